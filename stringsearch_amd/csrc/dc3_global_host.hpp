// dc3_global_host.hpp — host driver of the GLOBAL multi-GPU mode: ONE suffix array of a text whose bytes are sharded
// over P ranks (sacapart-style blocks, crates/sacapart/src/lib.rs:43-46), result sharded by suffix rank.  Included at
// the end of dc3hip.hip (same translation unit: it reuses the single-device building blocks above).
//
// The reference has no counterpart: its PartitionedSuffixArray keeps P independent local arrays and stitches at search
// time (sacapart/src/lib.rs:5-25, 69-97).  This is the other semantics SURVEY.md §8(e) names: the true SA[0..n) of the
// whole text, bit-identical to a single-device build (and so to divsufsort), built by the level structure of
// crates/dc3/src/lib.rs:44-193 with two rules:
//   1. replicate what is read at random — the level's string S and the sample ranks rank12, 4 B per element — into
//      every rank's HBM; then every rank can evaluate any key of the level from local memory, so
//   2. work is split by KEY RANGE, not by position: a rank streams over all positions, keeps the records whose key
//      falls into its range (splitters from a deterministic sample every rank computes identically), and sorts,
//      names and merges only those.  No (key, pos) record ever crosses xGMI.
// What does cross xGMI is the RANK EXCHANGE (rank_exchange below): (slot, name) / (position, rank) pairs go to the owner
// of their destination block (all-to-all), each owner builds its block with the windowed inversion, and the 4-byte
// blocks are all-gathered.  Transport behind a small interface (GComm): RCCL (ncclSend/ncclRecv groups over xGMI, one
// process per GPU) or an in-process loopback (P rank contexts on one device, hipMemcpyAsync as the wire) that exists so
// that P in {2,4,8} is parity-tested bit-exactly on a single-GPU box.
#pragma once

#include <condition_variable>
#include <memory>
#include <functional>
#include <mutex>
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the entry points are resolved with dlopen/dlsym at first use

// ---------------------------------------------------------------------------------------------
// transport
// ---------------------------------------------------------------------------------------------
constexpr double kXgmiLinkGBps = 153.0;    // one xGMI link of an MI355X (7 per GPU, point to point): the rate the link model prices with
struct GComm {
  int rank = 0, nranks = 1;
  double comm_ms = 0;                 // host wall time inside collectives (they are synchronous)
  uint64_t bytes_out = 0, bytes_in = 0;   // payload that left / reached this rank (self copies excluded)
  // prediction for P real GPUs from a run whose ranks may share one (dc3hip_gstats.work_ms / link_ms / collectives)
  double work_ms = 0, link_ms = 0; uint64_t ncoll = 0;
  std::chrono::steady_clock::time_point work_t0;
  bool working = false;
  // a rank starts / stops working on its device (a build's begin and end, and around every collective).  Loopback ranks on
  // one device pass a token (LoopComm): then work_ms is the rank's own work even though the ranks time-share the GPU.
  virtual void device_enter() { work_t0 = std::chrono::steady_clock::now(); working = true; }
  virtual void device_leave() {
    if (working) work_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - work_t0).count();
    working = false;
  }
  // one collective in which this rank exchanges at most `peer_bytes` with a single peer: P - 1 links work at once
  void note_link(size_t peer_bytes) { ncoll++; link_ms += (double)peer_bytes / (kXgmiLinkGBps * 1e9) * 1e3; }
  // GB/s of one link of THIS transport, for the policies that weigh recomputation against routing (0: the ranks share a device)
  virtual double link_GBps() const { return kXgmiLinkGBps; }
  virtual ~GComm() {}
  // every rank sends send[soff[r] .. +sbytes[r]) to rank r and receives rbytes[r] bytes from rank r at recv + roff[r]
  virtual int all_to_all_v(const void *send, const size_t *soff, const size_t *sbytes, void *recv, const size_t *roff,
                           const size_t *rbytes, hipStream_t st) = 0;
  // every rank contributes sbytes bytes; block r lands at recv + roff[r] (rbytes[r] bytes) on every rank.
  // send may be recv + roff[rank] (in place).
  virtual int all_gather_v(const void *send, size_t sbytes, void *recv, const size_t *roff, const size_t *rbytes,
                           hipStream_t st) = 0;
  // small host values: out[r*bytes ..] = rank r's in[0..bytes)
  virtual int all_gather_host(const void *in, void *out, size_t bytes) = 0;
  virtual int transport_ranks() { return nranks; }   // ranks the transport itself reports (RCCL: ncclCommCount)
  virtual void abort_all() {}
  virtual void leave_failed() {}      // this rank returns from a failed collective (loopback: see LoopWorld::leave)
  virtual void reset_all() {}          // before a new collective build of the whole group (no rank inside a collective)
  virtual const char *name() const = 0;
};
struct CommTimer {
  GComm *g; std::chrono::steady_clock::time_point t0; bool was_working;
  explicit CommTimer(GComm *gc) : g(gc), t0(std::chrono::steady_clock::now()), was_working(gc->working) { if (was_working) g->device_leave(); }
  ~CommTimer() {
    g->comm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (was_working) g->device_enter();
  }
};
static size_t max_peer_bytes(const size_t *a, const size_t *b, int P, int me) {
  size_t m = 0;
  for (int r = 0; r < P; r++) if (r != me) { if (a) m = std::max(m, a[r]); if (b) m = std::max(m, b[r]); }
  return m;
}

// ---- loopback: P ranks = P host threads of one process, each with its own context/stream on the same device -------
struct LoopWorld {
  int P;
  std::mutex mu; std::condition_variable cv;
  int arrived = 0; uint64_t gen = 0; bool failed = false;
  // DC3HIP_DEBUG=global_device_token: ranks that share one device work one at a time (the token is handed over inside
  // collectives) — a rank's work_ms is then its own work, the figure a prediction for P GPUs needs, at the price of the
  // overlap the ranks' streams otherwise find on the shared GPU.  Nothing depends on it but the timing.
  // one_device = all ranks of the group sit on the same device.
  std::mutex dev_mu; bool one_device = false, token = false;
  struct Post { const void *send; const size_t *soff; const size_t *sbytes; size_t one; };
  std::vector<Post> post;
  explicit LoopWorld(int p) : P(p), post((size_t)p) {}
  bool barrier() {                    // false once any rank has failed (nobody is left waiting for it)
    std::unique_lock<std::mutex> lk(mu);
    if (failed) return false;
    const uint64_t g = gen;
    if (++arrived == P) { arrived = 0; gen++; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return gen != g || failed; });
    return !failed;
  }
  void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
  void reset() { std::lock_guard<std::mutex> lk(mu); failed = false; arrived = 0; left_mask = 0; }
  // A rank returns from a failed collective entry point (build / sufcheck).  Once every rank has left it, the world is
  // clean again: the next collective of the group works whichever entry point it comes through.
  uint32_t left_mask = 0;
  void leave(int rank) {
    std::lock_guard<std::mutex> lk(mu);
    if (!failed) return;
    left_mask |= 1u << rank;
    if (left_mask == (P >= 32 ? 0xffffffffu : (1u << P) - 1u)) { failed = false; arrived = 0; left_mask = 0; }
  }
};
struct LoopComm : GComm {
  std::shared_ptr<LoopWorld> w;
  double model_link_GBps = 0;        // DC3HIP_DEBUG=global_link_gbps=.. (tests): the link rate the policies see (0: ranks share a device)
  void device_enter() override { if (w->token) w->dev_mu.lock(); GComm::device_enter(); }
  void device_leave() override { const bool had = working; GComm::device_leave(); if (had && w->token) w->dev_mu.unlock(); }
  double link_GBps() const override { return model_link_GBps > 0 ? model_link_GBps : (w->one_device ? 0.0 : kXgmiLinkGBps); }
  const char *name() const override { return "loopback (in-process, hipMemcpyAsync; peer copies between devices)"; }
  void abort_all() override { w->fail(); }
  void reset_all() override { w->reset(); }
  void leave_failed() override { w->leave(rank); }
  int sync_fail() { set_err("loopback transport: another rank failed"); return E_HIP; }
  int all_to_all_v(const void *send, const size_t *soff, const size_t *sbytes, void *recv, const size_t *roff,
                   const size_t *rbytes, hipStream_t st) override {
    HIPC(hipStreamSynchronize(st));                      // my outgoing bytes are complete
    CommTimer tm(this);
    note_link(max_peer_bytes(sbytes, rbytes, nranks, rank));
    w->post[(size_t)rank] = LoopWorld::Post{send, soff, sbytes, 0};
    if (!w->barrier()) return sync_fail();
    for (int r = 0; r < nranks; r++) {
      const LoopWorld::Post &p = w->post[(size_t)r];
      if (p.sbytes[rank] != rbytes[r]) { set_err("loopback all_to_all_v: rank %d sends %zu bytes, rank %d expects %zu", r, p.sbytes[rank], rank, rbytes[r]); w->fail(); return E_HIP; }
      if (rbytes[r] == 0) continue;
      HIPC(hipMemcpyAsync(static_cast<char *>(recv) + roff[r], static_cast<const char *>(p.send) + p.soff[rank], rbytes[r],
                          hipMemcpyDefault, st));                 // (peer copy when the ranks sit on different devices)
      if (r != rank) { bytes_in += rbytes[r]; bytes_out += sbytes[r]; }
    }
    HIPC(hipStreamSynchronize(st));
    if (!w->barrier()) return sync_fail();               // senders may reuse their buffers
    return E_OK;
  }
  int all_gather_v(const void *send, size_t sbytes, void *recv, const size_t *roff, const size_t *rbytes,
                   hipStream_t st) override {
    HIPC(hipStreamSynchronize(st));
    CommTimer tm(this);
    note_link(std::max(nranks > 1 ? sbytes : (size_t)0, max_peer_bytes(rbytes, nullptr, nranks, rank)));
    w->post[(size_t)rank] = LoopWorld::Post{send, nullptr, nullptr, sbytes};
    if (!w->barrier()) return sync_fail();
    for (int r = 0; r < nranks; r++) {
      const LoopWorld::Post &p = w->post[(size_t)r];
      if (p.one != rbytes[r]) { set_err("loopback all_gather_v: rank %d contributes %zu bytes, expected %zu", r, p.one, rbytes[r]); w->fail(); return E_HIP; }
      char *dst = static_cast<char *>(recv) + roff[r];
      if (rbytes[r] == 0 || dst == p.send) continue;     // in place
      HIPC(hipMemcpyAsync(dst, p.send, rbytes[r], hipMemcpyDefault, st));
      if (r != rank) { bytes_in += rbytes[r]; bytes_out += sbytes; }
    }
    HIPC(hipStreamSynchronize(st));
    if (!w->barrier()) return sync_fail();
    return E_OK;
  }
  int all_gather_host(const void *in, void *out, size_t bytes) override {
    CommTimer tm(this);
    w->post[(size_t)rank] = LoopWorld::Post{in, nullptr, nullptr, bytes};
    if (!w->barrier()) return sync_fail();
    for (int r = 0; r < nranks; r++) memcpy(static_cast<char *>(out) + (size_t)r * bytes, w->post[(size_t)r].send, bytes);
    if (!w->barrier()) return sync_fail();
    return E_OK;
  }
};

// ---- RCCL: one process per GPU, grouped ncclSend/ncclRecv over xGMI --------------------------------------------------
struct RcclApi {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;          // optional (self-test: the rank count RCCL itself reports)
  bool preloaded = false;                                // the host program had an RCCL mapped already and this is it
  bool load() {
    if (h) return true;
    // A process that already carries an RCCL (PyTorch maps its own torch/lib/librccl.so, soname librccl.so.1) must not get a
    // second instance: look for a mapped one first (RTLD_NOLOAD matches by soname), load one only when there is none.
    // dc3hip_rccl_library_path() reports which file the entry points came from; bench.py checks it against /proc/self/maps.
    for (const char *nm : {"librccl.so.1", "librccl.so"}) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (h) { preloaded = true; break; } }
    if (!h) {
      // None mapped yet.  Prefer the librccl that sits NEXT TO the HIP runtime this process runs on: a PyTorch wheel ships
      // its own libamdhip64 and a librccl built against it (and maps the latter only when torch.distributed is first
      // used) — the system's librccl on top of the wheel's runtime aborts at exit (double free, seen with ROCm 7.2's
      // librccl under a ROCm 7.0 wheel).
      Dl_info di;
      if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
          dir.resize(slash);
          for (const char *nm : {"/librccl.so", "/librccl.so.1"}) { h = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        }
      }
    }
    if (!h)
      for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { set_err("RCCL not found (dlopen librccl.so): %s", dlerror()); return false; }
#define DC3_RCCL_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(h, #sym)); if (!field) { set_err("RCCL symbol %s missing", #sym); h = nullptr; return false; }
    DC3_RCCL_SYM(GetUniqueId, ncclGetUniqueId) DC3_RCCL_SYM(CommInitRank, ncclCommInitRank) DC3_RCCL_SYM(CommDestroy, ncclCommDestroy)
    DC3_RCCL_SYM(GroupStart, ncclGroupStart) DC3_RCCL_SYM(GroupEnd, ncclGroupEnd) DC3_RCCL_SYM(Send, ncclSend) DC3_RCCL_SYM(Recv, ncclRecv)
    DC3_RCCL_SYM(AllGather, ncclAllGather) DC3_RCCL_SYM(GetErrorString, ncclGetErrorString)
#undef DC3_RCCL_SYM
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(h, "ncclCommCount"));
    return true;
  }
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;
#define NCCLC(expr)                                                                                              \
  do {                                                                                                           \
    ncclResult_t r__ = (expr);                                                                                   \
    if (r__ != ncclSuccess) { set_err("RCCL error %d (%s) at %s:%d: %s", (int)r__, g_rccl.GetErrorString(r__), __FILE__, __LINE__, #expr); return E_HIP; } \
  } while (0)
struct RcclComm : GComm {
  ncclComm_t comm = nullptr;
  unsigned char *d_small = nullptr;   // [ (P + 1) * kSmall ]
  static constexpr size_t kSmall = 1024;
  const char *name() const override { return "RCCL (grouped ncclSend/ncclRecv over xGMI)"; }
  int transport_ranks() override {
    int cnt = -1;
    if (g_rccl.CommCount && comm && g_rccl.CommCount(comm, &cnt) == ncclSuccess) return cnt;
    return -1;
  }
  ~RcclComm() override {
    if (comm) (void)g_rccl.CommDestroy(comm);
    if (d_small) (void)hipFree(d_small);
  }
  int all_to_all_v(const void *send, const size_t *soff, const size_t *sbytes, void *recv, const size_t *roff,
                   const size_t *rbytes, hipStream_t st) override {
    CommTimer tm(this);
    note_link(max_peer_bytes(sbytes, rbytes, nranks, rank));
    NCCLC(g_rccl.GroupStart());
    for (int r = 0; r < nranks; r++) {
      if (r == rank) continue;
      if (sbytes[r]) NCCLC(g_rccl.Send(static_cast<const char *>(send) + soff[r], sbytes[r], ncclUint8, r, comm, st));
      if (rbytes[r]) NCCLC(g_rccl.Recv(static_cast<char *>(recv) + roff[r], rbytes[r], ncclUint8, r, comm, st));
      bytes_out += sbytes[r]; bytes_in += rbytes[r];
    }
    NCCLC(g_rccl.GroupEnd());
    if (rbytes[rank])
      HIPC(hipMemcpyAsync(static_cast<char *>(recv) + roff[rank], static_cast<const char *>(send) + soff[rank], rbytes[rank],
                          hipMemcpyDeviceToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return E_OK;
  }
  int all_gather_v(const void *send, size_t sbytes, void *recv, const size_t *roff, const size_t *rbytes,
                   hipStream_t st) override {
    CommTimer tm(this);
    note_link(std::max(nranks > 1 ? sbytes : (size_t)0, max_peer_bytes(rbytes, nullptr, nranks, rank)));
    NCCLC(g_rccl.GroupStart());
    for (int r = 0; r < nranks; r++) {
      if (r == rank) continue;
      if (sbytes) NCCLC(g_rccl.Send(send, sbytes, ncclUint8, r, comm, st));
      if (rbytes[r]) NCCLC(g_rccl.Recv(static_cast<char *>(recv) + roff[r], rbytes[r], ncclUint8, r, comm, st));
      bytes_out += sbytes; bytes_in += rbytes[r];
    }
    NCCLC(g_rccl.GroupEnd());
    char *self = static_cast<char *>(recv) + roff[rank];
    if (sbytes && self != send) HIPC(hipMemcpyAsync(self, send, sbytes, hipMemcpyDeviceToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return E_OK;
  }
  int all_gather_host(const void *in, void *out, size_t bytes) override {
    CommTimer tm(this);
    if (bytes > kSmall) { set_err("all_gather_host: %zu bytes per rank exceed the staging buffer", bytes); return E_ARGS; }
    unsigned char *din = d_small, *dout = d_small + kSmall;
    HIPC(hipMemcpy(din, in, bytes, hipMemcpyHostToDevice));
    NCCLC(g_rccl.AllGather(din, dout, bytes, ncclUint8, comm, nullptr));
    HIPC(hipStreamSynchronize(nullptr));
    HIPC(hipMemcpy(out, dout, bytes * (size_t)nranks, hipMemcpyDeviceToHost));
    return E_OK;
  }
};

// ---- host-staged: the caller supplies the two collectives on HOST buffers (MPI, torch.distributed/gloo, ...); the
// library stages device data through pinned memory.  For nodes without peer access and for multi-process tests on
// a single GPU (RCCL refuses two ranks on one device).
struct HostComm : GComm {
  dc3hip_host_transport t;
  char *hs = nullptr, *hr = nullptr; size_t cap_s = 0, cap_r = 0;     // pinned staging
  const char *name() const override { return "host-staged (caller's collectives on pinned host buffers)"; }
  double link_GBps() const override { return 25.0; }      // staged through host memory: PCIe-class, not xGMI
  ~HostComm() override { if (hs) (void)hipHostFree(hs); if (hr) (void)hipHostFree(hr); }
  int grow(char **p, size_t *cap, size_t need) {
    if (need <= *cap) return E_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = std::max<size_t>(need + need / 4, 1u << 20);
    HIPC(hipHostMalloc(reinterpret_cast<void **>(p), want, hipHostMallocDefault));
    *cap = want;
    return E_OK;
  }
  int all_to_all_v(const void *send, const size_t *soff, const size_t *sbytes, void *recv, const size_t *roff,
                   const size_t *rbytes, hipStream_t st) override {
    CommTimer tm(this);
    note_link(max_peer_bytes(sbytes, rbytes, nranks, rank));
    uint64_t so[kMaxRanks], sb[kMaxRanks], ro[kMaxRanks], rb[kMaxRanks];
    size_t send_hi = 0, recv_hi = 0;
    for (int r = 0; r < nranks; r++) {
      so[r] = soff[r]; sb[r] = sbytes[r]; ro[r] = roff[r]; rb[r] = rbytes[r];
      send_hi = std::max(send_hi, soff[r] + sbytes[r]); recv_hi = std::max(recv_hi, roff[r] + rbytes[r]);
      if (r != rank) { bytes_out += sbytes[r]; bytes_in += rbytes[r]; }
    }
    RC(grow(&hs, &cap_s, send_hi)); RC(grow(&hr, &cap_r, recv_hi));
    if (send_hi) HIPC(hipMemcpyAsync(hs, send, send_hi, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (t.all_to_all_v(t.user, hs, so, sb, hr, ro, rb) != 0) { set_err("host transport: all_to_all_v failed"); return E_HIP; }
    if (recv_hi) HIPC(hipMemcpyAsync(recv, hr, recv_hi, hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return E_OK;
  }
  int all_gather_v(const void *send, size_t sbytes, void *recv, const size_t *roff, const size_t *rbytes,
                   hipStream_t st) override {
    CommTimer tm(this);
    note_link(std::max(nranks > 1 ? sbytes : (size_t)0, max_peer_bytes(rbytes, nullptr, nranks, rank)));
    uint64_t ro[kMaxRanks], rb[kMaxRanks];
    size_t recv_hi = 0;
    for (int r = 0; r < nranks; r++) {
      ro[r] = roff[r]; rb[r] = rbytes[r]; recv_hi = std::max(recv_hi, roff[r] + rbytes[r]);
      if (r != rank) { bytes_out += sbytes; bytes_in += rbytes[r]; }
    }
    RC(grow(&hs, &cap_s, sbytes)); RC(grow(&hr, &cap_r, recv_hi));
    if (sbytes) HIPC(hipMemcpyAsync(hs, send, sbytes, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (t.all_gather_v(t.user, hs, sbytes, hr, ro, rb) != 0) { set_err("host transport: all_gather_v failed"); return E_HIP; }
    // every block but my own (already in place, and possibly aliased by `send`)
    for (int r = 0; r < nranks; r++)
      if (r != rank && rbytes[r]) HIPC(hipMemcpyAsync(static_cast<char *>(recv) + roff[r], hr + roff[r], rbytes[r], hipMemcpyHostToDevice, st));
    char *self = static_cast<char *>(recv) + roff[rank];
    if (sbytes && self != send) HIPC(hipMemcpyAsync(self, send, sbytes, hipMemcpyDeviceToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return E_OK;
  }
  int all_gather_host(const void *in, void *out, size_t bytes) override {
    CommTimer tm(this);
    uint64_t ro[kMaxRanks], rb[kMaxRanks];
    for (int r = 0; r < nranks; r++) { ro[r] = (uint64_t)r * bytes; rb[r] = bytes; }
    if (t.all_gather_v(t.user, in, bytes, out, ro, rb) != 0) { set_err("host transport: all_gather_v failed"); return E_HIP; }
    return E_OK;
  }
};

// ---------------------------------------------------------------------------------------------
// one rank of a global build
// ---------------------------------------------------------------------------------------------
struct dc3hip_gctx {
  bool no_wide_msd = false;    // DC3HIP_NO_WIDE_MSD=1: wide mode always sorts 16-byte records with the LSD passes
  u64 wide_msd_min = 1ull << 22; // DC3HIP_WIDE_MSD_MIN (tests): fewest positions per rank for the wide bucket ordering
  bool wide_msd_forced = false;  // ... given explicitly: texts below 2^32 take the unrouted order at every rank count
  u32 w_depth = 0;             // wide mode: symbols the last tie pass of the last build compared (the verifier compares at least as deep)
  bool no_select = false;      // DC3HIP_GLOBAL_NO_SELECT=1 (tests): no selecting partition pass (MsdPass1KeysSel); the routed / scanned forms as before
  bool route = true;           // DC3HIP_GLOBAL_NO_ROUTE=1: every rank evaluates all positions and keeps its key range (the round-2 form)
  dc3hip_ctx *c = nullptr;
  GComm *comm = nullptr;
  int64_t max_total = 0, total_n = 0;
  bool text_set = false, built = false;
  int64_t shard_first = 0, shard_count = 0;     // this rank holds SA[shard_first .. shard_first + shard_count)
  const u32 *shard_ptr = nullptr;               // device
  u32 local_max = 1u << 22;                     // levels up to this length are finished on every rank redundantly
  bool no_text_order = false;
  bool force_dist = false;                      // run the distributed path even with one rank (transport tests)
  // wide mode (texts beyond DC3HIP_MAX_N, or DC3HIP_GLOBAL_FORCE_WIDE=1): 64-bit positions, whole-text order only
  bool wide = false;
  uint8_t *w_text = nullptr;                    // max_total + 64 bytes (the context's own text buffer is not used)
  // wide mode: records of this rank's image range (w_ra / w_rb: pack / partition / sort buffers) and its shard of 64-bit
  // positions; every array with its own capacity (records / words)
  Rec16 *w_ra = nullptr, *w_rb = nullptr; u64 *w_shard = nullptr;
  uint8_t *w_same = nullptr;            // one byte per word of the bucket ordering: same image as the word before
  size_t w_cap_a = 0, w_cap_b = 0, w_cap_s = 0, w_cap_same = 0;
  // wide mode, deepening by rank look-ups (wide_deepen): the whole order, its equal-window flags and its inverse on every rank
  // (w_sa_all: one rank's shard at a time while the inverse is built — the whole order is never held, round 5)
  u64 *w_sa_all = nullptr, *w_isa = nullptr; uint8_t *w_eq_all = nullptr, *w_eq2 = nullptr;
  size_t w_cap_sa = 0, w_cap_isa = 0, w_cap_eq = 0, w_cap_eq2 = 0;
  // groups beyond kWideTieBig members (a run of one symbol, a short period): group starts of the shard (w_aux, 4 bytes per
  // entry) and the compacted members with their sort records (w_aux2, 48 bytes per member of such a group)
  unsigned char *w_aux = nullptr, *w_aux2 = nullptr;
  size_t w_cap_aux = 0, w_cap_aux2 = 0;
  bool w_isa_valid = false;             // the last build ended with w_isa = the exact inverse of the order (the verifier uses it)
  bool no_wide_deepen = false;          // DC3HIP_NO_WIDE_DEEPEN=1 (tests): windows that repeat beyond the symbol compares' budget are refused, as before round 4
  dc3hip_gstats gs;
  char err[512] = "";
  std::vector<dc3hip_gctx *> group;             // loopback: all ranks of the group (rank 0 owns the list)
};

// SELECT (every rank walks all positions of the replicated string and keeps its key range: nothing is routed) or ROUTE (every
// rank packs its own block and sends each 8-byte record to its owner) — by the per-rank cost of the two forms on P GPUs, in
// ms per GiB of the level's string (MI355X, profiles/r04*): the selecting count + partition pass 1 walk ALL positions,
// 4.46; packing and partitioning a rank's own block costs 4.5 / P, and the all-to-all puts 8 / P^2 bytes per position on each
// link.  On xGMI (153 GB/s per link) that is select up to 4 ranks and route beyond (8 ranks: 4.46 against 0.56 + 0.88);
// ranks that share one device (loopback) have no link to pay and select.  Both forms give the same array and are tested.
static bool gselect_pays(const dc3hip_gctx *G, int P) {
  const double link = G->comm->link_GBps();
  if (link <= 0 || P <= 1) return true;
  const double walk = 4.46, pack = 4.5;
  const double xfer = 8.0 * 1073741824.0 / (link * 1e9) * 1e3;       // ms for 8 bytes per position of one GiB over one link
  return walk <= pack / P + xfer / ((double)P * P);
}

static void block_of(int64_t n, int P, int r, int64_t *off, int64_t *len) {
  const int64_t S = n / P + 1;                  // sacapart/src/lib.rs:43
  const int64_t o = std::min<int64_t>(n, (int64_t)r * S);
  *off = o; *len = std::min<int64_t>(S, n - o);
}

// counts of `bytes` per rank -> this rank's prefix and the total
static int gather_counts(GComm *cm, uint64_t mine, uint64_t *prefix, uint64_t *total, uint64_t *all = nullptr) {
  uint64_t buf[kMaxRanks];
  RC(cm->all_gather_host(&mine, buf, sizeof(uint64_t)));
  uint64_t pre = 0, tot = 0;
  for (int r = 0; r < cm->nranks; r++) { if (r < cm->rank) pre += buf[r]; tot += buf[r]; if (all) all[r] = buf[r]; }
  *prefix = pre; *total = tot;
  return E_OK;
}

// order-preserving selection; *out is allocated from the arena.  One evaluation of the selector per item when the
// arena has room for the chunk-local staging array (k_sel_stage / scan / k_sel_copy), else count / scan / write.
template <class Sel>
static int select_records(dc3hip_ctx *c, const Sel &sel, u32 nitems, typename Sel::Out **out, u32 *count, int phase) {
  typedef typename Sel::Out Out;
  const Chunking ck = make_chunks(c, nitems, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  const size_t stage_bytes = align_up((size_t)nitems * sizeof(Out), 256);
  // staged form needs the staging array ABOVE the result (it is released afterwards), so the result is placed first with
  // its worst-case size only when that is affordable; otherwise the two-evaluation form
  const bool staged = c->arena_bytes - c->arena_off >= 2 * stage_bytes + (64u << 20);
  if (staged) {
    Out *res = nullptr, *stage = nullptr;
    RC(arena_alloc(c, (size_t)nitems + 16, &res));         // shrunk to the real count below
    const ArenaMark mk_stage = arena_mark(c);
    RC(arena_alloc(c, (size_t)nitems, &stage));
    {
      PhaseScope ps(c, phase, nitems);
      hipLaunchKernelGGL((k_sel_stage<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, stage, counts);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 32);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 32, c->d_words + 32, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    *count = c->h_words[32];
    if (*count) {
      PhaseScope ps(c, phase, *count);
      hipLaunchKernelGGL((k_sel_copy<Out>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, stage, ck.chunk, ck.nchunks, counts,
                         *count, res);
      KCHECK();
    }
    // give back the staging array and the unused tail of the result
    arena_release(c, mk_stage);
    c->arena_off = (size_t)(reinterpret_cast<unsigned char *>(res) - c->arena) + align_up(((size_t)*count + 16) * sizeof(Out), 256);
    *out = res;
    return E_OK;
  }
  {
    PhaseScope ps(c, phase, nitems);
    hipLaunchKernelGGL((k_sel_count<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 32);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 32, c->d_words + 32, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  *count = c->h_words[32];
  RC(arena_alloc(c, (size_t)*count + 16, out));
  if (*count) {
    PhaseScope ps(c, phase, nitems);
    hipLaunchKernelGGL((k_sel_write<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, counts, *out);
    KCHECK();
  }
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// RANK EXCHANGE: every rank holds `cnt` (destination, value) pairs; all destinations together are a bijection onto
// [0, M).  On return out[0..M) is complete on EVERY rank.
//   1. local partition of the pairs by destination digit (<= 256 digits of >= 2^14 destinations; rank h owns a
//      contiguous digit range) — one stable radix pass, its digit table gives the send offsets;
//   2. all-to-all: pairs to the owner of their destination            (8 B x cnt x (P-1)/P per rank over xGMI)
//   3. the owner builds its block by the windowed inversion (inverse_permute)
//   4. all-gather of the blocks                                        (4 B x M x (P-1)/P per rank over xGMI)
// ---------------------------------------------------------------------------------------------
static int rank_exchange(dc3hip_gctx *G, Rec8 *pairs, u32 cnt, u32 M, u32 *out, int phase) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const ArenaMark mk = arena_mark(c);
  u32 sh = (u32)kInvWindowBits;
  while ((((u64)M - 1) >> sh) + 1 > 256) sh++;
  const u32 nd = (u32)((((u64)M - 1) >> sh) + 1);
  auto dlo = [&](int h) { return (u32)(((u64)h * nd + P - 1) / P); };       // first digit of rank h
  auto dest_lo = [&](int h) { return (u64)std::min<u64>((u64)M, (u64)dlo(h) << sh); };
  // 1. partition by digit
  u32 hdb[257];
  Rec8 *sorted = pairs;
  for (u32 d = 0; d <= 256; d++) hdb[d] = 0;
  if (cnt) {
    constexpr int kTile = SortCfg<Rec8, 256>::NW * 64 * SortCfg<Rec8, 256>::IPT;
    const Chunking ck = make_chunks(c, cnt, kTile);
    u32 *table = nullptr, *digit_base = nullptr;
    Rec8 *pb = nullptr;
    RC(arena_alloc(c, (size_t)256 * ck.nchunks, &table));
    RC(arena_alloc(c, (size_t)256, &digit_base));
    RC(arena_alloc(c, (size_t)cnt, &pb));
    KeyDig dig; dig.shift = 32 + sh; dig.mask = 255;
    {
      PhaseScope ps(c, phase, cnt);
      hipLaunchKernelGGL((k_rs_upsweep<Rec8, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, pairs, cnt, ck.chunk,
                         ck.nchunks, dig, table);
      KCHECK();
    }
    RC(scan_digit_table(c, table, ck.nchunks, digit_base, 256, phase));
    void *tmpp = nullptr;
    RC(stage_d2h_async(c, digit_base, 256 * sizeof(u32), &tmpp));
    const u32 *tmp = static_cast<const u32 *>(tmpp);
    ArrayLoader<Rec8> ld; ld.p = pairs;
    RC((launch_downsweep<Rec8, 256, ArrayLoader<Rec8>>(c, ld, pb, cnt, ck, dig, table, digit_base, phase)));
    HIPC(hipStreamSynchronize(c->stream));
    for (u32 d = 0; d < 256; d++) hdb[d] = tmp[d];
    hdb[256] = cnt;
    for (u32 d = nd; d < 256; d++) hdb[d] = cnt;
    sorted = pb;
  }
  // 2. all-to-all
  size_t soff[kMaxRanks], sbytes[kMaxRanks], roff[kMaxRanks], rbytes[kMaxRanks];
  uint64_t scount[kMaxRanks], mat[kMaxRanks * kMaxRanks];
  for (int h = 0; h < P; h++) {
    const u32 a = hdb[std::min<u32>(dlo(h), 256)], b = hdb[std::min<u32>(dlo(h + 1), 256)];
    soff[h] = (size_t)a * sizeof(Rec8); sbytes[h] = (size_t)(b - a) * sizeof(Rec8); scount[h] = b - a;
  }
  RC(cm->all_gather_host(scount, mat, sizeof(uint64_t) * (size_t)P));
  const u64 base = dest_lo(me), myblk = dest_lo(me + 1) - base;
  u64 got = 0;
  for (int r = 0; r < P; r++) { roff[r] = (size_t)got * sizeof(Rec8); rbytes[r] = (size_t)mat[(size_t)r * P + me] * sizeof(Rec8); got += mat[(size_t)r * P + me]; }
  if (got != myblk) { set_err("rank exchange: block of rank %d expects %llu pairs, received %llu (destinations are not a bijection)", me, (unsigned long long)myblk, (unsigned long long)got); return E_HIP; }
  Rec8 *rb = nullptr, *rt = nullptr;
  RC(arena_alloc(c, (size_t)myblk + 16, &rb));
  RC(arena_alloc(c, (size_t)myblk + 16, &rt));
  RC(cm->all_to_all_v(sorted, soff, sbytes, rb, roff, rbytes, c->stream));
  // 3. my block
  if (myblk) {
    if (base) {
      PhaseScope ps(c, phase, myblk);
      hipLaunchKernelGGL(k_rebase_keys, dim3(grid_for(c, myblk)), dim3(kBlock), 0, c->stream, rb, (u32)myblk, (u32)base);
      KCHECK();
    }
    RC(inverse_permute(c, rb, rt, (u32)myblk, out + base, phase));
  }
  // 4. all-gather of the blocks, in place
  size_t goff[kMaxRanks], gbytes[kMaxRanks];
  for (int r = 0; r < P; r++) { goff[r] = (size_t)dest_lo(r) * 4; gbytes[r] = (size_t)(dest_lo(r + 1) - dest_lo(r)) * 4; }
  RC(cm->all_gather_v(out + base, (size_t)myblk * 4, out, goff, gbytes, c->stream));
  G->gs.exchanges += 1;
  G->gs.exchange_pairs += cnt;
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// where a level's result goes
// ---------------------------------------------------------------------------------------------
enum GOut {
  G_TOP = 0,    // level 0: this rank's slice of the suffix array stays in c->d_sa (G->shard_*)
  G_RANK = 1,   // out[pos] = 1-based rank of suffix pos, complete on every rank (the parent's rank12)
  G_SA = 2      // out[k] = position of the k-th smallest suffix, complete on every rank (discarding parent)
};
// slice[0..cnt) = this rank's part of the level's suffix array, starting at global index `pre`
static int deliver(dc3hip_gctx *G, const u32 *slice, u32 cnt, u64 pre, const uint64_t *all, u32 m, u32 *out, GOut mode) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  if (mode == G_TOP) {
    G->shard_first = (int64_t)pre; G->shard_count = cnt; G->shard_ptr = slice;
    return E_OK;
  }
  if (mode == G_RANK) {
    const ArenaMark mk = arena_mark(c);
    Rec8 *pp = nullptr;
    RC(arena_alloc(c, (size_t)cnt + 16, &pp));
    if (cnt) {
      PhaseScope ps(c, DC3HIP_PH_RANKS, cnt);
      hipLaunchKernelGGL(k_sa_to_pairs, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, slice, cnt, (u32)pre, pp);
      KCHECK();
    }
    RC(rank_exchange(G, pp, cnt, m, out, DC3HIP_PH_RANKS));     // rank[pos] = global index + 1, everywhere
    arena_release(c, mk);
    return E_OK;
  }
  size_t roff[kMaxRanks], rbytes[kMaxRanks];
  u64 o = 0;
  for (int r = 0; r < cm->nranks; r++) { roff[r] = (size_t)o * 4; rbytes[r] = (size_t)all[r] * 4; o += all[r]; }
  return cm->all_gather_v(slice, (size_t)cnt * 4, out, roff, rbytes, c->stream);
}

// splitters of a 64-bit image order from ns sampled records ((image << pbits) | pos): every rank computes the same
static int image_splitters(dc3hip_ctx *c, const Rec8 *d_sample, u32 ns, u32 pbits, int P, int me, u64 *lo, u64 *hi) {
  void *hsp = nullptr;
  RC(stage_d2h(c, d_sample, (size_t)ns * sizeof(Rec8), &hsp));
  const Rec8 *hs = static_cast<const Rec8 *>(hsp);
  // a few thousand candidates per rank are plenty (a host sort of the whole 2^20-record predictor sample cost 60 ms)
  const u32 step = std::max<u32>(1, ns / (u32)(4096 * P));
  std::vector<u64> img;
  img.reserve(ns / step + 1);
  for (u32 i = 0; i < ns; i += step) img.push_back(((((u64)hs[i].key) << 32) | hs[i].val) >> pbits);
  std::sort(img.begin(), img.end());
  const size_t k = img.size();
  *lo = 0; *hi = ~0ull;
  if (me > 0) *lo = img[(size_t)((u64)me * k / P)];
  if (me + 1 < P) *hi = img[(size_t)((u64)(me + 1) * k / P)];
  return E_OK;
}

// Pass 1 of the bucket ordering that SELECTS (k_msd_part_keys<.., kSel>): the rank walks the replicated text / level
// string, makes every position's image (as the single device's pass 1 does) and partitions the words of its image
// range only.  Everything behind it — bucket sizes, pass 2, local order, tie pass — is the single device's code on a
// P-th of the words.  m = positions walked.
template <class KM>
struct MsdPass1KeysSel : MsdPass1Keys<KM> {
  MsdSel sel{0, 0, 1, 0}; u32 m = 0;
  // (the ordering gave up before pass 2: the plain words of the selection, in position order, for the LSD passes)
  int repack(dc3hip_ctx *c, Rec8 *out, u32 nrec, u32 **first_table) override {
    SelPosImageW<KM> s; s.km = this->km; s.hm = this->hm; s.sel = sel; s.pbits = this->hm.pbits + sel.sh;
    Rec8 *tmp = nullptr; u32 cnt = 0;
    RC(select_records(c, s, m, &tmp, &cnt, DC3HIP_PH_PACK));
    if (cnt != nrec) { set_err("internal: the selection repacked %u words of %u", cnt, nrec); return E_HIP; }
    HIPC(hipMemcpyAsync(out, tmp, (size_t)cnt * sizeof(Rec8), hipMemcpyDeviceToDevice, c->stream));
    *first_table = nullptr;
    return E_OK;
  }
  int launch(dc3hip_ctx *c, u64 *out, u32, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      attr_set[c->device & 15] = true;
    }
    if (this->strip && (this->hm.nbits + this->hm.pbits != 64 || g.d1 == 0)) { set_err("internal: a stripped image must fill the word"); return E_HIP; }
    if (this->strip)
      hipLaunchKernelGGL((k_msd_part_keys<KM, true, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, this->km, this->hm,
                         this->P1, out, m, base, sh1, g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon, sel);
    else
      hipLaunchKernelGGL((k_msd_part_keys<KM, false, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, this->km, this->hm,
                         this->P1, out, m, base, sh1, g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon, sel);
    KCHECK();
    return E_OK;
  }
};

// ---------------------------------------------------------------------------------------------
// Whole-level order, split by key range (the distributed form of order_all_positions): every rank orders the positions
// p in [0, m) whose key image falls into its range by prefix sort + tie refinement.  If every key on every rank is
// distinct the concatenated slices ARE the level's suffix array (suffixes differ inside the key: 9 bytes of text for
// Key9 at level 0, a K-S triple for Key3 below), *done = true and the result has been delivered; otherwise nothing
// was produced.  emit: where this rank's slice goes (c->d_sa at the top; nullptr = arena, valid until the caller's mark
// is released).
// ---------------------------------------------------------------------------------------------
template <class KM>
static int gorder_positions(dc3hip_gctx *G, KM km, u32 m, u32 kbits, const HiMap &hm, int depth, u32 *out, GOut mode,
                            bool *done) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  *done = false;
  const ArenaMark mk = arena_mark(c);
  // the slice first (worst case: every position in my range), so that the sort's temporaries can be released before
  // the result is delivered (the rank exchange needs the room)
  u32 *slice = (mode == G_TOP) ? c->d_sa : nullptr;
  if (!slice) RC(arena_alloc(c, (size_t)m + 16, &slice));
  const ArenaMark mk_tmp = arena_mark(c);
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
  u32 nrec = 0;
  u64 img_lo = 0, img_span = 0;
  // SELECTED (default up to 16 ranks, key makers whose image pass 1 of the bucket ordering can make itself): no records
  // are built or sent at all — see MsdPass1KeysSel.  A rank reads the m positions twice (count, partition) and orders
  // m / P words; on one GPU shared by P loopback ranks that is the least total work of the three forms, and on P GPUs
  // the walk (HBM rate) costs less than routing 8 m / P bytes over xGMI.
  constexpr bool kFusable = std::is_same<KM, Key9>::value || std::is_same<KM, Key3<SymU32>>::value;
  typename std::conditional<kFusable, MsdPass1KeysSel<KM>, MsdPass1Keys<KM>>::type psel;     // (the selecting kernels only where they are used)
  MsdGeom mgx;
  u32 *sel_table = nullptr;
  bool selected = false;
  if constexpr (kFusable) {
    const MsdGeom mg = msd_geometry(c, m, hm);
    if (!G->no_select && mg.on && gselect_pays(G, P)) {
      u64 lo = 0, hi = ~0ull;
      {
        u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
        const u32 stride = std::max<u32>(1, m / ns);
        ns = (m - 1) / stride + 1;
        Rec8 *smp = nullptr;
        RC(arena_alloc(c, (size_t)ns, &smp));
        hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
        KCHECK();
        RC(image_splitters(c, smp, ns, hm.pbits, P, me, &lo, &hi));
      }
      mgx = mg;
      psel.km = km; psel.hm = hm; psel.P1 = 0; psel.m = m;
      if (!c->no_pack_strip && hm.pbits >= 23 && !hm.exact && kbits >= hm.nbits + mg.d1) {      // (as order_all_positions)
        u64 limb = 0;
        if constexpr (std::is_same<KM, Key9>::value) limb = km.B3; else limb = km.B;
        psel.strip = true; psel.hm_plain = hm;
        psel.hm = make_himap(limb, kbits, m, hm.pbits - mg.d1);
        psel.hm.raw = hm.raw;
        mgx.ebits = psel.hm.nbits;
      }
      psel.sel = MsdSel{lo, hi, (me + 1 == P) ? 1u : 0u, psel.strip ? mg.d1 : 0u};
      RC(arena_alloc(c, (size_t)kMsdMaxDig * mg.ck.nchunks, &sel_table));
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, m);
        HIPC(hipMemsetAsync(c->d_words + 33, 0, sizeof(u32), c->stream));
        hipLaunchKernelGGL((k_msd_count_sel<KM>), dim3(mg.ck.nchunks), dim3(kBlock), 0, c->stream, km, psel.hm, 0ull, m, psel.sel, mg.ck.chunk,
                           mg.ck.nchunks, sel_table, psel.hm.nbits - mg.d1, c->d_words + 33);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 33, c->d_words + 33, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      nrec = c->h_words[33];
      RC(arena_alloc(c, (size_t)nrec + 16, &ha));            // (pass 1 makes the words: scratch of pass 2)
      selected = true;
      G->gs.select_p1 += 1;
    }
  }
  if (selected) {
  } else if (G->route && hm.nbits >= 8) {
    // ROUTED (default): every rank packs the records of ITS block of positions only (m / P of them), partitions them by
    // the top 8 image bits — rank h owns a contiguous digit range, chosen from a replicated sample so that the ranges hold
    // about m / P records each — and sends every record to its owner: one all-to-all of 8-byte records
    // (8 m (P-1) / P^2 bytes out per rank).  Work per rank is O(m / P); SURVEY.md §8(e) step 2.
    u32 dlo[kMaxRanks + 1];
    {
      u32 ns = (u32)std::min<u64>(m, (u64)4096 * P);
      const u32 stride = std::max<u32>(1, m / ns);
      ns = (m - 1) / stride + 1;
      Rec8 *smp = nullptr;
      RC(arena_alloc(c, (size_t)ns, &smp));
      hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
      KCHECK();
      void *hsp = nullptr;
      RC(stage_d2h(c, smp, (size_t)ns * sizeof(Rec8), &hsp));
      const Rec8 *hs = static_cast<const Rec8 *>(hsp);
      u32 cnt256[257] = {0};
      for (u32 i = 0; i < ns; i++) cnt256[(u32)((((((u64)hs[i].key) << 32) | hs[i].val) >> (hm.pbits + hm.nbits - 8)) & 255u)]++;
      // boundaries: rank h starts at the first digit whose prefix count reaches h * ns / P (identical on all ranks)
      dlo[0] = 0; dlo[P] = 256;
      u32 acc = 0, hnext = 1;
      for (u32 d = 0; d < 256 && hnext < (u32)P; d++) {
        while (hnext < (u32)P && (u64)acc * P >= (u64)hnext * ns) dlo[hnext++] = d;
        acc += cnt256[d];
      }
      while (hnext < (u32)P) dlo[hnext++] = 256;
      for (int r = 1; r <= P; r++) dlo[r] = std::max(dlo[r], dlo[r - 1]);
    }
    const u32 boff = (u32)((u64)m * me / P), blen = (u32)((u64)m * (me + 1) / P) - boff;
    Rec8 *mine = nullptr, *sorted = nullptr;
    RC(arena_alloc(c, (size_t)blen + 16, &mine));
    u32 hdb[257];
    for (u32 d = 0; d <= 256; d++) hdb[d] = 0;
    if (blen) {
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, blen);
        hipLaunchKernelGGL((k_pack_image_range<KM>), dim3(grid_for(c, blen)), dim3(kBlock), 0, c->stream, km, boff, blen, hm, mine);
        KCHECK();
      }
      constexpr int kTile = SortCfg<Rec8, 256>::NW * 64 * SortCfg<Rec8, 256>::IPT;
      const Chunking ck = make_chunks(c, blen, kTile);
      u32 *table = nullptr, *digit_base = nullptr;
      RC(arena_alloc(c, (size_t)256 * ck.nchunks, &table));
      RC(arena_alloc(c, (size_t)256, &digit_base));
      RC(arena_alloc(c, (size_t)blen + 16, &sorted));
      KeyDig dig; dig.shift = hm.pbits + hm.nbits - 8; dig.mask = 255;
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, blen);
        hipLaunchKernelGGL((k_rs_upsweep<Rec8, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, mine, blen, ck.chunk, ck.nchunks, dig, table);
        KCHECK();
      }
      RC(scan_digit_table(c, table, ck.nchunks, digit_base, 256, DC3HIP_PH_PACK));
      void *tmpp = nullptr;
      RC(stage_d2h_async(c, digit_base, 256 * sizeof(u32), &tmpp));
      const u32 *tmp = static_cast<const u32 *>(tmpp);
      ArrayLoader<Rec8> ld; ld.p = mine;
      RC((launch_downsweep<Rec8, 256, ArrayLoader<Rec8>>(c, ld, sorted, blen, ck, dig, table, digit_base, DC3HIP_PH_PACK)));
      HIPC(hipStreamSynchronize(c->stream));
      for (u32 d = 0; d < 256; d++) hdb[d] = tmp[d];
      hdb[256] = blen;
    }
    size_t soff[kMaxRanks], sbytes[kMaxRanks], roff[kMaxRanks], rbytes[kMaxRanks];
    uint64_t scount[kMaxRanks], mat[kMaxRanks * kMaxRanks];
    for (int r = 0; r < P; r++) {
      const u32 a0 = hdb[dlo[r]], b0 = hdb[dlo[r + 1]];
      soff[r] = (size_t)a0 * sizeof(Rec8); sbytes[r] = (size_t)(b0 - a0) * sizeof(Rec8); scount[r] = b0 - a0;
    }
    RC(cm->all_gather_host(scount, mat, sizeof(uint64_t) * (size_t)P));
    u64 got = 0;
    for (int r = 0; r < P; r++) { roff[r] = (size_t)got * sizeof(Rec8); rbytes[r] = (size_t)mat[(size_t)r * P + me] * sizeof(Rec8); got += mat[(size_t)r * P + me]; }
    if (got > (u64)m) { set_err("global order: %llu records routed to rank %d of a level of %u", (unsigned long long)got, me, m); return E_HIP; }
    nrec = (u32)got;
    RC(arena_alloc(c, (size_t)nrec + 16, &ha));
    RC(cm->all_to_all_v(sorted ? sorted : mine, soff, sbytes, ha, roff, rbytes, c->stream));
    G->gs.exchanges += 1;
    img_lo = (u64)dlo[me] << (hm.nbits - 8);
    img_span = (u64)(dlo[me + 1] - dlo[me]) << (hm.nbits - 8);
  } else {
    u64 lo = 0, hi = ~0ull;
    {
      u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
      const u32 stride = std::max<u32>(1, m / ns);
      ns = (m - 1) / stride + 1;
      Rec8 *smp = nullptr;
      RC(arena_alloc(c, (size_t)ns, &smp));
      hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
      KCHECK();
      RC(image_splitters(c, smp, ns, hm.pbits, P, me, &lo, &hi));
    }
    SelPosImage<KM> sel; sel.km = km; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
    RC(select_records(c, sel, m, &ha, &nrec, DC3HIP_PH_PACK));
  }
  RC(arena_alloc(c, (size_t)nrec + 16, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  bool ok = true, distinct = true;
  if (nrec && selected)
    RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &ok, depth, slice, 0, &distinct, sel_table, false, &mgx, 0, 0, &psel, nullptr, true)));
  else if (nrec)
    RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &ok, depth, slice, 0, &distinct, nullptr, false, nullptr,
                             img_lo, img_span, nullptr, nullptr, true)));      // (slots: the routed records fill their image range evenly)
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, (ok && distinct) ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != m) { set_err("global order: %llu of %u positions selected", (unsigned long long)tot, m); return E_HIP; }
  arena_release(c, mk_tmp);
  if (ngood == (uint64_t)P) {
    *done = true;
    RC(deliver(G, slice, nrec, pre, all, m, out, mode));
  }
  arena_release(c, mk);
  return E_OK;
}

// The whole-text order on 12-byte records (try_text_order12's distributed form; top level only): hm maps KM's image to
// hm.nbits <= 63 bits, sorted in full; the tie pass writes the slice.
template <class KM>
static int gorder_positions12(dc3hip_gctx *G, KM km, u32 m, u32 kbits, const HiMap &hm, bool *done) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  *done = false;
  const ArenaMark mk = arena_mark(c);
  u32 *slice = c->d_sa;
  u64 lo = 0, hi = ~0ull;
  {
    u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
    const u32 stride = std::max<u32>(1, m / ns);
    ns = (m - 1) / stride + 1;
    Rec8 *smp = nullptr;
    RC(arena_alloc(c, (size_t)ns, &smp));
    hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
    KCHECK();
    RC(image_splitters(c, smp, ns, 1u, P, me, &lo, &hi));
  }
  SelPosImage12<KM> sel; sel.km = km; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
  u32 nrec = 0;
  RC(select_records(c, sel, m, &ha, &nrec, DC3HIP_PH_PACK));
  RC(arena_alloc(c, (size_t)nrec + 16, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  bool ok = true, distinct = true;
  if (nrec) {
    RC(radix_sort<Rec12>(c, ha, hb, nrec, 0, hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    RC((hybrid12_refine<KM>(c, km, kbits, h, nrec, f, &ok, 0, slice, &distinct)));
  }
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, (ok && distinct) ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != m) { set_err("global order: %llu of %u positions selected", (unsigned long long)tot, m); return E_HIP; }
  arena_release(c, mk);
  if (ngood == (uint64_t)P) {
    *done = true;
    RC(deliver(G, slice, nrec, pre, all, m, nullptr, G_TOP));
  }
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// one level (lib.rs:44-193) on replicated S; the result goes where `mode` says (GOut).
// ---------------------------------------------------------------------------------------------
template <class Sym>
static int glevel(dc3hip_gctx *G, Sym S, u32 m, u64 K, int depth, u32 *out, GOut mode);

// Sorted naming of this rank's key range (lib.rs:80-100), generic over the accessor of the sorted order.
//   counts of distinct / unique names go around (all-gather of two words), names continue after those of the smaller
//   key ranges (equal keys never straddle ranks); the (slot, name [| unique << 31]) pairs land in the caller's buffer
//   pa (m02 entries), which the caller exchanges into R AFTER releasing its sort buffers.
template <class Acc0>
static int gname_pairs(dc3hip_gctx *G, Acc0 acc0, u32 cnt, u32 m0, u32 m02, Rec8 *pa, u32 *sslot, uint64_t *names_total,
                       uint64_t *uniq_total, uint64_t *cnt_pre, bool *discard, bool first_eq = false, bool last_eq_next = false) {
  typedef AccBound<Acc0> Acc;
  Acc acc; acc.a = acc0; acc.first_eq = first_eq ? 1u : 0u; acc.last_eq_next = last_eq_next ? 1u : 0u;
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, std::max<u32>(cnt, 1), kBlock * kNameIPT);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 distinct = 0, uniq = 0;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_NAMING, cnt);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, cnt, ck.chunk, counts, c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, 5 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    distinct = c->h_words[0]; uniq = c->h_words[4];
  }
  uint64_t name_off = 0, upre = 0, cnt_total = 0;
  RC(gather_counts(cm, distinct, &name_off, names_total));
  RC(gather_counts(cm, uniq, &upre, uniq_total));
  RC(gather_counts(cm, cnt, cnt_pre, &cnt_total));
  if (cnt_total != m02) { set_err("global naming: %llu of %u samples selected", (unsigned long long)cnt_total, m02); return E_HIP; }
  // discarding (see discard_recurse): worth it when ~1/6 of the slots would leave the recursion
  const double drop_est = (double)*uniq_total * (double)*uniq_total / (double)m02;
  *discard = sslot && *names_total != m02 && !c->no_discard && m02 < 0x7fffffffu && drop_est * kDiscardMinDropInv >= (double)m02;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_NAMING, cnt);
    if (name_off) {
      hipLaunchKernelGGL(k_add_scalar, dim3(grid_for(c, ck.nchunks)), dim3(kBlock), 0, c->stream, counts, ck.nchunks, (u32)name_off);
      KCHECK();
    }
    hipLaunchKernelGGL((k_name_assign<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, cnt, ck.chunk, counts, m0, pa,
                       *discard ? sslot : (u32 *)nullptr);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}

// Discarding recursion, distributed (the scheme of discard_recurse).  RU[p] = name | unique << 31 is replicated, so the
// reduced string R' and the kept-slot list are built by every rank (streaming); the child returns its suffix array
// REPLICATED (all-gather of slices: 4 B per kept slot — cheaper than a rank exchange); every rank derives the order of
// the non-unique slots (pt) from it and rewrites ITS range of the sorted array; one rank exchange gives rank12.
static int gdiscard(dc3hip_gctx *G, const u32 *RU, const u32 *sslot, u32 cnt, u64 cnt_pre, u32 m02, u64 names, u32 *rank12,
                    int depth) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, m02, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 5);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 5, c->d_words + 5, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 mp = c->h_words[5];
  c->stats.level_kept[depth] = mp;
  if (mp == 0) { set_err("internal: discarding kept no slot"); return E_HIP; }
  u32 *Rp = nullptr, *kept = nullptr, *sap = nullptr;
  RC(arena_alloc(c, (size_t)mp + 16, &Rp));
  RC(arena_alloc(c, (size_t)mp + 16, &kept));
  RC(arena_alloc(c, (size_t)mp + 16, &sap));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_write, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts, Rp, kept);
    KCHECK();
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, Rp, mp, 8u);
    KCHECK();
  }
  SymU32 RS; RS.s = Rp; RS.m = mp;
  if (mp == 1) { hipLaunchKernelGGL(k_base1, dim3(1), dim3(64), 0, c->stream, sap, (u32 *)nullptr); KCHECK(); }
  else RC(glevel<SymU32>(G, RS, mp, names, depth + 1, sap, G_SA));
  u32 *x = nullptr, *pt = nullptr;
  RC(arena_alloc(c, (size_t)mp + 16, &x));
  RC(arena_alloc(c, (size_t)mp + 16, &pt));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, mp);
    const Chunking ckp = make_chunks(c, mp, kBlock);
    u32 *cnt2 = nullptr;
    RC(arena_alloc(c, (size_t)ckp.nchunks + 16, &cnt2));
    hipLaunchKernelGGL(k_discard_gather, dim3(grid_for(c, mp)), dim3(kBlock), 0, c->stream, sap, mp, kept, x);
    KCHECK();
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ckp.nchunks), dim3(kBlock), 0, c->stream, x, mp, ckp.chunk, cnt2);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cnt2, ckp.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL(k_nonuniq_write, dim3(ckp.nchunks), dim3(kBlock), 0, c->stream, x, mp, ckp.chunk, cnt2, pt);
    KCHECK();
  }
  // my range [cnt_pre, cnt_pre + cnt) of the level's sorted array: unique entries keep their place, the t-th
  // non-unique entry (t counted over all ranks) receives pt[t]
  Rec8 *pa = nullptr;
  RC(arena_alloc(c, (size_t)cnt + 16, &pa));
  const Chunking ckl = make_chunks(c, std::max<u32>(cnt, 1), kBlock);
  u32 *cl = nullptr;
  RC(arena_alloc(c, (size_t)ckl.nchunks + 16, &cl));
  u32 nu_local = 0;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, cnt);
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ckl.nchunks), dim3(kBlock), 0, c->stream, sslot, cnt, ckl.chunk, cl);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cl, ckl.nchunks, c->d_words + 6);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 6, c->d_words + 6, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    nu_local = c->h_words[6];
  }
  uint64_t nu_pre = 0, nu_tot = 0;
  RC(gather_counts(cm, nu_local, &nu_pre, &nu_tot));
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, cnt);
    if (nu_pre) {
      hipLaunchKernelGGL(k_add_scalar, dim3(grid_for(c, ckl.nchunks)), dim3(kBlock), 0, c->stream, cl, ckl.nchunks, (u32)nu_pre);
      KCHECK();
    }
    hipLaunchKernelGGL(k_final_assign, dim3(ckl.nchunks), dim3(kBlock), 0, c->stream, sslot, cnt, ckl.chunk, cl, pt, (u32 *)nullptr, pa);
    KCHECK();
    if (cnt_pre) {
      hipLaunchKernelGGL(k_add_val, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, pa, cnt, (u32)cnt_pre);
      KCHECK();
    }
  }
  RC(rank_exchange(G, pa, cnt, m02, rank12, DC3HIP_PH_RANKS));
  arena_release(c, mk);
  return E_OK;
}

template <class Sym>
static int glevel(dc3hip_gctx *G, Sym S, u32 m, u64 K, int depth, u32 *out, GOut mode) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  if (depth >= DC3HIP_MAX_LEVELS) { set_err("recursion deeper than %d levels", DC3HIP_MAX_LEVELS); return E_HIP; }
  if (m <= G->local_max || m < 64) {
    // small level: every rank finishes the recursion on its own copy (no communication below this point)
    if (G->gs.local_from_level < 0) G->gs.local_from_level = depth;
    if (mode == G_TOP) {
      RC(dc3_level<Sym>(c, S, m, K, c->d_sa, nullptr, depth));
      int64_t off, len; block_of(m, P, me, &off, &len);
      G->shard_first = off; G->shard_count = len; G->shard_ptr = c->d_sa + off;
    } else if (mode == G_RANK) {
      RC(dc3_level<Sym>(c, S, m, K, nullptr, out, depth));
    } else {
      RC(dc3_level<Sym>(c, S, m, K, out, nullptr, depth));
    }
    return E_OK;
  }
  const u32 m0 = (m + 2) / 3, m1 = (m + 1) / 3, m2 = m / 3, m02 = m0 + m2;   // lib.rs:45-48
  c->stats.level_n[depth] = m; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
  const ArenaMark mk0 = arena_mark(c);
  u32 *R = nullptr, *rank12 = nullptr;
  RC(arena_alloc(c, (size_t)m02 + 16, &R));
  const u64 B = K + 1;
  const bool direct = (B * B * B) <= 0x7fffffffull;
  if (direct) {
    // order-preserving packed-triple names, computed by every rank for the whole level (one streaming pass)
    c->stats.level_sorted[depth] = 0; c->stats.level_name_width[depth] = 3;
    {
      PhaseScope ps(c, DC3HIP_PH_NAME_DIRECT, m02);
      hipLaunchKernelGGL((k_name_direct<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, (u32)B, 3u,
                         (u32)(B * B), R);
      KCHECK();
    }
    RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
    SymU32 RS; RS.s = R; RS.m = m02;
    RC(glevel<SymU32>(G, RS, m02, B * B * B, depth + 1, rank12, G_RANK));
  } else {
    // ---- sorted naming, split by key range (lib.rs:62-100) ---------------------------------------------------
    c->stats.level_sorted[depth] = 1;
    const u32 b = (u32)B;
    u32 kbits = 0;
    { unsigned __int128 mx = (unsigned __int128)B * B * B - 1; while (mx) { kbits++; mx >>= 1; } }
    u32 *sslot = nullptr;                          // my range of the sorted slots (| unique << 31), for the discarding
    RC(arena_alloc(c, (size_t)m02 + 16, &sslot));
    Rec8 *pa = nullptr;                            // my (slot, name) pairs: below the sort buffers, which go before the exchange
    RC(arena_alloc(c, (size_t)m02 + 16, &pa));
    const ArenaMark mk1 = arena_mark(c);
    const HiMap hm = make_himap(B, kbits, m);
    double pred = 1.0;
    const bool try_hybrid = m02 >= kHybridMinSamples / 4 && !c->no_hybrid && !c->no_hybrid8;
    if (try_hybrid) {
      RC(predict_tie_fraction<Sym>(c, S, m, m0, m02, b, hm, &pred));
      c->stats.level_tie_pred[depth] = pred;
    }
    Key3<Sym> km; km.S = S; km.B = b;
    if (try_hybrid && pred < kFullSortMaxPredicted && !c->no_fullsort) {
      // high-entropy level: if all its triples are distinct, ordering all its positions finishes it
      bool done = false;
      RC((gorder_positions<Key3<Sym>>(G, km, m, kbits, hm, depth, out, mode, &done)));
      if (done) { c->stats.level_sorted[depth] = 5; arena_release(c, mk0); return E_OK; }
      arena_release(c, mk1);
    }
    uint64_t names_total = 0, uniq_total = 0, cnt_pre = 0;
    bool discard = false, named = false;
    u32 cnt = 0;
    if (try_hybrid && pred < c->hybrid_max_pred) {
      // prefix sort + tie refinement of my IMAGE range (equal keys have equal images, so they stay on one rank)
      u64 lo = 0, hi = ~0ull;
      {
        const u32 stride = std::max<u32>(1, m0 / (u32)std::min<u64>(m0, (u64)1024 * P));
        const u32 ng = (m0 - 1) / stride + 1;
        Rec8 *smp = nullptr;
        RC(arena_alloc(c, (size_t)2 * ng, &smp));
        HIPC(hipMemsetAsync(smp, 0xff, (size_t)2 * ng * sizeof(Rec8), c->stream));   // (a missing last mod-2 sample stays a filler)
        hipLaunchKernelGGL((k_pack_image<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, hm, stride, ng, smp);
        KCHECK();
        RC(image_splitters(c, smp, 2 * ng, hm.pbits, P, me, &lo, &hi));
      }
      SelSampleImage<Sym> sel; sel.S = S; sel.B = b; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
      Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
      RC(select_records(c, sel, m02, &ha, &cnt, DC3HIP_PH_PACK));
      RC(arena_alloc(c, (size_t)cnt + 16, &hb));
      RC(arena_alloc(c, (size_t)cnt + 16, &f));
      bool ok = true;
      if (cnt) RC((hybrid_sort_core<Key3<Sym>>(c, km, kbits, hm, ha, hb, cnt, &h, f, &ok, depth)));
      uint64_t g0 = 0, ngood = 0;
      RC(gather_counts(cm, ok ? 1 : 0, &g0, &ngood));
      if (ngood == (uint64_t)P) {
        c->stats.level_sorted[depth] = 2;
        AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
        RC(gname_pairs<AccHyb>(G, acc, cnt, m0, m02, pa, sslot, &names_total, &uniq_total, &cnt_pre, &discard));
        named = true;
      } else {
        arena_release(c, mk1);                       // too many ties somewhere: every rank takes the straight sort
      }
    }
    if (!named) {
      // straight sort of my KEY range: splitters = every rank sorts the same deterministic sample of keys.
      // Where a rank's share is large enough for the splitter ordering (dc3_ssort.hip.hpp), whose cost does not depend on
      // the key width, the records are W-symbol windows instead of triples (order_wide of the single-device build): more
      // names are distinct, the discarding recursion keeps less, the distributed levels below shrink.
      const u32 Ww = wide_window_syms(c, m02 / (u32)P, K), wsb = bits_of(K);
      const u32 W = Ww > 3 ? Ww : 0u, sort_bits = W ? W * wsb : kbits;
      if (W) c->stats.level_name_width[depth] = (int32_t)W;
      Rec16 klo{0, 0, 0, 0}, khi{0, 0, 0, 0};
      {
        u32 ns = (u32)std::min<u64>(m02, (u64)1024 * P);
        const u32 stride = std::max<u32>(1, m02 / ns);
        ns = (m02 - 1) / stride + 1;
        Rec16 *smp = nullptr;
        RC(arena_alloc(c, (size_t)ns, &smp));
        hipLaunchKernelGGL((k_sample_triple_keys<Sym>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, S, b, ns, stride, smp, W, wsb);
        KCHECK();
        void *hsp = nullptr;
        RC(stage_d2h(c, smp, (size_t)ns * sizeof(Rec16), &hsp));
        std::vector<Rec16> hs(static_cast<const Rec16 *>(hsp), static_cast<const Rec16 *>(hsp) + ns);
        std::sort(hs.begin(), hs.end(), [](const Rec16 &x, const Rec16 &y) {
          if (x.k2 != y.k2) return x.k2 < y.k2;
          if (x.k1 != y.k1) return x.k1 < y.k1;
          if (x.k0 != y.k0) return x.k0 < y.k0;
          return x.pos < y.pos;                      // equal keys are split by position (see keypos_lt)
        });
        if (me > 0) klo = hs[(size_t)((u64)me * ns / P)];
        if (me + 1 < P) khi = hs[(size_t)((u64)(me + 1) * ns / P)];
      }
      SelTripleKey<Sym> sel; sel.S = S; sel.B = b; sel.klo = klo; sel.khi = khi; sel.has_lo = me > 0 ? 1u : 0u; sel.last = (me + 1 == P) ? 1u : 0u;
      sel.W = W; sel.sb = wsb;
      Rec16 *recA = nullptr, *recB = nullptr, *sorted = nullptr;
      RC(select_records(c, sel, m02, &recA, &cnt, DC3HIP_PH_PACK));
      RC(arena_alloc(c, (size_t)cnt + 16, &recB));
      sorted = recA;
      if (cnt) {
        bool by_splitters = false;
        RC(ssort<Rec16>(c, recA, recB, cnt, sort_bits, &sorted, &by_splitters));
        if (!by_splitters)
          RC(radix_sort<Rec16>(c, recA, recB, cnt, 0, sort_bits, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
      }
      // equal keys may straddle ranks: every rank learns its neighbours' boundary keys (first / last record of each
      // rank's sorted range, one small all-gather) and names continue across the boundary where they are equal
      struct Edge { u32 f[3], l[3], has, pad; } mine, all_e[kMaxRanks];
      memset(&mine, 0, sizeof(mine));
      if (cnt) {
        Rec16 fl[2];
        void *fp = nullptr;
        RC(stage_d2h(c, sorted, sizeof(Rec16), &fp)); fl[0] = *static_cast<const Rec16 *>(fp);
        RC(stage_d2h(c, sorted + (cnt - 1), sizeof(Rec16), &fp)); fl[1] = *static_cast<const Rec16 *>(fp);
        mine.f[0] = fl[0].k0; mine.f[1] = fl[0].k1; mine.f[2] = fl[0].k2;
        mine.l[0] = fl[1].k0; mine.l[1] = fl[1].k1; mine.l[2] = fl[1].k2; mine.has = 1;
      }
      RC(cm->all_gather_host(&mine, all_e, sizeof(Edge)));
      bool first_eq = false, last_eq_next = false;
      if (cnt) {
        for (int h = me - 1; h >= 0; h--) if (all_e[h].has) { first_eq = memcmp(all_e[h].l, mine.f, 12) == 0; break; }
        for (int h = me + 1; h < P; h++) if (all_e[h].has) { last_eq_next = memcmp(all_e[h].f, mine.l, 12) == 0; break; }
      }
      AccRec<Rec16> acc; acc.s = sorted;
      RC(gname_pairs<AccRec<Rec16>>(G, acc, cnt, m0, m02, pa, sslot, &names_total, &uniq_total, &cnt_pre, &discard, first_eq,
                                    last_eq_next));
    }
    arena_release(c, mk1);
    RC(rank_exchange(G, pa, cnt, m02, R, DC3HIP_PH_NAMING));      // R[slot] = name (| unique << 31), everywhere
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, R, m02, 8u);
    KCHECK();
    if (names_total == m02) {
      rank12 = R;                                                 // all names distinct: the names are the ranks (lib.rs:109-113)
    } else if (discard) {
      c->stats.level_sorted[depth] += 2;
      RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
      RC(gdiscard(G, R, sslot, cnt, cnt_pre, m02, names_total, rank12, depth));
    } else {
      RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
      SymU32 RS; RS.s = R; RS.m = m02;
      RC(glevel<SymU32>(G, RS, m02, names_total, depth + 1, rank12, G_RANK));   // lib.rs:104
    }
  }
  hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, rank12, m02, 8u);
  KCHECK();

  // ---- Step 2 + 3, split by rank range (lib.rs:118-192) -------------------------------------------------------
  // my slice of the level's suffix array comes first on the stack (worst case: everything), so that the tuples can be
  // released before it is delivered
  u32 *slice = (mode == G_TOP) ? c->d_sa : nullptr;
  if (!slice) RC(arena_alloc(c, (size_t)m + 16, &slice));
  const ArenaMark mk_merge = arena_mark(c);
  // rank g owns the output between splitter samples g and g+1; the splitters are the samples of rank bound[g]
  const u32 dskip = m0 - m1;                      // lib.rs:133: the dummy has sample rank 1 and is not a suffix
  const u32 first_rank = 1 + dskip, nAtot = m02 - dskip;
  u32 bound[kMaxRanks + 1];
  for (int h = 0; h <= P; h++) bound[h] = first_rank + (u32)((u64)nAtot * h / P);
  const u32 nsp = (u32)(P - 1);
  if (P > 1) {
    RankTargets t; memset(&t, 0, sizeof(t)); t.n = nsp;
    for (int h = 1; h < P; h++) t.r[h - 1] = bound[h];
    HIPC(hipMemsetAsync(c->d_words + 40, 0xff, kMaxRanks * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_find_ranks, dim3(grid_for(c, m02)), dim3(kBlock), 0, c->stream, rank12, m02, t, c->d_words + 40);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 40, c->d_words + 40, kMaxRanks * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int h = 1; h < P; h++)
      if (c->h_words[40 + h - 1] >= m02) { set_err("global merge: sample rank %u not found (rank12 is not a bijection)", bound[h]); return E_HIP; }
  }
  const u32 lo = bound[me], hi = bound[me + 1], nA = hi - lo;
  // A: my samples in rank order = windowed inversion of (rank - lo, slot), then the tuple gather (slot table built by
  // every rank, streaming).  The P-1 splitter samples ride along behind my nA slots; their tuples come back to the host.
  Tup12 *A = nullptr;
  RC(arena_alloc(c, (size_t)nA + nsp + 16, &A));
  Splitters sp; memset(&sp, 0, sizeof(sp)); sp.n = nsp;
  {
    const ArenaMark mkA = arena_mark(c);
    SelRankRange sr; sr.rank12 = rank12; sr.lo = lo; sr.hi = hi;
    Rec8 *pr = nullptr, *pt = nullptr; u32 got = 0;
    RC(select_records(c, sr, m02, &pr, &got, DC3HIP_PH_RANKS));
    if (got != nA) { set_err("global merge: %u samples in rank range [%u,%u), expected %u", got, lo, hi, nA); return E_HIP; }
    u32 *sa12l = nullptr;
    RC(arena_alloc(c, (size_t)nA + 16, &pt));
    RC(arena_alloc(c, (size_t)nA + nsp + 16, &sa12l));
    if (nA) RC(inverse_permute(c, pr, pt, nA, sa12l, DC3HIP_PH_RANKS));
    if (nsp) HIPC(hipMemcpyAsync(sa12l + nA, c->d_words + 40, nsp * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    const u32 cnt = nA + nsp;
    if (cnt) {
      constexpr u32 kTup0Tile = SortCfg<Tup0, 256>::NW * 64 * SortCfg<Tup0, 256>::IPT;
      const Chunking ckc = make_chunks(c, cnt, kTup0Tile);
      u32 *table0 = nullptr;
      RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
      RC((build_gather_tuples<Sym>(c, S, m, m0, m02, K, rank12, sa12l, cnt, ckc, A, table0)));
    }
    if (nsp) {
      void *spp = nullptr;
      RC(stage_d2h(c, A + nA, nsp * sizeof(Tup12), &spp));
      memcpy(sp.a, spp, nsp * sizeof(Tup12));
    }
    arena_release(c, mkA);
  }
  // B: my mod-0 tuples, sorted by (first symbol, rank of the suffix behind it)
  SelMod0<Sym> sm; sm.S = S; sm.rank12 = rank12; sm.m = m; sm.m0 = m0; sm.me = (u32)me; sm.sp = sp;
  Tup0 *z0 = nullptr; u32 nB = 0;
  RC(select_records(c, sm, m0, &z0, &nB, DC3HIP_PH_COMPACT));
  Tup0G *zs = reinterpret_cast<Tup0G *>(z0);
  if (nB) {
    Tup0G *z1 = nullptr;
    RC(arena_alloc(c, (size_t)nB + 16, &z1));
    Tup0G *t1 = nullptr;
    RC(radix_sort<Tup0G>(c, reinterpret_cast<Tup0G *>(z0), z1, nB, 0, bits_of((u64)m02), &t1, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
    Tup0G *other = (t1 == z1) ? reinterpret_cast<Tup0G *>(z0) : z1;
    RC(radix_sort<Tup0G>(c, t1, other, nB, 32, 32 + bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  }
  // my slice of the level's suffix array
  const u32 total = nA + nB;
  uint64_t pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, total, &pre, &tot, all));
  if (tot != m) { set_err("global merge: slices hold %llu of %u suffixes", (unsigned long long)tot, m); return E_HIP; }
  RC(merge_lists(c, A, nA, reinterpret_cast<const Tup0 *>(zs), nB, slice, nullptr, 0u));
  arena_release(c, mk_merge);
  RC(deliver(G, slice, total, pre, all, m, out, mode));
  arena_release(c, mk0);
  return E_OK;
}

// level 0 shortcut: the whole-text order by 9-symbol (Key9) or, on small alphabets, 3L-symbol windows (KeyT), split by
// key range (conditions as in build_core; no reuse of the order when windows repeat: the recursion decides then)
static int gorder_text_msd(dc3hip_gctx *G, u32 sigma, bool *done, bool *tried, bool have_select);     // (defined behind the wide mode's pieces)
template <class KM>
static int gtext_order_with(dc3hip_gctx *G, KM km, u64 BL, const HiMap &hm, u32 sigma, bool wide, bool *done) {
  dc3hip_ctx *c = G->c;
  const u32 n = (u32)G->total_n;
  u32 kbits = 0;
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  double pred = 1.0;
  if (wide) {                               // hm: image of hm.nbits bits for 12-byte records (see try_text_order12)
    const ArenaMark mk = arena_mark(c);
    const u32 stride = std::max<u32>(1, n >> 20);
    const u32 ns = (n - 1) / stride + 1;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
    KCHECK();
    u32 ts = 0;
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1));
    arena_release(c, mk);
    c->stats.level_tie_pred[0] = pred;
    if (!(pred < kTextSortMaxPredicted)) return E_OK;
    bool tried = false;
    RC(gorder_text_msd(G, sigma, done, &tried, false));
    if (!tried) RC((gorder_positions12<KM>(G, km, n, kbits, hm, done)));
  } else {
    RC(predict_tie_fraction_pos<KM>(c, km, n, hm, &pred));
    c->stats.level_tie_pred[0] = pred;
    if (!text_order_worth_trying(pred, (u64)n, hm.nbits)) return E_OK;
    bool tried = false;
    // (byte windows: gorder_positions has the selecting pass 1 of the single device's own kernels, cheaper still)
    const bool have_select = std::is_same<KM, Key9>::value && !G->no_select && gselect_pays(G, G->comm->nranks) &&
                             msd_geometry(c, n, hm).on;
    RC(gorder_text_msd(G, sigma, done, &tried, have_select));
    if (!tried) RC((gorder_positions<KM>(G, km, n, kbits, hm, 0, nullptr, G_TOP, done)));
  }
  if (*done) {
    c->stats.text_sort_state = 1;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = 5;
  } else {
    c->stats.text_sort_state = 3;       // some window repeats somewhere: the recursion decides
  }
  return E_OK;
}
static int gtext_order(dc3hip_gctx *G, SymU8 S, u32 sigma, bool *done) {
  dc3hip_ctx *c = G->c;
  const u32 n = (u32)G->total_n;
  *done = false;
  const u64 Bq = (u64)sigma + 1, B3 = Bq * Bq * Bq;
  const double need_bits = 2.0 * log2((double)n) + 2.0, sym_bits = log2((double)sigma);
  if (!(n >= kHybridMinSamples / 4 && !c->no_hybrid && !c->no_fullsort && !c->no_text_shortcut && !G->no_text_order)) return E_OK;
  const bool wide = c->text_order12 >= 0 ? c->text_order12 == 1 : bits_of((u64)n - 1) >= 32;
  const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2((double)n) + 4.2) / 9.0));
  if (9.0 * sym_bits >= need_bits && B3 * B3 * B3 > 0x7fffffffull) {
    u32 kbits = 0;
    { unsigned __int128 mx = (unsigned __int128)B3 * B3 * B3 - 1; while (mx) { kbits++; mx >>= 1; } }
    Key9 km; km.S = S; km.B = (u32)Bq; km.B3 = (u32)B3;
    HiMap hm = make_himap(B3, kbits, n, wide ? 64 - ibits : bits_of((u64)n - 1));
    hm.raw = sigma > 128 && !hm.exact ? 1u : 0u;       // (as build_core: byte alphabets)
    return gtext_order_with<Key9>(G, km, B3, hm, sigma, wide, done);
  }
  if (!c->no_long_keys) {
    u32 L = 1; u64 BL = Bq;
    while (L < 20 && BL * Bq <= 0xffffffffull) { BL *= Bq; L++; }
    KeyT km; HiMap hm;
    if (L > 3 && 3.0 * L * sym_bits >= need_bits && make_keyt(S, sigma, L, BL, n, &km, &hm, wide ? ibits : 0u))
      return gtext_order_with<KeyT>(G, km, BL, hm, sigma, wide, done);
  }
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// wide mode: texts of 2^32 bytes and more (kernels and the scope in dc3_wide.hip.hpp)
// ---------------------------------------------------------------------------------------------
static uint8_t *gtext(dc3hip_gctx *G) { return G->wide ? G->w_text : G->c->d_text; }

static int wide_key(dc3hip_gctx *G, u32 sigma, WideKey *k, u32 *ibits_out) {
  const double n = (double)G->total_n;
  // (a text over one symbol: every image is 0 and every window repeats — one group that the deepening orders; the image
  //  arithmetic runs as for two symbols)
  if (sigma < 2) { if (G->no_wide_deepen) { set_err("wide global mode: a text over one symbol has no distinct windows"); return E_TOOBIG; } sigma = 2; }
  const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2(n) + 4.2) / 9.0));
  u32 J = 1; u64 SJ = sigma;
  while (J < kWideMaxImageSyms && (SJ >> std::min<u32>(ibits + 2, 62)) == 0 && SJ * sigma < (1ull << 63)) { SJ *= sigma; J++; }
  if ((SJ >> ibits) == 0) { set_err("wide global mode: alphabet of %u symbols cannot fill a %u-bit image", sigma, ibits); return E_TOOBIG; }
  k->t = gtext(G); k->code = G->c->d_code; k->n = (u64)G->total_n; k->sigma = sigma; k->J = J; k->W = kWideWindow;
  k->mfix = (u64)(((((unsigned __int128)1) << (64 + ibits)) - 1) / SJ);
  k->P1 = SJ / sigma;
  k->lg = 0; k->sh = 0;
  if ((sigma & (sigma - 1)) == 0) {          // power of two: sigma^J = 2^(lg J), image = v >> (lg J - ibits) exactly
    const u32 lg = bits_of((u64)sigma) - 1, sh = lg * J - ibits;
    if (lg >= 1 && lg * J > ibits && sh < 64) { k->lg = lg; k->sh = sh; k->mfix = 1ull << (64 - sh); }
  }
  *ibits_out = ibits;
  return E_OK;
}

// grow one of the wide mode's arrays to at least `need` elements (never while it holds live data)
template <class T>
static int wide_ensure(dc3hip_ctx *c, T **p, size_t *cap, size_t need) {
  if (need <= *cap) return E_OK;
  HIPC(hipStreamSynchronize(c->stream));
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = need + need / 16 + 1024;
  if (hipMalloc(p, want * sizeof(T)) != hipSuccess) {
    (void)hipGetLastError();
    set_err("wide global mode: no device memory for %zu elements of %zu bytes", want, sizeof(T));
    return E_ALLOC;
  }
  *cap = want;
  return E_OK;
}
// ---- groups of any size (kernels: dc3_wide.hip.hpp, "Groups of any size") -------------------------------------------
// f(i, start of i's group) for the n entries whose run structure `same` describes (same[0] = 0)
template <class F>
static int wide_seg_apply(dc3hip_ctx *c, const uint8_t *same, u32 n, F f) {
  if (n == 0) return E_OK;
  const u32 ntiles = (n + kSegTile - 1) / kSegTile;
  const ArenaMark mk = arena_mark(c);
  u32 *tiles = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &tiles));
  hipLaunchKernelGGL(k_seg_last, dim3(ntiles), dim3(kBlock), 0, c->stream, same, n, tiles);
  KCHECK();
  hipLaunchKernelGGL(k_seg_carry, dim3(1), dim3(1024), 0, c->stream, tiles, ntiles);
  KCHECK();
  hipLaunchKernelGGL((k_seg_apply<F>), dim3(ntiles), dim3(kBlock), 0, c->stream, same, n, (const u32 *)tiles, f);
  KCHECK();
  arena_release(c, mk);      // (the stream orders the launches before whatever reuses the table)
  return E_OK;
}
// the members of this rank's groups of more than kWideTieBig entries, compacted in index order
struct WideBig { u32 nb = 0; u32 *cslot = nullptr, *gid = nullptr; u64 *cpos = nullptr; Rec16 *ra = nullptr, *rb = nullptr; };
template <class Pos>
static int wide_big_collect(dc3hip_gctx *G, const uint8_t *same, u32 nrec, Pos pos, WideBig *bg) {
  dc3hip_ctx *c = G->c;
  bg->nb = 0;
  if (nrec == 0) return E_OK;
  RC(wide_ensure(c, &G->w_aux, &G->w_cap_aux, ((size_t)nrec + 16) * 4));
  u32 *gstart = reinterpret_cast<u32 *>(G->w_aux);
  SegStore st; st.gstart = gstart;
  RC(wide_seg_apply(c, same, nrec, st));
  const u32 ntiles = (nrec + kSegTile - 1) / kSegTile;
  const ArenaMark mk = arena_mark(c);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &counts));
  hipLaunchKernelGGL(k_big_count, dim3(ntiles), dim3(kBlock), 0, c->stream, (const u32 *)gstart, nrec, kWideTieBig, counts);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ntiles, c->d_words + 34);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 34, c->d_words + 34, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  const u32 nb = c->h_words[34];
  if (nb) {
    const size_t per = 4 + 4 + 8 + 16 + 16;
    const int rc = wide_ensure(c, &G->w_aux2, &G->w_cap_aux2, ((size_t)nb + 16) * per);
    if (rc != E_OK) { arena_release(c, mk); return rc; }
    unsigned char *b = G->w_aux2;
    bg->ra = reinterpret_cast<Rec16 *>(b); b += ((size_t)nb + 16) * 16;
    bg->rb = reinterpret_cast<Rec16 *>(b); b += ((size_t)nb + 16) * 16;
    bg->cpos = reinterpret_cast<u64 *>(b); b += ((size_t)nb + 16) * 8;
    bg->cslot = reinterpret_cast<u32 *>(b); b += ((size_t)nb + 16) * 4;
    bg->gid = reinterpret_cast<u32 *>(b);
    hipLaunchKernelGGL((k_big_write<Pos>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const u32 *)gstart, nrec, kWideTieBig, (const u32 *)counts, pos,
                       bg->cslot, bg->gid, bg->cpos);
    KCHECK();
  }
  bg->nb = nb;
  arena_release(c, mk);
  return E_OK;
}
// Segmented sort of the collected members: `ncomp` key components of `bits` bits, most significant first, made by
// make(component, order so far, records out); the stable LSD passes order them last component first, the group's start
// index last.  *sorted = the records in final order (pos = index of the member in the compacted list).
template <class Make>
static int wide_big_sort(dc3hip_gctx *G, const WideBig &bg, u32 nrec, int ncomp, u32 bits, Make make, const Rec16 **sorted) {
  dc3hip_ctx *c = G->c;
  const Rec16 *prev = nullptr;
  for (int comp = ncomp - 1; comp >= -1; comp--) {
    // (the records of a component are made IN PLACE over the order so far: place j reads and writes element j only)
    Rec16 *in = prev ? const_cast<Rec16 *>(prev) : bg.ra, *other = (in == bg.ra) ? bg.rb : bg.ra, *res = nullptr;
    if (comp >= 0) RC(make(comp, prev, in));
    else {
      hipLaunchKernelGGL(k_seg_recs_gid, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u32 *)bg.gid, in);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, in, other, bg.nb, 0, comp >= 0 ? bits : bits_of((u64)nrec), &res, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    prev = res;
  }
  *sorted = prev;
  return E_OK;
}
// Big groups of the symbol tie pass (entries with one sort image): ordered by their first kWideBigSyms symbols, exactly.
// pos: positions of the sorted records; same: same-image flags.  The shard then is in order kWideBigSyms symbols deep
// wherever such a group stood (and deeper elsewhere): the caller lowers its depth to that.
constexpr u32 kWideBigSyms = 56;           // 8 components of 7 symbols (a multiple of 4: wide_cmp's depth)
template <class Pos>
static int wide_big_syms(dc3hip_gctx *G, const uint8_t *same, u32 nrec, Pos pos, const WideKey &k, u32 *nb_out) {
  dc3hip_ctx *c = G->c;
  WideBig bg;
  RC(wide_big_collect(G, same, nrec, pos, &bg));
  *nb_out = bg.nb;
  if (!bg.nb) return E_OK;
  PhaseScope ps(c, DC3HIP_PH_TIES, bg.nb);
  const Rec16 *sorted = nullptr;
  RC(wide_big_sort(G, bg, nrec, (int)(kWideBigSyms / 7), 63u, [&](int comp, const Rec16 *prev, Rec16 *out) -> int {
    SegKeySyms key; key.k = k; key.off = (u32)comp * 7u;
    hipLaunchKernelGGL((k_seg_recs<SegKeySyms>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u64 *)bg.cpos, key, k.code, out);
    KCHECK();
    return E_OK;
  }, &sorted));
  hipLaunchKernelGGL(k_seg_writeback, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u64 *)bg.cpos, G->w_shard);
  KCHECK();
  return E_OK;
}
// Big groups of a deepening round (entries that agree on D symbols): ordered by the W rank look-ups isa[p + j D], j = 1..W,
// the new flags written for all their members (words[2] of c->d_words + 10 counts those that still agree).
static int wide_big_isa(dc3hip_gctx *G, const uint8_t *eq, u32 nrec, u64 n, u64 D, u32 W, uint8_t *neweq) {
  dc3hip_ctx *c = G->c;
  WideBig bg;
  PosShard ps; ps.s = G->w_shard;
  RC(wide_big_collect(G, eq, nrec, ps, &bg));
  if (!bg.nb) return E_OK;
  PhaseScope pss(c, DC3HIP_PH_TIES, bg.nb);
  const Rec16 *sorted = nullptr;
  RC(wide_big_sort(G, bg, nrec, (int)W, bits_of(n), [&](int comp, const Rec16 *prev, Rec16 *out) -> int {
    SegKeyIsa key; key.isa = G->w_isa; key.n = n; key.add = (u64)(comp + 1) * D;
    hipLaunchKernelGGL((k_seg_recs<SegKeyIsa>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u64 *)bg.cpos, key, (const uint16_t *)nullptr, out);
    KCHECK();
    return E_OK;
  }, &sorted));
  hipLaunchKernelGGL(k_seg_writeback, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u64 *)bg.cpos, G->w_shard);
  KCHECK();
  SegCmpIsa cmp; cmp.isa = G->w_isa; cmp.n = n; cmp.D = D; cmp.W = W;
  hipLaunchKernelGGL((k_seg_neweq<SegCmpIsa>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u32 *)bg.gid,
                     (const u64 *)bg.cpos, cmp, neweq, c->d_words + 10);
  KCHECK();
  return E_OK;
}

// The tie rounds of a wide build over the sorted records h[0..nrec): positions to G->w_shard, statistics in
// c->h_words[10..12] (oversized group, tied records, windows that still agree after the last round).
template <class Launch>
static int wide_tie_rounds_with(dc3hip_gctx *G, u32 nrec, WideKey k, Launch launch) {
  dc3hip_ctx *c = G->c;
    // tie pass; while a few windows still agree completely it is repeated with a deeper compare: kWideWindow symbols, then
    // kWideWindowDeep, then 16 times deeper per round for as long as (windows that still agree) x (next depth) stays inside
    // a work budget — the compare is lazy, so the depth only costs where windows really agree that far.  This settles
    // repeats of any length a few of which exist (two copies of a 100 kB block: 10^5 tied pairs x 10^5 symbols); what
    // the budget does not cover is refused (there is no recursion with 64-bit positions).
    u32 depth = kWideWindow;
    for (int round = 0;; round++) {
      k.W = depth;
      G->w_depth = depth;
      HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
      if (nrec) {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        launch(k);
        KCHECK();
      }
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      if (c->h_words[10] != 0 || c->h_words[12] == 0) break;
      if (round == 0) { if (c->h_words[12] > (1u << 20)) break; depth = kWideWindowDeep; continue; }
      const u64 next = (u64)depth * 16;
      // (with the deepening by rank look-ups behind it, a symbol round is only worth its reads while they stay below what
      //  one exchange of the shards moves)
      const u64 budget = G->no_wide_deepen ? kWideTieBudget : std::max<u64>(1ull << 28, 4 * (u64)G->total_n);
      if (next > kWideMaxDepth || (u64)c->h_words[12] * next > budget) break;
      depth = (u32)next;
    }
    return E_OK;
}

static int wide_tie_rounds(dc3hip_gctx *G, const Rec16 *h, u32 nrec, WideKey k) {
  dc3hip_ctx *c = G->c;
  return wide_tie_rounds_with(G, nrec, k, [&](const WideKey &kk) {
    hipLaunchKernelGGL(k_wide_ties, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, h, nrec, kk, G->w_shard, c->d_words + 10);
  });
}

// Pass 1 of the bucket ordering of a wide rank: selection + x' + partition by its top d1 bits, straight from the text
struct WidePass1 : MsdPass1 {
  WideKey k; WideRange rg; u64 chunk = 0; u32 nchunks = 0, cpg = 0;
  int launch(dc3hip_ctx *c, u64 *out, u32, u64, u32, const MsdGeom &, u32, const u32 *, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
#define DC3_WIDE_ATTR(JM, PW) HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_wide_part1<JM, PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWidePartSmem))
      DC3_WIDE_ATTR(8, false); DC3_WIDE_ATTR(16, false); DC3_WIDE_ATTR(24, false); DC3_WIDE_ATTR(kWideMaxImageSyms, false);
      DC3_WIDE_ATTR(8, true); DC3_WIDE_ATTR(16, true); DC3_WIDE_ATTR(24, true); DC3_WIDE_ATTR(kWideMaxImageSyms, true);
#undef DC3_WIDE_ATTR
      attr_set[c->device & 15] = true;
    }
#define DC3_WIDE_P1(JM, PW) hipLaunchKernelGGL((k_wide_part1<JM, PW>), dim3(kMsdGroups * cpg), dim3(kWideNT), kWidePartSmem, c->stream, k, rg, chunk, nchunks, cpg, cur1, out, c->d_xcdmon)
    if (k.lg) switch (wide_jmax(k.J)) { case 8: DC3_WIDE_P1(8, true); break; case 16: DC3_WIDE_P1(16, true); break; case 24: DC3_WIDE_P1(24, true); break; default: DC3_WIDE_P1(kWideMaxImageSyms, true); }
    else switch (wide_jmax(k.J)) { case 8: DC3_WIDE_P1(8, false); break; case 16: DC3_WIDE_P1(16, false); break; case 24: DC3_WIDE_P1(24, false); break; default: DC3_WIDE_P1(kWideMaxImageSyms, false); }
#undef DC3_WIDE_P1
    KCHECK();
    return E_OK;
  }
};

// this rank's image range [lo, hi) from a strided sample of the replicated text (every rank computes the same sorted
// sample `img`; rank r takes the r-th P-quantile as its lower bound)
static int wide_splitters(dc3hip_gctx *G, const WideKey &k, std::vector<u64> *img, u64 *lo, u64 *hi) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = k.n;
  const ArenaMark mk = arena_mark(c);
  const u32 ns = (u32)std::min<u64>(n, (u64)4096 * P);
  const u64 stride = std::max<u64>(1, n / ns);
  const u32 cnt = (u32)((n - 1) / stride + 1);
  u64 *d_img = nullptr;
  RC(arena_alloc(c, (size_t)cnt, &d_img));
  hipLaunchKernelGGL(k_wide_sample, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, k, stride, cnt, d_img);
  KCHECK();
  void *ip = nullptr;
  RC(stage_d2h(c, d_img, (size_t)cnt * 8, &ip));
  img->assign(static_cast<const u64 *>(ip), static_cast<const u64 *>(ip) + cnt);
  arena_release(c, mk);
  std::sort(img->begin(), img->end());
  *lo = 0; *hi = ~0ull;
  if (me > 0) *lo = (*img)[(size_t)((u64)me * cnt / P)];
  if (me + 1 < P) *hi = (*img)[(size_t)((u64)(me + 1) * cnt / P)];
  return E_OK;
}

// whether a wide build of n bytes over P ranks uses the bucket ordering on 8-byte words: the same answer on every rank
static bool wide_msd_applies(const dc3hip_gctx *G, u64 n, int P, u32 ibits) {
  const u64 est = n / (u64)P;
  if (G->c->no_msd || G->no_wide_msd || est < G->wide_msd_min || est < 8192) return false;
  const u32 pb = bits_of(n - 1), lg = bits_of(est - 1);
  const u32 tb = std::min<u32>(20, lg > 10 ? lg - 10 : 1);
  // a rank's span is about 2^ibits / P: x' keeps min(bits of the span, 64 - pb + d1) bits and needs tb + 4 of them
  const u32 eb_typ = ibits > bits_of((u64)P) ? ibits - bits_of((u64)P) : 0;
  return pb < 54 && std::min<u32>(eb_typ, 64 - pb + (tb <= 10 ? tb : (tb + 1) / 2)) >= tb + 6;
}

// The order of this rank's image range [lo, hi) by the bucket ordering on 8-byte words (dc3_wide_msd.hip.hpp): counting
// pass over the text, partition pass 1 with selection, the 8-byte passes 2 and 3 of dc3_msd.hip.hpp, tie rounds.
// *done = false: does not apply (switched off, too few positions, too few image bits) or a sub-bucket outgrew the local
// sort — nothing was delivered and the caller runs the 16-byte LSD form.  *nrec_out = the rank's record count.
// OutT / bufs: where the words and the positions live — bufs(nrec, &wa, &wb, &out) hands out two arrays of nrec 8-byte words
// and the array of nrec positions (wide contexts: their own device buffers, 64-bit positions; texts below 2^32: the arena
// and the suffix-array buffer, 32-bit positions).
template <class OutT, class Bufs>
static int wide_msd_order(dc3hip_gctx *G, const WideKey &k, u32 ibits, u64 lo, u64 hi, bool last, u32 *nrec_out, bool *done, Bufs bufs) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks;
  const u64 n = (u64)G->total_n;
  *done = false;
  const u64 est = n / (u64)P;
  const u64 top = ibits >= 64 ? ~0ull : (1ull << ibits);
  const u64 span = (last ? top : hi) - lo;
  if (span < 2) return E_OK;
  const u32 eb = bits_of(span - 1), pb = bits_of(n - 1);
  const u32 lg = bits_of(est - 1);
  u32 tb = lg > 10 ? lg - 10 : 1;
  if (tb > 20) tb = 20;
  u32 d1, d2;
  if (tb <= 10) { d1 = tb; d2 = 0; } else { d1 = (tb + 1) / 2; d2 = tb - d1; }
  const u32 E = std::min<u32>(std::min<u32>(eb, 63u), 64u - pb + d1);
  if (pb >= 64 || E < tb + 4) return E_OK;
  WidePass1 p1;
  p1.k = k;
  p1.rg.lo = lo; p1.rg.hi = hi; p1.rg.last = last ? 1u : 0u; p1.rg.eb = eb; p1.rg.E = E; p1.rg.d1 = d1; p1.rg.pb = pb;
  p1.rg.M = (u64)((((unsigned __int128)1) << (63 + eb)) / span);
  p1.chunk = ((n + 2047) / 2048 + kWideRound - 1) / kWideRound * kWideRound;
  p1.nchunks = (u32)((n + p1.chunk - 1) / p1.chunk);
  p1.cpg = (p1.nchunks + kMsdGroups - 1) / kMsdGroups;
  const u32 nb1 = 1u << d1;
  const ArenaMark mk = arena_mark(c);
  u32 *table = nullptr, *cntg = nullptr;
  RC(arena_alloc(c, (size_t)1024 * p1.nchunks, &table));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cntg));
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, (int64_t)n);
#define DC3_WIDE_C1(JM, PW) hipLaunchKernelGGL((k_wide_count1<JM, PW>), dim3(p1.nchunks), dim3(kWideNT), 0, c->stream, k, p1.rg, p1.chunk, p1.nchunks, table)
    if (k.lg) switch (wide_jmax(k.J)) { case 8: DC3_WIDE_C1(8, true); break; case 16: DC3_WIDE_C1(16, true); break; case 24: DC3_WIDE_C1(24, true); break; default: DC3_WIDE_C1(kWideMaxImageSyms, true); }
    else switch (wide_jmax(k.J)) { case 8: DC3_WIDE_C1(8, false); break; case 16: DC3_WIDE_C1(16, false); break; case 24: DC3_WIDE_C1(24, false); break; default: DC3_WIDE_C1(kWideMaxImageSyms, false); }
#undef DC3_WIDE_C1
    KCHECK();
    hipLaunchKernelGGL(k_msd_cnt1, dim3(nb1), dim3(kBlock), 0, c->stream, (const u32 *)table, p1.nchunks, p1.cpg, cntg);
    KCHECK();
  }
  void *hcp = nullptr;
  RC(stage_d2h(c, cntg, (size_t)nb1 * kMsdGroups * 4, &hcp));
  u64 nrec64 = 0;
  for (size_t i = 0; i < (size_t)nb1 * kMsdGroups; i++) nrec64 += static_cast<const u32 *>(hcp)[i];
  if (nrec64 > (u64)DC3HIP_MAX_N) { set_err("wide global mode: rank %d would hold %llu suffixes (more ranks needed)", cm->rank, (unsigned long long)nrec64); return E_TOOBIG; }
  const u32 nrec = (u32)nrec64;
  *nrec_out = nrec;
  if (nrec < 4096) { arena_release(c, mk); return E_OK; }
  u64 *wa = nullptr, *wb = nullptr; OutT *shard = nullptr; uint8_t *same = nullptr;
  RC(bufs(nrec, &wa, &wb, &shard, &same));
  MsdGeom g;
  g.on = true; g.d1 = d1; g.d2 = d2; g.cpg = p1.cpg; g.ck.nchunks = p1.nchunks; g.ck.chunk = 0; g.img_lo = 0; g.ebits = E;
  HiMap hm; hm.mfix = 0; hm.shx = 0; hm.pbits = pb; hm.nbits = E; hm.exact = 0; hm.raw = 0;
  Rec8 *res = nullptr, *where = nullptr; MsdRedo redo; bool ok = false;
  RC(msd_sort(c, reinterpret_cast<Rec8 *>(wa), reinterpret_cast<Rec8 *>(wb), nrec, hm, g, table, nullptr, &res, &redo, &ok, &where, &p1, same));
  if (!ok) { arena_release(c, mk); return E_OK; }
  const u64 *h = reinterpret_cast<const u64 *>(res);
  RC(wide_tie_rounds_with(G, nrec, k, [&](const WideKey &kk) {
    hipLaunchKernelGGL((k_wide_ties8<OutT>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, h, (const uint8_t *)same, nrec, pb, kk, shard, c->d_words + 10);
  }));
  if constexpr (sizeof(OutT) == 8) {
    if (c->h_words[10] && G->wide && !G->no_wide_deepen) {      // (groups beyond kWideTieBig records: see gbuild_wide)
      PosWord8 ph; ph.h = h; ph.pmask = (1ull << pb) - 1ull;
      u32 nbig = 0;
      RC(wide_big_syms(G, same, nrec, ph, k, &nbig));
      G->w_depth = std::min<u32>(G->w_depth, kWideBigSyms);
      c->h_words[10] = 0; c->h_words[12] = std::max<u32>(c->h_words[12], 1u);
    }
  }
  arena_release(c, mk);
  G->gs.wide_msd = 1;
  *done = true;
  return E_OK;
}

// The whole-text order of a text below 2^32 bytes in the form the wide contexts use (gbuild_wide / wide_msd_order): every
// rank takes the images of its range straight from its replica of the text — counting pass, partition pass 1 with
// selection, 8-byte passes 2 and 3, tie rounds with lazily compared windows — and nothing but the text blocks has crossed
// the transport.  Replaces the routed order (pack own block, partition by owner, all-to-all of 8-byte records, count the
// top digit again) where the bucket ordering applies: numbers in DESIGN.md §6.  *done = false: some rank's windows repeat (or the ordering does not apply): the caller goes on as before.
static int gorder_text_msd(dc3hip_gctx *G, u32 sigma, bool *done, bool *tried, bool have_select) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = (u64)G->total_n;
  *done = false; *tried = false;
  if (sigma < 2) return E_OK;
  WideKey k; u32 ibits = 0;
  { char keep[sizeof(g_err)]; snprintf(keep, sizeof(keep), "%s", g_err); if (wide_key(G, sigma, &k, &ibits) != E_OK) { set_err("%s", keep); return E_OK; } }
  if (!wide_msd_applies(G, n, P, ibits)) return E_OK;
  if ((double)k.W * log2((double)sigma) < 2.0 * log2((double)n) + 2.0) return E_OK;
  // Where it pays (total work of P loopback ranks on one GPU, 256 MiB random bytes: routed 8.1 / 9.0 ms for P = 2 / 4,
  // unrouted 8.3 / 11.5 — every rank evaluates all n positions twice): from 2^31 positions on, where the routed order
  // would sort 12-byte records with LSD passes, and for two ranks.  DC3HIP_WIDE_MSD_MIN set explicitly (tests) forces it.
  if (!(G->wide_msd_forced || (P <= 2 && !have_select) || bits_of(n - 1) >= 32)) return E_OK;
  *tried = true;
  const ArenaMark mk = arena_mark(c);
  u64 lo = 0, hi = ~0ull;
  std::vector<u64> img;
  RC(wide_splitters(G, k, &img, &lo, &hi));
  u32 nrec = 0; bool ordered = false;
  c->h_words[10] = c->h_words[11] = c->h_words[12] = 0;
  const int rc = wide_msd_order<u32>(G, k, ibits, lo, hi, me + 1 == P, &nrec, &ordered, [&](u32 cnt, u64 **wa, u64 **wb, u32 **out, uint8_t **same) -> int {
    RC(arena_alloc(c, (size_t)cnt + 16, wa));
    RC(arena_alloc(c, (size_t)cnt + 16, wb));
    RC(arena_alloc(c, (size_t)cnt + 16, same));
    *out = c->d_sa;
    return E_OK;
  });
  if (rc != E_OK && rc != E_TOOBIG && rc != E_ALLOC) return rc;
  const bool mine_ok = rc == E_OK && ordered && c->h_words[10] == 0 && c->h_words[12] == 0;
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, mine_ok ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, mine_ok ? nrec : 0, &pre, &tot, all));
  arena_release(c, mk);
  if (ngood == (uint64_t)P) {
    if (tot != n) { set_err("global order: %llu of %llu positions selected", (unsigned long long)tot, (unsigned long long)n); return E_HIP; }
    c->stats.level_tied[0] = c->h_words[11];
    *done = true;
    RC(deliver(G, c->d_sa, nrec, pre, all, (u32)n, nullptr, G_TOP));
  }
  return E_OK;
}

template <class T> static void wide_release(T **p, size_t *cap) { if (*p) (void)hipFree(*p); *p = nullptr; *cap = 0; }
// Deepening by rank look-ups (kernels and the idea: dc3_wide.hip.hpp): collective; entered when some rank's windows still
// agree after the last symbol compare (depth G->w_depth) and no rank met an oversized group.  Every round all ranks
// exchange their shards and equal-window flags (9 bytes per suffix of the text), build the inverse, and order their
// groups by kWideDeepenW + 1 rank look-ups per compare.  *ok = every window of every rank is distinct now; the shards are
// in suffix order and G->w_isa is the exact inverse (kept for the verifier).  *ok = false: no memory, or an oversized group.
constexpr u32 kWideDeepenW = 16;
static int wide_deepen(dc3hip_gctx *G, WideKey k, u32 nrec, u64 pre, const uint64_t *all, bool *ok) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = k.n;
  *ok = false;
  HIPC(hipStreamSynchronize(c->stream));
  wide_release(&G->w_ra, &G->w_cap_a);                     // the sort's buffers are done with: room for the inverse
  wide_release(&G->w_rb, &G->w_cap_b);
  u64 maxshard = 0;
  for (int r = 0; r < P; r++) maxshard = std::max<u64>(maxshard, all[r]);
  // the inverse is built from one rank's shard at a time (w_sa_all = the largest shard), never from the whole order
  int rc_alloc = wide_ensure(c, &G->w_sa_all, &G->w_cap_sa, (size_t)maxshard + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_isa, &G->w_cap_isa, (size_t)n + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_eq_all, &G->w_cap_eq, (size_t)n + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_eq2, &G->w_cap_eq2, (size_t)nrec + 16);
  if (rc_alloc != E_OK && rc_alloc != E_ALLOC) return rc_alloc;
  uint64_t badp = 0, nbad = 0;
  RC(gather_counts(cm, rc_alloc != E_OK ? 1u : 0u, &badp, &nbad));
  if (nbad) return E_OK;                                   // (every rank returns here: the caller refuses the text as before)
  size_t roff1[kMaxRanks], rb1[kMaxRanks];
  u64 first[kMaxRanks];
  { u64 acc = 0; for (int r = 0; r < P; r++) { first[r] = acc; roff1[r] = (size_t)acc; rb1[r] = (size_t)all[r]; acc += all[r]; } }
  // the depth the look-ups start from: what EVERY rank's symbol compares reached (a rank stops deepening them by its own
  // count of agreeing windows; its shard is in order at least that deep)
  u64 D = G->w_depth;
  { uint64_t mine = G->w_depth, depths[kMaxRanks]; RC(cm->all_gather_host(&mine, depths, sizeof(uint64_t))); for (int r = 0; r < P; r++) D = std::min<u64>(D, depths[r]); }
  k.W = (u32)D;
  if (nrec) {
    PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
    hipLaunchKernelGGL(k_wide_eq, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_shard, nrec, k, G->w_eq_all + pre);
    KCHECK();
  }
  bool final_round = false;
  for (int round = 0; round < 64; round++) {
    RC(cm->all_gather_v(G->w_eq_all + pre, (size_t)nrec, G->w_eq_all, roff1, rb1, c->stream));
    G->gs.exchanges += 1;
    // isa[p] = 1 + index of the first entry of p's group, rank by rank: every rank receives rank r's shard (an all-gather
    // in which only r contributes) and scatters the group starts of that range; a rank's range begins with a new group
    HIPC(hipMemsetAsync(G->w_isa + n, 0, 8, c->stream));
    for (int r = 0; r < P; r++) {
      if (all[r] == 0) continue;
      size_t ro[kMaxRanks], rbz[kMaxRanks];
      for (int q = 0; q < P; q++) { ro[q] = 0; rbz[q] = 0; }
      rbz[r] = (size_t)all[r] * 8;
      RC(cm->all_gather_v(r == me ? (const void *)G->w_shard : (const void *)G->w_sa_all, r == me ? (size_t)nrec * 8 : 0, G->w_sa_all, ro, rbz, c->stream));
      PhaseScope ps(c, DC3HIP_PH_RANKS, (int64_t)all[r]);
      SegIsa f; f.sa = G->w_sa_all; f.isa = G->w_isa; f.base = first[r];
      RC(wide_seg_apply(c, G->w_eq_all + first[r], (u32)all[r], f));
    }
    if (final_round) { *ok = true; break; }
    HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
    int rc_big = E_OK;
    if (nrec) {
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        hipLaunchKernelGGL(k_wide_ties_isa, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, G->w_shard, (const uint8_t *)(G->w_eq_all + pre), nrec,
                           (const u64 *)G->w_isa, n, D, kWideDeepenW, G->w_eq2, c->d_words + 10);
        KCHECK();
      }
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      if (c->h_words[10]) {
        // groups beyond kWideTieBig members: a segmented sort by the same look-ups (rank-local; its failure is agreed on below)
        rc_big = wide_big_isa(G, G->w_eq_all + pre, nrec, n, D, kWideDeepenW, G->w_eq2);
        if (rc_big != E_OK && rc_big != E_ALLOC) return rc_big;
      }
      HIPC(hipMemcpyAsync(G->w_eq_all + pre, G->w_eq2, (size_t)nrec, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->h_words[10] = 0;
    G->gs.wide_deepen_rounds += 1;
    uint64_t p0 = 0, nfail = 0, ntied = 0;
    RC(gather_counts(cm, rc_big != E_OK ? 1u : 0u, &p0, &nfail));
    RC(gather_counts(cm, c->h_words[12] ? 1u : 0u, &p0, &ntied));
    if (nfail) { if (rc_big == E_OK) set_err("wide global mode: another rank has no memory for its groups of tied suffixes"); break; }
    D *= (u64)kWideDeepenW + 1;
    if (!ntied) final_round = true;                        // (one more exchange: the inverse of the finished order)
    else if (D > 2 * n * ((u64)kWideDeepenW + 1)) { set_err("internal: suffixes still tied %llu symbols deep", (unsigned long long)D); return E_HIP; }
  }
  if (*ok) { G->w_isa_valid = true; c->h_words[10] = 0; c->h_words[12] = 0; G->w_depth = (u32)std::min<u64>(D, 1u << 30); }
  return E_OK;
}

// The placement probe of context creation (xcd_rr -> no_msd) is a per-device observation, but no_msd decides which COLLECTIVE
// schedule a rank runs (selected pass 1 without an all-to-all, or the routed form; whether the wide bucket ordering is
// tried): ranks on different devices — or a probe disturbed on one of them — must not disagree.  One host all-gather per
// build: the bucket ordering is used only if every rank may use it.
static int gagree_placement(dc3hip_gctx *G) {
  GComm *cm = G->comm;
  if (cm->nranks == 1) return E_OK;
  uint64_t mine = G->c->no_msd ? 1u : 0u, all[kMaxRanks];
  RC(cm->all_gather_host(&mine, all, sizeof(uint64_t)));
  for (int r = 0; r < cm->nranks; r++) if (all[r]) G->c->no_msd = true;
  return E_OK;
}

static int gbuild_wide(dc3hip_gctx *G) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = (u64)G->total_n;
  c->n = 0;
  c->arena_off = 0;
  RC(gagree_placement(G));
  G->w_isa_valid = false;
  if (G->w_sa_all || G->w_isa) {             // (a deepened build's whole-order arrays: the sort needs the room again)
    HIPC(hipStreamSynchronize(c->stream));
    wide_release(&G->w_sa_all, &G->w_cap_sa); wide_release(&G->w_isa, &G->w_cap_isa); wide_release(&G->w_eq_all, &G->w_cap_eq); wide_release(&G->w_eq2, &G->w_cap_eq2);
  }
  RC(ensure_arena(c, (size_t)256 << 20));   // the sorts' tables: digit table 8 MB, 2 x 2^20 sub-buckets x 8 groups x 4 bytes, counts
  RC(build_begin(c));
  {
    size_t roff[kMaxRanks], rbytes[kMaxRanks];
    for (int r = 0; r < P; r++) { int64_t o, l; block_of((int64_t)n, P, r, &o, &l); roff[r] = (size_t)o; rbytes[r] = (size_t)l; }
    RC(cm->all_gather_v(G->w_text + roff[me], rbytes[me], G->w_text, roff, rbytes, c->stream));
    HIPC(hipMemsetAsync(G->w_text + n, 0, 64, c->stream));
  }
  // alphabet (the presence kernel counts in 32 bits: pieces of 2^30 bytes)
  HIPC(hipMemsetAsync(c->d_present, 0, 256 * sizeof(u32), c->stream));
  for (u64 off = 0; off < n; off += (u64)1 << 30) {
    const u32 len = (u32)std::min<u64>((u64)1 << 30, n - off);
    hipLaunchKernelGGL(k_byte_presence, dim3(grid_for(c, (u64)len / 16 + 1)), dim3(kBlock), 0, c->stream, G->w_text + off, len, c->d_present);
    KCHECK();
  }
  hipLaunchKernelGGL(k_make_codes, dim3(1), dim3(kBlock), 0, c->stream, c->d_present, c->d_code, c->d_words + 1);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 1, c->d_words + 1, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  const u32 sigma = c->h_words[1];
  WideKey k; u32 ibits = 0;
  RC(wide_key(G, sigma, &k, &ibits));
  if (G->no_wide_deepen && (double)k.W * log2((double)sigma) < 2.0 * log2((double)n) + 2.0) {
    set_err("wide global mode: %u-symbol windows over %u symbols cannot all be distinct in %llu bytes", k.W, sigma, (unsigned long long)n);
    return E_TOOBIG;
  }
  const ArenaMark mk = arena_mark(c);
  // splitters from a strided sample (every rank computes the same ones from the replicated text)
  u64 lo = 0, hi = ~0ull;
  std::vector<u64> img;
  RC(wide_splitters(G, k, &img, &lo, &hi));
  // The selection, the sort and the tie pass of this rank.  A refusal that depends on the data and on the rank (its share
  // exceeds 2^32 - 2^24 suffixes, no device memory for the records) must not leave the other ranks waiting in the
  // collectives below: the status is agreed on there and every rank returns the same error.
  u32 nrec = 0;
  c->h_words[10] = c->h_words[11] = c->h_words[12] = 0;
  // agreement on a refusal (see above): 0, or the code every rank returns
  auto agree = [&](int rc_local) -> int {
    uint64_t refp = 0, refused = 0;
    char local_err[sizeof(g_err)];
    snprintf(local_err, sizeof(local_err), "%s", g_err);
    RC(gather_counts(cm, rc_local == E_TOOBIG ? 1u : rc_local == E_ALLOC ? (1u << 20) : 0u, &refp, &refused));
    if (!refused) return E_OK;
    const int rc_all = (refused >> 20) ? E_ALLOC : E_TOOBIG;
    if (rc_local != E_OK) set_err("%s", local_err);
    else set_err("wide global mode: another rank refused its share (%s)", rc_all == E_ALLOC ? "no device memory for its records" : "more ranks needed");
    return rc_all;
  };
  // (no record is routed between ranks: every rank selects straight from its replica of the text, so what one rank cannot
  //  order by the bucket ordering it orders by the 16-byte LSD form on its own, and no collective sits in between)
  const bool msd_static = wide_msd_applies(G, n, P, ibits);
  const int local_rc = [&]() -> int {
    // bucket ordering on 8-byte words where it applies (the 16-byte LSD form below otherwise)
    if (msd_static) {
      bool msd_done = false;
      // (two arrays of 8-byte words inside the record buffers, and the shard)
      RC((wide_msd_order<u64>(G, k, ibits, lo, hi, me + 1 == P, &nrec, &msd_done, [&](u32 cnt, u64 **wa, u64 **wb, u64 **out, uint8_t **same) -> int {
        RC(wide_ensure(c, &G->w_ra, &G->w_cap_a, (size_t)cnt / 2 + 16));
        RC(wide_ensure(c, &G->w_rb, &G->w_cap_b, (size_t)cnt / 2 + 16));
        RC(wide_ensure(c, &G->w_shard, &G->w_cap_s, (size_t)cnt + 16));
        RC(wide_ensure(c, &G->w_same, &G->w_cap_same, (size_t)cnt + 16));
        *wa = reinterpret_cast<u64 *>(G->w_ra); *wb = reinterpret_cast<u64 *>(G->w_rb); *out = G->w_shard; *same = G->w_same;
        return E_OK;
      })));
      if (msd_done) return E_OK;
    }
    // count, allocate, write
    const u64 chunk = (u64)1 << 20;
    const u64 nblocks64 = (n + chunk - 1) / chunk;
    if (nblocks64 > 0x7fffffffull) { set_err("wide global mode: text too long"); return E_TOOBIG; }
    const u32 nblocks = (u32)nblocks64;
    u32 *counts = nullptr;
    RC(arena_alloc(c, (size_t)nblocks + 16, &counts));
    const u32 last = (me + 1 == P) ? 1u : 0u;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, n);
      hipLaunchKernelGGL((k_wide_select<false>), dim3(nblocks), dim3(kBlock), 0, c->stream, k, chunk, lo, hi, last, counts,
                         (const u32 *)nullptr, (Rec16 *)nullptr);
      KCHECK();
    }
    // (the per-block counts are summed in 64 bits on the host: a rank's share must stay below 2^32 - 2^24 records)
    void *hcp = nullptr;
    RC(stage_d2h(c, counts, (size_t)nblocks * 4, &hcp));
    std::vector<u32> hc(static_cast<const u32 *>(hcp), static_cast<const u32 *>(hcp) + nblocks);
    u64 nrec64 = 0;
    for (u32 b = 0; b < nblocks; b++) { const u32 v = hc[b]; hc[b] = (u32)nrec64; nrec64 += v; }
    if (nrec64 > (u64)DC3HIP_MAX_N) { set_err("wide global mode: rank %d would hold %llu suffixes (more ranks needed)", me, (unsigned long long)nrec64); return E_TOOBIG; }
    nrec = (u32)nrec64;
    HIPC(hipMemcpyAsync(counts, hc.data(), (size_t)nblocks * 4, hipMemcpyHostToDevice, c->stream));
    RC(wide_ensure(c, &G->w_ra, &G->w_cap_a, (size_t)nrec + 16));
    RC(wide_ensure(c, &G->w_rb, &G->w_cap_b, (size_t)nrec + 16));
    RC(wide_ensure(c, &G->w_shard, &G->w_cap_s, (size_t)nrec + 16));
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, n);
      hipLaunchKernelGGL((k_wide_select<true>), dim3(nblocks), dim3(kBlock), 0, c->stream, k, chunk, lo, hi, last, (u32 *)nullptr,
                         (const u32 *)counts, G->w_ra);
      KCHECK();
    }
    Rec16 *h = G->w_ra;
    if (nrec) RC(radix_sort<Rec16>(c, G->w_ra, G->w_rb, nrec, 0, ibits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    RC(wide_tie_rounds(G, h, nrec, k));
    if (c->h_words[10] && !G->no_wide_deepen) {
      // images shared by more than kWideTieBig records (a run of one symbol, a short period): those groups are ordered by
      // their first kWideBigSyms symbols here and go on through the deepening like every other repeat
      RC(wide_ensure(c, &G->w_same, &G->w_cap_same, (size_t)nrec + 16));
      hipLaunchKernelGGL(k_wide_same16, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, (const Rec16 *)h, nrec, G->w_same);
      KCHECK();
      PosRec16 ph; ph.h = h;
      u32 nbig = 0;
      RC(wide_big_syms(G, G->w_same, nrec, ph, k, &nbig));
      G->w_depth = std::min<u32>(G->w_depth, kWideBigSyms);
      c->h_words[10] = 0; c->h_words[12] = std::max<u32>(c->h_words[12], 1u);
    }
    return E_OK;
  }();
  if (local_rc != E_OK && local_rc != E_TOOBIG && local_rc != E_ALLOC) return local_rc;     // HIP / transport faults: as before
  arena_release(c, mk);
  c->stats.level_tied[0] = c->h_words[11];
  const bool mine_ok = local_rc == E_OK && c->h_words[10] == 0 && c->h_words[12] == 0;
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(agree(local_rc));
  RC(gather_counts(cm, mine_ok ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != n) { set_err("wide global order: %llu of %llu positions selected", (unsigned long long)tot, (unsigned long long)n); return E_HIP; }
  if (ngood != (uint64_t)P && !G->no_wide_deepen) {
    // windows repeat beyond what the symbol compares settle: rank look-ups (wide_deepen)
    bool deep_ok = false;
    RC(wide_deepen(G, k, nrec, pre, all, &deep_ok));
    if (deep_ok) ngood = (uint64_t)P;
  }
  if (ngood != (uint64_t)P) {
    set_err("wide global mode: some %u-symbol window of the text repeats; texts of 2^32 bytes and more are only built when all windows "
            "are distinct (no recursion with 64-bit positions) [rank %d: %u records, %u tied, %u equal windows, oversized group %u]",
            k.W, me, nrec, c->h_words[11], c->h_words[12], c->h_words[10]);
    return E_TOOBIG;
  }
  G->shard_first = (int64_t)pre; G->shard_count = (int64_t)nrec; G->shard_ptr = nullptr;
  long long corrupt = 0;
  if (dbg_num("wide_corrupt", &corrupt)) {
    // test hook for the verifier: 1 = swap two neighbours of the last rank's shard, 2 = put one position out of range
    const char e[2] = {(char)('0' + corrupt), 0};
    if (me == P - 1 && nrec >= 2 && (e[0] == '1' || e[0] == '2')) {
      u64 two[2];
      HIPC(hipMemcpy(two, G->w_shard + nrec / 2, 16, hipMemcpyDeviceToHost));
      if (e[0] == '1') std::swap(two[0], two[1]); else two[0] = n;
      HIPC(hipMemcpy(G->w_shard + nrec / 2, two, 16, hipMemcpyHostToDevice));
    }
  }
  c->stats.text_sort_state = 1;
  c->stats.level_n[0] = (int64_t)n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = 5;
  G->gs.local_from_level = -1;
  RC(build_end(c));
  return E_OK;
}

static int gbuild_inner(dc3hip_gctx *G) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const int64_t n = G->total_n;
  if (G->wide) return gbuild_wide(G);
  c->n = n;
  c->arena_off = 0;
  RC(gagree_placement(G));
  RC(ensure_arena(c, arena_requirement(n)));      // a rank may end up with a whole level's key range: the full budget
  RC(build_begin(c));
  // 1. the text, replicated: all-gather of the ranks' blocks (n x (P-1)/P bytes in per rank)
  {
    size_t roff[kMaxRanks], rbytes[kMaxRanks];
    for (int r = 0; r < P; r++) { int64_t o, l; block_of(n, P, r, &o, &l); roff[r] = (size_t)o; rbytes[r] = (size_t)l; }
    RC(cm->all_gather_v(c->d_text + roff[me], rbytes[me], c->d_text, roff, rbytes, c->stream));
    HIPC(hipMemsetAsync(c->d_text + n, 0, 64, c->stream));
  }
  G->gs.local_from_level = -1;
  if (n <= 2 || (P == 1 && !G->force_dist) || (u64)n <= (u64)G->local_max) {
    if (n >= 1) RC(build_core(c));
    int64_t off, len; block_of(n, P, me, &off, &len);
    G->shard_first = off; G->shard_count = len; G->shard_ptr = c->d_sa + off;
    G->gs.local_from_level = 0;
  } else {
    u32 sigma = 0;
    RC(build_alphabet(c, &sigma));
    SymU8 S; S.t = c->d_text; S.code = c->d_code; S.m = (u32)n;
    bool done = false;
    RC(gtext_order(G, S, sigma, &done));
    if (!done) RC(glevel<SymU8>(G, S, (u32)n, sigma, 0, nullptr, G_TOP));
  }
  RC(build_end(c));
  return E_OK;
}

static int gbuild(dc3hip_gctx *G) {
  if (!G || !G->c || !G->comm) { set_err("invalid global context"); return E_ARGS; }
  if (!G->text_set) { set_err("no text block set in this global context"); return E_ARGS; }
  G->built = false;
  // the calling thread may be a fresh one (loopback ranks, a host program's worker) whose current device is 0: every
  // allocation of the build (ensure_arena comes before build_begin) must land on the rank's own device
  HIPC(hipSetDevice(G->c->device));
  GComm *cm = G->comm;
  cm->comm_ms = 0; cm->bytes_in = cm->bytes_out = 0;
  cm->work_ms = 0; cm->link_ms = 0; cm->ncoll = 0;
  memset(&G->gs, 0, sizeof(G->gs));
  const auto t0 = std::chrono::steady_clock::now();
  cm->device_enter();
  const int rc = gbuild_inner(G);
  if (rc == E_OK && G->c->stream) (void)hipStreamSynchronize(G->c->stream);     // (the rank's last kernels are its own work)
  cm->device_leave();
  if (rc != E_OK) { snprintf(G->err, sizeof(G->err), "%s", g_err); cm->abort_all(); cm->leave_failed(); return rc; }
  G->gs.struct_size = (int32_t)sizeof(dc3hip_gstats);
  G->gs.nranks = cm->nranks; G->gs.rank = cm->rank;
  G->gs.total_n = G->total_n; G->gs.shard_first = G->shard_first; G->gs.shard_count = G->shard_count;
  G->gs.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  G->gs.comm_ms = cm->comm_ms; G->gs.comm_bytes_out = (int64_t)cm->bytes_out; G->gs.comm_bytes_in = (int64_t)cm->bytes_in;
  G->gs.device_ms = G->c->stats.build_ms;
  G->gs.work_ms = cm->work_ms; G->gs.link_ms = cm->link_ms; G->gs.collectives = (int64_t)cm->ncoll;
  G->gs.levels = G->c->stats.levels;
  G->gs.text_order = G->c->stats.text_sort_state == 1 ? 1 : 0;
  G->built = true;
  return E_OK;
}

static void gctx_env(dc3hip_gctx *G) {
  if (const char *e = getenv("DC3HIP_GLOBAL_LOCAL_MAX")) { const long long v = atoll(e); if (v >= 0) G->local_max = (u32)std::min<long long>(v, 0x7fffffffll); }
  // (test switches: DC3HIP_DEBUG, dc3_host_core.hpp)
  G->no_text_order = dbg_on("global_no_text_order"); G->force_dist = dbg_on("global_force_dist");
  G->route = !dbg_on("global_no_route"); G->no_select = dbg_on("global_no_select");
  G->no_wide_msd = dbg_on("no_wide_msd"); G->no_wide_deepen = dbg_on("no_wide_deepen");
  { long long v; if (dbg_num("wide_msd_min", &v) && v >= 0) { G->wide_msd_min = (u64)v; G->wide_msd_forced = true; } }
}

// the rank's device context: a full one (text, SA, arena for max_total_n) — or, in wide mode, a minimal one (stream,
// scratch words, a small arena for the sort's tables) next to the wide text buffer
static int gctx_make_ctx(dc3hip_gctx *G, int device, int64_t max_total_n) {
  bool force_wide = false;
  force_wide = dbg_on("global_force_wide");
  G->wide = force_wide || max_total_n > DC3HIP_MAX_N;
  // (use_vm = false: a rank's buffers are what RCCL and peer copies read and write — plain hipMalloc, as every multi-rank
  //  test so far ran; the reserve + commit buffers of round 6 are for the single-device contexts)
  if (!G->wide) return ctx_create_impl(&G->c, device, max_total_n, nullptr, false);
  if (max_total_n > ((int64_t)1 << 40)) { set_err("n=%lld exceeds 2^40", (long long)max_total_n); return E_TOOBIG; }
  RC(ctx_create_impl(&G->c, device, 0, nullptr, false));
  HIPC(hipSetDevice(G->c->device));
  HIPC(hipMalloc(&G->w_text, (size_t)max_total_n + 64));
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI of the global mode
// ---------------------------------------------------------------------------------------------
extern "C" {

// wide_ensure()'s growth rule (elements)
static size_t wide_slack(size_t need) { return need + need / 16 + 1024; }
int32_t dc3hip_global_plan(int64_t total_n, int32_t nranks, dc3hip_gplan *out) {
  if (!out || total_n < 0 || nranks < 1 || nranks > kMaxRanks) { set_err("dc3hip_global_plan: invalid arguments (1 <= nranks <= %d)", kMaxRanks); return E_ARGS; }
  if (total_n > ((int64_t)1 << 40)) { set_err("n=%lld exceeds 2^40", (long long)total_n); return E_TOOBIG; }
  memset(out, 0, sizeof(*out));
  out->struct_size = (int32_t)sizeof(*out);
  out->total_n = total_n; out->nranks = nranks;
  out->hbm_bytes = 288000000000ll;
  const size_t n = (size_t)total_n;
  const size_t share = (n + (size_t)nranks - 1) / (size_t)nranks;
  const size_t nrec = nranks == 1 ? n : std::min(n, share + share / 8 + 4096);
  out->records_per_rank = (int64_t)nrec;
  out->wide = total_n > DC3HIP_MAX_N ? 1 : 0;
  const size_t small = 256 * 4 + 256 * 2 + 64 * 4 + 4096 * 4 + 3 * DC3HIP_MAX_LEVELS * 8;     // d_present, d_code, d_words, d_xcdmon, d_trace
  if (!out->wide) {
    // a rank is a whole single-device context of the text (gctx_make_ctx): text, SA, and the arena at arena_requirement()
    out->text_bytes = (int64_t)(n + 64);
    out->context_bytes = (int64_t)((n + 16) * 4 + small);
    out->arena_bytes = (int64_t)arena_requirement(total_n);
    out->peak_bytes = out->text_bytes + out->context_bytes + out->arena_bytes;
    return E_OK;
  }
  out->text_bytes = (int64_t)(n + 64);
  out->context_bytes = (int64_t)(64 + 64 + small);
  out->arena_bytes = (int64_t)((size_t)256 << 20);                                            // gbuild_wide: ensure_arena(256 MiB)
  // ordering (gbuild_wide): two 16-byte record arrays + the 8-byte shard (the 16-byte LSD form: the larger of the two forms;
  // the bucket ordering holds two 8-byte word arrays, the shard and a flag byte per word)
  const size_t lsd = 2 * 16 * wide_slack(nrec + 16) + 8 * wide_slack(nrec + 16);
  const size_t msd = 2 * 16 * wide_slack(nrec / 2 + 16) + 8 * wide_slack(nrec + 16) + wide_slack(nrec + 16);
  out->order_bytes = (int64_t)std::max(lsd, msd);
  // deepening (wide_deepen): the record arrays are released; shard and flags stay, one rank's shard at a time (8 bytes per
  // suffix of the largest share), the inverse (8 n), the flags of the whole order (n) and of the rank's range, the group
  // starts of the rank's range (4 bytes per entry, only while a group beyond kWideTieBig members is being ordered)
  out->deepen_bytes = (int64_t)(8 * wide_slack(nrec + 16) + wide_slack(nrec + 16) + 8 * wide_slack(nrec + 16) + 8 * wide_slack(n + 16) +
                                wide_slack(n + 16) + wide_slack(nrec + 16) + wide_slack((nrec + 16) * 4));
  out->big_group_bytes_per_member = 48;                                                       // wide_big_collect: two records, position, slot, group
  out->peak_bytes = out->text_bytes + out->context_bytes + out->arena_bytes + std::max(out->order_bytes, out->deepen_bytes);
  return E_OK;
}

int32_t dc3hip_global_loopback_create(dc3hip_gctx **ranks, int32_t P, int32_t device, int64_t max_total_n) {
  if (!ranks || P < 1 || P > kMaxRanks || max_total_n < 0) { set_err("dc3hip_global_loopback_create: invalid arguments (1 <= P <= %d)", kMaxRanks); return E_ARGS; }
  for (int r = 0; r < P; r++) ranks[r] = nullptr;
  // device == DC3HIP_DEVICE_SPREAD: rank r on device r % (visible devices) — one process drives all GPUs of the node,
  // peer copies (xGMI where peer access exists, staged through the host otherwise) are the transport
  int ndev = 1;
  if (device == DC3HIP_DEVICE_SPREAD) {
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err("no HIP device visible"); return E_HIP; }
    for (int a = 0; a < ndev; a++) {
      if (hipSetDevice(a) != hipSuccess) continue;
      for (int b = 0; b < ndev; b++) if (a != b) (void)hipDeviceEnablePeerAccess(b, 0);   // (already enabled / unsupported: ignored)
    }
    (void)hipGetLastError();
  }
  auto world = std::make_shared<LoopWorld>(P);
  std::vector<dc3hip_gctx *> made;
  for (int r = 0; r < P; r++) {
    dc3hip_gctx *G = new (std::nothrow) dc3hip_gctx();
    if (!G) { set_err("host allocation failed"); for (auto *g : made) dc3hip_global_destroy(g); return E_ALLOC; }
    made.push_back(G);
    const int rc = gctx_make_ctx(G, device == DC3HIP_DEVICE_SPREAD ? r % ndev : device, max_total_n);
    if (rc != E_OK) { for (auto *g : made) dc3hip_global_destroy(g); return rc; }
    LoopComm *lc = new LoopComm(); lc->rank = r; lc->nranks = P; lc->w = world;
    (void)dbg_real("global_link_gbps", &lc->model_link_GBps);
    G->comm = lc; G->max_total = max_total_n;
    gctx_env(G);
  }
  world->one_device = device != DC3HIP_DEVICE_SPREAD || ndev == 1;
  world->token = world->one_device && dbg_on("global_device_token");
  for (int r = 0; r < P; r++) { ranks[r] = made[(size_t)r]; made[(size_t)r]->group = made; }
  return E_OK;
}

int32_t dc3hip_rccl_unique_id(uint8_t *id128) {
  if (!id128) { set_err("id is NULL"); return E_ARGS; }
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return E_HIP;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return E_OK;
}

int32_t dc3hip_rccl_library_path(char *buf, int32_t len, int32_t *was_already_mapped) {
  if (!buf || len < 2) { set_err("buffer is NULL or too short"); return E_ARGS; }
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return E_HIP;
  Dl_info di;
  if (!dladdr(reinterpret_cast<const void *>(g_rccl.CommInitRank), &di) || !di.dli_fname) { set_err("dladdr(ncclCommInitRank) failed"); return E_HIP; }
  snprintf(buf, (size_t)len, "%s", di.dli_fname);
  if (was_already_mapped) *was_already_mapped = g_rccl.preloaded ? 1 : 0;
  return E_OK;
}

int32_t dc3hip_global_rccl_create(dc3hip_gctx **out, const uint8_t *id128, int32_t rank, int32_t nranks, int32_t device,
                                  int64_t max_total_n) {
  if (!out || !id128 || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks || max_total_n < 0) {
    set_err("dc3hip_global_rccl_create: invalid arguments (1 <= nranks <= %d)", kMaxRanks); return E_ARGS;
  }
  *out = nullptr;
  { std::lock_guard<std::mutex> lk(g_rccl_mu); if (!g_rccl.load()) return E_HIP; }
  dc3hip_gctx *G = new (std::nothrow) dc3hip_gctx();
  if (!G) { set_err("host allocation failed"); return E_ALLOC; }
  int rc = gctx_make_ctx(G, device, max_total_n);
  if (rc != E_OK) { dc3hip_global_destroy(G); return rc; }
  RcclComm *rcm = new RcclComm(); rcm->rank = rank; rcm->nranks = nranks;
  G->comm = rcm; G->max_total = max_total_n;
  rc = [&]() -> int {
    HIPC(hipSetDevice(G->c->device));
    HIPC(hipMalloc(&rcm->d_small, RcclComm::kSmall * (size_t)(nranks + 1)));
    ncclUniqueId id; memcpy(&id, id128, 128);
    NCCLC(g_rccl.CommInitRank(&rcm->comm, nranks, id, rank));
    return E_OK;
  }();
  if (rc != E_OK) { dc3hip_global_destroy(G); return rc; }
  gctx_env(G);
  *out = G;
  return E_OK;
}

int32_t dc3hip_global_host_create(dc3hip_gctx **out, const dc3hip_host_transport *t, int32_t rank, int32_t nranks,
                                  int32_t device, int64_t max_total_n) {
  if (!out || !t || !t->all_to_all_v || !t->all_gather_v || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks ||
      max_total_n < 0) {
    set_err("dc3hip_global_host_create: invalid arguments (1 <= nranks <= %d)", kMaxRanks); return E_ARGS;
  }
  *out = nullptr;
  dc3hip_gctx *G = new (std::nothrow) dc3hip_gctx();
  if (!G) { set_err("host allocation failed"); return E_ALLOC; }
  const int rc = gctx_make_ctx(G, device, max_total_n);
  if (rc != E_OK) { dc3hip_global_destroy(G); return rc; }
  HostComm *hc = new HostComm(); hc->rank = rank; hc->nranks = nranks; hc->t = *t;
  G->comm = hc; G->max_total = max_total_n;
  gctx_env(G);
  *out = G;
  return E_OK;
}

void dc3hip_global_destroy(dc3hip_gctx *G) {
  if (!G) return;
  if (G->c) { (void)hipSetDevice(G->c->device); if (G->c->stream) (void)hipStreamSynchronize(G->c->stream); }
  delete G->comm;
  if (G->w_text) (void)hipFree(G->w_text);
  if (G->w_ra) (void)hipFree(G->w_ra);
  if (G->w_rb) (void)hipFree(G->w_rb);
  if (G->w_shard) (void)hipFree(G->w_shard);
  if (G->w_same) (void)hipFree(G->w_same);
  if (G->w_sa_all) (void)hipFree(G->w_sa_all);
  if (G->w_isa) (void)hipFree(G->w_isa);
  if (G->w_eq_all) (void)hipFree(G->w_eq_all);
  if (G->w_eq2) (void)hipFree(G->w_eq2);
  if (G->w_aux) (void)hipFree(G->w_aux);
  if (G->w_aux2) (void)hipFree(G->w_aux2);
  if (G->c) dc3hip_ctx_destroy(G->c);
  delete G;
}

static int gctx_set_total(dc3hip_gctx *G, int64_t total_n, int64_t *off, int64_t *len) {
  if (!G || total_n < 0) { set_err("invalid arguments"); return E_ARGS; }
  if (total_n > G->max_total) { set_err("n=%lld exceeds the context capacity %lld", (long long)total_n, (long long)G->max_total); return E_ARGS; }
  block_of(total_n, G->comm->nranks, G->comm->rank, off, len);
  G->total_n = total_n; G->built = false;
  return E_OK;
}

int32_t dc3hip_global_block(dc3hip_gctx *G, int64_t total_n, int64_t *offset, int64_t *length) {
  if (!G || !offset || !length || total_n < 0) { set_err("invalid arguments"); return E_ARGS; }
  block_of(total_n, G->comm->nranks, G->comm->rank, offset, length);
  return E_OK;
}

int32_t dc3hip_global_set_text_block(dc3hip_gctx *G, const uint8_t *block, int64_t total_n) {
  int64_t off, len;
  RC(gctx_set_total(G, total_n, &off, &len));
  if (!block && len > 0) { set_err("block is NULL"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(hipSetDevice(c->device));
  if (len > 0) HIPC(hipMemcpyAsync(gtext(G) + off, block, (size_t)len, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  G->text_set = true;
  return E_OK;
}

int32_t dc3hip_global_generate(dc3hip_gctx *G, int64_t total_n, uint64_t seed, int32_t kind) {
  int64_t off, len;
  RC(gctx_set_total(G, total_n, &off, &len));
  if (kind < 0 || kind > 2) { set_err("unknown generator kind %d", kind); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(hipSetDevice(c->device));
  if (len > 0) {
    // only this rank's block: the others arrive by the all-gather of the build
    hipLaunchKernelGGL(k_generate, dim3(grid_for(c, (u64)len / 8 + 1)), dim3(kBlock), 0, c->stream, gtext(G) + off, (u64)len,
                       (u64)seed, (int)kind, (u64)off);
    KCHECK();
  }
  HIPC(hipStreamSynchronize(c->stream));
  G->text_set = true;
  return E_OK;
}

int32_t dc3hip_global_build(dc3hip_gctx *G) { return gbuild(G); }

// loopback convenience: run the P ranks of a group on P host threads and wait for all of them.
// The threads are PERSISTENT (one process-wide set, started on demand, never joined): a round-4 hunt found the host heap
// damaged — use after free, always noticed by the main thread inside group creation / destruction — in a few per cent of
// short processes that built many loopback groups with fresh std::threads per build; nothing in this library frees what
// those threads touch, but every HIP call makes per-thread runtime state that dies with its thread while the streams it
// worked on live on.  Rank threads that never exit take that pattern away (profiles/r04u_fresh_process_crash_hunt.md).
struct LoopPool {
  std::mutex run_mu;                      // one group's build at a time through the pool
  std::mutex mu; std::condition_variable cv_work, cv_done;
  std::function<void()> job[kMaxRanks]; bool has[kMaxRanks] = {};
  int started = 0, pending = 0;
  void worker(int i) {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return has[i]; });
      std::function<void()> f = std::move(job[i]);
      has[i] = false;
      lk.unlock();
      f();
      lk.lock();
      if (--pending == 0) cv_done.notify_all();
    }
  }
  void run(int P, const std::function<void(int)> &f) {
    std::lock_guard<std::mutex> one(run_mu);
    std::unique_lock<std::mutex> lk(mu);
    while (started < P) { const int i = started++; std::thread([this, i] { worker(i); }).detach(); }
    for (int r = 0; r < P; r++) { job[r] = [&f, r] { f(r); }; has[r] = true; }
    pending = P;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};
static LoopPool *loop_pool() { static LoopPool *p = new LoopPool(); return p; }       // (leaked on purpose: its threads never exit)

int32_t dc3hip_global_loopback_build(dc3hip_gctx **ranks, int32_t P) {
  if (!ranks || P < 1 || P > kMaxRanks) { set_err("invalid arguments"); return E_ARGS; }
  for (int r = 0; r < P; r++) if (!ranks[r] || ranks[r]->comm->nranks != P) { set_err("not a loopback group of %d ranks", P); return E_ARGS; }
  ranks[0]->comm->reset_all();         // a failure of an earlier build no longer poisons the group
  std::vector<int> rcs((size_t)P, E_OK);
  loop_pool()->run(P, [&](int r) { rcs[(size_t)r] = gbuild(ranks[r]); });
  for (int r = 0; r < P; r++)
    if (rcs[(size_t)r] != E_OK && strstr(ranks[r]->err, "another rank failed") == nullptr) { set_err("rank %d: %s", r, ranks[r]->err); return rcs[(size_t)r]; }
  for (int r = 0; r < P; r++) if (rcs[(size_t)r] != E_OK) { set_err("rank %d: %s", r, ranks[r]->err); return rcs[(size_t)r]; }
  return E_OK;
}

int32_t dc3hip_global_shard(dc3hip_gctx *G, int64_t *first, int64_t *count) {
  if (!G || !first || !count) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  *first = G->shard_first; *count = G->shard_count;
  return E_OK;
}

int32_t dc3hip_global_get_shard_i64(dc3hip_gctx *G, int64_t *out) {
  if (!G || (!out && G->shard_count > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(hipSetDevice(c->device));
  if (G->shard_count == 0) return E_OK;
  if (G->wide) {
    HIPC(hipMemcpyAsync(out, G->w_shard, (size_t)G->shard_count * 8, hipMemcpyDefault, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return E_OK;
  }
  c->arena_off = 0;
  const size_t piece = std::min<size_t>((size_t)G->shard_count, std::max<size_t>(c->arena_bytes / 8, 1));
  int64_t *tmp = reinterpret_cast<int64_t *>(c->arena);
  for (size_t off = 0; off < (size_t)G->shard_count; off += piece) {
    const size_t cnt = std::min(piece, (size_t)G->shard_count - off);
    hipLaunchKernelGGL(k_widen_off, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, G->shard_ptr + off, tmp, (u32)cnt);
    KCHECK();
    HIPC(hipMemcpyAsync(out + off, tmp, cnt * 8, hipMemcpyDefault, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  return E_OK;
}

int32_t dc3hip_global_get_shard_u32(dc3hip_gctx *G, uint32_t *out) {
  if (!G || (!out && G->shard_count > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  if (G->wide) { set_err("this global context holds 64-bit positions: use dc3hip_global_get_shard_i64"); return E_TOOBIG; }
  HIPC(hipSetDevice(c->device));
  if (G->shard_count > 0) HIPC(hipMemcpyAsync(out, G->shard_ptr, (size_t)G->shard_count * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
}

// order-sensitive checksum of this rank's shard with GLOBAL indices: the sum over all ranks equals
// dc3hip_ctx_sa_checksum of a single-device build of the same text
int32_t dc3hip_global_shard_checksum(dc3hip_gctx *G, uint64_t *out) {
  if (!G || !out) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(hipSetDevice(c->device));
  u64 *acc = reinterpret_cast<u64 *>(c->d_words + 16);
  HIPC(hipMemsetAsync(acc, 0, sizeof(u64), c->stream));
  if (G->shard_count > 0 && G->wide) {       // (its own mixing: there is no single-device array to compare with)
    hipLaunchKernelGGL(k_wide_checksum, dim3(grid_for(c, G->shard_count)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_shard,
                       (u32)G->shard_count, (u64)G->shard_first, acc);
    KCHECK();
  } else if (G->shard_count > 0) {
    hipLaunchKernelGGL(k_checksum_off, dim3(grid_for(c, G->shard_count)), dim3(kBlock), 0, c->stream, G->shard_ptr,
                       (u32)G->shard_count, (u64)G->shard_first, acc);
    KCHECK();
  }
  HIPC(hipMemcpyAsync(c->h_words + 16, acc, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  memcpy(out, c->h_words + 16, sizeof(u64));
  return E_OK;
}

int32_t dc3hip_global_stats(dc3hip_gctx *G, dc3hip_gstats *out, dc3hip_stats *ctx_stats) {
  if (!G || !out) { set_err("invalid arguments"); return E_ARGS; }
  *out = G->gs;
  out->struct_size = (int32_t)sizeof(dc3hip_gstats);
  if (ctx_stats) { *ctx_stats = G->c->stats; ctx_stats->struct_size = (int32_t)sizeof(dc3hip_stats); }
  return E_OK;
}

// Collective check of a wide-mode result (every rank calls it): positions in range, every shard entry's suffix strictly
// smaller than its successor's — across the rank boundaries too — and the shard sizes add up to n.  Returns 0 when the
// concatenated shards are the suffix array, else the reference sufcheck's codes (-2 range, -3 order), the same on all
// ranks; < -10 = the check itself failed (dc3hip error code - 10).
int32_t dc3hip_global_sufcheck(dc3hip_gctx *G) {
  if (!G || !G->comm) { set_err("invalid arguments"); return E_ARGS - 10; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS - 10; }
  if (!G->wide) { set_err("dc3hip_global_sufcheck: only for contexts with 64-bit positions (fetch the shards and use dc3hip_ctx_sufcheck)"); return E_ARGS - 10; }
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  int verdict = 0;
  auto run = [&]() -> int {
    HIPC(hipSetDevice(c->device));
    // first entries of all shards (a rank with an empty shard passes ~0 and is skipped)
    u64 first_mine = ~0ull, firsts[kMaxRanks], counts[kMaxRanks];
    if (G->shard_count > 0) {
      void *fp = nullptr;
      RC(stage_d2h(c, G->w_shard, 8, &fp));
      memcpy(&first_mine, fp, 8);
    }
    RC(cm->all_gather_host(&first_mine, firsts, 8));
    const u64 cnt_mine = (u64)G->shard_count;
    RC(cm->all_gather_host(&cnt_mine, counts, 8));
    u64 tot = 0, next_first = ~0ull;
    for (int r = 0; r < P; r++) tot += counts[r];
    for (int r = me + 1; r < P; r++) if (counts[r]) { next_first = firsts[r]; break; }
    u32 sigma = 0;
    { HIPC(hipMemcpyAsync(c->h_words + 1, c->d_words + 1, sizeof(u32), hipMemcpyDeviceToHost, c->stream)); HIPC(hipStreamSynchronize(c->stream)); sigma = c->h_words[1]; }
    WideKey k; u32 ibits = 0;
    RC(wide_key(G, sigma, &k, &ibits));
    HIPC(hipMemsetAsync(c->d_words + 20, 0, sizeof(u32), c->stream));
    if (G->shard_count > 0 && G->w_isa_valid) {
      // a deepened order: linear-time check against its own inverse (symbol compares would be as deep as the repeats are long)
      hipLaunchKernelGGL(k_wide_check_isa, dim3(grid_for(c, G->shard_count)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_shard,
                         (u32)G->shard_count, (u64)G->shard_first, next_first, k, (const u64 *)G->w_isa, c->d_words + 20);
      KCHECK();
    } else if (G->shard_count > 0) {
      hipLaunchKernelGGL(k_wide_check, dim3(grid_for(c, G->shard_count)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_shard,
                         (u32)G->shard_count, next_first, k, std::max<u32>(4 * kWideWindowDeep, G->w_depth), c->d_words + 20);
      KCHECK();
    }
    HIPC(hipMemcpyAsync(c->h_words + 20, c->d_words + 20, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    u64 err = c->h_words[20], errs[kMaxRanks];
    if (tot != (u64)G->total_n) err = std::max<u64>(err, 2);
    {
      // A deepened order is checked against the rank's OWN copy of the inverse: each slice of it is verified by its owner
      // against the real shard (isa[p] == index, above), and a rank reads other ranks' slices for the neighbour test — so
      // all copies must be the same array.  Order-sensitive checksum of the whole inverse on every rank, compared.
      u64 have = G->w_isa_valid ? 1ull : 0ull, sum = 0, haves[kMaxRanks], sums[kMaxRanks];
      RC(cm->all_gather_host(&have, haves, 8));
      bool any = false, all_have = true;
      for (int r = 0; r < P; r++) { any = any || haves[r]; all_have = all_have && haves[r]; }
      if (any && !all_have) err = std::max<u64>(err, 3);
      if (all_have) {
        u64 *acc = reinterpret_cast<u64 *>(c->d_words + 22);
        HIPC(hipMemsetAsync(acc, 0, 8, c->stream));
        const u64 nn = (u64)G->total_n;
        for (u64 off = 0; off < nn; off += (u64)1 << 30) {
          const u32 cnt = (u32)std::min<u64>((u64)1 << 30, nn - off);
          hipLaunchKernelGGL(k_wide_checksum, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_isa + off, cnt, off, acc);
          KCHECK();
        }
        void *sp = nullptr;
        RC(stage_d2h(c, acc, 8, &sp));
        memcpy(&sum, sp, 8);
        RC(cm->all_gather_host(&sum, sums, 8));
        for (int r = 0; r < P; r++) if (sums[r] != sums[0]) err = std::max<u64>(err, 3);
      }
    }
    RC(cm->all_gather_host(&err, errs, 8));
    u64 worst = 0;
    for (int r = 0; r < P; r++) worst = std::max(worst, errs[r]);
    verdict = worst == 0 ? 0 : -(int)worst;
    return E_OK;
  };
  const int rc = run();
  if (rc != E_OK) { snprintf(G->err, sizeof(G->err), "%s", g_err); cm->abort_all(); cm->leave_failed(); return rc - 10; }
  return verdict;
}

const char *dc3hip_global_last_error(dc3hip_gctx *G) { return G ? G->err : ""; }
// Transport self-test (a COLLECTIVE): a ragged all_to_all_v, a ragged all_gather_v and a host all-gather of known bytes
// through this group's transport, every byte checked on every rank.  0 = all three delivered exactly what was sent.
static inline uint8_t selftest_byte(int from, int to, size_t k) {
  uint64_t x = ((uint64_t)(from + 1) << 40) ^ ((uint64_t)(to + 1) << 20) ^ (uint64_t)k;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
  return (uint8_t)x;
}
static int gselftest(dc3hip_gctx *G) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  HIPC(hipSetDevice(c->device));
  auto a2a_len = [](int from, int to) -> size_t { return 1000 + 37 * (size_t)from + 101 * (size_t)to + (size_t)((from * 7 + to * 3) % 11); };
  auto ag_len = [](int from) -> size_t { return 5000 + 313 * (size_t)from; };
  // ---- all_to_all_v
  size_t soff[kMaxRanks], sb[kMaxRanks], roff[kMaxRanks], rb[kMaxRanks], stot = 0, rtot = 0;
  for (int r = 0; r < P; r++) { soff[r] = stot; sb[r] = a2a_len(me, r); stot += sb[r]; roff[r] = rtot; rb[r] = a2a_len(r, me); rtot += rb[r]; }
  std::vector<uint8_t> hs(stot), hr(rtot);
  for (int r = 0; r < P; r++) for (size_t k = 0; k < sb[r]; k++) hs[soff[r] + k] = selftest_byte(me, r, k);
  unsigned char *ds = nullptr, *dr = nullptr;
  HIPC(hipMalloc(&ds, stot + 64)); 
  if (hipMalloc(&dr, rtot + 64) != hipSuccess) { (void)hipFree(ds); set_err("self-test: no device memory"); return E_ALLOC; }
  int rc = [&]() -> int {
    HIPC(hipMemcpyAsync(ds, hs.data(), stot, hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemsetAsync(dr, 0xEE, rtot, c->stream));
    RC(cm->all_to_all_v(ds, soff, sb, dr, roff, rb, c->stream));
    HIPC(hipMemcpyAsync(hr.data(), dr, rtot, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int r = 0; r < P; r++)
      for (size_t k = 0; k < rb[r]; k++)
        if (hr[roff[r] + k] != selftest_byte(r, me, k)) {
          set_err("transport self-test: all_to_all_v delivered a wrong byte (from rank %d to rank %d, offset %zu of %zu)", r, me, k, rb[r]);
          return E_HIP;
        }
    return E_OK;
  }();
  (void)hipFree(ds); (void)hipFree(dr);
  RC(rc);
  // ---- all_gather_v
  size_t goff[kMaxRanks], gb[kMaxRanks], gtot = 0;
  for (int r = 0; r < P; r++) { goff[r] = gtot; gb[r] = ag_len(r); gtot += gb[r]; }
  std::vector<uint8_t> hg(gtot);
  unsigned char *dg = nullptr;
  HIPC(hipMalloc(&dg, gtot + 64));
  rc = [&]() -> int {
    HIPC(hipMemsetAsync(dg, 0xEE, gtot, c->stream));
    for (size_t k = 0; k < gb[me]; k++) hg[goff[me] + k] = selftest_byte(me, 255, k);
    HIPC(hipMemcpyAsync(dg + goff[me], hg.data() + goff[me], gb[me], hipMemcpyHostToDevice, c->stream));
    RC(cm->all_gather_v(dg + goff[me], gb[me], dg, goff, gb, c->stream));
    HIPC(hipMemcpyAsync(hg.data(), dg, gtot, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int r = 0; r < P; r++)
      for (size_t k = 0; k < gb[r]; k++)
        if (hg[goff[r] + k] != selftest_byte(r, 255, k)) {
          set_err("transport self-test: all_gather_v delivered a wrong byte (block of rank %d seen by rank %d, offset %zu of %zu)", r, me, k, gb[r]);
          return E_HIP;
        }
    return E_OK;
  }();
  (void)hipFree(dg);
  RC(rc);
  // ---- host words
  uint64_t mine = 0x5eed000000000000ull + (uint64_t)me * 1000003ull, all[kMaxRanks];
  RC(cm->all_gather_host(&mine, all, sizeof(uint64_t)));
  for (int r = 0; r < P; r++)
    if (all[r] != 0x5eed000000000000ull + (uint64_t)r * 1000003ull) { set_err("transport self-test: all_gather_host delivered a wrong word for rank %d", r); return E_HIP; }
  return E_OK;
}
int32_t dc3hip_global_selftest(dc3hip_gctx *G, int32_t *transport_ranks) {
  if (!G || !G->c || !G->comm) { set_err("invalid global context"); return E_ARGS; }
  if (transport_ranks) *transport_ranks = G->comm->transport_ranks();
  const int rc = gselftest(G);
  if (rc != E_OK) { snprintf(G->err, sizeof(G->err), "%s", g_err); G->comm->abort_all(); G->comm->leave_failed(); }
  return rc;
}
const char *dc3hip_global_transport(dc3hip_gctx *G) { return (G && G->comm) ? G->comm->name() : ""; }

}  // extern "C"
