// dc3_global_comm.hpp — the transports of the GLOBAL multi-GPU mode behind one small interface (GComm): RCCL (grouped
// ncclSend / ncclRecv over xGMI, one process per GPU; the entry points are resolved with dlopen / dlsym at first use), the
// host-staged transport (the caller's two collectives on pinned buffers) and the in-process loopback (P rank contexts in
// one process).  Part of the host driver of libdc3hip's global mode: included by dc3_global_host.hpp (overview there).
#pragma once

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

