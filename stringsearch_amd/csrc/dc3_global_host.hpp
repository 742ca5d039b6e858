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

#include "dc3_global_comm.hpp"
#include "dc3_global_level.hpp"
#include "dc3_global_wide.hpp"

// ---------------------------------------------------------------------------------------------
// the build of one rank, group and context management, the C ABI of the global mode
// ---------------------------------------------------------------------------------------------
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
  HIPC(dc3_set_device(G->c->device));
  GComm *cm = G->comm;
  cm->comm_ms = 0; cm->bytes_in = cm->bytes_out = 0;
  cm->work_ms = 0; cm->link_ms = 0; cm->ncoll = 0;
  memset(&G->gs, 0, sizeof(G->gs));
  const auto t0 = std::chrono::steady_clock::now();
  cm->device_enter();
  int rc;
  try { rc = gbuild_inner(G); } catch (...) { rc = abi_exception(); }      // (a rank thread: the other ranks must be released, below)
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
  HIPC(dc3_set_device(G->c->device));
  HIPC(hipMalloc(&G->w_text, (size_t)max_total_n + 64));
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI of the global mode
// ---------------------------------------------------------------------------------------------
extern "C" {

// wide_ensure()'s growth rule (elements)
static size_t wide_slack(size_t need) { return need + need / 16 + 1024; }
int32_t dc3hip_global_plan(int64_t total_n, int32_t nranks, dc3hip_gplan *out) try {
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
} DC3_ABI_CATCH

int32_t dc3hip_global_loopback_create(dc3hip_gctx **ranks, int32_t P, int32_t device, int64_t max_total_n) try {
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
} DC3_ABI_CATCH

int32_t dc3hip_rccl_unique_id(uint8_t *id128) try {
  if (!id128) { set_err("id is NULL"); return E_ARGS; }
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return E_HIP;
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(id128, &id, 128);
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_rccl_library_path(char *buf, int32_t len, int32_t *was_already_mapped) try {
  if (!buf || len < 2) { set_err("buffer is NULL or too short"); return E_ARGS; }
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) return E_HIP;
  Dl_info di;
  if (!dladdr(reinterpret_cast<const void *>(g_rccl.CommInitRank), &di) || !di.dli_fname) { set_err("dladdr(ncclCommInitRank) failed"); return E_HIP; }
  snprintf(buf, (size_t)len, "%s", di.dli_fname);
  if (was_already_mapped) *was_already_mapped = g_rccl.preloaded ? 1 : 0;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_rccl_create(dc3hip_gctx **out, const uint8_t *id128, int32_t rank, int32_t nranks, int32_t device,
                                  int64_t max_total_n) try {
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
    HIPC(dc3_set_device(G->c->device));
    HIPC(hipMalloc(&rcm->d_small, RcclComm::kSmall * (size_t)(nranks + 1)));
    ncclUniqueId id; memcpy(&id, id128, 128);
    NCCLC(g_rccl.CommInitRank(&rcm->comm, nranks, id, rank));
    return E_OK;
  }();
  if (rc != E_OK) { dc3hip_global_destroy(G); return rc; }
  gctx_env(G);
  *out = G;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_host_create(dc3hip_gctx **out, const dc3hip_host_transport *t, int32_t rank, int32_t nranks,
                                  int32_t device, int64_t max_total_n) try {
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
} DC3_ABI_CATCH

void dc3hip_global_destroy(dc3hip_gctx *G) try {
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
} DC3_ABI_CATCH_VOID

static int gctx_set_total(dc3hip_gctx *G, int64_t total_n, int64_t *off, int64_t *len) {
  if (!G || total_n < 0) { set_err("invalid arguments"); return E_ARGS; }
  if (total_n > G->max_total) { set_err("n=%lld exceeds the context capacity %lld", (long long)total_n, (long long)G->max_total); return E_ARGS; }
  block_of(total_n, G->comm->nranks, G->comm->rank, off, len);
  G->total_n = total_n; G->built = false;
  return E_OK;
}

int32_t dc3hip_global_block(dc3hip_gctx *G, int64_t total_n, int64_t *offset, int64_t *length) try {
  if (!G || !offset || !length || total_n < 0) { set_err("invalid arguments"); return E_ARGS; }
  block_of(total_n, G->comm->nranks, G->comm->rank, offset, length);
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_set_text_block(dc3hip_gctx *G, const uint8_t *block, int64_t total_n) try {
  int64_t off, len;
  RC(gctx_set_total(G, total_n, &off, &len));
  if (!block && len > 0) { set_err("block is NULL"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(dc3_set_device(c->device));
  if (len > 0) HIPC(hipMemcpyAsync(gtext(G) + off, block, (size_t)len, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  G->text_set = true;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_generate(dc3hip_gctx *G, int64_t total_n, uint64_t seed, int32_t kind) try {
  int64_t off, len;
  RC(gctx_set_total(G, total_n, &off, &len));
  if (kind < 0 || kind > 2) { set_err("unknown generator kind %d", kind); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(dc3_set_device(c->device));
  if (len > 0) {
    // only this rank's block: the others arrive by the all-gather of the build
    hipLaunchKernelGGL(k_generate, dim3(grid_for(c, (u64)len / 8 + 1)), dim3(kBlock), 0, c->stream, gtext(G) + off, (u64)len,
                       (u64)seed, (int)kind, (u64)off);
    KCHECK();
  }
  HIPC(hipStreamSynchronize(c->stream));
  G->text_set = true;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_build(dc3hip_gctx *G) try { return gbuild(G); } DC3_ABI_CATCH

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
    while (started < P) { const int i = started; std::thread([this, i] { worker(i); }).detach(); started++; }   // (may throw: no rank thread to be had)
    for (int r = 0; r < P; r++) { job[r] = [&f, r] { f(r); }; has[r] = true; }
    pending = P;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};
static LoopPool *loop_pool() { static LoopPool *p = new LoopPool(); return p; }       // (leaked on purpose: its threads never exit)

int32_t dc3hip_global_loopback_build(dc3hip_gctx **ranks, int32_t P) try {
  if (!ranks || P < 1 || P > kMaxRanks) { set_err("invalid arguments"); return E_ARGS; }
  for (int r = 0; r < P; r++) if (!ranks[r] || ranks[r]->comm->nranks != P) { set_err("not a loopback group of %d ranks", P); return E_ARGS; }
  ranks[0]->comm->reset_all();         // a failure of an earlier build no longer poisons the group
  std::vector<int> rcs((size_t)P, E_OK);
  loop_pool()->run(P, [&](int r) { rcs[(size_t)r] = gbuild(ranks[r]); });
  for (int r = 0; r < P; r++)
    if (rcs[(size_t)r] != E_OK && strstr(ranks[r]->err, "another rank failed") == nullptr) { set_err("rank %d: %s", r, ranks[r]->err); return rcs[(size_t)r]; }
  for (int r = 0; r < P; r++) if (rcs[(size_t)r] != E_OK) { set_err("rank %d: %s", r, ranks[r]->err); return rcs[(size_t)r]; }
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_shard(dc3hip_gctx *G, int64_t *first, int64_t *count) try {
  if (!G || !first || !count) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  *first = G->shard_first; *count = G->shard_count;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_global_get_shard_i64(dc3hip_gctx *G, int64_t *out) try {
  if (!G || (!out && G->shard_count > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(dc3_set_device(c->device));
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
} DC3_ABI_CATCH

int32_t dc3hip_global_get_shard_u32(dc3hip_gctx *G, uint32_t *out) try {
  if (!G || (!out && G->shard_count > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  if (G->wide) { set_err("this global context holds 64-bit positions: use dc3hip_global_get_shard_i64"); return E_TOOBIG; }
  HIPC(dc3_set_device(c->device));
  if (G->shard_count > 0) HIPC(hipMemcpyAsync(out, G->shard_ptr, (size_t)G->shard_count * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
} DC3_ABI_CATCH

// order-sensitive checksum of this rank's shard with GLOBAL indices: the sum over all ranks equals
// dc3hip_ctx_sa_checksum of a single-device build of the same text
int32_t dc3hip_global_shard_checksum(dc3hip_gctx *G, uint64_t *out) try {
  if (!G || !out) { set_err("invalid arguments"); return E_ARGS; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS; }
  dc3hip_ctx *c = G->c;
  HIPC(dc3_set_device(c->device));
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
} DC3_ABI_CATCH

int32_t dc3hip_global_stats(dc3hip_gctx *G, dc3hip_gstats *out, dc3hip_stats *ctx_stats) try {
  if (!G || !out) { set_err("invalid arguments"); return E_ARGS; }
  *out = G->gs;
  out->struct_size = (int32_t)sizeof(dc3hip_gstats);
  if (ctx_stats) { *ctx_stats = G->c->stats; ctx_stats->struct_size = (int32_t)sizeof(dc3hip_stats); }
  return E_OK;
} DC3_ABI_CATCH

// Collective check of a wide-mode result (every rank calls it): positions in range, every shard entry's suffix strictly
// smaller than its successor's — across the rank boundaries too — and the shard sizes add up to n.  Returns 0 when the
// concatenated shards are the suffix array, else the reference sufcheck's codes (-2 range, -3 order), the same on all
// ranks; < -10 = the check itself failed (dc3hip error code - 10).
int32_t dc3hip_global_sufcheck(dc3hip_gctx *G) try {
  if (!G || !G->comm) { set_err("invalid arguments"); return E_ARGS - 10; }
  if (!G->built) { set_err("no suffix array built in this global context"); return E_ARGS - 10; }
  if (!G->wide) { set_err("dc3hip_global_sufcheck: only for contexts with 64-bit positions (fetch the shards and use dc3hip_ctx_sufcheck)"); return E_ARGS - 10; }
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  int verdict = 0;
  auto run = [&]() -> int {
    HIPC(dc3_set_device(c->device));
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
} DC3_ABI_CATCH_GLOBAL_SUFCHECK

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
  HIPC(dc3_set_device(c->device));
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
int32_t dc3hip_global_selftest(dc3hip_gctx *G, int32_t *transport_ranks) try {
  if (!G || !G->c || !G->comm) { set_err("invalid global context"); return E_ARGS; }
  if (transport_ranks) *transport_ranks = G->comm->transport_ranks();
  const int rc = gselftest(G);
  if (rc != E_OK) { snprintf(G->err, sizeof(G->err), "%s", g_err); G->comm->abort_all(); G->comm->leave_failed(); }
  return rc;
} DC3_ABI_CATCH
const char *dc3hip_global_transport(dc3hip_gctx *G) { return (G && G->comm) ? G->comm->name() : ""; }

}  // extern "C"
