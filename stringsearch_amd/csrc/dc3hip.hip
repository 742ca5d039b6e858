// dc3hip.hip — host driver + C ABI of libdc3hip.so (see include/dc3hip.h).
//
// Host side of the DC3/Skew recursion of crates/dc3/src/lib.rs:44-193, re-designed for MI355X:
//   * one context = one HIP stream + one device arena (no hipMalloc inside the recursion; the
//     reference allocates 4 Vecs per level, lib.rs:50-57);
//   * every level is a fixed sequence of streaming kernels (dc3_kernels.hip.hpp); the host only reads back a
//     few words per level: the number of distinct names that decides lib.rs:103 (recurse or not), and the
//     tie statistics that steer the ordering policy (which never affect the result);
//   * no CPU fallback of any kind: if HIP fails the call fails (-3).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <chrono>
#include <sys/mman.h>
#include <cmath>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>
#include <thread>
#include <string>

#include "../../include/dc3hip.h"
#include "dc3_kernels.hip.hpp"
#include "dc3_msd.hip.hpp"
#include "dc3_ssort.hip.hpp"
#include "dc3_wide_msd.hip.hpp"

using namespace dc3;

#define DC3HIP_VERSION_STR "dc3hip 0.2.0 (gfx950, HIP)"

#include "dc3_host_core.hpp"
#include "dc3_host_sort.hpp"
#include "dc3_host_order.hpp"
#include "dc3_host_level.hpp"

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char *dc3hip_version(void) { return DC3HIP_VERSION_STR; }
const char *dc3hip_last_error(void) { return g_err; }

int32_t dc3hip_hip_versions(int32_t *compiled, int32_t *runtime) try {
  int rt = 0;
  if (hipRuntimeGetVersion(&rt) != hipSuccess) { (void)hipGetLastError(); set_err("hipRuntimeGetVersion failed"); return E_HIP; }
  if (compiled) *compiled = (int32_t)HIP_VERSION;
  if (runtime) *runtime = rt;
  return (HIP_VERSION / 100000) == (rt / 100000) ? 1 : 0;
} DC3_ABI_CATCH
// once per process, from the first context: a runtime other than the one the library was compiled against is legal
// (same soname) but nothing this library's tests ran on; say so where a crash report would be read
static void warn_runtime_mismatch_once() {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  int32_t ct = 0, rt = 0;
  if (dc3hip_hip_versions(&ct, &rt) == 0 && !getenv("DC3HIP_QUIET"))
    std::fprintf(stderr, "dc3hip: compiled against HIP %d.%d.%d, running on HIP runtime %d.%d.%d (another libamdhip64 was mapped first)\n",
                 ct / 10000000, ct / 100000 % 100, ct % 100000, rt / 10000000, rt / 100000 % 100, rt % 100000);
}

int32_t dc3hip_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { set_err("hipGetDeviceCount failed"); return E_HIP; }
  return n;
} DC3_ABI_CATCH

int32_t dc3hip_device_synchronize(int32_t device) try {
  if (device >= 0) HIPC(dc3_set_device(device));
  HIPC(hipDeviceSynchronize());
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_device_info(int32_t device, char *arch, int32_t arch_len, int32_t *compute_units) try {
  if (device < 0) HIPC(hipGetDevice(&device));
  hipDeviceProp_t prop;
  HIPC(hipGetDeviceProperties(&prop, device));
  if (arch && arch_len > 0) { std::strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return E_OK;
} DC3_ABI_CATCH

static int ctx_create_impl(dc3hip_ctx **out, int32_t device, int64_t max_n, dc3hip_ctx *arena_from, bool use_vm = true);
int32_t dc3hip_ctx_create(dc3hip_ctx **out, int32_t device, int64_t max_n) try { return ctx_create_impl(out, device, max_n, nullptr); } DC3_ABI_CATCH
// one device buffer of a context: reserved + committed (DevBuf) from vm_min bytes on, else hipMalloc
static int ctx_big_alloc(dc3hip_ctx *c, DevBuf *vm, size_t bytes, void **out) {
  if (c->use_vm && bytes >= c->vm_min && devbuf_reserve(vm, c->device, bytes)) {
    if (devbuf_commit(vm, bytes)) { *out = vm->va; return E_OK; }
    devbuf_free(vm);
  }
  HIPC(hipMalloc(out, bytes));
  return E_OK;
}
// arena_from != nullptr: the new context works in that context's arena instead of allocating its own (the lender must not
// build meanwhile; dc3hip_ctx_build_partitions: the partitions are built one after the other in the parent's arena)
static int ctx_create_impl(dc3hip_ctx **out, int32_t device, int64_t max_n, dc3hip_ctx *arena_from, bool use_vm) {
  if (!out || max_n < 0) { set_err("dc3hip_ctx_create: invalid arguments"); return E_ARGS; }
  *out = nullptr;
  if (max_n > DC3HIP_MAX_N) { set_err("n=%lld exceeds DC3HIP_MAX_N", (long long)max_n); return E_TOOBIG; }
  int ndev = 0;
  HIPC(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { set_err("no HIP device visible (no CPU fallback exists)"); return E_HIP; }
  if (device < 0) HIPC(hipGetDevice(&device));
  if (device >= ndev) { set_err("device %d out of range (%d devices)", device, ndev); return E_ARGS; }
  warn_runtime_mismatch_once();
  dc3hip_ctx *c = new (std::nothrow) dc3hip_ctx();
  if (!c) { set_err("host allocation failed"); return E_ALLOC; }
  c->device = device; c->max_n = max_n;
  memset(&c->stats, 0, sizeof(c->stats));
  const char *prof = getenv("DC3HIP_PROFILE");
  c->profile = !(prof && prof[0] == '0');
  { const char *e = getenv("DC3HIP_TRACE"); c->trace = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_LEVEL_PHASES"); c->level_report = (e && e[0] == '1'); }
  // test / diagnosis switches (DC3HIP_DEBUG, dc3_host_core.hpp; the names: include/dc3hip.h): none changes a result
  c->no_hybrid = dbg_on("no_hybrid"); c->no_hybrid8 = dbg_on("no_hybrid8");
  c->no_small_ties = dbg_on("no_small_ties");
  c->no_long_keys = dbg_on("no_long_keys"); c->no_doubling = dbg_on("no_doubling");
  if (dbg_on("text_order12")) c->text_order12 = 1; else if (dbg_off("text_order12")) c->text_order12 = 0;
  c->no_msd = dbg_on("no_msd"); c->no_pack_strip = dbg_on("no_pack_strip");
  c->ssort_verify = dbg_on("ssort_verify"); c->no_wide_window = dbg_on("no_wide_window");
  c->no_text_shortcut = dbg_on("no_text_shortcut"); c->no_fullsort = dbg_on("no_fullsort"); c->no_discard = dbg_on("no_discard");
  { long long v; if (dbg_num("tup_scatter_min", &v)) c->tup_scatter_min = (u32)std::max(0ll, v); }
  { long long v; if (dbg_num("msd_min", &v)) c->msd_min = (u32)std::max(4096ll, v); }
  { long long v; if (dbg_num("msd_slot_cap", &v)) c->msd_slot_cap = (u32)std::max(1ll, v); }
  { long long v; if (dbg_num("ssort_min", &v)) c->ssort_min = (u32)std::max(8192ll, v); }
  { long long v; if (dbg_num("hybrid12_min", &v)) c->hybrid12_min = (u32)std::max(0ll, v); }
  c->use_vm = use_vm;
  { long long v; if (dbg_num("vmm_min", &v)) c->vm_min = (size_t)std::max(1ll, v); }
  int rc = [&]() -> int {
    HIPC(dc3_set_device(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    RC(ctx_big_alloc(c, &c->text_vm, (size_t)max_n + 64, reinterpret_cast<void **>(&c->d_text)));
    RC(ctx_big_alloc(c, &c->sa_vm, ((size_t)max_n + 16) * sizeof(u32), reinterpret_cast<void **>(&c->d_sa)));
    c->arena_bytes = std::min(arena_requirement(max_n), arena_text_requirement(max_n));   // grown on demand (ensure_arena)
    // testing aid: DC3HIP_ARENA_BYTES=<bytes> replaces the computed size (a build then either fits — possibly through
    // a fallback ordering — or fails loudly with -2; it never returns a wrong array)
    if (const char *e = getenv("DC3HIP_ARENA_BYTES")) { const long long v = atoll(e); if (v > 0) { c->arena_bytes = (size_t)v; c->arena_fixed = true; } }
    if (arena_from && arena_from->arena && arena_from->device == device) {
      // (a borrowed arena is used through this context's allocator, which cannot commit for the lender: all of it is backed first)
      if (arena_from->arena_vm.va && arena_from->arena == arena_from->arena_vm.va && !devbuf_commit(&arena_from->arena_vm, arena_from->arena_bytes)) {
        set_err("device allocation failed: arena of %zu bytes for a partition build", arena_from->arena_bytes); return E_ALLOC;
      }
      c->arena = arena_from->arena; c->arena_bytes = arena_from->arena_bytes; c->arena_borrowed = true; c->arena_fixed = true;
    } else {
      // address space for everything a context of max_n bytes can ever ask for, memory for what the first build needs
      const size_t want = c->arena_bytes;
      bool done = false;
      if (c->use_vm && !c->arena_fixed && arena_reserve_bytes(max_n) >= c->vm_min && devbuf_reserve(&c->arena_vm, device, arena_reserve_bytes(max_n))) {
        c->arena = c->arena_vm.va; c->arena_bytes = want; done = true;        // (nothing committed yet: arena_alloc does)
      }
      if (!done) HIPC(hipMalloc(&c->arena, c->arena_bytes));
    }
    HIPC(hipMalloc(&c->d_present, 256 * sizeof(u32)));
    HIPC(hipMalloc(&c->d_code, 256 * sizeof(uint16_t)));
    HIPC(hipMalloc(&c->d_words, 64 * sizeof(u32)));
    HIPC(hipMalloc(&c->d_xcdmon, 4096 * sizeof(u32)));         // (64 monitor words; 4096 for the placement probe below)
    HIPC(hipMalloc(&c->d_trace, 3 * DC3HIP_MAX_LEVELS * sizeof(u64)));
    c->h_words = static_cast<u32 *>(pinned_pool()->take(64 * sizeof(u32), &c->h_words_bytes));
    if (!c->h_words) { set_err("no pinned host memory"); return E_ALLOC; }
    HIPC(hipEventCreate(&c->ev_build_a));
    HIPC(hipEventCreate(&c->ev_build_b));
    {
      // Placement probe (dc3hip_stats.xcd_round_robin): 4096 one-wave blocks report the XCD they ran on.  The bucket
      // ordering's partition passes are only fast when blocks b and b + 8 share an XCD (XCD-grouped reservation,
      // dc3_msd.hip.hpp); on a device that places blocks otherwise the context keeps to the stable 256-bucket LSD passes,
      // whose speed does not depend on placement (DC3HIP_XCD_ASSUME=1: keep the bucket ordering anyway).
      // (a positive answer once per device and process: it is a property of the device's dispatcher, and a context per
      //  sacapart worker would otherwise launch 4096 blocks and wait for a copy each)
      static std::mutex probe_mu;
      static int probe_cache[64];
      static bool probe_init = false;
      std::lock_guard<std::mutex> lk(probe_mu);
      if (!probe_init) { for (int &v : probe_cache) v = -1; probe_init = true; }
      if (probe_cache[device & 63] < 0) {
        hipLaunchKernelGGL(k_xcd_probe, dim3(4096), dim3(64), 0, c->stream, c->d_xcdmon);
        void *hp = nullptr;
        RC(stage_d2h(c, c->d_xcdmon, 4096 * sizeof(u32), &hp));
        const u32 *xs = static_cast<const u32 *>(hp);
        u32 cnt[8][8] = {};
        for (u32 b = 0; b < 4096; b++) cnt[b & 7][xs[b] & 7]++;
        u32 hit = 0, seen = 0;
        for (int g = 0; g < 8; g++) {
          u32 mx = 0, arg = 0;
          for (int x = 0; x < 8; x++) if (cnt[g][x] > mx) { mx = cnt[g][x]; arg = (u32)x; }
          hit += mx; seen |= 1u << arg;
        }
        // (only a positive answer is kept for the process: a probe taken while other streams or ranks load the device can
        //  come out negative without the dispatcher being any different — the next context probes again)
        const int ok = (hit >= 4096 * 9 / 10 && seen == 0xffu) ? 1 : 0;
        if (ok) probe_cache[device & 63] = 1;
        c->xcd_rr = ok;
      } else {
        c->xcd_rr = 1;
      }
      const char *e = getenv("DC3HIP_XCD_ASSUME");
      if (!c->xcd_rr && !(e && e[0] == '1')) c->no_msd = true;
    }
    return E_OK;
  }();
  if (rc != E_OK) { dc3hip_ctx_destroy(c); return rc; }
  *out = c;
  return E_OK;
}

void dc3hip_ctx_destroy(dc3hip_ctx *c) try {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_build_a) (void)hipEventDestroy(c->ev_build_a);
  if (c->ev_build_b) (void)hipEventDestroy(c->ev_build_b);
  if (c->text_vm.va) devbuf_free(&c->text_vm); else if (c->d_text) (void)hipFree(c->d_text);
  if (c->sa_vm.va) devbuf_free(&c->sa_vm); else if (c->d_sa) (void)hipFree(c->d_sa);
  if (c->arena_vm.va) devbuf_free(&c->arena_vm); else if (c->arena && !c->arena_borrowed) (void)hipFree(c->arena);
  if (c->d_xcdmon) (void)hipFree(c->d_xcdmon);
  if (c->d_present) (void)hipFree(c->d_present);
  if (c->d_code) (void)hipFree(c->d_code);
  if (c->d_words) (void)hipFree(c->d_words);
  if (c->d_trace) (void)hipFree(c->d_trace);
  pinned_pool()->give(c->h_words, c->h_words_bytes);         // (recycled: PinnedPool)
  pinned_pool()->give(c->h_stage, c->h_stage_bytes);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
} DC3_ABI_CATCH_VOID

static int ctx_check_n(dc3hip_ctx *c, int64_t n) {
  if (!c || n < 0) { set_err("invalid arguments"); return E_ARGS; }
  if (n > c->max_n) { set_err("n=%lld exceeds the context capacity %lld", (long long)n, (long long)c->max_n); return E_ARGS; }
  return E_OK;
}

int32_t dc3hip_ctx_set_text(dc3hip_ctx *c, const uint8_t *T, int64_t n) try {
  RC(ctx_check_n(c, n));
  if (!T && n > 0) { set_err("T is NULL"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (n > 0) HIPC(hipMemcpyAsync(c->d_text, T, (size_t)n, hipMemcpyDefault, c->stream));
  HIPC(hipMemsetAsync(c->d_text + n, 0, 64, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->n = n; c->built = false;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_generate(dc3hip_ctx *c, int64_t n, uint64_t seed, int32_t kind) try {
  return dc3hip_ctx_generate_at(c, n, seed, kind, 0);
} DC3_ABI_CATCH

int32_t dc3hip_ctx_generate_at(dc3hip_ctx *c, int64_t n, uint64_t seed, int32_t kind, int64_t offset) try {
  RC(ctx_check_n(c, n));
  if (offset < 0) { set_err("negative offset"); return E_ARGS; }
  if (kind < 0 || kind > 2) { set_err("unknown generator kind %d", kind); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (n > 0) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(c, (u64)n / 8 + 1)), dim3(kBlock), 0, c->stream, c->d_text, (u64)n,
                       (u64)seed, (int)kind, (u64)offset);
    KCHECK();
  }
  HIPC(hipMemsetAsync(c->d_text + n, 0, 64, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->n = n; c->built = false;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_build(dc3hip_ctx *c) try {
  if (!c) { set_err("ctx is NULL"); return E_ARGS; }
  return ctx_build(c);
} DC3_ABI_CATCH

int32_t dc3hip_ctx_get_sa_i32(dc3hip_ctx *c, int32_t *SA) try {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (c->n > (int64_t)INT32_MAX) { set_err("text of %lld bytes needs 64-bit indices", (long long)c->n); return E_TOOBIG; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(SA, c->d_sa, (size_t)c->n * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_get_sa_i64(dc3hip_ctx *c, int64_t *SA) try {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (c->n == 0) return E_OK;
  // widen on the device in arena-sized pieces, then copy
  c->arena_off = 0;
  // (at most 1 GiB of it at a time: a reserved arena commits what is allocated)
  const size_t piece = std::min<size_t>((size_t)c->n, std::min<size_t>(c->arena_bytes / 8 > 0 ? c->arena_bytes / 8 : 1, (size_t)1 << 27));
  int64_t *tmp = nullptr;
  RC(arena_alloc(c, piece, &tmp));
  c->arena_off = 0;
  for (size_t off = 0; off < (size_t)c->n; off += piece) {
    const size_t cnt = std::min(piece, (size_t)c->n - off);
    hipLaunchKernelGGL(k_widen, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, c->d_sa + off, tmp, (u32)cnt);
    KCHECK();
    HIPC(hipMemcpyAsync(SA + off, tmp, cnt * 8, hipMemcpyDefault, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_get_text(dc3hip_ctx *c, uint8_t *T) try {
  if (!c || (!T && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(T, c->d_text, (size_t)c->n, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
} DC3_ABI_CATCH

// Runs the check; *code receives sufcheck()'s result (0, -2, -3, -4); the return value is the
// library status (E_OK / E_ALLOC / E_HIP), kept apart because the two code spaces overlap.
static int ctx_sufcheck(dc3hip_ctx *c, const u32 *d_sa, int *code, const uint8_t *text = nullptr, int64_t n_override = -1) {
  // utils.c:160-241: range and first characters in the gather kernel, the scan of :213-238 as one stable counting pass
  // whose sink compares (dc3_aux.hip.hpp).  text / n_override: a partition of the resident text
  const int64_t n = n_override >= 0 ? n_override : c->n;
  if (!text) text = c->d_text;
  *code = 0;
  if (n == 0) return E_OK;
  HIPC(dc3_set_device(c->device));
  c->arena_off = 0;
  const u32 n32 = (u32)n;
  int nb = 0; Chunking ck;
  radix_plan<Rec8>(c, n32, 8, &nb, &ck);                         // (8 key bits: 256 bins)
  uint8_t *bw = nullptr;
  u32 *table = nullptr, *digit_base = nullptr, *hist = nullptr;
  RC(arena_alloc(c, (size_t)n + 16, &bw));
  RC(arena_alloc(c, (size_t)256 * ck.nchunks, &table));
  RC(arena_alloc(c, (size_t)256, &digit_base));
  RC(arena_alloc(c, (size_t)256, &hist));
  int *err = reinterpret_cast<int *>(c->d_words + 8);
  u32 *total = c->d_words + 9, *qslot = c->d_words + 26;
  HIPC(hipMemsetAsync(err, 0, sizeof(int), c->stream));
  HIPC(hipMemsetAsync(hist, 0, 256 * sizeof(u32), c->stream));
  hipLaunchKernelGGL(k_check_text_hist, dim3(grid_for(c, (u64)n / 16 + 1)), dim3(kBlock), 0, c->stream, text, n32, hist);
  KCHECK();
  hipLaunchKernelGGL(k_check_gather, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, text, d_sa, n32, ck.chunk, ck.nchunks, bw, table, err);
  KCHECK();
  hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(kBlock), 0, c->stream, table, ck.nchunks, digit_base);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, digit_base, 256u, total);
  KCHECK();
  hipLaunchKernelGGL(k_check_counts, dim3(1), dim3(256), 0, c->stream, (const u32 *)hist, (const u32 *)digit_base, (const u32 *)total, text, d_sa, n32, qslot, err);
  KCHECK();
  {
    CheckLoader ld; ld.sa = d_sa; ld.bw = bw; ld.n = n32;
    CheckSink sk; sk.sa = d_sa; sk.q = qslot; sk.n = n32; sk.err = err;
    KeyDig dig; dig.shift = 32; dig.mask = 255;                  // (Rec8.key = the byte)
    const bool saved = c->profile; c->profile = false;           // (not a phase of a build)
    const int rc = launch_downsweep_to<Rec8, 256, CheckLoader, CheckSink>(c, ld, sk, n32, ck, dig, table, digit_base, DC3HIP_PH_OTHER);
    c->profile = saved;
    RC(rc);
  }
  HIPC(hipMemcpyAsync(c->h_words + 8, err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  const int v = (int)c->h_words[8];
  c->arena_off = 0;
  *code = v ? -(5 - v) : 0;
  return E_OK;
}

// By-products read T[SA[i]..] without range checks: an array handed in by dc3hip_ctx_set_sa_i32 is verified once
// (full sufcheck) before the first by-product call and refused if it is not the suffix array of the resident text.
static int ensure_trusted_sa(dc3hip_ctx *c) {
  if (c->sa_trusted || c->n == 0) return E_OK;
  int code = 0;
  RC(ctx_sufcheck(c, c->d_sa, &code));
  if (code != 0) { set_err("the resident array (dc3hip_ctx_set_sa_i32) is not the suffix array of the text: sufcheck %d", code); return E_ARGS; }
  c->sa_trusted = true;
  return E_OK;
}

int32_t dc3hip_ctx_sufcheck(dc3hip_ctx *c) try {
  if (!c) { set_err("ctx is NULL"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  int code = 0;
  const int rc = ctx_sufcheck(c, c->d_sa, &code);
  if (rc != E_OK) return rc == E_ALLOC ? -5 : -6;   // library failure, distinct from sufcheck's -1..-4
  return code;
} DC3_ABI_CATCH_SUFCHECK

int32_t dc3hip_ctx_sa_checksum(dc3hip_ctx *c, uint64_t *out) try {
  if (!c || !out) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  u64 *acc = reinterpret_cast<u64 *>(c->d_words + 16);
  HIPC(hipMemsetAsync(acc, 0, sizeof(u64), c->stream));
  if (c->n > 0) {
    hipLaunchKernelGGL(k_checksum, dim3(grid_for(c, c->n)), dim3(kBlock), 0, c->stream, c->d_sa, (u32)c->n, acc);
    KCHECK();
  }
  HIPC(hipMemcpyAsync(c->h_words + 16, acc, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  memcpy(out, c->h_words + 16, sizeof(u64));
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_set_sa_i32(dc3hip_ctx *c, const int32_t *SA) try {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(c->d_sa, SA, (size_t)c->n * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->built = true;
  c->sa_trusted = false;
  c->parts_trusted = 0;
  return E_OK;
} DC3_ABI_CATCH

// LCP array of the resident SA (kernels and method: dc3_aux.hip.hpp).  LCP may be a host or a device pointer (n x int32).
int32_t dc3hip_ctx_lcp_i32(dc3hip_ctx *c, int32_t *LCP) try {
  if (!c || (!LCP && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  const int64_t n64 = c->n;
  if (n64 == 0) return E_OK;
  if (n64 > (int64_t)INT32_MAX) { set_err("LCP values of %lld bytes do not fit int32", (long long)n64); return E_TOOBIG; }
  const u32 n = (u32)n64;
  HIPC(dc3_set_device(c->device));
  c->arena_off = 0;
  RC(ensure_arena(c, (size_t)n * 26 + ((size_t)64 << 20)));
  u32 *phi = nullptr, *plcp = nullptr;
  int32_t *dout = nullptr;
  RC(arena_alloc(c, (size_t)n + 16, &phi));
  RC(arena_alloc(c, (size_t)n + 16, &plcp));
  if (c->sa_trusted && n > 1) {
    Rec8 *pa = nullptr, *pb = nullptr;
    RC(arena_alloc(c, (size_t)n, &pa));
    RC(arena_alloc(c, (size_t)n, &pb));
    hipLaunchKernelGGL(k_phi_pairs, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, c->d_sa, n, pa);
    KCHECK();
    RC(inverse_permute(c, pa, pb, n, phi, DC3HIP_PH_OTHER));
  } else {
    HIPC(hipMemsetAsync(phi, 0xff, (size_t)n * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_phi_scatter, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, c->d_sa, n, phi);
    KCHECK();
  }
  const u32 ncoarse = (u32)(((u64)n + kLcpCoarse - 1) / kLcpCoarse);
  hipLaunchKernelGGL(k_plcp_coarse, dim3((ncoarse + kWaves - 1) / kWaves), dim3(kBlock), 0, c->stream, c->d_text, phi, n,
                     plcp);
  KCHECK();
  hipLaunchKernelGGL(k_plcp_fine, dim3(grid_for(c, (n + kLcpFine - 1) / kLcpFine)), dim3(kBlock), 0, c->stream, c->d_text,
                     phi, n, plcp);
  KCHECK();
  // phi is dead: the LCP values in rank order go there before they leave
  dout = reinterpret_cast<int32_t *>(phi);
  hipLaunchKernelGGL(k_lcp_gather, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, c->d_sa, plcp, n, dout);
  KCHECK();
  HIPC(hipMemcpyAsync(LCP, dout, (size_t)n * sizeof(int32_t), hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->arena_off = 0;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_bwt(dc3hip_ctx *c, uint8_t *U, int64_t *primary_index) try {
  if (!c || !primary_index || (!U && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }   // utils.c:60
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  const int64_t n = c->n;
  HIPC(dc3_set_device(c->device));
  RC(ensure_trusted_sa(c));
  if (n <= 1) {                                                                                    // utils.c:61-65
    if (n == 1) HIPC(hipMemcpy(U, c->d_text, 1, hipMemcpyDefault));
    *primary_index = n;
    return E_OK;
  }
  c->arena_off = 0;
  uint8_t *du = nullptr;
  RC(arena_alloc(c, (size_t)n + 16, &du));
  u32 *z = c->d_words + 24;
  HIPC(hipMemsetAsync(z, 0xff, sizeof(u32), c->stream));            // sentinel: "no entry is 0"
  hipLaunchKernelGGL(k_find_zero, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, c->d_sa, (u32)n, z);
  KCHECK();
  hipLaunchKernelGGL(k_bwt, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, c->d_text, c->d_sa, (u32)n, z, du);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 24, z, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipMemcpyAsync(U, du, (size_t)n, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  if (c->h_words[24] == 0xffffffffu) { set_err("internal: the suffix array holds no entry 0"); return E_HIP; }
  *primary_index = (int64_t)c->h_words[24] + 1;                                                    // utils.c:97
  c->arena_off = 0;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_search(dc3hip_ctx *c, const uint8_t *needles, const int64_t *offsets, int32_t count,
                          int64_t *out_start, int64_t *out_len) try {
  if (!c || count < 0 || (count > 0 && (!needles || !offsets || !out_start || !out_len))) {
    set_err("invalid arguments"); return E_ARGS;
  }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  if (c->n == 0) { set_err("empty suffix array (the reference indexes out of bounds here)"); return E_ARGS; }
  if (count == 0) return E_OK;
  HIPC(dc3_set_device(c->device));
  // offsets is a HOST array of count+1 non-decreasing byte offsets into needles, offsets[0] >= 0
  for (int32_t k = 0; k < count; k++)
    if (offsets[k] < 0 || offsets[k + 1] < offsets[k]) { set_err("invalid needle offsets (must be non-negative and non-decreasing)"); return E_ARGS; }
  const int64_t total = offsets[count];
  RC(ensure_trusted_sa(c));
  c->arena_off = 0;
  RC(ensure_arena(c, (size_t)total + (size_t)count * 24 + ((size_t)1 << 20)));
  uint8_t *dn = nullptr; int64_t *doff = nullptr, *ds = nullptr, *dl = nullptr;
  RC(arena_alloc(c, (size_t)total + 16, &dn));
  RC(arena_alloc(c, (size_t)count + 1, &doff));
  RC(arena_alloc(c, (size_t)count, &ds));
  RC(arena_alloc(c, (size_t)count, &dl));
  if (total > 0) HIPC(hipMemcpyAsync(dn, needles, (size_t)total, hipMemcpyDefault, c->stream));
  HIPC(hipMemcpyAsync(doff, offsets, ((size_t)count + 1) * 8, hipMemcpyDefault, c->stream));
  hipLaunchKernelGGL(k_search, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->d_text, (u32)c->n,
                     c->d_sa, dn, doff, (u32)count, ds, dl);
  KCHECK();
  HIPC(hipMemcpyAsync(out_start, ds, (size_t)count * 8, hipMemcpyDefault, c->stream));
  HIPC(hipMemcpyAsync(out_len, dl, (size_t)count * 8, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->arena_off = 0;
  return E_OK;
} DC3_ABI_CATCH

// Partition arrays (sacapart semantics) in the resident SA buffer: chunk c = text[c*S .. min(n,(c+1)*S)), S = n/P + 1
// (crates/sacapart/src/lib.rs:43-46), local indices, back to back — built here chunk by chunk on the device.
int32_t dc3hip_ctx_build_partitions(dc3hip_ctx *c, int32_t num_partitions) try {
  if (!c || num_partitions < 1) { set_err("invalid arguments"); return E_ARGS; }
  const int64_t n = c->n;
  if (n > (int64_t)INT32_MAX) { set_err("partitioned build of %lld bytes needs 64-bit indices", (long long)n); return E_TOOBIG; }
  HIPC(dc3_set_device(c->device));
  const int64_t S = n / num_partitions + 1;
  if (S >= n) {                      // one partition: its array is the suffix array of the whole text
    RC(dc3hip_ctx_build(c));
    c->sa_trusted = false; c->parts_trusted = num_partitions;
    return E_OK;
  }
  // the partitions are built one after the other by a child context that holds a chunk's text and array (5 S bytes) and
  // works in THIS context's arena (sized for n >= 2 S bytes: enough for every ordering of a chunk)
  c->arena_off = 0;                  // (no build of this context is running)
  if (!c->arena_fixed) RC(ensure_arena(c, arena_requirement(S)));
  dc3hip_ctx *child = nullptr;
  RC(ctx_create_impl(&child, c->device, S, c));
  int rc = E_OK;
  for (int64_t off = 0; off < n && rc == E_OK; off += S) {
    const int64_t len = std::min<int64_t>(S, n - off);
    rc = dc3hip_ctx_set_text(child, c->d_text + off, len);                       // (device pointer: hipMemcpyDefault)
    if (rc == E_OK) rc = dc3hip_ctx_build(child);
    if (rc == E_OK) rc = dc3hip_ctx_get_sa_i32(child, reinterpret_cast<int32_t *>(c->d_sa + off));
  }
  dc3hip_ctx_destroy(child);
  if (rc != E_OK) return rc;
  c->built = true; c->sa_trusted = false; c->parts_trusted = num_partitions;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_search_partitioned(dc3hip_ctx *c, int32_t num_partitions, const uint8_t *needles, const int64_t *offsets,
                                      int32_t count, int64_t *out_start, int64_t *out_len) try {
  if (!c || num_partitions < 1 || count < 0 || (count > 0 && (!needles || !offsets || !out_start || !out_len))) {
    set_err("invalid arguments"); return E_ARGS;
  }
  if (!c->built) { set_err("no partition arrays in this context (dc3hip_ctx_build_partitions / dc3hip_ctx_set_sa_i32)"); return E_ARGS; }
  if (c->n == 0) { set_err("empty text (the reference indexes out of bounds here)"); return E_ARGS; }
  if (c->n > (int64_t)INT32_MAX) { set_err("partitioned search of %lld bytes needs 64-bit indices", (long long)c->n); return E_TOOBIG; }
  if (count == 0) return E_OK;
  HIPC(dc3_set_device(c->device));
  for (int32_t k = 0; k < count; k++)
    if (offsets[k] < 0 || offsets[k + 1] < offsets[k]) { set_err("invalid needle offsets (must be non-negative and non-decreasing)"); return E_ARGS; }
  const int64_t n = c->n, S = n / num_partitions + 1;
  // arrays that were handed in (dc3hip_ctx_set_sa_i32) are verified once, partition by partition: the search reads
  // T[SA[i]..] without range checks
  if (c->parts_trusted != num_partitions) {
    for (int64_t off = 0; off < n; off += S) {
      int code = 0;
      RC(ctx_sufcheck(c, c->d_sa + off, &code, c->d_text + off, std::min<int64_t>(S, n - off)));
      if (code != 0) { set_err("the resident array is not %d partition suffix arrays of the text: partition at %lld fails sufcheck (%d)", num_partitions, (long long)off, code); return E_ARGS; }
    }
    c->parts_trusted = num_partitions;
  }
  const int64_t total = offsets[count];
  c->arena_off = 0;
  RC(ensure_arena(c, (size_t)total + (size_t)count * 24 + ((size_t)1 << 20)));
  uint8_t *dn = nullptr; int64_t *doff = nullptr, *ds = nullptr, *dl = nullptr;
  RC(arena_alloc(c, (size_t)total + 16, &dn));
  RC(arena_alloc(c, (size_t)count + 1, &doff));
  RC(arena_alloc(c, (size_t)count, &ds));
  RC(arena_alloc(c, (size_t)count, &dl));
  if (total > 0) HIPC(hipMemcpyAsync(dn, needles, (size_t)total, hipMemcpyDefault, c->stream));
  HIPC(hipMemcpyAsync(doff, offsets, ((size_t)count + 1) * 8, hipMemcpyDefault, c->stream));
  hipLaunchKernelGGL(k_search_partitioned, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->d_text, (u32)n, c->d_sa, (u32)S,
                     dn, doff, (u32)count, ds, dl);
  KCHECK();
  HIPC(hipMemcpyAsync(out_start, ds, (size_t)count * 8, hipMemcpyDefault, c->stream));
  HIPC(hipMemcpyAsync(out_len, dl, (size_t)count * 8, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->arena_off = 0;
  return E_OK;
} DC3_ABI_CATCH

// Test hook for kernel-level parity (the reference's radix_pass, crates/dc3/src/lib.rs:15-39): ONE stable pass of
// the product's radix scatter (up-sweep, scan, down-sweep) over n host words, digit = (word >> shift) & (nb - 1).
int32_t dc3hip_ctx_debug_radix_pass_u64(dc3hip_ctx *c, const uint64_t *words, uint64_t *out, int64_t n, int32_t shift,
                                        int32_t nb) try {
  if (!c || !words || !out || n < 1 || shift < 0 || shift > 55 || (nb != 256 && nb != 512)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(dc3_set_device(c->device));
  c->arena_off = 0;
  RC(ensure_arena(c, (size_t)n * 40 + ((size_t)64 << 20)));
  Rec8 *a = nullptr, *b = nullptr, *res = nullptr;
  RC(arena_alloc(c, (size_t)n, &a));
  RC(arena_alloc(c, (size_t)n, &b));
  // Rec8 = {key (high half), val (low half)}: the in-memory u64 is (val | key << 32) only on the device side of
  // rec8_word(); host words are split explicitly
  std::vector<Rec8> h((size_t)n);
  for (int64_t i = 0; i < n; i++) { h[(size_t)i].key = (u32)(words[i] >> 32); h[(size_t)i].val = (u32)words[i]; }
  HIPC(hipMemcpyAsync(a, h.data(), (size_t)n * sizeof(Rec8), hipMemcpyHostToDevice, c->stream));
  const bool saved = c->profile; c->profile = false;
  int rc = nb == 512 ? radix_passes<Rec8, 512>(c, a, b, (u32)n, (u32)shift, (u32)shift + 9, &res, 0, 0, 0)
                     : radix_passes<Rec8, 256>(c, a, b, (u32)n, (u32)shift, (u32)shift + 8, &res, 0, 0, 0);
  c->profile = saved;
  RC(rc);
  HIPC(hipMemcpyAsync(h.data(), res, (size_t)n * sizeof(Rec8), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  for (int64_t i = 0; i < n; i++) out[i] = ((uint64_t)h[(size_t)i].key << 32) | h[(size_t)i].val;
  c->arena_off = 0;
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_ctx_stats(dc3hip_ctx *c, dc3hip_stats *out) try {
  if (!c || !out) { set_err("invalid arguments"); return E_ARGS; }
  *out = c->stats;
  out->struct_size = (int32_t)sizeof(dc3hip_stats);
  return E_OK;
} DC3_ABI_CATCH

// ---- one-shot entry points -------------------------------------------------------------------

// n in {0,1,2} exactly as divsufsort.c:346-349 (host pointers only)
static bool tiny_host(const uint8_t *T, void *SA, int64_t n, int bits) {
  if (n > 2) return false;
  if (n == 0) return true;
  int64_t v[2] = {0, 0};
  if (n == 2) { const int m = (T[0] < T[1]); v[m ^ 1] = 0; v[m] = 1; }
  for (int64_t i = 0; i < n; i++) {
    if (bits == 32) static_cast<int32_t *>(SA)[i] = (int32_t)v[i]; else static_cast<int64_t *>(SA)[i] = v[i];
  }
  return true;
}

// One-shot calls keep their context (stream + device buffers) in a per-thread cache: a fresh
// hipMalloc of a multi-GB arena costs 0.2-0.6 s on MI355X, far more than the build itself.  The
// cache is private to the calling thread (so concurrent sacapart workers never share state), is
// released by dc3hip_release_cache() or at thread exit, and is disabled by DC3HIP_CACHE=0.
struct CtxCache {
  dc3hip_ctx *c = nullptr;
  ~CtxCache() { if (c) { dc3hip_ctx_destroy(c); c = nullptr; } }
};
static thread_local CtxCache g_cache;
static bool cache_enabled() {
  static const bool on = [] { const char *e = getenv("DC3HIP_CACHE"); return !(e && e[0] == '0'); }();
  return on;
}

static int acquire_ctx(dc3hip_ctx **out, int device, int64_t n, bool *cached) {
  *cached = false;
  if (cache_enabled()) {
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
    dc3hip_ctx *cc = g_cache.c;
    // reuse when it fits, unless the cached arena is far larger than this call needs (a context for 1 GiB holds
    // ~50 GB of HBM: do not pin that for a thread that has moved on to small texts)
    const bool oversized = cc && cc->max_n > 4 * std::max<int64_t>(n, 1) && cc->arena_bytes > ((size_t)4 << 30);
    if (cc && dev >= 0 && cc->device == dev && cc->max_n >= n && !oversized) { *out = cc; *cached = true; return E_OK; }
    if (cc) { dc3hip_ctx_destroy(cc); g_cache.c = nullptr; }
    RC(dc3hip_ctx_create(out, device, n));
    (*out)->one_shot = true;
    g_cache.c = *out; *cached = true;
    return E_OK;
  }
  RC(dc3hip_ctx_create(out, device, n));
  (*out)->one_shot = true;
  return E_OK;
}

static int sufsort_one(const uint8_t *T, void *SA, int64_t n, int bits, int device, bool devptrs) {
  if (!devptrs && tiny_host(T, SA, n, bits)) return E_OK;
  if (n == 0) return E_OK;
  dc3hip_ctx *c = nullptr;
  bool cached = false;
  typedef std::chrono::steady_clock clk;
  auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
  const clk::time_point t0 = clk::now();
  // The caller's array is usually fresh — `vec![0; n]` (cdivsufsort/src/lib.rs:27) is calloc: pages the OS has not handed out
  // yet — and a copy into such memory faults every page in inside the runtime's one copy thread (4 GiB: 220 ms against 78 ms
  // for resident pages).  A few host threads touch the pages while the text goes to the device and the build runs; the
  // array is overwritten completely either way (the contract of divsufsort()).
  struct Prefault {
    std::vector<std::thread> th;
    void join() { for (auto &t : th) if (t.joinable()) t.join(); th.clear(); }
    ~Prefault() { join(); }
  } prefault;
  const size_t sa_bytes = (size_t)n * (size_t)(bits / 8);
  // (only pages that do not exist yet — a resident array needs nothing: 64 sample pages are asked for with mincore first)
  auto mostly_absent = [](unsigned char *base, size_t bytes) {
    const uintptr_t pg = 4096;
    unsigned char *lo = reinterpret_cast<unsigned char *>((reinterpret_cast<uintptr_t>(base) + pg - 1) & ~(pg - 1));
    if (bytes < 128 * pg) return false;
    const size_t span = bytes - 2 * pg;
    unsigned absent = 0;
    for (unsigned i = 0; i < 64; i++) {
      unsigned char vec = 1;
      unsigned char *p = lo + ((span / 64 * i) & ~(size_t)(pg - 1));
      if (mincore(p, pg, &vec) != 0) return false;
      absent += (vec & 1) ? 0u : 1u;
    }
    return absent >= 48;
  };
  if (!devptrs && sa_bytes >= ((size_t)256 << 20) && mostly_absent(static_cast<unsigned char *>(SA), sa_bytes)) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = std::min(8u, std::max(1u, hw / 4));
    unsigned char *base = static_cast<unsigned char *>(SA);
    try {
      for (unsigned t = 0; t < nt; t++) {
        const size_t lo = sa_bytes / nt * t, hi = t + 1 == nt ? sa_bytes : sa_bytes / nt * (t + 1);
        prefault.th.emplace_back([base, lo, hi] { for (size_t o = lo; o < hi; o += 4096) *reinterpret_cast<volatile unsigned char *>(base + o) = 0; });
      }
    } catch (...) {}                           // (no thread to be had: the copy faults the pages in itself, as before)
  }
  RC(acquire_ctx(&c, device, n, &cached));
  const double t_ctx = ms_since(t0);
  double t_h2d = 0, t_build = 0, t_d2h = 0;
  int rc = [&]() -> int {
    clk::time_point t1 = clk::now();
    RC(dc3hip_ctx_set_text(c, T, n));          // hipMemcpyDefault handles host or device sources
    t_h2d = ms_since(t1); t1 = clk::now();
    RC(ctx_build(c));
    t_build = ms_since(t1);
    prefault.join();
    t1 = clk::now();
    if (bits == 32) RC(dc3hip_ctx_get_sa_i32(c, static_cast<int32_t *>(SA)));
    else RC(dc3hip_ctx_get_sa_i64(c, static_cast<int64_t *>(SA)));
    t_d2h = ms_since(t1);
    return E_OK;
  }();
  // DC3HIP_LEVEL_PHASES=1: where a one-shot call's wall time went (what crates/divsuftest/src/main.rs:145-151 times)
  if (c->level_report)
    std::fprintf(stderr, "dc3hip one-shot n=%lld: context %.1f ms (%s), text to device %.1f, build %.1f (device %.1f), array to host %.1f\n", (long long)n, t_ctx,
                 t_ctx > 1.0 ? "created" : "cached", t_h2d, t_build, c->stats.build_ms, t_d2h);
  if (!cached) dc3hip_ctx_destroy(c);
  else if (rc != E_OK) { dc3hip_ctx_destroy(c); g_cache.c = nullptr; }   // do not keep a context in an unknown state
  return rc;
}

int32_t dc3hip_sufsort_ex(const uint8_t *T, void *SA, int64_t n, const dc3hip_opts *o) try {
  dc3hip_opts d; memset(&d, 0, sizeof(d)); d.index_bits = 32; d.device = -1;
  if (o) {
    if (o->struct_size != (int32_t)sizeof(dc3hip_opts)) { set_err("dc3hip_opts.struct_size mismatch"); return E_ARGS; }
    d = *o;
  }
  if (T == nullptr || SA == nullptr || n < 0) { set_err("invalid arguments (NULL pointer or n < 0)"); return E_ARGS; }
  if (d.index_bits != 32 && d.index_bits != 64) { set_err("index_bits must be 32 or 64"); return E_ARGS; }
  const bool devptrs = (d.flags & DC3HIP_F_DEVICE_PTRS) != 0;
  const int64_t P = d.num_partitions > 1 ? d.num_partitions : 1;
  if (d.index_bits == 32 && (P == 1 ? n : n / P + 1) > (int64_t)INT32_MAX) {
    set_err("index_bits = 32 cannot address %lld bytes", (long long)n); return E_TOOBIG;
  }
  if (P == 1) {
    if (n > DC3HIP_MAX_N) { set_err("n=%lld exceeds DC3HIP_MAX_N=%lld", (long long)n, (long long)DC3HIP_MAX_N); return E_TOOBIG; }
    return sufsort_one(T, SA, n, d.index_bits, d.device, devptrs);
  }
  // sacapart semantics (sacapart/src/lib.rs:43-49): chunks of n/P + 1 bytes, independent SAs
  const int64_t S = n / P + 1;
  if (S > DC3HIP_MAX_N) { set_err("partition of %lld bytes exceeds DC3HIP_MAX_N", (long long)S); return E_TOOBIG; }
  const size_t isz = d.index_bits / 8;
  const int64_t nparts = (n + S - 1) / S;
  int ndev = 1;
  if ((d.flags & DC3HIP_F_ALL_DEVICES) && !devptrs) {
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { set_err("no HIP device"); return E_HIP; }
  }
  int per_dev = 1;
  if (const char *e = getenv("DC3HIP_WORKERS_PER_DEVICE")) per_dev = std::max(1, std::min(4, atoi(e)));
  const int workers = (int)std::min<int64_t>(nparts, (d.flags & DC3HIP_F_ALL_DEVICES) && !devptrs ? (int64_t)ndev * per_dev : 1);
  if (workers <= 1) {
    for (int64_t off = 0; off < n; off += S) {
      const int64_t len = std::min(S, n - off);
      RC(sufsort_one(T + off, static_cast<unsigned char *>(SA) + (size_t)off * isz, len, d.index_bits, d.device, devptrs));
    }
    return E_OK;
  }
  // The node's GPUs share the partitions (the rayon par_chunks of sacapart/src/lib.rs:45-49): worker w builds chunks
  // w, w+W, ... on device w % ndev with its own context and stream; no data is exchanged between partitions.
  std::vector<int> rcs((size_t)workers, E_OK);
  std::vector<std::string> msgs((size_t)workers);
  auto work = [&](int w) {
    try {
      const int dev = w % ndev;
      for (int64_t part = w; part < nparts; part += workers) {
        const int64_t off = part * S, len = std::min(S, n - off);
        const int rc = sufsort_one(T + off, static_cast<unsigned char *>(SA) + (size_t)off * isz, len, d.index_bits, dev,
                                   false);
        if (rc != E_OK) { rcs[(size_t)w] = rc; msgs[(size_t)w] = dc3hip_last_error(); break; }
      }
    } catch (...) { rcs[(size_t)w] = abi_exception(); try { msgs[(size_t)w] = dc3hip_last_error(); } catch (...) {} }
    dc3hip_release_cache();            // the worker thread ends here: give its arena back now
  };
  struct Pool {                        // (joined on every way out: a joinable std::thread must not be destroyed)
    std::vector<std::thread> th;
    ~Pool() { for (auto &t : th) if (t.joinable()) t.join(); }
  } pool;
  int started = 0;
  try {
    pool.th.reserve((size_t)workers);
    for (int w = 0; w < workers; w++) { pool.th.emplace_back(work, w); started++; }
  } catch (...) {}                     // (a container's process limit: the calling thread takes the chunks of the workers it could not start)
  for (int w = started; w < workers; w++) work(w);
  for (auto &t : pool.th) t.join();
  for (int w = 0; w < workers; w++)
    if (rcs[(size_t)w] != E_OK) { set_err("partition worker %d: %s", w, msgs[(size_t)w].c_str()); return rcs[(size_t)w]; }
  return E_OK;
} DC3_ABI_CATCH

int32_t dc3hip_sufsort_i32(const uint8_t *T, int32_t *SA, int32_t n) try {
  return dc3hip_sufsort_ex(T, SA, (int64_t)n, nullptr);
} DC3_ABI_CATCH

int32_t dc3hip_sufsort_i64(const uint8_t *T, int64_t *SA, int64_t n) try {
  dc3hip_opts o; memset(&o, 0, sizeof(o));
  o.struct_size = (int32_t)sizeof(o); o.index_bits = 64; o.device = -1;
  return dc3hip_sufsort_ex(T, SA, n, &o);
} DC3_ABI_CATCH

void dc3hip_release_cache(void) try {
  if (g_cache.c) { dc3hip_ctx_destroy(g_cache.c); g_cache.c = nullptr; }
} DC3_ABI_CATCH_VOID

// divbwt(T, U, A, n) (divsufsort.c:372-405): returns the primary index, -1 / -2 on error; A is an
// optional temporary in the reference and unused here.
int32_t dc3hip_divbwt_i32(const uint8_t *T, uint8_t *U, int32_t *A, int32_t n) try {
  (void)A;
  if (T == nullptr || U == nullptr || n < 0) { set_err("invalid arguments"); return -1; }
  if (n <= 1) { if (n == 1) U[0] = T[0]; return n; }
  dc3hip_ctx *c = nullptr;
  int rc = dc3hip_ctx_create(&c, -1, n);
  if (rc != E_OK) return rc == E_ARGS ? -1 : (rc == E_ALLOC ? -2 : rc);
  int64_t pidx = 0;
  rc = [&]() -> int {
    RC(dc3hip_ctx_set_text(c, T, n));
    RC(ctx_build(c));
    RC(dc3hip_ctx_bwt(c, U, &pidx));
    return E_OK;
  }();
  dc3hip_ctx_destroy(c);
  if (rc != E_OK) return rc;
  return (int32_t)pidx;
} DC3_ABI_CATCH

int32_t dc3hip_sufcheck_i32(const uint8_t *T, const int32_t *SA, int32_t n) try {
  if (T == nullptr || SA == nullptr || n < 0) { set_err("invalid arguments"); return -1; }  // utils.c:169-172
  if (n == 0) return 0;
  dc3hip_ctx *c = nullptr;
  int rc = dc3hip_ctx_create(&c, -1, n);
  if (rc != E_OK) return rc == E_ALLOC ? -5 : -6;   // distinct from sufcheck's own -1..-4
  int code = 0;
  rc = [&]() -> int {
    RC(dc3hip_ctx_set_text(c, T, n));
    HIPC(hipMemcpyAsync(c->d_sa, SA, (size_t)n * 4, hipMemcpyDefault, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    // negative entries become huge u32 values and fail the range check, like utils.c:179-188
    return ctx_sufcheck(c, c->d_sa, &code);
  }();
  dc3hip_ctx_destroy(c);
  if (rc != E_OK) return rc == E_ALLOC ? -5 : -6;
  return code;
} DC3_ABI_CATCH_SUFCHECK

}  // extern "C"

#include "dc3_global_host.hpp"
