// dc3_host_core.hpp — error strings, the context, the bump arena, per-phase HIP-event timing, chunking
// Host side of libdc3hip (single translation unit: included by dc3hip.hip in this order; everything here is static).
#pragma once

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
enum { E_OK = 0, E_ARGS = -1, E_ALLOC = -2, E_HIP = -3, E_TOOBIG = -4 };
// No C++ exception leaves the C ABI: every exported function is a function-try-block that ends in one of these.  The library
// throws nothing itself; what can arrive is the standard library's (std::bad_alloc of a host container, std::system_error of a
// thread that cannot be started in a container with a process limit).  Device buffers of the failed call may stay allocated
// until their context is destroyed.
static int abi_exception() {
  try { throw; }
  catch (const std::bad_alloc &) { set_err("host allocation failed (std::bad_alloc)"); return E_ALLOC; }
  catch (const std::exception &e) { set_err("C++ exception inside the library: %s", e.what()); return E_HIP; }
  catch (...) { set_err("unknown C++ exception inside the library"); return E_HIP; }
}
#define DC3_ABI_CATCH catch (...) { return abi_exception(); }
#define DC3_ABI_CATCH_SUFCHECK catch (...) { return abi_exception() == E_ALLOC ? -5 : -6; }        /* dc3hip_sufcheck_*: outside -1..-4 */
#define DC3_ABI_CATCH_GLOBAL_SUFCHECK catch (...) { return abi_exception() - 10; }
#define DC3_ABI_CATCH_VOID catch (...) { (void)abi_exception(); }

// hipFuncSetAttribute is reached by several host threads at once (the loopback ranks' first build, sacapart's workers, each
// behind its own "already set" flag): one at a time — a round-4 soak died twice, seconds into the process, with a corrupted
// host heap, and the runtime's per-function state is the one thing those threads all write.
static std::mutex g_func_attr_mu;
static inline hipError_t dc3_func_set_attribute(const void *fn, hipFuncAttribute attr, int value) {
  std::lock_guard<std::mutex> lk(g_func_attr_mu);
  return hipFuncSetAttribute(fn, attr, value);
}
#define HIPC(expr)                                                                              \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      (void)hipGetLastError(); /* the runtime keeps the last error until somebody reads it */   \
      set_err("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__), __FILE__,     \
              __LINE__, #expr);                                                                 \
      return (e__ == hipErrorOutOfMemory) ? E_ALLOC : E_HIP;                                    \
    }                                                                                           \
  } while (0)
#define RC(expr) do { int rc__ = (expr); if (rc__ != E_OK) return rc__; } while (0)
#define KCHECK() HIPC(hipGetLastError())
// Every entry point selects its context's device first; the same call drops an error that an EARLIER call left behind in
// the runtime's per-thread slot (one of ours that was reported or recovered from, or the application's own): hipGetLastError()
// behind a launch would otherwise blame this call for it (tools/oom_probe.py: a generator launch reported the out-of-memory
// of a context creation that had failed — and had been reported — two calls before).
static inline hipError_t dc3_set_device(int device) {
  const hipError_t e = hipSetDevice(device);
  (void)hipGetLastError();
  return e;
}

// ---------------------------------------------------------------------------------------------
// Large device buffers by virtual-memory reserve + commit (round 6).  Device memory that a process has freed is wiped by
// the driver before it is handed out again: about 30 ms per GiB on MI355X / ROCm 7.2, whichever call allocates it (hipMalloc
// of 44 GiB behind a free of as much: 1.7 s; the first 18 GiB of a context behind the destruction of two others: 2.4 s;
// clean memory: a millisecond — tools/alloc_probe.hip, tools/ctx_probe.py, profiles/r06c_*, r06l_*, r06m_*).  The un-warmed
// dc3hip_sufsort_i32 of 1 GiB that round 5 measured at 2.4 s (crates/divsuftest/src/main.rs:145-151 times exactly one such
// call) was waiting for that.  The wait cannot be avoided, only kept proportional to what a build USES: a DevBuf reserves
// the most its owner can ever need (hipMemAddressReserve) and commits pieces as they are needed (hipMemCreate / hipMemMap /
// hipMemSetAccess); growing never moves the buffer, so the arena of a context grows in the middle of a build and is
// committed as its bump allocator advances.  Where the calls are not available (a mock runtime, an old one) the owner falls
// back to hipMalloc.
// ---------------------------------------------------------------------------------------------
struct DevBuf {
  unsigned char *va = nullptr;
  size_t reserved = 0, mapped = 0, piece = 0;
  int device = 0;
  std::vector<hipMemGenericAllocationHandle_t> pieces;
};
// All pieces of a buffer have ONE size: on ROCm 7.2 hipMemSetAccess refuses ("invalid argument") a mapping that is smaller
// than the one mapped before it in the same reservation (976 + 464 MiB, 1024 + 464 MiB fail; 64 + 976, 512 x 3, 1024 x 6 pass:
// tools/vmm_probe.hip, profiles/r06f_vmm_probe.jsonl).  The piece is a 64th of the reservation, a power of two between
// 2 MiB and 1 GiB; commits are rounded up to whole pieces.
static constexpr size_t kDevBufPieceMax = (size_t)1 << 30, kDevBufPieceMin = (size_t)2 << 20;
static void devbuf_free(DevBuf *b) {
  if (!b->va) return;
  size_t off = 0;
  for (auto &h : b->pieces) { (void)hipMemUnmap(b->va + off, b->piece); (void)hipMemRelease(h); off += b->piece; }
  b->pieces.clear();
  (void)hipMemAddressFree(b->va, b->reserved);
  b->va = nullptr; b->reserved = b->mapped = b->piece = 0;
}
// (errors here are not reported through set_err: the caller falls back to hipMalloc or reports its own)
static bool devbuf_reserve(DevBuf *b, int device, size_t bytes) {
  size_t piece = kDevBufPieceMin;
  while (piece < kDevBufPieceMax && piece * 64 < bytes) piece <<= 1;
  void *p = nullptr;
  const size_t r = (bytes + piece - 1) / piece * piece;
  if (hipMemAddressReserve(&p, r, kDevBufPieceMin, nullptr, 0) != hipSuccess || !p) { (void)hipGetLastError(); return false; }
  b->va = static_cast<unsigned char *>(p); b->reserved = r; b->mapped = 0; b->piece = piece; b->device = device;
  return true;
}
// commit until at least `bytes` are mapped; false: out of memory or address space (what is mapped stays mapped)
static bool devbuf_commit(DevBuf *b, size_t bytes) {
  if (bytes <= b->mapped) return true;
  if (!b->va || bytes > b->reserved) return false;
  const size_t target = (bytes + b->piece - 1) / b->piece * b->piece;      // <= reserved (a multiple of the piece)
  hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = b->device;
  hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc));
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  while (b->mapped < target) {
    const size_t sz = b->piece;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemMap(b->va + b->mapped, sz, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
    if (hipMemSetAccess(b->va + b->mapped, sz, &acc, 1) != hipSuccess) {
      (void)hipGetLastError(); (void)hipMemUnmap(b->va + b->mapped, sz); (void)hipMemRelease(h); return false;
    }
    b->pieces.push_back(h);
    b->mapped += sz;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
static constexpr double kHybrid12MaxPredicted = 0.75;   // 12-byte prefix sort: taken below this predicted tied fraction (the sample
                                                        // extrapolation over-predicts on heavy-tailed repeats: 0.66 predicted, 0.08 measured on 1 GiB text)
struct PhaseMark { int phase; hipEvent_t a, b; int64_t elems; int kclass; int depth; };

struct dc3hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t max_n = 0, n = 0;
  bool built = false;
  bool sa_trusted = false;     // the resident SA was produced by ctx_build (a permutation), not handed in by set_sa
  int cur_depth = 0;           // recursion level the phase marks are charged to (DC3HIP_LEVEL_PHASES report)
  bool level_report = false;
  int parts_trusted = 0;       // the resident array is this many verified partition arrays (0: not known to be)
  uint8_t *d_text = nullptr;   // max_n + 64 bytes
  u32 *d_sa = nullptr;         // max_n + 16 words
  unsigned char *arena = nullptr;
  size_t arena_bytes = 0, arena_off = 0, arena_peak = 0;
  // round 6: big buffers are reserved address ranges, committed as needed (DevBuf): arena_vm.va == arena when the arena is
  // one; sa_vm / text_vm likewise for d_sa / d_text from vm_min bytes on.  use_vm = false (ranks of the global mode, whose
  // buffers RCCL and peer copies see): plain hipMalloc as before
  DevBuf arena_vm, sa_vm, text_vm;
  bool use_vm = true;
  size_t vm_min = (size_t)2 << 30;   // DC3HIP_DEBUG=vmm_min=<bytes> (tests): smallest buffer that is reserved + committed
  bool arena_fixed = false;    // DC3HIP_ARENA_BYTES given: never grown
  bool arena_borrowed = false; // the arena belongs to another context (ctx_create_impl): never grown, never freed here
  bool arena_exhausted = false; // the last E_ALLOC came from the bump allocator (not from hipMalloc)
  u32 builds_done = 0;          // builds this context has finished
  bool one_shot = false;        // created by a one-shot call (dc3hip_sufsort_*): its FIRST build does not commit the slots' 16 n
  // small device scratch
  u32 *d_present = nullptr;    // [256]
  uint16_t *d_code = nullptr;  // [256]
  u32 *d_words = nullptr;      // [64] misc totals / error words
  u32 *d_xcdmon = nullptr;     // [64] (block group, XCD) counts of the XCD-grouped partition kernels (xcd_note)
  int xcd_rr = -1;             // creation-time placement probe: 1 = blocks b and b + 8 shared an XCD and the 8 groups had 8 XCDs
  u32 *h_words = nullptr; size_t h_words_bytes = 0;   // pinned mirror (and the size the pool gave)
  unsigned char *h_stage = nullptr; size_t h_stage_bytes = 0;   // pinned landing area of every other device-to-host read (stage_d2h)
  // profiling
  bool profile = true;
  bool no_hybrid = false;
  bool no_small_ties = false;
  bool no_discard = false, no_fullsort = false, no_text_shortcut = false;
  bool no_long_keys = false;   // DC3HIP_NO_LONG_KEYS=1: the whole-text shortcut only with 9-symbol windows (no KeyT)
  bool no_doubling = false;    // DC3HIP_NO_DOUBLING=1: repeated windows always hand the whole-text order to level 1
  int text_order12 = -1;       // DC3HIP_TEXT_ORDER12=1/0: whole-text shortcut on 12-byte records always / never (default: n > 2^31)
  double hybrid_max_pred = 0.50;                      // 8-byte prefix sort of a level's samples: taken below this predicted tied fraction
  double hybrid12_max_pred = kHybrid12MaxPredicted;   // 12-byte prefix sort: taken below this predicted tied fraction
  bool no_hybrid8 = false;     // DC3HIP_DEBUG=no_hybrid8 (tests): skip the 8-byte prefix sort / whole-level order of a level (the 12-byte one is reached)
  u32 hybrid12_min = 1u << 22; // DC3HIP_HYBRID12_MIN: smallest level (samples) that tries it (tests lower it)
  u32 tup_scatter_min = 1u << 25; // DC3HIP_TUP_SCATTER_MIN (tests): smallest level (samples) whose tuples are scattered
  bool no_pack_strip = false;  // DC3HIP_DEBUG=no_pack_strip: partition pass 1 makes its words from an image no wider than the word (default: d1 bits wider, the bucket's own bits dropped)
  bool no_msd = false;         // DC3HIP_NO_MSD=1: the prefix sorts always run the stable LSD passes (no bucket ordering)
  u32 ssort_over = 24;         // splitter ordering: sample values per sub-bucket
  u32 ssort_mean = 1400;       // splitter ordering: records per sub-bucket it aims at (capacity 4096)
  bool no_wide_window = false; // DC3HIP_NO_WIDE_WINDOW=1: straight orderings always sort the triple (no wider window)
  bool ssort_verify = false;   // DC3HIP_SSORT_VERIFY=1 (tests): every splitter ordering checks its passes (record checksums, cursors, order); a mismatch fails the build
  u32 msd_slot_cap = 0;        // DC3HIP_DEBUG=msd_slot_cap=N (tests): slots of N words instead of twice the mean — small N makes them overflow
  u32 ssort_min = 1u << 23;    // DC3HIP_SSORT_MIN: fewest records the splitter ordering is used for (tests lower it)
  u32 msd_min = 1u << 20;      // DC3HIP_MSD_MIN: fewest records the bucket ordering is used for (tests lower it)
  bool trace = false;          // DC3HIP_TRACE=1: per-level checksums of SA12 / SA0 / SA (dc3hip_stats.trace_*)
  u64 *d_trace = nullptr;      // [3][DC3HIP_MAX_LEVELS]
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  std::vector<PhaseMark> marks;
  hipEvent_t ev_build_a = nullptr, ev_build_b = nullptr;
  dc3hip_stats stats;
  int num_cu = 256;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct ArenaMark { size_t off; };
static ArenaMark arena_mark(dc3hip_ctx *c) { return ArenaMark{c->arena_off}; }
static void arena_release(dc3hip_ctx *c, ArenaMark m) { c->arena_off = m.off; }
template <class T>
static int arena_alloc(dc3hip_ctx *c, size_t count, T **out) {
  const size_t bytes = align_up(count * sizeof(T), 256);
  if (c->arena_off + bytes > c->arena_bytes) {
    set_err("device work arena exhausted: need %zu more bytes (arena %zu, used %zu)", bytes, c->arena_bytes,
            c->arena_off);
    c->arena_exhausted = true;
    return E_ALLOC;
  }
  // a reserved arena (DevBuf) is committed as the allocator advances: a build pays for the memory it uses, not for what
  // the largest path of its size could use (arena_bytes is the LIMIT the policies see; arena_vm.mapped what is backed)
  if (c->arena_vm.va && c->arena == c->arena_vm.va && c->arena_off + bytes > c->arena_vm.mapped) {
    if (!devbuf_commit(&c->arena_vm, c->arena_off + bytes)) {
      set_err("device allocation failed: %zu bytes of the arena committed, %zu needed", c->arena_vm.mapped, c->arena_off + bytes);
      return E_ALLOC;
    }
  }
  *out = reinterpret_cast<T *>(c->arena + c->arena_off);
  c->arena_off += bytes;
  c->arena_peak = std::max(c->arena_peak, c->arena_off);
  return E_OK;
}

// Upper bound of the arena a build of n bytes can use (see DESIGN.md "Memory plan"):
// a level of length m holds 3 index arrays of m02 (+pad) while its child runs and at most
// 2 record arrays (16 B) or 2 tuple arrays (16 B) + 2 mod-0 tuple arrays (20 B) at its own peak.
static size_t arena_requirement(int64_t n) {
  size_t total = 0, held = 0;
  int64_t m = n;
  for (int lvl = 0; lvl < DC3HIP_MAX_LEVELS && m >= 2; lvl++) {
    const int64_t m0 = (m + 2) / 3, m02 = m0 + m / 3;
    const size_t keep = 4 * align_up((size_t)(m02 + 16) * 4, 256);
    const size_t tbl = 2 * align_up((size_t)4 * 4096 * 256, 256);
    const size_t recs = 2 * align_up((size_t)m02 * 16, 256) + 2 * align_up((size_t)m02 * 8, 256) + tbl;
    const size_t after = 2 * align_up((size_t)m0 * 20, 256) + align_up((size_t)(m / 1024 + 16) * 4, 256) +
                         (lvl > 0 ? 2 * align_up((size_t)m * 8, 256) : 0) + tbl;
    const size_t tups = align_up((size_t)m02 * 16, 256) + std::max(align_up((size_t)m02 * 16, 256), after);
    total = std::max(total, held + keep + std::max(recs, tups) + (1u << 20));
    held += keep;
    m = m02;
  }
  return total + (8u << 20);
}

// What the whole-text order (and every by-product except the LCP array) needs: two 8-byte word arrays, the image side
// array, a flag byte per record, radix tables and the tie predictor.  A context starts with this LIMIT and raises it to
// arena_requirement() the first time a build enters the DC3 recursion (ensure_arena): high-entropy texts never do.
static size_t arena_text_requirement(int64_t n) {
  // two 8-byte word arrays, 4 image bytes + 1 flag byte + 1 same byte per position, tables, the tie predictor: 22 n + tables
  // (+ the size tables of the bucket ordering: 2 x 8 words per sub-bucket, at most 2^20 sub-buckets)
  // The slots of the bucket ordering's second pass (16 bytes per position, msd_sort) are NOT part of it: a context whose
  // arena can grow in place (DevBuf) commits them when a sort first takes slots (arena_grow_in_use); every other context
  // — borrowed or fixed arenas, ranks of the global mode — keeps to the counted form.
  return (size_t)n * 24 + ((size_t)208 << 20);
}
// most a context of max_n bytes may ever ask of its arena: the recursion's requirement or the whole-text order with slots,
// the one retry of ctx_build (+ 50 %), the by-products (LCP: 26 n)
static size_t arena_reserve_bytes(int64_t n) {
  const size_t need = std::max(arena_requirement(n), (size_t)n * 42 + ((size_t)256 << 20));
  return need + need / 2 + ((size_t)1 << 30);
}

// Raise the arena's limit to at least `need` bytes.  Never shrinks; a size forced by DC3HIP_ARENA_BYTES stays as it is.  A
// reserved arena (DevBuf) grows where it lies — also while in use — and commits memory as the allocator advances; a
// hipMalloc'ed one is replaced and must be empty.
static int ensure_arena(dc3hip_ctx *c, size_t need) {
  if (c->arena_bytes >= need || c->arena_fixed) return E_OK;
  HIPC(dc3_set_device(c->device));            // (callers may be on a thread whose current device is another one)
  if (c->arena_vm.va && c->arena == c->arena_vm.va && need <= c->arena_vm.reserved) {
    c->arena_bytes = need;                 // (the limit; memory is committed when the allocator gets there: arena_alloc)
    return E_OK;
  }
  if (c->arena_off != 0) { set_err("internal: arena grown while in use"); return E_HIP; }
  HIPC(hipStreamSynchronize(c->stream));
  if (c->arena_vm.va && c->arena == c->arena_vm.va) { devbuf_free(&c->arena_vm); c->arena = nullptr; c->arena_bytes = 0; }
  else if (c->arena) { HIPC(hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
  if (c->use_vm && need >= c->vm_min && devbuf_reserve(&c->arena_vm, c->device, need + need / 2)) {
    c->arena = c->arena_vm.va; c->arena_bytes = need;
    return E_OK;
  }
  HIPC(hipMalloc(&c->arena, need));
  c->arena_bytes = need;
  return E_OK;
}
// In the middle of a build: a higher limit for what is in use, if the arena can grow where it lies (else false: the caller
// takes the path that needs no more memory).  Never an error; the memory itself is committed by arena_alloc.
static bool arena_grow_in_use(dc3hip_ctx *c, size_t need_total) {
  if (c->arena_bytes >= need_total) return true;
  if (c->arena_fixed || c->arena_borrowed || !c->arena_vm.va || c->arena != c->arena_vm.va || need_total > c->arena_vm.reserved) return false;
  c->arena_bytes = need_total;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Switches.  A deployment sets at most the POLICY variables — DC3HIP_PROFILE, DC3HIP_CACHE, DC3HIP_WORKERS_PER_DEVICE,
// DC3HIP_ARENA_BYTES, DC3HIP_XCD_ASSUME, DC3HIP_GLOBAL_LOCAL_MAX, DC3HIP_QUIET — and the diagnostics DC3HIP_TRACE and
// DC3HIP_LEVEL_PHASES.  Everything that only selects among orderings that are tested to give the same bytes, or plants a
// fault for a test, travels in ONE variable read when a context is created:
//     DC3HIP_DEBUG="name[=value],name,..."      e.g.  DC3HIP_DEBUG="no_text_shortcut,msd_min=4096"
// (round 4 had 48 getenv sites, one per switch: each a reachable code path of the shipped library that looked like an
// interface).  The names are the old variables' without the DC3HIP_ prefix, in lower case.
// ---------------------------------------------------------------------------------------------
static const char *dbg_find(const char *name) {       // -> the text behind "name" (at '=', ',' or the end), or nullptr
  const char *s = getenv("DC3HIP_DEBUG");
  if (!s) return nullptr;
  const size_t ln = strlen(name);
  while (*s) {
    const char *e = strchr(s, ',');
    const size_t tl = e ? (size_t)(e - s) : strlen(s);
    if (tl >= ln && strncmp(s, name, ln) == 0 && (tl == ln || s[ln] == '=')) return s + ln;
    if (!e) break;
    s = e + 1;
  }
  return nullptr;
}
static bool dbg_on(const char *name) { const char *v = dbg_find(name); return v && !(v[0] == '=' && v[1] == '0'); }   // "name" or "name=1"
static bool dbg_off(const char *name) { const char *v = dbg_find(name); return v && v[0] == '=' && v[1] == '0'; }    // "name=0"
static bool dbg_num(const char *name, long long *out) { const char *v = dbg_find(name); if (!v || v[0] != '=') return false; *out = atoll(v + 1); return true; }
static bool dbg_real(const char *name, double *out) { const char *v = dbg_find(name); if (!v || v[0] != '=') return false; *out = atof(v + 1); return true; }

// ---------------------------------------------------------------------------------------------
// Device-to-host reads of small results (samples, digit tables, monitor words) land in PINNED memory of the context.  An
// asynchronous copy into pageable memory — a std::vector, a stack array of a rank thread — makes the runtime register
// and unregister the caller's pages on the fly; with several rank threads doing that at once a round-4 hunt saw glibc's
// free() called on the base address of the runtime's host aperture a few per cent of the time
// (profiles/r04u_fresh_process_crash_hunt.md).  stage_d2h_async: the copy is queued, the caller synchronises the stream
// before reading *host; the area is reused by the next call.
// ---------------------------------------------------------------------------------------------
// Pinned host buffers are RECYCLED across contexts: a context pins at least 1 MiB and sacapart-style callers create many
// worker and child contexts.  (Round 4 never handed them back at all: the one pointer its crash hunt saw in glibc's free()
// was the base of the runtime's host-memory mapping, the kind hipHostMalloc returns.  Round 5: the deaths follow the HIP
// runtime — 0 of 110 perturbed fresh processes on the ROCm 7.2 runtime the library is compiled against, 1 of 110 and all 13
// of round 4 on the 7.0 runtime bundled with a PyTorch wheel that was mapped first — and the host half runs clean under
// ASan and TSan on a mock runtime, tools/hostmock; profiles/r05_crash_hunt.md.)  The pool is best-fit and capped: beyond
// kKeepBuffers / kKeepBytes idle buffers go back to the runtime.
struct PinnedPool {
  static constexpr size_t kKeepBuffers = 64, kKeepBytes = (size_t)256 << 20;
  std::mutex mu;
  std::vector<std::pair<void *, size_t>> free_list;
  size_t idle_bytes = 0;
  void *take(size_t bytes, size_t *got) {
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = free_list.size();
      for (size_t i = 0; i < free_list.size(); i++)
        if (free_list[i].second >= bytes && free_list[i].second <= 4 * bytes + 4096 && (best == free_list.size() || free_list[i].second < free_list[best].second)) best = i;
      if (best != free_list.size()) {
        void *p = free_list[best].first; *got = free_list[best].second;
        idle_bytes -= free_list[best].second;
        free_list[best] = free_list.back(); free_list.pop_back();
        return p;
      }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *got = bytes;
    return p;
  }
  // bytes: the size take() reported for this buffer
  void give(void *p, size_t bytes) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (free_list.size() < kKeepBuffers && idle_bytes + bytes <= kKeepBytes) { free_list.emplace_back(p, bytes); idle_bytes += bytes; return; }
    }
    (void)hipHostFree(p);
  }
};
static PinnedPool *pinned_pool() { static PinnedPool *p = new PinnedPool(); return p; }      // (lives as long as the process)

static int stage_d2h_async(dc3hip_ctx *c, const void *dev, size_t bytes, void **host) {
  if (bytes > c->h_stage_bytes) {
    HIPC(hipStreamSynchronize(c->stream));
    pinned_pool()->give(c->h_stage, c->h_stage_bytes);
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 20);
    size_t got = 0;
    c->h_stage = static_cast<unsigned char *>(pinned_pool()->take(want, &got));
    if (!c->h_stage) { set_err("no pinned host memory for %zu bytes", want); return E_ALLOC; }
    c->h_stage_bytes = got;
  }
  if (bytes) HIPC(hipMemcpyAsync(c->h_stage, dev, bytes, hipMemcpyDeviceToHost, c->stream));
  *host = c->h_stage;
  return E_OK;
}
static int stage_d2h(dc3hip_ctx *c, const void *dev, size_t bytes, void **host) {
  RC(stage_d2h_async(c, dev, bytes, host));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------------------------
static hipEvent_t get_event(dc3hip_ctx *c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  return c->ev_pool[c->ev_used++];
}
struct PhaseScope {
  dc3hip_ctx *c; size_t idx; bool on;
  PhaseScope(dc3hip_ctx *ctx, int phase, int64_t elems = 0, int kclass = -1) : c(ctx), idx(0), on(ctx->profile) {
    if (!on) return;
    PhaseMark m; m.phase = phase; m.a = get_event(c); m.b = get_event(c); m.elems = elems; m.kclass = kclass; m.depth = c->cur_depth;
    if (!m.a || !m.b) { on = false; return; }
    (void)hipEventRecord(m.a, c->stream);
    idx = c->marks.size(); c->marks.push_back(m);
  }
  ~PhaseScope() { if (on) (void)hipEventRecord(c->marks[idx].b, c->stream); }
};

static inline u32 bits_of(u64 v) { u32 b = 0; while (v) { b++; v >>= 1; } return b ? b : 1; }
static inline int grid_for(dc3hip_ctx *c, u64 work_items, int per_block = kBlock) {
  u64 g = (work_items + per_block - 1) / per_block;
  const u64 cap = (u64)c->num_cu * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------
// chunking shared by the up-/down-sweep style kernels
// ---------------------------------------------------------------------------------------------
struct Chunking { u32 chunk, nchunks; };
static Chunking make_chunks(dc3hip_ctx *c, u32 n, u32 tile) {
  const u32 target_blocks = (u32)c->num_cu * 8;
  u32 chunk = (n + target_blocks - 1) / target_blocks;
  chunk = (chunk + tile - 1) / tile * tile;
  if (chunk < tile) chunk = tile;
  Chunking k; k.chunk = chunk; k.nchunks = (n + chunk - 1) / chunk;
  if (k.nchunks == 0) k.nchunks = 1;
  return k;
}

