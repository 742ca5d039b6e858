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
#include <cmath>
#include <cstring>
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

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
enum { E_OK = 0, E_ARGS = -1, E_ALLOC = -2, E_HIP = -3, E_TOOBIG = -4 };

#define HIPC(expr)                                                                              \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      set_err("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__), __FILE__,     \
              __LINE__, #expr);                                                                 \
      return (e__ == hipErrorOutOfMemory) ? E_ALLOC : E_HIP;                                    \
    }                                                                                           \
  } while (0)
#define RC(expr) do { int rc__ = (expr); if (rc__ != E_OK) return rc__; } while (0)
#define KCHECK() HIPC(hipGetLastError())

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
  bool arena_fixed = false;    // DC3HIP_ARENA_BYTES given: never grown
  bool arena_borrowed = false; // the arena belongs to another context (ctx_create_impl): never grown, never freed here
  bool arena_exhausted = false; // the last E_ALLOC came from the bump allocator (not from hipMalloc)
  // small device scratch
  u32 *d_present = nullptr;    // [256]
  uint16_t *d_code = nullptr;  // [256]
  u32 *d_words = nullptr;      // [64] misc totals / error words
  u32 *d_xcdmon = nullptr;     // [64] (block group, XCD) counts of the XCD-grouped partition kernels (xcd_note)
  int xcd_rr = -1;             // creation-time placement probe: 1 = blocks b and b + 8 shared an XCD and the 8 groups had 8 XCDs
  u32 *h_words = nullptr;      // pinned mirror
  // profiling
  bool profile = true;
  bool no_hybrid = false;
  bool no_small_ties = false;
  bool no_nine_bit = false, no_rec12 = false, no_discard = false, no_fullsort = false, no_text_shortcut = false;
  bool no_split_emit = false;
  bool no_long_keys = false;   // DC3HIP_NO_LONG_KEYS=1: the whole-text shortcut only with 9-symbol windows (no KeyT)
  bool no_doubling = false;    // DC3HIP_NO_DOUBLING=1: repeated windows always hand the whole-text order to level 1
  int text_order12 = -1;       // DC3HIP_TEXT_ORDER12=1/0: whole-text shortcut on 12-byte records always / never (default: n > 2^31)
  double hybrid_max_pred = 0.50;                      // 8-byte prefix sort of a level's samples: taken below this predicted tied fraction
  double hybrid12_max_pred = kHybrid12MaxPredicted;   // 12-byte prefix sort: taken below this predicted tied fraction
  u32 hybrid12_min = 1u << 22; // DC3HIP_HYBRID12_MIN: smallest level (samples) that tries it (tests lower it)
  bool no_hybrid8 = false;     // DC3HIP_NO_HYBRID8=1 (tests): skip the 8-byte prefix sort / whole-level order of a level
  bool no_hybrid12 = false;    // DC3HIP_NO_HYBRID12=1: no 63-bit-prefix sort on 12-byte records for keys wider than 64 bits
  bool no_tup_scatter = false; // DC3HIP_NO_TUP_SCATTER=1: sample tuples always by the random gather
  u32 tup_scatter_min = 1u << 25; // DC3HIP_TUP_SCATTER_MIN (tests): smallest level (samples) whose tuples are scattered
  bool no_tup_rec8 = false;    // DC3HIP_NO_TUP_REC8=1 (tests): level 0 moves 12-byte records through the tuple scatter, as deeper levels do
  bool no_xcd_map = false;     // DC3HIP_NO_XCD_MAP=1: window partitions without the segment -> XCD-group tile order (measurement aid)
  bool pack_fuse = true;       // DC3HIP_PACK_FUSE=0: whole-text order of bytes with a pack kernel that WRITES the words (default: it only counts, partition pass 1 makes them on the fly)
  bool tup_bigtile = true;     // DC3HIP_TUP_BIGTILE=0 (lab / tests): level 0's tuple scatter pass 1 in the 4096-slot, 512-thread shape of the deeper levels
  bool no_pack_strip = false;  // DC3HIP_NO_PACK_STRIP=1: ... from an image no wider than the word (default: d1 bits wider, the bucket's own bits dropped)
  bool no_msd = false;         // DC3HIP_NO_MSD=1: the prefix sorts always run the stable LSD passes (no bucket ordering)
  u32 ssort_over = 24;         // splitter ordering: sample values per sub-bucket
  u32 ssort_mean = 1400;       // splitter ordering: records per sub-bucket it aims at (capacity 4096)
  bool no_wide_window = false; // DC3HIP_NO_WIDE_WINDOW=1: straight orderings always sort the triple (no wider window)
  bool ssort_rec12 = false;    // DC3HIP_SSORT_REC12=1: the splitter ordering also for keys of at most 64 bits (tests)
  bool no_pack_count = false;  // DC3HIP_NO_PACK_COUNT=1: the wide-window records are packed by their own kernel, then counted
  bool ssort_verify = false;   // DC3HIP_SSORT_VERIFY=1 (tests): every splitter ordering checks its passes (record checksums, cursors, order); a mismatch fails the build
  bool no_ssort = false;       // DC3HIP_NO_SSORT=1: the straight orderings always run the stable LSD passes (no splitter ordering)
  u32 ssort_min = 1u << 23;    // DC3HIP_SSORT_MIN: fewest records the splitter ordering is used for (tests lower it)
  u32 msd_min = 1u << 20;      // DC3HIP_MSD_MIN: fewest records the bucket ordering is used for (tests lower it)
  bool no_tup8 = false;        // DC3HIP_NO_TUP8=1: the slot table of the merge tuples is always 16 bytes per sample
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

// What the whole-text order (and every by-product except the LCP array) needs: two 8-byte record arrays, the image
// side array, a flag byte per record, radix tables and the tie predictor.  A context starts with this much and grows
// to arena_requirement() the first time a build enters the DC3 recursion (ensure_arena): high-entropy texts never
// do, so their contexts hold half the memory and the first hipMalloc is half as long.
static size_t arena_text_requirement(int64_t n) {
  // (beyond 2^31 positions the whole-text order runs on 12-byte records: 2 x 12 + 1 bytes per position + tables)
  // (+ the size tables of the bucket ordering: 2 x 8 words per sub-bucket, at most 2^20 sub-buckets)
  return n > ((int64_t)1 << 31) ? (size_t)n * 26 + ((size_t)256 << 20) : (size_t)n * 24 + ((size_t)208 << 20);
}

// Grow the (empty) arena to at least `need` bytes.  Never shrinks; a size forced by DC3HIP_ARENA_BYTES stays as it is.
static int ensure_arena(dc3hip_ctx *c, size_t need) {
  if (c->arena_bytes >= need || c->arena_fixed) return E_OK;
  if (c->arena_off != 0) { set_err("internal: arena grown while in use"); return E_HIP; }
  HIPC(hipSetDevice(c->device));            // (callers may be on a thread whose current device is another one)
  HIPC(hipStreamSynchronize(c->stream));
  if (c->arena) { HIPC(hipFree(c->arena)); c->arena = nullptr; c->arena_bytes = 0; }
  HIPC(hipMalloc(&c->arena, need));
  c->arena_bytes = need;
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

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort over a bit range of the key (lib.rs:15-39 per digit)
// ---------------------------------------------------------------------------------------------
// tile shapes per record type and digit width (NB bins); LDS = records + NW*NB counters (<= 160 KiB)
template <class Rec, int NB> struct SortCfg;
template <> struct SortCfg<Rec8, 256>  { static constexpr int IPT = 12, NW = 16; static constexpr bool PF = true; };
template <> struct SortCfg<Rec8, 512>  { static constexpr int IPT = 12, NW = 16; static constexpr bool PF = true; };
template <> struct SortCfg<Rec12, 256> { static constexpr int IPT = 10, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec12, 512> { static constexpr int IPT = 10, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec16, 256> { static constexpr int IPT = 8, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec16, 512> { static constexpr int IPT = 7, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0, 256>  { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0, 512>  { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0C, 256> { static constexpr int IPT = 8, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0C, 512> { static constexpr int IPT = 7, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0G, 256> { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0G, 512> { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <class Rec> struct RecClass;      // index into dc3hip_stats.downsweep_*
template <> struct RecClass<Tup0G> { static constexpr int k = 2; };
template <> struct RecClass<Rec8>  { static constexpr int k = 0; };
template <> struct RecClass<Rec12> { static constexpr int k = 1; };
template <> struct RecClass<Rec16> { static constexpr int k = 1; };
template <> struct RecClass<Tup0>  { static constexpr int k = 2; };
template <> struct RecClass<Tup0C> { static constexpr int k = 2; };

template <class Rec, int NB, class Loader, class Sink>
static int launch_downsweep_to(dc3hip_ctx *c, Loader in, Sink dst, u32 n, const Chunking &ck, KeyDig dig,
                               const u32 *table, const u32 *digit_base, int phase) {
  constexpr int IPT = SortCfg<Rec, NB>::IPT, NW = SortCfg<Rec, NB>::NW;
  constexpr bool PF = SortCfg<Rec, NB>::PF && std::is_same<Loader, ArrayLoader<Rec>>::value;
  const size_t smem = DownsweepSmem<Rec, IPT, NW, NB>::kBytes;
  auto kern = k_rs_downsweep<Rec, NB, IPT, NW, PF, Loader, Sink>;
  static std::atomic<bool> attr_set[16];   // (per function and device, process-wide; a double set is harmless)
  if (!attr_set[c->device & 15]) {
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)smem));
    attr_set[c->device & 15] = true;
  }
  PhaseScope ps(c, phase, n, RecClass<Rec>::k);
  hipLaunchKernelGGL(kern, dim3(ck.nchunks), dim3(NW * 64), smem, c->stream, in, dst, n, ck.chunk, ck.nchunks, dig,
                     table, digit_base, 0u);
  KCHECK();
  return E_OK;
}
template <class Rec, int NB, class Loader>
static int launch_downsweep(dc3hip_ctx *c, Loader in, Rec *dst, u32 n, const Chunking &ck, KeyDig dig,
                            const u32 *table, const u32 *digit_base, int phase) {
  RecSink<Rec> sink; sink.p = dst;
  return launch_downsweep_to<Rec, NB, Loader, RecSink<Rec>>(c, in, sink, n, ck, dig, table, digit_base, phase);
}
static int scan_digit_table(dc3hip_ctx *c, u32 *table, u32 nchunks, u32 *digit_base, u32 nb, int phase) {
  PhaseScope ps(c, phase, nb * nchunks);
  hipLaunchKernelGGL(k_scan_rows, dim3(nb), dim3(kBlock), 0, c->stream, table, nchunks, digit_base);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, digit_base, nb, (u32 *)nullptr);
  KCHECK();
  return E_OK;
}

// Stable LSD sort of key bits [bit_lo, bit_hi) of the records in `a` (ping-pong with `b`).
// Digit width: 9 bits where that saves a pass over 8-bit digits, else 8.
// first_table: digit table of the first pass already produced by whoever wrote the records (k_pack_image_text);
// it must have been made for radix_plan()'s chunking and bin count.
// final_sink (Rec8 only): the LAST pass writes through it instead of into the other record buffer; *last then
// describes that pass (source buffer, destination buffer, digit) so that it can be repeated into records
// (radix_redo_last) if the caller turns out to need them after all.
struct LastPass { void *src = nullptr, *dst = nullptr; u32 lo = 0; int nb = 0; };
template <class Rec, int NB>
static int radix_passes(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 bit_lo, u32 bit_hi, Rec **result, int ph_up,
                        int ph_scan, int ph_down, u32 *first_table = nullptr, const SplitSink *final_sink = nullptr,
                        LastPass *last = nullptr) {
  constexpr u32 kBits = NB == 512 ? 9 : 8;
  constexpr int kTile = SortCfg<Rec, NB>::NW * 64 * SortCfg<Rec, NB>::IPT;
  const Chunking ck = make_chunks(c, n, kTile);
  const ArenaMark mk = arena_mark(c);
  u32 *table = first_table, *digit_base = nullptr;
  if (!table) RC(arena_alloc(c, (size_t)NB * ck.nchunks, &table));
  RC(arena_alloc(c, (size_t)NB, &digit_base));
  Rec *src = a, *dst = b;
  for (u32 lo = bit_lo; lo < bit_hi; lo += kBits) {
    KeyDig dig; dig.shift = lo; dig.mask = NB - 1;
    if (!(first_table && lo == bit_lo)) {
      PhaseScope ps(c, ph_up, n);
      hipLaunchKernelGGL((k_rs_upsweep<Rec, NB>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, src, n, ck.chunk,
                         ck.nchunks, dig, table);
      KCHECK();
    }
    RC(scan_digit_table(c, table, ck.nchunks, digit_base, NB, ph_scan));
    ArrayLoader<Rec> ld; ld.p = src;
    if constexpr (std::is_same<Rec, Rec8>::value) {
      if (final_sink && lo + kBits >= bit_hi) {
        RC((launch_downsweep_to<Rec, NB, ArrayLoader<Rec>, SplitSink>(c, ld, *final_sink, n, ck, dig, table, digit_base,
                                                                      ph_down)));
        if (last) { last->src = src; last->dst = dst; last->lo = lo; last->nb = NB; }
        arena_release(c, mk);
        *result = nullptr;                 // the order lives in the sink
        return E_OK;
      }
    }
    RC((launch_downsweep<Rec, NB, ArrayLoader<Rec>>(c, ld, dst, n, ck, dig, table, digit_base, ph_down)));
    std::swap(src, dst);
  }
  arena_release(c, mk);
  *result = src;
  return E_OK;
}
// repeat the last pass of a sort that ended in a SplitSink, this time into records
template <int NB>
static int radix_redo_last_nb(dc3hip_ctx *c, const LastPass &lp, u32 n, Rec8 **result, int ph_up, int ph_scan, int ph_down) {
  constexpr int kTile = SortCfg<Rec8, NB>::NW * 64 * SortCfg<Rec8, NB>::IPT;
  const Chunking ck = make_chunks(c, n, kTile);
  const ArenaMark mk = arena_mark(c);
  u32 *table = nullptr, *digit_base = nullptr;
  RC(arena_alloc(c, (size_t)NB * ck.nchunks, &table));
  RC(arena_alloc(c, (size_t)NB, &digit_base));
  KeyDig dig; dig.shift = lp.lo; dig.mask = NB - 1;
  Rec8 *src = static_cast<Rec8 *>(lp.src), *dst = static_cast<Rec8 *>(lp.dst);
  {
    PhaseScope ps(c, ph_up, n);
    hipLaunchKernelGGL((k_rs_upsweep<Rec8, NB>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, src, n, ck.chunk, ck.nchunks,
                       dig, table);
    KCHECK();
  }
  RC(scan_digit_table(c, table, ck.nchunks, digit_base, NB, ph_scan));
  ArrayLoader<Rec8> ld; ld.p = src;
  RC((launch_downsweep<Rec8, NB, ArrayLoader<Rec8>>(c, ld, dst, n, ck, dig, table, digit_base, ph_down)));
  arena_release(c, mk);
  *result = dst;
  return E_OK;
}
static int radix_redo_last(dc3hip_ctx *c, const LastPass &lp, u32 n, Rec8 **result, int ph_up, int ph_scan, int ph_down) {
  return lp.nb == 512 ? radix_redo_last_nb<512>(c, lp, n, result, ph_up, ph_scan, ph_down)
                      : radix_redo_last_nb<256>(c, lp, n, result, ph_up, ph_scan, ph_down);
}
static bool radix_nine(const dc3hip_ctx *c, u32 bits) { return !c->no_nine_bit && ((bits + 8) / 9 < (bits + 7) / 8); }
template <class Rec>
static int radix_sort(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 bit_lo, u32 bit_hi, Rec **result, int ph_up,
                      int ph_scan, int ph_down, u32 *first_table = nullptr, const SplitSink *final_sink = nullptr,
                      LastPass *last = nullptr) {
  const u32 bits = bit_hi > bit_lo ? bit_hi - bit_lo : 0;
  if (bits == 0) { *result = a; return E_OK; }
  if (radix_nine(c, bits))
    return radix_passes<Rec, 512>(c, a, b, n, bit_lo, bit_hi, result, ph_up, ph_scan, ph_down, first_table, final_sink, last);
  return radix_passes<Rec, 256>(c, a, b, n, bit_lo, bit_hi, result, ph_up, ph_scan, ph_down, first_table, final_sink, last);
}
// bins and chunking radix_sort<Rec> will use for n records and `bits` key bits
template <class Rec>
static void radix_plan(dc3hip_ctx *c, u32 n, u32 bits, int *nb, Chunking *ck) {
  const bool nine = radix_nine(c, bits);
  *nb = nine ? 512 : 256;
  const int tile = nine ? SortCfg<Rec, 512>::NW * 64 * SortCfg<Rec, 512>::IPT : SortCfg<Rec, 256>::NW * 64 * SortCfg<Rec, 256>::IPT;
  *ck = make_chunks(c, n, (u32)tile);
}

// ---------------------------------------------------------------------------------------------
// Bucket (MSD) ordering of the prefix-sort words (dc3_msd.hip.hpp): the same array the stable LSD passes over the image
// bits produce, in two non-stable partition passes + an in-LDS order of the sub-buckets.
// ---------------------------------------------------------------------------------------------
struct MsdGeom {
  bool on = false;
  u32 d1 = 0, d2 = 0;                            // digit widths of the two partition passes (d2 = 0: one pass)
  u32 ntiles1 = 0, tpc = 0, cpg = 0, cpx1 = 0;   // pass-1 tiles; tiles per pack chunk, chunks and tiles per group
  Chunking ck{0, 0};                             // chunking of the pack kernel that produces the bucket sizes
  u64 img_lo = 0;                                // the records' images lie in [img_lo, img_lo + 2^ebits): digits come from
  u32 ebits = 0;                                 // image - img_lo, ebits wide (the whole range: 0, hm.nbits)
};
static constexpr u32 kMsdCapSmall = 2048, kMsdCapLarge = 4096;     // sub-bucket capacities of the two local-sort shapes
// Geometry for nrec words with hm's layout, or .on = false when the bucket ordering does not apply (switched off, too
// few records, 32-bit positions, or too few image bits below the bucket bits for the local sort's bins).
// img_lo / img_span: the records hold only the images in [img_lo, img_lo + img_span) (0 = the whole range).
static MsdGeom msd_geometry(const dc3hip_ctx *c, u32 nrec, const HiMap &hm, u64 img_lo = 0, u64 img_span = 0) {
  MsdGeom g;
  if (c->no_msd || nrec < c->msd_min || nrec < 4096 || hm.pbits >= 32 || hm.pbits + hm.nbits > 64) return g;
  g.img_lo = img_span ? img_lo : 0;
  g.ebits = img_span ? std::min<u32>(hm.nbits, bits_of(img_span - 1)) : hm.nbits;
  const u32 lg = bits_of((u64)nrec - 1);                       // ceil(log2 nrec)
  u32 tb = lg > 10 ? lg - 10 : 1;                              // sub-buckets of 512..1024 words on uniform images
  if (tb > 20) tb = 20;
  if (g.ebits < tb + 4) return g;
  if (tb <= 10) { g.d1 = tb; g.d2 = 0; } else { g.d1 = (tb + 1) / 2; g.d2 = tb - g.d1; }
  g.ntiles1 = (nrec + kMsdTile - 1) / kMsdTile;
  g.tpc = std::max<u32>(1, (g.ntiles1 + 2047) / 2048);
  g.cpg = ((g.ntiles1 + kMsdGroups - 1) / kMsdGroups + g.tpc - 1) / g.tpc;
  g.cpx1 = g.cpg * g.tpc;
  g.ck.chunk = g.tpc * (u32)kMsdTile;
  g.ck.nchunks = (nrec + g.ck.chunk - 1) / g.ck.chunk;
  g.on = true;
  return g;
}
// Pass 1 of a sort whose words are made on the fly from a key maker (k_msd_part_keys) instead of being read from `ha`:
// the pack kernel then only counted.  launch() = that kernel with the sort's geometry.
struct MsdPass1 {
  virtual ~MsdPass1() {}
  virtual int launch(dc3hip_ctx *c, u64 *out, u32 n, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) = 0;
  // a caller that has to give the bucket ordering up has no words to continue from (the pack kernel only counted):
  // repack() writes the plain words of all positions, in position order, with the LSD passes' first digit table
  virtual int repack(dc3hip_ctx *, Rec8 *, u32, u32 **) { set_err("internal: this pass 1 cannot repack"); return E_HIP; }
};
template <class KM> static int launch_pack_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table,
                                               const MsdGeom *mg = nullptr, bool store = true);
template <class KM>
struct MsdPass1Keys : MsdPass1 {
  KM km; HiMap hm; u64 P1 = 0;
  bool strip = false; HiMap hm_plain{};      // strip: hm is the WIDER image (hm.pbits = position bits - d1); hm_plain the words' own layout
  int repack(dc3hip_ctx *c, Rec8 *out, u32 nrec, u32 **first_table) override {
    PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
    return launch_pack_all<KM>(c, km, nrec, hm_plain, out, first_table, nullptr, true);
  }
  int launch(dc3hip_ctx *c, u64 *out, u32 n, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
      HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      attr_set[c->device & 15] = true;
    }
    if (strip)
      hipLaunchKernelGGL((k_msd_part_keys<KM, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, km, hm, P1, out, n, base, sh1,
                         g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon);
    else
      hipLaunchKernelGGL((k_msd_part_keys<KM, false>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, km, hm, P1, out, n, base, sh1,
                         g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon);
    KCHECK();
    return E_OK;
  }
};
// what a finished sort leaves behind so that its last pass can be repeated into records (cf. LastPass)
struct MsdRedo { const u64 *src = nullptr; u64 *dst = nullptr; const u32 *start = nullptr; u32 nsub = 0, shb = 0; bool large = false; u64 base = 0; };
template <class Sink>
static int msd_launch_local(dc3hip_ctx *c, const MsdRedo &r, u32 n, Sink sink) {
  PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, 6);
  if (r.large)
    hipLaunchKernelGGL((k_msd_local<512, (int)kMsdCapLarge, 12, Sink>), dim3(r.nsub), dim3(512), kMsdCapLarge * 8, c->stream, r.src,
                       r.start, r.base, r.shb, sink);
  else
    hipLaunchKernelGGL((k_msd_local<256, (int)kMsdCapSmall, 10, Sink>), dim3(r.nsub), dim3(256), kMsdCapSmall * 8, c->stream, r.src,
                       r.start, r.base, r.shb, sink);
  KCHECK();
  return E_OK;
}
// Sort the n words of `ha` (scratch `hb`) by image bits [pbits, pbits + nbits).  table = the pack kernel's digit table of
// the top g.d1 image bits ([1024][g.ck.nchunks]).  split != nullptr: the last pass writes positions + 32 image bits
// through it (as the LSD passes do with a SplitSink) and *result = nullptr; else *result = the sorted records.
// *ok = false: a sub-bucket was too large for the local sort — seen BEFORE pass 2 is launched, so `ha` still holds the
// caller's words in their original (position) order (*where = ha; with p1 they were never written: the caller repacks)
// and the stable LSD passes start from there, exactly as if the bucket ordering had not been tried: the order of equal
// images the tie pass meets does not depend on which way the sort went.  The small tables stay allocated in the arena
// until the caller releases its mark (redo reads them).
static int msd_sort(dc3hip_ctx *c, Rec8 *ha, Rec8 *hb, u32 n, const HiMap &hm, const MsdGeom &g, const u32 *table,
                    const SplitSink *split, Rec8 **result, MsdRedo *redo, bool *ok, Rec8 **where, MsdPass1 *p1 = nullptr,
                    uint8_t *same_out = nullptr) {
  // same_out (record form only): same_out[i] = 1 iff sorted record i has the image of record i - 1
  // p1 != nullptr: the words do not exist yet — pass 1 makes them from the key maker (`ha` is then only the scratch of
  // pass 2); needs `table` (the counting pack kernel's)
  *ok = false; *where = ha; *result = nullptr;
  if (p1 && !table) { set_err("internal: on-the-fly pass 1 without a digit table"); return E_HIP; }
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    attr_set[c->device & 15] = true;
  }
  const u32 nb1 = 1u << g.d1, tb = g.d1 + g.d2, n2 = 1u << tb;
  const u32 sh1 = hm.pbits + g.ebits - g.d1, sh2 = sh1 - g.d2, rb = g.ebits - tb;
  const u64 base = g.img_lo << hm.pbits;
  u32 *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *segsum = nullptr;
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cntg));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &startg));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb1 + 16, &bstart)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpre)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpreh));
  RC(arena_alloc(c, (size_t)kMsdW_COUNT + 12, &plan)); RC(arena_alloc(c, (size_t)1024 + 16, &segsum));
  u64 *wa = reinterpret_cast<u64 *>(ha), *wb = reinterpret_cast<u64 *>(hb);
  if (!table) {                              // records packed elsewhere: count the top digit here (one read of the records)
    u32 *t = nullptr;
    RC(arena_alloc(c, (size_t)kMsdMaxDig * g.ck.nchunks, &t));
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL(k_msd_hist1, dim3(g.ck.nchunks), dim3(kBlock), 0, c->stream, (const u64 *)wa, n, base, sh1, g.ck.chunk, g.ck.nchunks, t);
    KCHECK();
    table = t;
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
    HIPC(hipMemsetAsync(plan, 0, (kMsdW_COUNT + 12) * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_msd_cnt1, dim3(nb1), dim3(kBlock), 0, c->stream, table, g.ck.nchunks, g.cpg, cntg);
    KCHECK();
    hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb1, n, startg, cur1, bstart, tpre, tpreh, plan);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, p1 ? 9 : 5);     // (class 9: pass 1 that also makes the words, timed on its own)
    if (p1) {
      RC(p1->launch(c, wb, n, base, sh1, g, nb1, plan, cur1));
    } else {
      hipLaunchKernelGGL((k_msd_part<false>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, (const u64 *)wa, wb, n,
                         base, sh1, g.d1, g.cpx1, g.ntiles1, (const u32 *)nullptr, (const u32 *)nullptr, nb1, (const u32 *)plan, cur1, nb1, c->d_xcdmon);
      KCHECK();
    }
  }
  MsdRedo r;
  r.base = base;
  if (g.d2 > 0) {
    const size_t N = (size_t)n2 * kMsdGroups;
    u32 *cnt2g = nullptr, *cur2 = nullptr;
    RC(arena_alloc(c, N + 16, &cnt2g));
    RC(arena_alloc(c, N + 16, &cur2));
    const u32 nseg = (u32)((N + kMsdScanSeg - 1) / kMsdScanSeg);       // <= 1024
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
      HIPC(hipMemsetAsync(cnt2g, 0, (N + 1) * sizeof(u32), c->stream));
      hipLaunchKernelGGL(k_msd_hist2, dim3(n / kMsdHistTile + nb1 + 1), dim3(1024), 0, c->stream, (const u64 *)wb, base, sh2, g.d2,
                         (const u32 *)tpre, (const u32 *)tpreh, (const u32 *)bstart, nb1, (const u32 *)plan, cnt2g);
      KCHECK();
    }
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, N);
      hipLaunchKernelGGL(k_msd_scan2a, dim3(nseg), dim3(1024), 0, c->stream, (const u32 *)cnt2g, (u32)N, segsum, plan);
      KCHECK();
      hipLaunchKernelGGL(k_msd_scan2c, dim3(nseg), dim3(1024), 0, c->stream, cnt2g, (u32)N, n2, (const u32 *)segsum, cur2);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));                  // (pass 2 overwrites `ha`: decide first)
    if (c->h_words[20 + kMsdW_MAXSUB] > kMsdCapLarge) { c->stats.msd_max_subbucket = c->h_words[20 + kMsdW_MAXSUB]; c->stats.msd_fallbacks++; return E_OK; }
    {
      PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, 5);
      const u32 grid2 = kMsdGroups * ((n / kMsdTile + nb1 + 1 + kMsdGroups - 1) / kMsdGroups);
      hipLaunchKernelGGL((k_msd_part<true>), dim3(grid2), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, (const u64 *)wb, wa, n, base, sh2, g.d2,
                         0u, 0u, (const u32 *)tpre, (const u32 *)bstart, nb1, (const u32 *)plan, cur2, n2, c->d_xcdmon);
      KCHECK();
    }
    r.src = wa; r.dst = wb; r.start = cnt2g; r.nsub = n2;
  } else {
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
      hipLaunchKernelGGL(k_msd_scan2a, dim3((nb1 * kMsdGroups + kMsdScanSeg - 1) / kMsdScanSeg), dim3(1024), 0, c->stream, (const u32 *)cntg,
                         nb1 * kMsdGroups, segsum, plan);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    r.src = wb; r.dst = wa; r.start = startg; r.nsub = nb1;
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 maxsub = c->h_words[20 + kMsdW_MAXSUB];
  c->stats.msd_max_subbucket = maxsub;
  if (maxsub > kMsdCapLarge) { c->stats.msd_fallbacks++; return E_OK; }      // (d2 == 0: only `hb` was written)
  r.large = maxsub > kMsdCapSmall;
  r.shb = sh2 - std::min<u32>(r.large ? 12u : 10u, rb);
  if (split) {
    MsdSplitSink sk; sk.sa = split->sa; sk.same = split->same; sk.pbits = split->pbits;
    RC(msd_launch_local(c, r, n, sk));
  } else if (same_out) {
    MsdRecSameSink sk; sk.p = r.dst; sk.same = same_out; sk.pbits = hm.pbits;
    RC(msd_launch_local(c, r, n, sk));
    *result = reinterpret_cast<Rec8 *>(r.dst);
  } else {
    MsdRecSink sk; sk.p = r.dst;
    RC(msd_launch_local(c, r, n, sk));
    *result = reinterpret_cast<Rec8 *>(r.dst);
  }
  c->stats.msd_sorts++;
  *redo = r;
  *ok = true;
  return E_OK;
}
// repeat the last pass of an MSD sort that ended in a split sink, this time into records
static int msd_redo(dc3hip_ctx *c, const MsdRedo &r, u32 n, Rec8 **result) {
  MsdRecSink sk; sk.p = r.dst;
  RC(msd_launch_local(c, r, n, sk));
  *result = reinterpret_cast<Rec8 *>(r.dst);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// Splitter ordering of the sample-triple records (dc3_ssort.hip.hpp): the array radix_sort<Rec>(a, b, n, 0, kbits) makes
// from records in position order, in two partition passes over sampled splitters + an in-LDS order of the sub-buckets.
// *ok = false: not applied (too few records, switched off, or a sub-bucket beyond the local capacity — `a` is untouched
// in every such case and the caller runs the LSD passes).
// ---------------------------------------------------------------------------------------------
static constexpr u32 kSsCap = 4096;
// Measured on MI355X (1 GiB text): 318 M 16-byte records with 81-bit keys, 21 ms against 32 ms for the 63-bit
// prefix + tie rounds (and 9 LSD passes for the straight order); 477 M 12-byte records with 45-bit keys, 28 ms against
// 21 ms for the 5 LSD passes — so the keys of at most 64 bits stay with the LSD passes (DC3HIP_SSORT_REC12=1: tests).
static bool ssort_applies(const dc3hip_ctx *c, u32 n, u32 kbits) {
  return !c->no_ssort && n >= c->ssort_min && n >= 8192 && (kbits > 64 || c->ssort_rec12);
}
// A caller whose records do not exist yet hands in a producer: sample() computes S of them (ascending index),
// pack_count() makes all of them into `a` while it counts the coarse buckets (k_ss_count1's arguments).
struct SsProducer {
  virtual ~SsProducer() {}
  virtual int sample(dc3hip_ctx *c, u32 n, u32 S, void *out) = 0;
  virtual int pack_count(dc3hip_ctx *c, void *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb,
                         u32 grid, u32 *cntg, uint16_t *dig) = 0;
};
// nb1 coarse buckets x F2 sub-buckets of about ssort_mean records, S sample values; false: the ordering does not apply
struct SsGeom { u32 nb1, F2, n2, S; };
static bool ssort_geometry(const dc3hip_ctx *c, u32 n, u32 kbits, SsGeom *g, size_t rec_bytes = 16) {
  if (!ssort_applies(c, n, kbits)) return false;
  const u64 want = ((u64)n + c->ssort_mean - 1) / c->ssort_mean;           // sub-buckets
  u32 nb1 = kSsMaxDig, F2 = (u32)((want + nb1 - 1) / nb1);
  if (F2 < 2) { F2 = 2; nb1 = (u32)std::max<u64>(2, (want + 1) / 2); }
  if (F2 > kSsMaxDig) return false;                                        // (beyond 1.4e9 records)
  g->nb1 = nb1; g->F2 = F2; g->n2 = nb1 * F2; g->S = g->n2 * c->ssort_over;
  if ((u64)g->S * 4 > n) return false;
  // scratch on top of the caller's two record arrays: the sample twice, a digit per record, splitters and size tables;
  // when the arena cannot hold it the LSD passes run (arena_requirement() models those)
  const size_t need = 2 * (size_t)g->S * rec_bytes + (size_t)n * 2 + (size_t)g->n2 * (16 + 2 * 8 * 4) + ((size_t)48 << 20);
  return c->arena_bytes - c->arena_off >= need;
}
template <class Rec>
static int ssort(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 kbits, Rec **result, bool *ok, SsProducer *prod = nullptr) {
  // prod != nullptr: only when ssort_geometry() holds (the caller checked); on return `a` holds the records either way
  *ok = false; *result = nullptr;
  SsGeom geo;
  if (!ssort_geometry(c, n, kbits, &geo, sizeof(Rec))) {
    if (prod) { set_err("internal: splitter ordering with a producer outside its range"); return E_HIP; }
    return E_OK;
  }
  constexpr int IPT = SsCfg<Rec>::IPT;
  constexpr u32 tile = (u32)kSsNT * IPT, htile = tile * kSsHistTiles;
  constexpr int kLocNT = 1024, kLocIPT = (int)(kSsCap / kLocNT);
  constexpr size_t part_smem = ss_part_smem<Rec>(), loc_smem = sizeof(Rec) * kSsCap + kSsCap;
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss_part<Rec, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_smem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss_part<Rec, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_smem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_ss_local<Rec, kLocNT, kLocIPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)loc_smem));
    attr_set[c->device & 15] = true;
  }
  const u32 nb1 = geo.nb1, F2 = geo.F2, n2 = geo.n2, S = geo.S;
  const u32 ntiles1 = (n + tile - 1) / tile, cpx1 = (ntiles1 + kSsGroups - 1) / kSsGroups;
  const u32 tpb = std::max<u32>(1, (cpx1 + 255) / 256);
  const ArenaMark mk = arena_mark(c);
  Rec *sa = nullptr, *sb = nullptr, *ss = nullptr;
  SsVal *fine = nullptr, *coarse = nullptr;
  u32 *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *segsum = nullptr;
  u32 *cnt2g = nullptr, *cur2 = nullptr;
  uint16_t *dig = nullptr;
  const size_t N2 = (size_t)n2 * kSsGroups;
  RC(arena_alloc(c, (size_t)S, &sa)); RC(arena_alloc(c, (size_t)S, &sb));
  RC(arena_alloc(c, (size_t)n2 + 16, &fine)); RC(arena_alloc(c, (size_t)kSsMaxDig + 16, &coarse));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &cntg));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &startg));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb1 + 16, &bstart)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpre)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpreh));
  RC(arena_alloc(c, (size_t)kMsdW_COUNT + 12, &plan)); RC(arena_alloc(c, (size_t)1024 + 16, &segsum));
  RC(arena_alloc(c, N2 + 16, &cnt2g)); RC(arena_alloc(c, N2 + 16, &cur2));
  RC(arena_alloc(c, (size_t)n + 16, &dig));
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, S);
    if (prod) RC(prod->sample(c, n, S, sa));
    else {
      hipLaunchKernelGGL((k_ss_sample<Rec>), dim3((S + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (const Rec *)a, n, S, sa);
      KCHECK();
    }
  }
  RC(radix_sort<Rec>(c, sa, sb, S, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL((k_ss_splitters<Rec>), dim3((n2 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (const Rec *)ss, n2, F2, c->ssort_over, fine, coarse);
    KCHECK();
    HIPC(hipMemsetAsync(cntg, 0, ((size_t)nb1 * kSsGroups + 16) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(plan, 0, (kMsdW_COUNT + 12) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(cnt2g, 0, (N2 + 1) * sizeof(u32), c->stream));
    const u32 grid1 = kSsGroups * ((cpx1 + tpb - 1) / tpb);
    if (prod) RC(prod->pack_count(c, a, n, coarse, nb1, tile, cpx1, ntiles1, tpb, grid1, cntg, dig));
    else {
      hipLaunchKernelGGL((k_ss_count1<Rec>), dim3(grid1), dim3(kSsNT), 0, c->stream, (const Rec *)a, n,
                         (const SsVal *)coarse, nb1, tile, cpx1, ntiles1, tpb, cntg, dig);
      KCHECK();
    }
  }
  unsigned long long *vsum = nullptr;
  if (c->ssort_verify) {
    RC(arena_alloc(c, (size_t)8, &vsum));
    HIPC(hipMemsetAsync(vsum, 0, 8 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)a, n, vsum);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
    hipLaunchKernelGGL(k_ss_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb1, n, tile, htile, startg, cur1, bstart, tpre, tpreh, plan);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 7);
    hipLaunchKernelGGL((k_ss_part<Rec, false>), dim3(kSsGroups * cpx1), dim3(kSsNT), part_smem, c->stream, (const Rec *)a, b, n,
                       (const uint16_t *)dig, F2, cpx1, ntiles1, (const u32 *)nullptr, (const u32 *)nullptr, nb1, (const u32 *)plan, cur1, nb1);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)b, n, vsum + 4);
    KCHECK();
  }
  const u32 nseg = (u32)((N2 + kMsdScanSeg - 1) / kMsdScanSeg);              // <= 1024
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL((k_ss_hist2<Rec>), dim3(n / htile + nb1 + 1), dim3(kSsNT), 0, c->stream, (const Rec *)b, (const SsVal *)fine, F2, tile,
                       (const u32 *)tpre, (const u32 *)tpreh, (const u32 *)bstart, nb1, (const u32 *)plan, cnt2g, dig);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, N2);
    hipLaunchKernelGGL(k_msd_scan2a, dim3(nseg), dim3(1024), 0, c->stream, (const u32 *)cnt2g, (u32)N2, segsum, plan);
    KCHECK();
    hipLaunchKernelGGL(k_msd_scan2c, dim3(nseg), dim3(1024), 0, c->stream, cnt2g, (u32)N2, n2, (const u32 *)segsum, cur2);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 maxsub = c->h_words[20 + kMsdW_MAXSUB];
  c->stats.ssort_max_subbucket = maxsub;
  if (maxsub > kSsCap) { c->stats.ssort_fallbacks++; arena_release(c, mk); return E_OK; }     // (a is still the input)
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 7);
    const u32 grid2 = kSsGroups * ((n / tile + nb1 + 1 + kSsGroups - 1) / kSsGroups);
    hipLaunchKernelGGL((k_ss_part<Rec, true>), dim3(grid2), dim3(kSsNT), part_smem, c->stream, (const Rec *)b, a, n, (const uint16_t *)dig, F2, 0u, 0u,
                       (const u32 *)tpre, (const u32 *)bstart, nb1, (const u32 *)plan, cur2, n2);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)a, n, vsum + 6);
    KCHECK();
    unsigned long long *vc = nullptr;
    RC(arena_alloc(c, (size_t)8, &vc));
    HIPC(hipMemsetAsync(vc, 0, 8 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_ss_verify_cursors, dim3((u32)((N2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, (const u32 *)cnt2g, (const u32 *)cur2, n2, vc);
    KCHECK();
    unsigned long long hc[4];
    HIPC(hipMemcpyAsync(hc, vc, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (hc[0]) {
      set_err("DC3HIP_SSORT_VERIFY: after pass 2 %llu regions are off; first (sub-bucket %llu, group %llu): cursor %llu, expected %llu (n=%u F2=%u)",
              hc[0], hc[1] / 8, hc[1] % 8, hc[2], hc[3], n, F2);
      return E_HIP;
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 8);
    hipLaunchKernelGGL((k_ss_local<Rec, kLocNT, kLocIPT>), dim3(n2), dim3(kLocNT), loc_smem, c->stream, (const Rec *)a, (const u32 *)cnt2g, b);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)b, n, vsum + 2);
    KCHECK();
    unsigned long long h[8];
    HIPC(hipMemcpyAsync(h, vsum, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (h[0] != h[2] || h[3] != 0) {
      set_err("DC3HIP_SSORT_VERIFY: n=%u rec=%zu nb1=%u F2=%u S=%u checksums in=%llx pass1=%llx pass2=%llx out=%llx, %llu descents, largest sub-bucket %u",
              n, sizeof(Rec), nb1, F2, S, h[0], h[4], h[6], h[2], h[3], maxsub);
      return E_HIP;
    }
  }
  c->stats.ssort_sorts++;
  arena_release(c, mk);        // (the stream orders the kernels above before whatever reuses the scratch)
  *result = b;
  *ok = true;
  return E_OK;
}

// Digit table of a pack kernel (k_pack_image_*): bins, chunking and which image bits it counts.
// mg (bucket ordering, msd_geometry): the table counts the TOP mg->d1 image bits in mg's chunking instead (1024 rows).
static void pack_plan(dc3hip_ctx *c, u32 nrec, const HiMap &hm, const MsdGeom *mg, int *nb, Chunking *ck, u32 *hshift) {
  if (mg && mg->on) { *nb = 1024; *ck = mg->ck; *hshift = hm.nbits - mg->d1; }
  else { radix_plan<Rec8>(c, nrec, hm.nbits, nb, ck); *hshift = 0; }
}
#define DC3_PACK_LAUNCH(KERNEL_NB, ...)                                                                              \
  do {                                                                                                               \
    if (nb == 1024) hipLaunchKernelGGL(KERNEL_NB(1024), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__);   \
    else if (nb == 512) hipLaunchKernelGGL(KERNEL_NB(512), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL_NB(256), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__);               \
    KCHECK();                                                                                                        \
  } while (0)

// ---------------------------------------------------------------------------------------------
// out[key] = val for pairs whose keys are a bijection onto [0,n)  (R[SA12[i]] = i+1, lib.rs:106-108;
// SA12[R[i]-1] = i, lib.rs:111-113).  Two partition passes by the high key bits, then windows of
// 16384 destinations are assembled in LDS and stored with full lines.
// ---------------------------------------------------------------------------------------------
// `first`: the source of the first partition pass (PairArray of `a`, or pairs made on the fly — then the pass writes
// into `a` and `a`'s contents on entry do not matter).  Needs n > 2^14 when `first` is not `a` itself.
template <class Src>
static int inverse_permute_from(dc3hip_ctx *c, Src first, bool first_is_a, Rec8 *a, Rec8 *b, u32 n, u32 *out, int phase) {
  static std::atomic<bool> attr_set[16];   // (per function and device, process-wide; a double set is harmless)
  if (!attr_set[c->device & 15]) {
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_invperm_local),
                             hipFuncAttributeMaxDynamicSharedMemorySize, kInvWindow * 4));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_msd<Src>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kPartSmem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_msd<PairArray>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kPartSmem));
    attr_set[c->device & 15] = true;
  }
  const u32 kb = bits_of(n > 0 ? n - 1 : 0);
  const ArenaMark mk = arena_mark(c);
  const u32 ntiles = (n + kPartTile - 1) / kPartTile;
  // (a pass fed by `first` writes into a when first is not a itself, else into b)
  Rec8 *src = a, *dst = first_is_a ? b : a;
  bool at_first = true;
  if (kb > 22) {                       // pass 1: top digit = key >> 22 (<= 1024 values for n < 2^32)
    const u32 ndig = ((n - 1) >> 22) + 1;
    u32 *cur = nullptr;
    RC(arena_alloc(c, (size_t)1024, &cur));
    PhaseScope ps(c, phase, n, 3);
    HIPC(hipMemsetAsync(cur, 0, 1024 * sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_part_msd<Src>), dim3(ntiles), dim3(kPartNW * 64), kPartSmem, c->stream, first, dst, n, 22u, 32u,
                       ndig, cur, 0u);
    KCHECK();
    src = dst; dst = (src == a) ? b : a;
    at_first = false;
  }
  if (kb > (u32)kInvWindowBits) {      // pass 2: bits [14,22) inside every 2^22-pair segment
    const u32 nseg = kb > 22 ? ((n - 1) >> 22) + 1 : 1;
    u32 *cur = nullptr;
    RC(arena_alloc(c, (size_t)nseg * 256, &cur));
    PhaseScope ps(c, phase, n, 3);
    HIPC(hipMemsetAsync(cur, 0, (size_t)nseg * 256 * sizeof(u32), c->stream));
    // (with more than one 2^22-pair segment: segment s on the XCD group s % 8, see k_part_msd)
    const u32 tps = (1u << 22) / kPartTile;
    const bool xcd = kb > 22 && !c->no_xcd_map;
    const u32 grid = xcd ? 8u * ((nseg + 7) / 8) * tps : ntiles;
    if (at_first)
      hipLaunchKernelGGL((k_part_msd<Src>), dim3(grid), dim3(kPartNW * 64), kPartSmem, c->stream, first, dst, n,
                         (u32)kInvWindowBits, kb > 22 ? 22u : 32u, 256u, cur, xcd ? tps : 0u);
    else {
      PairArray pa; pa.p = src;
      hipLaunchKernelGGL((k_part_msd<PairArray>), dim3(grid), dim3(kPartNW * 64), kPartSmem, c->stream, pa, dst, n,
                         (u32)kInvWindowBits, kb > 22 ? 22u : 32u, 256u, cur, xcd ? tps : 0u);
    }
    KCHECK();
    src = dst; dst = (src == a) ? b : a;
    at_first = false;
  }
  if (at_first && !first_is_a) { set_err("inverse_permute_from: %u pairs are too few for an on-the-fly source", n); return E_ARGS; }
  {
    PhaseScope ps(c, phase, n);
    hipLaunchKernelGGL(k_invperm_local, dim3((n + kInvWindow - 1) / kInvWindow), dim3(1024), kInvWindow * 4,
                       c->stream, src, n, out);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}
static int inverse_permute(dc3hip_ctx *c, Rec8 *a, Rec8 *b, u32 n, u32 *out, int phase) {
  PairArray pa; pa.p = a;
  return inverse_permute_from<PairArray>(c, pa, true, a, b, n, out, phase);
}

// ---------------------------------------------------------------------------------------------
// naming + rank/name placement shared by both ordering paths (lib.rs:80-113).
//   unique names  -> sa12[i] = slot(pos_i), rank12 = inverse            (lib.rs:109-113)
//   otherwise     -> R[slot(pos_i)] = name_i (+ zero tail), caller recurses (lib.rs:93-104)
// ---------------------------------------------------------------------------------------------
static constexpr double kDiscardMinDropInv = 6.0;  // discard when ~1/6 of the slots would leave the recursion

struct Presort { const u32 *spos, *snf; };   // level-1 samples in sorted order + full names (whole-text sort)
template <class Sym>
static int dc3_level(dc3hip_ctx *c, Sym S, u32 m, u64 K, u32 *out_sa, u32 *out_rank, int depth,
                     const Presort *pre = nullptr);

// mode: 0 = names unique, sa12/rank12 complete; 1 = R holds the names, caller recurses on R (lib.rs:104);
//       2 = R holds name | unique<<31 and sslot the sorted slots: caller runs discard_recurse()
template <class Acc>
static int name_and_rank(dc3hip_ctx *c, Acc acc, u32 m02, u32 m0, u32 *sa12, u32 *rank12, u32 *R, u32 *sslot,
                         u32 *names_out, int *mode) {
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, m02, kBlock * kNameIPT);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_NAMING, m02);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, m02, ck.chunk, counts,
                       c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, 5 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));     // the lib.rs:103 decision needs the name count
  const u32 names = c->h_words[0], uniq = c->h_words[4];
  *names_out = names;
  Rec8 *pa = nullptr, *pb = nullptr;
  RC(arena_alloc(c, (size_t)m02, &pa));
  RC(arena_alloc(c, (size_t)m02, &pb));
  if (names == m02) {
    *mode = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_RANKS, m02);
      hipLaunchKernelGGL((k_assign_unique<Acc>), dim3(grid_for(c, m02)), dim3(kBlock), 0, c->stream, acc, m02, m0,
                         sa12, pa);
      KCHECK();
    }
    RC(inverse_permute(c, pa, pb, m02, rank12, DC3HIP_PH_RANKS));
  } else {
    // discard unique names from the recursion when enough slots would leave it to pay for the bookkeeping:
    // a unique slot is dropped iff its predecessor is unique too, so about uniq^2/m02 slots go
    const double drop_est = (double)uniq * (double)uniq / (double)m02;
    const bool discard = sslot && !c->no_discard && m02 < 0x7fffffffu && drop_est * kDiscardMinDropInv >= (double)m02 &&
                         c->arena_bytes - c->arena_off >= (size_t)m02 * 16 + (64u << 20);
    *mode = discard ? 2 : 1;
    {
      PhaseScope ps(c, DC3HIP_PH_NAMING, m02);
      hipLaunchKernelGGL((k_name_assign<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, m02, ck.chunk,
                         counts, m0, pa, discard ? sslot : (u32 *)nullptr);
      KCHECK();
    }
    RC(inverse_permute(c, pa, pb, m02, R, DC3HIP_PH_NAMING));
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, R, m02, 8u);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}

// Discarding recursion: see dc3_kernels.hip.hpp.  RU[p] = name | unique<<31 (slot order), sslot[i] =
// slot | unique<<31 (sorted order).  Recurses on the reduced string only; fills sa12 and rank12.
static int discard_recurse(dc3hip_ctx *c, const u32 *RU, const u32 *sslot, u32 m02, u32 names, u32 *sa12,
                           u32 *rank12, int depth) {
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, m02, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 mp = 0;
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 5);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 5, c->d_words + 5, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  mp = c->h_words[5];
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
  RC(dc3_level<SymU32>(c, RS, mp, names, sap, nullptr, depth + 1));   // m == 1 is the child's base case
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
  Rec8 *pa = nullptr, *pb = nullptr;
  RC(arena_alloc(c, (size_t)m02, &pa));
  RC(arena_alloc(c, (size_t)m02, &pb));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sslot, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL(k_final_assign, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sslot, m02, ck.chunk, counts, pt,
                       sa12, pa);
    KCHECK();
  }
  RC(inverse_permute(c, pa, pb, m02, rank12, DC3HIP_PH_RANKS));
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// prefix-sort + tie-refine ordering (see dc3_kernels.hip.hpp).  Policy:
//   * a strided sample of ~2^20 triples predicts the fraction of samples whose N-bit key image
//     collide; the path is taken when the prediction is below kHybridMaxPredicted,
//   * and abandoned (falling back to the straight 16-byte LSD sort) if the measured fraction turns
//     out above kHybridMaxMeasured.  Correctness never depends on the policy.
// ---------------------------------------------------------------------------------------------
static constexpr u32 kHybridMinSamples = 1u << 22;
// (kHybridMaxPredicted = dc3hip_ctx::hybrid_max_pred = 0.50)
static constexpr double kHybridMaxMeasured = 0.60;
static constexpr double kFullSortMaxPredicted = 0.10;   // whole-level shortcut only for very few predicted ties
static constexpr double kTextSortMaxPredicted = 0.30;   // whole-text shortcut (33-bit images at 2^30 bytes tie ~12 %)
static constexpr double kTextSortMaxBirthday = 0.55;    // ... or more, if the image width alone explains the ties
// Whole-text shortcut: go when few image ties are predicted, or when the predicted ties are no more than what a
// uniformly random text has at this image width (1 - exp(-n / 2^nbits): the 32-bit images of 2^31 positions tie 39 %
// and the tie pass still costs far less than the recursion), which says the text itself is not repetitive.
static bool text_order_worth_trying(double pred, u64 n, u32 nbits) {
  if (pred < kTextSortMaxPredicted) return true;
  const double birthday = 1.0 - exp(-(double)n / ldexp(1.0, (int)nbits));
  return pred < kTextSortMaxBirthday && pred <= 1.25 * birthday + 0.02;
}
// hi = floor(X * mfix / 2^64) in N = min(64 - pbits, kbits) bits; X = key >> shx; see HiMap
static HiMap make_himap(u64 B, u32 kbits, u32 m, u32 pbits = 0) {
  const unsigned __int128 mx = (unsigned __int128)B * B * B - 1;      // largest key
  HiMap hm;
  hm.pbits = pbits ? pbits : bits_of((u64)m + 2);
  hm.nbits = std::min<u32>(64 - hm.pbits, kbits);
  hm.exact = kbits <= hm.nbits ? 1u : 0u;
  hm.shx = kbits > 64 ? kbits - 64 : 0;
  hm.mfix = 0;
  if (!hm.exact) {
    const unsigned __int128 xmax1 = (mx >> hm.shx) + 1;                // > 2^nbits
    const unsigned __int128 num = (((unsigned __int128)1) << (64 + hm.nbits)) - 1;
    hm.mfix = (u64)(num / xmax1);
  }
  return hm;
}

static int count_ties(dc3hip_ctx *c, const Rec8 *h, u32 n, u32 pbits, u32 *counts, const Chunking &ck, u32 *total) {
  hipLaunchKernelGGL(k_tie_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, h, n, ck.chunk, pbits, counts);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  *total = c->h_words[2];
  return E_OK;
}

// sample records whose key image equals another sample's (hash table in the arena; see k_hash_ties)
static int sample_ties(dc3hip_ctx *c, const Rec8 *a, u32 ns, u32 pbits, u32 *ts) {
  u32 slots = 1; while (slots < 2 * ns) slots <<= 1;
  unsigned long long *table = nullptr;
  RC(arena_alloc(c, (size_t)slots, &table));
  HIPC(hipMemsetAsync(table, 0, (size_t)slots * sizeof(unsigned long long), c->stream));
  HIPC(hipMemsetAsync(c->d_words + 2, 0, sizeof(u32), c->stream));
  hipLaunchKernelGGL(k_hash_ties, dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, a, ns, pbits, table, slots - 1,
                     c->d_words + 2);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  *ts = c->h_words[2];
  return E_OK;
}

template <class Sym>
static int predict_tie_fraction(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, HiMap sh, double *pred) {
  const ArenaMark mk = arena_mark(c);
  const u32 stride = std::max<u32>(1, m0 >> 19);
  const u32 ng = (m0 - 1) / stride + 1;      // sampled groups, 2 records each
  const u32 ns = 2 * ng;
  Rec8 *a = nullptr;
  RC(arena_alloc(c, (size_t)ns, &a));
  PhaseScope ps(c, DC3HIP_PH_PACK, ns);
  hipLaunchKernelGGL((k_pack_image<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, sh,
                     stride, ng, a);
  KCHECK();
  u32 ts = 0;
  RC(sample_ties(c, a, ns, sh.pbits, &ts));
  const double fs = (double)ts / (double)ns;
  const double ratio = (double)(m02 - 1) / (double)(ns > 1 ? ns - 1 : 1);
  *pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
  arena_release(c, mk);
  return E_OK;
}

// straight ordering: full-key records (12 bytes when the key fits 64 bits, else 16), LSD over all key bits
template <class Sym, class Rec>
static int order_straight(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12, u32 *rank12,
                          u32 *R, u32 *sslot, u32 *names, int *mode) {
  Rec *recA = nullptr, *recB = nullptr, *sorted = nullptr;
  RC(arena_alloc(c, (size_t)m02, &recA));
  RC(arena_alloc(c, (size_t)m02, &recB));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck;
    radix_plan<Rec>(c, m02, kbits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    if (nb == 512)
      hipLaunchKernelGGL((k_pack_triples_hist<Sym, Rec, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         b, recA, ck.chunk, ck.nchunks, first_table);
    else
      hipLaunchKernelGGL((k_pack_triples_hist<Sym, Rec, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         b, recA, ck.chunk, ck.nchunks, first_table);
    KCHECK();
  }
  bool by_splitters = false;
  RC(ssort<Rec>(c, recA, recB, m02, kbits, &sorted, &by_splitters));
  if (!by_splitters)
    RC(radix_sort<Rec>(c, recA, recB, m02, 0, kbits, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                       DC3HIP_PH_SORT12_DOWN, first_table));
  AccRec<Rec> acc; acc.s = sorted;
  return name_and_rank<AccRec<Rec>>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode);
}

// Straight ordering with a wider window than the triple (dc3_ssort.hip.hpp): W = floor(96 / sb) symbols (4..7) in
// 16-byte records, ordered by the splitter ordering whose cost does not depend on the key width.  For the levels whose
// triples repeat everywhere (text: the 3-symbol names of level 0 make a level-1 string whose triples are 9 characters):
// their names would send a string of the same length down the recursion; W symbols settle most samples here.
static u32 wide_window_syms(const dc3hip_ctx *c, u32 m02, u64 K) {
  const u32 sb = bits_of(K);
  if (c->no_wide_window || c->no_hybrid || !ssort_applies(c, m02, 96) || sb > 24) return 0;
  return std::min<u32>(7, 96 / sb);             // (the zero tail behind a level's string is 8 symbols)
}
template <class Sym>
struct WideProducer : SsProducer {
  Sym S; u32 sb, W;
  template <int WW> int sample_w(dc3hip_ctx *c, u32 n, u32 Sn, Rec16 *out) {
    hipLaunchKernelGGL((k_ss_sample_window<Sym, WW>), dim3((Sn + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, S, sb, n, Sn, out);
    KCHECK();
    return E_OK;
  }
  template <int WW> int pack_w(dc3hip_ctx *c, Rec16 *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 grid,
                               u32 *cntg, uint16_t *dig) {
    hipLaunchKernelGGL((k_ss_pack_count1<Sym, WW>), dim3(grid), dim3(kSsNT), 0, c->stream, S, sb, a, n, coarse, nb1, tile, cpx, ntiles, tpb, cntg, dig);
    KCHECK();
    return E_OK;
  }
  int sample(dc3hip_ctx *c, u32 n, u32 Sn, void *out) override {
    Rec16 *o = static_cast<Rec16 *>(out);
    switch (W) { case 4: return sample_w<4>(c, n, Sn, o); case 5: return sample_w<5>(c, n, Sn, o); case 6: return sample_w<6>(c, n, Sn, o); default: return sample_w<7>(c, n, Sn, o); }
  }
  int pack_count(dc3hip_ctx *c, void *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 grid, u32 *cntg,
                 uint16_t *dig) override {
    Rec16 *r = static_cast<Rec16 *>(a);
    switch (W) {
      case 4: return pack_w<4>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      case 5: return pack_w<5>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      case 6: return pack_w<6>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      default: return pack_w<7>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
    }
  }
};
template <class Sym>
static int order_wide(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 sb, u32 W, u32 *sa12, u32 *rank12, u32 *R,
                      u32 *sslot, u32 *names, int *mode) {
  Rec16 *recA = nullptr, *recB = nullptr, *sorted = nullptr;
  RC(arena_alloc(c, (size_t)m02, &recA));
  RC(arena_alloc(c, (size_t)m02, &recB));
  bool by_splitters = false;
  SsGeom geo;
  if (ssort_geometry(c, m02, W * sb, &geo) && !c->no_pack_count) {
    // the records are made by the kernel that counts the coarse buckets (written once, not read back for the count)
    WideProducer<Sym> prod; prod.S = S; prod.sb = sb; prod.W = W;
    RC(ssort<Rec16>(c, recA, recB, m02, W * sb, &sorted, &by_splitters, &prod));
  } else {
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, m02);
      const dim3 grid((m02 / 2 + kBlock) / kBlock);
      switch (W) {
        case 4: hipLaunchKernelGGL((k_pack_window16<Sym, 4>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        case 5: hipLaunchKernelGGL((k_pack_window16<Sym, 5>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        case 6: hipLaunchKernelGGL((k_pack_window16<Sym, 6>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        default: hipLaunchKernelGGL((k_pack_window16<Sym, 7>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
      }
      KCHECK();
    }
    RC(ssort<Rec16>(c, recA, recB, m02, W * sb, &sorted, &by_splitters));
  }
  if (!by_splitters)
    RC(radix_sort<Rec16>(c, recA, recB, m02, 0, W * sb, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
  AccRec<Rec16> acc; acc.s = sorted;
  return name_and_rank<AccRec<Rec16>>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode);
}

static constexpr u32 kDeepSyms = 2048;                     // symbols compared by the second tie pass of a whole-text order
template <class KM, class Acc>
static int doubling_finish(dc3hip_ctx *c, KM km, Acc acc, u32 n, u32 W, u32 *out_sa, bool *done, bool order_in_place = false);
// key makers of the whole-text order (they know their window; Key3 is a level's triple)
template <class KM> struct IsTextKey { static constexpr bool value = false; };
template <> struct IsTextKey<Key9> { static constexpr bool value = true; };
template <> struct IsTextKey<KeyT> { static constexpr bool value = true; };

// Core of the prefix-sort + tie-refine ordering: `ha` holds nrec packed (image << pbits | pos) records of the
// positions to order; on return (ok) h = records sorted by the full key, f[i] = key differs from predecessor.
template <class KM>
static int hybrid_sort_core(dc3hip_ctx *c, KM km, u32 kbits, const HiMap &hm, Rec8 *ha, Rec8 *hb, u32 nrec,
                            Rec8 **h_out, uint8_t *f, bool *ok, int depth, u32 *emit_sa = nullptr, u32 skip = 0,
                            bool *emitted_distinct = nullptr, u32 *first_table = nullptr, bool whole_text = false,
                            const MsdGeom *mg = nullptr, u64 img_lo = 0, u64 img_span = 0, MsdPass1 *p1 = nullptr,
                            bool *keys_distinct = nullptr) {
  // keys_distinct (record form): set when the tie pass settled every tied group and found no two equal keys — the caller
  // then knows that all nrec keys are distinct without counting the flags
  // p1 (only with mg->on): the records of `ha` were NOT written — the pack kernel only counted, pass 1 of the bucket
  // ordering makes them on the fly
  // img_lo / img_span (only with mg == nullptr): the records hold the images of [img_lo, img_lo + img_span) only
  // mg (and mg->on): the records were packed for the bucket ordering — first_table is then the digit table of the TOP
  // image bits in mg's chunking, and the sort runs msd_sort(); should that give up, the LSD passes start from scratch.
  // mg == nullptr (callers that build their records elsewhere): the bucket ordering counts its top digit itself.
  MsdGeom mg_self;
  if (!mg) {
    mg_self = msd_geometry(c, nrec, hm, img_lo, img_span);
    if (mg_self.on) { mg = &mg_self; first_table = nullptr; }
  }
  // whole_text: the records are ALL positions of the text (single device): few repeated windows may be settled here by
  // prefix doubling.  (A rank of the global mode orders only its image range and must not: ranks are global.)
  *ok = false;
  if (emitted_distinct) *emitted_distinct = false;
  if (keys_distinct) *keys_distinct = false;
  Rec8 *h = nullptr;
  bool msd_ok = false;                 // (record form) the bucket ordering delivered, with its same-image bytes in same_rec
  uint8_t *same_rec = nullptr;
  if (emit_sa && emitted_distinct && skip == 0 && !c->no_small_ties && !c->no_split_emit && hm.pbits < 32) {
    // optimistic end of the whole-text order: the last pass writes positions to the SA buffer and 32 image bits to a
    // side array; the tie pass settles the tied groups in place.  Complete unless a key repeats or a group is large.
    // (what the tie pass reads: a "same image as the record before" byte from the bucket ordering, or the 32 image bits
    //  the LSD passes leave when that ordering does not apply or gave up; the image array is only touched in that case)
    u32 *img = nullptr;
    uint8_t *same = nullptr;
    RC(arena_alloc(c, (size_t)nrec + 16, &img));
    RC(arena_alloc(c, (size_t)nrec + 16, &same));
    SplitSink sink; sink.sa = emit_sa; sink.img = img; sink.same = same; sink.pbits = hm.pbits;
    LastPass lp;
    MsdRedo mredo; bool msd_ok = false;
    if (mg && mg->on) {
      Rec8 *where = ha;
      RC(msd_sort(c, ha, hb, nrec, hm, *mg, first_table, &sink, &h, &mredo, &msd_ok, &where, p1));
      if (msd_ok) lp.src = const_cast<u64 *>(mredo.src);               // (non-null = "the order lives in the sink")
      else { first_table = nullptr; if (p1) RC(p1->repack(c, ha, nrec, &first_table)); }      // from scratch: `ha` in position order
    }
    if (!msd_ok)
      RC(radix_sort<Rec8>(c, ha, hb, nrec, hm.pbits, hm.pbits + hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                          DC3HIP_PH_SORT8_DOWN, first_table, &sink, &lp));
    if (lp.src) {
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
        if (msd_ok)
          hipLaunchKernelGGL((k_tie_resolve_split<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km,
                             SameFlag{same}, emit_sa, nrec, c->d_words + 10);
        else
          hipLaunchKernelGGL((k_tie_resolve_split<KM, SameImg>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km,
                             SameImg{img}, emit_sa, nrec, c->d_words + 10);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      c->stats.level_tied[depth] = c->h_words[11];
      if ((double)c->h_words[11] > kHybridMaxMeasured * (double)nrec) return E_OK;   // *ok stays false -> straight LSD
      if (c->h_words[10] == 0 && c->h_words[12] == 0) { *emitted_distinct = true; *h_out = nullptr; *ok = true; return E_OK; }
      if constexpr (IsTextKey<KM>::value) {
        // few windows repeat: a second tie pass that compares kDeepSyms symbols instead of the window settles the repeats
        // shorter than that (the compare is lazy: the depth only costs where windows really agree that far) — single
        // device and global mode alike
        if (c->h_words[10] == 0 && c->h_words[12] <= nrec / 4096 + 16 && !c->no_doubling) {
          KM kd = km; kd.deep = kDeepSyms;
          {
            PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
            HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
            if (msd_ok)
              hipLaunchKernelGGL((k_tie_resolve_split<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, kd,
                                 SameFlag{same}, emit_sa, nrec, c->d_words + 10);
            else
              hipLaunchKernelGGL((k_tie_resolve_split<KM, SameImg>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, kd,
                                 SameImg{img}, emit_sa, nrec, c->d_words + 10);
            KCHECK();
            HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
          }
          HIPC(hipStreamSynchronize(c->stream));
          if (c->h_words[10] == 0 && c->h_words[12] == 0) { *emitted_distinct = true; *h_out = nullptr; *ok = true; return E_OK; }
        }
        // few windows repeat and no group was too large for the tie pass: the positions are in window order in the SA
        // buffer; flag the window changes and let the prefix doubling finish from there (no records needed)
        if (whole_text && c->h_words[10] == 0 && c->h_words[12] <= nrec / 128 && !c->no_doubling && depth == 0) {
          {
            PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
            if (msd_ok)
              hipLaunchKernelGGL((k_split_flags<KM, SameFlag>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, km, SameFlag{same},
                                 (const u32 *)emit_sa, nrec, f);
            else
              hipLaunchKernelGGL((k_split_flags<KM, SameImg>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, km, SameImg{img},
                                 (const u32 *)emit_sa, nrec, f);
            KCHECK();
          }
          // (the doubling reads the order from the very buffer whose tied slots it rewrites: a slot of a tied group always
          //  holds SOME member of that group, whose window — all the binary searches look at — is the group's)
          AccSplit acc; acc.sa = emit_sa; acc.f = f;
          bool finished = false;
          RC((doubling_finish<KM, AccSplit>(c, km, acc, nrec, km.window_syms(), emit_sa, &finished, true)));
          if (finished) { *emitted_distinct = true; *h_out = nullptr; *ok = true; c->stats.level_sorted[0] = 6; return E_OK; }
        }
      }
      // keys repeat (or a large group): the records are needed after all
      if (msd_ok) RC(msd_redo(c, mredo, nrec, &h));
      else RC(radix_redo_last(c, lp, nrec, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT8_DOWN));
    }
    emit_sa = nullptr;                       // from here on: the record path, positions are emitted by the caller
  } else {
    if (mg && mg->on) {
      MsdRedo mredo; Rec8 *where = ha;
      if (!c->no_small_ties) RC(arena_alloc(c, (size_t)nrec + 16, &same_rec));
      RC(msd_sort(c, ha, hb, nrec, hm, *mg, first_table, nullptr, &h, &mredo, &msd_ok, &where, p1, same_rec));
      if (!msd_ok) { first_table = nullptr; if (p1) RC(p1->repack(c, ha, nrec, &first_table)); }
    }
    if (!msd_ok)
      RC(radix_sort<Rec8>(c, ha, hb, nrec, hm.pbits, hm.pbits + hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                          DC3HIP_PH_SORT8_DOWN, first_table));
  }
  const Chunking ck = make_chunks(c, nrec, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 tied = 0;
  bool general = false;
  HIPC(hipMemsetAsync(f, 1, (size_t)nrec, c->stream));
  if (!c->no_small_ties) {
    // one in-place pass counts the tied records and settles every tied group of at most kTieSmallMax members
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
      HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
      if (msd_ok && same_rec)
        hipLaunchKernelGGL((k_tie_resolve<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km, SameFlag{same_rec}, h, nrec,
                           hm.pbits, f, c->d_words + 10, emit_sa, skip);
      else
        hipLaunchKernelGGL((k_tie_resolve<KM, SameRec>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km, SameRec{h, hm.pbits}, h, nrec,
                           hm.pbits, f, c->d_words + 10, emit_sa, skip);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    tied = c->h_words[11];
    c->stats.level_tied[depth] = tied;
    if ((double)tied > kHybridMaxMeasured * (double)nrec) return E_OK;   // *ok stays false -> straight LSD
    general = c->h_words[10] != 0;       // some group is larger: redo the ties with the general path
    if (!general && emit_sa && emitted_distinct && c->h_words[12] == 0) *emitted_distinct = true;
    if (!general && keys_distinct && c->h_words[12] == 0) *keys_distinct = true;
  }
  if (general || c->no_small_ties) {
    PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
    RC(count_ties(c, h, nrec, hm.pbits, counts, ck, &tied));
    c->stats.level_tied[depth] = tied;
    if ((double)tied > kHybridMaxMeasured * (double)nrec) return E_OK;
    general = tied > 0;
  }
  if (general) {
    // the tied subset is re-sorted as 16-byte records (2 x 16 B + index + radix tables): when the arena cannot hold
    // that on top of what the caller holds, give the ordering up (*ok stays false -> the caller's next ordering
    // runs instead); arena_requirement() only bounds the straight ordering
    if (c->arena_bytes - c->arena_off < (size_t)tied * 36 + (32u << 20)) return E_OK;
    const ArenaMark mk_general = arena_mark(c);     // the tied subset is dead after the write-back: released there
    Rec16 *sa = nullptr, *sb = nullptr, *ss = nullptr;
    u32 *tiedidx = nullptr;
    RC(arena_alloc(c, (size_t)tied, &sa));
    RC(arena_alloc(c, (size_t)tied, &sb));
    RC(arena_alloc(c, (size_t)tied, &tiedidx));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL((k_tie_compact<KM>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, h, nrec, ck.chunk,
                         hm.pbits, counts, sa, tiedidx);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, sa, sb, tied, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                         DC3HIP_PH_SORT12_DOWN));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL(k_tie_writeback, dim3(grid_for(c, tied)), dim3(kBlock), 0, c->stream, ss, tiedidx, tied, h, f);
      KCHECK();
    }
    arena_release(c, mk_general);
  }
  *h_out = h;
  *ok = true;
  return E_OK;
}

// Tie refinement of records sorted by their 63-bit key prefix (see order_hybrid12): f[i] = full key differs from the
// predecessor's, tied groups ordered by the full key.  *ok = false: too many ties, or no room for the general path.
template <class KM>
static int hybrid12_refine(dc3hip_ctx *c, KM km, u32 kbits, Rec12 *h, u32 n, uint8_t *f, bool *ok, int depth,
                           u32 *emit_sa = nullptr, bool *distinct = nullptr, bool *deep_flags = nullptr) {
  // *deep_flags: on return f[] (and the order inside tied groups) reflects equality over kDeepSyms symbols, not over
  // the window — whoever continues from f[] (doubling_finish) must compare at the same depth
  *ok = false;
  if (distinct) *distinct = false;
  if (deep_flags) *deep_flags = false;
  bool deep_ran = false;
  HIPC(hipMemsetAsync(f, 1, (size_t)n, c->stream));
  u32 tied = 0;
  bool general = false;
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, n);
    HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_tie_resolve12<KM>), dim3(grid_for(c, n / 4 + 1)), dim3(kBlock), 0, c->stream, km, h, n, f,
                       c->d_words + 10, emit_sa);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  tied = c->h_words[11];
  c->stats.level_tied[depth] = tied;
  general = c->h_words[10] != 0;
  if constexpr (IsTextKey<KM>::value) {
    // whole-text order: a few windows agree completely -> the tie pass once more, comparing kDeepSyms symbols (see
    // hybrid_sort_core); settles the repeats shorter than that
    if (emit_sa && !general && c->h_words[12] > 0 && c->h_words[12] <= n / 4096 + 16 && !c->no_doubling) {
      KM kd = km; kd.deep = kDeepSyms;
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, n);
        HIPC(hipMemsetAsync(f, 1, (size_t)n, c->stream));
        HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
        hipLaunchKernelGGL((k_tie_resolve12<KM>), dim3(grid_for(c, n / 4 + 1)), dim3(kBlock), 0, c->stream, kd, h, n, f,
                           c->d_words + 10, emit_sa);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      tied = c->h_words[11];
      general = c->h_words[10] != 0;
      deep_ran = true;
    }
  }
  // (the general path below re-sorts every tied record by the window's full key and rewrites f[] from it)
  if (deep_flags) *deep_flags = deep_ran && !general;
  // no group overflowed and no full key repeats: the positions the tie pass wrote to emit_sa are the sorted order
  if (distinct) *distinct = emit_sa && !general && c->h_words[12] == 0;
  if ((double)tied > std::max(kHybridMaxMeasured, c->hybrid12_max_pred + 0.1) * (double)n) return E_OK;
  if (general) {
    // some tied group is larger than kTieSmallMax: re-sort ALL tied records by the full key (the small groups that were
    // already settled are re-done consistently)
    if (c->arena_bytes - c->arena_off < (size_t)tied * 36 + (32u << 20)) return E_OK;
    const ArenaMark mk_general = arena_mark(c);     // the tied subset is dead after the write-back: released there
    const Chunking ck = make_chunks(c, n, kBlock);
    u32 *counts = nullptr, *tiedidx = nullptr;
    Rec16 *sa = nullptr, *sb = nullptr, *ss = nullptr;
    RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, n);
      hipLaunchKernelGGL(k_tie_count12, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, h, n, ck.chunk, counts);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    tied = c->h_words[2];
    RC(arena_alloc(c, (size_t)tied, &sa));
    RC(arena_alloc(c, (size_t)tied, &sb));
    RC(arena_alloc(c, (size_t)tied, &tiedidx));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL((k_tie_compact12<KM>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, h, n, ck.chunk, counts,
                         sa, tiedidx);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, sa, sb, tied, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL(k_tie_writeback12, dim3(grid_for(c, tied)), dim3(kBlock), 0, c->stream, ss, tiedidx, tied, h, f);
      KCHECK();
    }
    arena_release(c, mk_general);
  }
  *ok = true;
  return E_OK;
}

// Prefix sort + tie refinement on 12-byte records (kernels: "Prefix sort ... on 12-byte records" in dc3_order.hip.hpp):
// for keys wider than 64 bits.  *ok = false: too many ties (predicted or measured), nothing was produced.
template <class Sym>
static int order_hybrid12(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12, u32 *rank12, u32 *R,
                          u32 *sslot, u32 *names, int *mode, bool *ok, int depth) {
  *ok = false;
  const ArenaMark mk = arena_mark(c);
  Key3<Sym> km; km.S = S; km.B = b;
  // predicted fraction of samples whose 63-bit prefix collides with another sample's
  {
    const u32 stride = std::max<u32>(1, m0 >> 19);
    const u32 ng = (m0 - 1) / stride + 1, ns = 2 * ng;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    u32 ts = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, ns);
      hipLaunchKernelGGL((k_pack_image12_sample<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m02, b, kbits,
                         stride, ng, a);
      KCHECK();
    }
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    const double ratio = (double)(m02 - 1) / (double)(ns > 1 ? ns - 1 : 1);
    const double pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
    c->stats.level_tie_pred[depth] = pred;
    arena_release(c, mk);
    if (!(pred < c->hybrid12_max_pred)) return E_OK;
  }
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)m02, &ha));
  RC(arena_alloc(c, (size_t)m02, &hb));
  RC(arena_alloc(c, (size_t)m02 + 16, &f));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck;
    radix_plan<Rec12>(c, m02, kImg12Bits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    if (nb == 512)
      hipLaunchKernelGGL((k_pack_image12_hist<Sym, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, kbits,
                         ha, ck.chunk, ck.nchunks, first_table);
    else
      hipLaunchKernelGGL((k_pack_image12_hist<Sym, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, kbits,
                         ha, ck.chunk, ck.nchunks, first_table);
    KCHECK();
  }
  RC(radix_sort<Rec12>(c, ha, hb, m02, 0, kImg12Bits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN,
                       first_table));
  {
    bool refined = false;
    RC((hybrid12_refine<Key3<Sym>>(c, km, kbits, h, m02, f, &refined, depth)));
    if (!refined) { arena_release(c, mk); return E_OK; }
  }
  c->stats.level_sorted[depth] = 2;
  AccHyb12 acc; acc.h = h; acc.f = f;
  RC(name_and_rank<AccHyb12>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode));
  *ok = true;
  arena_release(c, mk);
  return E_OK;
}

// tie-rate predictor over all positions (stride sample) for the whole-text shortcut of level 0
template <class KM>
static int predict_tie_fraction_pos(dc3hip_ctx *c, KM km, u32 n, const HiMap &hm, double *pred) {
  const ArenaMark mk = arena_mark(c);
  const u32 stride = std::max<u32>(1, n >> 20);
  const u32 ns = (n - 1) / stride + 1;
  Rec8 *a = nullptr;
  RC(arena_alloc(c, (size_t)ns, &a));
  PhaseScope ps(c, DC3HIP_PH_PACK, ns);
  hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
  KCHECK();
  u32 ts = 0;
  RC(sample_ties(c, a, ns, hm.pbits, &ts));
  const double fs = (double)ts / (double)ns;
  const double ratio = (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1);
  *pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
  arena_release(c, mk);
  return E_OK;
}

template <class Sym>
static int order_hybrid(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12,
                        u32 *rank12, u32 *R, u32 *sslot, u32 *names, int *mode, bool *ok, int depth) {
  *ok = false;
  const HiMap hm = make_himap((u64)b, kbits, m);
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)m02, &ha));
  RC(arena_alloc(c, (size_t)m02, &hb));
  RC(arena_alloc(c, (size_t)m02 + 16, &f));
  u32 *first_table = nullptr;
  const MsdGeom mg = msd_geometry(c, m02, hm);
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck; u32 hshift = 0;
    pack_plan(c, m02, hm, &mg, &nb, &ck, &hshift);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
#define K_(NB) (k_pack_image_hist<Sym, NB>)
    DC3_PACK_LAUNCH(K_, S, m, m0, m02, b, hm, ha, ck.chunk, ck.nchunks, first_table, hshift);
#undef K_
  }
  bool sorted_ok = false;
  Key3<Sym> km; km.S = S; km.B = b;
  RC((hybrid_sort_core<Key3<Sym>>(c, km, kbits, hm, ha, hb, m02, &h, f, &sorted_ok, depth, nullptr, 0, nullptr,
                                  first_table, false, &mg)));
  if (!sorted_ok) return E_OK;
  c->stats.level_sorted[depth] = 2;
  AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
  RC(name_and_rank<AccHyb>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode));
  *ok = true;
  return E_OK;
}

// Whole-level shortcut for high-entropy levels: order ALL m positions (plus the dummy sample) by their triple.
//   state 1: every triple distinct -> the result is the suffix array of the level (suffixes differ within 3
//            symbols): sampling, tuples and the merge (lib.rs:62-192) are skipped altogether;
//   state 2: duplicates exist -> the samples are filtered out of the sorted order (spos/snf), so the usual
//            naming continues from there and the sort is not repeated;
//   state 0: too many collisions in the key image, nothing was produced.
// spos/snf (m02 entries each) must be allocated by the caller below this function's arena mark.
// Packs the records of all positions; *first_table != nullptr on return when the kernel also produced the digit
// table of the first radix pass (whole text: k_pack_image_text).
// store = false (only with mg->on): count only — pass 1 of the bucket ordering makes the records on the fly (MsdPass1Keys).
template <class KM>
static int launch_pack_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table,
                           const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  if (!store) {
    hipLaunchKernelGGL((k_pack_image_all_hist<KM, 1024, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out, ck.chunk,
                       ck.nchunks, table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_all_hist<KM, NB>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
template <>
int launch_pack_all<Key9>(dc3hip_ctx *c, Key9 km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  if (!store) {
    hipLaunchKernelGGL((k_pack_image_text<1024, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out, ck.chunk, ck.nchunks,
                       table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_text<NB>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
static u64 keyt_p1(const KeyT &km) {
  u64 P1 = 1;
  for (u32 i = 0; i + 1 < km.J; i++) P1 *= km.sigma;
  return P1;
}
template <>
int launch_pack_all<KeyT>(dc3hip_ctx *c, KeyT km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  const u64 P1 = keyt_p1(km);
  if (!store) {
    hipLaunchKernelGGL((k_pack_image_textT<1024, false, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1, (void *)out,
                       ck.chunk, ck.nchunks, table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_textT<NB, false>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, P1, (void *)out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
template <class KM> static u64 pass1_p1(const KM &) { return 0; }
template <> u64 pass1_p1<KeyT>(const KeyT &km) { return keyt_p1(km); }
// The records of all m (+dummy) positions are in key order behind accessor `acc` (pos, neq): all keys distinct -> the
// order is the suffix array (*state = 1; out_sa / out_rank written); else, with spos/snf given, the samples are
// filtered out with their full names (*state = 2); else *state stays 0.
template <class Acc, class Map>
static int finish_position_order(dc3hip_ctx *c, Acc acc, Map mp, u32 nrec, u32 m, u32 dummy, u32 *out_sa, u32 *out_rank,
                                 u32 *spos, u32 *snf, int *state, bool known_distinct = false) {
  // known_distinct: the tie pass already established that no two keys are equal (no counting pass, no host round trip)
  const Chunking ck = make_chunks(c, nrec, kBlock);
  u32 *counts = nullptr, *scounts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &scounts));
  if (known_distinct) {
    c->h_words[0] = nrec;
  } else {
    PhaseScope ps(c, DC3HIP_PH_NAMING, nrec);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, nrec, ck.chunk,
                       counts, c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  if (c->h_words[0] == nrec) {          // every key distinct: the sorted order is the suffix array
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    if (out_rank && m > (1u << kInvWindowBits)) {
      // the pairs (pos_k, k + 1) are made by the first partition pass itself, which also leaves out_sa[k] = pos_k
      PairsOfOrder<Acc> po; po.acc = acc; po.skip = dummy; po.out_sa = out_sa;
      RC((inverse_permute_from<PairsOfOrder<Acc>>(c, po, false, pa, pb, m, out_rank, DC3HIP_PH_RANKS)));
    } else {
      {
        PhaseScope ps(c, DC3HIP_PH_RANKS, m);
        hipLaunchKernelGGL((k_emit_sorted<Acc>), dim3(grid_for(c, m)), dim3(kBlock), 0, c->stream, acc, m, dummy,
                           out_sa, pa);
        KCHECK();
      }
      if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
    }
    *state = 1;
  } else if (spos && snf) {             // keep the sort: filter the samples with their full names
    PhaseScope ps(c, DC3HIP_PH_NAMING, nrec);
    hipLaunchKernelGGL((k_filter_count<Acc, Map>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, mp, nrec,
                       ck.chunk, scounts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, scounts, ck.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL((k_filter_write<Acc, Map>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, mp, nrec,
                       ck.chunk, counts, scounts, spos, snf);
    KCHECK();
    *state = 2;
  }
  return E_OK;
}

// Few windows repeat (dc3_doubling.hip.hpp): refine the tied positions alone by prefix doubling and finish the suffix
// array at level 0.  acc = the n positions in window order with their "differs from predecessor" flags; W = symbols per
// window.  *done = false (nothing lost: out_sa is scratch until a caller declares it the result) when too many
// positions are tied, the arena is short, or the rounds do not converge.
static constexpr u32 kDoublingMaxTied = 4u << 20;          // records; and at most 1/64 of the positions
template <class KM, class Acc>
static int doubling_finish(dc3hip_ctx *c, KM km, Acc acc, u32 n, u32 W, u32 *out_sa, bool *done, bool order_in_place) {
  *done = false;
  if (c->no_doubling || !out_sa || n < 2) return E_OK;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, n, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, n);
    hipLaunchKernelGGL((k_dbl_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, n, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 t = c->h_words[2];
  if (t == 0 || t > kDoublingMaxTied || t > n / 64 || c->arena_bytes - c->arena_off < (size_t)t * 128 + (32u << 20)) {
    arena_release(c, mk);
    return E_OK;
  }
  u32 *slot = nullptr, *pos = nullptr, *start = nullptr, *gid = nullptr, *mapidx = nullptr, *map_pos = nullptr, *map_val = nullptr;
  Rec8 *pa = nullptr, *pb = nullptr, *ps_sorted = nullptr;
  Rec16 *act = nullptr, *tmp = nullptr, *next = nullptr;
  RC(arena_alloc(c, (size_t)t + 16, &slot)); RC(arena_alloc(c, (size_t)t + 16, &pos)); RC(arena_alloc(c, (size_t)t + 16, &start));
  RC(arena_alloc(c, (size_t)t + 16, &gid)); RC(arena_alloc(c, (size_t)t + 16, &mapidx));
  RC(arena_alloc(c, (size_t)t + 16, &map_pos)); RC(arena_alloc(c, (size_t)t + 16, &map_val));
  RC(arena_alloc(c, (size_t)t + 16, &pa)); RC(arena_alloc(c, (size_t)t + 16, &pb));
  RC(arena_alloc(c, (size_t)t + 16, &act)); RC(arena_alloc(c, (size_t)t + 16, &tmp)); RC(arena_alloc(c, (size_t)t + 16, &next));
  u32 *sums = nullptr, *carry = nullptr;                      // per-tile summaries of the regrouping
  RC(arena_alloc(c, (size_t)3 * (t / kDblTile + 2), &sums)); RC(arena_alloc(c, (size_t)3 * (t / kDblTile + 2), &carry));
  const u32 kb = bits_of((u64)n);                            // ranks + 1 and slots are below 2^kb
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, t);
    // the order as it stands (final for every untied position) — unless acc already reads it from out_sa
    if (!order_in_place) {
      hipLaunchKernelGGL((k_emit_sorted<Acc>), dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, acc, n, 0u, out_sa, (Rec8 *)nullptr);
      KCHECK();
    }
    hipLaunchKernelGGL((k_dbl_collect<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, n, ck.chunk, (const u32 *)counts, slot,
                       pos, start);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_gid, dim3(1), dim3(kBlock), 0, c->stream, (const u32 *)slot, (const u32 *)start, t, gid);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_map_pairs, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const u32 *)pos, t, pa);
    KCHECK();
  }
  RC(radix_sort<Rec8>(c, pa, pb, t, 32, 32 + bits_of((u64)n - 1), &ps_sorted, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, t);
    hipLaunchKernelGGL(k_dbl_map_build, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const Rec8 *)ps_sorted, t, (const u32 *)gid,
                       map_pos, map_val, mapidx);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_init, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const u32 *)pos, (const u32 *)gid,
                       (const u32 *)mapidx, t, act);
    KCHECK();
  }
  u32 a = t;
  u64 d = W;
  int rounds = 0;
  Rec16 *X = act, *Y = tmp, *Z = next;                       // X: this round's records, Y: sort scratch, Z: next round's records
  for (; a > 0 && rounds < 40 && d < (u64)n * 2; rounds++, d *= 2) {
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, a);
      hipLaunchKernelGGL((k_dbl_key<KM, Acc>), dim3(grid_for(c, a)), dim3(kBlock), 0, c->stream, km, acc, n,
                         (u32)std::min<u64>(d, 0xffffffffull), (const u32 *)map_pos, (const u32 *)map_val, t, X, a);
      KCHECK();
    }
    // by (group, rank of p + d): LSD, the rank first
    Rec16 *s1 = nullptr, *s2 = nullptr;
    RC(radix_sort<Rec16>(c, X, Y, a, 0, kb, &s1, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
    RC(radix_sort<Rec16>(c, s1, s1 == X ? Y : X, a, 32, 32 + kb, &s2, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, a);
      const u32 ntiles = (a + kDblTile - 1) / kDblTile;
      hipLaunchKernelGGL((k_dbl_regroup<false>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const Rec16 *)s2, a, sums, (const u32 *)nullptr,
                         (u32 *)nullptr, (u32 *)nullptr, (Rec16 *)nullptr);
      KCHECK();
      hipLaunchKernelGGL(k_dbl_regroup_scan, dim3(1), dim3(kBlock), 0, c->stream, (const u32 *)sums, ntiles, carry, c->d_words + 2);
      KCHECK();
      hipLaunchKernelGGL((k_dbl_regroup<true>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const Rec16 *)s2, a, (u32 *)nullptr,
                         (const u32 *)carry, out_sa, map_val, Z);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    a = c->h_words[2];
    Rec16 *nx = Z; Z = Y; Y = X; X = nx;
  }
  c->stats.level_tied[0] = t;
  c->stats.level_kept[0] = rounds;                           // (rounds of prefix doubling over the tied positions)
  arena_release(c, mk);
  *done = a == 0;
  return E_OK;
}

template <class KM, class Map>
static int order_all_positions(dc3hip_ctx *c, KM km, Map mp, u32 m, u32 kbits, const HiMap &hm, u32 dummy, u32 *out_sa,
                               u32 *out_rank, u32 *spos, u32 *snf, int *state, int depth) {
  *state = 0;
  const ArenaMark mk = arena_mark(c);
  const u32 nrec = m + dummy;              // dummy = 1: include the dummy sample at position m (lib.rs:61-64)
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)nrec, &ha));
  RC(arena_alloc(c, (size_t)nrec, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  u32 *first_table = nullptr;
  const MsdGeom mg = msd_geometry(c, nrec, hm);
  // bucket ordering: the pack kernel only counts, partition pass 1 makes the records on the fly (8 bytes per position
  // neither written nor read back)
  // — opt-in (DC3HIP_PACK_FUSE=1), Key9 only.  Measured at 1 GiB: bytes: pack 3.3 -> 1.7 ms counting only, pass 1
  // 3.7 -> 4.4 ms (it becomes VALU-bound: 9 bytes moved per word instead of 16, but the key arithmetic on top of the
  // ranking), build 20.4 -> 19.5 ms; DNA (KeyT): the rolling image inside the partition pass costs more than the bytes
  // save, 22.8 -> 27.9 ms.  A 5 % gain on one input class against a second variant of the dominant kernel: off by default.
  // (byte windows at level 0, name triples at the levels below; the small-alphabet windows KeyT keep a pack kernel
  //  that writes: their rolling image inside the partition pass was measured slower, 22.8 -> 27.9 ms at 1 GiB DNA)
  constexpr bool kFusable = std::is_same<KM, Key9>::value || std::is_same<KM, Key3<SymU32>>::value;
  const bool fuse = mg.on && c->pack_fuse && kFusable;
  MsdPass1Keys<KM> p1; p1.km = km; p1.hm = hm; p1.P1 = pass1_p1<KM>(km);
  MsdGeom mgx = mg;
  if constexpr (kFusable) {
    // ... and since the words are made inside pass 1, they can come from an image d1 bits wider than a word has room
    // for (k_msd_part_keys<.., true>): the tie pass then finds next to nothing tied
    if (fuse && !c->no_pack_strip && hm.pbits >= 23 && !hm.exact && kbits >= hm.nbits + mg.d1) {
      u64 limb = 0;                                  // base of the key's three limbs (make_himap's B)
      if constexpr (std::is_same<KM, Key9>::value) limb = km.B3; else limb = km.B;
      p1.strip = true; p1.hm_plain = hm;
      p1.hm = make_himap(limb, kbits, m, hm.pbits - mg.d1);
      mgx.ebits = p1.hm.nbits;                       // (= hm.nbits + d1: the shifts of passes 2 and 3 follow from it)
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
    RC(launch_pack_all<KM>(c, km, nrec, p1.strip ? p1.hm : hm, ha, &first_table, &mgx, !fuse));
  }
  bool sorted_ok = false, distinct = false, all_distinct = false;
  RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &sorted_ok, depth, out_rank ? nullptr : out_sa, dummy,
                           &distinct, first_table, std::is_same<Map, MapText>::value && dummy == 0 && !out_rank, &mgx, 0, 0,
                           fuse ? &p1 : nullptr, &all_distinct)));
  if (sorted_ok && distinct) {
    *state = 1;                            // the tie pass already wrote the suffix array
  } else if (sorted_ok) {
    AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
    bool finished = false;
    if constexpr (std::is_same<Map, MapText>::value) {       // whole text: few repeated windows are settled right here
      if (dummy == 0 && out_sa && !out_rank) RC((doubling_finish<KM, AccHyb>(c, km, acc, nrec, km.window_syms(), out_sa, &finished)));
    }
    if (finished) { *state = 1; c->stats.level_sorted[0] = 6; }
    else RC((finish_position_order<AccHyb, Map>(c, acc, mp, nrec, m, dummy, out_sa, out_rank, spos, snf, state, all_distinct)));
  }
  arena_release(c, mk);
  return E_OK;
}

template <int NT, int VT, class TA, class TB>
static int launch_merge(dc3hip_ctx *c, u32 ntiles, const TA *A, u32 nA, const TB *B, u32 nB, const u32 *part,
                        u32 *out_sa, Rec8 *out_pairs, u32 rank_base = 0) {
  auto kern = k_merge<NT, VT, TA, TB>;
  const size_t smem = MergeSmem<NT, VT>::kBytes;
  HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), smem, c->stream, A, nA, B, nB, part, out_sa, out_pairs, rank_base);
  return E_OK;
}

// DC3HIP_TRACE=1 (stage-level parity, the counterpart of the reference's crosscheck! macro,
// crates/divsufsort/src/crosscheck.rs:17-84): order-sensitive checksums of a level's three canonical arrays — the
// sorted samples SA12 (as text positions of the level, the dummy included), the sorted mod-0 suffixes SA0 and the
// level's suffix array — which do not depend on HOW names were made (dense by sorting or packed directly), so the
// CPU restatement the tests check against emits the same words and can be compared level by level.
enum { TR_SA12 = 0, TR_SA0 = 1, TR_SA = 2 };
static int trace_sum(dc3hip_ctx *c, int which, int depth, const void *arr, u32 n, int kind /*0 u32 positions, 1 slots, 2 Tup0, 3 Tup0C*/, u32 m0) {
  if (!c->trace || depth >= DC3HIP_MAX_LEVELS || n == 0) return E_OK;
  u64 *acc = c->d_trace + (size_t)which * DC3HIP_MAX_LEVELS + depth;
  hipLaunchKernelGGL(k_trace_sum, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, arr, n, kind, m0, acc);
  KCHECK();
  return E_OK;
}

// Sample tuples of slots sa12l[0..cnt) in that order -> t12 (lib.rs:136-162's reads, gathered once): the slot table is
// built by streaming (8-byte entries when the level's symbols fit 16 bits, else 16-byte) and gathered.  table0
// ([256][chunks of cnt]) receives the digit table of the fused mod-0 selection pass.  The slot table lives above the
// caller's arena mark and is released here.
// Sample tuples in SA12 order by scattering instead of gathering (dc3_merge.hip.hpp, "WITHOUT the random gather"), as
// COMPACT tuples (TupC, 12 bytes).  Level 0 (bytes) moves 8-byte records through the two partition passes and reads the
// first symbol off a table of cumulative counts; deeper levels whose symbols fit 16 bits move 12-byte records.
// *done = false when the level is too small or the arena too short for the two record arrays (the caller gathers).
template <class Sym, class Out, bool kDerive>
static int scatter_tuples_run(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, const u32 *rank12, const u32 *sa12, const Chunking &ckc,
                              TupC *t12, u32 *table0, bool *done) {
  typedef typename Out::Rec Rec;
  *done = false;
  // (level 0, 8-byte words: pass 1 in tiles of 6144 slots on 1024 threads, k_tup8_part1, unless DC3HIP_TUP_BIGTILE=0)
  const bool big = kDerive && c->tup_bigtile;
  const u32 tile1 = big ? (u32)kTup8Tile : (u32)kTupTile;
  const u32 ntiles = (m02 + tile1 - 1) / tile1;
  const u32 tpc = std::max<u32>(1, (ntiles + 2047) / 2048);
  const u32 cpg = ((ntiles + 7) / 8 + tpc - 1) / tpc, cpx = cpg * tpc;
  const u32 chunk = tpc * tile1, nchunks = (m02 + chunk - 1) / chunk;
  const u32 nb = ((m02 - 1) >> kTupSh1) + 1;                     // buckets of 2^22 destinations (<= 1024)
  if (c->arena_bytes - c->arena_off < (size_t)m02 * 2 * sizeof(Rec) + (size_t)1024 * nchunks * 4 + ((size_t)nb << 11) + (16u << 20)) return E_OK;
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tup_part1<Sym, Out>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTupPartSmem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tup_part2<Out>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTupPartSmem));
    HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tup_local<Out, kDerive>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTupWin * 4)));
    attr_set[c->device & 15] = true;
  }
  const ArenaMark mk = arena_mark(c);
  Rec *ra = nullptr, *rb = nullptr;
  u32 *table1 = nullptr, *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *cur2 = nullptr;
  u32 *cum = nullptr;
  const u32 nsym = 258;                                          // level 0: codes 0..sigma <= 256 (+ slack)
  RC(arena_alloc(c, (size_t)m02, &ra)); RC(arena_alloc(c, (size_t)m02, &rb));
  RC(arena_alloc(c, (size_t)1024 * nchunks, &table1));
  RC(arena_alloc(c, (size_t)nb * 8 + 16, &cntg)); RC(arena_alloc(c, (size_t)nb * 8 + 16, &startg)); RC(arena_alloc(c, (size_t)nb * 8 + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb + 16, &bstart)); RC(arena_alloc(c, (size_t)nb + 16, &tpre)); RC(arena_alloc(c, (size_t)nb + 16, &tpreh));
  RC(arena_alloc(c, (size_t)16, &plan)); RC(arena_alloc(c, (size_t)nb * 512, &cur2));
  RC(arena_alloc(c, (size_t)nsym + 16, &cum));
  const u32 rbits = bits_of(m02);                                // r <= m02
  Out oa, ob;
  oa.p = ra; ob.p = rb;
  if constexpr (kDerive) { oa.rb = rbits; ob.rb = rbits; }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
    hipLaunchKernelGGL(k_tup_hist1, dim3(nchunks), dim3(kBlock), 0, c->stream, rank12, m02, chunk, nchunks, table1);
    KCHECK();
    hipLaunchKernelGGL(k_msd_cnt1, dim3(nb), dim3(kBlock), 0, c->stream, (const u32 *)table1, nchunks, cpg, cntg);
    KCHECK();
    hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb, m02, startg, cur1, bstart, tpre, tpreh, plan);
    KCHECK();
    HIPC(hipMemsetAsync(cur2, 0, (size_t)nb * 512 * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(table0, 0, (size_t)256 * ckc.nchunks * sizeof(u32), c->stream));
    if (kDerive) {
      HIPC(hipMemsetAsync(cum, 0, (size_t)(nsym + 1) * sizeof(u32), c->stream));
      hipLaunchKernelGGL((k_sample_sym_hist<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, nsym, cum);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cum, nsym + 1, (u32 *)nullptr);
      KCHECK();
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02, 3);
    if constexpr (kDerive) {
      if (big) {
        static std::atomic<bool> attr8[16];
        if (!attr8[c->device & 15]) {
          HIPC(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tup8_part1<Sym>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTup8PartSmem));
          attr8[c->device & 15] = true;
        }
        hipLaunchKernelGGL((k_tup8_part1<Sym>), dim3(8 * cpx), dim3(kTup8NT), kTup8PartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa);
      } else {
        hipLaunchKernelGGL((k_tup_part1<Sym, Out>), dim3(8 * cpx), dim3(kTupNT), kTupPartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa);
      }
    } else {
      hipLaunchKernelGGL((k_tup_part1<Sym, Out>), dim3(8 * cpx), dim3(kTupNT), kTupPartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa);
    }
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02, 3);
    const u32 tpb = (1u << kTupSh1) / kTupTile;
    hipLaunchKernelGGL((k_tup_part2<Out>), dim3(8 * ((nb + 7) / 8) * tpb), dim3(kTupNT), kTupPartSmem, c->stream, (const Rec *)ra, m02, nb, cur2, ob, rbits);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
    hipLaunchKernelGGL((k_tup_local<Out, kDerive>), dim3((m02 + kTupWin - 1) / kTupWin), dim3(1024), 2 * kTupWin * 4, c->stream, (const Rec *)rb, rbits, sa12, m02, m0,
                       ckc.chunk, ckc.nchunks, (const u32 *)cum, nsym, t12, table0);
    KCHECK();
  }
  arena_release(c, mk);
  *done = true;
  return E_OK;
}
template <class Sym>
static int scatter_tuples(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, const u32 *rank12, const u32 *sa12, const Chunking &ckc,
                          TupC *t12, u32 *table0, bool *done) {
  *done = false;
  if (c->no_tup_scatter || m02 < c->tup_scatter_min || m02 < 2 || ckc.chunk < kTupWin) return E_OK;
  if constexpr (std::is_same<Sym, SymU8>::value) {
    if (!c->no_tup_rec8) return scatter_tuples_run<Sym, TupOut8, true>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done);
  }
  return scatter_tuples_run<Sym, TupOut12, false>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done);
}

template <class Sym>
static int build_gather_tuples(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u64 K, const u32 *rank12, const u32 *sa12l,
                               u32 cnt, const Chunking &ckc, Tup12 *t12, u32 *table0) {
  const ArenaMark mk = arena_mark(c);
  PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
  if (K < 65536 && !c->no_tup8) {
    TupS8 *ts = nullptr;
    RC(arena_alloc(c, (size_t)m02, &ts));
    hipLaunchKernelGGL((k_build_tuples8<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, rank12, ts);
    KCHECK();
    if (cnt) {
      PhaseScope pg(c, DC3HIP_PH_OTHER, cnt, 4);   // timed separately as kernel class 4 (gather)
      hipLaunchKernelGGL(k_gather_tuples8, dim3(ckc.nchunks), dim3(kBlock), 0, c->stream, ts, sa12l, cnt, m0, ckc.chunk,
                         ckc.nchunks, t12, table0);
      KCHECK();
    }
  } else {
    Tup12 *ts = nullptr;
    RC(arena_alloc(c, (size_t)m02, &ts));
    hipLaunchKernelGGL((k_build_tuples<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, rank12, ts);
    KCHECK();
    if (cnt) {
      PhaseScope pg(c, DC3HIP_PH_OTHER, cnt, 4);
      hipLaunchKernelGGL(k_gather_tuples, dim3(ckc.nchunks), dim3(kBlock), 0, c->stream, ts, sa12l, cnt, ckc.chunk, ckc.nchunks,
                         t12, table0);
      KCHECK();
    }
  }
  arena_release(c, mk);
  return E_OK;
}

template <int kMergeNT, int kMergeVT, class TA, class TB>
static int merge_lists_shape(dc3hip_ctx *c, const TA *A, u32 nA, const TB *B, u32 nB, u32 *out_sa, Rec8 *out_pairs,
                             u32 rank_base) {
  const u32 total = nA + nB;
  if (total == 0) return E_OK;
  const u32 tile = (u32)kMergeNT * kMergeVT;
  const u32 ntiles = (total + tile - 1) / tile;
  const ArenaMark mk = arena_mark(c);
  u32 *part = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &part));
  {
    PhaseScope ps(c, DC3HIP_PH_MERGE, total);
    // coarse split of every 16th tile boundary first, then the bounded per-tile searches
    constexpr u32 kRatio = 16;
    const u32 nco = (ntiles + kRatio - 1) / kRatio;              // coarse tiles of kRatio*tile outputs
    u32 *coarse = nullptr;
    RC(arena_alloc(c, (size_t)nco + 16, &coarse));
    hipLaunchKernelGGL((k_merge_partition<TA, TB>), dim3((nco + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, A, nA, B, nB,
                       nco, tile * kRatio, (const u32 *)nullptr, 1u, coarse);
    KCHECK();
    hipLaunchKernelGGL((k_merge_partition<TA, TB>), dim3((ntiles + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, A, nA, B, nB,
                       ntiles, tile, (const u32 *)coarse, kRatio, part);
    KCHECK();
    RC((launch_merge<kMergeNT, kMergeVT, TA, TB>(c, ntiles, A, nA, B, nB, part, out_sa, out_pairs, rank_base)));
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}
// Step 3 (lib.rs:131-192): merge-path merge of the sorted sample tuples A and the sorted mod-0 tuples B into
// out_sa[0 .. nA+nB) (and, when out_pairs != nullptr, the (pos, rank_base + k + 1) pairs of the rank inversion).
template <class TA, class TB>
static int merge_lists(dc3hip_ctx *c, const TA *A, u32 nA, const TB *B, u32 nB, u32 *out_sa, Rec8 *out_pairs,
                       u32 rank_base) {
  // 1024 threads x 2 outputs: re-measured in round 4 on the compact tuples against 512 x 4, 1024 x 4, 256 x 8, 512 x 8
  // (merge of 1.07 G suffixes: 6.2 / 7.0 / 8.0 / 10.2 / 10.7 ms, profiles/r04g_lab_shapes.jsonl)
  return merge_lists_shape<1024, 2, TA, TB>(c, A, nA, B, nB, out_sa, out_pairs, rank_base);
}

// Steps 2 + 3 of a level (lib.rs:118-192) on compact tuples: sample tuples scattered into SA12 order (TupC), mod-0
// tuples (Tup0C) selected and ordered by the fused radix pass(es), merge.  *done = false: nothing happened, the caller
// runs the general form.
template <class Sym>
static int unwind_compact(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m1, u32 m02, u64 K, const u32 *rank12, const u32 *sa12,
                          u32 *out_sa, u32 *out_rank, int depth, bool *done) {
  *done = false;
  const ArenaMark mk = arena_mark(c);
  TupC *t12 = nullptr;
  RC(arena_alloc(c, (size_t)m02, &t12));
  constexpr u32 kTup0Tile = SortCfg<Tup0C, 256>::NW * 64 * SortCfg<Tup0C, 256>::IPT;
  const Chunking ckc = make_chunks(c, m02, kTup0Tile);
  u32 *table0 = nullptr, *dbase0 = nullptr;
  RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
  RC(arena_alloc(c, (size_t)256, &dbase0));
  RC((scatter_tuples<Sym>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done)));
  if (!*done) { arena_release(c, mk); return E_OK; }
  Tup0C *z0 = nullptr, *z1 = nullptr, *zs = nullptr;
  RC(arena_alloc(c, (size_t)m0, &z0));
  RC(arena_alloc(c, (size_t)m0, &z1));
  {
    // pass 0 of the mod-0 sort reads the sample tuples directly (selection fused in the loader)
    RC(scan_digit_table(c, table0, ckc.nchunks, dbase0, 256, DC3HIP_PH_COMPACT));
    Mod0LoaderC ld; ld.t = t12;
    KeyDig dig; dig.shift = 0; dig.mask = 255;
    // (tiles of 8192 slots; 6144 and 4096 — two blocks per CU — were measured at 5.6 and 6.1 ms against 5.2 for 716 M slots)
    RC((launch_downsweep<Tup0C, 256, Mod0LoaderC>(c, ld, z0, m02, ckc, dig, table0, dbase0, DC3HIP_PH_COMPACT)));
  }
  RC(radix_sort<Tup0C>(c, z0, z1, m0, 8, bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  RC(trace_sum(c, TR_SA0, depth, zs, m0, 3, m0));
  {
    const u32 dskip = m0 - m1;                  // lib.rs:133: skip the dummy, which sorts first
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    RC(merge_lists(c, t12 + dskip, m02 - dskip, zs, m0, out_sa, pa, 0u));
    if (out_sa) RC(trace_sum(c, TR_SA, depth, out_sa, m, 0, m0));
    if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
  }
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// one DC3 level (lib.rs:44-193) on the device.
//   S: symbols in 1..K with zero tail, m >= 2
//   out_sa  : [m]      k-th smallest suffix -> position   (may be null)
//   out_rank: [m+3..]  position -> 1-based rank, caller zeroes the tail (may be null)
// ---------------------------------------------------------------------------------------------
template <class Sym>
static int dc3_level(dc3hip_ctx *c, Sym S, u32 m, u64 K, u32 *out_sa, u32 *out_rank, int depth, const Presort *pre) {
  if (depth >= DC3HIP_MAX_LEVELS) { set_err("recursion deeper than %d levels", DC3HIP_MAX_LEVELS); return E_HIP; }
  if (m == 1) {   // single suffix (only reachable as the child of a 2- or 3-symbol level)
    c->stats.level_n[depth] = 1; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
    hipLaunchKernelGGL(k_base1, dim3(1), dim3(64), 0, c->stream, out_sa, out_rank);
    KCHECK();
    return E_OK;
  }
  const u32 m0 = (m + 2) / 3, m1 = (m + 1) / 3, m2 = m / 3, m02 = m0 + m2;   // lib.rs:45-48
  struct DepthScope { dc3hip_ctx *c; int was; DepthScope(dc3hip_ctx *x, int d) : c(x), was(x->cur_depth) { c->cur_depth = d; } ~DepthScope() { c->cur_depth = was; } } depth_scope(c, depth);
  c->stats.level_n[depth] = m; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
  const ArenaMark mk0 = arena_mark(c);

  u32 *rank12 = nullptr, *sa12 = nullptr, *R = nullptr;
  RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
  RC(arena_alloc(c, (size_t)m02 + 16, &sa12));
  RC(arena_alloc(c, (size_t)m02 + 16, &R));

  const u64 B = K + 1;
  // (level 1 takes its sample order from the whole-text order when there is one, whatever its alphabet)
  const bool direct = (B * B * B) <= 0x7fffffffull && !(pre && depth == 1);
  c->stats.level_sorted[depth] = direct ? 0 : 1;   // 2 = prefix-sort + tie-refine
  if (direct) {
    // names = the K–S triple packed in base B (order-preserving); always recurse (distinctness unknown)
    // (packing more symbols per name is order-isomorphic too but was measured slower)
    const u32 w = 3; const u64 Bw = B * B * B;       // B^w
    c->stats.level_name_width[depth] = (int32_t)w;
    {
      PhaseScope ps(c, DC3HIP_PH_NAME_DIRECT, m02);
      hipLaunchKernelGGL((k_name_direct<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         (u32)B, w, (u32)(Bw / B), R);
      KCHECK();
    }
    SymU32 RS; RS.s = R; RS.m = m02;
    RC(dc3_level<SymU32>(c, RS, m02, Bw, sa12, rank12, depth + 1, pre));
  } else {
    const u32 b = (u32)B;                          // packing base of make_rec (K < 2^31)
    u32 kbits = 0;                                 // bit width of B^3 - 1; > 32 here (else direct path)
    { unsigned __int128 mx = (unsigned __int128)B * B * B - 1; while (mx) { kbits++; mx >>= 1; } }
    u32 *sslot = nullptr;                           // sorted slots, only used by the discarding recursion
    RC(arena_alloc(c, (size_t)m02 + 16, &sslot));
    const ArenaMark mk1 = arena_mark(c);
    u32 names = 0;
    int mode = 0;
    bool done = false;
    if (pre && depth == 1) {
      // the whole-text sort of level 0 found duplicate keys; its order, filtered down to this level's samples,
      // is the sorted sample order: name it and continue as usual
      c->stats.level_sorted[depth] = 2;
      AccFilt acc; acc.spos = pre->spos; acc.snf = pre->snf;
      RC(name_and_rank<AccFilt>(c, acc, m02, m0, sa12, rank12, R, sslot, &names, &mode));
      done = true;
    }
    // ---- prefix-sort + tie-refine ordering when the N-bit key image separates most samples ------
    if (!done && m02 >= kHybridMinSamples && !c->no_hybrid && !c->no_hybrid8) {
      double pred = 1.0;
      RC(predict_tie_fraction<Sym>(c, S, m, m0, m02, b, make_himap(B, kbits, m), &pred));
      c->stats.level_tie_pred[depth] = pred;
      // (the whole-level order holds 17 B per position + the filtered samples; skipped when the arena is short)
      if (pred < kFullSortMaxPredicted && !c->no_fullsort &&
          c->arena_bytes - c->arena_off >= (size_t)(m + 1) * 17 + (size_t)m02 * 8 + (64u << 20)) {
        // high entropy: try to finish the whole level by sorting all of its positions
        u32 *spos = nullptr, *snf = nullptr;
        RC(arena_alloc(c, (size_t)m02 + 16, &spos));
        RC(arena_alloc(c, (size_t)m02 + 16, &snf));
        int state = 0;
        Key3<Sym> km; km.S = S; km.B = b;
        RC((order_all_positions<Key3<Sym>, MapSelf>(c, km, MapSelf{}, m, kbits, make_himap(B, kbits, m),
                                                    (m % 3 == 1) ? 1u : 0u, out_sa, out_rank, spos, snf, &state,
                                                    depth)));
        if (state == 1) {
          c->stats.level_sorted[depth] = 5;
          arena_release(c, mk0);
          return E_OK;
        }
        if (state == 2) {      // sorted sample order is already there: name it and continue as usual
          c->stats.level_sorted[depth] = 2;
          AccFilt acc; acc.spos = spos; acc.snf = snf;
          RC(name_and_rank<AccFilt>(c, acc, m02, m0, sa12, rank12, R, sslot, &names, &mode));
          done = true;
        }
      }
      if (!done && pred < c->hybrid_max_pred) {
        bool ok = false;
        RC(order_hybrid<Sym>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode, &ok, depth));
        done = ok;
        if (!ok) arena_release(c, mk1);
      }
    }
    // (with the splitter ordering the full 96-bit key costs three passes: no prefix + tie rounds then)
    if (!done && kbits > 64 && m02 >= c->hybrid12_min && !c->no_hybrid && !c->no_hybrid12 && !ssort_applies(c, m02, kbits)) {
      // wide keys whose 34-bit image collides everywhere: try the 63-bit prefix on 12-byte records
      bool ok = false;
      RC(order_hybrid12<Sym>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode, &ok, depth));
      done = ok;
      if (!ok) arena_release(c, mk1);
    }
    if (!done) {
      const u32 W = wide_window_syms(c, m02, K);
      if (W > 3) {
        c->stats.level_sorted[depth] = 1;
        c->stats.level_name_width[depth] = (int32_t)W;
        RC((order_wide<Sym>(c, S, m, m0, m02, bits_of(K), W, sa12, rank12, R, sslot, &names, &mode)));
        done = true;
      }
    }
    if (!done) {
      c->stats.level_sorted[depth] = 1;
      if (kbits <= 64 && !c->no_rec12)
        RC((order_straight<Sym, Rec12>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode)));
      else
        RC((order_straight<Sym, Rec16>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode)));
    }
    arena_release(c, mk1);
    c->stats.trace_names[depth] = (int64_t)names;
    if (mode == 1) {
      SymU32 RS; RS.s = R; RS.m = m02;
      RC(dc3_level<SymU32>(c, RS, m02, names, sa12, rank12, depth + 1));   // lib.rs:104
    } else if (mode == 2) {
      c->stats.level_sorted[depth] += 2;                                    // 3 / 4 = straight / prefix-sort + discarding
      RC(discard_recurse(c, R, sslot, m02, names, sa12, rank12, depth));
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_OTHER);
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, rank12, m02, 8u);
    KCHECK();
  }
  RC(trace_sum(c, TR_SA12, depth, sa12, m02, 1, m0));

  // ---- Step 2 + 3: tuples, mod-0 order, merge -------------------------------------------------
  // t12 = sample tuples in SA12 order.  The gather also produces the digit table of the fused
  // "select mod-0 + first radix pass" (Step 2, lib.rs:118-126).
  // Levels whose symbols fit 16 bits and that are large enough for the scatter: compact tuples (12 / 16 bytes).
  if (K < 65536 && !c->no_tup8) {
    bool done = false;
    RC((unwind_compact<Sym>(c, S, m, m0, m1, m02, K, rank12, sa12, out_sa, out_rank, depth, &done)));
    if (done) { arena_release(c, mk0); return E_OK; }
  }
  Tup12 *t12 = nullptr;
  RC(arena_alloc(c, (size_t)m02, &t12));
  constexpr u32 kTup0Tile = SortCfg<Tup0, 256>::NW * 64 * SortCfg<Tup0, 256>::IPT;
  const Chunking ckc = make_chunks(c, m02, kTup0Tile);
  u32 *table0 = nullptr, *dbase0 = nullptr;
  RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
  RC(arena_alloc(c, (size_t)256, &dbase0));
  RC((build_gather_tuples<Sym>(c, S, m, m0, m02, K, rank12, sa12, m02, ckc, t12, table0)));   // slot table released inside
  Tup0 *z0 = nullptr, *z1 = nullptr, *zs = nullptr;
  RC(arena_alloc(c, (size_t)m0, &z0));
  RC(arena_alloc(c, (size_t)m0, &z1));
  {
    // pass 0 of the mod-0 sort reads the sample tuples directly (selection fused in the loader)
    RC(scan_digit_table(c, table0, ckc.nchunks, dbase0, 256, DC3HIP_PH_COMPACT));
    Mod0Loader ld; ld.t = t12;
    KeyDig dig; dig.shift = 0; dig.mask = 255;
    RC((launch_downsweep<Tup0, 256, Mod0Loader>(c, ld, z0, m02, ckc, dig, table0, dbase0, DC3HIP_PH_COMPACT)));
  }
  RC(radix_sort<Tup0>(c, z0, z1, m0, 8, bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  RC(trace_sum(c, TR_SA0, depth, zs, m0, 2, m0));
  {
    const u32 dskip = m0 - m1;                  // lib.rs:133: skip the dummy, which sorts first
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    RC(merge_lists(c, t12 + dskip, m02 - dskip, zs, m0, out_sa, pa, 0u));
    if (out_sa) RC(trace_sum(c, TR_SA, depth, out_sa, m, 0, m0));
    if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
  }
  arena_release(c, mk0);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// build: level 0 = bytes through the dense code table
// ---------------------------------------------------------------------------------------------
// prologue / epilogue shared by the single-device build and the global (multi-rank) build
static int build_begin(dc3hip_ctx *c) {
  c->built = false;
  c->arena_off = 0; c->arena_peak = 0;
  c->ev_used = 0; c->marks.clear();
  memset(&c->stats, 0, sizeof(c->stats));
  c->stats.struct_size = (int32_t)sizeof(dc3hip_stats);
  c->stats.arena_bytes = (int64_t)c->arena_bytes;
  if (c->n < 0) return E_ARGS;
  HIPC(hipSetDevice(c->device));
  for (int l = 0; l < DC3HIP_MAX_LEVELS; l++) c->stats.trace_names[l] = -1;
  if (c->trace) HIPC(hipMemsetAsync(c->d_trace, 0, 3 * DC3HIP_MAX_LEVELS * sizeof(u64), c->stream));
  HIPC(hipMemsetAsync(c->d_xcdmon, 0, 64 * sizeof(u32), c->stream));
  if (c->profile) HIPC(hipEventRecord(c->ev_build_a, c->stream));
  return E_OK;
}
static int build_end(dc3hip_ctx *c) {
  if (c->profile) HIPC(hipEventRecord(c->ev_build_b, c->stream));
  c->stats.trace_on = c->trace ? 1 : 0;
  if (c->trace) {
    static_assert(sizeof(c->stats.trace_sa12[0]) == sizeof(u64), "trace words");
    HIPC(hipMemcpyAsync(c->stats.trace_sa12, c->d_trace, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->stats.trace_sa0, c->d_trace + DC3HIP_MAX_LEVELS, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->stats.trace_sa, c->d_trace + 2 * DC3HIP_MAX_LEVELS, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  u32 mon[64];
  HIPC(hipMemcpyAsync(mon, c->d_xcdmon, sizeof(mon), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  {
    // where the XCD-grouped partition blocks of this build really ran: share of them on their group's majority XCD
    u64 all = 0, hit = 0;
    for (int g = 0; g < 8; g++) { u32 mx = 0; for (int x = 0; x < 8; x++) { all += mon[g * 8 + x]; mx = std::max(mx, mon[g * 8 + x]); } hit += mx; }
    c->stats.xcd_blocks = (int64_t)all;
    c->stats.xcd_group_hit = all ? (double)hit / (double)all : 0.0;
    c->stats.xcd_round_robin = c->xcd_rr;
  }
  c->stats.arena_peak = (int64_t)c->arena_peak;
  c->stats.arena_bytes = (int64_t)c->arena_bytes;
  if (c->profile) {
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, c->ev_build_a, c->ev_build_b));
    c->stats.build_ms = ms;
    double by_level[DC3HIP_MAX_LEVELS][DC3HIP_PH_COUNT] = {};
    for (const PhaseMark &m : c->marks) {
      float t = 0;
      if (hipEventElapsedTime(&t, m.a, m.b) != hipSuccess) continue;
      if (m.kclass != 4 && m.depth >= 0 && m.depth < DC3HIP_MAX_LEVELS) by_level[m.depth][m.phase] += t;
      if (m.kclass != 4) {   // class 4 is nested inside the TUPLES phase mark
        c->stats.phase_ms[m.phase] += t;
        c->stats.phase_launches[m.phase] += 1;
      }
      if (m.kclass == 4) { c->stats.gather_ms += t; c->stats.gather_launches += 1; c->stats.gather_elems += m.elems; continue; }
      if (m.kclass == 3) { c->stats.partition_ms += t; c->stats.partition_launches += 1; c->stats.partition_elems += m.elems; }
      if (m.kclass == 5) { c->stats.msd_part_ms += t; c->stats.msd_part_launches += 1; c->stats.msd_part_elems += m.elems; }
      if (m.kclass == 6) { c->stats.msd_local_ms += t; c->stats.msd_local_launches += 1; c->stats.msd_local_elems += m.elems; }
      if (m.kclass == 9) { c->stats.msd_part_keys_ms += t; c->stats.msd_part_keys_launches += 1; c->stats.msd_part_keys_elems += m.elems; }
      if (m.kclass == 7) { c->stats.ssort_part_ms += t; c->stats.ssort_part_launches += 1; c->stats.ssort_part_elems += m.elems; }
      if (m.kclass == 8) { c->stats.ssort_local_ms += t; c->stats.ssort_local_launches += 1; c->stats.ssort_local_elems += m.elems; }
      if (m.kclass >= 0 && m.kclass < 3) {
        c->stats.downsweep_ms[m.kclass] += t; c->stats.downsweep_launches[m.kclass] += 1;
        c->stats.downsweep_elems[m.kclass] += m.elems;
      }
    }
    if (c->level_report) {      // DC3HIP_LEVEL_PHASES=1: the phase times level by level, on stderr (a tuning aid)
      for (int l = 0; l < c->stats.levels && l < DC3HIP_MAX_LEVELS; l++) {
        double sum = 0;
        for (int p = 0; p < DC3HIP_PH_COUNT; p++) sum += by_level[l][p];
        std::fprintf(stderr, "dc3hip level %d n=%lld K=%lld mode=%d total=%.2f ms:", l, (long long)c->stats.level_n[l], (long long)c->stats.level_K[l],
                     c->stats.level_sorted[l], sum);
        for (int p = 0; p < DC3HIP_PH_COUNT; p++) if (by_level[l][p] > 0.005) std::fprintf(stderr, " p%d=%.2f", p, by_level[l][p]);
        std::fprintf(stderr, "\n");
      }
    }
  }
  c->built = true;
  c->sa_trusted = true;
  c->parts_trusted = 0;
  return E_OK;
}
// level-0 alphabet: dense order-preserving codes 1..sigma of the bytes that occur
static int build_alphabet(dc3hip_ctx *c, u32 *sigma_out) {
  const int64_t n = c->n;
  {
    PhaseScope ps(c, DC3HIP_PH_ALPHABET, n);
    HIPC(hipMemsetAsync(c->d_present, 0, 256 * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_byte_presence, dim3(grid_for(c, (u64)n / 16 + 1)), dim3(kBlock), 0, c->stream, c->d_text,
                       (u32)n, c->d_present);
    KCHECK();
    hipLaunchKernelGGL(k_make_codes, dim3(1), dim3(kBlock), 0, c->stream, c->d_present, c->d_code, c->d_words + 1);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 1, c->d_words + 1, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 sigma = c->h_words[1];
  if (sigma < 1 || sigma > 256) { set_err("internal: alphabet size %u", sigma); return E_HIP; }
  *sigma_out = sigma;
  return E_OK;
}

// KeyT and the map of its image (see the struct): J = fewest symbols whose base-sigma value exceeds the image width by
// two bits, within 63 bits, the key's 3L symbols and kKeyTMaxImageSyms.  false = no such J (the caller skips the path).
static bool make_keyt(SymU8 S, u32 sigma, u32 L, u64 BL, u32 n, KeyT *km, HiMap *hm, u32 image_bits = 0) {
  if (sigma < 2) return false;
  hm->pbits = image_bits ? 64 - image_bits : bits_of((u64)n - 1);        // positions 0..n-1 only
  hm->nbits = 64 - hm->pbits;
  hm->shx = 0; hm->exact = 0;
  u32 J = 1; u64 SJ = sigma;                                             // sigma^J
  const u32 jmax = std::min<u32>(3 * L, kKeyTMaxImageSyms);
  while (J < jmax && (SJ >> std::min<u32>(hm->nbits + 2, 62)) == 0 && SJ * sigma < (1ull << 63)) { SJ *= sigma; J++; }
  if ((SJ >> hm->nbits) == 0) return false;                              // the image must be a proper scaling
  hm->mfix = (u64)(((((unsigned __int128)1) << (64 + hm->nbits)) - 1) / SJ);
  km->S = S; km->B = sigma + 1; km->BL = (u32)BL; km->L = L; km->sigma = sigma; km->J = J;
  return true;
}

// Whole-text shortcut with key maker KM (three limbs of base BL: Key9's 9 symbols or KeyT's 3L): predicted ties
// permitting, order all n positions by their windows.  All windows distinct: that order is the suffix array
// (*whole_text).  Otherwise the order, filtered down to level 1's samples with the dense ranks of the windows as
// their names, still serves level 1 (*pre): a name built from a window LONGER than the K-S triple orders the samples
// consistently and equal names still imply equal triples, which is all lib.rs:78-104 needs of a name.
template <class KM>
static int try_text_order(dc3hip_ctx *c, KM km, u64 BL, const HiMap &hm, u32 sigma, bool *whole_text, Presort *pre) {
  const int64_t n = c->n;
  u32 kbits = 0;                          // of the full key (limb base BL)
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  double pred = 1.0;
  RC(predict_tie_fraction_pos<KM>(c, km, (u32)n, hm, &pred));
  c->stats.level_tie_pred[0] = pred;
  if (!text_order_worth_trying(pred, (u64)n, hm.nbits)) return E_OK;
  const u32 m0 = (u32)((n + 2) / 3), m1 = m0 + (u32)(n / 3);            // level 1 = string of m1 names
  const u32 m02_1 = (m1 + 2) / 3 + m1 / 3;                              // its samples (incl. the dummy)
  // The filtered order (2 * m02_1 words < n) lives in the output buffer: the optimistic SA written there
  // by the tie pass is void when keys repeat, and nothing else writes d_sa before the final merge.
  u32 *spos = c->d_sa, *snf = c->d_sa + m02_1 + 16;
  MapText mp; mp.m0 = m0; mp.npre = 0; mp.ppos[0] = mp.ppos[1] = 0;
  if (m1 % 3 == 1) mp.ppos[mp.npre++] = m1;                              // level 1's dummy sample
  if (n % 3 == 1 && (m0 - 1) % 3 != 0) mp.ppos[mp.npre++] = m0 - 1;      // level 0's dummy, a level-1 position
  int state = 0;
  RC((order_all_positions<KM, MapText>(c, km, mp, (u32)n, kbits, hm, 0u, c->d_sa, nullptr, spos, snf, &state, 0)));
  c->stats.text_sort_state = state == 1 ? 1 : state == 2 ? 2 : 3;
  if (state == 1) {
    *whole_text = true;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1;
    if (c->stats.level_sorted[0] != 6) c->stats.level_sorted[0] = 5;      // 6 = finished by prefix doubling of the tied positions
  } else if (state == 2) {
    pre->spos = spos; pre->snf = snf;      // duplicates: the order still serves level 1
  }
  return E_OK;
}

template <class KM>
static int launch_pack12_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec12 *out, int nb, const Chunking &ck, u32 *table) {
  if (nb == 512)
    hipLaunchKernelGGL((k_pack_image12_all_hist<KM, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out,
                       ck.chunk, ck.nchunks, table);
  else
    hipLaunchKernelGGL((k_pack_image12_all_hist<KM, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out,
                       ck.chunk, ck.nchunks, table);
  KCHECK();
  return E_OK;
}
template <>
int launch_pack12_all<KeyT>(dc3hip_ctx *c, KeyT km, u32 nrec, const HiMap &hm, Rec12 *out, int nb, const Chunking &ck, u32 *table) {
  u64 P1 = 1;
  for (u32 i = 0; i + 1 < km.J; i++) P1 *= km.sigma;
  if (nb == 512)
    hipLaunchKernelGGL((k_pack_image_textT<512, true>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1,
                       (void *)out, ck.chunk, ck.nchunks, table, 0u);
  else
    hipLaunchKernelGGL((k_pack_image_textT<256, true>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1,
                       (void *)out, ck.chunk, ck.nchunks, table, 0u);
  KCHECK();
  return E_OK;
}
// The same shortcut on 12-byte records (image of ibits <= 63 bits NEXT TO the position instead of sharing a 64-bit
// word with it): beyond 2^31 positions the 8-byte record has 32 image bits left and ties 39 % of even random
// positions; here the image is as wide as the text needs (log2 n + 4.2 bits rounded up to whole 9-bit digits: 36 bits =
// 4 passes of 24 B per record up to 3.5 GiB, 3-5 % ties; 45 bits above).  hm = map of KM's image to ibits.
template <class KM>
static int try_text_order12(dc3hip_ctx *c, KM km, u64 BL, const HiMap &hm, u32 sigma, bool *whole_text, Presort *pre) {
  const int64_t n = c->n;
  u32 kbits = 0;
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  const size_t need = (size_t)n * 26 + ((size_t)256 << 20);
  if (c->arena_bytes - c->arena_off < need) {
    if (c->arena_fixed || c->arena_off != 0) return E_OK;
    if (ensure_arena(c, need) != E_OK) return E_OK;
  }
  const ArenaMark mk = arena_mark(c);
  {
    const u32 stride = std::max<u32>(1, (u32)n >> 20);
    const u32 ns = ((u32)n - 1) / stride + 1;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    u32 ts = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, ns);
      hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
      KCHECK();
    }
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    const double ratio = (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1);
    const double pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
    c->stats.level_tie_pred[0] = pred;
    arena_release(c, mk);
    if (!(pred < kTextSortMaxPredicted)) return E_OK;
  }
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)n, &ha));
  RC(arena_alloc(c, (size_t)n, &hb));
  RC(arena_alloc(c, (size_t)n + 16, &f));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, n);
    int nb = 0; Chunking ck;
    radix_plan<Rec12>(c, (u32)n, hm.nbits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    RC(launch_pack12_all(c, km, (u32)n, hm, ha, nb, ck, first_table));
  }
  RC(radix_sort<Rec12>(c, ha, hb, (u32)n, 0, hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN,
                       first_table));
  bool refined = false, distinct = false, deep_flags = false;
  RC((hybrid12_refine<KM>(c, km, kbits, h, (u32)n, f, &refined, 0, c->d_sa, &distinct, &deep_flags)));
  int state = 0;
  const u32 m0 = (u32)((n + 2) / 3), m1 = m0 + (u32)(n / 3);
  const u32 m02_1 = (m1 + 2) / 3 + m1 / 3;
  u32 *spos = c->d_sa, *snf = c->d_sa + m02_1 + 16;                        // (as in try_text_order)
  bool doubled = false;
  if (refined && !distinct) {
    AccHyb12 acc; acc.h = h; acc.f = f;
    // the rank look-ups of the doubling (binary searches with km.cmp) must compare as deep as the flags were made:
    // after the second tie pass the groups of f[] agree on kDeepSyms symbols, and a search with the window alone
    // would return the lower bound of the whole window-equal run for an untied position behind such a group
    KM kd = km;
    if (deep_flags) kd.deep = kDeepSyms;
    RC((doubling_finish<KM, AccHyb12>(c, kd, acc, (u32)n, deep_flags ? kDeepSyms : km.window_syms(), c->d_sa, &doubled)));
  }
  if (refined && (distinct || doubled)) {
    state = 1;                             // the tie pass (or the doubling rounds) already wrote the suffix array
  } else if (refined) {
    MapText mp; mp.m0 = m0; mp.npre = 0; mp.ppos[0] = mp.ppos[1] = 0;
    if (m1 % 3 == 1) mp.ppos[mp.npre++] = m1;
    if (n % 3 == 1 && (m0 - 1) % 3 != 0) mp.ppos[mp.npre++] = m0 - 1;
    AccHyb12 acc; acc.h = h; acc.f = f;
    RC((finish_position_order<AccHyb12, MapText>(c, acc, mp, (u32)n, (u32)n, 0u, c->d_sa, nullptr, spos, snf, &state)));
  }
  arena_release(c, mk);
  c->stats.text_sort_state = state == 1 ? 1 : state == 2 ? 2 : 3;
  if (state == 1) {
    *whole_text = true;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = doubled ? 6 : 5;
  } else if (state == 2) {
    pre->spos = spos; pre->snf = snf;
  }
  return E_OK;
}

// the device-resident build proper: SA of c->d_text[0..n) into c->d_sa
static int build_core(dc3hip_ctx *c) {
  const int64_t n = c->n;
  if (n == 1) {
    HIPC(hipMemsetAsync(c->d_sa, 0, 4, c->stream));
  } else if (n >= 2) {
    u32 sigma = 0;
    RC(build_alphabet(c, &sigma));
    SymU8 S; S.t = c->d_text; S.code = c->d_code; S.m = (u32)n;
    bool whole_text = false;
    Presort pre{nullptr, nullptr};
    const u64 Bq = (u64)sigma + 1, B3 = Bq * Bq * Bq;
    // (even uniformly random symbols repeat a w-symbol window once sigma^w is not well above n^2/2: skip then)
    const double need_bits = 2.0 * log2((double)n) + 2.0, sym_bits = log2((double)sigma);
    if ((u64)n >= kHybridMinSamples && !c->no_hybrid && !c->no_fullsort && !c->no_text_shortcut &&
        c->arena_bytes - c->arena_off >= (size_t)n * 22 + (64u << 20)) {
      // whole-text shortcut: if all w-symbol windows of a high-entropy text are distinct, sorting all positions by
      // them is the suffix array (the same test level 1 would make on its triples, without building level 1)
      // 12-byte records (image beside the position) once positions take all 32 bits; DC3HIP_TEXT_ORDER12=1/0 forces
      // / forbids them (tests).  Image width: log2 n + 4.2 bits, rounded up to whole 9-bit digits.
      const bool wide = c->text_order12 >= 0 ? c->text_order12 == 1 : bits_of((u64)n - 1) >= 32;
      const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2((double)n) + 4.2) / 9.0));
      if (9.0 * sym_bits >= need_bits && B3 * B3 * B3 > 0x7fffffffull) {
        Key9 km; km.S = S; km.B = (u32)Bq; km.B3 = (u32)B3;
        u32 kbits = 0;
        { unsigned __int128 mx = (unsigned __int128)B3 * B3 * B3 - 1; while (mx) { kbits++; mx >>= 1; } }
        if (wide)
          RC(try_text_order12<Key9>(c, km, B3, make_himap(B3, kbits, (u32)n, 64 - ibits), sigma, &whole_text, &pre));
        else
          RC(try_text_order<Key9>(c, km, B3, make_himap(B3, kbits, (u32)n, bits_of((u64)n - 1)), sigma, &whole_text, &pre));
      } else if (!c->no_long_keys) {
        // small alphabets: limbs of L > 3 symbols (as many as fit 32 bits), 3L-symbol windows
        u32 L = 1; u64 BL = Bq;
        while (L < 20 && BL * Bq <= 0xffffffffull) { BL *= Bq; L++; }
        KeyT km; HiMap hm;
        if (L > 3 && 3.0 * L * sym_bits >= need_bits && make_keyt(S, sigma, L, BL, (u32)n, &km, &hm, wide ? ibits : 0u)) {
          if (wide) RC(try_text_order12<KeyT>(c, km, BL, hm, sigma, &whole_text, &pre));
          else RC(try_text_order<KeyT>(c, km, BL, hm, sigma, &whole_text, &pre));
        }
      }
    }
    if (!whole_text) {
      RC(ensure_arena(c, arena_requirement(n)));          // (the arena is empty here: the filtered order lives in d_sa)
      RC(dc3_level<SymU8>(c, S, (u32)n, sigma, c->d_sa, nullptr, 0, pre.spos ? &pre : nullptr));
    }
  }
  return E_OK;
}

static int ctx_build_once(dc3hip_ctx *c) {
  RC(build_begin(c));
  RC(build_core(c));
  return build_end(c);
}
// arena_requirement() is a model of the paths' peaks, not a proof: if the bump allocator (not hipMalloc) runs out, the
// arena is grown by half and the build — deterministic, nothing was returned yet — is repeated once.
static int ctx_build(dc3hip_ctx *c) {
  c->arena_exhausted = false;
  int rc = ctx_build_once(c);
  if (rc == E_ALLOC && c->arena_exhausted && !c->arena_fixed) {
    (void)hipStreamSynchronize(c->stream);
    c->arena_off = 0;
    if (ensure_arena(c, c->arena_bytes + c->arena_bytes / 2 + ((size_t)64 << 20)) == E_OK) {
      c->arena_exhausted = false;
      rc = ctx_build_once(c);
    }
  }
  return rc;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char *dc3hip_version(void) { return DC3HIP_VERSION_STR; }
const char *dc3hip_last_error(void) { return g_err; }

int32_t dc3hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { set_err("hipGetDeviceCount failed"); return E_HIP; }
  return n;
}

static int ctx_create_impl(dc3hip_ctx **out, int32_t device, int64_t max_n, dc3hip_ctx *arena_from);
int32_t dc3hip_ctx_create(dc3hip_ctx **out, int32_t device, int64_t max_n) { return ctx_create_impl(out, device, max_n, nullptr); }
// arena_from != nullptr: the new context works in that context's arena instead of allocating its own (the lender must not
// build meanwhile; dc3hip_ctx_build_partitions: the partitions are built one after the other in the parent's arena)
static int ctx_create_impl(dc3hip_ctx **out, int32_t device, int64_t max_n, dc3hip_ctx *arena_from) {
  if (!out || max_n < 0) { set_err("dc3hip_ctx_create: invalid arguments"); return E_ARGS; }
  *out = nullptr;
  if (max_n > DC3HIP_MAX_N) { set_err("n=%lld exceeds DC3HIP_MAX_N", (long long)max_n); return E_TOOBIG; }
  int ndev = 0;
  HIPC(hipGetDeviceCount(&ndev));
  if (ndev <= 0) { set_err("no HIP device visible (no CPU fallback exists)"); return E_HIP; }
  if (device < 0) HIPC(hipGetDevice(&device));
  if (device >= ndev) { set_err("device %d out of range (%d devices)", device, ndev); return E_ARGS; }
  dc3hip_ctx *c = new (std::nothrow) dc3hip_ctx();
  if (!c) { set_err("host allocation failed"); return E_ALLOC; }
  c->device = device; c->max_n = max_n;
  memset(&c->stats, 0, sizeof(c->stats));
  const char *prof = getenv("DC3HIP_PROFILE");
  c->profile = !(prof && prof[0] == '0');
  const char *nh = getenv("DC3HIP_NO_HYBRID");
  c->no_hybrid = (nh && nh[0] == '1');
  const char *nst = getenv("DC3HIP_NO_SMALL_TIES");
  c->no_small_ties = (nst && nst[0] == '1');
  { const char *e = getenv("DC3HIP_NO_SPLIT_EMIT"); c->no_split_emit = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_TRACE"); c->trace = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_LONG_KEYS"); c->no_long_keys = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_DOUBLING"); c->no_doubling = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_TEXT_ORDER12"); if (e && (e[0] == '0' || e[0] == '1')) c->text_order12 = e[0] - '0'; }
  { const char *e = getenv("DC3HIP_NO_TUP8"); c->no_tup8 = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_MSD"); c->no_msd = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_LEVEL_PHASES"); c->level_report = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_PACK_FUSE"); c->pack_fuse = !(e && e[0] == '0'); }
  { const char *e = getenv("DC3HIP_NO_PACK_STRIP"); c->no_pack_strip = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_TUP_BIGTILE"); c->tup_bigtile = !(e && e[0] == '0'); }
  { const char *e = getenv("DC3HIP_NO_XCD_MAP"); c->no_xcd_map = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_TUP_SCATTER"); c->no_tup_scatter = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_TUP_REC8"); c->no_tup_rec8 = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_TUP_SCATTER_MIN"); if (e && *e) c->tup_scatter_min = (u32)strtoul(e, nullptr, 10); }
  { const char *e = getenv("DC3HIP_MSD_MIN"); if (e) c->msd_min = (u32)std::max(4096ll, atoll(e)); }
  { const char *e = getenv("DC3HIP_NO_SSORT"); c->no_ssort = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_SSORT_VERIFY"); c->ssort_verify = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_PACK_COUNT"); c->no_pack_count = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_SSORT_REC12"); c->ssort_rec12 = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_WIDE_WINDOW"); c->no_wide_window = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_SSORT_MIN"); if (e) c->ssort_min = (u32)std::max(8192ll, atoll(e)); }
  { const char *e = getenv("DC3HIP_NO_HYBRID12"); c->no_hybrid12 = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_NO_HYBRID8"); c->no_hybrid8 = (e && e[0] == '1'); }
  { const char *e = getenv("DC3HIP_HYBRID12_MIN"); if (e) c->hybrid12_min = (u32)std::max(0ll, atoll(e)); }
  const char *nts = getenv("DC3HIP_NO_TEXT_SHORTCUT");
  c->no_text_shortcut = (nts && nts[0] == '1');
  const char *nf = getenv("DC3HIP_NO_FULLSORT");
  c->no_fullsort = (nf && nf[0] == '1');
  const char *nd = getenv("DC3HIP_NO_DISCARD");
  c->no_discard = (nd && nd[0] == '1');
  const char *n9 = getenv("DC3HIP_NO_9BIT");
  c->no_nine_bit = (n9 && n9[0] == '1');
  const char *n12 = getenv("DC3HIP_NO_REC12");
  c->no_rec12 = (n12 && n12[0] == '1');
  int rc = [&]() -> int {
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPC(hipMalloc(&c->d_text, (size_t)max_n + 64));
    HIPC(hipMalloc(&c->d_sa, ((size_t)max_n + 16) * sizeof(u32)));
    c->arena_bytes = std::min(arena_requirement(max_n), arena_text_requirement(max_n));   // grown on demand (ensure_arena)
    // testing aid: DC3HIP_ARENA_BYTES=<bytes> replaces the computed size (a build then either fits — possibly through
    // a fallback ordering — or fails loudly with -2; it never returns a wrong array)
    if (const char *e = getenv("DC3HIP_ARENA_BYTES")) { const long long v = atoll(e); if (v > 0) { c->arena_bytes = (size_t)v; c->arena_fixed = true; } }
    if (arena_from && arena_from->arena && arena_from->device == device) {
      c->arena = arena_from->arena; c->arena_bytes = arena_from->arena_bytes; c->arena_borrowed = true; c->arena_fixed = true;
    } else {
      HIPC(hipMalloc(&c->arena, c->arena_bytes));
    }
    HIPC(hipMalloc(&c->d_present, 256 * sizeof(u32)));
    HIPC(hipMalloc(&c->d_code, 256 * sizeof(uint16_t)));
    HIPC(hipMalloc(&c->d_words, 64 * sizeof(u32)));
    HIPC(hipMalloc(&c->d_xcdmon, 4096 * sizeof(u32)));         // (64 monitor words; 4096 for the placement probe below)
    HIPC(hipMalloc(&c->d_trace, 3 * DC3HIP_MAX_LEVELS * sizeof(u64)));
    HIPC(hipHostMalloc(&c->h_words, 64 * sizeof(u32), hipHostMallocDefault));
    HIPC(hipEventCreate(&c->ev_build_a));
    HIPC(hipEventCreate(&c->ev_build_b));
    {
      // Placement probe (dc3hip_stats.xcd_round_robin): 4096 one-wave blocks report the XCD they ran on.  The bucket
      // ordering's partition passes are only fast when blocks b and b + 8 share an XCD (XCD-grouped reservation,
      // dc3_msd.hip.hpp); on a device that places blocks otherwise the context keeps to the stable 256-bucket LSD passes,
      // whose speed does not depend on placement (DC3HIP_XCD_ASSUME=1: keep the bucket ordering anyway).
      std::vector<u32> xs(4096);
      hipLaunchKernelGGL(k_xcd_probe, dim3(4096), dim3(64), 0, c->stream, c->d_xcdmon);
      HIPC(hipMemcpyAsync(xs.data(), c->d_xcdmon, 4096 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      u32 cnt[8][8] = {};
      for (u32 b = 0; b < 4096; b++) cnt[b & 7][xs[b] & 7]++;
      u32 hit = 0, seen = 0;
      for (int g = 0; g < 8; g++) {
        u32 mx = 0, arg = 0;
        for (int x = 0; x < 8; x++) if (cnt[g][x] > mx) { mx = cnt[g][x]; arg = (u32)x; }
        hit += mx; seen |= 1u << arg;
      }
      c->xcd_rr = (hit >= 4096 * 9 / 10 && seen == 0xffu) ? 1 : 0;
      const char *e = getenv("DC3HIP_XCD_ASSUME");
      if (!c->xcd_rr && !(e && e[0] == '1')) c->no_msd = true;
    }
    return E_OK;
  }();
  if (rc != E_OK) { dc3hip_ctx_destroy(c); return rc; }
  *out = c;
  return E_OK;
}

void dc3hip_ctx_destroy(dc3hip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_build_a) (void)hipEventDestroy(c->ev_build_a);
  if (c->ev_build_b) (void)hipEventDestroy(c->ev_build_b);
  if (c->d_text) (void)hipFree(c->d_text);
  if (c->d_sa) (void)hipFree(c->d_sa);
  if (c->arena && !c->arena_borrowed) (void)hipFree(c->arena);
  if (c->d_xcdmon) (void)hipFree(c->d_xcdmon);
  if (c->d_present) (void)hipFree(c->d_present);
  if (c->d_code) (void)hipFree(c->d_code);
  if (c->d_words) (void)hipFree(c->d_words);
  if (c->d_trace) (void)hipFree(c->d_trace);
  if (c->h_words) (void)hipHostFree(c->h_words);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static int ctx_check_n(dc3hip_ctx *c, int64_t n) {
  if (!c || n < 0) { set_err("invalid arguments"); return E_ARGS; }
  if (n > c->max_n) { set_err("n=%lld exceeds the context capacity %lld", (long long)n, (long long)c->max_n); return E_ARGS; }
  return E_OK;
}

int32_t dc3hip_ctx_set_text(dc3hip_ctx *c, const uint8_t *T, int64_t n) {
  RC(ctx_check_n(c, n));
  if (!T && n > 0) { set_err("T is NULL"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (n > 0) HIPC(hipMemcpyAsync(c->d_text, T, (size_t)n, hipMemcpyDefault, c->stream));
  HIPC(hipMemsetAsync(c->d_text + n, 0, 64, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->n = n; c->built = false;
  return E_OK;
}

int32_t dc3hip_ctx_generate(dc3hip_ctx *c, int64_t n, uint64_t seed, int32_t kind) {
  return dc3hip_ctx_generate_at(c, n, seed, kind, 0);
}

int32_t dc3hip_ctx_generate_at(dc3hip_ctx *c, int64_t n, uint64_t seed, int32_t kind, int64_t offset) {
  RC(ctx_check_n(c, n));
  if (offset < 0) { set_err("negative offset"); return E_ARGS; }
  if (kind < 0 || kind > 2) { set_err("unknown generator kind %d", kind); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (n > 0) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(c, (u64)n / 8 + 1)), dim3(kBlock), 0, c->stream, c->d_text, (u64)n,
                       (u64)seed, (int)kind, (u64)offset);
    KCHECK();
  }
  HIPC(hipMemsetAsync(c->d_text + n, 0, 64, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->n = n; c->built = false;
  return E_OK;
}

int32_t dc3hip_ctx_build(dc3hip_ctx *c) {
  if (!c) { set_err("ctx is NULL"); return E_ARGS; }
  return ctx_build(c);
}

int32_t dc3hip_ctx_get_sa_i32(dc3hip_ctx *c, int32_t *SA) {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (c->n > (int64_t)INT32_MAX) { set_err("text of %lld bytes needs 64-bit indices", (long long)c->n); return E_TOOBIG; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(SA, c->d_sa, (size_t)c->n * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
}

int32_t dc3hip_ctx_get_sa_i64(dc3hip_ctx *c, int64_t *SA) {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (c->n == 0) return E_OK;
  // widen on the device in arena-sized pieces, then copy
  c->arena_off = 0;
  const size_t piece = std::min<size_t>((size_t)c->n, c->arena_bytes / 8 > 0 ? c->arena_bytes / 8 : 1);
  int64_t *tmp = reinterpret_cast<int64_t *>(c->arena);
  for (size_t off = 0; off < (size_t)c->n; off += piece) {
    const size_t cnt = std::min(piece, (size_t)c->n - off);
    hipLaunchKernelGGL(k_widen, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, c->d_sa + off, tmp, (u32)cnt);
    KCHECK();
    HIPC(hipMemcpyAsync(SA + off, tmp, cnt * 8, hipMemcpyDefault, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  return E_OK;
}

int32_t dc3hip_ctx_get_text(dc3hip_ctx *c, uint8_t *T) {
  if (!c || (!T && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(T, c->d_text, (size_t)c->n, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  return E_OK;
}

// Runs the check; *code receives sufcheck()'s result (0, -2, -3, -4); the return value is the
// library status (E_OK / E_ALLOC / E_HIP), kept apart because the two code spaces overlap.
static int ctx_sufcheck(dc3hip_ctx *c, const u32 *d_sa, int *code, const uint8_t *text = nullptr, int64_t n_override = -1) {
  // utils.c:160-241 as parallel passes; isa lives in the arena.  text / n_override: a partition of the resident text
  const int64_t n = n_override >= 0 ? n_override : c->n;
  if (!text) text = c->d_text;
  *code = 0;
  if (n == 0) return E_OK;
  HIPC(hipSetDevice(c->device));
  c->arena_off = 0;
  u32 *isa = nullptr;
  RC(arena_alloc(c, (size_t)n + 16, &isa));
  int *err = reinterpret_cast<int *>(c->d_words + 8);
  HIPC(hipMemsetAsync(err, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_check_fill, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, d_sa, (u32)n, isa, err);
  KCHECK();
  hipLaunchKernelGGL(k_check_order, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, text, d_sa, isa, (u32)n,
                     err);
  KCHECK();
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

int32_t dc3hip_ctx_sufcheck(dc3hip_ctx *c) {
  if (!c) { set_err("ctx is NULL"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  int code = 0;
  const int rc = ctx_sufcheck(c, c->d_sa, &code);
  if (rc != E_OK) return rc == E_ALLOC ? -5 : -6;   // library failure, distinct from sufcheck's -1..-4
  return code;
}

int32_t dc3hip_ctx_sa_checksum(dc3hip_ctx *c, uint64_t *out) {
  if (!c || !out) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
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
}

int32_t dc3hip_ctx_set_sa_i32(dc3hip_ctx *c, const int32_t *SA) {
  if (!c || (!SA && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
  if (c->n > 0) HIPC(hipMemcpyAsync(c->d_sa, SA, (size_t)c->n * 4, hipMemcpyDefault, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  c->built = true;
  c->sa_trusted = false;
  c->parts_trusted = 0;
  return E_OK;
}

// LCP array of the resident SA (kernels and method: dc3_aux.hip.hpp).  LCP may be a host or a device pointer (n x int32).
int32_t dc3hip_ctx_lcp_i32(dc3hip_ctx *c, int32_t *LCP) {
  if (!c || (!LCP && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  const int64_t n64 = c->n;
  if (n64 == 0) return E_OK;
  if (n64 > (int64_t)INT32_MAX) { set_err("LCP values of %lld bytes do not fit int32", (long long)n64); return E_TOOBIG; }
  const u32 n = (u32)n64;
  HIPC(hipSetDevice(c->device));
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
}

int32_t dc3hip_ctx_bwt(dc3hip_ctx *c, uint8_t *U, int64_t *primary_index) {
  if (!c || !primary_index || (!U && c->n > 0)) { set_err("invalid arguments"); return E_ARGS; }   // utils.c:60
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  const int64_t n = c->n;
  HIPC(hipSetDevice(c->device));
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
}

int32_t dc3hip_ctx_search(dc3hip_ctx *c, const uint8_t *needles, const int64_t *offsets, int32_t count,
                          int64_t *out_start, int64_t *out_len) {
  if (!c || count < 0 || (count > 0 && (!needles || !offsets || !out_start || !out_len))) {
    set_err("invalid arguments"); return E_ARGS;
  }
  if (!c->built) { set_err("no suffix array built in this context"); return E_ARGS; }
  if (c->n == 0) { set_err("empty suffix array (the reference indexes out of bounds here)"); return E_ARGS; }
  if (count == 0) return E_OK;
  HIPC(hipSetDevice(c->device));
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
}

// Partition arrays (sacapart semantics) in the resident SA buffer: chunk c = text[c*S .. min(n,(c+1)*S)), S = n/P + 1
// (crates/sacapart/src/lib.rs:43-46), local indices, back to back — built here chunk by chunk on the device.
int32_t dc3hip_ctx_build_partitions(dc3hip_ctx *c, int32_t num_partitions) {
  if (!c || num_partitions < 1) { set_err("invalid arguments"); return E_ARGS; }
  const int64_t n = c->n;
  if (n > (int64_t)INT32_MAX) { set_err("partitioned build of %lld bytes needs 64-bit indices", (long long)n); return E_TOOBIG; }
  HIPC(hipSetDevice(c->device));
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
}

int32_t dc3hip_ctx_search_partitioned(dc3hip_ctx *c, int32_t num_partitions, const uint8_t *needles, const int64_t *offsets,
                                      int32_t count, int64_t *out_start, int64_t *out_len) {
  if (!c || num_partitions < 1 || count < 0 || (count > 0 && (!needles || !offsets || !out_start || !out_len))) {
    set_err("invalid arguments"); return E_ARGS;
  }
  if (!c->built) { set_err("no partition arrays in this context (dc3hip_ctx_build_partitions / dc3hip_ctx_set_sa_i32)"); return E_ARGS; }
  if (c->n == 0) { set_err("empty text (the reference indexes out of bounds here)"); return E_ARGS; }
  if (c->n > (int64_t)INT32_MAX) { set_err("partitioned search of %lld bytes needs 64-bit indices", (long long)c->n); return E_TOOBIG; }
  if (count == 0) return E_OK;
  HIPC(hipSetDevice(c->device));
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
}

// Test hook for kernel-level parity (the reference's radix_pass, crates/dc3/src/lib.rs:15-39): ONE stable pass of
// the product's radix scatter (up-sweep, scan, down-sweep) over n host words, digit = (word >> shift) & (nb - 1).
int32_t dc3hip_ctx_debug_radix_pass_u64(dc3hip_ctx *c, const uint64_t *words, uint64_t *out, int64_t n, int32_t shift,
                                        int32_t nb) {
  if (!c || !words || !out || n < 1 || shift < 0 || shift > 55 || (nb != 256 && nb != 512)) { set_err("invalid arguments"); return E_ARGS; }
  HIPC(hipSetDevice(c->device));
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
}

int32_t dc3hip_ctx_stats(dc3hip_ctx *c, dc3hip_stats *out) {
  if (!c || !out) { set_err("invalid arguments"); return E_ARGS; }
  *out = c->stats;
  out->struct_size = (int32_t)sizeof(dc3hip_stats);
  return E_OK;
}

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
    g_cache.c = *out; *cached = true;
    return E_OK;
  }
  return dc3hip_ctx_create(out, device, n);
}

static int sufsort_one(const uint8_t *T, void *SA, int64_t n, int bits, int device, bool devptrs) {
  if (!devptrs && tiny_host(T, SA, n, bits)) return E_OK;
  if (n == 0) return E_OK;
  dc3hip_ctx *c = nullptr;
  bool cached = false;
  RC(acquire_ctx(&c, device, n, &cached));
  int rc = [&]() -> int {
    RC(dc3hip_ctx_set_text(c, T, n));          // hipMemcpyDefault handles host or device sources
    RC(ctx_build(c));
    if (bits == 32) RC(dc3hip_ctx_get_sa_i32(c, static_cast<int32_t *>(SA)));
    else RC(dc3hip_ctx_get_sa_i64(c, static_cast<int64_t *>(SA)));
    return E_OK;
  }();
  if (!cached) dc3hip_ctx_destroy(c);
  else if (rc != E_OK) { dc3hip_ctx_destroy(c); g_cache.c = nullptr; }   // do not keep a context in an unknown state
  return rc;
}

int32_t dc3hip_sufsort_ex(const uint8_t *T, void *SA, int64_t n, const dc3hip_opts *o) {
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
  std::vector<std::thread> pool;
  for (int w = 0; w < workers; w++) {
    pool.emplace_back([&, w]() {
      const int dev = w % ndev;
      for (int64_t part = w; part < nparts; part += workers) {
        const int64_t off = part * S, len = std::min(S, n - off);
        const int rc = sufsort_one(T + off, static_cast<unsigned char *>(SA) + (size_t)off * isz, len, d.index_bits, dev,
                                   false);
        if (rc != E_OK) { rcs[(size_t)w] = rc; msgs[(size_t)w] = dc3hip_last_error(); break; }
      }
      dc3hip_release_cache();            // the worker thread ends here: give its arena back now
    });
  }
  for (auto &t : pool) t.join();
  for (int w = 0; w < workers; w++)
    if (rcs[(size_t)w] != E_OK) { set_err("partition worker %d: %s", w, msgs[(size_t)w].c_str()); return rcs[(size_t)w]; }
  return E_OK;
}

int32_t dc3hip_sufsort_i32(const uint8_t *T, int32_t *SA, int32_t n) {
  return dc3hip_sufsort_ex(T, SA, (int64_t)n, nullptr);
}

int32_t dc3hip_sufsort_i64(const uint8_t *T, int64_t *SA, int64_t n) {
  dc3hip_opts o; memset(&o, 0, sizeof(o));
  o.struct_size = (int32_t)sizeof(o); o.index_bits = 64; o.device = -1;
  return dc3hip_sufsort_ex(T, SA, n, &o);
}

void dc3hip_release_cache(void) {
  if (g_cache.c) { dc3hip_ctx_destroy(g_cache.c); g_cache.c = nullptr; }
}

// divbwt(T, U, A, n) (divsufsort.c:372-405): returns the primary index, -1 / -2 on error; A is an
// optional temporary in the reference and unused here.
int32_t dc3hip_divbwt_i32(const uint8_t *T, uint8_t *U, int32_t *A, int32_t n) {
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
}

int32_t dc3hip_sufcheck_i32(const uint8_t *T, const int32_t *SA, int32_t n) {
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
}

}  // extern "C"

#include "dc3_global_host.hpp"
