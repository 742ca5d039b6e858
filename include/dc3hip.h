/*
 * dc3hip.h — C ABI of libdc3hip.so: MI355X-native (gfx950, HIP) suffix-array construction by
 * DC3/Skew, drop-in for the reference's SACA plug-in boundary.
 *
 * Reference interface this replaces (paths relative to the stringsearch repository):
 *   crates/cdivsufsort/src/lib.rs:1-4       extern "C" { fn divsufsort(T:*const u8, SA:*mut i32, n:i32) -> i32; }
 *   crates/cdivsufsort/c-sources/divsufsort.h:67-76   saint_t divsufsort(const sauchar_t*, saidx_t*, saidx_t)
 *   crates/cdivsufsort/c-sources/divsufsort.c:346-349 argument checks and n in {0,1,2}
 *   crates/dc3/src/lib.rs:44                pub fn suffix_array(T, SA, n, K)   (the algorithm; orphan crate)
 *   crates/sacabase/src/lib.rs:127-149      verify()                            (dc3hip_sufcheck_*)
 *   crates/cdivsufsort/c-sources/utils.c:160-241      sufcheck()                (same return convention)
 *
 * Contract (identical to divsufsort()):
 *   - T: exactly n bytes, borrowed, read-only, no padding required.  SA: caller-allocated n indices,
 *     fully overwritten.  Nothing is retained after return.
 *   - return 0 on success, -1 for invalid arguments (NULL pointers, n < 0), -2 for (device or host)
 *     allocation failure; additionally -3 for a HIP runtime error and -4 when n exceeds what this
 *     build supports (n > DC3HIP_MAX_N).  dc3hip_last_error() describes the failure (thread-local).
 *   - all entry points are re-entrant and thread-safe: every call owns its context and HIP stream
 *     (sacapart calls the SACA concurrently from rayon workers, crates/sacapart/src/lib.rs:41-49).
 *     The one-shot calls keep that context (device buffers only, never caller data) in a per-thread
 *     cache so that repeated calls do not allocate again; dc3hip_release_cache() or thread exit
 *     frees it, DC3HIP_CACHE=0 disables it.  Footprint: a cached context holds n + 4n + 24n..44n bytes of HBM for
 *     the largest n the thread has sorted (29-50 GB per 1 GiB of text: what its builds committed); it is dropped when a later call needs
 *     less than a quarter of it.  Partitioned use should go through dc3hip_sufsort_ex(num_partitions), which
 *     runs one worker per GPU, rather than through concurrent one-shot calls on one device.
 *   - shared HIP runtime state: a call leaves the calling thread's CURRENT DEVICE at the device it worked on, and it reads
 *     (clears) the thread's hipGetLastError() slot on entry — an error the application left there is dropped, not reported
 *     as this call's.  Nothing else of the runtime is touched (own non-blocking stream; no device-wide synchronisation outside
 *     dc3hip_device_synchronize; no hipDeviceReset).
 *   - no C++ exception leaves the library (std::bad_alloc, a thread that cannot be started: -2 / -3 and a message; where a
 *     helper thread is optional — partition workers, the page-touching threads of a one-shot call — the calling thread
 *     does its work instead).
 *   - there is NO CPU fallback: without a usable gfx950 device the calls fail with -3.
 *
 * Plain C, no torch / HIP types in any signature.
 *
 * Environment (read when a context is created; NONE changes a result — every ordering is tested to give the same bytes):
 *   policy       DC3HIP_PROFILE=1 (per-phase HIP events -> dc3hip_stats), DC3HIP_CACHE=0, DC3HIP_WORKERS_PER_DEVICE,
 *                DC3HIP_ARENA_BYTES, DC3HIP_XCD_ASSUME=1 (keep the bucket ordering although the placement probe failed),
 *                DC3HIP_GLOBAL_LOCAL_MAX (global mode: levels this small are finished on every rank's replica),
 *                DC3HIP_QUIET=1 (no line on stderr when the HIP runtime in use is not the one the library was compiled against)
 *   diagnostics  DC3HIP_TRACE=1 (stage checksums), DC3HIP_LEVEL_PHASES=1 (phase times per level on stderr)
 *   test-only    ONE variable, DC3HIP_DEBUG="name[=value],name,...": forces or forbids one of the orderings so that the
 *                parity suite can compare them, lowers a size threshold so that small inputs reach a path, or plants a
 *                fault for a verifier test.  The 30 names (round 6: pruned from 47; tests/test_debug_switches.py compares
 *                this list with the sources):
 *                  structure   no_text_shortcut, no_fullsort, no_hybrid, no_hybrid8, hybrid12_min=, no_long_keys,
 *                              no_doubling, text_order12=0|1, no_small_ties, no_discard
 *                  sorts       no_msd, msd_min=, msd_slot_cap=, no_pack_strip, ssort_min=, ssort_verify, no_wide_window,
 *                              tup_scatter_min=
 *                  memory      vmm_min= (smallest device buffer that is reserved + committed instead of hipMalloc'ed)
 *                  global mode global_no_text_order, global_force_dist, global_force_wide, global_no_route,
 *                              global_no_select, global_link_gbps=, global_device_token, no_wide_msd, wide_msd_min=,
 *                              no_wide_deepen, wide_corrupt=1|2
 */
#ifndef DC3HIP_H
#define DC3HIP_H 1

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC3HIP_API __attribute__((visibility("default")))

/* Largest supported text length: positions and ranks are unsigned 32-bit on the device.  Texts of
 * 2^31 bytes or more need the 64-bit index entry points (int32 cannot hold their positions). */
#define DC3HIP_MAX_N ((int64_t)4278190080)   /* 2^32 - 2^24 */

/* ---- one-shot entry points (the FFI surface a Rust/Go/... shim binds) ------------------------ */

/* Replaces divsufsort(T, SA, n) (cdivsufsort/src/lib.rs:1-4, divsufsort.h:74-76). */
DC3HIP_API int32_t dc3hip_sufsort_i32(const uint8_t *T, int32_t *SA, int32_t n);

/* 64-bit index variant (sacabase only needs Index: ToPrimitive, sacabase/src/lib.rs:165-167). */
DC3HIP_API int32_t dc3hip_sufsort_i64(const uint8_t *T, int64_t *SA, int64_t n);

typedef struct dc3hip_opts {
  int32_t struct_size;   /* = sizeof(dc3hip_opts), for forward compatibility */
  int32_t index_bits;    /* 32 or 64: element type of SA */
  int32_t device;        /* HIP device ordinal, -1 = current device */
  int32_t num_partitions;/* 0/1 = one SA for the whole text; P>1 = sacapart semantics
                            (sacapart/src/lib.rs:39-58): chunk size n/P+1, SA holds the P local
                            suffix arrays back to back (chunk c at offset c*(n/P+1)), local indices */
  int32_t flags;         /* DC3HIP_F_* */
} dc3hip_opts;
#define DC3HIP_F_DEVICE_PTRS 1 /* T and SA are device pointers on `device` (no H2D/D2H).  The library works on a stream of its own
                                  (non-blocking: NOT ordered behind the null stream): whatever produced T must have completed
                                  before the call; SA is complete when the call returns */
#define DC3HIP_F_ALL_DEVICES 2 /* num_partitions > 1, host pointers: the partitions are shared by all visible GPUs, one
                                  host worker thread per GPU (the rayon par_chunks of sacapart/src/lib.rs:45-49);
                                  `device` is ignored.  DC3HIP_WORKERS_PER_DEVICE=2 lets a GPU overlap one chunk's
                                  PCIe transfers with another chunk's build. */

DC3HIP_API int32_t dc3hip_sufsort_ex(const uint8_t *T, void *SA, int64_t n, const dc3hip_opts *opts);

/* sufcheck() twin, computed on the GPU (utils.c:160-241 return codes: 0 ok, -1 invalid arguments,
 * -2 out of range, -3 first characters out of order, -4 suffix in wrong position — an in-range array with
 * duplicate entries reports -3 / -4 like the reference, which has no separate permutation test; library
 * failures are reported as -5 (allocation) / -6 (HIP error), outside sufcheck's own range). */
DC3HIP_API int32_t dc3hip_sufcheck_i32(const uint8_t *T, const int32_t *SA, int32_t n);

/* divbwt() twin (divsufsort.c:372-405, divsufsort.h:78-88): Burrows-Wheeler transform via the GPU suffix
 * array; returns the primary index (>= 0) or a negative error code.  A is ignored (may be NULL). */
DC3HIP_API int32_t dc3hip_divbwt_i32(const uint8_t *T, uint8_t *U, int32_t *A, int32_t n);

/* Free the calling thread's cached one-shot context (see the contract above). */
DC3HIP_API void dc3hip_release_cache(void);

DC3HIP_API const char *dc3hip_version(void);
DC3HIP_API const char *dc3hip_last_error(void); /* thread-local, never NULL */
DC3HIP_API int32_t dc3hip_device_count(void);   /* number of visible HIP devices, <0 on error */
/* Waits until every stream of `device` (-1 = current) has drained: what a host program without HIP bindings of its own
 * brackets a timed region with (bench.py at one GPU runs without torch, on the HIP runtime the library was compiled for). */
DC3HIP_API int32_t dc3hip_device_synchronize(int32_t device);
/* Architecture name ("gfx950:sramecc+:xnack-") and compute-unit count of `device` (-1 = current): lets a host program
 * without HIP bindings tell an MI355X from anything else (the perf guards of the test suite do). */
DC3HIP_API int32_t dc3hip_device_info(int32_t device, char *arch, int32_t arch_len, int32_t *compute_units);
/* HIP_VERSION the library was compiled against and hipRuntimeGetVersion() of the runtime the process really runs on (a
 * host program that maps its own libamdhip64 first — a PyTorch wheel does — makes the library run on that one).  Returns 1
 * when major and minor agree, 0 when they differ (the library also says so once on stderr), <0 on error. */
DC3HIP_API int32_t dc3hip_hip_versions(int32_t *compiled, int32_t *runtime);

/* ---- context API: device-resident builds (bench / repeated calls / multi-GPU hosts) ---------- */

typedef struct dc3hip_ctx dc3hip_ctx;

/* Creates a context on `device` (-1 = current) able to index texts of up to max_n bytes.
 * Device memory: text (max_n + 64 bytes), suffix array (4 max_n) and a work arena.  From 2 GiB on a buffer is a reserved
 * address range (hipMemAddressReserve / hipMemMap) that is committed as it is used, and the arena grows where it lies: a
 * build of random bytes commits about 17 bytes per text byte + 16 for the bucket ordering's slots (a one-shot call's
 * context takes those from its second build on), about 44 when a build enters the DC3 recursion (high-entropy texts never do).  Memory another process
 * or context freed shortly before is handed out by the driver at about 30 ms per GiB (it is wiped first), so a first build
 * can take that much longer than the next.  dc3hip_stats.arena_bytes (the limit) / arena_peak (what the last build used). */
DC3HIP_API int32_t dc3hip_ctx_create(dc3hip_ctx **out, int32_t device, int64_t max_n);
DC3HIP_API void dc3hip_ctx_destroy(dc3hip_ctx *ctx);

/* Load the text: from host memory (H2D copy) ... */
DC3HIP_API int32_t dc3hip_ctx_set_text(dc3hip_ctx *ctx, const uint8_t *T, int64_t n);
/* ... or generate it on the device: byte i = byte (i&7) of splitmix64(seed + (i>>3)) for kind 0,
 * "ACGT"[2-bit field] for kind 1, low-entropy text with 1-8 KiB repeated spans for kind 2
 * (BASELINE.md §3 configs 2/5/3). */
DC3HIP_API int32_t dc3hip_ctx_generate(dc3hip_ctx *ctx, int64_t n, uint64_t seed, int32_t kind);
/* Same stream, bytes [offset, offset+n): lets each GPU of a sacapart run generate its own chunk. */
DC3HIP_API int32_t dc3hip_ctx_generate_at(dc3hip_ctx *ctx, int64_t n, uint64_t seed, int32_t kind, int64_t offset);

/* Build SA[0..n) of the loaded text on the device (text and SA stay in HBM).  Blocking. */
DC3HIP_API int32_t dc3hip_ctx_build(dc3hip_ctx *ctx);

/* Copy results back. */
DC3HIP_API int32_t dc3hip_ctx_get_sa_i32(dc3hip_ctx *ctx, int32_t *SA);
DC3HIP_API int32_t dc3hip_ctx_get_sa_i64(dc3hip_ctx *ctx, int64_t *SA);
DC3HIP_API int32_t dc3hip_ctx_get_text(dc3hip_ctx *ctx, uint8_t *T);

/* Verify the device-resident SA against the device-resident text on the GPU.
 * Returns the sufcheck() codes above. */
DC3HIP_API int32_t dc3hip_ctx_sufcheck(dc3hip_ctx *ctx);

/* Load a suffix array built elsewhere (e.g. by libdivsufsort) next to the text already set.  The by-product calls
 * below (bwt, search) verify such an array once with the GPU sufcheck and fail with -1 if it is not the suffix
 * array of the resident text; lcp tolerates any array (entries out of range yield 0). */
DC3HIP_API int32_t dc3hip_ctx_set_sa_i32(dc3hip_ctx *ctx, const int32_t *SA);

/* bw_transform() of utils.c:53-108 on the device-resident text/SA: U receives n bytes. */
DC3HIP_API int32_t dc3hip_ctx_bwt(dc3hip_ctx *ctx, uint8_t *U, int64_t *primary_index);
/* LCP array of the resident SA (SURVEY §8f "BWT/LCP by-products"; the reference ships no LCP routine, the definition
 * is the usual one): LCP[0] = 0, LCP[i] = length of the longest common prefix of the suffixes SA[i-1] and SA[i].
 * LCP: n x int32, host or device pointer.  n < 2^31. */
DC3HIP_API int32_t dc3hip_ctx_lcp_i32(dc3hip_ctx *ctx, int32_t *LCP);

/* Batched sacabase::longest_substring_match (sacabase/src/lib.rs:39-99): needle k is
 * needles[offsets[k] .. offsets[k+1]); out_start/out_len receive the match exactly as the reference's
 * binary-search narrowing would return it (LongestCommonSubstring.start / .len).  needles, offsets (count+1
 * non-negative, non-decreasing entries), out_start and out_len are HOST pointers. */
DC3HIP_API int32_t dc3hip_ctx_search(dc3hip_ctx *ctx, const uint8_t *needles, const int64_t *offsets, int32_t count,
                                     int64_t *out_start, int64_t *out_len);

/* Batched sacapart::PartitionedSuffixArray::longest_substring_match (crates/sacapart/src/lib.rs:69-97) on the device.
 * The resident array holds the P partition suffix arrays back to back — chunk c = text[c*S .. min(n,(c+1)*S)),
 * S = n/P + 1 (lib.rs:43-46), LOCAL indices: exactly what dc3hip_sufsort_ex(num_partitions = P) writes.  It gets there
 * either by dc3hip_ctx_build_partitions (built on the device, chunk by chunk) or by dc3hip_ctx_set_sa_i32 (then every
 * partition is verified once by the device sufcheck before the first search).  Every needle is searched in every
 * partition with the reference's narrowing loop; a match that reaches its partition's end is re-extended over the whole
 * text (lib.rs:77-84); the strictly longer match wins (lib.rs:86-92).  Arguments and results as dc3hip_ctx_search, starts
 * absolute.  n < 2^31. */
DC3HIP_API int32_t dc3hip_ctx_build_partitions(dc3hip_ctx *ctx, int32_t num_partitions);
DC3HIP_API int32_t dc3hip_ctx_search_partitioned(dc3hip_ctx *ctx, int32_t num_partitions, const uint8_t *needles,
                                                 const int64_t *offsets, int32_t count, int64_t *out_start, int64_t *out_len);

/* 64-bit order-sensitive checksum of the device-resident SA (sum over k of mix(k, SA[k])). */
DC3HIP_API int32_t dc3hip_ctx_sa_checksum(dc3hip_ctx *ctx, uint64_t *out);

#define DC3HIP_MAX_LEVELS 48
enum dc3hip_phase {
  DC3HIP_PH_ALPHABET = 0,   /* byte histogram + dense code table (level 0) */
  DC3HIP_PH_NAME_DIRECT,    /* order-preserving packed-triple names (no sort needed) */
  DC3HIP_PH_PACK,           /* triple records (key, pos) in position order  (ref lib.rs:62-70 + keys) */
  DC3HIP_PH_SORT12_UP,      /* radix passes on sample triples: digit histograms   (ref lib.rs:20-22) */
  DC3HIP_PH_SORT12_SCAN,    /*                                 prefix sums        (ref lib.rs:25-32) */
  DC3HIP_PH_SORT12_DOWN,    /*                                 stable scatter     (ref lib.rs:35-38) */
  DC3HIP_PH_SORT8_DOWN,     /* same, 8-byte (top-32-bit key, pos) records of the prefix-sort path */
  DC3HIP_PH_TIES,           /* tie detection / compaction / write-back of the prefix-sort path */
  DC3HIP_PH_NAMING,         /* flag/scan/assign lexicographic names               (ref lib.rs:80-100) */
  DC3HIP_PH_DISCARD,        /* discarding recursion bookkeeping (reduced string, merge back) */
  DC3HIP_PH_RANKS,          /* rank <- SA12 inversion                             (ref lib.rs:106-113) */
  DC3HIP_PH_TUPLES,         /* merge tuples in slot order + gather to SA12 order */
  DC3HIP_PH_COMPACT,        /* mod-0 suffixes ordered by rank of suffix i+1       (ref lib.rs:118-125) */
  DC3HIP_PH_SORT0,          /* stable radix sort of mod-0 tuples by first symbol  (ref lib.rs:126) */
  DC3HIP_PH_MERGE,          /* merge-path merge of SA12 and SA0                   (ref lib.rs:131-192) */
  DC3HIP_PH_OTHER,
  DC3HIP_PH_COUNT
};

typedef struct dc3hip_stats {
  int32_t struct_size;
  int32_t levels;                         /* recursion depth reached by the last build */
  int64_t level_n[DC3HIP_MAX_LEVELS];     /* string length per level (level 0 = n) */
  int64_t level_K[DC3HIP_MAX_LEVELS];     /* alphabet bound per level */
  int32_t level_sorted[DC3HIP_MAX_LEVELS];/* 0 = direct packed names, 1 = names by full radix sort,
                                             2 = names by prefix sort + tie refinement,
                                             3 / 4 = 1 / 2 followed by the discarding recursion,
                                             5 = whole level ordered at once (all its triples distinct; at level 0:
                                                 all windows of the text distinct - 9 bytes, or 3L symbols of a small
                                                 alphabet, L = symbols per 32-bit limb in base sigma+1),
                                             6 = level 0 only: the whole text ordered at once and the few positions whose
                                                 windows repeat settled by prefix doubling (level_tied[0] = those positions,
                                                 level_kept[0] = doubling rounds) */
  int64_t level_kept[DC3HIP_MAX_LEVELS];  /* length of the reduced recursive string (discarding) */
  int32_t level_name_width[DC3HIP_MAX_LEVELS]; /* symbols packed per direct name (0 on sorted levels) */
  int64_t level_tied[DC3HIP_MAX_LEVELS];  /* samples re-sorted by the full key (prefix-sort path) */
  double  level_tie_pred[DC3HIP_MAX_LEVELS]; /* predicted tied fraction (policy input) */
  double  build_ms;                       /* HIP-event time of the whole device-resident build */
  double  phase_ms[DC3HIP_PH_COUNT];      /* HIP-event time per phase, summed over levels */
  int64_t phase_launches[DC3HIP_PH_COUNT];
  /* the stable radix scatter kernel k_rs_downsweep<Rec,...>, per record type:
   * [0] 8-byte (key,value) pairs, [1] 16-byte triple records, [2] 20-byte mod-0 tuples */
  double  downsweep_ms[3];                /* summed HIP-event time of the launches */
  int64_t downsweep_launches[3];
  int64_t downsweep_elems[3];             /* records moved, summed over launches */
  /* k_part_msd: non-stable window partition of (destination,value) pairs (inverse permutations) */
  double  partition_ms; int64_t partition_launches; int64_t partition_elems;
  /* k_gather_tuples: the one random 16-byte gather per sample suffix */
  double  gather_ms; int64_t gather_launches; int64_t gather_elems;
  int64_t arena_bytes;                    /* device work arena size */
  int64_t arena_peak;                     /* high-water mark of the last build */
  /* whole-text shortcut (all n positions ordered by their 9-byte - small alphabets: 3L-symbol - windows before any
   * recursion level is built; 8-byte records, 12-byte ones beyond 2^31 positions):
   * 0 = not tried, 1 = all keys distinct, that order is the SA (levels == 1, level_sorted[0] == 5),
   * 2 = duplicate keys, the order was filtered into level 1's sorted samples, 3 = abandoned (too many ties) */
  int32_t text_sort_state;
  int32_t trace_on;                       /* DC3HIP_TRACE=1 was set when the context was created */
  /* Stage-level parity trace (DC3HIP_TRACE=1; the counterpart of the reference's crosscheck! macro,
   * crates/divsufsort/src/crosscheck.rs:17-84).  Per level: sum over k of mix(k, x_k) for the sorted samples
   * (x = position in the level's string, dummy included), the sorted mod-0 suffixes and the level's suffix array;
   * 0 where a level did not build the array (whole-level / whole-text order).  trace_names = distinct triples of
   * a sorted level (lib.rs:104 `name`), -1 where names were packed directly.  The CPU restatement produces the same
   * words (the restatement the tests check against), so a failing build can be narrowed to a level and a stage. */
  uint64_t trace_sa12[DC3HIP_MAX_LEVELS];
  uint64_t trace_sa0[DC3HIP_MAX_LEVELS];
  uint64_t trace_sa[DC3HIP_MAX_LEVELS];
  int64_t  trace_names[DC3HIP_MAX_LEVELS];
  /* bucket (MSD) ordering of the prefix-sort words (dc3_msd.hip.hpp), which replaces the stable LSD passes over
   * 8-byte records where the key images are well spread: k_msd_part = one non-stable partition pass with XCD-grouped
   * reservation (2^30 words per launch at 1 GiB), k_msd_local = the in-LDS order of the sub-buckets.
   * msd_sorts = sorts that took this path in the last build, msd_fallbacks = sorts that gave it up (a sub-bucket too
   * large for LDS) and ran the LSD passes instead. */
  double  msd_part_ms;  int64_t msd_part_launches;  int64_t msd_part_elems;
  double  msd_local_ms; int64_t msd_local_launches; int64_t msd_local_elems;
  int32_t msd_sorts, msd_fallbacks;
  int64_t msd_max_subbucket;              /* largest sub-bucket of the last MSD sort */
  /* splitter (sample) ordering of the 12- / 16-byte sample-triple records (dc3_ssort.hip.hpp), which replaces the stable
   * LSD passes of the straight orderings on large levels: k_ss_part = one partition pass over sampled splitters,
   * k_ss_local = the in-LDS comparison order of the sub-buckets.  ssort_fallbacks = sorts that gave it up before
   * pass 2 (a sub-bucket beyond the local capacity) and ran the LSD passes instead. */
  double  ssort_part_ms;  int64_t ssort_part_launches;  int64_t ssort_part_elems;
  double  ssort_local_ms; int64_t ssort_local_launches; int64_t ssort_local_elems;
  int32_t ssort_sorts, ssort_fallbacks;
  int64_t ssort_max_subbucket;
  /* The XCD-grouped partition kernels (k_msd_part, k_wide_part1) owe their speed to "blocks with equal blockIdx % 8 run
   * on one XCD" (a placement the dispatcher is observed to use, not a promise).  xcd_round_robin: the probe of context
   * creation (4096 small blocks report HW_REG_XCC_ID): 1 = they did; 0 = they did not, and the context then orders with the
   * stable 256-bucket LSD passes instead (DC3HIP_XCD_ASSUME=1 keeps the bucket ordering).  xcd_blocks / xcd_group_hit:
   * the grouped partition blocks of the LAST build and the share of them that ran on their group's majority XCD (1.0 = the
   * assumption held throughout, 0.125 = placement unrelated to blockIdx) — also under other streams on the device. */
  int32_t xcd_round_robin;
  int32_t msd_slot_sorts;                 /* MSD sorts of the last build whose pass 2 wrote into sub-bucket slots (no counting sweep) */
  int64_t xcd_blocks;
  double  xcd_group_hit;
  /* k_msd_part_keys / k_wide_part1: partition pass 1 of a bucket ordering that also MAKES the words it partitions (from
   * the text: the pack kernel only counted) — Step 0 (lib.rs:62-70) and one radix pass (lib.rs:35-38) in one launch;
   * timed apart from msd_part_*, which then holds pass 2 only. */
  double  msd_part_keys_ms; int64_t msd_part_keys_launches; int64_t msd_part_keys_elems;
} dc3hip_stats;

DC3HIP_API int32_t dc3hip_ctx_stats(dc3hip_ctx *ctx, dc3hip_stats *out);

/* Test hook for kernel-level parity against the reference's radix_pass (crates/dc3/src/lib.rs:15-39): ONE stable pass
 * of the product's radix scatter over n host words, digit = (word >> shift) & (nb - 1), nb = 256 or 512. */
DC3HIP_API int32_t dc3hip_ctx_debug_radix_pass_u64(dc3hip_ctx *ctx, const uint64_t *words, uint64_t *out, int64_t n,
                                                   int32_t shift, int32_t nb);

/* ---- GLOBAL mode: one suffix array over P ranks (one rank per GPU) -------------------------------------------
 * The other multi-GPU semantics of SURVEY.md §8(e).  sacapart (crates/sacapart/src/lib.rs:39-58; here
 * dc3hip_sufsort_ex(num_partitions)) keeps P independent local arrays.  A global context builds the TRUE SA[0..n) of
 * the whole text — bit-identical to dc3hip_sufsort_* / divsufsort on the concatenated text — with the text sharded
 * over the ranks in sacapart-style blocks (rank r owns bytes [r*S, min(n,(r+1)*S)), S = n/P + 1, lib.rs:43-46) and
 * the result sharded by suffix rank (rank r holds SA[first_r .. first_r + count_r), the shards in rank order
 * concatenate to the suffix array).  dc3hip_global_build is a COLLECTIVE: every rank of the group must call it.
 * Transport: RCCL over xGMI (one process per GPU; the host program carries the 128-byte unique id from rank 0 to
 * the others, e.g. with torch.distributed / MPI), or the in-process loopback (P ranks on ONE device, used to
 * parity-test P in {2,4,8} on a single-GPU box).  n <= DC3HIP_MAX_N; P <= 16.
 * Failure semantics: a collective call (build, sufcheck) that fails on one rank (e.g. -2) returns there at once; loopback
 * peers are released and return -3 ("another rank failed"), and once EVERY rank has returned from the failed call the
 * group can be used again through any entry point.  Refusals of the wide mode that depend on the data and on the rank (a
 * rank's share above DC3HIP_MAX_N suffixes: -4, no device memory for its records: -2) are agreed on inside the build: every
 * rank returns the same code, under every transport.  Other faults under RCCL / the host-staged transport leave the peers
 * inside a collective, as in any NCCL program: the host job's watchdog has to tear the group down.
 * Environment: DC3HIP_GLOBAL_LOCAL_MAX (levels up to this length are finished by every rank on its own replicated
 * copy, default 2^22).
 * WIDE contexts (max_total_n > DC3HIP_MAX_N, up to 2^40): positions are 64-bit, and the
 * order is the distributed whole-text order (BASELINE.json configs[3] random bytes at 4 GiB, configs[4] random DNA at
 * 16 GiB).  Windows that repeat are compared deeper (256, 8192, then 16x more symbols per round while that is cheap) and
 * beyond that settled by rank look-ups: all ranks exchange their shards, build the inverse of the order so far, and 17
 * look-ups per compare settle 17x the depth per round (9 bytes per suffix of the TEXT on every rank plus one rank's shard
 * at a time: dc3hip_global_plan).  Groups of tied suffixes of any size are ordered (beyond 1024 members by a segmented
 * sort of the members: a run of one symbol, a short period, a text over one symbol).  Refused with -4 on every rank: a rank
 * whose share would exceed DC3HIP_MAX_N suffixes, no memory for the look-ups.
 * Shards are fetched with dc3hip_global_get_shard_i64 (…_u32 returns -4) and verified with the
 * collective dc3hip_global_sufcheck. */
typedef struct dc3hip_gctx dc3hip_gctx;

typedef struct dc3hip_gstats {
  int32_t struct_size;
  int32_t nranks, rank;
  int32_t levels;            /* recursion depth of the last build */
  int32_t text_order;        /* 1 = finished by the distributed whole-text order (all 9-byte / 3L-symbol windows distinct) */
  int32_t local_from_level;  /* first level that every rank finished locally on its replicated copy (-1: none) */
  int64_t total_n, shard_first, shard_count;
  int64_t exchanges;         /* rank exchanges (all-to-all + all-gather) of the last build */
  int64_t exchange_pairs;    /* (destination, value) pairs this rank fed into them */
  int64_t comm_bytes_out, comm_bytes_in;   /* payload over the transport, self copies excluded */
  double  comm_ms;           /* host wall time inside collectives (incl. waiting for the slowest rank) */
  double  device_ms;         /* HIP-event time of this rank's build stream, start to end (incl. transport) */
  double  wall_ms;           /* host wall time of dc3hip_global_build on this rank */
  int64_t wide_msd;          /* wide mode: 1 = this rank's order came from the bucket ordering on 8-byte words, 0 = 16-byte LSD passes */
  int64_t select_p1;         /* orderings of this rank whose partition pass 1 selected the rank's key range from the replicated string (no records built or routed) */
  int64_t wide_deepen_rounds; /* wide mode: rounds of deepening by rank look-ups (windows repeated beyond the symbol compares; 0 = not needed) */
  /* What one rank costs on its OWN GPU, from a run in which the ranks may share one (loopback): */
  double  work_ms;           /* host wall time this rank spent OUTSIDE collectives.  Loopback ranks that share a device can be made to
                              * hold a device token while they work and hand it over inside collectives (DC3HIP_DEBUG=
                              * global_device_token): then this is the rank's own work, not its share of a time-sliced GPU */
  double  link_ms;           /* model of the transport on xGMI: per collective, the most bytes this rank exchanges with ONE peer
                              * (a link) / 153 GB/s, summed over the build's collectives */
  int64_t collectives;       /* collectives of the build (device-side all-to-all / all-gather) */
} dc3hip_gstats;

/* Per-rank HBM need of a global-mode build of total_n bytes over nranks ranks, computed by the library's own sizing rules
 * (the formulas its allocations use) WITHOUT touching a device: what a deployment checks before it asks for an 8-GPU node.
 * wide = 1: total_n > DC3HIP_MAX_N, 64-bit positions.  All figures in bytes, per rank. */
typedef struct dc3hip_gplan {
  int32_t struct_size, wide;
  int64_t total_n;
  int32_t nranks, reserved;
  int64_t records_per_rank;   /* suffixes a rank is planned to hold: ceil(n / P) * 9/8 (key-range splitters from a sample) */
  int64_t text_bytes;         /* the replicated text */
  int64_t context_bytes;      /* SA / small buffers of the rank's context (not wide: 4 bytes per text byte) */
  int64_t arena_bytes;        /* work arena at its largest */
  int64_t order_bytes;        /* wide: record buffers + shard (+ flags) of the ordering at their largest; else 0 (inside the arena) */
  int64_t deepen_bytes;       /* wide: shard + inverse + flags + one rank's shard while repeats are settled by rank look-ups */
  int64_t big_group_bytes_per_member; /* wide: on top, per suffix that sits in a group of more than 1024 tied suffixes */
  int64_t peak_bytes;         /* text + context + arena + max(order, deepen) */
  int64_t hbm_bytes;          /* 288e9: one MI355X */
} dc3hip_gplan;
DC3HIP_API int32_t dc3hip_global_plan(int64_t total_n, int32_t nranks, dc3hip_gplan *out);

/* P loopback ranks on `device` (-1 = current), each able to take part in builds of up to max_total_n bytes.
 * device = DC3HIP_DEVICE_SPREAD puts rank r on device r % (visible devices): ONE process drives all GPUs of the node
 * (the global-mode counterpart of DC3HIP_F_ALL_DEVICES), with peer copies as the transport — no RCCL, no MPI. */
#define DC3HIP_DEVICE_SPREAD (-2)
DC3HIP_API int32_t dc3hip_global_loopback_create(dc3hip_gctx **ranks /*[P]*/, int32_t P, int32_t device, int64_t max_total_n);
/* runs dc3hip_global_build of the P ranks on P host threads and waits for all of them */
DC3HIP_API int32_t dc3hip_global_loopback_build(dc3hip_gctx **ranks, int32_t P);
/* RCCL: rank 0 obtains the id, the host program distributes it, every process creates its rank */
DC3HIP_API int32_t dc3hip_rccl_unique_id(uint8_t *id128);
/* Which RCCL the library bound to: the file its entry points were resolved from (dladdr), and whether the host program
 * had that library mapped already (PyTorch maps its own torch/lib/librccl.so; two RCCL instances in one process must not
 * happen — the library looks for a mapped one first).  bench.py prints it and compares it with /proc/self/maps. */
DC3HIP_API int32_t dc3hip_rccl_library_path(char *buf, int32_t len, int32_t *was_already_mapped);
DC3HIP_API int32_t dc3hip_global_rccl_create(dc3hip_gctx **out, const uint8_t *id128, int32_t rank, int32_t nranks,
                                             int32_t device, int64_t max_total_n);
/* Host-staged transport: the caller supplies the two collectives on HOST buffers (MPI, gloo, ...); the library
 * stages device data through pinned memory.  For nodes without GPU peer access, and for multi-process runs on one
 * GPU (RCCL refuses two ranks on one device).  Offsets and sizes are bytes, arrays have nranks entries.
 *   all_to_all_v: send[soff[r] .. +sbytes[r]) goes to rank r; rbytes[r] bytes from rank r land at recv + roff[r]
 *   all_gather_v: every rank contributes sbytes bytes; rank r's land at recv + roff[r] (rbytes[r] bytes) everywhere
 * Both return 0 on success. */
typedef struct dc3hip_host_transport {
  void *user;
  int32_t (*all_to_all_v)(void *user, const void *send, const uint64_t *soff, const uint64_t *sbytes, void *recv,
                          const uint64_t *roff, const uint64_t *rbytes);
  int32_t (*all_gather_v)(void *user, const void *send, uint64_t sbytes, void *recv, const uint64_t *roff,
                          const uint64_t *rbytes);
} dc3hip_host_transport;
DC3HIP_API int32_t dc3hip_global_host_create(dc3hip_gctx **out, const dc3hip_host_transport *t, int32_t rank,
                                             int32_t nranks, int32_t device, int64_t max_total_n);
DC3HIP_API void dc3hip_global_destroy(dc3hip_gctx *g);
/* the block of a text of total_n bytes this rank owns */
DC3HIP_API int32_t dc3hip_global_block(dc3hip_gctx *g, int64_t total_n, int64_t *offset, int64_t *length);
/* load this rank's block (host or device pointer, `length` bytes of dc3hip_global_block) ... */
DC3HIP_API int32_t dc3hip_global_set_text_block(dc3hip_gctx *g, const uint8_t *block, int64_t total_n);
/* ... or generate it on the device (the stream of dc3hip_ctx_generate, bytes [offset, offset+length)) */
DC3HIP_API int32_t dc3hip_global_generate(dc3hip_gctx *g, int64_t total_n, uint64_t seed, int32_t kind);
DC3HIP_API int32_t dc3hip_global_build(dc3hip_gctx *g);
/* this rank's shard: SA[first .. first+count) */
DC3HIP_API int32_t dc3hip_global_shard(dc3hip_gctx *g, int64_t *first, int64_t *count);
DC3HIP_API int32_t dc3hip_global_get_shard_i64(dc3hip_gctx *g, int64_t *out);
DC3HIP_API int32_t dc3hip_global_get_shard_u32(dc3hip_gctx *g, uint32_t *out);
/* sum over the shard of mix(global index, SA[index]); the sum over all ranks equals dc3hip_ctx_sa_checksum of a
 * single-device build of the same text */
DC3HIP_API int32_t dc3hip_global_shard_checksum(dc3hip_gctx *g, uint64_t *out);
/* WIDE contexts only; a COLLECTIVE like the build.  0 = the concatenated shards are the suffix array (every position in
 * range, every entry's suffix strictly smaller than its successor's — across rank boundaries too —, sizes add up to n);
 * -2 / -3 = position out of range / order violated (the codes of the reference's sufcheck, utils.c:160-241), identical
 * on all ranks; <= -11 = the check could not run (dc3hip error code - 10). */
DC3HIP_API int32_t dc3hip_global_sufcheck(dc3hip_gctx *g);
DC3HIP_API int32_t dc3hip_global_stats(dc3hip_gctx *g, dc3hip_gstats *out, dc3hip_stats *ctx_stats /* may be NULL */);
DC3HIP_API const char *dc3hip_global_last_error(dc3hip_gctx *g);   /* error of this rank's last build (loopback threads) */
DC3HIP_API const char *dc3hip_global_transport(dc3hip_gctx *g);
/* Transport self-test, a COLLECTIVE like the build: a ragged all-to-all, a ragged all-gather and a host all-gather of
 * known bytes through this group's transport, every byte checked on every rank.  0 = delivered exactly; -3 = a byte was
 * wrong or the transport failed (dc3hip_global_last_error says where).  *transport_ranks (may be NULL) = the number of
 * ranks the transport itself reports (RCCL: ncclCommCount; -1 if it cannot tell; other transports: nranks) — a host
 * program uses it to refuse to label a run "RCCL over xGMI" that is not (bench.py does, at the start of every N > 1 run). */
DC3HIP_API int32_t dc3hip_global_selftest(dc3hip_gctx *g, int32_t *transport_ranks);

#ifdef __cplusplus
}
#endif
#endif /* DC3HIP_H */
