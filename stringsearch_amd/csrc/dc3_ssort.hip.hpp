// dc3_ssort.hip.hpp — splitter (sample) ordering of the 12- and 16-byte sample-triple records.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); namespace dc3.
//
// The straight ordering of a level (lib.rs:62-78: the three radix_pass calls over the sample triples) sorts records
// (key, pos) whose keys are names of the level above: on text they are badly skewed (a few triples carry most of the
// mass) and wider than a word, so the bucket ordering of dc3_msd.hip.hpp — digits taken from key BITS — does not
// apply, and the stable LSD passes cost 5 (45-bit keys) to 9 (81-bit keys) sweeps over the records.  Here the buckets
// come from the data instead:
//   sample    one record per cell of n / S records (jittered), S = 24 per sub-bucket; the sample is sorted by the LSD passes (it is ~2 % of n)
//   splitters every 24th sample value = one of n2 - 1 fine splitters; every F2-th fine splitter = a coarse one
//   pass 1    k_ss_part<.., false>   partition by the 1023 coarse splitters   (sizes: k_ss_count1)
//   pass 2    k_ss_part<.., true>    partition every bucket by its F2 - 1 fine splitters  (sizes: k_ss_hist2 + scans)
//   pass 3    k_ss_local             every sub-bucket (about 1400 records, at most 4096) is ordered inside LDS
// A record's sort value is (key, pos): all values are distinct, so "ascending" is one array whatever the partition
// passes do (they are not stable: XCD-grouped atomic reservation, see dc3_msd.hip.hpp), and it is the array the stable
// LSD passes produce from records in position order.  Buckets are balanced by construction — a sub-bucket holds the
// records between two sample values 24 samples apart (its size is Gamma(24)-distributed around the mean of 1400: beyond
// the capacity of 4096 with probability 3e-11; 16 samples gave 4e-8, i.e. one sort in fifty with a fallback) — so skew
// and repeated keys (which the position breaks) do not matter; a sub-bucket above the local capacity is reported before
// pass 2 and the caller runs the LSD passes instead.
//
// Local order: a comparison sort, since the keys inside a sub-bucket share no usable bit structure.  127 or 255 of the
// sub-bucket's own records are ranked against each other (all pairs, broadcast reads) and become bin boundaries; every
// record finds its bin by binary search, bins are laid out in LDS by arrival, and each record counts the smaller
// records of its bin (5-11 of them).
#pragma once

namespace dc3 {

struct __attribute__((aligned(16))) SsVal { u64 hi, lo; };
__device__ __forceinline__ SsVal ss_val(const Rec12 &r) { SsVal v; v.hi = ((u64)r.k1 << 32) | r.k0; v.lo = (u64)r.pos; return v; }
template <class Rec> __device__ __forceinline__ Rec ss_max_rec();
template <> __device__ __forceinline__ Rec12 ss_max_rec<Rec12>() { Rec12 r; r.k0 = r.k1 = r.pos = ~0u; return r; }
template <> __device__ __forceinline__ Rec16 ss_max_rec<Rec16>() { Rec16 r; r.k0 = r.k1 = r.k2 = r.pos = ~0u; return r; }
__device__ __forceinline__ SsVal ss_val(const Rec16 &r) { SsVal v; v.hi = ((u64)r.k2 << 32) | r.k1; v.lo = ((u64)r.k0 << 32) | r.pos; return v; }
// a < b is the borrow out of the 128-bit difference a - b: four subtract-with-borrow instructions on the 32-bit words.
// (Written as a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo) it compiles to three 64-bit compares, two mask operations and
// two moves — the halves of lo, k0 and pos, are not neighbours in a Rec16 — and the compiler folds __builtin_subc chains
// back into that form: hence the assembly.  The ordering kernels are bound by exactly these instructions.)
#define DC3_SS_BORROW(A, B)                                                                                              \
  "v_sub_co_u32 %[t], vcc, %[" #A "0], %[" #B "0]\n\tv_subb_co_u32 %[t], vcc, %[" #A "1], %[" #B "1], vcc\n\t"            \
  "v_subb_co_u32 %[t], vcc, %[" #A "2], %[" #B "2], vcc\n\tv_subb_co_u32 %[t], vcc, %[" #A "3], %[" #B "3], vcc\n\t"
#define DC3_SS_WORDS(a, b)                                                                                               \
  [a0] "v"((u32)a.lo), [a1] "v"((u32)(a.lo >> 32)), [a2] "v"((u32)a.hi), [a3] "v"((u32)(a.hi >> 32)), [b0] "v"((u32)b.lo),  \
      [b1] "v"((u32)(b.lo >> 32)), [b2] "v"((u32)b.hi), [b3] "v"((u32)(b.hi >> 32))
// acc += (a < b)
__device__ __forceinline__ void ss_acc_lt(u32 &acc, const SsVal &a, const SsVal &b) {
  u32 t;
  asm(DC3_SS_BORROW(a, b) "v_addc_co_u32 %[acc], vcc, 0, %[acc], vcc" : [t] "=&v"(t), [acc] "+v"(acc) : DC3_SS_WORDS(a, b) : "vcc");
}
// s <= x ? yes : no      (s <= x is "no borrow out of x - s")
__device__ __forceinline__ u32 ss_sel_le(const SsVal &s, const SsVal &x, u32 yes, u32 no) {
  u32 t, r;
  asm(DC3_SS_BORROW(b, a) "v_cndmask_b32 %[r], %[yes], %[no], vcc" : [t] "=&v"(t), [r] "=v"(r) : DC3_SS_WORDS(s, x), [yes] "v"(yes), [no] "v"(no) : "vcc");
  return r;
}
__device__ __forceinline__ bool ss_lt(const SsVal &a, const SsVal &b) { u32 c = 0; ss_acc_lt(c, a, b); return c != 0u; }
__device__ __forceinline__ bool ss_le(const SsVal &a, const SsVal &b) { u32 c = 0; ss_acc_lt(c, b, a); return c == 0u; }

constexpr int kSsNT = 1024;                    // threads of a partition block
constexpr u32 kSsGroups = 8;                   // XCDs (as kMsdGroups)
constexpr u32 kSsMaxDig = 1024;                // coarse buckets, and the most fine splitters + 1 per bucket
constexpr u32 kSsLocPad = 4;                   // records of all ones behind a sub-bucket's LDS image (k_ss_local's last step)
constexpr u32 kSsHistTiles = 8;                // partition tiles per block of k_ss_hist2
template <class Rec> struct SsCfg;
template <> struct SsCfg<Rec12> { static constexpr int IPT = 8; };      // 8192-record tiles (96 KB)
template <> struct SsCfg<Rec16> { static constexpr int IPT = 6; };      // 6144-record tiles (96 KB)
template <class Rec> constexpr size_t ss_part_smem() {
  return sizeof(Rec) * kSsNT * SsCfg<Rec>::IPT + sizeof(u32) * (2 * kSsMaxDig + 64) + sizeof(uint16_t) * kSsNT * SsCfg<Rec>::IPT;
}

// Splitter tables in LDS are search trees in breadth-first order: (1 << steps) - 1 entries (padded with +inf), the root
// at slot 0, the children of slot j - 1 at slots 2j - 1 and 2j.  ss_tree_slot(t) is the slot of the t-th smallest entry.
// (In ascending order the probes of one step of a binary search lie a power of two of 16-byte entries apart — from the
// second step to the seventh every lane's probe falls into the same four LDS banks, and the counters showed three of
// four LDS cycles of k_ss_hist2 as bank conflicts; in breadth-first order the entries one step can probe are neighbours.)
__device__ __forceinline__ u32 ss_tree_slot(u32 t, u32 steps) {
  const u32 z = (u32)__builtin_ctz(t + 1u);
  return (1u << (steps - 1u - z)) + ((t + 1u) >> (z + 1u)) - 1u;
}
// how many of the table's entries are <= x, for K values at once, the K searches advancing in lockstep (K independent
// LDS reads in flight per step instead of one dependent chain per value).  x, pos: K registers each.
template <int K>
__device__ __forceinline__ void ss_count_le_n(const SsVal *spl, u32 steps, const SsVal *x, u32 *pos) {
#pragma unroll
  for (int k = 0; k < K; k++) pos[k] = 1;                  // node numbers (slot + 1)
  for (u32 l = 0; l < steps; l++) {
    SsVal s[K];
#pragma unroll
    for (int k = 0; k < K; k++) s[k] = spl[pos[k] - 1u];
#pragma unroll
    for (int k = 0; k < K; k++) pos[k] = ss_sel_le(s[k], x[k], 2u * pos[k] + 1u, 2u * pos[k]);
  }
#pragma unroll
  for (int k = 0; k < K; k++) pos[k] -= 1u << steps;
}
template <int K>
__device__ __forceinline__ void ss_count_le_multi(const SsVal *spl, u32 steps, const SsVal (&x)[K], u32 (&pos)[K]) {
  ss_count_le_n<K>(spl, steps, x, pos);
}
// stage ns ascending splitters from global memory as such a table; blockDim.x >= (1 << steps) - 1
__device__ __forceinline__ void ss_stage(SsVal *spl, const SsVal *__restrict__ g, u32 ns, u32 steps) {
  const u32 t = threadIdx.x;
  if (t < (1u << steps) - 1u) {
    SsVal v;
    if (t < ns) v = g[t]; else { v.hi = ~0ull; v.lo = ~0ull; }
    spl[ss_tree_slot(t, steps)] = v;
  }
}
__device__ __forceinline__ u32 ss_steps(u32 ns) { u32 s = 1; while (((1u << s) - 1u) < ns) s++; return s; }   // ns >= 1

// Wide windows: the record of sample position p holds W > 3 symbols, sb bits each, first symbol most significant (the
// 96-bit key of Rec16).  Names taken from such records refine the triple names of lib.rs:80-100 — equal windows imply
// equal triples — and order the sample suffixes just as well: comparing (W(p), W(p+3), ...) with (W(q), W(q+3), ...) is
// comparing the suffixes at p and q, the windows that reach past the end are distinct (they differ in where their first
// sentinel is), so the recursion of lib.rs:104 is entered with fewer repeated names, or not at all.
// (ss_window_rec, the record of a window, is in dc3_names.hip.hpp next to make_rec)
// records of the sample positions in position order (k_pack_triples with W symbols): thread g makes those of 3g+1, 3g+2
template <class Sym, int W>
__global__ __launch_bounds__(kBlock) void k_pack_window16(Sym S, u32 m, u32 m02, u32 sb, Rec16 *__restrict__ out) {
  const u32 g = blockIdx.x * kBlock + threadIdx.x;
  if (2 * g >= m02) return;
  const u32 i = 3 * g + 1;
  u32 s[W + 1];
#pragma unroll
  for (int j = 0; j <= W; j++) s[j] = S.get(i + j);
  out[2 * g] = ss_window_rec(s, W, sb, i);
  if (2 * g + 1 < m02) out[2 * g + 1] = ss_window_rec(s + 1, W, sb, i + 1);
}

// Index of sample i of S among n records: one per cell [i n / S, (i + 1) n / S), at a pseudo-random offset inside the cell
// (a fixed function of i: the build stays deterministic).  Ascending in i, i.e. ascending pos.  A regular stride
// resonates with texts that repeat at fixed distances: a generated text of 3 * 2^23 bytes sampled every 58th record
// made one sub-bucket of more than 4096 records — the fallback, not an error, but 10 LSD passes instead of 3.
__device__ __forceinline__ u32 ss_sample_index(u32 i, u32 n, u32 S) {
  const u64 lo = (u64)i * n / S, hi = (u64)(i + 1) * n / S;      // hi - lo >= 4 (S <= n / 4)
  u32 h = i * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
  const u64 idx = lo + h % (u32)(hi - lo);
  return (u32)(idx < n ? idx : n - 1);
}
template <class Rec>
__global__ __launch_bounds__(kBlock) void k_ss_sample(const Rec *__restrict__ in, u32 n, u32 S, Rec *__restrict__ out) {
  const u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= S) return;
  out[i] = in[ss_sample_index(i, n, S)];
}

// fine[j] = value of sorted sample (j + 1) * over, j < n2 - 1 (fine[n2 - 1] = +inf); coarse[b] = fine[(b + 1) * F2 - 1]
template <class Rec>
__global__ __launch_bounds__(kBlock) void k_ss_splitters(const Rec *__restrict__ ss, u32 n2, u32 F2, u32 over, SsVal *__restrict__ fine,
                                                        SsVal *__restrict__ coarse) {
  const u32 j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= n2) return;
  SsVal v;
  if (j == n2 - 1) { v.hi = ~0ull; v.lo = ~0ull; }
  else v = ss_val(ss[(size_t)(j + 1) * over]);
  fine[j] = v;
  if ((j + 1) % F2 == 0) coarse[(j + 1) / F2 - 1] = v;      // (coarse[nb1 - 1] = +inf, never searched)
}

// Sizes of the coarse buckets per group: block j belongs to group g = j % 8 and counts tiles
// [g * cpx + idx * tpb, ... + tpb) of that group's eighth (idx = j / 8); cntg[d * 8 + g] += its counts.
template <class Rec>
__global__ __launch_bounds__(kSsNT, 8) void k_ss_count1(const Rec *__restrict__ in, u32 n, const SsVal *__restrict__ coarse, u32 nb1,
                                                    u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 *__restrict__ cntg,
                                                    uint16_t *__restrict__ dig) {
  __shared__ SsVal spl[kSsMaxDig];
  __shared__ u32 hist[kSsMaxDig];
  const u32 tid = threadIdx.x;
  const u32 g = blockIdx.x % kSsGroups, idx = blockIdx.x / kSsGroups;
  const u32 t0 = g * cpx + idx * tpb;
  const u32 t1 = min(min(t0 + tpb, (g + 1) * cpx), ntiles);
  if (idx * tpb >= cpx || t0 >= t1) return;
  const u32 steps = ss_steps(nb1 - 1);
  ss_stage(spl, coarse, nb1 - 1, steps);
  hist[tid] = 0;
  __syncthreads();
  const u32 begin = t0 * tile, end = min(n, t1 * tile);     // (32-bit: n < 2^32 - tile)
  // (four searches in lockstep: eight held 64 registers of values and splitters, past the two-blocks-per-CU bound)
  for (u32 i = begin + tid; i < end; i += 4 * kSsNT) {
    SsVal v[4];
    u32 d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = ss_val(in[min(i + (u32)k * kSsNT, end - 1u)]);
    ss_count_le_multi<4>(spl, steps, v, d);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + (u32)k * kSsNT < end) { atomicAdd(&hist[d[k]], 1u); dig[i + (u32)k * kSsNT] = (uint16_t)d[k]; }
  }
  __syncthreads();
  if (tid < nb1 && hist[tid]) atomicAdd(&cntg[tid * kSsGroups + g], hist[tid]);
}

// The two kernels above for records that do not exist yet (the wide-window ordering): the sample is computed from the
// level's string, and the counting kernel MAKES the records (k_pack_window16's job), stores them, and counts — the
// records are written once and not read back for the count.  Packed index x is position 3 (x / 2) + 1 + (x & 1).
template <class Sym, int W>
__device__ __forceinline__ Rec16 ss_window_at(const Sym &S, u32 x, u32 sb) {
  const u32 p = 3 * (x >> 1) + 1 + (x & 1u);
  u32 s[W];
#pragma unroll
  for (int j = 0; j < W; j++) s[j] = S.get(p + j);
  return ss_window_rec(s, W, sb, p);
}
template <class Sym, int W>
__global__ __launch_bounds__(kBlock) void k_ss_sample_window(Sym S, u32 sb, u32 n, u32 Sn, Rec16 *__restrict__ out) {
  const u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= Sn) return;
  out[i] = ss_window_at<Sym, W>(S, ss_sample_index(i, n, Sn), sb);
}
template <class Sym, int W>
__global__ __launch_bounds__(kSsNT, 8) void k_ss_pack_count1(Sym S, u32 sb, Rec16 *__restrict__ recs, u32 n, const SsVal *__restrict__ coarse,
                                                         u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 *__restrict__ cntg,
                                                         uint16_t *__restrict__ dig) {
  __shared__ SsVal spl[kSsMaxDig];
  __shared__ u32 hist[kSsMaxDig];
  const u32 tid = threadIdx.x;
  const u32 g = blockIdx.x % kSsGroups, idx = blockIdx.x / kSsGroups;
  const u32 t0 = g * cpx + idx * tpb;
  const u32 t1 = min(min(t0 + tpb, (g + 1) * cpx), ntiles);
  if (idx * tpb >= cpx || t0 >= t1) return;
  const u32 steps = ss_steps(nb1 - 1);
  ss_stage(spl, coarse, nb1 - 1, steps);
  hist[tid] = 0;
  __syncthreads();
  const u32 begin = t0 * tile, end = min(n, t1 * tile);
  for (u32 i = begin + tid; i < end; i += 4 * kSsNT) {
    Rec16 r[4];
    SsVal v[4];
    u32 d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { r[k] = ss_window_at<Sym, W>(S, min(i + (u32)k * kSsNT, end - 1u), sb); v[k] = ss_val(r[k]); }
    ss_count_le_multi<4>(spl, steps, v, d);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 x = i + (u32)k * kSsNT;
      if (x < end) { recs[x] = r[k]; atomicAdd(&hist[d[k]], 1u); dig[x] = (uint16_t)d[k]; }
    }
  }
  __syncthreads();
  if (tid < nb1 && hist[tid]) atomicAdd(&cntg[tid * kSsGroups + g], hist[tid]);
}

// k_msd_plan1 with the tile sizes as arguments (the record types have different tiles).
__global__ __launch_bounds__(1024) void k_ss_plan1(const u32 *__restrict__ cntg, u32 nb1, u32 n, u32 tile, u32 htile,
                                                  u32 *__restrict__ startg, u32 *__restrict__ cur1, u32 *__restrict__ bstart,
                                                  u32 *__restrict__ tpre, u32 *__restrict__ tpreh, u32 *__restrict__ plan) {
  __shared__ u32 tmp[16];
  const u32 d = threadIdx.x;
  u32 v[kSsGroups], c = 0;
#pragma unroll
  for (u32 g = 0; g < kSsGroups; g++) { v[g] = d < nb1 ? cntg[d * kSsGroups + g] : 0u; c += v[g]; }
  u32 tot;
  u32 ex = block_excl_scan<16>(c, tmp, tot);
  if (d < nb1) {
    bstart[d] = ex;
#pragma unroll
    for (u32 g = 0; g < kSsGroups; g++) { startg[d * kSsGroups + g] = ex; cur1[g * nb1 + d] = ex; ex += v[g]; }
  }
  const u32 ext = block_excl_scan<16>((c + tile - 1) / tile, tmp, tot);
  if (d < nb1) tpre[d] = ext;
  if (d == 0) { tpre[nb1] = tot; plan[kMsdW_T2] = tot; plan[kMsdW_CPX2] = max(1u, (tot + kSsGroups - 1) / kSsGroups); }
  const u32 exh = block_excl_scan<16>((c + htile - 1) / htile, tmp, tot);
  if (d < nb1) tpreh[d] = exh;
  if (d == 0) { tpreh[nb1] = tot; bstart[nb1] = n; startg[nb1 * kSsGroups] = n; }
}

// a record in registers as one vector (the fourth word of a Rec12 is unused)
__device__ __forceinline__ u32x4 ss_ld(const Rec16 *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ void ss_st(Rec16 *p, const u32x4 &v) { *reinterpret_cast<u32x4 *>(p) = v; }
__device__ __forceinline__ u32x4 ss_ld(const Rec12 *p) { u32x4 v; v.x = p->k0; v.y = p->k1; v.z = p->pos; v.w = 0u; return v; }
__device__ __forceinline__ void ss_st(Rec12 *p, const u32x4 &v) { p->k0 = v.x; p->k1 = v.y; p->pos = v.z; }

// One partition pass (cf. k_msd_part): block j belongs to group g = j % 8 and works that group's tile number j / 8.
// The digit of record i is dig[i], left there by the counting kernel that sized the buckets (k_ss_count1 / k_ss_hist2):
// the splitter search is done once per pass pair, 2 bytes per record carry it over.
// kSeg = false (pass 1): tile t = records [t * tile, ...), digit = coarse bucket, cursors[g * nb1 + digit].
// kSeg = true (pass 2): the tiles are those of the bucket list (tpre / bstart), digit = sub-bucket inside bucket b,
//   cursors[g * gstride + b * F2 + digit].
// Not stable.
template <class Rec, bool kSeg>
__global__ __launch_bounds__(kSsNT, 8) void k_ss_part(const Rec *__restrict__ in, Rec *__restrict__ out, u32 n, const uint16_t *__restrict__ dig,
                                                  u32 F2, u32 cpx, u32 ntiles, const u32 *__restrict__ tpre,
                                                  const u32 *__restrict__ bstart, u32 nb1, const u32 *__restrict__ plan,
                                                  u32 *__restrict__ cursors, u32 gstride) {
  constexpr int NT = kSsNT, IPT = SsCfg<Rec>::IPT, T = NT * IPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *srec = reinterpret_cast<Rec *>(smem);
  u32 *hist = reinterpret_cast<u32 *>(smem + sizeof(Rec) * T);                               // counts -> tile-exclusive prefix
  u32 *gbase = hist + kSsMaxDig;                                                              // global start of the tile's run
  u32 *tmp = gbase + kSsMaxDig;
  uint16_t *sdig = reinterpret_cast<uint16_t *>(tmp + 64);                                    // digit of the record in srec[]
  const u32 tid = threadIdx.x;
  const u32 g = blockIdx.x % kSsGroups, idx = blockIdx.x / kSsGroups;
  u32 begin, end, ndig;
  u32 *cur = cursors + (size_t)g * gstride;
  if (kSeg) {
    const u32 cpx2 = plan[kMsdW_CPX2], t2 = plan[kMsdW_T2];
    const u32 tile = g * cpx2 + idx;
    if (idx >= cpx2 || tile >= t2) return;
    const u32 b = msd_find_bucket(tpre, nb1, tile);
    begin = bstart[b] + (tile - tpre[b]) * (u32)T;
    end = min(begin + (u32)T, bstart[b + 1]);
    // a bucket's records are counted for ONE group, that of its first tile (k_ss_hist2): the few tiles of a bucket that
    // reach into the next group's tile range still reserve in the first group's regions
    cur = cursors + (size_t)(tpre[b] / cpx2) * gstride + (size_t)b * F2;
    ndig = F2;
  } else {
    const u32 tile = g * cpx + idx;
    if (idx >= cpx || tile >= ntiles) return;
    begin = tile * (u32)T;
    end = min(n, begin + (u32)T);
    ndig = nb1;
  }
  const u32 nvalid = end - begin;                  // >= 1
  hist[tid] = 0;
  __syncthreads();
  // The digits first, the records after the ranks are known: with the six 16-byte records held across the ranking and
  // the scan the kernel did not fit 64 VGPRs and hipcc kept them in scratch — 112 bytes per thread written and read
  // back, as much memory traffic again as the tile itself (profiles/r05a: WRITE_SIZE 2.1 x the records).
  u32 pk[IPT];                                     // digit | rank inside the tile's run << 16   (tile < 65536 records)
  // (clamped, not guarded: all loads in flight at once)
#pragma unroll
  for (int k = 0; k < IPT; k++) pk[k] = dig[begin + min((u32)(k * NT) + tid, nvalid - 1u)];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < nvalid) pk[k] |= atomicAdd(&hist[pk[k]], 1u) << 16;
  }
  __syncthreads();
  u32 cnt = 0;
  if (tid < ndig) {
    cnt = hist[tid];
    if (cnt) gbase[tid] = atomicAdd(&cur[tid], cnt);
  }
  // (held as vectors: hipcc left an array of record structs in scratch whatever the register count was)
  u32x4 r[IPT];
#pragma unroll
  for (int k = 0; k < IPT; k++) r[k] = ss_ld(in + begin + min((u32)(k * NT) + tid, nvalid - 1u));
  u32 tot;
  const u32 ex = block_excl_scan<NT / 64>(cnt, tmp, tot);
  hist[tid] = ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < nvalid) { const u32 dd = pk[k] & 0xffffu, q = hist[dd] + (pk[k] >> 16); ss_st(srec + q, r[k]); sdig[q] = (uint16_t)dd; }
  }
  __syncthreads();
  for (u32 q = tid; q < nvalid; q += NT) {
    const u32 dd = sdig[q];
    out[gbase[dd] + (q - hist[dd])] = srec[q];
  }
}

// Sizes of the sub-buckets per group (cf. k_msd_hist2): block h counts the digits of its piece (kSsHistTiles pass-2
// tiles) of bucket b and adds them to cnt2g[(b * F2 + digit) * 8 + g], g = the group of the bucket's FIRST pass-2 tile.
// (No barrier inside the tile loop, on purpose: the first version flushed the LDS histogram whenever a bucket's tiles
// crossed into the next group's tile range, and hipcc emitted that in-loop barrier without the lgkmcnt(0) wait for the
// ds_add_u32 pending around the loop's back-edge — counts off by a few records; tools/isa_barrier_scan.py
// checks every barrier of every kernel for this.)
template <class Rec>
__global__ __launch_bounds__(kSsNT, 8) void k_ss_hist2(const Rec *__restrict__ in, const SsVal *__restrict__ fine, u32 F2, u32 tile,
                                                   const u32 *__restrict__ tpre, const u32 *__restrict__ tpreh,
                                                   const u32 *__restrict__ bstart, u32 nb1, const u32 *__restrict__ plan,
                                                   u32 *__restrict__ cnt2g, uint16_t *__restrict__ dig) {
  __shared__ SsVal spl[kSsMaxDig];
  __shared__ u32 hist[kSsMaxDig];
  if (blockIdx.x >= tpreh[nb1]) return;
  const u32 tid = threadIdx.x;
  const u32 cpx2 = plan[kMsdW_CPX2];
  const u32 b = msd_find_bucket(tpreh, nb1, blockIdx.x);
  const u32 hh = blockIdx.x - tpreh[b];
  const u32 htile = tile * kSsHistTiles;
  const u32 begin = bstart[b] + hh * htile;
  const u32 end = min(begin + htile, bstart[b + 1]);
  const u32 gcur = tpre[b] / cpx2;                  // the whole bucket belongs to the group of its first tile
  const u32 steps = ss_steps(F2 - 1);
  ss_stage(spl, fine + (size_t)b * F2, F2 - 1, steps);
  hist[tid] = 0;
  __syncthreads();
  for (u32 pt = 0; pt < kSsHistTiles; pt++) {
    const u32 pb = begin + pt * tile;
    if (pb >= end) break;
    const u32 pe = min(pb + tile, end);
    for (u32 i = pb + tid; i < pe; i += 4 * kSsNT) {
      SsVal v[4];
      u32 d[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ss_val(in[min(i + (u32)k * kSsNT, pe - 1u)]);
      ss_count_le_multi<4>(spl, steps, v, d);
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i + (u32)k * kSsNT < pe) { atomicAdd(&hist[d[k]], 1u); dig[i + (u32)k * kSsNT] = (uint16_t)d[k]; }
    }
  }
  __syncthreads();
  if (tid < F2 && hist[tid]) atomicAdd(&cnt2g[((size_t)b * F2 + tid) * kSsGroups + gcur], hist[tid]);
}

// (debug, DC3HIP_SSORT_VERIFY=1) order-independent checksum of a record array: out[0] += sum of mixes, out[1] += count of
// records whose successor is smaller (0 for a sorted array)
template <class Rec>
__global__ __launch_bounds__(kBlock) void k_ss_verify(const Rec *__restrict__ p, u32 n, unsigned long long *out) {
  unsigned long long s = 0, bad = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const SsVal v = ss_val(p[i]);
    unsigned long long x = v.hi * 0x9E3779B97F4A7C15ull ^ (v.lo + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29;
    s += x;
    if (i + 1 < n && ss_lt(ss_val(p[i + 1]), v)) bad++;
  }
  atomicAdd(&out[0], s);
  if (bad) atomicAdd(&out[1], bad);
}

// (debug) after pass 2 every cursor must stand at the start of the next region: out[0] = mismatching (sub, group)
// regions, out[1] = the first such index (sub * 8 + group), out[2] = its cursor, out[3] = the expected value
__global__ __launch_bounds__(kBlock) void k_ss_verify_cursors(const u32 *__restrict__ start, const u32 *__restrict__ cur2, u32 n2,
                                                             unsigned long long *out) {
  const u32 i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n2 * kSsGroups) return;
  const u32 s = i / kSsGroups, g = i % kSsGroups;
  const u32 want = start[i + 1], got = cur2[(size_t)g * n2 + s];
  if (want != got) {
    if (atomicAdd(&out[0], 1ull) == 0) { out[1] = i; out[2] = got; out[3] = want; }
  }
}

// Pass 3: block s orders sub-bucket s = records [start[8 s], start[8 (s + 1)]) (at most NT * IPT of them; larger ones
// were refused on the host) and writes it to out at the same indices, ascending by (key, pos).
//   1. the records go to LDS in arrival order; ns = 127 (255 above 2048 records) of them, spread over the arrival order,
//      are ranked against each other — all pairs, the block's threads sharing the pairs — and become the bin boundaries
//   2. every record finds its bin (the thread's IPT searches in lockstep) and its arrival number in the bin
//   3. the bins are laid out in LDS; every record counts the smaller records of its bin: index = bin start + count
template <class Rec, int NT, int IPT>
__global__ __launch_bounds__(NT) void k_ss_local(const Rec *__restrict__ in, const u32 *__restrict__ start, Rec *__restrict__ out) {
  constexpr int CAP = NT * IPT, NSMAX = 256;
  static_assert(NT >= NSMAX && NT % NSMAX == 0, "threads share the candidate pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *A = reinterpret_cast<Rec *>(smem);                                   // CAP records
  uint8_t *binid = reinterpret_cast<uint8_t *>(smem + sizeof(Rec) * (CAP + kSsLocPad));   // bin of the record placed at A[q]
  __shared__ SsVal cand[NSMAX], spl[NSMAX];
  __shared__ u32 crank[NSMAX], cnt[NSMAX], cex[NSMAX + 1];
  __shared__ u32 tmp[NT / 64];
  const u32 tid = threadIdx.x;
  const u32 begin = start[(size_t)blockIdx.x * kSsGroups], end = start[(size_t)(blockIdx.x + 1) * kSsGroups];
  const u32 m = end - begin;
  if (m == 0) return;
  Rec r[IPT];
  // (clamped, not guarded: all loads in flight at once)
#pragma unroll
  for (int k = 0; k < IPT; k++) r[k] = in[begin + min((u32)(k * NT) + tid, m - 1u)];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < m) A[t] = r[k];
  }
  if (tid < NSMAX) { cnt[tid] = 0; crank[tid] = 0; }
  __syncthreads();
  if (m <= (u32)NSMAX) {                    // small: every record counts the smaller ones directly
    if (tid < m) {
      const SsVal v = ss_val(r[0]);
      u32 less = 0;
      for (u32 j = 0; j < m; j++) ss_acc_lt(less, ss_val(A[j]), v);
      out[begin + less] = r[0];
    }
    return;
  }
  const u32 steps = m > 2048u ? 8u : 7u, ns = (1u << steps) - 1u;          // 255 / 127 boundaries
  if (tid < ns) cand[tid] = ss_val(A[(u32)(((u64)tid * m) / ns)]);
  __syncthreads();
  {
    // all pairs: thread t holds two candidates, c and c + (ns + 1) / 2, and compares them with its share of all of them
    // (two per LDS read)
    const u32 hs = steps - 1u, hn = (ns + 1u) >> 1, ci = tid & (hn - 1u), part = tid >> hs, parts = (u32)NT >> hs, span = (ns + parts - 1u) / parts;
    const SsVal va = cand[ci], vb = cand[min(ci + hn, ns - 1u)];
    const u32 j0 = min(ns, part * span), j1 = min(ns, j0 + span);
    u32 la = 0, lb = 0;
    for (u32 j = j0; j < j1; j++) { const SsVal w = cand[j]; ss_acc_lt(la, w, va); ss_acc_lt(lb, w, vb); }
    if (la) atomicAdd(&crank[ci], la);
    if (lb && ci + hn < ns) atomicAdd(&crank[ci + hn], lb);
  }
  __syncthreads();
  if (tid < ns) spl[ss_tree_slot(crank[tid], steps)] = cand[tid];
  __syncthreads();
  u32 bin[IPT], rk[IPT];
  {
    SsVal v[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) v[k] = ss_val(r[k]);
    // 0 .. ns; a wave searches only for the rounds of the load in which it holds records (a sub-bucket has about 1400 of
    // the 4096 the block has room for: the searches are LDS reads at scattered addresses, what this kernel is bound by)
    static_assert(IPT == 4, "rounds of the search");
#pragma unroll
    for (int k = 0; k < IPT; k++) bin[k] = 0;
    const u32 w0 = tid & ~63u;                              // the wave's first record of round 0
    if (w0 + 3u * NT < m) ss_count_le_n<4>(spl, steps, v, bin);
    else if (w0 + 2u * NT < m) ss_count_le_n<3>(spl, steps, v, bin);
    else if (w0 + (u32)NT < m) ss_count_le_n<2>(spl, steps, v, bin);
    else if (w0 < m) ss_count_le_n<1>(spl, steps, v, bin);
  }
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    rk[k] = t < m ? atomicAdd(&cnt[bin[k]], 1u) : 0u;
  }
  __syncthreads();
  {
    const u32 c = tid < NSMAX ? cnt[tid] : 0u;                             // bins 0 .. ns <= NSMAX - 1
    u32 tot;
    const u32 ex = block_excl_scan<NT / 64>(c, tmp, tot);
    if (tid < NSMAX) cex[tid] = ex;
    if (tid == NSMAX - 1) cex[NSMAX] = tot;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < m) { const u32 q = cex[bin[k]] + rk[k]; A[q] = r[k]; binid[q] = (uint8_t)bin[k]; }
  }
  if (tid < kSsLocPad) A[m + tid] = ss_max_rec<Rec>();
  __syncthreads();
  for (u32 q = tid; q < m; q += NT) {
    const Rec x = A[q];
    const SsVal v = ss_val(x);
    const u32 bb = binid[q];
    const u32 lo = cex[bb], hi = cex[bb + 1];
    u32 less = 0;
    // 4 independent reads per round (bins hold about 11 records: rounds of 8 would read 16 for them, rounds of 4 read 12),
    // and no bound inside the round: what follows the bin in LDS are the later bins — every record there is above the
    // bin's upper boundary, so "smaller than mine" is false for it — and behind the last bin kSsLocPad records of all ones
    for (u32 j0 = lo; j0 < hi; j0 += 4) {
      SsVal w[4];
#pragma unroll
      for (int i = 0; i < 4; i++) w[i] = ss_val(A[j0 + (u32)i]);
#pragma unroll
      for (int i = 0; i < 4; i++) ss_acc_lt(less, w[i], v);
    }
    out[begin + lo + less] = x;
  }
}

}  // namespace dc3
