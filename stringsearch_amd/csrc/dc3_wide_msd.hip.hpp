// dc3_wide_msd.hip.hpp — bucket (MSD) ordering for texts of 2^32 bytes and more (64-bit positions), global mode.
// Part of the gfx950 kernel set of libdc3hip; namespace dc3.  Included after dc3_wide.hip.hpp and dc3_msd.hip.hpp.
//
// The wide whole-text order (dc3_wide.hip.hpp) sorted 16-byte records {image <= 63 bits, position 40 bits}
// with five stable LSD passes — 4x the bytes per position of the 8-byte bucket ordering that builds texts below 2^32
// (crates/dc3/src/lib.rs:44-57 has no width limit: everything is usize).  An image of log2 n + 8 bits and a position of
// log2 n bits do not fit one 64-bit word, but they do not have to:
//   * a rank sorts only the images of ITS range [lo, hi).  x' = floor((image - lo) * 2^E / (hi - lo)) maps the range onto
//     [0, 2^E) monotonically (E <= bits of the span), so digits are plain bit fields and every bucket is used;
//   * partition pass 1 computes x' of every text position ON THE FLY (no record is read) and leaves each selected position
//     in the bucket of the top d1 bits of x' — which the stored word then need not repeat:
//         word = (x' mod 2^(E - d1)) << pb | position,      pb = bits of n - 1,  E - d1 + pb <= 64.
// From there on the words are ordinary bucket-ordering words: k_msd_hist2 / k_msd_part<true> / k_msd_local order every
// bucket by the remaining image bits (then by position), and k_wide_ties8 settles equal images by comparing windows.
// Non-injective maps (two images with one x') and words that agree across a bucket boundary only create extra tie
// groups; the window compare orders those consistently with the image.
//
// Pass 1 selects: a block walks its chunk of TEXT positions, appends the selected words to an LDS stage and partitions the
// stage whenever it holds a full tile of 8192 words — so the runs a tile writes per bucket are as long as those of a dense
// pass although only one position in P is selected.
#pragma once

namespace dc3 {

// this rank's range and the map onto [0, 2^E): x' = umulhi((image - lo) << (64 - eb), M) >> (63 - E), M = floor(2^(63+eb) / span)
struct WideRange {
  u64 lo, hi, M;          // images in [lo, hi) (hi ignored when last)
  u32 last, eb, E, d1, pb;
};
__device__ __forceinline__ bool wide_in_range(const WideRange &r, u64 img) { return img >= r.lo && (r.last || img < r.hi); }
__device__ __forceinline__ u64 wide_xprime(const WideRange &r, u64 img) {
  return __umul64hi((img - r.lo) << (64 - r.eb), r.M) >> (63 - r.E);
}

// images of the four positions p0 .. p0 + 3 (p0 % 4 == 0) by rolling, as k_wide_select does.  JMAX >= k.J bounds the
// unrolled loops at compile time (bytes need 5-8 image symbols, DNA about 20, a binary alphabet up to 48): with the one
// bound of 48 every position paid 51 predicated iterations — 4.6 ms per GiB and pass against 1.7 for Key9's pack kernel.
// the text words behind position p0 that the images of p0 .. p0 + 3 are made of (loaded apart from the arithmetic so that a
// kernel can fetch the next round's words before it works on this round's)
template <u32 JMAX>
struct WideWords { u32 w[(JMAX + 3 + 3) / 4]; };
template <u32 JMAX>
__device__ __forceinline__ void wide_load4(const WideKey &k, u64 p0, WideWords<JMAX> &ww) {
  const u32 nw = (k.J + 3 + 3) / 4;
  const u32 *tw = reinterpret_cast<const u32 *>(k.t + p0);
#pragma unroll
  for (u32 i = 0; i < (JMAX + 3 + 3) / 4; i++) ww.w[i] = i < nw ? tw[i] : 0u;
}
template <u32 JMAX, bool kPow2>
__device__ __forceinline__ void wide_images4(const WideKey &k, u64 p0, const WideWords<JMAX> &ww, const uint16_t *lcode, u64 (&img)[4]) {
  const u32 J = k.J, sigma = k.sigma;
  const u32 (&w)[(JMAX + 3 + 3) / 4] = ww.w;
  u64 v = 0;
  u32 dh[3] = {0, 0, 0}, dt0 = 0, dt1 = 0, dt2 = 0;
#pragma unroll
  for (u32 s = 0; s < JMAX + 3; s++) {
    if (s < J + 3) {
      u32 q = (p0 + s < k.n) ? (u32)lcode[(w[s >> 2] >> (8 * (s & 3u))) & 255u] : 0u;
      q = q ? q - 1 : 0u;
      if (s < 3) dh[s] = q;
      if (s < J) v = kPow2 ? ((v << k.lg) | q) : v * sigma + q;
      else if (s == J) dt0 = q;
      else if (s == J + 1) dt1 = q;
      else dt2 = q;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    img[j] = kPow2 ? (v >> k.sh) : __umul64hi(v, k.mfix);
    const u64 nd = j == 0 ? dt0 : j == 1 ? dt1 : dt2;
    if (j < 3) v = kPow2 ? (((v & (k.P1 - 1ull)) << k.lg) | nd) : (v - (u64)dh[j] * k.P1) * sigma + nd;
  }
}
// the smallest of the compiled bounds that holds J image symbols
inline u32 wide_jmax(u32 J) { return J <= 8 ? 8u : J <= 16 ? 16u : J <= 24 ? 24u : kWideMaxImageSyms; }

constexpr int kWideNT = 1024;                                  // threads of a block of the two kernels below
constexpr u32 kWideRound = 4u * kWideNT;                       // positions per round (4 consecutive ones per thread)
constexpr u32 kWideTile = 8192;                                // words per partition tile
constexpr u32 kWideStage = kWideTile + kWideRound;             // capacity of the LDS stage
constexpr size_t kWidePartSmem = sizeof(u64) * kWideStage + sizeof(uint16_t) * kWideStage + sizeof(u32) * (2 * 1024 + 64);

// Counting pass: table[d * nchunks + c] = selected positions of chunk c (positions [c * chunk, (c + 1) * chunk), chunk a
// multiple of kWideRound) whose x' has top-d1 digit d — the digit table format of the pack kernels (k_msd_cnt1 sums it
// per XCD group: chunk c belongs to group c / cpg).
template <u32 JMAX, bool kPow2>
__global__ __launch_bounds__(kWideNT) void k_wide_count1(WideKey k, WideRange rg, u64 chunk, u32 nchunks, u32 *__restrict__ table) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[4][1024];
  const u32 tid = threadIdx.x;
  if (tid < 256) lcode[tid] = k.code[tid];
  for (u32 j = tid; j < 4 * 1024; j += kWideNT) (&hist[0][0])[j] = 0;
  __syncthreads();
  u32 *myh = hist[(tid >> 6) & 3u];
  const u64 begin = (u64)blockIdx.x * chunk, end = min(k.n, begin + chunk);
  const u32 sh = rg.E - rg.d1;
  for (u64 p0 = begin + 4ull * tid; p0 < end; p0 += kWideRound) {
    u64 img[4];
    WideWords<JMAX> ww;
    wide_load4<JMAX>(k, p0, ww);
    wide_images4<JMAX, kPow2>(k, p0, ww, lcode, img);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (p0 + j < end && wide_in_range(rg, img[j])) atomicAdd(&myh[(u32)(wide_xprime(rg, img[j]) >> sh)], 1u);
  }
  __syncthreads();
  const u32 ndig = 1u << rg.d1;
  if (tid < ndig) table[(size_t)tid * nchunks + blockIdx.x] = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
}

// Partition pass 1 with selection (see the header).  Block j belongs to group j % 8 and works that group's chunk number
// j / 8 (chunk c = g * cpg + idx); cursors[g * ndig + d] = the group's cursor of bucket d (k_msd_plan1).  Output words in
// the memory form of the bucket ordering (msd_word).  Not stable.
template <u32 JMAX, bool kPow2>
__global__ __launch_bounds__(kWideNT) void k_wide_part1(WideKey k, WideRange rg, u64 chunk, u32 nchunks, u32 cpg,
                                                       u32 *__restrict__ cursors, u64 *__restrict__ out, u32 *__restrict__ xcdmon) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *stage = reinterpret_cast<u64 *>(smem);
  uint16_t *sdig = reinterpret_cast<uint16_t *>(smem + sizeof(u64) * kWideStage);
  u32 *hist = reinterpret_cast<u32 *>(smem + sizeof(u64) * kWideStage + sizeof(uint16_t) * kWideStage);   // [1024]
  u32 *gbase = hist + 1024;                                                                                // [1024]
  u32 *tmp = gbase + 1024;                                                                                 // [16] + cursor
  __shared__ uint16_t lcode[256];
  const u32 tid = threadIdx.x;
  const u32 g = blockIdx.x % kMsdGroups, idx = blockIdx.x / kMsdGroups;
  const u32 c = g * cpg + idx;
  if (idx >= cpg || c >= nchunks) return;
  xcd_note(xcdmon, g);
  const u32 ndig = 1u << rg.d1, sh = rg.E - rg.d1;
  const u64 wmask = sh >= 64 ? ~0ull : ((1ull << sh) - 1ull);
  u32 *cur = cursors + (size_t)g * ndig;
  u32 *fill = tmp + 32;                       // words in the stage
  if (tid < 256) lcode[tid] = k.code[tid];
  if (tid == 0) *fill = 0;
  __syncthreads();
  const u64 begin = (u64)c * chunk, end = min(k.n, begin + chunk);

  // partition the first `nv` words of the stage (nv <= kWideTile) and move the rest to its front
  auto flush = [&](u32 nv, u32 have) {
    constexpr int IPT = kWideTile / kWideNT;
    u64 r[IPT]; u32 d[IPT], rk[IPT];
    u64 lw[kWideRound / kWideNT]; u32 ld[kWideRound / kWideNT];
#pragma unroll
    for (int q = 0; q < IPT; q++) { const u32 t = q * kWideNT + tid; r[q] = stage[min(t, nv - 1u)]; d[q] = sdig[min(t, nv - 1u)]; }
    const u32 left = have - nv;               // < kWideRound
#pragma unroll
    for (int q = 0; q < (int)(kWideRound / kWideNT); q++) {
      const u32 t = q * kWideNT + tid;
      if (t < left) { lw[q] = stage[nv + t]; ld[q] = sdig[nv + t]; }
    }
    hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IPT; q++) { const u32 t = q * kWideNT + tid; if (t < nv) rk[q] = atomicAdd(&hist[d[q]], 1u); }
    __syncthreads();
    u32 cnt = 0;
    if (tid < ndig) { cnt = hist[tid]; if (cnt) gbase[tid] = atomicAdd(&cur[tid], cnt); }
    u32 tot;
    const u32 ex = block_excl_scan<kWideNT / 64>(cnt, tmp, tot);
    hist[tid] = ex;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IPT; q++) {
      const u32 t = q * kWideNT + tid;
      if (t < nv) { const u32 at = hist[d[q]] + rk[q]; stage[at] = r[q]; sdig[at] = (uint16_t)d[q]; }
    }
    __syncthreads();
    for (u32 q = tid; q < nv; q += kWideNT) {
      const u32 dd = sdig[q];
      out[gbase[dd] + (q - hist[dd])] = msd_word(stage[q]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (int)(kWideRound / kWideNT); q++) {
      const u32 t = q * kWideNT + tid;
      if (t < left) { stage[t] = lw[q]; sdig[t] = (uint16_t)ld[q]; }
    }
    if (tid == 0) *fill = left;
    __syncthreads();
  };

  // (one block of 1024 threads fills a CU: the next round's text words are fetched before this round is worked, or every
  //  round would wait out a full memory latency with nothing else to run)
  WideWords<JMAX> wnext;
  if (begin + 4ull * tid < end) wide_load4<JMAX>(k, begin + 4ull * tid, wnext);
  for (u64 r0 = begin; r0 < end; r0 += kWideRound) {
    const u64 p0 = r0 + 4ull * tid;
    const WideWords<JMAX> wcur = wnext;
    if (p0 + kWideRound < end) wide_load4<JMAX>(k, p0 + kWideRound, wnext);
    u64 img[4] = {0, 0, 0, 0};
    if (p0 < end) wide_images4<JMAX, kPow2>(k, p0, wcur, lcode, img);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool sel = p0 + j < end && wide_in_range(rg, img[j]);
      const u64 m = __ballot(sel);
      if (m) {
        u32 wbase = 0;
        if (lane_id() == 0) wbase = atomicAdd(fill, (u32)__popcll(m));
        wbase = __shfl(wbase, 0);
        if (sel) {
          const u64 x = wide_xprime(rg, img[j]);
          const u32 at = wbase + mbcnt(m);
          stage[at] = ((x & wmask) << rg.pb) | (p0 + j);
          sdig[at] = (uint16_t)(x >> sh);
        }
      }
    }
    __syncthreads();
    const u32 have = *fill;                   // (block-uniform)
    __syncthreads();
    if (have >= kWideTile) flush(kWideTile, have);
  }
  const u32 have = *fill;
  if (have) flush(have, have);
}

// Tie pass over a rank's words in ascending order (x' rest, position): shard[i] = position of the i-th smallest window.
// Words whose image part agrees with a neighbour's are ordered by their windows, one thread per group — k_wide_ties on
// 8-byte words.  words[0] = a group larger than kWideTieBig, words[1] += tied words, words[2] += windows equal within k.W.
// OutT: 64-bit positions (wide contexts) or 32-bit ones (the same order for texts below 2^32, whose slices are u32).
// same[i] = 1 iff word i has the image part of word i - 1 (the byte the local sort leaves, MsdRecSameSink): the scan reads
// one word and two bytes per entry instead of three words.
template <class OutT>
__global__ __launch_bounds__(kBlock) void k_wide_ties8(const u64 *__restrict__ h, const uint8_t *__restrict__ same, u32 nrec, u32 pb, WideKey k,
                                                      OutT *__restrict__ shard, u32 *words) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  const u64 pmask = (1ull << pb) - 1ull;
  u32 tied = 0, dup = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nrec; i += gridDim.x * kBlock) {
    const bool eqp = same[i] != 0;
    const bool eqn = i + 1 < nrec && same[i + 1] != 0;
    if (!eqp && !eqn) { shard[i] = (OutT)(msd_word(h[i]) & pmask); continue; }
    tied++;
    if (eqp) continue;                                   // the group's first thread does the work
    u32 e = i + 2;
    while (e < nrec && e - i <= kWideTieMax && same[e] != 0) e++;
    const u32 len = e - i;
    if (len > kWideTieMax) {
      while (e < nrec && e - i <= kWideTieBig && same[e] != 0) e++;
      const u32 big = e - i;
      if (big > kWideTieBig) { words[0] = 1u; continue; }
      for (u32 x = 0; x < big; x++) {
        const u64 v = msd_word(h[i + x]) & pmask;
        u32 y = x;
        while (y > 0) {
          const u64 prev = shard[i + y - 1];
          const int c = wide_cmp(k, v, prev, k.W, lcode);
          if (c == 0) dup++;
          if (c >= 0) break;
          shard[i + y] = (OutT)prev; y--;
        }
        shard[i + y] = (OutT)v;
      }
      continue;
    }
    u64 loc[kWideTieMax];
    for (u32 x = 0; x < len; x++) {
      const u64 v = msd_word(h[i + x]) & pmask;
      u32 y = x;
      while (y > 0) {
        const int c = wide_cmp(k, v, loc[y - 1], k.W, lcode);
        if (c == 0) dup++;
        if (c >= 0) break;
        loc[y] = loc[y - 1]; y--;
      }
      loc[y] = v;
    }
    for (u32 x = 0; x < len; x++) shard[i + x] = (OutT)loc[x];
  }
  tied = wave_reduce(tied); dup = wave_reduce(dup);
  if (lane_id() == 0) { if (tied) atomicAdd(&words[1], tied); if (dup) atomicAdd(&words[2], dup); }
}

}  // namespace dc3
