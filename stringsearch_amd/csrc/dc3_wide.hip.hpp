// dc3hip — texts of 2^32 bytes and more in the global mode: positions no longer fit the 32-bit fields every record of
// the recursion uses, so only the distributed WHOLE-TEXT ORDER exists at that size: every rank orders
// the positions whose window image falls into its range — 16-byte records {image (<= 63 bits), position (40 bits)} —
// and the text's windows must all be distinct (high-entropy inputs: BASELINE.json configs[3] random bytes at 4 GiB,
// configs[4] random DNA at 16 GiB).  A text whose windows repeat is refused (no recursion with 64-bit positions).
//
// Window = W symbols of the text (W = kWideWindow = 256, or the whole rest of the text if shorter); its order is decided lazily, a
// 4-byte word at a time (wide_cmp).  Sort image = the first J symbols in base sigma (digit = code - 1, past the end = 0)
// scaled to `ibits` bits, as KeyT's (dc3_order.hip.hpp).
#pragma once
#include "dc3_common.hip.hpp"

namespace dc3 {

struct WideKey {
  const uint8_t *t; const uint16_t *code; u64 n;
  u32 sigma, J, W;          // alphabet size, image symbols, compare depth in symbols (multiple of 4)
  u64 mfix, P1;             // floor((2^(64+ibits) - 1) / sigma^J), sigma^(J-1)
  // sigma a power of two (bytes: 256, DNA: 4 — both of BASELINE's input classes): lg = log2 sigma and the image is
  // v >> sh exactly (mfix = 2^(64 - sh)), so the kernels that evaluate every position of the text can shift and mask
  // instead of multiplying 64-bit numbers (v_mul_*_u32 run at quarter rate: 12 ms per 4.3 G positions and pass).  lg = 0: general.
  u32 lg = 0, sh = 0;
};
constexpr u32 kWideMaxImageSyms = 48;
constexpr u32 kWideWindow = 256;          // symbols compared before two positions count as having the same window ...
constexpr u32 kWideWindowDeep = 8192;     // ... and in the second attempt, made when only few windows agree on 256 symbols
constexpr u64 kWideMaxDepth = 1ull << 30; // further attempts go 16 times deeper each, up to this many symbols ...
constexpr u64 kWideTieBudget = 1ull << 37;// ... while (windows that still agree) x depth stays below this many symbol compares

__device__ __forceinline__ u64 wide_pos(const Rec16 &r) { return ((u64)r.k2 << 32) | r.pos; }
__device__ __forceinline__ u64 wide_img(const Rec16 &r) { return ((u64)r.k1 << 32) | r.k0; }

// image of one position (splitter sample)
__global__ __launch_bounds__(kBlock) void k_wide_sample(WideKey k, u64 stride, u32 ns, u64 *__restrict__ out) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < ns; i += gridDim.x * kBlock) {
    const u64 p = (u64)i * stride;
    u64 v = 0;
#pragma unroll 1
    for (u32 j = 0; j < k.J; j++) {
      u32 q = (p + j < k.n) ? (u32)lcode[k.t[p + j]] : 0u;
      q = q ? q - 1 : 0u;
      v = v * k.sigma + q;
    }
    out[i] = __umul64hi(v, k.mfix);
  }
}

// Selection of the positions whose image lies in [lo, hi) (hi ignored on the last rank): block b owns the positions
// [b * chunk, (b + 1) * chunk) (chunk % 4 == 0), four consecutive positions per thread, images by rolling as in
// k_pack_image_textT.  kWrite = false: counts[b] = selected positions of the block.  kWrite = true: the records go to
// out[bases[b] ..] in arbitrary order (the sort that follows decides).
template <bool kWrite>
__global__ __launch_bounds__(kBlock) void k_wide_select(WideKey k, u64 chunk, u64 lo, u64 hi, u32 last,
                                                       u32 *__restrict__ counts, const u32 *__restrict__ bases,
                                                       Rec16 *__restrict__ out) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 cursor;
  __shared__ u32 tmp[kWaves];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  if (threadIdx.x == 0) cursor = 0;
  __syncthreads();
  const u32 J = k.J, sigma = k.sigma, nw = (J + 3 + 3) / 4;
  constexpr u32 kW = (kWideMaxImageSyms + 3 + 3) / 4;
  const u64 begin = (u64)blockIdx.x * chunk, end = min(k.n, begin + chunk);
  const u32 base = kWrite ? bases[blockIdx.x] : 0u;
  u32 mine = 0;
  for (u64 p0 = begin + 4ull * threadIdx.x; p0 < end; p0 += 4ull * kBlock) {
    const u32 *tw = reinterpret_cast<const u32 *>(k.t + p0);
    u32 w[kW];
#pragma unroll
    for (u32 i = 0; i < kW; i++) w[i] = i < nw ? tw[i] : 0u;
    u64 v = 0;
    u32 dh[3] = {0, 0, 0}, dt0 = 0, dt1 = 0, dt2 = 0;
#pragma unroll
    for (u32 s = 0; s < kWideMaxImageSyms + 3; s++) {
      if (s < J + 3) {
        u32 q = (p0 + s < k.n) ? (u32)lcode[(w[s >> 2] >> (8 * (s & 3u))) & 255u] : 0u;
        q = q ? q - 1 : 0u;
        if (s < 3) dh[s] = q;
        if (s < J) v = v * sigma + q;
        else if (s == J) dt0 = q;
        else if (s == J + 1) dt1 = q;
        else dt2 = q;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 img = __umul64hi(v, k.mfix);
      const u64 p = p0 + j;
      if (p < end && img >= lo && (last || img < hi)) {
        if (kWrite) {
          const u32 slot = atomicAdd(&cursor, 1u);
          out[(size_t)base + slot] = Rec16{(u32)img, (u32)(img >> 32), (u32)(p >> 32), (u32)p};
        } else {
          mine++;
        }
      }
      if (j < 3) v = (v - (u64)dh[j] * k.P1) * sigma + (j == 0 ? dt0 : j == 1 ? dt1 : dt2);
    }
  }
  if (!kWrite) {
    mine = wave_reduce(mine);
    if (lane_id() == 0) tmp[wave_id()] = mine;
    __syncthreads();
    if (threadIdx.x == 0) { u32 tot = 0; for (int i = 0; i < kWaves; i++) tot += tmp[i]; counts[blockIdx.x] = tot; }
  }
}

// Order (< 0, 0, > 0) of the windows of positions p and q, up to `depth` symbols (multiple of 4): a word at a time, only a
// differing (or end-crossing) word is decoded; a suffix that ends sorts before its extensions.
__device__ __forceinline__ int wide_cmp(const WideKey &k, u64 p, u64 q, u32 depth, const uint16_t *lds) {
#pragma unroll 1
  for (u32 s = 0; s < depth; s += 4) {
    u32 wp, wq;
    __builtin_memcpy(&wp, k.t + p + s, 4);
    __builtin_memcpy(&wq, k.t + q + s, 4);
    if (wp == wq && p + s + 4 <= k.n && q + s + 4 <= k.n) continue;
#pragma unroll
    for (u32 b = 0; b < 4; b++) {
      const bool ep = p + s + b >= k.n, eq = q + s + b >= k.n;
      const u32 cp = ep ? 0u : (u32)lds[(wp >> (8 * b)) & 255u];
      const u32 cq = eq ? 0u : (u32)lds[(wq >> (8 * b)) & 255u];
      if (cp != cq) return cp < cq ? -1 : 1;
      if (ep) return 0;                      // both ended at once: only possible for p == q
    }
  }
  return 0;
}

// Tie pass over this rank's records in image order: shard[i] = position of the i-th smallest window.  Records whose image
// is shared (rare: the image has log2 n + 4 bits and more) are ordered by their windows, one thread per group.
// words[0] = a group larger than kWideTieBig, words[1] += tied records, words[2] += windows equal within k.W symbols.
constexpr u32 kWideTieMax = 16, kWideTieBig = 1024;
__global__ __launch_bounds__(kBlock) void k_wide_ties(const Rec16 *__restrict__ h, u32 nrec, WideKey k, u64 *__restrict__ shard,
                                                     u32 *words) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  u32 tied = 0, dup = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nrec; i += gridDim.x * kBlock) {
    const Rec16 r = h[i];
    const u64 a = wide_img(r);
    const bool eqp = i > 0 && wide_img(h[i - 1]) == a;
    const bool eqn = i + 1 < nrec && wide_img(h[i + 1]) == a;
    if (!eqp && !eqn) { shard[i] = wide_pos(r); continue; }
    tied++;
    if (eqp) continue;                                   // the group's first thread does the work
    u32 e = i + 2;
    while (e < nrec && e - i <= kWideTieMax && wide_img(h[e]) == a) e++;
    const u32 len = e - i;
    if (len > kWideTieMax) {
      // a larger group (a run of the smallest symbol, a short repeat): extend it to its end and insertion-sort it in
      // place in the shard; beyond kWideTieBig records the text is refused
      while (e < nrec && e - i <= kWideTieBig && wide_img(h[e]) == a) e++;
      const u32 big = e - i;
      if (big > kWideTieBig) { words[0] = 1u; continue; }
      for (u32 x = 0; x < big; x++) {
        const u64 v = wide_pos(h[i + x]);
        u32 y = x;
        while (y > 0) {
          const u64 prev = shard[i + y - 1];
          const int c = wide_cmp(k, v, prev, k.W, lcode);
          if (c == 0) dup++;
          if (c >= 0) break;
          shard[i + y] = prev; y--;
        }
        shard[i + y] = v;
      }
      continue;
    }
    u64 loc[kWideTieMax];
    for (u32 x = 0; x < len; x++) {
      const u64 v = wide_pos(h[i + x]);
      u32 y = x;
      while (y > 0) {
        const int c = wide_cmp(k, v, loc[y - 1], k.W, lcode);
        if (c == 0) dup++;
        if (c >= 0) break;
        loc[y] = loc[y - 1]; y--;
      }
      loc[y] = v;
    }
    for (u32 x = 0; x < len; x++) shard[i + x] = loc[x];
  }
  tied = wave_reduce(tied); dup = wave_reduce(dup);
  if (lane_id() == 0) { if (tied) atomicAdd(&words[1], tied); if (dup) atomicAdd(&words[2], dup); }
}

// Verifier of a shard (dc3hip_global_sufcheck): every position in range and every entry's suffix strictly smaller than
// its successor's (next_first = the first entry of the next rank's shard, or ~0 on the last rank) — compared as
// suffixes, up to `depth` symbols; equal within `depth` counts as an error.  Strict order implies distinct positions, so
// together with the shard sizes adding up to n this is exactly "the concatenated shards are the suffix array".
// err: 0 ok, 2 position out of range, 3 order violated (atomicMax).
__global__ __launch_bounds__(kBlock) void k_wide_check(const u64 *__restrict__ shard, u32 cnt, u64 next_first, WideKey k,
                                                      u32 depth, u32 *err) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < cnt; i += gridDim.x * kBlock) {
    const u64 p = shard[i];
    if (p >= k.n) { atomicMax(err, 2u); continue; }
    const u64 q = i + 1 < cnt ? shard[i + 1] : next_first;
    if (q == ~0ull) continue;
    if (q >= k.n) { atomicMax(err, 2u); continue; }
    if (wide_cmp(k, p, q, depth, lcode) >= 0) atomicMax(err, 3u);
  }
}

// ---------------------------------------------------------------------------------------------
// Deepening by rank look-ups (host: wide_deepen): the form of prefix doubling a range-partitioned order allows.  When
// windows still agree after the last symbol compare of depth D, every rank receives the whole order and builds
//     isa[p] = 1 + global index of the first entry whose D-symbol window equals that of p      (isa[n] = 0: the empty suffix)
// — the rank of p by its first D symbols.  Two suffixes that agree on D symbols are then ordered by the ranks of their
// continuations p + D, p + 2D, ..., p + W D: W + 1 look-ups settle (W + 1) D symbols, and the next round starts from
// ranks that deep.  Depth grows (W + 1)-fold per round whatever the text looks like; no symbol is compared again.
// ---------------------------------------------------------------------------------------------
// eq[i] = 1 iff entry i's window equals entry i - 1's within k.W symbols (eq[0] = 0: a rank's range begins with a new window)
__global__ __launch_bounds__(kBlock) void k_wide_eq(const u64 *__restrict__ shard, u32 nrec, WideKey k, uint8_t *__restrict__ eq) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nrec; i += gridDim.x * kBlock)
    eq[i] = (i > 0 && wide_cmp(k, shard[i - 1], shard[i], k.W, lcode) == 0) ? 1 : 0;
}
__device__ __forceinline__ int wide_cmp_isa(const u64 *__restrict__ isa, u64 n, u64 p, u64 q, u64 D, u32 W) {
#pragma unroll 1
  for (u32 j = 0; j <= W; j++) {
    const u64 pp = p + (u64)j * D, qq = q + (u64)j * D;
    const u64 a = pp < n ? isa[pp] : 0ull, b = qq < n ? isa[qq] : 0ull;
    if (a != b) return a < b ? -1 : 1;
    if (pp >= n) return 0;
  }
  return 0;
}
// One thread per group of entries that agree on D symbols (eq): the group is put in the order of (W + 1) D symbols, and
// neweq says which neighbours still agree that far.  words[0] = a group beyond kWideTieBig, words[2] += such neighbours.
__global__ __launch_bounds__(kBlock) void k_wide_ties_isa(u64 *__restrict__ shard, const uint8_t *__restrict__ eq, u32 nrec,
                                                         const u64 *__restrict__ isa, u64 n, u64 D, u32 W,
                                                         uint8_t *__restrict__ neweq, u32 *words) {
  u32 dup = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nrec; i += gridDim.x * kBlock) {
    if (eq[i]) continue;                                  // a member: its group's first thread does the work
    neweq[i] = 0;
    if (!(i + 1 < nrec && eq[i + 1])) continue;
    u32 e = i + 1;
    while (e < nrec && eq[e] && e - i <= kWideTieBig) e++;
    if (e - i > kWideTieBig) { words[0] = 1u; continue; }
    for (u32 x = i + 1; x < e; x++) {                     // binary insertion: log2 compares (W + 1 look-ups each) per entry
      const u64 v = shard[x];
      u32 lo = i, hi = x;                                 // first y in [i, x) whose entry is greater than v
      while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);             // (indices reach beyond 2^31: no lo + hi)
        if (wide_cmp_isa(isa, n, v, shard[mid], D, W) < 0) hi = mid; else lo = mid + 1;
      }
      for (u32 y = x; y > lo; y--) shard[y] = shard[y - 1];
      shard[lo] = v;
    }
    for (u32 x = i + 1; x < e; x++) {
      const bool same = wide_cmp_isa(isa, n, shard[x - 1], shard[x], D, W) == 0;
      neweq[x] = same ? 1 : 0;
      dup += same ? 1u : 0u;
    }
  }
  dup = wave_reduce(dup);
  if (lane_id() == 0 && dup) atomicAdd(&words[2], dup);
}
// The verifier of an order that was deepened (isa is then the exact inverse: all ranks distinct), linear in the shard:
// the entry's rank is its index, and neighbours p < q satisfy (T[p], rank of p + 1) < (T[q], rank of q + 1) — by induction
// over the ranks that is the suffix order.  err as k_wide_check (3 also when isa is not the inverse of the order).
__global__ __launch_bounds__(kBlock) void k_wide_check_isa(const u64 *__restrict__ shard, u32 cnt, u64 first, u64 next_first, WideKey k,
                                                          const u64 *__restrict__ isa, u32 *err) {
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = k.code[threadIdx.x];
  __syncthreads();
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < cnt; i += gridDim.x * kBlock) {
    const u64 p = shard[i];
    if (p >= k.n) { atomicMax(err, 2u); continue; }
    if (isa[p] != first + i + 1) { atomicMax(err, 3u); continue; }
    const u64 q = i + 1 < cnt ? shard[i + 1] : next_first;
    if (q == ~0ull) continue;
    if (q >= k.n) { atomicMax(err, 2u); continue; }
    const u32 cp = lcode[k.t[p]], cq = lcode[k.t[q]];
    const u64 rp = p + 1 < k.n ? isa[p + 1] : 0ull, rq = q + 1 < k.n ? isa[q + 1] : 0ull;
    if (cp > cq || (cp == cq && rp >= rq)) atomicMax(err, 3u);
  }
}

// ---------------------------------------------------------------------------------------------
// Groups of any size (round 5: the one-thread-per-group kernels above stop at kWideTieBig members and used to refuse the
// text: a run of one symbol, a short period).  A group = a maximal run of entries whose flag same[i] says "belongs to the
// group of entry i - 1" (same[0] = 0).
//   k_seg_last / k_seg_carry / k_seg_apply   start of every entry's group, in three streaming launches (no walk back)
//   k_big_count / k_big_write                 the members of groups beyond T entries, compacted in index order
//   k_seg_recs*                               16-byte records (one key component, index of the member) for the stable LSD
//                                             passes: the members are ordered component by component, last component
//                                             first, the group's start index last — a segmented sort by a key of any width
//   k_seg_writeback, k_seg_neweq              the members go back to their slots in the new order, with the flags
// The key components are 7 text symbols (9 bits each: 0 = past the end) or one rank look-up isa[p + j D].
// ---------------------------------------------------------------------------------------------
constexpr u32 kSegTile = 4096, kSegIPT = kSegTile / kBlock;     // entries per block, per thread (consecutive)
constexpr u32 kSegNone = 0xffffffffu;
// last[b] = index of the last entry of tile b that starts a group, kSegNone if it has none
__global__ __launch_bounds__(kBlock) void k_seg_last(const uint8_t *__restrict__ same, u32 n, u32 *__restrict__ last) {
  __shared__ u32 tmp[kWaves];
  const u32 base = blockIdx.x * kSegTile + threadIdx.x * kSegIPT;
  u32 cur = 0;                                             // index + 1 of the last start seen (0 = none)
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) { const u32 i = base + j; if (i < n && same[i] == 0) cur = i + 1; }
  cur = wave_reduce_max(cur);
  if (lane_id() == 0) tmp[wave_id()] = cur;
  __syncthreads();
  if (threadIdx.x == 0) { u32 m = 0; for (int w = 0; w < kWaves; w++) m = max(m, tmp[w]); last[blockIdx.x] = m ? m - 1 : kSegNone; }
}
// in place: last[b] -> start of the group that reaches into tile b (entry 0 starts a group, so tile 0 carries 0 in)
__global__ __launch_bounds__(1024) void k_seg_carry(u32 *__restrict__ last, u32 ntiles) {
  __shared__ u32 tmp[16];
  u32 carry = 1;                                           // index + 1
  for (u32 base = 0; base < ntiles; base += 1024) {
    const u32 b = base + threadIdx.x;
    const u32 v = (b < ntiles && last[b] != kSegNone) ? last[b] + 1 : 0u;
    u32 inc = v;                                           // inclusive running max inside the wave, then across waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane_id() >= (u32)o) inc = max(inc, t); }
    if (lane_id() == 63) tmp[wave_id()] = inc;
    __syncthreads();
    u32 pre = carry;
    for (u32 w = 0; w < wave_id(); w++) pre = max(pre, tmp[w]);
    u32 all = carry;
    for (u32 w = 0; w < 16; w++) all = max(all, tmp[w]);
    const u32 up = __shfl_up(inc, 1);
    const u32 excl = max(pre, lane_id() ? up : 0u);        // max over everything before tile b
    if (b < ntiles) last[b] = excl - 1;
    __syncthreads();
    carry = all;
  }
}
// f(i, start of i's group) for every entry
template <class F>
__global__ __launch_bounds__(kBlock) void k_seg_apply(const uint8_t *__restrict__ same, u32 n, const u32 *__restrict__ carry, F f) {
  __shared__ u32 tmp[kWaves];
  const u32 base = blockIdx.x * kSegTile + threadIdx.x * kSegIPT;
  u32 g[kSegIPT], cur = 0;
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) { const u32 i = base + j; if (i < n && same[i] == 0) cur = i + 1; g[j] = cur; }
  u32 inc = cur;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(inc, o); if (lane_id() >= (u32)o) inc = max(inc, t); }
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  u32 pre = carry[blockIdx.x] + 1;
  for (u32 w = 0; w < wave_id(); w++) pre = max(pre, tmp[w]);
  const u32 up = __shfl_up(inc, 1);
  pre = max(pre, lane_id() ? up : 0u);
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) { const u32 i = base + j; if (i < n) f(i, (g[j] ? g[j] : pre) - 1u); }
}
struct SegStore { u32 *gstart; __device__ __forceinline__ void operator()(u32 i, u32 s) const { gstart[i] = s; } };
// isa[sa[i]] = base + (start of i's group) + 1: the rank of the suffix by the symbols its group agrees on
struct SegIsa { const u64 *sa; u64 *isa; u64 base; __device__ __forceinline__ void operator()(u32 i, u32 s) const { isa[sa[i]] = base + s + 1; } };

// member of a group of more than T entries?  (its group's start + T is still inside the group)
__device__ __forceinline__ bool seg_big(const u32 *__restrict__ gstart, u32 n, u32 i, u32 T) {
  const u32 s = gstart[i];
  return s + T < n && gstart[s + T] == s;
}
__global__ __launch_bounds__(kBlock) void k_big_count(const u32 *__restrict__ gstart, u32 n, u32 T, u32 *__restrict__ counts) {
  __shared__ u32 tmp[kWaves];
  const u32 base = blockIdx.x * kSegTile + threadIdx.x * kSegIPT;
  u32 c = 0;
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) { const u32 i = base + j; if (i < n && seg_big(gstart, n, i, T)) c++; }
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int w = 0; w < kWaves; w++) t += tmp[w]; counts[blockIdx.x] = t; }
}
// where the position of entry i comes from: the sorted records of the tie pass (16-byte, 8-byte words) or the shard
struct PosRec16 { const Rec16 *h; __device__ __forceinline__ u64 operator()(u32 i) const { return wide_pos(h[i]); } };
struct PosWord8 { const u64 *h; u64 pmask; __device__ __forceinline__ u64 operator()(u32 i) const { return h[i] & pmask; } };
struct PosShard { const u64 *s; __device__ __forceinline__ u64 operator()(u32 i) const { return s[i]; } };
template <class Pos>
__global__ __launch_bounds__(kBlock) void k_big_write(const u32 *__restrict__ gstart, u32 n, u32 T, const u32 *__restrict__ base_excl, Pos pos,
                                                     u32 *__restrict__ cslot, u32 *__restrict__ gid, u64 *__restrict__ cpos) {
  __shared__ u32 tmp[kWaves];
  const u32 base = blockIdx.x * kSegTile + threadIdx.x * kSegIPT;
  bool b[kSegIPT]; u32 c = 0;
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) { const u32 i = base + j; b[j] = i < n && seg_big(gstart, n, i, T); c += b[j] ? 1u : 0u; }
  u32 tot;
  u32 ex = base_excl[blockIdx.x] + block_excl_scan<kWaves>(c, tmp, tot);
#pragma unroll
  for (u32 j = 0; j < kSegIPT; j++) {
    const u32 i = base + j;
    if (b[j]) { cslot[ex] = i; gid[ex] = gstart[i]; cpos[ex] = pos(i); ex++; }
  }
}
// key components
struct SegKeySyms {                       // 7 symbols from offset off: 9 bits each, 0 = past the end of the text
  WideKey k; u32 off;
  __device__ __forceinline__ u64 operator()(u64 p, const uint16_t *lcode) const {
    u64 v = 0;
#pragma unroll
    for (u32 j = 0; j < 7; j++) { const u64 q = p + off + j; v = (v << 9) | (q < k.n ? (u64)lcode[k.t[q]] : 0ull); }
    return v;
  }
};
struct SegKeyIsa {                        // rank of the continuation add symbols further on (0 = the suffix has ended)
  const u64 *isa; u64 n, add;
  __device__ __forceinline__ u64 operator()(u64 p, const uint16_t *) const { return p + add < n ? isa[p + add] : 0ull; }
};
// records of one LSD component: out[j] = {key of the member that stands at place j now, its index}; prev = the order so far
// (nullptr: members in index order)
template <class Key>
__global__ __launch_bounds__(kBlock) void k_seg_recs(const Rec16 *prev, u32 nb, const u64 *__restrict__ cpos, Key key,
                                                    const uint16_t *__restrict__ code, Rec16 *out) {   // (out may be prev: element j only)
  __shared__ uint16_t lcode[256];
  if (threadIdx.x < 256) lcode[threadIdx.x] = code ? code[threadIdx.x] : (uint16_t)0;
  __syncthreads();
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < nb; j += gridDim.x * kBlock) {
    const u32 idx = prev ? prev[j].pos : j;
    const u64 v = key(cpos[idx], lcode);
    out[j] = Rec16{(u32)v, (u32)(v >> 32), 0u, idx};
  }
}
__global__ __launch_bounds__(kBlock) void k_seg_recs_gid(const Rec16 *prev, u32 nb, const u32 *__restrict__ gid, Rec16 *out) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < nb; j += gridDim.x * kBlock) {
    const u32 idx = prev ? prev[j].pos : j;
    out[j] = Rec16{gid[idx], 0u, 0u, idx};
  }
}
// the members in their new order: place j of the compacted list is slot cslot[j] of the shard (groups are index ranges and
// the last component ordered was the group's start, so every member stays inside its group's range)
__global__ __launch_bounds__(kBlock) void k_seg_writeback(const Rec16 *__restrict__ sorted, u32 nb, const u32 *__restrict__ cslot,
                                                         const u64 *__restrict__ cpos, u64 *__restrict__ shard) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < nb; j += gridDim.x * kBlock) shard[cslot[j]] = cpos[sorted[j].pos];
}
// neweq[slot] = the member agrees with the one before it in ALL components (same group, compared by cmp); words[2] += those
struct SegCmpIsa { const u64 *isa; u64 n, D; u32 W; __device__ __forceinline__ bool operator()(u64 p, u64 q) const { return wide_cmp_isa(isa, n, p, q, D, W) == 0; } };
template <class Cmp>
__global__ __launch_bounds__(kBlock) void k_seg_neweq(const Rec16 *__restrict__ sorted, u32 nb, const u32 *__restrict__ cslot, const u32 *__restrict__ gid,
                                                     const u64 *__restrict__ cpos, Cmp cmp, uint8_t *__restrict__ neweq, u32 *words) {
  u32 dup = 0;
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < nb; j += gridDim.x * kBlock) {
    const u32 a = sorted[j].pos;
    bool same = false;
    if (j > 0) { const u32 b = sorted[j - 1].pos; same = gid[a] == gid[b] && cmp(cpos[b], cpos[a]); }
    neweq[cslot[j]] = same ? 1 : 0;
    dup += same ? 1u : 0u;
  }
  dup = wave_reduce(dup);
  if (lane_id() == 0 && dup) atomicAdd(&words[2], dup);
}
// same-image flags of the 16-byte records of the tie pass (the 8-byte form has them from its local sort)
__global__ __launch_bounds__(kBlock) void k_wide_same16(const Rec16 *__restrict__ h, u32 nrec, uint8_t *__restrict__ same) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nrec; i += gridDim.x * kBlock)
    same[i] = (i > 0 && wide_img(h[i]) == wide_img(h[i - 1])) ? 1 : 0;
}

// order-sensitive checksum of a shard with global indices (64-bit values: sum of mix(mix(index) ^ position))
__global__ __launch_bounds__(kBlock) void k_wide_checksum(const u64 *__restrict__ shard, u32 cnt, u64 first, u64 *out) {
  u64 acc = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < cnt; i += gridDim.x * kBlock)
    acc += splitmix64(splitmix64(first + i) ^ shard[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) atomicAdd(reinterpret_cast<unsigned long long *>(out), (unsigned long long)acc);
}

}  // namespace dc3
