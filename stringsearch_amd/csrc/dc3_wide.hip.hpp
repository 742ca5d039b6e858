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
// sa / eq: the whole order and its flags (all ranks' shards in rank order); groups are at most kWideTieBig long
__global__ __launch_bounds__(kBlock) void k_wide_isa_scatter(const u64 *__restrict__ sa, const uint8_t *__restrict__ eq, u64 n, u64 *__restrict__ isa) {
  for (u64 g = (u64)blockIdx.x * kBlock + threadIdx.x; g < n; g += (u64)gridDim.x * kBlock) {
    u64 s = g;
    while (eq[s]) s--;
    isa[sa[g]] = s + 1;
  }
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
