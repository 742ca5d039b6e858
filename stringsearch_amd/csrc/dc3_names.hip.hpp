// dc3_names.hip.hpp — level-0 alphabet, direct (sort-free) names, triple records.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Level-0 alphabet: byte histogram -> dense order-preserving code table (codes 1..sigma).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_byte_presence(const uint8_t *t, u32 n, u32 *present /*[256]*/) {
  __shared__ u32 loc[256];
  loc[threadIdx.x] = 0;
  __syncthreads();
  const u32 nvec = n / 16;
  const uint4 *tv = reinterpret_cast<const uint4 *>(t);
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += gridDim.x * kBlock) {
    uint4 v = tv[i];
    u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      loc[w[j] & 255] = 1; loc[(w[j] >> 8) & 255] = 1; loc[(w[j] >> 16) & 255] = 1; loc[w[j] >> 24] = 1;
    }
  }
  if (blockIdx.x == 0) for (u32 i = nvec * 16 + threadIdx.x; i < n; i += kBlock) loc[t[i]] = 1;
  __syncthreads();
  if (loc[threadIdx.x]) present[threadIdx.x] = 1;   // benign race: every writer stores 1
}
// one block of 256 threads: code[b] = 1 + #present bytes below b (0 if absent); sigma_out = #present
__global__ __launch_bounds__(kBlock) void k_make_codes(const u32 *present, uint16_t *code /*[256]*/, u32 *sigma_out) {
  __shared__ u32 tmp[kWaves];
  const u32 p = present[threadIdx.x] ? 1u : 0u;
  u32 tot;
  const u32 ex = block_excl_scan<kWaves>(p, tmp, tot);
  code[threadIdx.x] = (uint16_t)(p ? ex + 1 : 0);   // dense, order-preserving, 1..sigma (sigma <= 256)
  if (threadIdx.x == 0) *sigma_out = tot;
}

// ---------------------------------------------------------------------------------------------
// Direct (sort-free) naming: when (K+1)^3 fits 31 bits the packed symbols themselves are an order-
// and equality-preserving name, so the sample string
//   R[slot(i)] for i%3 != 0   (slot = i/3 for mod 1, i/3 + m0 for mod 2; lib.rs:93-98)
// is produced by one streaming pass (names need not be dense; replaces lib.rs:62-100 for small
// alphabets).  The name packs w >= 3 symbols, B = K+1:   name(i) = sum_{t<w} S[i+t] * B^(w-1-t) + 1.
// w = 3 is the K–S triple and the default.  Wider names (overlapping neighbours) are also valid —
// comparing name(i), name(i+3), ... still compares the suffixes in order, and as in K–S two sample
// suffixes of one residue differ no later than the name covering the shorter one's end (a zero in its
// first 3 symbols) — but they were measured SLOWER (DNA 1 GiB 199 -> 217 ms): they save cheap direct
// levels and make the first sorted level's alphabet huge and sparse (93-bit keys, all top-32-bit
// prefixes tied).  The host always passes w = 3 (the switch that tried wider names is gone since round 3).
// Thread g owns positions 3g+1 and 3g+2.
// ---------------------------------------------------------------------------------------------
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_name_direct(Sym S, u32 m, u32 m0, u32 m02, u32 B, u32 w, u32 Bw1,
                                                       u32 *R) {
  const u32 ngroups = m0;   // group g: samples 3g+1 (slot g) and 3g+2 (slot m0+g)
  __shared__ uint16_t lcode[256];
  S.stage(lcode);
  if (w == 3) {               // the K–S triple: symbols 3g+1 .. 3g+4 in one get4
    for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += gridDim.x * kBlock) {
      const u32 i = 3 * g + 1;
      u32 q[4]; S.get4(i, lcode, q);
      R[g] = ((q[0] * B + q[1]) * B + q[2]) + 1;
      if (i + 1 < m) R[m0 + g] = ((q[1] * B + q[2]) * B + q[3]) + 1;
    }
  } else {
    for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += gridDim.x * kBlock) {
      const u32 i = 3 * g + 1;
      const u32 first = S.get(i);
      u32 acc = first;
      for (u32 t = 1; t < w; t++) acc = acc * B + S.get(i + t);      // Horner over S[i .. i+w)
      // mod-1 sample exists for every g < m0 (includes the dummy at i == m when m%3 == 1)
      R[g] = acc + 1;
      if (i + 1 < m) R[m0 + g] = (acc - first * Bw1) * B + S.get(i + w) + 1;   // S[i+1 .. i+1+w)
    }
  }
  // zero tail of R (sentinels of the next level, lib.rs:51-53)
  if (blockIdx.x == 0 && threadIdx.x < 8) R[m02 + threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// Triple records in position order (lib.rs:62-70 fused with the key reads of :74-76).
// key = (s0*B + s1)*B + s2 (up to 93 bits, B = K+1), thread g emits records of 3g+1, 3g+2
// at indices 2g, 2g+1 — i.e. ascending text position like the reference's R.
// n12 = number of sample positions = m02.
// ---------------------------------------------------------------------------------------------
// key = (s0*B + s1)*B + s2 with B = K+1 (dense arithmetic packing: no bits are wasted when K is
// not a power of two, which keeps the top key bits discriminating for the prefix-sort path)
__device__ __forceinline__ Rec16 make_rec(u32 s0, u32 s1, u32 s2, u32 B, u32 pos) {
  const u64 lo = (u64)s1 * B + s2;                     // < B^2 <= 2^62
  const u64 B2 = (u64)B * B;
  const u64 p_lo = (u64)s0 * B2;
  u64 p_hi = __umul64hi((u64)s0, B2);
  const u64 s_lo = p_lo + lo;
  p_hi += (s_lo < p_lo) ? 1u : 0u;
  Rec16 r; r.k0 = (u32)s_lo; r.k1 = (u32)(s_lo >> 32); r.k2 = (u32)p_hi; r.pos = pos;
  return r;
}
// the record of a WIDE window: W > 3 symbols of sb bits each, first symbol most significant, in the 96-bit key
// (dc3_ssort.hip.hpp explains why names taken from such records order the samples as the triple names do)
__device__ __forceinline__ Rec16 ss_window_rec(const u32 *s, u32 W, u32 sb, u32 pos) {
  u64 lo = 0; u32 hi = 0;
  for (u32 j = 0; j < W; j++) { hi = (hi << sb) | (u32)(lo >> (64 - sb)); lo = (lo << sb) | s[j]; }
  Rec16 r; r.k0 = (u32)lo; r.k1 = (u32)(lo >> 32); r.k2 = hi; r.pos = pos;
  return r;
}
// the sample record of position i: the K-S triple in base B (W == 0), or the W-symbol window
template <class Sym>
__device__ __forceinline__ Rec16 sample_rec(const Sym &S, u32 i, u32 B, u32 W, u32 sb) {
  if (W == 0) return make_rec(S.get(i), S.get(i + 1), S.get(i + 2), B, i);
  u32 s[7];
#pragma unroll
  for (int j = 0; j < 7; j++) s[j] = (u32)j < W ? S.get(i + j) : 0u;
  return ss_window_rec(s, W, sb, i);
}
__device__ __forceinline__ void store_rec(Rec16 *out, u32 i, const Rec16 &r) { out[i] = r; }
__device__ __forceinline__ void store_rec(Rec12 *out, u32 i, const Rec16 &r) { out[i] = Rec12{r.k0, r.k1, r.pos}; }
template <class Sym, class Rec>
__global__ __launch_bounds__(kBlock) void k_pack_triples(Sym S, u32 m, u32 m0, u32 m02, u32 b, Rec *out) {
  // sample positions in ascending order: 1,2,4,5,7,8,...; index of 3g+1 is 2g, of 3g+2 is 2g+1
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < m0; g += gridDim.x * kBlock) {
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    store_rec(out, 2 * g, make_rec(s1, s2, s3, b, i));
    if (2 * g + 1 < m02) store_rec(out, 2 * g + 1, make_rec(s2, s3, s4, b, i + 1));
  }
}

// chunked variant that also produces the digit table of the first radix pass over the records (digit = key bits
// [0, log2 NB), table[d*nchunks + block]); output indices [block*chunk, ...), chunk even
template <class Sym, class Rec, int NB>
__global__ __launch_bounds__(kBlock) void k_pack_triples_hist(Sym S, u32 m, u32 m0, u32 m02, u32 b, Rec *out, u32 chunk,
                                                             u32 nchunks, u32 *__restrict__ table) {
  __shared__ u32 hist[kWaves][NB];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(m02, begin + chunk);
  for (u32 g = begin / 2 + threadIdx.x; 2 * g < end; g += kBlock) {
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    const Rec16 r0 = make_rec(s1, s2, s3, b, i);
    store_rec(out, 2 * g, r0);
    atomicAdd(&myh[r0.k0 & (NB - 1)], 1u);
    if (2 * g + 1 < m02) {
      const Rec16 r1 = make_rec(s2, s3, s4, b, i + 1);
      store_rec(out, 2 * g + 1, r1);
      atomicAdd(&myh[r1.k0 & (NB - 1)], 1u);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}

}  // namespace dc3
