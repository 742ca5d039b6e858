// dc3_kernels.cuh — gfx950 (CDNA4, wave64) device kernels of the DC3/Skew suffix-array path.
//
// Every kernel maps to a loop of the reference's crates/dc3/src/lib.rs (cited per kernel).
// Design rules: the work is HBM-bound integer/index traffic, so
//   * arrays are moved as coalesced records ((key,pos) / merge tuples), never sorted indirectly;
//   * the K+1-counter counting sort of the reference (lib.rs:15-39) becomes 8-bit-digit LSD passes
//     with per-wave LDS digit counters, wave64 ballot ranking (stable) and an LDS reorder so that
//     each (tile,digit) run leaves the CU as one contiguous burst;
//   * blocks own contiguous chunks (up-sweep / scan / down-sweep), so the only inter-block
//     communication is through kernel boundaries — no spin waits, no placement assumptions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dc3 {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kBlock = 256;         // 4 waves of 64
constexpr int kWaves = kBlock / 64;

// ---------------------------------------------------------------------------------------------
// Record types
// ---------------------------------------------------------------------------------------------
// Sample-triple record: 96-bit packed key (k[0] least significant) + text position.
struct __attribute__((aligned(16))) Rec16 { u32 k0, k1, k2, pos; };
// (destination, value) pair of an inverse-permutation pass (rank <- SA inversion, lib.rs:106-113).
struct __attribute__((aligned(8))) Rec8 { u32 key, val; };
// Sample-triple record when the packed key fits 64 bits (straight ordering of mid-size alphabets).
struct Rec12 { u32 k0, k1, pos; };
// Merge tuple of a sample (mod-1 / mod-2) suffix, 16 B:
//   pos%3==1: (c0=S[pos], r=rank[pos+1]), cx = S[pos-1]   (cx feeds the derived mod-0 tuple)
//   pos%3==2: (c0=S[pos], cx=S[pos+1], r=rank[pos+2])
struct __attribute__((aligned(16))) Tup12 { u32 pos, r, c0, cx; };
// Merge tuple of a mod-0 suffix j: (c0=S[j], c1=S[j+1], r1=rank[j+1], r2=rank[j+2]), 20 B.
struct Tup0 { u32 pos, c0, c1, r1, r2; };

// ---------------------------------------------------------------------------------------------
// Symbol readers: level 0 reads bytes through the dense code table (codes 1..sigma, 0 past the
// end = the sentinel of lib.rs:41-42); deeper levels read u32 names whose zero tail is physical.
// ---------------------------------------------------------------------------------------------
// get4(i, lds, out): symbols i..i+3 — for bytes one (unaligned) dword load + 4 look-ups in a per-block
// LDS copy of the code table (stage() fills it; the text buffer is padded with 64 zero bytes).
struct SymU8 {
  const uint8_t *t; const uint16_t *code; u32 m;
  static constexpr bool kTable = true;
  __device__ __forceinline__ u32 get(u32 i) const { return i < m ? (u32)code[t[i]] : 0u; }
  __device__ __forceinline__ void stage(uint16_t *lds) const {     // blockDim.x >= 256
    if (threadIdx.x < 256) lds[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
  }
  __device__ __forceinline__ void get4(u32 i, const uint16_t *lds, u32 *out) const {
    u32 w; __builtin_memcpy(&w, t + i, 4);
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (i + k < m) ? (u32)lds[(w >> (8 * k)) & 255u] : 0u;
  }
};
struct SymU32 {
  const u32 *s; u32 m;   // s has >= 8 zero words after s[m-1]
  static constexpr bool kTable = false;
  __device__ __forceinline__ u32 get(u32 i) const { return s[i]; }
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ void get4(u32 i, const uint16_t *, u32 *out) const {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = s[i + k];
  }
};

// ---------------------------------------------------------------------------------------------
// Wave / block primitives (wave64)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  const u32 lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(v, o); if (lane >= (u32)o) v += t; }
  return v;
}
__device__ __forceinline__ u32 wave_reduce_max(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (u32)__shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ u32 wave_reduce(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Exclusive scan of one value per thread over a block of NW waves; tmp needs NW words of LDS.
template <int NW>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *tmp, u32 &total) {
  const u32 inc = wave_incl_scan(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  u32 woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) { u32 t = tmp[i]; if ((u32)i < wave_id()) woff += t; tot += t; }
  __syncthreads();
  total = tot;
  return woff + inc - v;
}

// popcount of mask bits below this lane
__device__ __forceinline__ u32 mbcnt(u64 mask) {
  return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// ---------------------------------------------------------------------------------------------
// Single-block exclusive scan of a (small) u32 array in place; used for digit tables and
// per-chunk counts.  total_out (optional) receives the grand total.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan_excl_inplace(u32 *data, u32 n, u32 *total_out) {
  __shared__ u32 tmp[16];
  u32 carry = 0;
  const u32 tid = threadIdx.x;
  for (u32 base = 0; base < n; base += 1024 * 4) {
    const u32 i0 = base + tid * 4;
    u32 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (i0 + j < n) ? data[i0 + j] : 0u;
    const u32 s = v[0] + v[1] + v[2] + v[3];
    u32 tot;
    u32 ex = block_excl_scan<16>(s, tmp, tot) + carry;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (i0 + j < n) data[i0 + j] = ex; ex += v[j]; }
    carry += tot;
  }
  if (tid == 0 && total_out) *total_out = carry;
}

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
// prefixes tied).  Kept behind DC3HIP_WIDE_NAMES=1 for experiments; both widths are parity-tested.
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

// ---------------------------------------------------------------------------------------------
// Stable LSD radix pass (lib.rs:15-39 with the K+1 counters replaced by digits of NB = 256 or 512
// bins; 9-bit digits are used where they save a pass, e.g. 25-27-bit symbols).
//   up-sweep  : per-chunk digit histogram (lib.rs:20-22)          -> table[digit][chunk]
//   scan      : exclusive prefix sums over table (lib.rs:25-32)   (k_scan_rows + k_scan_excl_inplace)
//   down-sweep: stable scatter (lib.rs:35-38)
// A digit is (key >> shift) & mask of the record's sort key.
// ---------------------------------------------------------------------------------------------
struct KeyDig { u32 shift, mask; };
// Rec8 as a 64-bit sort word: key = high half, val = low half (prefix-sort records keep the position
// in the low pbits of val, the rest is the monotone key image)
__device__ __forceinline__ u64 rec8_word(const Rec8 &r) { return ((u64)r.key << 32) | r.val; }
__device__ __forceinline__ u32 digit_of(const Rec8 &r, KeyDig d) { return (u32)(rec8_word(r) >> d.shift) & d.mask; }
__device__ __forceinline__ u32 digit_of(const Rec12 &r, KeyDig d) {
  const u64 k = (u64)r.k0 | ((u64)r.k1 << 32);
  return (u32)(k >> d.shift) & d.mask;
}
__device__ __forceinline__ u32 digit_of(const Rec16 &r, KeyDig d) {     // 96-bit key, shift < 96
  const u32 w = d.shift >> 5, off = d.shift & 31;
  const u32 a = w == 0 ? r.k0 : (w == 1 ? r.k1 : r.k2);
  const u32 b = w == 0 ? r.k1 : (w == 1 ? r.k2 : 0u);
  return (off ? ((a >> off) | (b << (32 - off))) : a) & d.mask;
}
// mod-0 positions are real symbols (c0 >= 1), so the key is c0-1 in [0, K)
__device__ __forceinline__ u32 digit_of(const Tup0 &r, KeyDig d) { return ((r.c0 - 1u) >> d.shift) & d.mask; }

template <class Rec, int NB>
__global__ __launch_bounds__(kBlock) void k_rs_upsweep(const Rec *__restrict__ in, u32 n, u32 chunk, u32 nchunks,
                                                      KeyDig dig, u32 *__restrict__ table) {
  __shared__ u32 hist[kWaves][NB];
  const u32 tid = threadIdx.x;
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = tid; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  const u32 begin = blockIdx.x * chunk;
  const u32 end = min(n, begin + chunk);
  u32 *myh = hist[wave_id()];
  for (u32 i = begin + tid; i < end; i += kBlock) {
    const Rec r = in[i];
    atomicAdd(&myh[digit_of(r, dig)], 1u);
  }
  __syncthreads();
  for (int j = tid; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}

template <class Rec, int IPT, int NW, int NB>
struct DownsweepSmem {
  static constexpr int kTile = NW * 64 * IPT;
  static constexpr size_t kBytes = sizeof(Rec) * kTile + sizeof(u32) * (NW * NB + NB + NB + 32);
};

// table rows are scanned per digit (k_scan_rows) and the 256 digit totals separately
// (k_scan_excl_inplace on digit_base), so the global base of (digit d, chunk c) is
// digit_base[d] + table[d*nchunks + c].
// NW waves per block (measured on MI355X, profiles/r01_radix_downsweep_variants.txt: 16 waves x 8
// items = 8192-record tiles move 3.5 TB/s vs 2.9 TB/s for 4 waves x 16 items).
// Loaders: where a down-sweep takes its records from.  ArrayLoader = a plain record array.  A loader
// may also DROP elements (load() returns false), which fuses an order-preserving selection into the
// pass (used for Step 2: mod-0 tuples are made from the mod-1 entries of the sorted sample tuples
// and immediately partitioned by their first key byte — lib.rs:118-126 in one pass).
template <class Rec>
struct ArrayLoader {
  const Rec *p;
  __device__ __forceinline__ bool load(u32 i, Rec &r) const { r = p[i]; return true; }
};

// PF: prefetch the next tile into registers while the current one is ranked/reordered (pays for
// 8-byte records: 2.3 -> 3.4 TB/s; costs registers and loses for 16/20-byte records, see
// profiles/r01_radix_downsweep_variants_v2.txt).
template <class Rec, int NB, int IPT, int NW, bool PF, class Loader>
__global__ __launch_bounds__(NW * 64) void k_rs_downsweep(Loader in, Rec *__restrict__ out, u32 n,
                                                         u32 chunk, u32 nchunks, KeyDig dig,
                                                         const u32 *__restrict__ table,
                                                         const u32 *__restrict__ digit_base) {
  constexpr int kB = NW * 64;
  constexpr int kTile = kB * IPT;
  constexpr int kWItems = 64 * IPT;
  constexpr int kBits = NB == 512 ? 9 : 8;
  static_assert(NB == 256 || NB == 512, "digit bins");
  static_assert(NW * 64 >= NB, "one thread per digit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *srec = reinterpret_cast<Rec *>(smem);
  u32 *wcnt = reinterpret_cast<u32 *>(smem + sizeof(Rec) * kTile);   // [NW][NB]
  u32 *dbase = wcnt + NW * NB;                                       // [NB] running global base
  u32 *texcl = dbase + NB;                                           // [NB] tile-exclusive prefix
  u32 *tmp = texcl + NB;                                             // [NW]
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 begin = blockIdx.x * chunk;
  const u32 end = min(n, begin + chunk);
  if (tid < NB) dbase[tid] = digit_base[tid] + table[(size_t)tid * nchunks + blockIdx.x];
  u32 *mycnt = wcnt + w * NB;
  Rec r[IPT], rn[PF ? IPT : 1];
  bool okn[PF ? IPT : 1];
  if (PF) {
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      okn[PF ? k : 0] = (begin + t < end) && in.load(begin + t, rn[PF ? k : 0]);
    }
  }

  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 nin = min((u32)kTile, end - tile);
#pragma unroll
    for (int j = 0; j < NB / 64; j++) mycnt[lane + 64 * j] = 0;
    u32 d[IPT], rk[IPT];
    bool ok[IPT];
    // wave w owns tile items [w*kWItems, (w+1)*kWItems); round k covers 64 consecutive items
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if (PF) { r[k] = rn[PF ? k : 0]; ok[k] = okn[PF ? k : 0]; }
      else ok[k] = (t < nin) && in.load(tile + t, r[k]);
      d[k] = ok[k] ? digit_of(r[k], dig) : 0u;
    }
    if (PF) {
      const u32 nt = tile + kTile;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const u32 t = w * kWItems + k * 64 + lane;
        okn[PF ? k : 0] = (nt + t < end) && in.load(nt + t, rn[PF ? k : 0]);
      }
    }
    // stable ranking: items of one wave-round with equal digit are ordered by lane.  The lowest
    // peer lane bumps the wave's digit counter with one LDS atomic per round; the atomics of all
    // rounds are issued back to back (LDS executes them in order, so the returned values are the
    // running prefix) and the bases are broadcast afterwards.  Dropped / out-of-range lanes take
    // no part.
    {
      u32 below[IPT], leader[IPT], old[IPT];
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        u64 peers = __ballot(ok[k]);
#pragma unroll
        for (int bit = 0; bit < kBits; bit++) {
          const bool one = (d[k] >> bit) & 1u;
          const u64 mk = __ballot(one);
          peers &= one ? mk : ~mk;
        }
        below[k] = mbcnt(peers);
        leader[k] = ok[k] ? (u32)__ffsll((unsigned long long)peers) - 1u : lane;
        old[k] = 0;
        if (ok[k] && below[k] == 0) old[k] = atomicAdd(&mycnt[d[k]], (u32)__popcll(peers));
      }
#pragma unroll
      for (int k = 0; k < IPT; k++) rk[k] = __shfl(old[k], leader[k]) + below[k];
    }
    __syncthreads();
    // per digit (thread tid = digit): prefix over waves, tile total, tile-exclusive prefix
    u32 tot = 0;
    if (tid < NB) {
#pragma unroll
      for (int i = 0; i < NW; i++) { const u32 c = wcnt[i * NB + tid]; wcnt[i * NB + tid] = tot; tot += c; }
    }
    u32 nkeep;
    const u32 ex = block_excl_scan<NW>(tid < NB ? tot : 0u, tmp, nkeep);
    if (tid < NB) texcl[tid] = ex;
    __syncthreads();
    // reorder through LDS so every digit run is contiguous
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      if (ok[k]) srec[texcl[d[k]] + wcnt[w * NB + d[k]] + rk[k]] = r[k];
    }
    __syncthreads();
    for (u32 q = tid; q < nkeep; q += kB) {
      const Rec x = srec[q];
      const u32 dd = digit_of(x, dig);
      out[dbase[dd] + (q - texcl[dd])] = x;
    }
    __syncthreads();
    if (tid < NB) dbase[tid] += tot;
    // (the barrier after ranking in the next iteration orders this update before its use)
  }
}

// Row-wise exclusive scan of the [NB][nchunks] digit table: block d scans row d in place and
// writes the row total to totals[d] (then scanned by k_scan_excl_inplace over NB entries).
__global__ __launch_bounds__(kBlock) void k_scan_rows(u32 *__restrict__ table, u32 nchunks, u32 *__restrict__ totals) {
  __shared__ u32 tmp[kWaves];
  u32 *row = table + (size_t)blockIdx.x * nchunks;
  u32 carry = 0;
  for (u32 base = 0; base < nchunks; base += kBlock * 4) {
    const u32 i0 = base + threadIdx.x * 4;
    u32 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (i0 + j < nchunks) ? row[i0 + j] : 0u;
    u32 tot;
    u32 ex = block_excl_scan<kWaves>(v[0] + v[1] + v[2] + v[3], tmp, tot) + carry;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (i0 + j < nchunks) row[i0 + j] = ex; ex += v[j]; }
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// ---------------------------------------------------------------------------------------------
// Naming (lib.rs:80-100): name = 1 + number of key changes before i in the sorted order.
// The kernels are generic over an accessor of the sorted sample order:
//   AccRec<Rec16|Rec12> : fully sorted records (straight LSD path)
//   AccHyb   : (pos, "differs from predecessor" byte) arrays of the prefix-sort + tie-refine path
//   k_name_count  : per-chunk count of "key differs from predecessor" flags
//   (scan of the counts, total = number of distinct names)
//   k_name_assign : emits (slot(pos_i), name_i) pairs      (R[..] = name, lib.rs:93-98)
//   k_assign_unique: when every name is unique (lib.rs:109-113), SA12[i] = slot(pos_i) and the
//                    pairs (slot(pos_i), i+1)
// The pairs go through the windowed inversion (k_invperm_local) instead of a random scatter.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_neq(const Rec16 &a, const Rec16 &b) {
  return (a.k0 != b.k0) | (a.k1 != b.k1) | (a.k2 != b.k2);
}
__device__ __forceinline__ bool key_neq(const Rec12 &a, const Rec12 &b) { return (a.k0 != b.k0) | (a.k1 != b.k1); }
__device__ __forceinline__ u32 slot_of(u32 pos, u32 m0) {
  const u32 q = pos / 3, rem = pos - 3 * q;
  return rem == 1 ? q : q + m0;
}
template <class Rec>
struct AccRec {
  const Rec *s;
  __device__ __forceinline__ u32 pos(u32 i) const { return s[i].pos; }
  __device__ __forceinline__ u32 neq(u32 i) const {
    if (i == 0) return 1u;
    const Rec a = s[i], b = s[i - 1];
    return key_neq(a, b) ? 1u : 0u;
  }
};
struct AccHyb {
  const Rec8 *h; const uint8_t *f; u32 posmask;   // pos = low bits of h[i].val; f[i] = 1 iff key(i) != key(i-1)
  __device__ __forceinline__ u32 pos(u32 i) const { return h[i].val & posmask; }
  __device__ __forceinline__ u32 neq(u32 i) const { return f[i]; }
};

constexpr int kNameIPT = 4;
// a name is unique iff its key differs from both neighbours in the sorted order
template <class Acc>
__device__ __forceinline__ u32 acc_unique(const Acc &acc, u32 i, u32 n) {
  return (acc.neq(i) && (i + 1 == n || acc.neq(i + 1))) ? 1u : 0u;
}
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_name_count(Acc acc, u32 n, u32 chunk, u32 *counts, u32 *uniq_total) {
  __shared__ u32 tmp[kWaves], tmpu[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0, u = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) { c += acc.neq(i); u += acc_unique(acc, i, n); }
  c = wave_reduce(c); u = wave_reduce(u);
  if (lane_id() == 0) { tmp[wave_id()] = c; tmpu[wave_id()] = u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0, tu = 0;
    for (int i = 0; i < kWaves; i++) { t += tmp[i]; tu += tmpu[i]; }
    counts[blockIdx.x] = t;
    if (tu) atomicAdd(uniq_total, tu);
  }
}
// sslot (optional, discarding recursion): sslot[i] = slot(pos_i) | unique_i << 31, and the pair value
// carries the same unique bit (names < 2^31 on that path).
constexpr u32 kUniqBit = 0x80000000u;
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_name_assign(Acc acc, u32 n, u32 chunk, const u32 *__restrict__ base_excl,
                                                       u32 m0, Rec8 *__restrict__ pairs, u32 *__restrict__ sslot) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  constexpr u32 kTile = kBlock * kNameIPT;
  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 i0 = tile + threadIdx.x * kNameIPT;
    u32 f[kNameIPT];
    u32 local = 0;
#pragma unroll
    for (int j = 0; j < kNameIPT; j++) { f[j] = (i0 + j < end) ? acc.neq(i0 + j) : 0u; local += f[j]; }
    u32 tot;
    u32 name = running + block_excl_scan<kWaves>(local, tmp, tot);
#pragma unroll
    for (int j = 0; j < kNameIPT; j++) {
      if (i0 + j < end) {
        name += f[j];
        const u32 sl = slot_of(acc.pos(i0 + j), m0);
        if (sslot) {
          const u32 ub = acc_unique(acc, i0 + j, n) ? kUniqBit : 0u;
          sslot[i0 + j] = sl | ub;
          pairs[i0 + j] = Rec8{sl, name | ub};
        } else {
          pairs[i0 + j] = Rec8{sl, name};
        }
      }
    }
    running += tot;
  }
}
// Emits (slot, i+1) pairs (coalesced) for the windowed inversion below instead of scattering 4-byte
// ranks: random 4-byte stores run at ~25 G/s on MI355X (profiles/r01_membench_access_patterns.txt),
// a partition by destination window + LDS-local placement is > 2x faster.
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_assign_unique(Acc acc, u32 n, u32 m0, u32 *__restrict__ sa12,
                                                         Rec8 *__restrict__ pairs) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 sl = slot_of(acc.pos(i), m0);
    sa12[i] = sl;
    pairs[i] = Rec8{sl, i + 1};
  }
}

// ---------------------------------------------------------------------------------------------
// Discarding recursion (Dementiev/Kärkkäinen/Mehnert/Sanders' refinement of lib.rs:103-108).
// A sample whose name is unique needs no further sorting — its rank is its index in the sorted
// array — and a comparison of two suffixes of R stops at the first unique name.  So the recursive
// string only has to contain the non-unique slots and the unique slots that directly follow a
// non-unique one (they terminate the comparisons that start before them).  RU[p] = name | unique<<31.
//   k_keep_count/k_keep_write : R'[j] = name of the j-th kept slot, kept[j] = slot | unique<<31
//   (child: SA' of R')
//   k_discard_gather          : x[r] = kept[SA'[r]]            (kept slots in suffix order)
//   k_nonuniq_count/_write    : pt[t] = t-th non-unique slot of x (their final relative order)
//   k_final_count/_assign     : walk the level's sorted array; unique entries keep their place, the
//                               t-th non-unique entry receives pt[t]   -> SA12 and (slot, rank) pairs
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool keep_slot(const u32 *__restrict__ RU, u32 p) {
  return !((RU[p] & kUniqBit) && (p == 0 || (RU[p - 1] & kUniqBit)));
}
__device__ __forceinline__ void block_count_store(u32 c, u32 *tmp, u32 *counts) {
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
__global__ __launch_bounds__(kBlock) void k_keep_count(const u32 *__restrict__ RU, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 p = begin + threadIdx.x; p < end; p += kBlock) c += keep_slot(RU, p) ? 1u : 0u;
  block_count_store(c, tmp, counts);
}
__global__ __launch_bounds__(kBlock) void k_keep_write(const u32 *__restrict__ RU, u32 n, u32 chunk,
                                                      const u32 *__restrict__ base_excl, u32 *__restrict__ Rp,
                                                      u32 *__restrict__ kept) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 p = tile + threadIdx.x;
    const bool f = (p < end) && keep_slot(RU, p);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(f ? 1u : 0u, tmp, tot);
    if (f) { const u32 v = RU[p]; Rp[running + ex] = v & ~kUniqBit; kept[running + ex] = p | (v & kUniqBit); }
    running += tot;
  }
}
__global__ __launch_bounds__(kBlock) void k_discard_gather(const u32 *__restrict__ sap, u32 n,
                                                          const u32 *__restrict__ kept, u32 *__restrict__ x) {
  for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) x[r] = kept[sap[r]];
}
__global__ __launch_bounds__(kBlock) void k_nonuniq_count(const u32 *__restrict__ x, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += (x[i] & kUniqBit) ? 0u : 1u;
  block_count_store(c, tmp, counts);
}
__global__ __launch_bounds__(kBlock) void k_nonuniq_write(const u32 *__restrict__ x, u32 n, u32 chunk,
                                                         const u32 *__restrict__ base_excl, u32 *__restrict__ pt) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    const bool f = (i < end) && !(x[i] & kUniqBit);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(f ? 1u : 0u, tmp, tot);
    if (f) pt[running + ex] = x[i];
    running += tot;
  }
}
// sslot[i] = slot | unique<<31 of the i-th entry of the level's sorted array
__global__ __launch_bounds__(kBlock) void k_final_assign(const u32 *__restrict__ sslot, u32 n, u32 chunk,
                                                        const u32 *__restrict__ base_excl,
                                                        const u32 *__restrict__ pt, u32 *__restrict__ sa12,
                                                        Rec8 *__restrict__ pairs) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    const u32 v = (i < end) ? sslot[i] : kUniqBit;
    const bool nonu = (i < end) && !(v & kUniqBit);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(nonu ? 1u : 0u, tmp, tot);
    if (i < end) {
      const u32 sl = nonu ? pt[running + ex] : (v & ~kUniqBit);
      sa12[i] = sl;
      pairs[i] = Rec8{sl, i + 1};
    }
    running += tot;
  }
}

// Partition pass of the windowed inversion (MSD order, not stable — order inside a window is
// irrelevant because k_invperm_local places by exact key).  The keys are a bijection onto [0,n), so
// the destination region of digit d of segment s is known analytically:
//   [ (s << seg_bits) + (d << shift), ... )  and holds exactly the keys that belong there;
// a tile only has to reserve space inside the region: one global atomicAdd per (tile, digit) on
// cursors[s*ndig + d].  No up-sweep, no scan: 16 B moved per pair instead of 24.
//   pass 1: shift = 22, seg_bits = 32 (one segment), ndig = ceil(n / 2^22) <= 1024
//   pass 2: shift = 14, seg_bits = 22, ndig = 256   (tiles never straddle a 2^22-pair segment)
constexpr int kPartNW = 16, kPartIPT = 8, kPartTile = kPartNW * 64 * kPartIPT;   // 8192 pairs
constexpr size_t kPartSmem = sizeof(Rec8) * kPartTile + sizeof(u32) * (2 * 1024 + 64);
__global__ __launch_bounds__(kPartNW * 64) void k_part_msd(const Rec8 *__restrict__ in, Rec8 *__restrict__ out, u32 n,
                                                          u32 shift, u32 seg_bits, u32 ndig,
                                                          u32 *__restrict__ cursors) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec8 *srec = reinterpret_cast<Rec8 *>(smem);
  u32 *hist = reinterpret_cast<u32 *>(smem + sizeof(Rec8) * kPartTile);   // [1024] counts -> tile-exclusive prefix
  u32 *gbase = hist + 1024;                                                // [1024] global base of the tile's run
  u32 *tmp = gbase + 1024;
  const u32 tid = threadIdx.x;
  const u32 begin = blockIdx.x * (u32)kPartTile;
  const u32 nvalid = min((u32)kPartTile, n - begin);
  const u32 seg = seg_bits >= 32 ? 0u : (begin >> seg_bits);
  const u32 seg_base = seg_bits >= 32 ? 0u : (seg << seg_bits);
  hist[tid] = 0;
  __syncthreads();
  Rec8 r[kPartIPT];
  u32 d[kPartIPT], rk[kPartIPT];
#pragma unroll
  for (int k = 0; k < kPartIPT; k++) {
    const u32 t = k * (kPartNW * 64) + tid;
    if (t < nvalid) {
      r[k] = in[begin + t];
      d[k] = ((r[k].key - seg_base) >> shift);
      rk[k] = atomicAdd(&hist[d[k]], 1u);
    }
  }
  __syncthreads();
  u32 cnt = 0;
  if (tid < ndig) {
    cnt = hist[tid];
    if (cnt) gbase[tid] = seg_base + (tid << shift) + atomicAdd(&cursors[seg * ndig + tid], cnt);
  }
  u32 tot;
  const u32 ex = block_excl_scan<kPartNW>(tid < ndig ? cnt : 0u, tmp, tot);
  hist[tid] = ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPartIPT; k++) {
    const u32 t = k * (kPartNW * 64) + tid;
    if (t < nvalid) srec[hist[d[k]] + rk[k]] = r[k];
  }
  __syncthreads();
  for (u32 q = tid; q < nvalid; q += kPartNW * 64) {
    const Rec8 x = srec[q];
    const u32 dd = (x.key - seg_base) >> shift;
    out[gbase[dd] + (q - hist[dd])] = x;
  }
}

// Final step of the windowed inversion.  The keys are a bijection onto [0,n), and the pairs are
// already partitioned by key >> kInvWindowBits, so pair range [w*W, (w+1)*W) holds exactly the
// destinations of window w: place them in LDS, then store the window with full coalesced lines.
constexpr int kInvWindowBits = 14;
constexpr int kInvWindow = 1 << kInvWindowBits;   // 16384 ranks = 64 KiB of LDS
__global__ __launch_bounds__(1024) void k_invperm_local(const Rec8 *__restrict__ pairs, u32 n, u32 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *win = reinterpret_cast<u32 *>(smem);
  const u32 base = blockIdx.x * (u32)kInvWindow;
  const u32 cnt = min((u32)kInvWindow, n - base);
  for (u32 i = threadIdx.x; i < cnt; i += 1024) {
    const Rec8 r = pairs[base + i];
    win[r.key - base] = r.val;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < cnt; i += 1024) out[base + i] = win[i];
}

// ---------------------------------------------------------------------------------------------
// Prefix-sort + tie-refine ordering of the sample triples (replaces the 3x radix_pass of
// lib.rs:74-76 when most keys are already distinct in an N-bit monotone image, N = 64 - pbits):
//   1. (image, pos) packed in one 64-bit word, 4 stable LSD passes over the image bits   [all samples]
//   2. elements whose image equals a neighbour's are "tied"; only those are re-sorted by the full
//      3b-bit key as 16-byte records and written back into the tied slots (same relative order)
// Result: h[i].val = position of the i-th smallest triple, f[i] = key differs from predecessor.
// ---------------------------------------------------------------------------------------------
// Monotone N-bit image of the full key (N = 64 - pbits, pbits = bits of a position; 34 bits at 1 GiB):
// X = key >> shx (its top 64 bits), hi = floor(X * mfix / 2^64) with mfix = floor(2^(64+N) / (Xmax+1)) —
// uses the whole N-bit range whatever the packing base is, so as few samples as possible collide
// (exact: the key itself fits N bits).  Any monotone map is valid for the tie-refine scheme.
// The record is the 64-bit word (hi << pbits) | pos.
struct HiMap { u64 mfix; u32 shx, pbits, nbits, exact; };
__device__ __forceinline__ Rec8 hyb_rec(const Rec16 &r, HiMap hm) {
  const u64 lo = (u64)r.k0 | ((u64)r.k1 << 32);
  u64 hi;
  if (hm.exact) hi = lo;
  else {
    const u64 x = hm.shx ? ((lo >> hm.shx) | ((u64)r.k2 << (64 - hm.shx))) : lo;
    hi = __umul64hi(x, hm.mfix);
  }
  const u64 w = (hi << hm.pbits) | r.pos;
  return Rec8{(u32)(w >> 32), (u32)w};
}
// stride > 1 samples every stride-th group (tie-rate predictor); out index = g / stride
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_pack_image(Sym S, u32 m, u32 m0, u32 m02, u32 b, HiMap hm, u32 stride,
                                                     u32 ngroups_out, Rec8 *out) {
  for (u32 go = blockIdx.x * kBlock + threadIdx.x; go < ngroups_out; go += gridDim.x * kBlock) {
    const u32 g = go * stride;
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    out[2 * go] = hyb_rec(make_rec(s1, s2, s3, b, i), hm);
    if (stride > 1 || 2 * g + 1 < m02) {
      // (in sampling mode a possibly non-existent last mod-2 sample only perturbs the estimate)
      if (2 * g + 1 < m02) out[2 * go + 1] = hyb_rec(make_rec(s2, s3, s4, b, i + 1), hm);
      else out[2 * go + 1] = Rec8{0xffffffffu, 0xffffffffu};
    }
  }
}
__device__ __forceinline__ bool hyb_tied(const Rec8 *h, u32 i, u32 n, u32 pbits) {
  const u64 a = rec8_word(h[i]) >> pbits;
  return (i > 0 && (rec8_word(h[i - 1]) >> pbits) == a) || (i + 1 < n && (rec8_word(h[i + 1]) >> pbits) == a);
}
__global__ __launch_bounds__(kBlock) void k_tie_count(const Rec8 *__restrict__ h, u32 n, u32 chunk, u32 pbits,
                                                     u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += hyb_tied(h, i, n, pbits) ? 1u : 0u;
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
// compacts the tied elements (order preserving): full-key record rebuilt from S, and the index of
// the slot it came from
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_tie_compact(Sym S, u32 b, const Rec8 *__restrict__ h, u32 n, u32 chunk,
                                                       u32 pbits,
                                                       const u32 *__restrict__ base_excl, Rec16 *__restrict__ sub,
                                                       u32 *__restrict__ tiedidx, u32 *__restrict__ gkey) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    const bool f = (i < end) && hyb_tied(h, i, n, pbits);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(f ? 1u : 0u, tmp, tot);
    if (f) {
      const u32 p = h[i].val & (pbits >= 32 ? 0xffffffffu : ((1u << pbits) - 1u));
      sub[running + ex] = make_rec(S.get(p), S.get(p + 1), S.get(p + 2), b, p);
      tiedidx[running + ex] = i;
      // group id = low 32 bits of the key image; merging two adjacent groups that differ only above
      // bit 31 is harmless (the union is sorted by the full key)
      gkey[running + ex] = (u32)(rec8_word(h[i]) >> pbits);
    }
    running += tot;
  }
}
// Tied samples form groups (equal key image) that are tiny on high-entropy input (Poisson: almost all of
// size 2-3).  When the largest group has at most kTieSmallMax members, one thread per group sorts
// it by the full key with a stable insertion sort — instead of 10 radix passes over the subset.
constexpr u32 kTieSmallMax = 16;
__global__ __launch_bounds__(kBlock) void k_tie_groupmax(const u32 *__restrict__ gkey, u32 t, u32 *maxlen) {
  u32 best = 0;
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) {
    const u32 k = gkey[j];
    if (j > 0 && gkey[j - 1] == k) continue;            // not a group start
    u32 e = j + 1;
    while (e < t && e - j <= kTieSmallMax && gkey[e] == k) e++;
    best = max(best, e - j);
  }
  best = wave_reduce_max(best);
  if (lane_id() == 0 && best) atomicMax(maxlen, best);
}
__device__ __forceinline__ bool key_less(const Rec16 &a, const Rec16 &b) {
  if (a.k2 != b.k2) return a.k2 < b.k2;
  if (a.k1 != b.k1) return a.k1 < b.k1;
  return a.k0 < b.k0;
}
__global__ __launch_bounds__(kBlock) void k_tie_sort_small(const Rec16 *__restrict__ sub, const u32 *__restrict__ gkey,
                                                          u32 t, Rec16 *__restrict__ out) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) {
    const u32 k = gkey[j];
    if (j > 0 && gkey[j - 1] == k) continue;
    u32 e = j + 1;
    while (e < t && gkey[e] == k) e++;
    const u32 len = e - j;                               // <= kTieSmallMax (checked by the host)
    Rec16 loc[kTieSmallMax];
    for (u32 x = 0; x < len; x++) {                      // stable insertion sort (input is in position order)
      const Rec16 v = sub[j + x];
      u32 y = x;
      while (y > 0 && key_less(v, loc[y - 1])) { loc[y] = loc[y - 1]; y--; }
      loc[y] = v;
    }
    for (u32 x = 0; x < len; x++) out[j + x] = loc[x];
  }
}
__global__ __launch_bounds__(kBlock) void k_tie_writeback(const Rec16 *__restrict__ sub, const u32 *__restrict__ tiedidx,
                                                         u32 t, Rec8 *__restrict__ h, uint8_t *__restrict__ f) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) {
    const Rec16 cur = sub[j];
    const u32 i = tiedidx[j];
    h[i].val = cur.pos;
    bool ne = true;
    if (j > 0) { const Rec16 prev = sub[j - 1]; ne = key_neq(cur, prev); }
    f[i] = ne ? 1 : 0;
  }
}

__global__ void k_base1(u32 *out_sa, u32 *out_rank) {
  if (threadIdx.x == 0) { if (out_sa) out_sa[0] = 0; if (out_rank) out_rank[0] = 1; }
}
__global__ void k_zero_tail(u32 *p, u32 from, u32 count) {
  if (threadIdx.x < count) p[from + threadIdx.x] = 0;
}

// ---------------------------------------------------------------------------------------------
// Merge tuples.  Built in slot order with coalesced reads of S and rank (thread g owns text
// positions 3g..3g+2), then gathered into SA12 order — one 16-byte gather per sample suffix
// instead of the 4-6 scattered reads per output of lib.rs:136-162.
// rank = 1-based rank of sample suffixes in slot order, with >= 3 zero words after rank[m02-1].
// ---------------------------------------------------------------------------------------------
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_build_tuples(Sym S, u32 m, u32 m0, u32 m02,
                                                        const u32 *__restrict__ rank, Tup12 *__restrict__ tslot) {
  const bool dummy = (m % 3) == 1;
  __shared__ uint16_t lcode[256];
  S.stage(lcode);
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < m0; g += gridDim.x * kBlock) {
    const u32 j = 3 * g;
    u32 q[4]; S.get4(j, lcode, q);
    const u32 s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3];
    // mod-1 sample at j+1 (slot g); exists for all g < m0 (dummy when j+1 == m)
    Tup12 a;
    a.pos = j + 1; a.c0 = s1; a.cx = s0;
    a.r = (j + 2 < m) ? rank[m0 + g] : 0u;                       // rank of suffix j+2 (mod 2)
    tslot[g] = a;
    if (j + 2 < m) {                                            // mod-2 sample at j+2 (slot m0+g)
      Tup12 c;
      c.pos = j + 2; c.c0 = s2; c.cx = s3;
      const bool has = (j + 4 < m) || (dummy && j + 4 == m);    // suffix j+4 is mod 1, slot g+1
      c.r = has ? rank[g + 1] : 0u;
      tslot[m0 + g] = c;
    }
  }
}
// Block b gathers the contiguous chunk [b*chunk, (b+1)*chunk) and also histograms, for its mod-1
// entries, the low key byte of the mod-0 tuple each of them yields (c_prev - 1): that is the
// up-sweep of the fused "select mod-0 + first radix pass" below, for free.
__global__ __launch_bounds__(kBlock) void k_gather_tuples(const Tup12 *__restrict__ tslot,
                                                         const u32 *__restrict__ sa12, u32 n, u32 chunk,
                                                         u32 nchunks, Tup12 *__restrict__ out,
                                                         u32 *__restrict__ table /*[256][nchunks]*/) {
  __shared__ u32 hist[kWaves][256];
#pragma unroll
  for (int w = 0; w < kWaves; w++) hist[w][threadIdx.x] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 i = begin + threadIdx.x;
  // the index stream and the output stream are touched once: non-temporal
  typedef u32 u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *tv = reinterpret_cast<const u32x4 *>(tslot);
  u32x4 *ov = reinterpret_cast<u32x4 *>(out);
  // 4 independent 16-byte gathers in flight per thread;  .x = pos, .w = cx
  for (; i + 3 * kBlock < end; i += 4 * kBlock) {
    const u32 s0 = __builtin_nontemporal_load(&sa12[i]), s1 = __builtin_nontemporal_load(&sa12[i + kBlock]);
    const u32 s2 = __builtin_nontemporal_load(&sa12[i + 2 * kBlock]), s3 = __builtin_nontemporal_load(&sa12[i + 3 * kBlock]);
    const u32x4 a = tv[s0], b = tv[s1], c = tv[s2], d = tv[s3];
    __builtin_nontemporal_store(a, &ov[i]); __builtin_nontemporal_store(b, &ov[i + kBlock]);
    __builtin_nontemporal_store(c, &ov[i + 2 * kBlock]); __builtin_nontemporal_store(d, &ov[i + 3 * kBlock]);
    if (a.x % 3 == 1) atomicAdd(&myh[(a.w - 1u) & 255u], 1u);
    if (b.x % 3 == 1) atomicAdd(&myh[(b.w - 1u) & 255u], 1u);
    if (c.x % 3 == 1) atomicAdd(&myh[(c.w - 1u) & 255u], 1u);
    if (d.x % 3 == 1) atomicAdd(&myh[(d.w - 1u) & 255u], 1u);
  }
  for (; i < end; i += kBlock) {
    const u32x4 a = tv[sa12[i]]; ov[i] = a;
    if (a.x % 3 == 1) atomicAdd(&myh[(a.w - 1u) & 255u], 1u);
  }
  __syncthreads();
  u32 sum = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) sum += hist[w][threadIdx.x];
  table[threadIdx.x * nchunks + blockIdx.x] = sum;
}

// ---------------------------------------------------------------------------------------------
// Step 2 (lib.rs:118-125): order-preserving selection of the mod-1 entries of SA12; each yields
// the mod-0 suffix one position to the left, already ordered by rank of suffix j+1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_mod1(u32 pos) { return pos % 3 == 1; }

// Loader of the fused Step-2 pass: element i of the sorted sample tuples yields a mod-0 tuple iff it
// is a mod-1 suffix; r1 = i+1 is the rank of suffix j+1, so the stream is already ordered by it.
struct Mod0Loader {
  const Tup12 *t;
  __device__ __forceinline__ bool load(u32 i, Tup0 &z) const {
    const Tup12 a = t[i];
    if (!is_mod1(a.pos)) return false;
    z.pos = a.pos - 1; z.c0 = a.cx; z.c1 = a.c0; z.r1 = i + 1; z.r2 = a.r;
    return true;
  }
};

// ---------------------------------------------------------------------------------------------
// Step 3 (lib.rs:131-192): merge of SA12 and SA0 as a merge-path merge.
// Comparator = leq2 / leq3 of lib.rs:3-11 in Kärkkäinen–Sanders argument order (the reference's
// leq3 parameter list is scrambled, lib.rs:9 vs :154-161).  Suffixes are distinct, so < == <=.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sample_before(const Tup12 &a, const Tup0 &z) {
  if (is_mod1(a.pos)) return (a.c0 < z.c0) || (a.c0 == z.c0 && a.r <= z.r1);               // leq2
  return (a.c0 < z.c0) || (a.c0 == z.c0 && ((a.cx < z.c1) || (a.cx == z.c1 && a.r <= z.r2))); // leq3
}

// Merge-path split points: part[t] = number of A elements among the first t*tile outputs.
// Two levels: `coarse` (optional) holds the split of every `ratio`-th tile boundary, which bounds the
// binary search of the tiles in between to a window of ratio*tile elements (L2-resident, ~half the
// dependent steps) — the unbounded search over 10^9 elements fetched 15 GB per build.
__global__ __launch_bounds__(kBlock) void k_merge_partition(const Tup12 *__restrict__ A, u32 nA,
                                                           const Tup0 *__restrict__ B, u32 nB, u32 ntiles,
                                                           u32 tile, const u32 *__restrict__ coarse, u32 ratio,
                                                           u32 *__restrict__ part /*[ntiles+1]*/) {
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t > ntiles) return;
  const u32 total = nA + nB;
  const u32 diag = (u32)min((u64)t * tile, (u64)total);
  u32 lo = diag > nB ? diag - nB : 0u, hi = min(diag, nA);
  if (coarse) {
    const u32 cidx = t / ratio;
    if (cidx * ratio == t) { part[t] = coarse[cidx]; return; }     // on a coarse boundary
    lo = max(lo, coarse[cidx]);
    hi = min(hi, coarse[cidx + 1]);
  }
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);     // (lo + hi) would overflow u32 beyond 2^31 samples
    if (sample_before(A[mid], B[diag - 1 - mid])) lo = mid + 1; else hi = mid;
  }
  part[t] = lo;
}

// out_sa[k] = text position of the k-th smallest suffix (coalesced); out_pairs[k] = (pos, k+1) feeds
// the windowed inversion that gives the parent level rank[pos] = k+1 (R[SA12[i]] = i+1, lib.rs:106-108).
// NT threads, VT outputs per thread; the tile's inputs are staged in LDS, outputs are staged in LDS
// too so that global stores are coalesced.
// LDS image of a tile: sample tuples as 16-byte words (pos, r, c0, cx); mod-0 tuples split into a
// 16-byte comparison key (c0, c1, r1, r2) and a separate pos array, so that every comparison is two
// ds_read_b128 (the packed 20-byte Tup0 would be five ds_read_b32).
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool sample_before4(const u32x4 a /*pos,r,c0,cx*/, const u32x4 z /*c0,c1,r1,r2*/) {
  if (is_mod1(a.x)) return (a.z < z.x) || (a.z == z.x && a.y <= z.z);                                   // leq2
  return (a.z < z.x) || (a.z == z.x && ((a.w < z.y) || (a.w == z.y && a.y <= z.w)));                    // leq3
}
template <int NT, int VT>
struct MergeSmem { static constexpr size_t kBytes = (16 + 16 + 4) * (size_t)(NT * VT) + 64; };

template <int NT, int VT>
__global__ __launch_bounds__(NT) void k_merge(const Tup12 *__restrict__ A, u32 nA, const Tup0 *__restrict__ B, u32 nB,
                                             const u32 *__restrict__ part, u32 *__restrict__ out_sa,
                                             Rec8 *__restrict__ out_pairs) {
  constexpr u32 kTile = NT * VT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4 *sa = reinterpret_cast<u32x4 *>(smem);
  u32x4 *sbk = reinterpret_cast<u32x4 *>(smem + 16 * kTile);
  u32 *sbpos = reinterpret_cast<u32 *>(smem + 32 * kTile);
  const u32 total = nA + nB;
  const u32 d0 = blockIdx.x * kTile;
  const u32 d1 = min(d0 + kTile, total);
  const u32 a0 = part[blockIdx.x], a1 = part[blockIdx.x + 1];
  const u32 b0 = d0 - a0, b1 = d1 - a1;
  const u32 na = a1 - a0, nb = b1 - b0;
  const u32x4 *Av = reinterpret_cast<const u32x4 *>(A);
  for (u32 i = threadIdx.x; i < na; i += NT) sa[i] = Av[a0 + i];
  for (u32 i = threadIdx.x; i < nb; i += NT) {
    const Tup0 z = B[b0 + i];
    u32x4 k; k.x = z.c0; k.y = z.c1; k.z = z.r1; k.w = z.r2;
    sbk[i] = k; sbpos[i] = z.pos;
  }
  __syncthreads();
  const u32 dl = min(threadIdx.x * (u32)VT, na + nb);
  u32 lo = dl > nb ? dl - nb : 0u, hi = min(dl, na);
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);     // (lo + hi) would overflow u32 beyond 2^31 samples
    if (sample_before4(sa[mid], sbk[dl - 1 - mid])) lo = mid + 1; else hi = mid;
  }
  u32 ai = lo, bi = dl - lo;
  u32 outp[VT];
#pragma unroll
  for (int v = 0; v < VT; v++) {
    const u32 k = dl + v;
    outp[v] = 0;
    if (k < na + nb) {
      const bool takeA = (bi >= nb) || (ai < na && sample_before4(sa[ai], sbk[bi]));
      outp[v] = takeA ? sa[ai].x : sbpos[bi];
      ai += takeA ? 1u : 0u; bi += takeA ? 0u : 1u;
    }
  }
  __syncthreads();                       // inputs are dead: reuse the front of LDS as the output stage
  u32 *so = reinterpret_cast<u32 *>(smem);
#pragma unroll
  for (int v = 0; v < VT; v++) so[threadIdx.x * VT + v] = outp[v];
  __syncthreads();
  const u32 nout = d1 - d0;
  for (u32 q = threadIdx.x; q < nout; q += NT) {
    const u32 pos = so[q];
    if (out_sa) out_sa[d0 + q] = pos;
    if (out_pairs) out_pairs[d0 + q] = Rec8{pos, d0 + q + 1};
  }
}

// ---------------------------------------------------------------------------------------------
// Synthetic text generator: byte i = byte (i&7) of splitmix64(seed + (i>>3)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// kind 2: low-entropy text with deep LCPs (position-addressable): 16-byte
// cells "word1 word2.." from a 4096-word vocabulary with a skewed id distribution, newline every
// 80 bytes, and every 64 KiB block starts with probability 1/2 with a 1-8 KiB copy of an earlier span.
__device__ __forceinline__ uint8_t text_base(u64 i, u64 seed) {
  const u64 cell = i >> 4; const u32 off = (u32)(i & 15);
  if (cell % 5 == 4 && off == 15) return '\n';
  const u64 hc = splitmix64(seed + cell * 0x9E3779B97F4A7C15ull);
  const u64 a = hc & 0xFFFF, b = (hc >> 16) & 0xFFFF, c = (hc >> 32) & 0xFFFF;
  const u64 wid = (((a * b) >> 16) * c) >> 20;
  const u64 hw = splitmix64(0x5EEDull ^ (wid << 1));
  const u32 wlen = 2 + (u32)(hw % 11);
  if (off < wlen) return (uint8_t)('a' + ((hw >> (8 + 4 * off)) % 26));
  if (off == wlen) return ' ';
  const u64 wid2 = (((hc >> 48) & 0xFFF) * ((hc >> 40) & 0xFF)) >> 8;
  const u64 hw2 = splitmix64(0x5EEDull ^ (wid2 << 1));
  const u32 wlen2 = 2 + (u32)(hw2 % 11), o2 = off - wlen - 1;
  if (o2 < wlen2) return (uint8_t)('a' + ((hw2 >> (8 + 4 * o2)) % 26));
  return ' ';
}
__device__ __forceinline__ uint8_t text_byte(u64 i, u64 seed) {
  const u64 block = i >> 16, within = i & 0xFFFF;
  if (block > 0) {
    const u64 hb = splitmix64((seed ^ 0xB10Cull) + block * 0xD1B54A32D192ED03ull);
    if (hb & 1) {
      const u64 len = 1024 + ((hb >> 8) % 7169);
      if (within < len) {
        const u64 sb = (hb >> 24) % block, so = (hb >> 44) % (65536 - 8192);
        return text_base(sb * 65536 + so + within, seed);
      }
    }
  }
  return text_base(i, seed);
}
__device__ __forceinline__ uint8_t gen_byte(u64 gi, u64 seed, int kind) {
  if (kind == 2) return text_byte(gi, seed);
  if (kind == 0) return (uint8_t)(splitmix64(seed + (gi >> 3)) >> (8 * (gi & 7)));
  const u32 code = (u32)(splitmix64(seed + (gi >> 5)) >> (2 * (gi & 31))) & 3u;
  return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
}
// t[i] = byte (off+i) of the stream; one thread per 8 output bytes, 8-byte stores
__global__ __launch_bounds__(kBlock) void k_generate(uint8_t *t, u64 n, u64 seed, int kind, u64 off) {
  const u64 nw = (n + 7) / 8;
  for (u64 wi = blockIdx.x * (u64)kBlock + threadIdx.x; wi < nw; wi += (u64)gridDim.x * kBlock) {
    u64 v = 0;
    if (kind == 0 && (off & 7) == 0) {
      v = splitmix64(seed + ((off + wi * 8) >> 3));
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) v |= (u64)gen_byte(off + wi * 8 + j, seed, kind) << (8 * j);
    }
    if (wi * 8 + 8 <= n) *reinterpret_cast<u64 *>(t + wi * 8) = v;
    else for (u64 j = wi * 8; j < n; j++) t[j] = (uint8_t)(v >> (8 * (j & 7)));
  }
}

// ---------------------------------------------------------------------------------------------
// GPU verifier = sufcheck() of crates/cdivsufsort/c-sources/utils.c:160-241 restated as parallel
// passes (equivalently sacabase::verify, sacabase/src/lib.rs:127-149).  With ISA = inverse of SA:
//   (1) range + permutation: every SA[i] in [0,n) and ISA is a bijection        (-2)
//   (2) first characters non-decreasing                                          (-3)
//   (3) for T[SA[i]] == T[SA[i+1]]: rank of suffix SA[i]+1 < rank of suffix SA[i+1]+1, the end
//       of text ranking lowest                                                   (-4)
// (1)-(3) hold iff SA is the suffix array.  err receives the smallest failing code seen
// (as in sufcheck, -2 is reported before -3 before -4).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_check_fill(const u32 *__restrict__ sa, u32 n, u32 *__restrict__ isa,
                                                      int *err) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 p = sa[i];
    if (p >= n) { atomicMax(err, 3); continue; }   // code = -(5 - v): 3 -> -2
    isa[p] = i + 1;
  }
}
__global__ __launch_bounds__(kBlock) void k_check_order(const uint8_t *__restrict__ t, const u32 *__restrict__ sa,
                                                       const u32 *__restrict__ isa, u32 n, int *err) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 p = sa[i];
    if (p >= n) continue;
    if (isa[p] != i + 1) { atomicMax(err, 3); continue; }   // not a permutation
    if (i + 1 >= n) continue;
    const u32 q = sa[i + 1];
    if (q >= n) continue;
    const uint8_t cp = t[p], cq = t[q];
    if (cp > cq) { atomicMax(err, 2); continue; }           // -3
    if (cp == cq) {
      const u32 rp = (p + 1 < n) ? isa[p + 1] : 0u;
      const u32 rq = (q + 1 < n) ? isa[q + 1] : 0u;
      if (!(rp < rq)) atomicMax(err, 1);                    // -4
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_checksum(const u32 *__restrict__ sa, u32 n, u64 *out) {
  u64 acc = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    acc += splitmix64(((u64)i << 32) | sa[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}
// ---------------------------------------------------------------------------------------------
// By-products of the suffix array ("next" rows of the scope table).
// BWT: bw_transform()/divbwt() of crates/cdivsufsort/c-sources (utils.c:53-108, divsufsort.c:372-405):
//   U[0] = T[n-1]; then T[SA[i]-1] for every i with SA[i] != 0, in order; primary index = (i: SA[i]==0) + 1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_find_zero(const u32 *__restrict__ sa, u32 n, u32 *zpos) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) if (sa[i] == 0) *zpos = i;
}
__global__ __launch_bounds__(kBlock) void k_bwt(const uint8_t *__restrict__ t, const u32 *__restrict__ sa, u32 n,
                                               const u32 *__restrict__ zpos, uint8_t *__restrict__ u) {
  const u32 z = *zpos;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    if (i == z) continue;
    const uint8_t ch = t[sa[i] - 1];
    u[i < z ? i + 1 : i] = ch;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) u[0] = t[n - 1];
}

// Batched search = sacabase::longest_substring_match (crates/sacabase/src/lib.rs:39-99), one thread
// per needle, the reference's own narrowing loop (mid = len/2; needle > suffix(mid) ? right : left
// inclusive; 1-2 survivors compared by common prefix) so that (start, len) are identical.
__device__ __forceinline__ u32 d_common_prefix(const uint8_t *a, u64 la, const uint8_t *b, u64 lb) {
  const u64 l = la < lb ? la : lb;
  u64 i = 0;
  while (i < l && a[i] == b[i]) i++;
  return (u32)i;
}
__global__ __launch_bounds__(kBlock) void k_search(const uint8_t *__restrict__ t, u32 n, const u32 *__restrict__ sa,
                                                  const uint8_t *__restrict__ needles,
                                                  const int64_t *__restrict__ off, u32 q,
                                                  int64_t *__restrict__ out_start, int64_t *__restrict__ out_len) {
  const u32 id = blockIdx.x * kBlock + threadIdx.x;
  if (id >= q) return;
  const uint8_t *nd = needles + off[id];
  const u64 nl = (u64)(off[id + 1] - off[id]);
  u32 lo = 0, len = n;
  for (;;) {
    if (len == 1) {
      const u32 s = sa[lo];
      out_start[id] = s; out_len[id] = d_common_prefix(t + s, n - s, nd, nl);
      return;
    }
    if (len == 2) {
      const u32 s0 = sa[lo], s1 = sa[lo + 1];
      const u32 x = d_common_prefix(t + s0, n - s0, nd, nl), y = d_common_prefix(t + s1, n - s1, nd, nl);
      if (x > y) { out_start[id] = s0; out_len[id] = x; } else { out_start[id] = s1; out_len[id] = y; }
      return;
    }
    const u32 mid = len / 2;
    const u32 s = sa[lo + mid];
    const u64 sl = n - s;
    const u32 c = d_common_prefix(t + s, sl, nd, nl);
    // needle > suffix: first differing byte larger, or suffix is a proper prefix of the needle
    const bool gt = (c < nl && c < sl) ? (nd[c] > t[s + c]) : (nl > sl);
    if (gt) { lo += mid; len -= mid; } else { len = mid + 1; }
  }
}

__global__ __launch_bounds__(kBlock) void k_widen(const u32 *__restrict__ in, int64_t *__restrict__ out, u32 n) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = (int64_t)in[i];
}

}  // namespace dc3
