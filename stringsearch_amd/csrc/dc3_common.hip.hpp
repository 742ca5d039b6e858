// dc3_common.hip.hpp — record types, symbol readers, wave/block primitives, small scans.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
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
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

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
  __device__ __forceinline__ void get2(u32 i, const uint16_t *lds, u32 *out) const {      // symbols i, i + 1
    uint16_t w; __builtin_memcpy(&w, t + i, 2);
    out[0] = (i < m) ? (u32)lds[w & 255u] : 0u;
    out[1] = (i + 1 < m) ? (u32)lds[w >> 8] : 0u;
  }
};
struct SymU32 {
  const u32 *s; u32 m;   // s has >= 8 zero words after s[m-1]
  static constexpr bool kTable = false;
  __device__ __forceinline__ u32 get(u32 i) const { return s[i]; }
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ void get2(u32 i, const uint16_t *, u32 *out) const { out[0] = s[i]; out[1] = s[i + 1]; }
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

// v + (the value a DPP pattern brings from another lane, 0 where the pattern has no source lane or the row is masked out)
template <int kCtrl, int kRowMask>
__device__ __forceinline__ u32 dpp_add(u32 v) { return v + (u32)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xf, false); }
// inclusive prefix sums inside every row of 16 lanes (row_shr:1, 2, 4, 8)
__device__ __forceinline__ u32 row_incl_scan(u32 v) {
  v = dpp_add<0x111, 0xf>(v);
  v = dpp_add<0x112, 0xf>(v);
  v = dpp_add<0x114, 0xf>(v);
  v = dpp_add<0x118, 0xf>(v);
  return v;
}
// ... and across the wave: lane 15 of a row into the next row (rows 1 and 3), then lane 31 into rows 2 and 3.  Six data
// parallel primitive adds in the VALU where __shfl_up makes six round trips through the LDS crossbar (ds_bpermute) with a
// compare and a select each: the block scans were 14 of the 62 VALU instructions per word of the partition kernels.
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
  v = row_incl_scan(v);
  v = dpp_add<0x142, 0xa>(v);      // row_bcast:15
  v = dpp_add<0x143, 0xc>(v);      // row_bcast:31
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
// Exclusive scan of one value per thread over a block of NW <= 16 waves; tmp needs NW words of LDS.  Every wave scans the
// NW wave totals itself (one row of lanes) and reads its own offset and the block's total out of that row.
template <int NW>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32 *tmp, u32 &total) {
  static_assert(NW >= 1 && NW <= 16, "the wave totals fit one row of lanes");
  const u32 inc = wave_incl_scan(v);
  if (lane_id() == 63) tmp[wave_id()] = inc;
  __syncthreads();
  const u32 t = lane_id() < (u32)NW ? tmp[lane_id()] : 0u;
  const u32 p = row_incl_scan(t);
  const u32 w = (u32)__builtin_amdgcn_readfirstlane((int)wave_id());
  const u32 woff = (u32)__builtin_amdgcn_readlane((int)(p - t), (int)w);
  total = (u32)__builtin_amdgcn_readlane((int)p, NW - 1);
  __syncthreads();
  return woff + inc - v;
}

// The XCD this wave runs on (0..7), HW_REG_XCC_ID.  A speed/diagnostic aid only: nothing may depend on it for correctness.
__device__ __forceinline__ u32 xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
// One word per (block group, XCD): mon[g * 8 + xcd] += 1 for a block of group g — the partition kernels whose speed rests
// on "blocks with equal blockIdx % 8 share an XCD" record where they really ran (dc3hip_stats.xcd_group_hit).
__device__ __forceinline__ void xcd_note(u32 *mon, u32 g) {
  if (mon && threadIdx.x == 0) atomicAdd(&mon[(g & 7u) * 8u + xcc_id()], 1u);
}
// the placement probe of context creation: out[b] = XCD of block b
__global__ void k_xcd_probe(u32 *__restrict__ out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

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

}  // namespace dc3
