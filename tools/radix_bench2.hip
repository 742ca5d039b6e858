// radix_bench2.hip — down-sweep variants (pipelined LDS atomics, next-tile prefetch) for 8/16/20-byte records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "dc3_kernels.cuh"
using namespace dc3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct KeyOf { __device__ static u32 get(const Rec8 &r) { return r.key; } __device__ static u32 get(const Rec16 &r) { return r.k0; } __device__ static u32 get(const Tup0 &r) { return r.c0; } };
struct DigAny { u32 shift; template <class R> __device__ __forceinline__ u32 operator()(const R &r) const { return (KeyOf::get(r) >> shift) & 255u; } };

// MODE 0: baseline ranking (read-modify-write chain), 1: pipelined ds_add_rtn + shfl, PF: prefetch next tile
template <class Rec, int IPT, int NW, int MODE, bool PF>
__global__ __launch_bounds__(NW * 64) void k_down2(const Rec *__restrict__ in, Rec *__restrict__ out, u32 n, u32 chunk,
                                                  u32 nchunks, DigAny dig, const u32 *__restrict__ table,
                                                  const u32 *__restrict__ digit_base) {
  constexpr int kB = NW * 64, kTile = kB * IPT, kWItems = 64 * IPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *srec = reinterpret_cast<Rec *>(smem);
  u32 *wcnt = reinterpret_cast<u32 *>(smem + sizeof(Rec) * kTile);
  u32 *dbase = wcnt + NW * 256, *texcl = dbase + 256, *tmp = texcl + 256;
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  if (tid < 256) dbase[tid] = digit_base[tid] + table[tid * nchunks + blockIdx.x];
  u32 *mycnt = wcnt + w * 256;
  Rec r[IPT], rn[IPT];
  if (PF) {
#pragma unroll
    for (int k = 0; k < IPT; k++) { const u32 t = w * kWItems + k * 64 + lane; if (begin + t < end) rn[k] = in[begin + t]; }
  }
  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 nvalid = min((u32)kTile, end - tile);
#pragma unroll
    for (int j = 0; j < 4; j++) mycnt[lane + 64 * j] = 0;
    u32 d[IPT], rk[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if (PF) r[k] = rn[k]; else if (t < nvalid) r[k] = in[tile + t];
      d[k] = (t < nvalid) ? dig(r[k]) : 255u;
    }
    if (PF) {
      const u32 nt = tile + kTile;
#pragma unroll
      for (int k = 0; k < IPT; k++) { const u32 t = w * kWItems + k * 64 + lane; if (nt + t < end) rn[k] = in[nt + t]; }
    }
    if (MODE == 0) {
      volatile u32 *vc = mycnt;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        u64 peers = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) { const bool one = (d[k] >> bit) & 1u; const u64 mk = __ballot(one); peers &= one ? mk : ~mk; }
        const u32 below = mbcnt(peers), cnt = __popcll(peers);
        const u32 base = vc[d[k]];
        rk[k] = base + below;
        if (below == cnt - 1) vc[d[k]] = base + cnt;
      }
    } else {
      u32 below[IPT], leader[IPT], old[IPT];
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        u64 peers = ~0ull;
#pragma unroll
        for (int bit = 0; bit < 8; bit++) { const bool one = (d[k] >> bit) & 1u; const u64 mk = __ballot(one); peers &= one ? mk : ~mk; }
        below[k] = mbcnt(peers);
        leader[k] = __ffsll((unsigned long long)peers) - 1;
        old[k] = 0;
        if (below[k] == 0) old[k] = atomicAdd(&mycnt[d[k]], (u32)__popcll(peers));
      }
#pragma unroll
      for (int k = 0; k < IPT; k++) rk[k] = __shfl(old[k], leader[k]) + below[k];
    }
    __syncthreads();
    u32 tot = 0;
    if (tid < 256) {
#pragma unroll
      for (int i = 0; i < NW; i++) { const u32 c = wcnt[i * 256 + tid]; wcnt[i * 256 + tid] = tot; tot += c; }
    }
    u32 dummy;
    const u32 ex = block_excl_scan<NW>(tid < 256 ? tot : 0u, tmp, dummy);
    if (tid < 256) texcl[tid] = ex;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if (t < nvalid) srec[texcl[d[k]] + wcnt[w * 256 + d[k]] + rk[k]] = r[k];
    }
    __syncthreads();
    for (u32 q = tid; q < nvalid; q += kB) {
      const Rec x = srec[q];
      const u32 dd = dig(x);
      out[dbase[dd] + (q - texcl[dd])] = x;
    }
    __syncthreads();
    if (tid < 256) dbase[tid] += tot;
  }
}

template <class Rec>
__global__ __launch_bounds__(256) void k_up2(const Rec *__restrict__ in, u32 n, u32 chunk, u32 nchunks, DigAny dig, u32 *__restrict__ table) {
  __shared__ u32 hist[4][256];
  const u32 tid = threadIdx.x;
  for (int w = 0; w < 4; w++) hist[w][tid] = 0;
  __syncthreads();
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 *myh = hist[tid >> 6];
  for (u32 i = begin + tid; i < end; i += 256) atomicAdd(&myh[dig(in[i])], 1u);
  __syncthreads();
  table[tid * nchunks + blockIdx.x] = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
}
__device__ void setrec(Rec8 &r, u32 key, u32 i) { r.key = key; r.val = i; }
__device__ void setrec(Rec16 &r, u32 key, u32 i) { r.k0 = key; r.k1 = i * 7; r.k2 = 3; r.pos = i; }
__device__ void setrec(Tup0 &r, u32 key, u32 i) { r.pos = i; r.c0 = key; r.c1 = 1; r.r1 = i; r.r2 = i ^ 5; }
template <class Rec> __global__ void k_fill2(Rec *r, u32 n) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) setrec(r[i], (u32)splitmix64(i), i);
}
static u32 hkey(const Rec8 &r) { return r.key; } static u32 hkey(const Rec16 &r) { return r.k0; } static u32 hkey(const Tup0 &r) { return r.c0; }
static u32 hid(const Rec8 &r) { return r.val; } static u32 hid(const Rec16 &r) { return r.pos; } static u32 hid(const Tup0 &r) { return r.pos; }

template <class Rec, int IPT, int NW, int MODE, bool PF>
void run(const Rec *in, Rec *out, u32 n, u32 *table, u32 *dbase, bool check, const char *label) {
  constexpr int kTile = NW * 64 * IPT;
  const u32 target_blocks = 2048;
  u32 chunk = (n + target_blocks - 1) / target_blocks; chunk = (chunk + kTile - 1) / kTile * kTile;
  const u32 nchunks = (n + chunk - 1) / chunk;
  size_t smem = sizeof(Rec) * kTile + 4 * (NW * 256 + 512 + 32);
  auto kern = k_down2<Rec, IPT, NW, MODE, PF>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  DigAny dig; dig.shift = 8;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    hipLaunchKernelGGL((k_up2<Rec>), dim3(nchunks), dim3(256), 0, 0, in, n, chunk, nchunks, dig, table);
    hipLaunchKernelGGL(k_scan_rows, dim3(256), dim3(kBlock), 0, 0, table, nchunks, dbase);
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, 0, dbase, 256u, (u32 *)nullptr);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(nchunks), dim3(NW * 64), smem, 0, in, out, n, chunk, nchunks, dig, table, dbase);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
  }
  printf("%-10s rec=%2zuB IPT=%2d NW=%2d mode=%d pf=%d smem=%6zu: %.3f ms  %.0f GB/s moved  %.1f Gelem/s\n", label, sizeof(Rec), IPT, NW, MODE, (int)PF, smem, best, 2.0 * sizeof(Rec) * n / best / 1e6, n / best / 1e6);
  if (check) {
    std::vector<Rec> h(n), g(n);
    CK(hipMemcpy(h.data(), in, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g.data(), out, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    std::stable_sort(h.begin(), h.end(), [&](const Rec &x, const Rec &y) { return ((hkey(x) >> 8) & 255) < ((hkey(y) >> 8) & 255); });
    size_t bad = 0; for (u32 i = 0; i < n; i++) if (hid(h[i]) != hid(g[i]) || hkey(h[i]) != hkey(g[i])) bad++;
    printf("   check n=%u: %zu mismatches\n", n, bad);
  }
}

template <class Rec> void suite(u32 n, bool chk, const char *label);
template <> void suite<Rec8>(u32 n, bool chk, const char *label) {
  Rec8 *A, *B; u32 *table, *dbase;
  CK(hipMalloc(&A, sizeof(Rec8) * (size_t)n)); CK(hipMalloc(&B, sizeof(Rec8) * (size_t)n)); CK(hipMalloc(&table, 4u * 256 * 16384)); CK(hipMalloc(&dbase, 1024));
  hipLaunchKernelGGL(k_fill2<Rec8>, dim3(2048), dim3(256), 0, 0, A, n); CK(hipDeviceSynchronize());
  run<Rec8, 16, 16, 0, false>(A, B, n, table, dbase, chk, label);
  run<Rec8, 16, 16, 1, false>(A, B, n, table, dbase, chk, label);
  run<Rec8, 16, 16, 0, true>(A, B, n, table, dbase, chk, label);
  run<Rec8, 16, 16, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec8, 8, 16, 1, false>(A, B, n, table, dbase, chk, label);
  run<Rec8, 8, 16, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec8, 16, 8, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec8, 8, 8, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec8, 12, 16, 1, true>(A, B, n, table, dbase, chk, label);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(table)); CK(hipFree(dbase));
}
template <> void suite<Rec16>(u32 n, bool chk, const char *label) {
  Rec16 *A, *B; u32 *table, *dbase;
  CK(hipMalloc(&A, sizeof(Rec16) * (size_t)n)); CK(hipMalloc(&B, sizeof(Rec16) * (size_t)n)); CK(hipMalloc(&table, 4u * 256 * 16384)); CK(hipMalloc(&dbase, 1024));
  hipLaunchKernelGGL(k_fill2<Rec16>, dim3(2048), dim3(256), 0, 0, A, n); CK(hipDeviceSynchronize());
  run<Rec16, 8, 16, 0, false>(A, B, n, table, dbase, chk, label);
  run<Rec16, 8, 16, 1, false>(A, B, n, table, dbase, chk, label);
  run<Rec16, 8, 16, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec16, 4, 16, 1, true>(A, B, n, table, dbase, chk, label);
  run<Rec16, 8, 8, 1, true>(A, B, n, table, dbase, chk, label);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(table)); CK(hipFree(dbase));
}
template <> void suite<Tup0>(u32 n, bool chk, const char *label) {
  Tup0 *A, *B; u32 *table, *dbase;
  CK(hipMalloc(&A, sizeof(Tup0) * (size_t)n)); CK(hipMalloc(&B, sizeof(Tup0) * (size_t)n)); CK(hipMalloc(&table, 4u * 256 * 16384)); CK(hipMalloc(&dbase, 1024));
  hipLaunchKernelGGL(k_fill2<Tup0>, dim3(2048), dim3(256), 0, 0, A, n); CK(hipDeviceSynchronize());
  run<Tup0, 6, 16, 0, false>(A, B, n, table, dbase, chk, label);
  run<Tup0, 6, 16, 1, false>(A, B, n, table, dbase, chk, label);
  run<Tup0, 6, 16, 1, true>(A, B, n, table, dbase, chk, label);
  run<Tup0, 4, 16, 1, true>(A, B, n, table, dbase, chk, label);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(table)); CK(hipFree(dbase));
}
int main(int argc, char **argv) {
  const u32 n = argc > 1 ? (u32)atol(argv[1]) : (1u << 28);
  const bool chk = n <= (1u << 22);
  suite<Rec8>(n, chk, "Rec8"); suite<Rec16>(n, chk, "Rec16"); suite<Tup0>(n, chk, "Tup0");
  return 0;
}
