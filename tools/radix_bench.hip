// radix_bench.hip — tuning harness for the stable 8-bit radix pass (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -I stringsearch_amd/csrc tools/radix_bench.hip -o tools/radix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "dc3_kernels.cuh"
using namespace dc3;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NW>
__device__ __forceinline__ u32 block_excl_scan_nw(u32 v, u32 *tmp, u32 &total) { return block_excl_scan<NW>(v, tmp, total); }

// generalized down-sweep: NW waves per block, IPT items per thread, optional LDS reorder
template <class Rec, class Dig, int IPT, int NW, bool REORDER>
__global__ __launch_bounds__(NW * 64) void k_down(const Rec *__restrict__ in, Rec *__restrict__ out, u32 n, u32 chunk,
                                                 u32 nchunks, Dig dig, const u32 *__restrict__ table) {
  constexpr int kB = NW * 64;
  constexpr int kTile = kB * IPT;
  constexpr int kWItems = 64 * IPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *srec = reinterpret_cast<Rec *>(smem);
  u32 *wcnt = reinterpret_cast<u32 *>(smem + (REORDER ? sizeof(Rec) * kTile : 0));
  u32 *dbase = wcnt + NW * 256;
  u32 *texcl = dbase + 256;
  u32 *tmp = texcl + 256;
  const u32 tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const u32 begin = blockIdx.x * chunk;
  const u32 end = min(n, begin + chunk);
  if (tid < 256) dbase[tid] = table[tid * nchunks + blockIdx.x];
  volatile u32 *mycnt = wcnt + w * 256;
  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 nvalid = min((u32)kTile, end - tile);
#pragma unroll
    for (int j = 0; j < 4; j++) mycnt[lane + 64 * j] = 0;
    Rec r[IPT]; u32 d[IPT], rk[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if (t < nvalid) { r[k] = in[tile + t]; d[k] = dig(r[k]); } else d[k] = 255u;
    }
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      u64 peers = ~0ull;
#pragma unroll
      for (int bit = 0; bit < 8; bit++) { const bool one = (d[k] >> bit) & 1u; const u64 mk = __ballot(one); peers &= one ? mk : ~mk; }
      const u32 below = mbcnt(peers), cnt = __popcll(peers);
      const u32 base = mycnt[d[k]];
      rk[k] = base + below;
      if (below == cnt - 1) mycnt[d[k]] = base + cnt;
    }
    __syncthreads();
    u32 tot = 0;
    if (tid < 256) {
#pragma unroll
      for (int i = 0; i < NW; i++) { const u32 c = wcnt[i * 256 + tid]; wcnt[i * 256 + tid] = tot; tot += c; }
    }
    if (REORDER) {
      u32 dummy;
      // exclusive scan over 256 digit totals: threads >= 256 contribute 0
      const u32 ex = block_excl_scan_nw<NW>(tid < 256 ? tot : 0u, tmp, dummy);
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
    } else {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const u32 t = w * kWItems + k * 64 + lane;
        if (t < nvalid) out[dbase[d[k]] + wcnt[w * 256 + d[k]] + rk[k]] = r[k];
      }
      __syncthreads();
    }
    if (tid < 256) dbase[tid] += tot;
  }
}

template <class Rec, class Dig>
__global__ __launch_bounds__(256) void k_up(const Rec *__restrict__ in, u32 n, u32 chunk, u32 nchunks, Dig dig, u32 *__restrict__ table) {
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

__global__ void k_fill(Rec16 *r, u32 n) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    u64 a = splitmix64(i), b = splitmix64(i ^ 0xabcdef1234ull);
    r[i] = Rec16{(u32)a, (u32)(a >> 32), (u32)b & 0x7ff, i};
  }
}

template <int IPT, int NW, bool REORDER>
float run(const Rec16 *in, Rec16 *out, u32 n, u32 *table, u32 target_blocks, u32 byte, bool check, const char *label) {
  constexpr int kTile = NW * 64 * IPT;
  u32 chunk = (n + target_blocks - 1) / target_blocks; chunk = (chunk + kTile - 1) / kTile * kTile;
  const u32 nchunks = (n + chunk - 1) / chunk;
  size_t smem = (REORDER ? sizeof(Rec16) * kTile : 0) + 4 * (NW * 256 + 512 + 16);
  auto kern = k_down<Rec16, Rec16Byte, IPT, NW, REORDER>;
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  Rec16Byte dig; dig.p = byte;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f, bestu = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_up<Rec16, Rec16Byte>), dim3(nchunks), dim3(256), 0, 0, in, n, chunk, nchunks, dig, table);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float mu; CK(hipEventElapsedTime(&mu, a, b)); bestu = std::min(bestu, mu);
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, 0, table, 256u * nchunks, (u32 *)nullptr);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(nchunks), dim3(NW * 64), smem, 0, in, out, n, chunk, nchunks, dig, table);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
  }
  printf("%-28s IPT=%2d NW=%2d reorder=%d blocks=%5u smem=%6zu: down %.3f ms  %.0f GB/s moved | up %.3f ms\n", label, IPT, NW, (int)REORDER, nchunks, smem, best, 32.0 * n / best / 1e6, bestu);
  if (check) {
    std::vector<Rec16> h(n), g(n);
    CK(hipMemcpy(h.data(), in, sizeof(Rec16) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g.data(), out, sizeof(Rec16) * n, hipMemcpyDeviceToHost));
    std::stable_sort(h.begin(), h.end(), [&](const Rec16 &x, const Rec16 &y) { return ((x.k0 >> (8 * byte)) & 255) < ((y.k0 >> (8 * byte)) & 255); });
    size_t bad = 0; for (u32 i = 0; i < n; i++) if (h[i].pos != g[i].pos || h[i].k0 != g[i].k0) { bad++; }
    printf("   check n=%u: %zu mismatches\n", n, bad);
  }
  return best;
}

int main(int argc, char **argv) {
  const u32 n = argc > 1 ? (u32)atol(argv[1]) : (1u << 28);
  Rec16 *A, *B; u32 *table;
  CK(hipMalloc(&A, sizeof(Rec16) * (size_t)n)); CK(hipMalloc(&B, sizeof(Rec16) * (size_t)n)); CK(hipMalloc(&table, 4u * 256 * 16384));
  hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, n); CK(hipDeviceSynchronize());
  const bool chk = n <= (1u << 22);
  for (u32 tb : {2048u, 4096u}) {
    run<16, 4, true>(A, B, n, table, tb, 1, chk, "base");
    run<8, 4, true>(A, B, n, table, tb, 1, chk, "ipt8");
    run<12, 4, true>(A, B, n, table, tb, 1, chk, "ipt12");
    run<8, 8, true>(A, B, n, table, tb, 1, chk, "512thr ipt8");
    run<4, 16, true>(A, B, n, table, tb, 1, chk, "1024thr ipt4");
    run<8, 16, true>(A, B, n, table, tb, 1, chk, "1024thr ipt8");
    run<6, 8, true>(A, B, n, table, tb, 1, chk, "512thr ipt6");
    run<16, 4, false>(A, B, n, table, tb, 1, chk, "noreorder ipt16");
    run<8, 8, false>(A, B, n, table, tb, 1, chk, "noreorder 512 ipt8");
  }
  return 0;
}
