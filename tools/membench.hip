// membench.hip — MI355X access-pattern microbenchmark that drives the DC3 data-layout choices
// (DESIGN.md §"Why records are sorted, not gathered").  Not part of the product path.
//   build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t bij(uint32_t x, int k) {  // bijection on k bits
  const uint32_t m = (k == 32) ? 0xffffffffu : ((1u << k) - 1);
  const int h = k / 2;
  x = (x * 0x9E3779B1u) & m; x ^= x >> h;
  x = (x * 0x85EBCA6Bu) & m; x ^= x >> h;
  x = (x * 0xC2B2AE35u) & m; x ^= x >> h;
  return x;
}

__global__ void k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
template <typename T>
__global__ void k_scatter(T* __restrict__ out, uint32_t n, int k, uint32_t wshift) {
  // destination = window(i) + bij(i within window); wshift = log2(window elements) (== k for one window)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t w = i >> wshift, o = i & ((1u << wshift) - 1);
    uint32_t d = (w << wshift) | bij(o, wshift);
    T v; 
    if constexpr (sizeof(T) == 4) v = i; else if constexpr (sizeof(T) == 8) v = make_uint2(i, i); else v = make_uint4(i, i, i, i);
    out[d] = v;
  }
}
template <typename T>
__global__ void k_gather(const T* __restrict__ in, uint32_t* __restrict__ sink, uint32_t n, int k, uint32_t wshift) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t w = i >> wshift, o = i & ((1u << wshift) - 1);
    uint32_t d = (w << wshift) | bij(o, wshift);
    T v = in[d];
    if constexpr (sizeof(T) == 4) acc += v; else if constexpr (sizeof(T) == 8) acc += v.x ^ v.y; else acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
// gather with coalesced output write (closer to the real use: out[i] = table[idx(i)])
template <typename T>
__global__ void k_gather_store(const T* __restrict__ in, T* __restrict__ out, uint32_t n, uint32_t wshift) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t w = i >> wshift, o = i & ((1u << wshift) - 1);
    uint32_t d = (w << wshift) | bij(o, wshift);
    out[i] = in[d];
  }
}
__global__ void k_atomic_hist(uint32_t* __restrict__ bins, uint32_t n, int kbins) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    atomicAdd(&bins[bij(i, 32) >> (32 - kbins)], 1u);
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d mem=%.1f GiB\n", p.name, p.multiProcessorCount, p.totalGlobalMem / 1073741824.0);
  const int grid = p.multiProcessorCount * 8, block = 256;
  size_t bytes = 4ull << 30;
  void *A, *B; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMemset(B, 2, bytes));
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  { size_t n16 = (2ull << 30) / 16; float ms = timeit([&] { k_copy16<<<grid, block>>>((uint4*)A, (uint4*)B, n16); });
    printf("copy16 2GiB: %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * (2ull << 30) / ms / 1e6); }
  // full-range random scatter / gather, element sizes 4/8/16, table = 2^k elements
  for (int k : {24, 26, 28, 29}) {
    uint32_t n = 1u << k;
    { float ms = timeit([&] { k_scatter<uint32_t><<<grid, block>>>((uint32_t*)A, n, k, k); });
      printf("scatter4  table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s  useful %.0f GB/s\n", n * 4.0 / 1048576, ms, n / ms / 1e6, n * 4.0 / ms / 1e6); }
    { float ms = timeit([&] { k_gather<uint32_t><<<grid, block>>>((uint32_t*)A, sink, n, k, k); });
      printf("gather4   table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 4.0 / 1048576, ms, n / ms / 1e6); }
    { float ms = timeit([&] { k_gather_store<uint32_t><<<grid, block>>>((uint32_t*)A, (uint32_t*)B, n, k); });
      printf("gather4+st table=%5.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 4.0 / 1048576, ms, n / ms / 1e6); }
    if (k <= 28) {
      { float ms = timeit([&] { k_scatter<uint2><<<grid, block>>>((uint2*)A, n, k, k); });
        printf("scatter8  table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 8.0 / 1048576, ms, n / ms / 1e6); }
      { float ms = timeit([&] { k_gather<uint2><<<grid, block>>>((uint2*)A, sink, n, k, k); });
        printf("gather8   table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 8.0 / 1048576, ms, n / ms / 1e6); }
    }
    if (k <= 27 || k == 28) {
      { float ms = timeit([&] { k_scatter<uint4><<<grid, block>>>((uint4*)A, n, k, k); });
        printf("scatter16 table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 16.0 / 1048576, ms, n / ms / 1e6); }
      { float ms = timeit([&] { k_gather<uint4><<<grid, block>>>((uint4*)A, sink, n, k, k); });
        printf("gather16  table=%6.0f MiB: %8.3f ms  %6.2f Gelem/s\n", n * 16.0 / 1048576, ms, n / ms / 1e6); }
    }
  }
  // windowed scatter/gather: n = 2^29 elements of 4 B (2 GiB table), random only inside windows of 2^w elements
  for (int w : {12, 16, 20, 22, 24, 25, 26}) {
    uint32_t n = 1u << 29;
    float ms = timeit([&] { k_scatter<uint32_t><<<grid, block>>>((uint32_t*)A, n, 29, w); });
    float mg = timeit([&] { k_gather<uint32_t><<<grid, block>>>((uint32_t*)A, sink, n, 29, w); });
    printf("windowed 4B window=%8.2f MiB: scatter %8.3f ms %6.2f Gelem/s | gather %8.3f ms %6.2f Gelem/s\n", (4.0 * (1u << w)) / 1048576, ms, n / ms / 1e6, mg, n / mg / 1e6);
  }
  for (int w : {16, 20, 22, 23, 24}) {
    uint32_t n = 1u << 27;
    float ms = timeit([&] { k_scatter<uint4><<<grid, block>>>((uint4*)A, n, 27, w); });
    float mg = timeit([&] { k_gather<uint4><<<grid, block>>>((uint4*)A, sink, n, 27, w); });
    printf("windowed 16B window=%8.2f MiB: scatter %8.3f ms %6.2f Gelem/s | gather %8.3f ms %6.2f Gelem/s\n", (16.0 * (1u << w)) / 1048576, ms, n / ms / 1e6, mg, n / mg / 1e6);
  }
  // global atomic histogram with 2^kbins counters (the reference's K+1 counter array on device)
  for (int kb : {8, 16, 20, 24}) {
    uint32_t n = 1u << 28; CK(hipMemset(A, 0, (size_t)4 << kb));
    float ms = timeit([&] { k_atomic_hist<<<grid, block>>>((uint32_t*)A, n, kb); }, 3);
    printf("atomic hist bins=2^%d: %8.3f ms %6.2f Gelem/s\n", kb, ms, n / ms / 1e6);
  }
  return 0;
}
