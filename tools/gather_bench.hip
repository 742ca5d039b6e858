// gather_bench.hip — what limits out[i] = table[idx[i]] (16-byte elements) on MI355X: table size, loads in flight, nt hints
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32; typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 mix(u64 x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void k_idx(u32 *idx, u32 n, u32 range) { for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) idx[i] = (u32)(mix(i) % range); }
template <class T, int U>
__global__ __launch_bounds__(256) void k_gather(const T *__restrict__ tab, const u32 *__restrict__ idx, T *__restrict__ out, u32 n, u32 chunk) {
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 i = begin + threadIdx.x;
  for (; i + (U - 1) * 256 < end; i += U * 256) {
    u32 s[U]; T v[U];
#pragma unroll
    for (int k = 0; k < U; k++) s[k] = idx[i + k * 256];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = tab[s[k]];
#pragma unroll
    for (int k = 0; k < U; k++) out[i + k * 256] = v[k];
  }
  for (; i < end; i += 256) out[i] = tab[idx[i]];
}
template <class T, int U> void run(const T *tab, const u32 *idx, T *out, u32 n, const char *label, size_t tabbytes) {
  const u32 nblocks = 2048; u32 chunk = (n + nblocks - 1) / nblocks; chunk = (chunk + 255) / 256 * 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 3; r++) { CK(hipEventRecord(a)); k_gather<T, U><<<(n + chunk - 1) / chunk, 256>>>(tab, idx, out, n, chunk); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  printf("%-8s elem=%2zuB table=%6.2f GB U=%d: %8.3f ms %6.2f Gelem/s\n", label, sizeof(T), tabbytes / 1e9, U, best, n / best / 1e6);
}
int main() {
  const u32 n = 1u << 28;
  u32 *idx; CK(hipMalloc(&idx, (size_t)n * 4));
  void *tab, *out; CK(hipMalloc(&tab, 12ull << 30)); CK(hipMalloc(&out, (size_t)n * 16)); CK(hipMemset(tab, 1, 12ull << 30));
  for (double gb : {1.0, 2.9, 5.7, 11.4}) {
    const u32 r16 = (u32)(gb * 1e9 / 16), r8 = (u32)(gb * 1e9 / 8);
    k_idx<<<2048, 256>>>(idx, n, r16); CK(hipDeviceSynchronize());
    run<u32x4, 1>((u32x4 *)tab, idx, (u32x4 *)out, n, "g16", (size_t)r16 * 16);
    run<u32x4, 4>((u32x4 *)tab, idx, (u32x4 *)out, n, "g16", (size_t)r16 * 16);
    run<u32x4, 8>((u32x4 *)tab, idx, (u32x4 *)out, n, "g16", (size_t)r16 * 16);
    k_idx<<<2048, 256>>>(idx, n, r8); CK(hipDeviceSynchronize());
    run<u32x2, 4>((u32x2 *)tab, idx, (u32x2 *)out, n, "g8", (size_t)r8 * 8);
    run<u32x2, 8>((u32x2 *)tab, idx, (u32x2 *)out, n, "g8", (size_t)r8 * 8);
  }
  return 0;
}
