// GPU box probe: does ds_add_rtn resolve the lanes of ONE wave instruction that hit the same LDS address in ascending
// lane order?  (Needed if a stable radix ranking is to use one returning LDS atomic per record instead of ballots.)
// Prints the number of violations over many random digit patterns, for 1..16 waves hammering their own counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void probe(const uint32_t *digits, uint32_t *ranks, int rounds, int nb) {
  extern __shared__ uint32_t cnt[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t *my = cnt + w * nb;
  for (int j = lane; j < nb; j += 64) my[j] = 0;
  __syncthreads();
  const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * rounds;
  for (int r = 0; r < rounds; r++) {
    const uint32_t d = digits[base + r];
    ranks[base + r] = __hip_atomic_fetch_add(&my[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}
int main() {
  const int blocks = 1024, threads = 1024, rounds = 12;
  for (int nb : {1, 2, 7, 64, 512}) {
    const size_t n = (size_t)blocks * threads * rounds;
    std::vector<uint32_t> h(n), r(n);
    uint64_t s = 88172645463325252ull + nb;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % nb); }
    uint32_t *dd, *dr;
    hipMalloc(&dd, n * 4); hipMalloc(&dr, n * 4);
    hipMemcpy(dd, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 16 * 512 * 4, 0, dd, dr, rounds, nb);
    hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    // expected: per wave, per round (in order), lanes ascending within equal digit
    size_t bad = 0;
    for (int b = 0; b < blocks; b++)
      for (int w = 0; w < threads / 64; w++) {
        std::vector<uint32_t> c(nb, 0);
        for (int rd = 0; rd < rounds; rd++)
          for (int lane = 0; lane < 64; lane++) {
            const size_t i = ((size_t)b * threads + w * 64 + lane) * rounds + rd;
            if (r[i] != c[h[i]]++) bad++;
          }
      }
    printf("nb=%d violations=%zu of %zu\n", nb, bad, n);
    hipFree(dd); hipFree(dr);
  }
  return 0;
}
