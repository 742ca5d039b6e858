// GPU box: what a first call pays before it can build (round-5 verdict item 6: dc3hip_sufsort_i32 un-warmed at 1 GiB took
// 2.4 s against 107 ms warm).  Times, on this box: hipMalloc / hipFree by size, a second hipMalloc of the same size,
// hipMallocAsync from the default pool, virtual-memory reserve + map in 1 GiB pieces, the first kernel launch (module load),
// H2D / D2H of pageable, registered and pinned host memory.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/alloc_probe tools/alloc_probe.hip      Run: tools/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_touch(unsigned *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (unsigned)i; }
int main() {
  double t0 = now();
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  std::printf("{\"what\": \"runtime init (hipSetDevice + hipFree(0))\", \"ms\": %.2f}\n", now() - t0);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  { void *p; t0 = now(); CK(hipMalloc(&p, 1 << 20)); double a = now() - t0; t0 = now(); hipLaunchKernelGGL(k_touch, dim3(1), dim3(256), 0, st, (unsigned *)p, (size_t)256); CK(hipStreamSynchronize(st));
    std::printf("{\"what\": \"first 1 MiB hipMalloc\", \"ms\": %.2f, \"first_kernel_launch_ms\": %.2f}\n", a, now() - t0); CK(hipFree(p)); }
  const size_t G = (size_t)1 << 30;
  for (size_t gb : {1, 4, 8, 24, 44}) {
    void *p = nullptr;
    t0 = now(); CK(hipMalloc(&p, gb * G)); double a = now() - t0;
    t0 = now(); hipLaunchKernelGGL(k_touch, dim3((unsigned)(gb * G / 4 / 256 / 64)), dim3(256), 0, st, (unsigned *)p, gb * G / 4 / 64); CK(hipStreamSynchronize(st)); double k = now() - t0;
    t0 = now(); CK(hipMemsetAsync(p, 0, gb * G, st)); CK(hipStreamSynchronize(st)); double ms1 = now() - t0;
    t0 = now(); CK(hipMemsetAsync(p, 0, gb * G, st)); CK(hipStreamSynchronize(st)); double ms2 = now() - t0;
    t0 = now(); CK(hipFree(p)); double f = now() - t0;
    t0 = now(); CK(hipMalloc(&p, gb * G)); double a2 = now() - t0;
    t0 = now(); CK(hipFree(p)); double f2 = now() - t0;
    std::printf("{\"what\": \"hipMalloc\", \"GiB\": %zu, \"malloc_ms\": %.2f, \"sparse_touch_ms\": %.2f, \"memset1_ms\": %.2f, \"memset2_ms\": %.2f, \"free_ms\": %.2f, \"malloc_again_ms\": %.2f, \"free_again_ms\": %.2f}\n", gb, a, k, ms1, ms2, f, a2, f2);
  }
  {  // stream-ordered allocation from the default pool
    hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = UINT64_MAX; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (size_t gb : {4, 24, 44}) {
      void *p = nullptr;
      t0 = now(); CK(hipMallocAsync(&p, gb * G, st)); CK(hipStreamSynchronize(st)); double a = now() - t0;
      t0 = now(); CK(hipMemsetAsync(p, 0, gb * G, st)); CK(hipStreamSynchronize(st)); double m = now() - t0;
      t0 = now(); CK(hipFreeAsync(p, st)); CK(hipStreamSynchronize(st)); double f = now() - t0;
      t0 = now(); CK(hipMallocAsync(&p, gb * G, st)); CK(hipStreamSynchronize(st)); double a2 = now() - t0;
      CK(hipFreeAsync(p, st)); CK(hipStreamSynchronize(st));
      std::printf("{\"what\": \"hipMallocAsync\", \"GiB\": %zu, \"malloc_ms\": %.2f, \"memset_ms\": %.2f, \"free_ms\": %.2f, \"malloc_again_ms\": %.2f}\n", gb, a, m, f, a2);
    }
    CK(hipMemPoolTrimTo(pool, 0));
  }
  {  // virtual memory: reserve 44 GiB, map 1 GiB at a time
    hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e == hipSuccess) {
      void *va = nullptr; t0 = now(); e = hipMemAddressReserve(&va, 44 * G, 0, nullptr, 0); double r = now() - t0;
      if (e == hipSuccess) {
        std::vector<hipMemGenericAllocationHandle_t> hs; double create = 0, map = 0, acc = 0;
        hipMemAccessDesc ad; memset(&ad, 0, sizeof(ad)); ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        bool ok = true;
        for (size_t i = 0; i < 44 && ok; i++) {
          hipMemGenericAllocationHandle_t h; t0 = now(); ok = hipMemCreate(&h, G, &prop, 0) == hipSuccess; create += now() - t0; if (!ok) break;
          t0 = now(); ok = hipMemMap((char *)va + i * G, G, 0, h, 0) == hipSuccess; map += now() - t0; if (!ok) break;
          t0 = now(); ok = hipMemSetAccess((char *)va + i * G, G, &ad, 1) == hipSuccess; acc += now() - t0;
          hs.push_back(h);
        }
        double m = -1;
        if (ok) { t0 = now(); CK(hipMemsetAsync(va, 0, 44 * G, st)); CK(hipStreamSynchronize(st)); m = now() - t0; }
        std::printf("{\"what\": \"VMM reserve 44 GiB + map 1 GiB pieces\", \"granularity\": %zu, \"reserve_ms\": %.2f, \"create_ms\": %.2f, \"map_ms\": %.2f, \"set_access_ms\": %.2f, \"pieces\": %zu, \"memset_ms\": %.2f}\n", gran, r, create, map, acc, hs.size(), m);
        t0 = now();
        for (size_t i = 0; i < hs.size(); i++) { (void)hipMemUnmap((char *)va + i * G, G); (void)hipMemRelease(hs[i]); }
        (void)hipMemAddressFree(va, 44 * G);
        std::printf("{\"what\": \"VMM unmap + release\", \"ms\": %.2f}\n", now() - t0);
      } else std::printf("{\"what\": \"VMM\", \"error\": \"reserve: %s\"}\n", hipGetErrorString(e));
    } else std::printf("{\"what\": \"VMM\", \"error\": \"granularity: %s\"}\n", hipGetErrorString(e));
    (void)hipGetLastError();
  }
  {  // host <-> device copies of 1 GiB / 4 GiB
    void *d; CK(hipMalloc(&d, 4 * G));
    char *h = (char *)aligned_alloc(4096, 4 * G); memset(h, 1, 4 * G);
    t0 = now(); CK(hipMemcpyAsync(d, h, G, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double a1 = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(d, h, G, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); double a2 = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h, d, 4 * G, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double b1 = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h, d, 4 * G, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); double b2 = now() - t0;
    std::printf("{\"what\": \"pageable\", \"h2d_1GiB_first_ms\": %.2f, \"h2d_1GiB_ms\": %.2f, \"d2h_4GiB_first_ms\": %.2f, \"d2h_4GiB_ms\": %.2f}\n", a1, a2, b1, b2);
    t0 = now(); CK(hipHostRegister(h, 4 * G, hipHostRegisterDefault)); double r = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(d, h, G, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st)); a1 = now() - t0;
    t0 = now(); CK(hipMemcpyAsync(h, d, 4 * G, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); b1 = now() - t0;
    t0 = now(); CK(hipHostUnregister(h)); double u = now() - t0;
    std::printf("{\"what\": \"registered\", \"register_4GiB_ms\": %.2f, \"h2d_1GiB_ms\": %.2f, \"d2h_4GiB_ms\": %.2f, \"unregister_ms\": %.2f}\n", r, a1, b1, u);
    void *ph; t0 = now(); CK(hipHostMalloc(&ph, 64 << 20, hipHostMallocDefault)); double pm = now() - t0;
    // chunked through a 2 x 32 MiB pinned ring with a host memcpy (what a library can do without touching the caller's pages)
    t0 = now();
    { const size_t C = 32 << 20; hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
      for (size_t off = 0, i = 0; off < G; off += C, i++) { char *s = (char *)ph + (i & 1) * C; if (i >= 2) CK(hipEventSynchronize(ev[i & 1])); memcpy(s, h + off, C);
        CK(hipMemcpyAsync((char *)d + off, s, C, hipMemcpyHostToDevice, st)); CK(hipEventRecord(ev[i & 1], st)); }
      CK(hipStreamSynchronize(st)); }
    double ring = now() - t0;
    std::printf("{\"what\": \"pinned ring\", \"hipHostMalloc_64MiB_ms\": %.2f, \"h2d_1GiB_through_ring_ms\": %.2f}\n", pm, ring);
    CK(hipHostFree(ph)); CK(hipFree(d)); free(h);
  }
  return 0;
}
