// A malloc-backed stand-in for the HIP runtime, for running the HOST half of libdc3hip under AddressSanitizer and
// ThreadSanitizer in a container without a GPU (tools/hostmock/Makefile: the library's own translation unit is compiled by
// hipcc with the sanitizer on the host pass only and linked against this file instead of libamdhip64).
// Kernels do not run: a launch only checks its stream.  Builds therefore fail at their first device-to-host read (the
// alphabet has no symbols) — which is the point: group / context lifecycle, rank threads, pinned-buffer recycling and every
// error path run for real, on allocations the sanitizers track:
//   hipMalloc      = calloc, registered; hipFree of anything else aborts
//   hipHostMalloc  = mmap (NOT the heap: a stray free() / delete of a pinned pointer is reported by ASan as "attempting
//                    free on address which was not malloc()-ed", the signature of the round-4 crash), registered
//   streams/events = small heap objects, poisoned on destroy (use after destroy = heap-use-after-free)
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {
std::mutex g_mu;
std::map<void *, size_t> g_dev, g_pinned;
struct MockStream { unsigned magic; int device; };
struct MockEvent { unsigned magic; };
constexpr unsigned kStreamMagic = 0x5753u, kEventMagic = 0x4556u;
thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;
struct Cfg { dim3 g, b; size_t sh; hipStream_t st; };
thread_local std::vector<Cfg> t_cfg;
[[noreturn]] void die(const char *what, const void *p) { std::fprintf(stderr, "hip_mock: %s (%p)\n", what, p); std::abort(); }
void check_stream(hipStream_t s) {
  if (!s) return;
  if (reinterpret_cast<MockStream *>(s)->magic != kStreamMagic) die("call on a stream that is not alive", s);
}
int ndev() { const char *e = getenv("HIP_MOCK_DEVICES"); return e ? atoi(e) : 1; }
}  // namespace

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = ndev(); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= ndev()) return t_last = hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) { memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipGetLastError(void) { const hipError_t e = t_last; t_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock error"; }
hipError_t hipRuntimeGetVersion(int *v) { *v = HIP_VERSION; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
  // (sizes are those of real device buffers: cap what the container can hold, the host code never touches them itself)
  const size_t cap = (size_t)1 << 26;
  void *q = calloc(1, n < cap ? (n ? n : 1) : cap);
  if (!q) return t_last = hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mu); g_dev[q] = n; *p = q; return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  { std::lock_guard<std::mutex> lk(g_mu); if (!g_dev.erase(p)) die("hipFree of a pointer hipMalloc did not return (or freed twice)", p); }
  free(p); return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
  const size_t len = (n + 4095) / 4096 * 4096 + 4096;
  void *q = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (q == MAP_FAILED) return t_last = hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mu); g_pinned[q] = len; *p = q; return hipSuccess;
}
hipError_t hipHostFree(void *p) {
  if (!p) return hipSuccess;
  size_t len = 0;
  { std::lock_guard<std::mutex> lk(g_mu); auto it = g_pinned.find(p); if (it == g_pinned.end()) die("hipHostFree of a pointer hipHostMalloc did not return (or freed twice)", p); len = it->second; g_pinned.erase(it); }
  munmap(p, len); return hipSuccess;
}
static size_t room(const void *p) {   // bytes the mock really holds behind a device pointer (SIZE_MAX: not a device pointer)
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_dev.upper_bound(const_cast<void *>(p));
  if (it == g_dev.begin()) return (size_t)-1;
  --it;
  const char *b = static_cast<const char *>(it->first);
  const size_t held = it->second < ((size_t)1 << 26) ? it->second : ((size_t)1 << 26);
  if (static_cast<const char *>(p) >= b + it->second) return (size_t)-1;
  const size_t off = (size_t)(static_cast<const char *>(p) - b);
  return off < held ? held - off : 0;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) {
  check_stream(st);
  if (!n) return hipSuccess;
  if (!d || !s) die("hipMemcpy with a null pointer", d ? s : d);
  const size_t rd = room(d), rs = room(s);
  const size_t m = std::min(n, std::min(rd, rs));   // (device buffers are capped: copy what exists; host sides stay exact)
  if (rd == (size_t)-1 && rs == (size_t)-1) memcpy(d, s, n); else if (m) memcpy(d, s, m);
  return hipSuccess;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(d, s, n, k, nullptr); }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) {
  check_stream(st);
  const size_t r = room(d);
  if (n) memset(d, v, r == (size_t)-1 ? n : std::min(n, r));
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { MockStream *m = new MockStream{kStreamMagic, t_device}; *s = reinterpret_cast<hipStream_t>(m); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { check_stream(s); MockStream *m = reinterpret_cast<MockStream *>(s); m->magic = 0; delete m; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { check_stream(s); return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
// no virtual-memory management in the mock: the library falls back to hipMalloc (DevBuf, dc3_host_core.hpp)
hipError_t hipMemAddressReserve(void **, size_t, size_t, void *, unsigned long long) { return t_last = hipErrorNotSupported; }
hipError_t hipMemAddressFree(void *, size_t) { return hipSuccess; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *, size_t, const hipMemAllocationProp *, unsigned long long) { return t_last = hipErrorNotSupported; }
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t) { return hipSuccess; }
hipError_t hipMemMap(void *, size_t, size_t, hipMemGenericAllocationHandle_t, unsigned long long) { return t_last = hipErrorNotSupported; }
hipError_t hipMemUnmap(void *, size_t) { return hipSuccess; }
hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return t_last = hipErrorNotSupported; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(new MockEvent{kEventMagic}); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { MockEvent *m = reinterpret_cast<MockEvent *>(e); if (m->magic != kEventMagic) die("hipEventDestroy of a dead event", e); m->magic = 0; delete m; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { check_stream(s); if (reinterpret_cast<MockEvent *>(e)->magic != kEventMagic) die("hipEventRecord on a dead event", e); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t st) { check_stream(st); return hipSuccess; }
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { t_cfg.push_back(Cfg{g, b, sh, st}); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *sh, hipStream_t *st) {
  const Cfg c = t_cfg.back(); t_cfg.pop_back(); *g = c.g; *b = c.b; *sh = c.sh; *st = c.st; return hipSuccess;
}
}
