// GPU box: which shapes of hipMemCreate / hipMemMap the runtime accepts (a 976 MiB piece followed by a 464 MiB one failed inside
// devbuf_commit).  Build: hipcc --offload-arch=gfx950 -O2 -o tools/vmm_probe tools/vmm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
static const size_t M = (size_t)1 << 20;
int main() {
  hipSetDevice(0); hipFree(nullptr);
  hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
  hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
  std::printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu}\n", gmin, grec);
  hipMemAccessDesc acc; memset(&acc, 0, sizeof(acc)); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  struct Case { const char *name; std::vector<size_t> pieces_mib; };
  std::vector<Case> cases = {
    {"976+464", {976, 464}}, {"1024+1024", {1024, 1024}}, {"1024+464", {1024, 464}}, {"512+512+512", {512, 512, 512}}, {"976 then 48 then 464", {976, 48, 464}},
    {"2+2+2", {2, 2, 2}}, {"1000+1000", {1000, 1000}}, {"64+976", {64, 976}}, {"3000", {3000}}, {"1024+1024+1024+1024+1024+1024", {1024, 1024, 1024, 1024, 1024, 1024}}};
  for (auto &cs : cases) {
    size_t total = 0; for (size_t p : cs.pieces_mib) total += p * M;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, total + 1024 * M, 2 * M, nullptr, 0);
    if (e != hipSuccess) { std::printf("{\"case\": \"%s\", \"reserve\": \"%s\"}\n", cs.name, hipGetErrorString(e)); continue; }
    size_t off = 0; std::vector<hipMemGenericAllocationHandle_t> hs; std::vector<size_t> szs;
    const char *fail = "none"; hipError_t fe = hipSuccess; size_t fpiece = 0;
    for (size_t p : cs.pieces_mib) {
      hipMemGenericAllocationHandle_t h;
      if ((fe = hipMemCreate(&h, p * M, &prop, 0)) != hipSuccess) { fail = "create"; fpiece = p; break; }
      if ((fe = hipMemMap((char *)va + off, p * M, 0, h, 0)) != hipSuccess) { fail = "map"; fpiece = p; hipMemRelease(h); break; }
      if ((fe = hipMemSetAccess((char *)va + off, p * M, &acc, 1)) != hipSuccess) { fail = "set_access"; fpiece = p; hipMemUnmap((char *)va + off, p * M); hipMemRelease(h); break; }
      hs.push_back(h); szs.push_back(p * M); off += p * M;
    }
    hipError_t me = hipSuccess;
    if (fe == hipSuccess) { me = hipMemset(va, 1, total); if (me == hipSuccess) me = hipDeviceSynchronize(); }
    std::printf("{\"case\": \"%s\", \"va\": \"%p\", \"failed_at\": \"%s\", \"piece_mib\": %zu, \"error\": \"%s\", \"memset\": \"%s\"}\n", cs.name, va, fail, fpiece, hipGetErrorString(fe), hipGetErrorString(me));
    (void)hipGetLastError();
    size_t o2 = 0; for (size_t i = 0; i < hs.size(); i++) { hipMemUnmap((char *)va + o2, szs[i]); hipMemRelease(hs[i]); o2 += szs[i]; }
    hipMemAddressFree(va, total + 1024 * M);
  }
  // one handle mapped, then SetAccess over a growing range (does access have to be set per mapping or may it span?)
  return 0;
}
