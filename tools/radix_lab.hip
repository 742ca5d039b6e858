// radix_lab.hip — kernel lab for the record sorts of libdc3hip: times ONE pass of every sort kernel on synthetic
// (image << pbits | pos) words and checks the MSD pipeline (dc3_msd.hip.hpp) against the stable LSD passes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/radix_lab tools/radix_lab.hip
//   tools/radix_lab [log2n=30] [reps=5]
// Output: one JSON line per measurement (profiles/r03*_radix_lab.jsonl).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../stringsearch_amd/csrc/dc3_kernels.hip.hpp"
#include "../stringsearch_amd/csrc/dc3_msd.hip.hpp"

using namespace dc3;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ u64 mix64(u64 x) {
  x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
// records in the product's Rec8 layout: high half first
__device__ __forceinline__ u64 memw(u64 w) { return (w << 32) | (w >> 32); }
__global__ void k_gen(u64 *w, u32 n, u32 pbits, u32 nbits, u32 skew) {
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    u64 img = mix64(i) >> (64 - nbits);
    if (skew) {   // squared distribution: dense near 0
      const double u = (double)(mix64(i) >> 11) * (1.0 / 9007199254740992.0);
      img = (u64)(u * u * (double)(1ull << nbits));
    }
    w[i] = memw((img << pbits) | i);
  }
}
__global__ void k_copy16(const u32x4 *a, u32x4 *b, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// sorted ascending + xor/sum of the words
__global__ void k_check_words(const u64 *w, u32 n, unsigned long long *acc, u32 *bad) {
  unsigned long long s = 0, x = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u64 v = memw(w[i]);
    s += v; x ^= mix64(v);
    if (i > 0 && memw(w[i - 1]) >= v) atomicAdd(bad, 1u);
  }
  atomicAdd(&acc[0], s); atomicXor(&acc[1], x);
}
__global__ void k_check_split(const u32 *sa, const u32 *img, u32 n, u32 pbits, u32 nbits, u32 skew, unsigned long long *acc, u32 *bad) {
  unsigned long long s = 0, x = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u32 p = sa[i];
    u64 im = mix64(p) >> (64 - nbits);
    if (skew) { const double u = (double)(mix64(p) >> 11) * (1.0 / 9007199254740992.0); im = (u64)(u * u * (double)(1ull << nbits)); }
    const u64 v = (im << pbits) | p;
    s += v; x ^= mix64(v);
    if ((u32)im != img[i]) atomicAdd(bad, 1u);
    if (i > 0) {
      const u32 q = sa[i - 1];
      u64 jm = mix64(q) >> (64 - nbits);
      if (skew) { const double u = (double)(mix64(q) >> 11) * (1.0 / 9007199254740992.0); jm = (u64)(u * u * (double)(1ull << nbits)); }
      if (((jm << pbits) | q) >= v) atomicAdd(bad, 1u);
    }
  }
  atomicAdd(&acc[0], s); atomicXor(&acc[1], x);
}

// digits non-decreasing along the array (a partition pass's postcondition) + checksum
__global__ void k_check_part(const u64 *w, u32 n, u32 shift, unsigned long long *acc, u32 *bad) {
  unsigned long long s = 0, x = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u64 v = memw(w[i]);
    s += v; x ^= mix64(v);
    if (i > 0 && (memw(w[i - 1]) >> shift) > (v >> shift)) atomicAdd(bad, 1u);
  }
  atomicAdd(&acc[0], s); atomicXor(&acc[1], x);
}
// one digit non-decreasing along the array (an LSD pass's postcondition on a random input) + checksum
__global__ void k_check_digit(const u64 *w, u32 n, u32 shift, u32 mask, unsigned long long *acc, u32 *bad) {
  unsigned long long s = 0, x = 0;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u64 v = memw(w[i]);
    s += v; x ^= mix64(v);
    if (i > 0) {
      const u64 u = memw(w[i - 1]);
      const u32 da = (u32)(u >> shift) & mask, db = (u32)(v >> shift) & mask;
      if (da > db || (da == db && (u & ((1ull << shift) - 1)) > (v & ((1ull << shift) - 1)))) atomicAdd(bad, 1u);    // stable: positions ascend inside a digit
    }
  }
  atomicAdd(&acc[0], s); atomicXor(&acc[1], x);
}
// per-group bucket sizes: tile t = words [t * tile, ...) belongs to group t % 8; cntg[digit * 8 + g]
__global__ __launch_bounds__(1024) void k_lab_hist_g(const u64 *__restrict__ in, u32 n, u32 shift, u32 dbits, u32 tile, u32 *__restrict__ cntg) {
  __shared__ u32 hist[8 * 1024];
  const u32 tid = threadIdx.x, ndig = 1u << dbits;
  for (u32 j = tid; j < 8 * 1024; j += 1024) hist[j] = 0;
  __syncthreads();
  const u32 begin = blockIdx.x * 65536u;
  for (u32 j = 0; j < 64; j++) {
    const u32 i = begin + j * 1024 + tid;
    if (i < n) atomicAdd(&hist[(((i / tile) & 7u) << 10) + (u32)(memw(in[i]) >> shift)], 1u);
  }
  __syncthreads();
  for (u32 j = tid; j < 8 * ndig; j += 1024) {
    const u32 g = j / ndig, d = j % ndig;
    if (hist[(g << 10) + d]) atomicAdd(&cntg[d * 8 + g], hist[(g << 10) + d]);
  }
}
__global__ void k_lab_transpose_cur(const u32 *start_dg, u32 ndig, u32 *cur_gd) {
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < 8 * ndig; j += gridDim.x * blockDim.x) cur_gd[(j % 8) * ndig + j / 8] = start_dg[j];
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
  void start() { CK(hipEventRecord(a, 0)); }
  float stop() { CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
};
static u32 bits_of(u64 v) { u32 b = 0; while (v) { b++; v >>= 1; } return b ? b : 1; }

template <class F>
static float time_it(int reps, F f) {
  Timer t; std::vector<float> ms;
  for (int r = 0; r < reps; r++) { t.start(); f(); ms.push_back(t.stop()); }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const u32 skew = argc > 3 ? atoi(argv[3]) : 0;
  const u32 sections = argc > 4 ? (u32)atoi(argv[4]) : 15u;     // 1 copy, 2 LSD, 4 pass-1 experiments, 8 MSD pipeline
  const u32 n = lg >= 32 ? 0xffffff00u : (1u << lg);
  const u32 pbits = bits_of((u64)n + 2), nbits = 64 - pbits;
  printf("{\"lab\":\"config\",\"n\":%u,\"pbits\":%u,\"nbits\":%u,\"skew\":%u}\n", n, pbits, nbits, skew);
  u64 *src, *a, *b; u32 *sa, *img;
  CK(hipMalloc(&src, (size_t)n * 8)); CK(hipMalloc(&a, (size_t)n * 8)); CK(hipMalloc(&b, (size_t)n * 8));
  CK(hipMalloc(&sa, (size_t)n * 4 + 64)); CK(hipMalloc(&img, (size_t)n * 4 + 64));
  unsigned long long *acc; u32 *bad; u32 *words;
  CK(hipMalloc(&acc, 64)); CK(hipMalloc(&bad, 64)); CK(hipMalloc(&words, 256));
  hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, src, n, pbits, nbits, skew);
  CK(hipDeviceSynchronize());
  auto checksum_words = [&](const u64 *w, unsigned long long out[2], u32 *nbad) {
    CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_check_words, dim3(4096), dim3(256), 0, 0, w, n, acc, bad);
    CK(hipMemcpy(out, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(nbad, bad, 4, hipMemcpyDeviceToHost));
  };
  unsigned long long ref[2]; u32 nb_unsorted;
  checksum_words(src, ref, &nb_unsorted);

  // ---- copy roofline of this box
  if (sections & 1u) {
    const size_t n16 = (size_t)n / 2;
    for (int g : {2048, 4096, 16384}) {
      float ms = time_it(reps, [&] { hipLaunchKernelGGL(k_copy16, dim3(g), dim3(256), 0, 0, (const u32x4 *)src, (u32x4 *)a, n16); });
      printf("{\"lab\":\"copy16\",\"grid\":%d,\"ms\":%.3f,\"TBps_rw\":%.3f}\n", g, ms, 16.0 * n / ms * 1e-9);
    }
    float ms = time_it(reps, [&] { CK(hipMemcpyAsync(a, src, (size_t)n * 8, hipMemcpyDeviceToDevice, 0)); });
    printf("{\"lab\":\"hipMemcpyD2D\",\"ms\":%.3f,\"TBps_rw\":%.3f}\n", ms, 16.0 * n / ms * 1e-9);
  }

  // ---- one stable LSD pass: as the product runs it (2048 chunks), and with one tile per block + XCD-swizzled chunk order
  if (sections & 2u) {
    auto lsd = [&](const char *name, auto kern, auto smem_c, int NB, int kTile, int NW, u32 blocks_target, bool swz) {
      const size_t smem = smem_c;
      u32 chunk = blocks_target ? (n + blocks_target - 1) / blocks_target : kTile; chunk = (chunk + kTile - 1) / kTile * kTile;
      const u32 nchunks = (n + chunk - 1) / chunk;
      u32 *table, *dbase;
      CK(hipMalloc(&table, (size_t)NB * nchunks * 4)); CK(hipMalloc(&dbase, NB * 4));
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      KeyDig dig; dig.shift = pbits; dig.mask = NB - 1;
      float up = time_it(reps, [&] {
        if (NB == 512) hipLaunchKernelGGL((k_rs_upsweep<Rec8, 512>), dim3(nchunks), dim3(kBlock), 0, 0, (const Rec8 *)src, n, chunk, nchunks, dig, table);
        else hipLaunchKernelGGL((k_rs_upsweep<Rec8, 256>), dim3(nchunks), dim3(kBlock), 0, 0, (const Rec8 *)src, n, chunk, nchunks, dig, table);
      });
      float sc = time_it(1, [&] {
        hipLaunchKernelGGL(k_scan_rows, dim3(NB), dim3(kBlock), 0, 0, table, nchunks, dbase);
        hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, 0, dbase, (u32)NB, (u32 *)nullptr);
      });
      ArrayLoader<Rec8> ld; ld.p = (const Rec8 *)src; RecSink<Rec8> sk; sk.p = (Rec8 *)a;
      const u32 cpx = swz ? (nchunks + 7) / 8 : 0u;
      const u32 grid = swz ? cpx * 8 : nchunks;
      float down = time_it(reps, [&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), smem, 0, ld, sk, n, chunk, nchunks, dig, table, dbase, cpx); });
      // stable pass postcondition: digits non-decreasing; and the checksum
      CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(k_check_digit, dim3(4096), dim3(256), 0, 0, (const u64 *)a, n, pbits, (u32)NB - 1, acc, bad);
      unsigned long long got[2]; u32 nbad;
      CK(hipMemcpy(got, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
      printf("{\"lab\":\"lsd_pass\",\"variant\":\"%s\",\"nchunks\":%u,\"swizzle\":%d,\"upsweep_ms\":%.3f,\"scan_ms\":%.3f,\"downsweep_ms\":%.3f,\"down_TBps_moved\":%.3f,\"ok\":%s}\n", name,
             nchunks, (int)swz, up, sc, down, 16.0 * n / down * 1e-9, (nbad == 0 && got[0] == ref[0] && got[1] == ref[1]) ? "true" : "false");
      CK(hipFree(table)); CK(hipFree(dbase));
    };
#define LSD(NB, IPT, NW, PF) k_rs_downsweep<Rec8, NB, IPT, NW, PF, ArrayLoader<Rec8>, RecSink<Rec8>>, DownsweepSmem<Rec8, IPT, NW, NB>::kBytes, NB, NW * 64 * IPT, NW
    lsd("product 512x12x16 pf", LSD(512, 12, 16, true), 2048, false);
    lsd("512x12x16 pf, 2048 chunks, swz", LSD(512, 12, 16, true), 2048, true);
    lsd("512x12x16, tile chunks", LSD(512, 12, 16, false), 0, false);
    lsd("512x12x16, tile chunks, swz", LSD(512, 12, 16, false), 0, true);
    lsd("512x12x16 pf, 4-tile chunks, swz", LSD(512, 12, 16, true), n / (12288 * 4), true);
    lsd("512x16x8, tile chunks, swz", LSD(512, 16, 8, false), 0, true);
    lsd("512x16x8, tile chunks", LSD(512, 16, 8, false), 0, false);
    lsd("512x8x16, tile chunks, swz", LSD(512, 8, 16, false), 0, true);
    lsd("256x12x16, tile chunks, swz", LSD(256, 12, 16, false), 0, true);
    lsd("256x16x8, tile chunks, swz", LSD(256, 16, 8, false), 0, true);
  }

  // ---- pass-1 experiments: digit width, tile size, grouped cursors, non-temporal accesses
  if (sections & 4u) {
    u32 *cnt, *bst, *cur, *tp, *tph, *cntg, *stg, *curg;
    CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&bst, 4100 * 4)); CK(hipMalloc(&cur, 4096 * 4)); CK(hipMalloc(&tp, 4100 * 4)); CK(hipMalloc(&tph, 4100 * 4));
    CK(hipMalloc(&cntg, 8200 * 4)); CK(hipMalloc(&stg, 8200 * 4)); CK(hipMalloc(&curg, 8200 * 4));
    auto set_attr = [&](auto kern, int ipt) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)msd_part_smem(ipt))); };
    set_attr(k_msd_part<false, 8, false, false>, 8); set_attr(k_msd_part<false, 8, false, true>, 8); set_attr(k_msd_part<false, 16, false, false>, 16);
    set_attr(k_msd_part<false, 8, true, false>, 8); set_attr(k_msd_part<false, 16, true, false>, 16); set_attr(k_msd_part<false, 4, false, false>, 4); set_attr(k_msd_part<false, 4, true, false>, 4);
    for (u32 d1 : {8u, 9u, 10u}) {
      const u32 sh1 = pbits + nbits - d1, nb1 = 1u << d1;
      CK(hipMemset(cnt, 0, 4096 * 4));
      const u32 nt = (n + kMsdHistTile - 1) / kMsdHistTile;
      u32 h_t[2] = {0, nt}, h_b[2] = {0, n};
      CK(hipMemcpy(tph, h_t, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bst, h_b, 8, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(k_msd_hist2, dim3(nt), dim3(1024), 0, 0, (const u64 *)src, sh1, d1, tph, bst, 1u, cnt);
      auto run = [&](const char *name, auto kern, int ipt, bool grouped) {
        const u32 tile = kMsdNW * 64 * ipt, grid = (n + tile - 1) / tile;
        if (grouped) {
          CK(hipMemset(cntg, 0, 8200 * 4));
          hipLaunchKernelGGL(k_lab_hist_g, dim3((n + 65535) / 65536), dim3(1024), 0, 0, (const u64 *)src, n, sh1, d1, tile, cntg);
          hipLaunchKernelGGL(k_msd_scan2, dim3(1), dim3(1024), 0, 0, cntg, 8 * nb1, stg, curg, words);   // stg[d*8+g]
        }
        float ms = time_it(reps, [&] {
          if (grouped) hipLaunchKernelGGL(k_lab_transpose_cur, dim3(8), dim3(1024), 0, 0, stg, nb1, curg);
          else hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(1024), 0, 0, cnt, nb1, n, bst, cur, tp, tph);
          hipLaunchKernelGGL(kern, dim3(grid), dim3(kMsdNW * 64), msd_part_smem(ipt), 0, (const u64 *)src, a, n, sh1, d1, (const u32 *)nullptr, (const u32 *)nullptr, 0u, grouped ? curg : cur, nb1);
        });
        CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check_part, dim3(4096), dim3(256), 0, 0, (const u64 *)a, n, sh1, acc, bad);
        unsigned long long got[2]; u32 nbad;
        CK(hipMemcpy(got, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"lab\":\"part1\",\"variant\":\"%s\",\"d1\":%u,\"ms\":%.3f,\"TBps_moved\":%.3f,\"ok\":%s}\n", name, d1, ms, 16.0 * n / ms * 1e-9,
               (nbad == 0 && got[0] == ref[0] && got[1] == ref[1]) ? "true" : "false");
      };
      run("tile8192", k_msd_part<false, 8, false, false>, 8, false);
      run("tile8192_nt", k_msd_part<false, 8, false, true>, 8, false);
      run("tile16384", k_msd_part<false, 16, false, false>, 16, false);
      run("tile4096", k_msd_part<false, 4, false, false>, 4, false);
      run("tile8192_grouped", k_msd_part<false, 8, true, false>, 8, true);
      run("tile16384_grouped", k_msd_part<false, 16, true, false>, 16, true);
      run("tile4096_grouped", k_msd_part<false, 4, true, false>, 4, true);
    }
  }

  // ---- MSD pipeline
  for (int variant = 0; variant < 3 && (sections & 8u); variant++) {
    MsdPlan pl; pl.pbits = pbits; pl.nbits = nbits;
    // sub-buckets of ~2048 words (variant 0), ~4096 (1), ~1024 (2)
    const u32 target = variant == 0 ? 11 : variant == 1 ? 12 : 10;
    const u32 lgn = bits_of(n - 1);
    u32 tb = lgn > target ? lgn - target : 0;
    if (tb > nbits) tb = nbits;
    pl.d1 = std::min<u32>(10, (tb + 1) / 2); pl.d2 = std::min<u32>(10, tb - pl.d1);
    pl.sh1 = pbits + nbits - pl.d1; pl.sh2 = pl.sh1 - pl.d2;
    const u32 rb = nbits - pl.d1 - pl.d2;
    const u32 nb1 = 1u << pl.d1, n2 = 1u << (pl.d1 + pl.d2);
    u32 *cnt1, *bstart, *cur1, *tpre, *tpreh, *cnt2, *start2, *cur2;
    CK(hipMalloc(&cnt1, 4096 * 4)); CK(hipMalloc(&bstart, 4100 * 4)); CK(hipMalloc(&cur1, 4096 * 4));
    CK(hipMalloc(&tpre, 4100 * 4)); CK(hipMalloc(&tpreh, 4100 * 4));
    CK(hipMalloc(&cnt2, ((size_t)n2 + 16) * 4)); CK(hipMalloc(&start2, ((size_t)n2 + 16) * 4)); CK(hipMalloc(&cur2, ((size_t)n2 + 16) * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<false, 8, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<true, 8, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    // bucket sizes (in the product: the pack kernel's digit table); here an up-sweep over the top digit
    {
      CK(hipMemset(cnt1, 0, 4096 * 4));
      // reuse k_msd_hist2 with one pseudo bucket covering everything: tpreh = {0, ntiles}, bstart = {0, n}
      const u32 nt = (n + kMsdHistTile - 1) / kMsdHistTile;
      u32 h_t[2] = {0, nt}, h_b[2] = {0, n};
      CK(hipMemcpy(tpreh, h_t, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(bstart, h_b, 8, hipMemcpyHostToDevice));
      float ms = time_it(1, [&] { hipLaunchKernelGGL(k_msd_hist2, dim3(nt), dim3(1024), 0, 0, (const u64 *)src, pl.sh1, pl.d1, tpreh, bstart, 1u, cnt1); });
      printf("{\"lab\":\"msd_hist1\",\"variant\":%d,\"d1\":%u,\"d2\":%u,\"ms\":%.3f}\n", variant, pl.d1, pl.d2, ms);
    }
    float t_tiles = time_it(1, [&] { hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(1024), 0, 0, cnt1, nb1, n, bstart, cur1, tpre, tpreh); });
    const u32 grid1 = (n + kMsdTile - 1) / kMsdTile;
    float t_p1 = time_it(reps, [&] {
      hipLaunchKernelGGL(k_msd_tiles, dim3(1), dim3(1024), 0, 0, cnt1, nb1, n, bstart, cur1, tpre, tpreh);
      hipLaunchKernelGGL((k_msd_part<false, 8, false, false>), dim3(grid1), dim3(kMsdNW * 64), kMsdPartSmem, 0, (const u64 *)src, a, n, pl.sh1, pl.d1, (const u32 *)nullptr, (const u32 *)nullptr, 0u, cur1, 0u);
    });
    float t_h2 = 0, t_s2 = 0, t_p2 = 0;
    const u64 *sorted2 = a;
    if (pl.d2 > 0) {
      const u32 gridh = n / kMsdHistTile + nb1 + 1;
      t_h2 = time_it(reps, [&] {
        CK(hipMemsetAsync(cnt2, 0, (size_t)n2 * 4, 0));
        hipLaunchKernelGGL(k_msd_hist2, dim3(gridh), dim3(1024), 0, 0, (const u64 *)a, pl.sh2, pl.d2, tpreh, bstart, nb1, cnt2);
      });
      t_s2 = time_it(reps, [&] { hipLaunchKernelGGL(k_msd_scan2, dim3(1), dim3(1024), 0, 0, cnt2, n2, start2, cur2, words); });
      const u32 grid2 = n / kMsdTile + nb1 + 1;
      t_p2 = time_it(reps, [&] {
        hipLaunchKernelGGL(k_msd_scan2, dim3(1), dim3(1024), 0, 0, cnt2, n2, start2, cur2, words);
        hipLaunchKernelGGL((k_msd_part<true, 8, false, false>), dim3(grid2), dim3(kMsdNW * 64), kMsdPartSmem, 0, (const u64 *)a, b, n, pl.sh2, pl.d2, tpre, bstart, nb1, cur2, 0u);
      });
      sorted2 = b;
    } else {
      // sub-buckets = buckets
      CK(hipMemcpy(start2, bstart, ((size_t)nb1 + 1) * 4, hipMemcpyDeviceToDevice));
      hipLaunchKernelGGL(k_msd_scan2, dim3(1), dim3(1024), 0, 0, cnt1, nb1, start2, cur2, words);
    }
    u32 maxsub = 0;
    CK(hipMemcpy(&maxsub, words, 4, hipMemcpyDeviceToHost));
    printf("{\"lab\":\"msd_parts\",\"variant\":%d,\"d1\":%u,\"d2\":%u,\"rem_bits\":%u,\"tiles_ms\":%.3f,\"part1_ms\":%.3f,\"hist2_ms\":%.3f,\"scan2_ms\":%.3f,\"part2_ms\":%.3f,\"max_subbucket\":%u,"
           "\"part1_TBps_moved\":%.3f}\n", variant, pl.d1, pl.d2, rb, t_tiles, t_p1, t_h2, t_s2, t_p2, maxsub, 16.0 * n / t_p1 * 1e-9);
    // local sort shapes
    auto run_local = [&](const char *name, auto kern_rec, auto kern_split, u32 cap, u32 bb, u32 nt = 256) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_rec), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap * 8));
      CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern_split), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap * 8));
      if (maxsub > cap) { printf("{\"lab\":\"msd_local\",\"shape\":\"%s\",\"skipped\":\"max sub-bucket %u > cap %u\"}\n", name, maxsub, cap); return; }
      const u32 ubb = std::min(bb, rb);
      const u32 shb = pl.sh2 - ubb;
      (void)ubb;
      MsdRecSink rs; rs.p = a == sorted2 ? b : a;
      MsdSplitSink ss; ss.sa = sa; ss.img = img; ss.pbits = pbits;
      float t_rec = time_it(reps, [&] { hipLaunchKernelGGL(kern_rec, dim3(n2), dim3(nt), cap * 8, 0, sorted2, start2, shb, rs); });
      unsigned long long got[2]; u32 nbad;
      checksum_words(rs.p, got, &nbad);
      float t_split = time_it(reps, [&] { hipLaunchKernelGGL(kern_split, dim3(n2), dim3(nt), cap * 8, 0, sorted2, start2, shb, ss); });
      CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(k_check_split, dim3(4096), dim3(256), 0, 0, sa, img, n, pbits, nbits, skew, acc, bad);
      unsigned long long got2[2]; u32 nbad2;
      CK(hipMemcpy(got2, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nbad2, bad, 4, hipMemcpyDeviceToHost));
      printf("{\"lab\":\"msd_local\",\"variant\":%d,\"shape\":\"%s\",\"rec_ms\":%.3f,\"split_ms\":%.3f,\"rec_ok\":%s,\"split_ok\":%s,\"bad\":[%u,%u]}\n", variant, name,
             t_rec, t_split, (nbad == 0 && got[0] == ref[0] && got[1] == ref[1]) ? "true" : "false",
             (nbad2 == 0 && got2[0] == ref[0] && got2[1] == ref[1]) ? "true" : "false", nbad, nbad2);
    };
    if (rb >= 10) {
      run_local("256x4096,bb12", k_msd_local<256, 4096, 12, MsdRecSink>, k_msd_local<256, 4096, 12, MsdSplitSink>, 4096, 12);
      run_local("256x4096,bb11", k_msd_local<256, 4096, 11, MsdRecSink>, k_msd_local<256, 4096, 11, MsdSplitSink>, 4096, 11);
      run_local("512x8192,bb12", k_msd_local<512, 8192, 12, MsdRecSink>, k_msd_local<512, 8192, 12, MsdSplitSink>, 8192, 12, 512);
      run_local("512x8192,bb13", k_msd_local<512, 8192, 13, MsdRecSink>, k_msd_local<512, 8192, 13, MsdSplitSink>, 8192, 13, 512);
      run_local("512x4096,bb12", k_msd_local<512, 4096, 12, MsdRecSink>, k_msd_local<512, 4096, 12, MsdSplitSink>, 4096, 12, 512);
      run_local("256x2048,bb11", k_msd_local<256, 2048, 11, MsdRecSink>, k_msd_local<256, 2048, 11, MsdSplitSink>, 2048, 11);
      run_local("256x2048,bb10", k_msd_local<256, 2048, 10, MsdRecSink>, k_msd_local<256, 2048, 10, MsdSplitSink>, 2048, 10);
    }
    CK(hipFree(cnt1)); CK(hipFree(bstart)); CK(hipFree(cur1)); CK(hipFree(tpre)); CK(hipFree(tpreh));
    CK(hipFree(cnt2)); CK(hipFree(start2)); CK(hipFree(cur2));
  }
  return 0;
}
