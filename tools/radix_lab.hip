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
// pass-2 cursors (one plane per group) back from the scanned starts: cur2[g * n2 + s] = start[s * 8 + g]
__global__ void k_lab_starts_to_cur(const u32 *start, u32 n2, u32 *cur2) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)n2 * 8; i += (size_t)gridDim.x * blockDim.x) cur2[(i % 8) * n2 + i / 8] = start[i];
}
__global__ void k_lab_transpose_cur(const u32 *start_dg, u32 ndig, u32 *cur_gd) {
  for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < 8 * ndig; j += gridDim.x * blockDim.x) cur_gd[(j % 8) * ndig + j / 8] = start_dg[j];
}

// ---- lab-only: the partition pass with its knobs exposed (tile size, grouped cursors, non-temporal accesses); the
// product kernel (k_msd_part in dc3_msd.hip.hpp) is the tile8192 / grouped / temporal point of this family with the
// groups owning contiguous eighths of the input instead of every eighth tile
constexpr size_t lab_part_smem(int ipt) { return sizeof(u64) * kMsdNW * 64 * ipt + sizeof(u32) * (2 * kMsdMaxDig + 64); }
template <int IPT, bool kGroup, bool kNT>
__global__ __launch_bounds__(kMsdNW * 64) void k_lab_part(const u64 *__restrict__ in, u64 *__restrict__ out, u32 n, u32 shift, u32 dbits,
                                                         u32 *__restrict__ cursors, u32 gstride) {
  constexpr int NT = kMsdNW * 64, kTile = NT * IPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *srec = reinterpret_cast<u64 *>(smem);
  u32 *hist = reinterpret_cast<u32 *>(smem + sizeof(u64) * kTile);
  u32 *gbase = hist + kMsdMaxDig;
  u32 *tmp = gbase + kMsdMaxDig;
  const u32 tid = threadIdx.x;
  const u32 ndig = 1u << dbits, mask = ndig - 1u;
  u32 *cur = cursors;
  if (kGroup) cur += (size_t)(blockIdx.x & 7u) * gstride;
  const u32 begin = blockIdx.x * (u32)kTile, end = min(n, begin + (u32)kTile);
  const u32 nvalid = end - begin;
  hist[tid] = 0;
  __syncthreads();
  u64 r[IPT];
  u32 rk[IPT];
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = min((u32)(k * NT) + tid, nvalid - 1u);
    r[k] = kNT ? __builtin_nontemporal_load(in + begin + t) : in[begin + t];
  }
#pragma unroll
  for (int k = 0; k < IPT; k++) r[k] = msd_word(r[k]);
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < nvalid) rk[k] = atomicAdd(&hist[(u32)(r[k] >> shift) & mask], 1u);
  }
  __syncthreads();
  u32 cnt = 0;
  if (tid < ndig) { cnt = hist[tid]; if (cnt) gbase[tid] = atomicAdd(&cur[tid], cnt); }
  u32 tot;
  const u32 ex = block_excl_scan<kMsdNW>(cnt, tmp, tot);
  hist[tid] = ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < nvalid) srec[hist[(u32)(r[k] >> shift) & mask] + rk[k]] = r[k];
  }
  __syncthreads();
  for (u32 q = tid; q < nvalid; q += NT) {
    const u64 x = srec[q];
    const u32 dd = (u32)(x >> shift) & mask;
    if (kNT) __builtin_nontemporal_store(msd_word(x), out + gbase[dd] + (q - hist[dd]));
    else out[gbase[dd] + (q - hist[dd])] = msd_word(x);
  }
}
// bucket sizes of the top digit (lab: one global counter per digit)
__global__ __launch_bounds__(1024) void k_lab_hist(const u64 *__restrict__ in, u32 n, u32 shift, u32 *__restrict__ cnt) {
  __shared__ u32 hist[1024];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const u32 begin = blockIdx.x * 65536u;
  for (u32 j = 0; j < 64; j++) { const u32 i = begin + j * 1024 + threadIdx.x; if (i < n) atomicAdd(&hist[(u32)(memw(in[i]) >> shift)], 1u); }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], hist[threadIdx.x]);
}
__global__ __launch_bounds__(1024) void k_lab_scan(const u32 *cnt, u32 n, u32 *cur) {
  __shared__ u32 tmp[16];
  u32 carry = 0;
  for (u32 base = 0; base < n; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < n ? cnt[i] : 0u;
    u32 tot;
    const u32 ex = block_excl_scan<16>(v, tmp, tot) + carry;
    if (i < n) cur[i] = ex;
    carry += tot;
  }
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
#define LSDA(NB, IPT, NW, PF) k_rs_downsweep<Rec8, NB, IPT, NW, PF, ArrayLoader<Rec8>, RecSink<Rec8>, true>, DownsweepSmem<Rec8, IPT, NW, NB>::kBytes, NB, NW * 64 * IPT, NW
    lsd("atomic rank 512x12x16 pf", LSDA(512, 12, 16, true), 2048, false);
    lsd("atomic rank 512x12x8 (2 blocks/CU)", LSDA(512, 12, 8, false), 4096, false);
    lsd("atomic rank 512x12x8 pf (2 blocks/CU)", LSDA(512, 12, 8, true), 4096, false);
    lsd("atomic rank 512x16x8 (2 blocks/CU)", LSDA(512, 16, 8, false), 4096, false);
    lsd("atomic rank 512x8x8 (3 blocks/CU)", LSDA(512, 8, 8, false), 8192, false);
    lsd("atomic rank 512x5x16 (2 blocks/CU)", LSDA(512, 5, 16, false), 4096, false);
    lsd("ballots 512x12x8 (2 blocks/CU)", LSD(512, 12, 8, false), 4096, false);
    lsd("ballots 512x12x8 pf (2 blocks/CU)", LSD(512, 12, 8, true), 4096, false);
  }

  // ---- pass-1 experiments: digit width, tile size, grouped cursors, non-temporal accesses
  if (sections & 4u) {
    u32 *cnt, *cur, *cntg, *stg, *curg;
    CK(hipMalloc(&cnt, 4096 * 4)); CK(hipMalloc(&cur, 4096 * 4));
    CK(hipMalloc(&cntg, 8200 * 4)); CK(hipMalloc(&stg, 8200 * 4)); CK(hipMalloc(&curg, 8200 * 4));
    auto set_attr = [&](auto kern, int ipt) { CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lab_part_smem(ipt))); };
    set_attr(k_lab_part<8, false, false>, 8); set_attr(k_lab_part<8, false, true>, 8); set_attr(k_lab_part<16, false, false>, 16);
    set_attr(k_lab_part<8, true, false>, 8); set_attr(k_lab_part<16, true, false>, 16); set_attr(k_lab_part<4, false, false>, 4); set_attr(k_lab_part<4, true, false>, 4);
    for (u32 d1 : {8u, 9u, 10u}) {
      const u32 sh1 = pbits + nbits - d1, nb1 = 1u << d1;
      CK(hipMemset(cnt, 0, 4096 * 4));
      hipLaunchKernelGGL(k_lab_hist, dim3((n + 65535) / 65536), dim3(1024), 0, 0, (const u64 *)src, n, sh1, cnt);
      auto run = [&](const char *name, auto kern, int ipt, bool grouped) {
        const u32 tile = kMsdNW * 64 * ipt, grid = (n + tile - 1) / tile;
        if (grouped) {
          CK(hipMemset(cntg, 0, 8200 * 4));
          hipLaunchKernelGGL(k_lab_hist_g, dim3((n + 65535) / 65536), dim3(1024), 0, 0, (const u64 *)src, n, sh1, d1, tile, cntg);
          hipLaunchKernelGGL(k_lab_scan, dim3(1), dim3(1024), 0, 0, (const u32 *)cntg, 8 * nb1, stg);   // stg[d*8+g]
        }
        float ms = time_it(reps, [&] {
          if (grouped) hipLaunchKernelGGL(k_lab_transpose_cur, dim3(8), dim3(1024), 0, 0, stg, nb1, curg);
          else hipLaunchKernelGGL(k_lab_scan, dim3(1), dim3(1024), 0, 0, (const u32 *)cnt, nb1, cur);
          hipLaunchKernelGGL(kern, dim3(grid), dim3(kMsdNW * 64), lab_part_smem(ipt), 0, (const u64 *)src, a, n, sh1, d1, grouped ? curg : cur, nb1);
        });
        CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
        hipLaunchKernelGGL(k_check_part, dim3(4096), dim3(256), 0, 0, (const u64 *)a, n, sh1, acc, bad);
        unsigned long long got[2]; u32 nbad;
        CK(hipMemcpy(got, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
        printf("{\"lab\":\"part1\",\"variant\":\"%s\",\"d1\":%u,\"ms\":%.3f,\"TBps_moved\":%.3f,\"ok\":%s}\n", name, d1, ms, 16.0 * n / ms * 1e-9,
               (nbad == 0 && got[0] == ref[0] && got[1] == ref[1]) ? "true" : "false");
      };
      run("tile8192", k_lab_part<8, false, false>, 8, false);
      run("tile8192_nt", k_lab_part<8, false, true>, 8, false);
      run("tile16384", k_lab_part<16, false, false>, 16, false);
      run("tile4096", k_lab_part<4, false, false>, 4, false);
      run("tile8192_grouped", k_lab_part<8, true, false>, 8, true);
      run("tile16384_grouped", k_lab_part<16, true, false>, 16, true);
      run("tile4096_grouped", k_lab_part<4, true, false>, 4, true);
    }
  }

  // ---- MSD pipeline as the product runs it (dc3hip.hip: msd_geometry / msd_sort), timed kernel by kernel
  for (int variant = 0; variant < 3 && (sections & 8u); variant++) {
    const u32 target = variant == 0 ? 10 : variant == 1 ? 11 : 12;      // log2 of the average sub-bucket
    const u32 lgn = bits_of(n - 1);
    u32 tb = lgn > target ? lgn - target : 1;
    if (tb > 20) tb = 20;
    const u32 d1 = tb <= 10 ? tb : (tb + 1) / 2, d2 = tb - d1;
    const u32 sh1 = pbits + nbits - d1, sh2 = sh1 - d2, rb = nbits - tb;
    const u32 nb1 = 1u << d1, n2 = 1u << tb;
    const u32 ntiles1 = (n + kMsdTile - 1) / kMsdTile, tpc = std::max<u32>(1, (ntiles1 + 2047) / 2048);
    const u32 cpg = ((ntiles1 + 7) / 8 + tpc - 1) / tpc, cpx1 = cpg * tpc, chunk = tpc * (u32)kMsdTile, nchunks = (n + chunk - 1) / chunk;
    u32 *table, *cntg, *startg, *cur1, *bstart, *tpre, *tpreh, *plan, *segsum, *cnt2g, *cur2;
    const size_t N = (size_t)n2 * 8;
    CK(hipMalloc(&table, (size_t)1024 * nchunks * 4)); CK(hipMalloc(&cntg, 8200 * 4)); CK(hipMalloc(&startg, 8200 * 4)); CK(hipMalloc(&cur1, 8200 * 4));
    CK(hipMalloc(&bstart, 1040 * 4)); CK(hipMalloc(&tpre, 1040 * 4)); CK(hipMalloc(&tpreh, 1040 * 4)); CK(hipMalloc(&plan, 64)); CK(hipMalloc(&segsum, 1040 * 4));
    CK(hipMalloc(&cnt2g, (N + 16) * 4)); CK(hipMalloc(&cur2, (N + 16) * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_msd_part<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    float t_h1 = time_it(reps, [&] { hipLaunchKernelGGL(k_msd_hist1, dim3(nchunks), dim3(kBlock), 0, 0, (const u64 *)src, n, (u64)0, sh1, chunk, nchunks, table); });
    float t_pl = time_it(reps, [&] {
      CK(hipMemsetAsync(plan, 0, 64, 0));
      hipLaunchKernelGGL(k_msd_cnt1, dim3(nb1), dim3(kBlock), 0, 0, (const u32 *)table, nchunks, cpg, cntg);
      hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, 0, (const u32 *)cntg, nb1, n, startg, cur1, bstart, tpre, tpreh, plan);
    });
    auto plan1 = [&] {
      CK(hipMemsetAsync(plan, 0, 64, 0));
      hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, 0, (const u32 *)cntg, nb1, n, startg, cur1, bstart, tpre, tpreh, plan);
    };
    float t_p1 = time_it(reps, [&] {
      plan1();
      hipLaunchKernelGGL((k_msd_part<false>), dim3(8 * cpx1), dim3(kMsdNW * 64), kMsdPartSmem, 0, (const u64 *)src, a, n, (u64)0, sh1, d1, cpx1, ntiles1,
                         (const u32 *)nullptr, (const u32 *)nullptr, nb1, (const u32 *)plan, cur1, nb1);
    });
    float t_h2 = 0, t_s2 = 0, t_p2 = 0;
    const u64 *sorted2 = a; u64 *other = b; const u32 *starts = startg; u32 nsub = nb1;
    if (d2 > 0) {
      const u32 nseg = (u32)((N + kMsdScanSeg - 1) / kMsdScanSeg);
      t_h2 = time_it(reps, [&] {
        CK(hipMemsetAsync(cnt2g, 0, (N + 1) * 4, 0));
        hipLaunchKernelGGL(k_msd_hist2, dim3(n / kMsdHistTile + nb1 + 1), dim3(1024), 0, 0, (const u64 *)a, (u64)0, sh2, d2, (const u32 *)tpre, (const u32 *)tpreh,
                           (const u32 *)bstart, nb1, (const u32 *)plan, cnt2g);
      });
      t_s2 = time_it(1, [&] {
        hipLaunchKernelGGL(k_msd_scan2a, dim3(nseg), dim3(1024), 0, 0, (const u32 *)cnt2g, (u32)N, segsum, plan);
        hipLaunchKernelGGL(k_msd_scan2c, dim3(nseg), dim3(1024), 0, 0, cnt2g, (u32)N, n2, (const u32 *)segsum, cur2);
      });
      // (the scan is in place: later repetitions of pass 2 restore the cursors from the scanned starts)
      const u32 grid2 = 8 * ((n / kMsdTile + nb1 + 1 + 7) / 8);
      t_p2 = time_it(reps, [&] {
        hipLaunchKernelGGL(k_lab_starts_to_cur, dim3(1024), dim3(256), 0, 0, (const u32 *)cnt2g, n2, cur2);
        hipLaunchKernelGGL((k_msd_part<true>), dim3(grid2), dim3(kMsdNW * 64), kMsdPartSmem, 0, (const u64 *)a, b, n, (u64)0, sh2, d2, 0u, 0u, (const u32 *)tpre,
                           (const u32 *)bstart, nb1, (const u32 *)plan, cur2, n2);
      });
      sorted2 = b; other = a; starts = cnt2g; nsub = n2;
    } else {
      hipLaunchKernelGGL(k_msd_scan2a, dim3((nb1 * 8 + kMsdScanSeg - 1) / kMsdScanSeg), dim3(1024), 0, 0, (const u32 *)cntg, nb1 * 8, segsum, plan);
    }
    u32 hp[4];
    CK(hipMemcpy(hp, plan, 16, hipMemcpyDeviceToHost));
    const u32 maxsub = hp[kMsdW_MAXSUB];
    printf("{\"lab\":\"msd_parts\",\"variant\":%d,\"d1\":%u,\"d2\":%u,\"rem_bits\":%u,\"hist1_ms\":%.3f,\"plan_ms\":%.3f,\"part1_ms\":%.3f,\"hist2_ms\":%.3f,\"scan2_ms\":%.3f,"
           "\"part2_ms\":%.3f,\"max_subbucket\":%u,\"part1_TBps_moved\":%.3f,\"part2_TBps_moved\":%.3f}\n", variant, d1, d2, rb, t_h1, t_pl, t_p1, t_h2, t_s2, t_p2, maxsub,
           16.0 * n / t_p1 * 1e-9, t_p2 > 0 ? 16.0 * n / t_p2 * 1e-9 : 0.0);
    auto run_local = [&](const char *name, auto kern_rec, auto kern_split, u32 cap, u32 bb, u32 nt) {
      if (maxsub > cap) { printf("{\"lab\":\"msd_local\",\"variant\":%d,\"shape\":\"%s\",\"skipped\":\"max sub-bucket %u > cap %u\"}\n", variant, name, maxsub, cap); return; }
      const u32 shb = sh2 - std::min(bb, rb);
      MsdRecSink rs; rs.p = other;
      MsdSplitSink ss; ss.sa = sa; ss.img = img; ss.pbits = pbits;
      float t_rec = time_it(reps, [&] { hipLaunchKernelGGL(kern_rec, dim3(nsub), dim3(nt), cap * 8, 0, sorted2, starts, (u64)0, shb, rs); });
      unsigned long long got[2]; u32 nbad;
      checksum_words(rs.p, got, &nbad);
      float t_split = time_it(reps, [&] { hipLaunchKernelGGL(kern_split, dim3(nsub), dim3(nt), cap * 8, 0, sorted2, starts, (u64)0, shb, ss); });
      CK(hipMemset(acc, 0, 16)); CK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(k_check_split, dim3(4096), dim3(256), 0, 0, sa, img, n, pbits, nbits, skew, acc, bad);
      unsigned long long got2[2]; u32 nbad2;
      CK(hipMemcpy(got2, acc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(&nbad2, bad, 4, hipMemcpyDeviceToHost));
      printf("{\"lab\":\"msd_local\",\"variant\":%d,\"shape\":\"%s\",\"rec_ms\":%.3f,\"split_ms\":%.3f,\"rec_ok\":%s,\"split_ok\":%s,\"bad\":[%u,%u]}\n", variant, name,
             t_rec, t_split, (nbad == 0 && got[0] == ref[0] && got[1] == ref[1]) ? "true" : "false",
             (nbad2 == 0 && got2[0] == ref[0] && got2[1] == ref[1]) ? "true" : "false", nbad, nbad2);
    };
    run_local("256x2048,bb10 (product, small)", k_msd_local<256, 2048, 10, MsdRecSink>, k_msd_local<256, 2048, 10, MsdSplitSink>, 2048, 10, 256);
    run_local("512x4096,bb12 (product, large)", k_msd_local<512, 4096, 12, MsdRecSink>, k_msd_local<512, 4096, 12, MsdSplitSink>, 4096, 12, 512);
    run_local("256x4096,bb12", k_msd_local<256, 4096, 12, MsdRecSink>, k_msd_local<256, 4096, 12, MsdSplitSink>, 4096, 12, 256);
    CK(hipFree(table)); CK(hipFree(cntg)); CK(hipFree(startg)); CK(hipFree(cur1)); CK(hipFree(bstart)); CK(hipFree(tpre)); CK(hipFree(tpreh));
    CK(hipFree(plan)); CK(hipFree(segsum)); CK(hipFree(cnt2g)); CK(hipFree(cur2));
  }
  return 0;
}
