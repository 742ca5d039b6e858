// dc3_merge.hip.hpp — merge tuples, gather, fused mod-0 selection, merge-path merge.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Merge tuples.  Built in slot order with coalesced reads of S and rank (thread g owns text
// positions 3g..3g+2), then gathered into SA12 order — one 16-byte gather per sample suffix
// instead of the 4-6 scattered reads per output of lib.rs:136-162.
// rank = 1-based rank of sample suffixes in slot order, with >= 3 zero words after rank[m02-1].
// ---------------------------------------------------------------------------------------------
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_build_tuples(Sym S, u32 m, u32 m0, u32 m02,
                                                        const u32 *__restrict__ rank, Tup12 *__restrict__ tslot) {
  const bool dummy = (m % 3) == 1;
  __shared__ uint16_t lcode[256];
  S.stage(lcode);
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < m0; g += gridDim.x * kBlock) {
    const u32 j = 3 * g;
    u32 q[4]; S.get4(j, lcode, q);
    const u32 s0 = q[0], s1 = q[1], s2 = q[2], s3 = q[3];
    // mod-1 sample at j+1 (slot g); exists for all g < m0 (dummy when j+1 == m)
    Tup12 a;
    a.pos = j + 1; a.c0 = s1; a.cx = s0;
    a.r = (j + 2 < m) ? rank[m0 + g] : 0u;                       // rank of suffix j+2 (mod 2)
    tslot[g] = a;
    if (j + 2 < m) {                                            // mod-2 sample at j+2 (slot m0+g)
      Tup12 c;
      c.pos = j + 2; c.c0 = s2; c.cx = s3;
      const bool has = (j + 4 < m) || (dummy && j + 4 == m);    // suffix j+4 is mod 1, slot g+1
      c.r = has ? rank[g + 1] : 0u;
      tslot[m0 + g] = c;
    }
  }
}
// Block b gathers the contiguous chunk [b*chunk, (b+1)*chunk) and also histograms, for its mod-1
// entries, the low key byte of the mod-0 tuple each of them yields (c_prev - 1): that is the
// up-sweep of the fused "select mod-0 + first radix pass" below, for free.
__global__ __launch_bounds__(kBlock) void k_gather_tuples(const Tup12 *__restrict__ tslot,
                                                         const u32 *__restrict__ sa12, u32 n, u32 chunk,
                                                         u32 nchunks, Tup12 *__restrict__ out,
                                                         u32 *__restrict__ table /*[256][nchunks]*/) {
  __shared__ u32 hist[kWaves][256];
#pragma unroll
  for (int w = 0; w < kWaves; w++) hist[w][threadIdx.x] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 i = begin + threadIdx.x;
  // the index stream and the output stream are touched once: non-temporal
  const u32x4 *tv = reinterpret_cast<const u32x4 *>(tslot);
  u32x4 *ov = reinterpret_cast<u32x4 *>(out);
  // 4 independent 16-byte gathers in flight per thread;  .x = pos, .w = cx
  for (; i + 3 * kBlock < end; i += 4 * kBlock) {
    const u32 s0 = __builtin_nontemporal_load(&sa12[i]), s1 = __builtin_nontemporal_load(&sa12[i + kBlock]);
    const u32 s2 = __builtin_nontemporal_load(&sa12[i + 2 * kBlock]), s3 = __builtin_nontemporal_load(&sa12[i + 3 * kBlock]);
    const u32x4 a = tv[s0], b = tv[s1], c = tv[s2], d = tv[s3];
    __builtin_nontemporal_store(a, &ov[i]); __builtin_nontemporal_store(b, &ov[i + kBlock]);
    __builtin_nontemporal_store(c, &ov[i + 2 * kBlock]); __builtin_nontemporal_store(d, &ov[i + 3 * kBlock]);
    if (a.x % 3 == 1) atomicAdd(&myh[(a.w - 1u) & 255u], 1u);
    if (b.x % 3 == 1) atomicAdd(&myh[(b.w - 1u) & 255u], 1u);
    if (c.x % 3 == 1) atomicAdd(&myh[(c.w - 1u) & 255u], 1u);
    if (d.x % 3 == 1) atomicAdd(&myh[(d.w - 1u) & 255u], 1u);
  }
  for (; i < end; i += kBlock) {
    const u32x4 a = tv[sa12[i]]; ov[i] = a;
    if (a.x % 3 == 1) atomicAdd(&myh[(a.w - 1u) & 255u], 1u);
  }
  __syncthreads();
  u32 sum = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) sum += hist[w][threadIdx.x];
  table[threadIdx.x * nchunks + blockIdx.x] = sum;
}

// Compact slot table for levels whose symbols fit 16 bits (bytes at level 0, small alphabets below): 8 bytes per
// sample — the rank behind it and both symbols — because the position is a function of the slot (lib.rs:136-144) and
// need not be gathered.  A random 8-byte gather from the half-size table runs 12 % faster on MI355X than the 16-byte
// one (tools/gather_bench.hip: 42.7 vs 38.1 G/s at these sizes) and the streaming build writes half the bytes; the
// gathered output is the same Tup12, so everything downstream is unchanged.
struct __attribute__((aligned(8))) TupS8 { u32 r, cc; };     // cc = c0 | cx << 16
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_build_tuples8(Sym S, u32 m, u32 m0, u32 m02, const u32 *__restrict__ rank,
                                                         TupS8 *__restrict__ tslot) {
  const bool dummy = (m % 3) == 1;
  __shared__ uint16_t lcode[256];
  S.stage(lcode);
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < m0; g += gridDim.x * kBlock) {
    const u32 j = 3 * g;
    u32 q[4]; S.get4(j, lcode, q);
    TupS8 a;                                                     // mod-1 sample at j+1 (slot g): c0 = S[j+1], cx = S[j]
    a.cc = q[1] | (q[0] << 16);
    a.r = (j + 2 < m) ? rank[m0 + g] : 0u;
    tslot[g] = a;
    if (j + 2 < m) {                                            // mod-2 sample at j+2 (slot m0+g): c0 = S[j+2], cx = S[j+3]
      TupS8 c;
      c.cc = q[2] | (q[3] << 16);
      const bool has = (j + 4 < m) || (dummy && j + 4 == m);
      c.r = has ? rank[g + 1] : 0u;
      tslot[m0 + g] = c;
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_gather_tuples8(const TupS8 *__restrict__ tslot, const u32 *__restrict__ sa12,
                                                          u32 n, u32 m0, u32 chunk, u32 nchunks, Tup12 *__restrict__ out,
                                                          u32 *__restrict__ table /*[256][nchunks]*/) {
  __shared__ u32 hist[kWaves][256];
#pragma unroll
  for (int w = 0; w < kWaves; w++) hist[w][threadIdx.x] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 i = begin + threadIdx.x;
  typedef u32 u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 *tv = reinterpret_cast<const u32x2 *>(tslot);
  u32x4 *ov = reinterpret_cast<u32x4 *>(out);
  auto emit = [&](u32 idx, u32 s, u32x2 v) {
    const bool mod1 = s < m0;
    u32x4 o;
    o.x = mod1 ? 3 * s + 1 : 3 * (s - m0) + 2;                   // position of the sample in slot s
    o.y = v.x; o.z = v.y & 0xffffu; o.w = v.y >> 16;
    __builtin_nontemporal_store(o, &ov[idx]);
    if (mod1) atomicAdd(&myh[(o.w - 1u) & 255u], 1u);
  };
  for (; i + 3 * kBlock < end; i += 4 * kBlock) {
    const u32 s0 = __builtin_nontemporal_load(&sa12[i]), s1 = __builtin_nontemporal_load(&sa12[i + kBlock]);
    const u32 s2 = __builtin_nontemporal_load(&sa12[i + 2 * kBlock]), s3 = __builtin_nontemporal_load(&sa12[i + 3 * kBlock]);
    const u32x2 a = tv[s0], b = tv[s1], c = tv[s2], d = tv[s3];
    emit(i, s0, a); emit(i + kBlock, s1, b); emit(i + 2 * kBlock, s2, c); emit(i + 3 * kBlock, s3, d);
  }
  for (; i < end; i += kBlock) { const u32 s = sa12[i]; emit(i, s, tv[s]); }
  __syncthreads();
  u32 sum = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) sum += hist[w][threadIdx.x];
  table[threadIdx.x * nchunks + blockIdx.x] = sum;
}

// ---------------------------------------------------------------------------------------------
// The same sample tuples WITHOUT the random gather: scattered to their rank by two partition passes and a window
// placement (the windowed inversion of dc3_order.hip.hpp with a payload).  The gather fetches a 64-byte line per
// 8-byte tuple at ~35 G gathers/s (82 B of HBM traffic per sample, 1.7 TB/s); the scatter moves about as many bytes
// as streams.  Record of slot s: (dest = rank12[s] - 1, r, cc) — 12 bytes; dest is a bijection onto [0, m02), so the
// region of every bucket and window is known analytically and nothing has to be counted except how a bucket's region
// splits between the XCD groups in pass 1 (k_tup_hist1: 4 bytes read per slot).
//   k_tup_hist1  digit table of dest >> kTupSh1 per chunk of slots (same format as the pack kernels' tables; the
//                per-group bucket sizes and cursors then come from k_msd_cnt1 / k_msd_plan1)
//   k_tup_part1  builds the records of a tile of slots (coalesced reads of S and rank) and partitions them by
//                dest >> kTupSh1, XCD-grouped like k_msd_part
//   k_tup_part2  inside every 2^22-record bucket: by (dest >> kTupWinBits) & 511 into the windows; bucket b is worked by
//                the XCD group b % 8 (all writes into a bucket go through one L2)
//   k_tup_local  window w: payloads placed in LDS by dest, then written out in order as Tup12 next to sa12 (which gives
//                the position), with the c_prev histogram of the fused mod-0 pass
// ---------------------------------------------------------------------------------------------
constexpr u32 kTupSh1 = 22, kTupWinBits = 13, kTupWin = 1u << kTupWinBits;       // 8192 destinations per window
constexpr int kTupNT = 512, kTupIPT = 8, kTupTile = kTupNT * kTupIPT;             // 4096 records per partition tile
constexpr size_t kTupPartSmem = sizeof(u32) * (3 * kTupTile + 2 * 1024 + 64);
struct TupRec { u32 dest, r, cc; };

// gstride 0 (round 5): dest is a bijection onto [0, m02), so bucket d of pass 1 is exactly the region [d << 22, ...) and a
// tile only reserves inside it — ONE cursor per bucket, starting at the bucket's base, shared by all blocks; no counting
// pass (k_tup_hist1 / k_msd_cnt1 / k_msd_plan1: 1.4 ms per GiB of text, 4 bytes read per slot) and no split of a bucket
// between the XCD groups: a tile's run in a bucket is ~36 records (290-430 bytes), long enough without it (as k_part_msd).
__global__ void k_tup_cur_init(u32 *__restrict__ cur, u32 nb) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < nb) cur[d] = d << kTupSh1;
}
__global__ __launch_bounds__(kBlock) void k_tup_hist1(const u32 *__restrict__ rank12, u32 m02, u32 chunk, u32 nchunks,
                                                     u32 *__restrict__ table /*[1024][nchunks]*/) {
  __shared__ u32 hist[kWaves][1024];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < 1024; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(m02, begin + chunk);
  for (u32 s = begin + threadIdx.x; s < end; s += kBlock) atomicAdd(&myh[(rank12[s] - 1u) >> kTupSh1], 1u);
  __syncthreads();
  for (int j = threadIdx.x; j < 1024; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}

// the (r, cc) payload of the sample in slot s (the fields of k_build_tuples8)
// Branch-free (one text load of two symbols, one rank load whatever the residue of the sample: the loads of a thread's
// eight slots are issued back to back instead of behind each slot's branches).
template <class Sym>
__device__ __forceinline__ void tup_payload(const Sym &S, const uint16_t *lcode, u32 m, u32 m0, bool dummy, const u32 *__restrict__ rank,
                                            u32 s, u32 &r, u32 &cc) {
  // mod-1 sample at 3s+1 (slot s < m0): c0 = S[3s+1], cx = S[3s], r = rank of suffix 3s+2 = slot m0 + s
  // mod-2 sample at 3g+2 (slot m0 + g): c0 = S[3g+2], cx = S[3g+3], r = rank of suffix 3g+4 = slot g + 1
  const bool mod1 = s < m0;
  const u32 g = mod1 ? s : s - m0, j = 3 * g;
  u32 q[2];
  S.get2(j + (mod1 ? 0u : 2u), lcode, q);
  cc = mod1 ? (q[1] | (q[0] << 16)) : (q[0] | (q[1] << 16));
  const bool has = mod1 ? (j + 2 < m) : ((j + 4 < m) || (dummy && j + 4 == m));
  const u32 rv = rank[has ? (mod1 ? m0 + s : g + 1) : 0u];
  r = has ? rv : 0u;
}

// shared tail of the two partition passes: the tile's records sit in registers (dest/r/cc[k], valid for t < nvalid,
// t = k * kTupNT + tid); rank them by digit in LDS, reserve, reorder, write the runs.
template <class DigitOf, class BaseOf, class Emit>
__device__ __forceinline__ void tup_partition_tile(const u32 (&dest)[kTupIPT], const u32 (&rr)[kTupIPT], const u32 (&cc)[kTupIPT],
                                                   u32 nvalid, u32 ndig, DigitOf digit_of, BaseOf reserve, Emit emit,
                                                   unsigned char *smem) {
  u32 *sd = reinterpret_cast<u32 *>(smem), *sr = sd + kTupTile, *sc = sr + kTupTile;
  u32 *hist = sc + kTupTile, *gbase = hist + 1024, *tmp = gbase + 1024;
  const u32 tid = threadIdx.x;
  for (u32 j = tid; j < 1024; j += kTupNT) hist[j] = 0;
  __syncthreads();
  u32 rk[kTupIPT];
#pragma unroll
  for (int k = 0; k < kTupIPT; k++) {
    const u32 t = k * kTupNT + tid;
    if (t < nvalid) rk[k] = atomicAdd(&hist[digit_of(dest[k])], 1u);
  }
  __syncthreads();
  // digits of thread tid: 2 * tid, 2 * tid + 1 (ndig <= 1024 = 2 * kTupNT)
  u32 c0 = 0, c1 = 0;
  const u32 d0 = 2 * tid, d1 = 2 * tid + 1;
  if (d0 < ndig) { c0 = hist[d0]; if (c0) gbase[d0] = reserve(d0, c0); }
  if (d1 < ndig) { c1 = hist[d1]; if (c1) gbase[d1] = reserve(d1, c1); }
  u32 tot;
  const u32 ex = block_excl_scan<kTupNT / 64>(c0 + c1, tmp, tot);
  hist[d0] = ex; hist[d1] = ex + c0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kTupIPT; k++) {
    const u32 t = k * kTupNT + tid;
    if (t < nvalid) { const u32 q = hist[digit_of(dest[k])] + rk[k]; sd[q] = dest[k]; sr[q] = rr[k]; sc[q] = cc[k]; }
  }
  __syncthreads();
  for (u32 q = tid; q < nvalid; q += kTupNT) {
    const u32 d = sd[q], dd = digit_of(d);
    emit(gbase[dd] + (q - hist[dd]), d, sr[q], sc[q]);
  }
}
// what a partition pass leaves per sample:
//   TupOut12  (dest, r, cc) as three words — any level whose symbols fit 16 bits (cc = c0 | cx << 16)
//   TupOut8   one 64-bit word  cx << (22 + rb) | r << 22 | (dest & (2^22 - 1))  — level 0 (bytes: cx has 9 bits, r has
//             rb = bits of m02 <= 32): the top bits of dest are the bucket the record lies in, and c0 is not carried at all
//             — in SA12 order the first symbols are non-decreasing, so k_tup_local reads c0 off a 258-entry table of
//             cumulative first-symbol counts (k_sample_sym_hist)
struct TupOut12 {
  TupRec *p;
  typedef TupRec Rec;
  __device__ __forceinline__ void operator()(u32 i, u32 d, u32 r, u32 cc) const { p[i] = TupRec{d, r, cc}; }
  __device__ __forceinline__ static void unpack(const TupRec &x, u32, u32 &d, u32 &r, u32 &cc) { d = x.dest; r = x.r; cc = x.cc; }
};
struct TupOut8 {
  u64 *p; u32 rb;
  typedef u64 Rec;
  __device__ __forceinline__ void operator()(u32 i, u32 d, u32 r, u32 cc) const {
    p[i] = (u64)(d & ((1u << kTupSh1) - 1u)) | ((u64)r << kTupSh1) | ((u64)(cc >> 16) << (kTupSh1 + rb));
  }
  // d = dest & (2^22 - 1): what the second pass and the window placement need of it;  cc = cx << 16 (c0 absent)
  __device__ __forceinline__ static void unpack(u64 x, u32 rb, u32 &d, u32 &r, u32 &cc) {
    d = (u32)x & ((1u << kTupSh1) - 1u);
    r = (u32)(x >> kTupSh1) & (rb >= 32 ? 0xffffffffu : ((1u << rb) - 1u));
    cc = (u32)(x >> (kTupSh1 + rb)) << 16;
  }
};

template <class Sym, class Out>
__global__ __launch_bounds__(kTupNT) void k_tup_part1(Sym S, u32 m, u32 m0, u32 m02, const u32 *__restrict__ rank12, u32 cpx,
                                                     u32 ntiles, u32 ndig, u32 *__restrict__ cursors /*[8][ndig], or [ndig] with gstride 0*/,
                                                     Out out, u32 gstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint16_t lcode[256];
  const u32 g = blockIdx.x % 8u, idx = blockIdx.x / 8u;
  const u32 tile = g * cpx + idx;
  if (idx >= cpx || tile >= ntiles) return;
  const bool dummy = (m % 3) == 1;
  S.stage(lcode);
  const u32 begin = tile * (u32)kTupTile, nvalid = min((u32)kTupTile, m02 - begin);
  u32 dest[kTupIPT], rr[kTupIPT], cc[kTupIPT];
#pragma unroll
  for (int k = 0; k < kTupIPT; k++) {
    const u32 s = begin + min((u32)(k * kTupNT) + threadIdx.x, nvalid - 1u);
    dest[k] = rank12[s] - 1u;
    tup_payload(S, lcode, m, m0, dummy, rank12, s, rr[k], cc[k]);
  }
  u32 *cur = cursors + (size_t)g * gstride;
  tup_partition_tile(dest, rr, cc, nvalid, ndig, [](u32 d) { return d >> kTupSh1; },
                     [&](u32 d, u32 cnt) { return atomicAdd(&cur[d], cnt); }, out, smem);
}

// pass 2: bucket b = records [b << 22, min(m02, (b + 1) << 22)); tile list: bucket b on the XCD group b % 8
template <class Out>
__global__ __launch_bounds__(kTupNT) void k_tup_part2(const typename Out::Rec *__restrict__ in, u32 m02, u32 nbuckets,
                                                     u32 *__restrict__ cursors /*[nbuckets][512], zeroed*/, Out out, u32 rb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr u32 tpb = (1u << kTupSh1) / kTupTile;         // tiles per bucket
  const u32 g = blockIdx.x % 8u, idx = blockIdx.x / 8u;
  const u32 b = g + 8u * (idx / tpb);
  if (b >= nbuckets) return;
  const u64 begin64 = ((u64)b << kTupSh1) + (u64)(idx % tpb) * kTupTile;
  if (begin64 >= m02) return;
  const u32 begin = (u32)begin64, nvalid = min((u32)kTupTile, m02 - begin);
  u32 dest[kTupIPT], rr[kTupIPT], cc[kTupIPT];
#pragma unroll
  for (int k = 0; k < kTupIPT; k++) {
    const typename Out::Rec x = in[begin + min((u32)(k * kTupNT) + threadIdx.x, nvalid - 1u)];
    Out::unpack(x, rb, dest[k], rr[k], cc[k]);
  }
  u32 *cur = cursors + ((size_t)b << 9);
  const u32 base = b << kTupSh1;
  // ((d >> 13) & 511 is the same digit whether d is the whole destination or its low 22 bits)
  tup_partition_tile(dest, rr, cc, nvalid, 512u, [](u32 d) { return (d >> kTupWinBits) & 511u; },
                     [&](u32 d, u32 cnt) { return base + (d << kTupWinBits) + atomicAdd(&cur[d], cnt); }, out, smem);
}

// Pass 1 of the 8-byte form in the shape of the bucket ordering's partition pass (k_msd_part): 1024 threads, tiles of
// 6144 slots, the packed words and a 2-byte digit per word in LDS (70 KB: two blocks per CU, 32 waves) instead of three
// 4-byte planes per record in tiles of 4096 (one tile per 13 us chain of load -> rank -> reserve -> reorder -> write, two
// blocks of 8 waves per CU): the chain is walked once per 6144 records by twice the waves, and a tile's run in a bucket
// is 1.5 times as long.
constexpr int kTup8NT = 1024, kTup8IPT = 6, kTup8Tile = kTup8NT * kTup8IPT;
constexpr size_t kTup8PartSmem = sizeof(u64) * kTup8Tile + sizeof(uint16_t) * kTup8Tile + sizeof(u32) * (2 * 1024 + 64);
template <class Sym>
__global__ __launch_bounds__(kTup8NT, 8) void k_tup8_part1(Sym S, u32 m, u32 m0, u32 m02, const u32 *__restrict__ rank12, u32 cpx,
                                                       u32 ntiles, u32 ndig, u32 *__restrict__ cursors /*[8][ndig], or [ndig] with gstride 0*/,
                                                       TupOut8 out, u32 gstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64 *srec = reinterpret_cast<u64 *>(smem);
  uint16_t *sdig = reinterpret_cast<uint16_t *>(smem + sizeof(u64) * kTup8Tile);
  u32 *hist = reinterpret_cast<u32 *>(smem + sizeof(u64) * kTup8Tile + sizeof(uint16_t) * kTup8Tile);
  u32 *gbase = hist + 1024, *tmp = gbase + 1024;
  __shared__ uint16_t lcode[256];
  const u32 tid = threadIdx.x;
  const u32 g = blockIdx.x % 8u, idx = blockIdx.x / 8u;
  const u32 tile = g * cpx + idx;
  if (idx >= cpx || tile >= ntiles) return;
  const bool dummy = (m % 3) == 1;
  S.stage(lcode);
  hist[tid] = 0;
  __syncthreads();
  const u32 begin = tile * (u32)kTup8Tile, nvalid = min((u32)kTup8Tile, m02 - begin);
  u64 w[kTup8IPT];
  u32 dg[kTup8IPT], rk[kTup8IPT];
#pragma unroll
  for (int k = 0; k < kTup8IPT; k++) {
    const u32 s = begin + min((u32)(k * kTup8NT) + tid, nvalid - 1u);
    const u32 dest = rank12[s] - 1u;
    u32 r, cc;
    tup_payload(S, lcode, m, m0, dummy, rank12, s, r, cc);
    dg[k] = dest >> kTupSh1;
    w[k] = (u64)(dest & ((1u << kTupSh1) - 1u)) | ((u64)r << kTupSh1) | ((u64)(cc >> 16) << (kTupSh1 + out.rb));
  }
#pragma unroll
  for (int k = 0; k < kTup8IPT; k++)
    if ((u32)(k * kTup8NT) + tid < nvalid) rk[k] = atomicAdd(&hist[dg[k]], 1u);
  __syncthreads();
  u32 *cur = cursors + (size_t)g * gstride;
  u32 cnt = 0;
  if (tid < ndig) { cnt = hist[tid]; if (cnt) gbase[tid] = atomicAdd(&cur[tid], cnt); }
  u32 tot;
  const u32 ex = block_excl_scan<kTup8NT / 64>(cnt, tmp, tot);
  hist[tid] = ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kTup8IPT; k++)
    if ((u32)(k * kTup8NT) + tid < nvalid) { const u32 at = hist[dg[k]] + rk[k]; srec[at] = w[k]; sdig[at] = (uint16_t)dg[k]; }
  __syncthreads();
  for (u32 q = tid; q < nvalid; q += kTup8NT) {
    const u32 dd = sdig[q];
    out.p[gbase[dd] + (q - hist[dd])] = srec[q];
  }
}

// Cumulative first-symbol counts of the level's sample suffixes (level 0: codes 0..sigma, 0 = the dummy sample behind
// the text): hist[c] += samples whose first symbol is c.  One pass over S; `hist` (nsym words, zeroed) is turned into the
// exclusive prefix cum[0..nsym] by k_scan_excl_inplace.  A sample of SA12 rank k starts with the symbol c for which
// cum[c] <= k < cum[c + 1] — SA12 is sorted, its first symbols are non-decreasing.
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_sample_sym_hist(Sym S, u32 m, u32 m0, u32 nsym, u32 *__restrict__ hist) {
  __shared__ u32 h[kWaves][264];
  __shared__ uint16_t lcode[256];
  for (u32 j = threadIdx.x; j < kWaves * 264; j += kBlock) (&h[0][0])[j] = 0;
  S.stage(lcode);
  __syncthreads();
  u32 *myh = h[wave_id()];
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < m0; g += gridDim.x * kBlock) {
    const u32 j = 3 * g;
    u32 q[4]; S.get4(j, lcode, q);
    atomicAdd(&myh[q[1]], 1u);                       // mod-1 sample at j + 1 (the dummy when j + 1 == m: symbol 0)
    if (j + 2 < m) atomicAdd(&myh[q[2]], 1u);        // mod-2 sample at j + 2
  }
  __syncthreads();
  for (u32 c = threadIdx.x; c < nsym; c += kBlock) {
    const u32 v = h[0][c] + h[1][c] + h[2][c] + h[3][c];
    if (v) atomicAdd(&hist[c], v);
  }
}

// Compact merge tuples (levels whose symbols fit 16 bits): 12 bytes per sample and 16 per mod-0 suffix instead of 16 and
// 20 — the unwinding of a level (tuples, mod-0 order, merge: lib.rs:118-192) is pure record traffic.
struct TupC { u32 pos, r, cc; };                                      // cc = c0 | cx << 16   (fields as Tup12)
struct __attribute__((aligned(16))) Tup0C { u32 pos, cc, r1, r2; };   // cc = c0 | c1 << 16   (fields as Tup0)

// window w: the records [w * kTupWin, ...) are exactly those with dest in the window
// kDerive (8-byte records): c0 is not in the record; cum[0..nsym] gives it (see k_sample_sym_hist).
template <class Out, bool kDerive>
__global__ __launch_bounds__(1024) void k_tup_local(const typename Out::Rec *__restrict__ in, u32 rb, const u32 *__restrict__ sa12, u32 m02, u32 m0,
                                                   u32 chunk, u32 nchunks, const u32 *__restrict__ cum, u32 nsym,
                                                   TupC *__restrict__ out, u32 *__restrict__ table /*[256][nchunks], zeroed*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 * kTupWin words (dynamic: 64 KiB)
  u32 *wr = reinterpret_cast<u32 *>(smem), *wc = wr + kTupWin;
  __shared__ u32 hist[2][256];
  __shared__ u32 scum[kDerive ? 264 : 1];
  const u32 base = blockIdx.x * kTupWin, cnt = min(kTupWin, m02 - base);
  if (threadIdx.x < 512) hist[threadIdx.x >> 8][threadIdx.x & 255] = 0;
  if (kDerive && threadIdx.x <= nsym) scum[threadIdx.x] = cum[threadIdx.x];            // nsym <= 258
  for (u32 i = threadIdx.x; i < cnt; i += 1024) {
    u32 d, r, cc;
    Out::unpack(in[base + i], rb, d, r, cc);
    wr[d & (kTupWin - 1u)] = r; wc[d & (kTupWin - 1u)] = cc;
  }
  __syncthreads();
  u32 c0 = 0;
  if (kDerive) {
    // symbol of the window's first rank: the largest c with cum[c] <= base (block-uniform, broadcast reads)
    u32 lo = 0, hi = nsym;                    // invariant: cum[lo] <= base < cum[hi]   (cum[0] = 0, cum[nsym] = m02 > base)
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (scum[mid] <= base) lo = mid; else hi = mid; }
    c0 = lo;
  }
  const u32 c_lo = base / chunk;                          // a window touches at most two chunks of the mod-0 pass (chunk >= kTupWin)
  for (u32 i = threadIdx.x; i < cnt; i += 1024) {
    const u32 s = sa12[base + i];
    const bool mod1 = s < m0;
    u32 cc = wc[i];
    if (kDerive) {
      while (scum[c0 + 1] <= base + i) c0++;              // (ranks grow with i: the walk never goes back)
      cc |= c0;
    }
    out[base + i] = TupC{mod1 ? 3 * s + 1 : 3 * (s - m0) + 2, wr[i], cc};
    if (mod1) atomicAdd(&hist[(base + i) / chunk - c_lo][((cc >> 16) - 1u) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 512) {
    const u32 h = threadIdx.x >> 8, d = threadIdx.x & 255, v = hist[h][d];
    if (v && c_lo + h < nchunks) atomicAdd(&table[(size_t)d * nchunks + c_lo + h], v);
  }
}

// ---------------------------------------------------------------------------------------------
// Step 2 (lib.rs:118-125): order-preserving selection of the mod-1 entries of SA12; each yields
// the mod-0 suffix one position to the left, already ordered by rank of suffix j+1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_mod1(u32 pos) { return pos % 3 == 1; }

// The two tuple formats behind one face: a sample tuple as (pos, r, c0, cx), a mod-0 tuple as its comparison key
// (c0, c1, r1, r2) + pos.
__device__ __forceinline__ u32x4 tupa_words(const Tup12 &a) { u32x4 v; v.x = a.pos; v.y = a.r; v.z = a.c0; v.w = a.cx; return v; }
__device__ __forceinline__ u32x4 tupa_words(const TupC &a) { u32x4 v; v.x = a.pos; v.y = a.r; v.z = a.cc & 0xffffu; v.w = a.cc >> 16; return v; }
__device__ __forceinline__ u32x4 tupb_key(const Tup0 &z) { u32x4 k; k.x = z.c0; k.y = z.c1; k.z = z.r1; k.w = z.r2; return k; }
__device__ __forceinline__ u32x4 tupb_key(const Tup0C &z) { u32x4 k; k.x = z.cc & 0xffffu; k.y = z.cc >> 16; k.z = z.r1; k.w = z.r2; return k; }
template <class TA> struct Mod0Of;
template <> struct Mod0Of<Tup12> { typedef Tup0 type; };
template <> struct Mod0Of<TupC> { typedef Tup0C type; };

// Loader of the fused Step-2 pass: element i of the sorted sample tuples yields a mod-0 tuple iff it
// is a mod-1 suffix; r1 = i+1 is the rank of suffix j+1, so the stream is already ordered by it.
struct Mod0Loader {
  const Tup12 *t;
  __device__ __forceinline__ bool load(u32 i, Tup0 &z) const {
    const Tup12 a = t[i];
    if (!is_mod1(a.pos)) return false;
    z.pos = a.pos - 1; z.c0 = a.cx; z.c1 = a.c0; z.r1 = i + 1; z.r2 = a.r;
    return true;
  }
};
struct Mod0LoaderC {
  const TupC *t;
  __device__ __forceinline__ bool load(u32 i, Tup0C &z) const {
    const TupC a = t[i];
    if (!is_mod1(a.pos)) return false;
    z.pos = a.pos - 1; z.cc = (a.cc >> 16) | (a.cc << 16); z.r1 = i + 1; z.r2 = a.r;     // (c0, c1) = (cx, c0) of the sample
    return true;
  }
};
// mod-0 positions are real symbols (c0 >= 1), so the key is c0-1 in [0, K)
__device__ __forceinline__ u32 digit_of(const Tup0C &r, KeyDig d) { return (((r.cc & 0xffffu) - 1u) >> d.shift) & d.mask; }

// ---------------------------------------------------------------------------------------------
// Step 3 (lib.rs:131-192): merge of SA12 and SA0 as a merge-path merge.
// Comparator = leq2 / leq3 of lib.rs:3-11 in Kärkkäinen–Sanders argument order (the reference's
// leq3 parameter list is scrambled, lib.rs:9 vs :154-161).  Suffixes are distinct, so < == <=.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sample_before4(const u32x4 a /*pos,r,c0,cx*/, const u32x4 z /*c0,c1,r1,r2*/) {
  if (is_mod1(a.x)) return (a.z < z.x) || (a.z == z.x && a.y <= z.z);                                   // leq2
  return (a.z < z.x) || (a.z == z.x && ((a.w < z.y) || (a.w == z.y && a.y <= z.w)));                    // leq3
}

__device__ __forceinline__ bool sample_before(const Tup12 &a, const Tup0 &z) { return sample_before4(tupa_words(a), tupb_key(z)); }

// Merge-path split points: part[t] = number of A elements among the first t*tile outputs.
// Two levels: `coarse` (optional) holds the split of every `ratio`-th tile boundary, which bounds the
// binary search of the tiles in between to a window of ratio*tile elements (L2-resident, ~half the
// dependent steps) — the unbounded search over 10^9 elements fetched 15 GB per build.
template <class TA, class TB>
__global__ __launch_bounds__(kBlock) void k_merge_partition(const TA *__restrict__ A, u32 nA,
                                                           const TB *__restrict__ B, u32 nB, u32 ntiles,
                                                           u32 tile, const u32 *__restrict__ coarse, u32 ratio,
                                                           u32 *__restrict__ part /*[ntiles+1]*/) {
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t > ntiles) return;
  const u32 total = nA + nB;
  const u32 diag = (u32)min((u64)t * tile, (u64)total);
  u32 lo = diag > nB ? diag - nB : 0u, hi = min(diag, nA);
  if (coarse) {
    const u32 cidx = t / ratio;
    if (cidx * ratio == t) { part[t] = coarse[cidx]; return; }     // on a coarse boundary
    lo = max(lo, coarse[cidx]);
    hi = min(hi, coarse[cidx + 1]);
  }
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);     // (lo + hi) would overflow u32 beyond 2^31 samples
    if (sample_before4(tupa_words(A[mid]), tupb_key(B[diag - 1 - mid]))) lo = mid + 1; else hi = mid;
  }
  part[t] = lo;
}

// out_sa[k] = text position of the k-th smallest suffix (coalesced); out_pairs[k] = (pos, k+1) feeds
// the windowed inversion that gives the parent level rank[pos] = k+1 (R[SA12[i]] = i+1, lib.rs:106-108).
// NT threads, VT outputs per thread; the tile's inputs are staged in LDS, outputs are staged in LDS
// too so that global stores are coalesced.
// LDS image of a tile: sample tuples as 16-byte words (pos, r, c0, cx); mod-0 tuples split into a
// 16-byte comparison key (c0, c1, r1, r2) and a separate pos array, so that every comparison is two
// ds_read_b128 (the packed 20-byte Tup0 would be five ds_read_b32).
template <int NT, int VT>
struct MergeSmem { static constexpr size_t kBytes = (16 + 16 + 4) * (size_t)(NT * VT) + 64; };

// kKeys64 (compact tuples whose symbols fit 15 bits: level 0, and level 1 of texts): the tile's LDS image holds ONE 64-bit
// comparison key per sample — (c0, r) for a mod-1 sample, bit 63 set; (c0, c1, r) for a mod-2 sample — and BOTH keys of a
// mod-0 tuple, (c0, r1) and (c0, c1, r2), side by side: a comparison is two ds_read_b64 and one 64-bit compare instead of two
// ds_read_b128, a position mod 3 and leq2 / leq3 field by field (the kernel was bound by vector instructions: 65 % of the
// SIMD time at 1024 x 2, profiles/r05a_pmc_text SQ counters).
__device__ __forceinline__ u64 merge_key_a(u32 pos, u32 r, u32 c0, u32 cx) {
  return is_mod1(pos) ? ((((u64)c0 << 32) | r) | (1ull << 63)) : (((u64)c0 << 47) | ((u64)cx << 32) | r);
}
template <int NT, int VT, class TA, class TB, bool kKeys64 = false>
__global__ __launch_bounds__(NT) void k_merge(const TA *__restrict__ A, u32 nA, const TB *__restrict__ B, u32 nB,
                                             const u32 *__restrict__ part, u32 *__restrict__ out_sa,
                                             Rec8 *__restrict__ out_pairs, u32 rank_base) {
  constexpr u32 kTile = NT * VT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const u32 total = nA + nB;
  const u32 d0 = blockIdx.x * kTile;
  const u32 d1 = min(d0 + kTile, total);
  const u32 a0 = part[blockIdx.x], a1 = part[blockIdx.x + 1];
  const u32 b0 = d0 - a0, b1 = d1 - a1;
  const u32 na = a1 - a0, nb = b1 - b0;
  const u32 dl = min(threadIdx.x * (u32)VT, na + nb);
  u32 outp[VT];
  if constexpr (kKeys64) {
    u64 *sk = reinterpret_cast<u64 *>(smem);                         // [kTile]     sample keys
    u64 *sb = reinterpret_cast<u64 *>(smem + 8 * kTile);             // [kTile][2]  mod-0 keys: against a mod-1 / a mod-2 sample
    u32 *spos = reinterpret_cast<u32 *>(smem + 24 * kTile);          // [kTile]
    u32 *sbpos = reinterpret_cast<u32 *>(smem + 28 * kTile);         // [kTile]
    for (u32 i = threadIdx.x; i < na; i += NT) {
      const u32x4 a = tupa_words(A[a0 + i]);
      sk[i] = merge_key_a(a.x, a.y, a.z, a.w); spos[i] = a.x;
    }
    for (u32 i = threadIdx.x; i < nb; i += NT) {
      const TB z = B[b0 + i];
      const u32x4 k = tupb_key(z);
      sb[2 * i] = ((u64)k.x << 32) | k.z; sb[2 * i + 1] = ((u64)k.x << 47) | ((u64)k.y << 32) | k.w;
      sbpos[i] = z.pos;
    }
    __syncthreads();
    auto before = [&](u32 i, u32 j) -> bool {                        // sample i of the tile before mod-0 suffix j?
      const u64 ks = sk[i];
      const u64 kb = sb[2 * j + ((ks >> 63) ? 0u : 1u)];
      return (ks & ~(1ull << 63)) <= kb;
    };
    u32 lo = dl > nb ? dl - nb : 0u, hi = min(dl, na);
    while (lo < hi) {
      const u32 mid = lo + ((hi - lo) >> 1);
      if (before(mid, dl - 1 - mid)) lo = mid + 1; else hi = mid;
    }
    u32 ai = lo, bi = dl - lo;
#pragma unroll
    for (int v = 0; v < VT; v++) {
      const u32 k = dl + v;
      outp[v] = 0;
      if (k < na + nb) {
        const bool takeA = (bi >= nb) || (ai < na && before(ai, bi));
        outp[v] = takeA ? spos[ai] : sbpos[bi];
        ai += takeA ? 1u : 0u; bi += takeA ? 0u : 1u;
      }
    }
  } else {
  u32x4 *sa = reinterpret_cast<u32x4 *>(smem);
  u32x4 *sbk = reinterpret_cast<u32x4 *>(smem + 16 * kTile);
  u32 *sbpos = reinterpret_cast<u32 *>(smem + 32 * kTile);
  for (u32 i = threadIdx.x; i < na; i += NT) sa[i] = tupa_words(A[a0 + i]);
  for (u32 i = threadIdx.x; i < nb; i += NT) {
    const TB z = B[b0 + i];
    sbk[i] = tupb_key(z); sbpos[i] = z.pos;
  }
  __syncthreads();
  u32 lo = dl > nb ? dl - nb : 0u, hi = min(dl, na);
  while (lo < hi) {
    const u32 mid = lo + ((hi - lo) >> 1);     // (lo + hi) would overflow u32 beyond 2^31 samples
    if (sample_before4(sa[mid], sbk[dl - 1 - mid])) lo = mid + 1; else hi = mid;
  }
  u32 ai = lo, bi = dl - lo;
#pragma unroll
  for (int v = 0; v < VT; v++) {
    const u32 k = dl + v;
    outp[v] = 0;
    if (k < na + nb) {
      const bool takeA = (bi >= nb) || (ai < na && sample_before4(sa[ai], sbk[bi]));
      outp[v] = takeA ? sa[ai].x : sbpos[bi];
      ai += takeA ? 1u : 0u; bi += takeA ? 0u : 1u;
    }
  }
  }
  __syncthreads();                       // inputs are dead: reuse the front of LDS as the output stage
  u32 *so = reinterpret_cast<u32 *>(smem);
#pragma unroll
  for (int v = 0; v < VT; v++) so[threadIdx.x * VT + v] = outp[v];
  __syncthreads();
  const u32 nout = d1 - d0;
  for (u32 q = threadIdx.x; q < nout; q += NT) {
    const u32 pos = so[q];
    if (out_sa) out_sa[d0 + q] = pos;
    if (out_pairs) out_pairs[d0 + q] = Rec8{pos, rank_base + d0 + q + 1};   // rank_base: ranks of a slice of the merge (global mode)
  }
}

}  // namespace dc3
