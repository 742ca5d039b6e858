// dc3_order.hip.hpp — naming, windowed inverse permutation, discarding recursion, prefix-sort + tie-refine.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Naming (lib.rs:80-100): name = 1 + number of key changes before i in the sorted order.
// The kernels are generic over an accessor of the sorted sample order:
//   AccRec<Rec16|Rec12> : fully sorted records (straight LSD path)
//   AccHyb   : (pos, "differs from predecessor" byte) arrays of the prefix-sort + tie-refine path
//   k_name_count  : per-chunk count of "key differs from predecessor" flags
//   (scan of the counts, total = number of distinct names)
//   k_name_assign : emits (slot(pos_i), name_i) pairs      (R[..] = name, lib.rs:93-98)
//   k_assign_unique: when every name is unique (lib.rs:109-113), SA12[i] = slot(pos_i) and the
//                    pairs (slot(pos_i), i+1)
// The pairs go through the windowed inversion (k_invperm_local) instead of a random scatter.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_neq(const Rec16 &a, const Rec16 &b) {
  return (a.k0 != b.k0) | (a.k1 != b.k1) | (a.k2 != b.k2);
}
__device__ __forceinline__ bool key_neq(const Rec12 &a, const Rec12 &b) { return (a.k0 != b.k0) | (a.k1 != b.k1); }
__device__ __forceinline__ u32 slot_of(u32 pos, u32 m0) {
  const u32 q = pos / 3, rem = pos - 3 * q;
  return rem == 1 ? q : q + m0;
}
template <class Rec>
struct AccRec {
  const Rec *s;
  __device__ __forceinline__ u32 pos(u32 i) const { return s[i].pos; }
  __device__ __forceinline__ u32 neq(u32 i) const {
    if (i == 0) return 1u;
    const Rec a = s[i], b = s[i - 1];
    return key_neq(a, b) ? 1u : 0u;
  }
  __device__ __forceinline__ u32 tail_differs() const { return 1u; }
};
struct AccHyb {
  const Rec8 *h; const uint8_t *f; u32 posmask;   // pos = low bits of h[i].val; f[i] = 1 iff key(i) != key(i-1)
  __device__ __forceinline__ u32 pos(u32 i) const { return h[i].val & posmask; }
  __device__ __forceinline__ u32 neq(u32 i) const { return f[i]; }
  __device__ __forceinline__ u32 tail_differs() const { return 1u; }
};

constexpr int kNameIPT = 4;
// a name is unique iff its key differs from both neighbours in the sorted order
// (tail_differs(): does the key BEHIND the last entry differ from it?  Always for a complete sorted array; the global
// mode, where a rank holds a range of the sorted order, answers from its right neighbour's first key: AccBound.)
template <class Acc>
__device__ __forceinline__ u32 acc_unique(const Acc &acc, u32 i, u32 n) {
  return (acc.neq(i) && (i + 1 == n ? acc.tail_differs() : acc.neq(i + 1))) ? 1u : 0u;
}
// A range [0, n) of a longer sorted order: entry 0 continues the left neighbour's last key iff first_eq, the right
// neighbour's first key equals the last entry iff last_eq_next.
template <class Acc>
struct AccBound {
  Acc a; u32 first_eq, last_eq_next;
  __device__ __forceinline__ u32 pos(u32 i) const { return a.pos(i); }
  __device__ __forceinline__ u32 neq(u32 i) const { return i == 0 ? (first_eq ? 0u : 1u) : a.neq(i); }
  __device__ __forceinline__ u32 tail_differs() const { return last_eq_next ? 0u : 1u; }
};
// flag of entry i + 1 for every lane of a wave whose lane l holds entry i = row + l and that entry's flag f (f_nextrow:
// the flags of the 64 entries behind this row when the caller has them, else pass have_next = false): the neighbour
// lane's flag where there is one, a direct comparison at the end of the row / range
template <class Acc>
__device__ __forceinline__ u32 next_flag(const Acc &acc, u32 i, u32 n, u32 end, u32 f, bool have_next, u32 f_nextrow) {
  u32 fn = __shfl_down(f, 1);
  const u32 f0 = __shfl(f_nextrow, 0);
  const bool from_lanes = lane_id() < 63u ? (i + 1 < end) : (have_next && i + 1 < end);
  if (lane_id() == 63u) fn = f0;
  if (!from_lanes) fn = (i + 1 >= n) ? (i + 1 == n ? acc.tail_differs() : 0u) : acc.neq(i + 1);
  return fn;
}
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_name_count(Acc acc, u32 n, u32 chunk, u32 *counts, u32 *uniq_total) {
  __shared__ u32 tmp[kWaves], tmpu[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0, u = 0;
  for (u32 base = begin; base < end; base += kBlock) {        // (chunk is a multiple of kBlock: whole waves)
    const u32 i = base + threadIdx.x;
    const u32 f = i < end ? acc.neq(i) : 0u;
    const u32 fn = next_flag(acc, i, n, end, f, false, 0u);
    c += f; u += (i < end) ? (f & fn) : 0u;
  }
  c = wave_reduce(c); u = wave_reduce(u);
  if (lane_id() == 0) { tmp[wave_id()] = c; tmpu[wave_id()] = u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t = 0, tu = 0;
    for (int i = 0; i < kWaves; i++) { t += tmp[i]; tu += tmpu[i]; }
    counts[blockIdx.x] = t;
    if (tu) atomicAdd(uniq_total, tu);
  }
}
// sslot (optional, discarding recursion): sslot[i] = slot(pos_i) | unique_i << 31, and the pair value
// carries the same unique bit (names < 2^31 on that path).
// A wave works 64 * kNameIPT consecutive entries as kNameIPT rows of 64 (coalesced record reads); the running name
// inside the wave comes from ballots, the waves' sums meet in LDS once per tile.
constexpr u32 kUniqBit = 0x80000000u;
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_name_assign(Acc acc, u32 n, u32 chunk, const u32 *__restrict__ base_excl,
                                                       u32 m0, Rec8 *__restrict__ pairs, u32 *__restrict__ sslot) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  constexpr u32 kTile = kBlock * kNameIPT, kWaveItems = 64 * kNameIPT;
  const u32 lane = lane_id(), w = wave_id();
  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 wb = tile + w * kWaveItems;
    u32 f[kNameIPT], pre[kNameIPT];
    u32 wsum = 0;
#pragma unroll
    for (int j = 0; j < kNameIPT; j++) {
      const u32 i = wb + j * 64 + lane;
      f[j] = (i < end) ? acc.neq(i) : 0u;
      const u64 b = __ballot(f[j] != 0u);
      pre[j] = wsum + mbcnt(b) + f[j];             // flags among the wave's entries up to and including i
      wsum += (u32)__popcll(b);
    }
    if (lane == 0) tmp[w] = wsum;
    __syncthreads();
    u32 woff = 0, tot = 0;
#pragma unroll
    for (int x = 0; x < kWaves; x++) { const u32 t = tmp[x]; if ((u32)x < w) woff += t; tot += t; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kNameIPT; j++) {
      const u32 i = wb + j * 64 + lane;
      u32 fn = 0;
      if (sslot) fn = next_flag(acc, i, n, end, f[j], j + 1 < kNameIPT, j + 1 < kNameIPT ? f[j + 1 < kNameIPT ? j + 1 : j] : 0u);
      if (i < end) {
        const u32 name = running + woff + pre[j];
        const u32 sl = slot_of(acc.pos(i), m0);
        if (sslot) {
          const u32 ub = (f[j] & fn) ? kUniqBit : 0u;
          sslot[i] = sl | ub;
          pairs[i] = Rec8{sl, name | ub};
        } else {
          pairs[i] = Rec8{sl, name};
        }
      }
    }
    running += tot;
  }
}
// Emits (slot, i+1) pairs (coalesced) for the windowed inversion below instead of scattering 4-byte
// ranks: random 4-byte stores run at ~25 G/s on MI355X (profiles/r01_membench_access_patterns.txt),
// a partition by destination window + LDS-local placement is > 2x faster.
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_assign_unique(Acc acc, u32 n, u32 m0, u32 *__restrict__ sa12,
                                                         Rec8 *__restrict__ pairs) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 sl = slot_of(acc.pos(i), m0);
    sa12[i] = sl;
    pairs[i] = Rec8{sl, i + 1};
  }
}

// ---------------------------------------------------------------------------------------------
// Discarding recursion (Dementiev/Kärkkäinen/Mehnert/Sanders' refinement of lib.rs:103-108).
// A sample whose name is unique needs no further sorting — its rank is its index in the sorted
// array — and a comparison of two suffixes of R stops at the first unique name.  So the recursive
// string only has to contain the non-unique slots and the unique slots that directly follow a
// non-unique one (they terminate the comparisons that start before them).  RU[p] = name | unique<<31.
//   k_keep_count/k_keep_write : R'[j] = name of the j-th kept slot, kept[j] = slot | unique<<31
//   (child: SA' of R')
//   k_discard_gather          : x[r] = kept[SA'[r]]            (kept slots in suffix order)
//   k_nonuniq_count/_write    : pt[t] = t-th non-unique slot of x (their final relative order)
//   k_final_count/_assign     : walk the level's sorted array; unique entries keep their place, the
//                               t-th non-unique entry receives pt[t]   -> SA12 and (slot, rank) pairs
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool keep_slot(const u32 *__restrict__ RU, u32 p) {
  return !((RU[p] & kUniqBit) && (p == 0 || (RU[p - 1] & kUniqBit)));
}
__device__ __forceinline__ void block_count_store(u32 c, u32 *tmp, u32 *counts) {
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
__global__ __launch_bounds__(kBlock) void k_keep_count(const u32 *__restrict__ RU, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  // (four independent loads per thread and round: one 4-byte load per round ran at 2 TB/s, round 5)
  for (u32 p0 = begin + threadIdx.x; p0 < end; p0 += 4 * kBlock) {
    u32 v[4], w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { const u32 p = min(p0 + (u32)j * kBlock, end - 1u); v[j] = RU[p]; w[j] = p ? RU[p - 1] : 0u; }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u32 p = p0 + (u32)j * kBlock;
      if (p < end) c += !((v[j] & kUniqBit) && (p == 0 || (w[j] & kUniqBit))) ? 1u : 0u;
    }
  }
  block_count_store(c, tmp, counts);
}
// Order-preserving selection inside a block: a tile is kBlock * kSelRows consecutive entries, wave w owns
// [tile + w * 64 * kSelRows, ...) as kSelRows rows of 64 (entry of row j, lane l: + j * 64 + l).  From the flags of the
// thread's kSelRows entries: ex[j] = selected entries of the tile before that entry, tot = selected entries of the tile
// (ballots inside the wave, one LDS exchange per tile).  tmp: kWaves words.
constexpr int kSelRows = 8;      // (8 rows: 2048 entries per pair of barriers; 4 rows ran k_keep_write at 2 TB/s, round 5)
constexpr u32 kSelTile = kBlock * kSelRows, kSelWave = 64 * kSelRows;
__device__ __forceinline__ void block_select_rows(const bool (&f)[kSelRows], u32 (&ex)[kSelRows], u32 *tmp, u32 &tot) {
  u32 wsum = 0;
#pragma unroll
  for (int j = 0; j < kSelRows; j++) {
    const u64 b = __ballot(f[j]);
    ex[j] = wsum + mbcnt(b);
    wsum += (u32)__popcll(b);
  }
  if (lane_id() == 0) tmp[wave_id()] = wsum;
  __syncthreads();
  u32 woff = 0, t = 0;
#pragma unroll
  for (int x = 0; x < kWaves; x++) { const u32 v = tmp[x]; if ((u32)x < wave_id()) woff += v; t += v; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kSelRows; j++) ex[j] += woff;
  tot = t;
}
__global__ __launch_bounds__(kBlock) void k_keep_write(const u32 *__restrict__ RU, u32 n, u32 chunk,
                                                      const u32 *__restrict__ base_excl, u32 *__restrict__ Rp,
                                                      u32 *__restrict__ kept) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kSelTile) {
    const u32 p0 = tile + wave_id() * kSelWave + lane_id();
    bool f[kSelRows];
    u32 ex[kSelRows], tot;
#pragma unroll
    for (int j = 0; j < kSelRows; j++) { const u32 p = p0 + j * 64; f[j] = (p < end) && keep_slot(RU, p); }
    block_select_rows(f, ex, tmp, tot);
#pragma unroll
    for (int j = 0; j < kSelRows; j++) {
      const u32 p = p0 + j * 64;
      if (f[j]) { const u32 v = RU[p]; Rp[running + ex[j]] = v & ~kUniqBit; kept[running + ex[j]] = p | (v & kUniqBit); }
    }
    running += tot;
  }
}
__global__ __launch_bounds__(kBlock) void k_discard_gather(const u32 *__restrict__ sap, u32 n,
                                                          const u32 *__restrict__ kept, u32 *__restrict__ x) {
  for (u32 r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) x[r] = kept[sap[r]];
}
__global__ __launch_bounds__(kBlock) void k_nonuniq_count(const u32 *__restrict__ x, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += (x[i] & kUniqBit) ? 0u : 1u;
  block_count_store(c, tmp, counts);
}
__global__ __launch_bounds__(kBlock) void k_nonuniq_write(const u32 *__restrict__ x, u32 n, u32 chunk,
                                                         const u32 *__restrict__ base_excl, u32 *__restrict__ pt) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kSelTile) {
    const u32 i0 = tile + wave_id() * kSelWave + lane_id();
    bool f[kSelRows];
    u32 v[kSelRows], ex[kSelRows], tot;
#pragma unroll
    for (int j = 0; j < kSelRows; j++) { const u32 i = i0 + j * 64; v[j] = (i < end) ? x[i] : kUniqBit; f[j] = !(v[j] & kUniqBit); }
    block_select_rows(f, ex, tmp, tot);
#pragma unroll
    for (int j = 0; j < kSelRows; j++)
      if (f[j]) pt[running + ex[j]] = v[j];
    running += tot;
  }
}
// sslot[i] = slot | unique<<31 of the i-th entry of the level's sorted array
__global__ __launch_bounds__(kBlock) void k_final_assign(const u32 *__restrict__ sslot, u32 n, u32 chunk,
                                                        const u32 *__restrict__ base_excl,
                                                        const u32 *__restrict__ pt, u32 *__restrict__ sa12,
                                                        Rec8 *__restrict__ pairs) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kSelTile) {
    const u32 i0 = tile + wave_id() * kSelWave + lane_id();
    bool nonu[kSelRows];
    u32 v[kSelRows], ex[kSelRows], tot;
#pragma unroll
    for (int j = 0; j < kSelRows; j++) { const u32 i = i0 + j * 64; v[j] = (i < end) ? sslot[i] : kUniqBit; nonu[j] = !(v[j] & kUniqBit); }
    block_select_rows(nonu, ex, tmp, tot);
#pragma unroll
    for (int j = 0; j < kSelRows; j++) {
      const u32 i = i0 + j * 64;
      if (i < end) {
        const u32 sl = nonu[j] ? pt[running + ex[j]] : (v[j] & ~kUniqBit);
        if (sa12) sa12[i] = sl;
        pairs[i] = Rec8{sl, i + 1};
      }
    }
    running += tot;
  }
}

// Partition pass of the windowed inversion (MSD order, not stable — order inside a window is
// irrelevant because k_invperm_local places by exact key).  The keys are a bijection onto [0,n), so
// the destination region of digit d of segment s is known analytically:
//   [ (s << seg_bits) + (d << shift), ... )  and holds exactly the keys that belong there;
// a tile only has to reserve space inside the region: one global atomicAdd per (tile, digit) on
// cursors[s*ndig + d].  No up-sweep, no scan: 16 B moved per pair instead of 24.
//   pass 1: shift = 22, seg_bits = 32 (one segment), ndig = ceil(n / 2^22) <= 1024
//   pass 2: shift = 14, seg_bits = 22, ndig = 256   (tiles never straddle a 2^22-pair segment)
constexpr int kPartNW = 16, kPartIPT = 8, kPartTile = kPartNW * 64 * kPartIPT;   // 8192 pairs
constexpr size_t kPartSmem = sizeof(Rec8) * kPartTile + sizeof(u32) * (2 * 1024 + 64 + kPartNW * kPartIPT);
// xcd_tps != 0 (pass 2): the tiles of segment s (xcd_tps tiles each) are worked by the blocks with blockIdx % 8 == s % 8 —
// the blocks that share an XCD under the dispatcher's round-robin placement (a speed assumption only) — so that every
// run written into a segment's windows goes through ONE L2 and the partial lines of neighbouring runs meet there
// (dc3_msd.hip.hpp explains the effect; grid = 8 * ceil(nseg / 8) * xcd_tps).
// Src: where the pairs come from — an array (PairArray), or made on the fly from a sorted order (PairsOfOrder: pair k =
// (pos_k, k + 1), which also leaves out_sa[k] = pos_k; saves writing the pairs and reading them back).
// Sources with kScan make a whole tile at once, because pair k needs a running count over the entries before it (the
// name of lib.rs:86-92, the number of non-unique entries of the discarding recursion): the count up to the tile comes
// from a table the caller scanned (one entry per tile of kPartTile pairs), the count inside the tile from ballots and one
// exchange through LDS.  PairsOfNames replaces k_name_assign + the read of its pairs, PairsOfFinal k_final_assign.
struct PairArray {
  static constexpr bool kScan = false;
  const Rec8 *p;
  __device__ __forceinline__ Rec8 load(u32 i) const { return p[i]; }
};
template <class Acc>
struct PairsOfOrder {
  static constexpr bool kScan = false;
  Acc acc; u32 skip; u32 *out_sa;
  __device__ __forceinline__ Rec8 load(u32 k) const {
    const u32 p = acc.pos(k + skip);
    if (out_sa) out_sa[k] = p;
    return Rec8{p, k + 1};
  }
};
// Flags (bit k of fm: the tile entry k * NT + tid, row k, NT = kPartNW * 64 threads) -> ex[k] = flags set among the
// tile's entries before that entry (index order).  lds: kPartNW * kPartIPT words.  Two barriers.
__device__ __forceinline__ void part_tile_scan(u32 fm, u32 (&ex)[kPartIPT], u32 *lds) {
  const u32 w = wave_id(), lane = lane_id();
#pragma unroll
  for (int k = 0; k < kPartIPT; k++) {
    const u64 b = __ballot(((fm >> k) & 1u) != 0u);
    ex[k] = mbcnt(b);
    if (lane == 0) lds[k * kPartNW + w] = (u32)__popcll(b);
  }
  __syncthreads();
  if (w == 0) {                                   // exclusive scan of the kPartIPT * kPartNW = 128 row-wave counts, two per lane
    static_assert(kPartNW * kPartIPT == 128, "two counts per lane of one wave");
    const u32 a = lds[2 * lane], b = lds[2 * lane + 1];
    const u32 inc = wave_incl_scan(a + b);
    lds[2 * lane] = inc - a - b;
    lds[2 * lane + 1] = inc - b;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPartIPT; k++) ex[k] += lds[k * kPartNW + w];
}
template <class Acc>
struct PairsOfNames {
  static constexpr bool kScan = true;
  Acc acc; u32 n, m0; const u32 *base_excl; u32 *sslot;      // sslot != nullptr: names and slots carry the unique bit
  __device__ __forceinline__ void load_tile(u32 begin, u32 nvalid, Rec8 (&r)[kPartIPT], u32 *lds) const {
    constexpr u32 NT = kPartNW * 64;
    const u32 tid = threadIdx.x;
    u32 sl[kPartIPT], ex[kPartIPT], fm = 0, um = 0;         // slots; flags and unique bits as masks over the rows
#pragma unroll
    for (int k = 0; k < kPartIPT; k++) {
      const u32 t = k * NT + tid, i = begin + min(t, nvalid - 1u);
      const u32 f = t < nvalid ? acc.neq(i) : 0u;
      sl[k] = slot_of(acc.pos(i), m0);
      fm |= f << k;
      if (sslot) {
        u32 fn = __shfl_down(f, 1);
        if (lane_id() == 63u || t + 1 >= nvalid) fn = i + 1 >= n ? acc.tail_differs() : acc.neq(i + 1);
        const u32 u = f & fn;
        um |= u << k;
        if (t < nvalid) sslot[i] = sl[k] | (u ? kUniqBit : 0u);
      }
    }
    part_tile_scan(fm, ex, lds);
    const u32 base = base_excl[begin / (u32)kPartTile];
#pragma unroll
    for (int k = 0; k < kPartIPT; k++)
      r[k] = Rec8{sl[k], (base + ex[k] + ((fm >> k) & 1u)) | (((um >> k) & 1u) ? kUniqBit : 0u)};
  }
};
struct PairsOfFinal {
  static constexpr bool kScan = true;
  const u32 *sslot, *pt, *base_excl; u32 *sa12;
  __device__ __forceinline__ void load_tile(u32 begin, u32 nvalid, Rec8 (&r)[kPartIPT], u32 *lds) const {
    constexpr u32 NT = kPartNW * 64;
    const u32 tid = threadIdx.x;
    u32 v[kPartIPT], ex[kPartIPT], fm = 0;
#pragma unroll
    for (int k = 0; k < kPartIPT; k++) {
      const u32 t = k * NT + tid;
      v[k] = sslot[begin + min(t, nvalid - 1u)];
      fm |= ((t < nvalid && !(v[k] & kUniqBit)) ? 1u : 0u) << k;
    }
    part_tile_scan(fm, ex, lds);
    const u32 base = base_excl[begin / (u32)kPartTile];
#pragma unroll
    for (int k = 0; k < kPartIPT; k++) {
      const u32 t = k * NT + tid, i = begin + min(t, nvalid - 1u);
      const u32 sl = ((fm >> k) & 1u) ? pt[base + ex[k]] : (v[k] & ~kUniqBit);
      if (sa12 && t < nvalid) sa12[i] = sl;
      r[k] = Rec8{sl, i + 1};
    }
  }
};
template <class Src>
__global__ __launch_bounds__(kPartNW * 64, 8) void k_part_msd(Src in, Rec8 *__restrict__ out, u32 n,
                                                          u32 shift, u32 seg_bits, u32 ndig,
                                                          u32 *__restrict__ cursors, u32 xcd_tps) {
  // LDS: the two digit tables first (their reads take a constant offset from the digit's own address), the pairs behind
  // them; after the scan gbase[d] = (start of the tile's run of d in the output) - (its start inside the tile), so the
  // pair at tile index q goes to gbase[d] + q.  Full tiles run without the per-pair guards (cf. k_msd_part).
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *hist = reinterpret_cast<u32 *>(smem);                               // [1024] counts -> tile-exclusive prefix
  u32 *gbase = hist + 1024;                                                // [1024]
  u32 *tmp = gbase + 1024;                                                 // [64 + kPartNW * kPartIPT]
  Rec8 *srec = reinterpret_cast<Rec8 *>(smem + sizeof(u32) * (2 * 1024 + 64 + kPartNW * kPartIPT));
  constexpr u32 NT = kPartNW * 64;
  const u32 tid = threadIdx.x;
  u32 tile = blockIdx.x;
  if (xcd_tps) {
    const u32 g = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    tile = (g + 8u * (idx / xcd_tps)) * xcd_tps + idx % xcd_tps;
    if ((u64)tile * kPartTile >= n) return;
  }
  const u32 begin = tile * (u32)kPartTile;
  const u32 nvalid = min((u32)kPartTile, n - begin);
  const u32 seg = seg_bits >= 32 ? 0u : (begin >> seg_bits);
  const u32 seg_base = seg_bits >= 32 ? 0u : (seg << seg_bits);
  hist[tid] = 0;
  __syncthreads();
  auto body = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
    Rec8 r[kPartIPT];
    u32 d[kPartIPT], rk[kPartIPT];
    // (all loads issued back to back: the index is clamped instead of guarded, a guard would put a wait behind every load)
    if constexpr (Src::kScan) in.load_tile(begin, nvalid, r, tmp + 64);
    else {
#pragma unroll
      for (int k = 0; k < kPartIPT; k++) r[k] = in.load(begin + (kFull ? (u32)(k * NT) + tid : min((u32)(k * NT) + tid, nvalid - 1u)));
    }
#pragma unroll
    for (int k = 0; k < kPartIPT; k++) {
      d[k] = ((r[k].key - seg_base) >> shift);
      if (kFull || (u32)(k * NT) + tid < nvalid) rk[k] = atomicAdd(&hist[d[k]], 1u);
    }
    __syncthreads();
    u32 cnt = 0, gb = 0;
    if (tid < ndig) {
      cnt = hist[tid];
      if (cnt) gb = seg_base + (tid << shift) + atomicAdd(&cursors[seg * ndig + tid], cnt);
    }
    u32 tot;
    const u32 ex = block_excl_scan<kPartNW>(tid < ndig ? cnt : 0u, tmp, tot);
    hist[tid] = ex;
    gbase[tid] = gb - ex;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPartIPT; k++)
      if (kFull || (u32)(k * NT) + tid < nvalid) srec[hist[d[k]] + rk[k]] = r[k];
    __syncthreads();
    if (kFull) {
#pragma unroll
      for (int k = 0; k < kPartIPT; k++) {
        const u32 q = (u32)(k * NT) + tid;
        const Rec8 x = srec[q];
        out[gbase[(x.key - seg_base) >> shift] + q] = x;
      }
    } else {
      for (u32 q = tid; q < nvalid; q += NT) {
        const Rec8 x = srec[q];
        out[gbase[(x.key - seg_base) >> shift] + q] = x;
      }
    }
  };
  if (nvalid == (u32)kPartTile) body(std::true_type{}); else body(std::false_type{});
}

// Final step of the windowed inversion.  The keys are a bijection onto [0,n), and the pairs are
// already partitioned by key >> kInvWindowBits, so pair range [w*W, (w+1)*W) holds exactly the
// destinations of window w: place them in LDS, then store the window with full coalesced lines.
constexpr int kInvWindowBits = 14;
constexpr int kInvWindow = 1 << kInvWindowBits;   // 16384 ranks = 64 KiB of LDS
__global__ __launch_bounds__(1024) void k_invperm_local(const Rec8 *__restrict__ pairs, u32 n, u32 *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32 *win = reinterpret_cast<u32 *>(smem);
  const u32 base = blockIdx.x * (u32)kInvWindow;
  const u32 cnt = min((u32)kInvWindow, n - base);
  for (u32 i = threadIdx.x; i < cnt; i += 1024) {
    const Rec8 r = pairs[base + i];
    win[r.key - base] = r.val;
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < cnt; i += 1024) out[base + i] = win[i];
}

// ---------------------------------------------------------------------------------------------
// Prefix-sort + tie-refine ordering of the sample triples (replaces the 3x radix_pass of
// lib.rs:74-76 when most keys are already distinct in an N-bit monotone image, N = 64 - pbits):
//   1. (image, pos) packed in one 64-bit word, 4 stable LSD passes over the image bits   [all samples]
//   2. elements whose image equals a neighbour's are "tied"; only those are re-sorted by the full
//      3b-bit key as 16-byte records and written back into the tied slots (same relative order)
// Result: h[i].val = position of the i-th smallest triple, f[i] = key differs from predecessor.
// ---------------------------------------------------------------------------------------------
// Monotone N-bit image of the full key (N = 64 - pbits, pbits = bits of a position; 34 bits at 1 GiB):
// X = key >> shx (its top 64 bits), hi = floor(X * mfix / 2^64) with mfix = floor(2^(64+N) / (Xmax+1)) —
// uses the whole N-bit range whatever the packing base is, so as few samples as possible collide
// (exact: the key itself fits N bits).  Any monotone map is valid for the tie-refine scheme.
// The record is the 64-bit word (hi << pbits) | pos.
// raw (Key9 only, alphabets that use most byte values): the image is the window's first nbits BITS read big-endian
// straight off the text — no code table, no multiplies.  Byte order = code order, a bit prefix of the window is a
// monotone map of it, and the zero padding behind the text reads as the smallest byte: valid for the tie refinement.
struct HiMap { u64 mfix; u32 shx, pbits, nbits, exact, raw; };
// an image range (one rank's share of the global mode's orderings), tested sh bits above the image's lowest
struct MsdSel { u64 lo, hi; u32 last, sh; };                  // keep image x iff lo <= (x >> sh) and ((x >> sh) < hi or last)
__device__ __forceinline__ bool msd_sel_keep(const MsdSel &s, u64 img) {
  const u64 v = img >> s.sh;
  return v >= s.lo && (s.last || v < s.hi);
}
__device__ __forceinline__ u64 hyb_hi(const Rec16 &r, const HiMap &hm) {
  const u64 lo = (u64)r.k0 | ((u64)r.k1 << 32);
  if (hm.exact) return lo;
  const u64 x = hm.shx ? ((lo >> hm.shx) | ((u64)r.k2 << (64 - hm.shx))) : lo;
  return __umul64hi(x, hm.mfix);
}
__device__ __forceinline__ Rec8 hyb_rec(const Rec16 &r, HiMap hm) {
  const u64 w = (hyb_hi(r, hm) << hm.pbits) | r.pos;
  return Rec8{(u32)(w >> 32), (u32)w};
}
// Order (< 0, 0, > 0) of the nsym-symbol windows of text positions p and q without building their keys: compared a
// word at a time, only a differing (or end-crossing) word is decoded (equal bytes have equal codes; past the end = 0).
__device__ __forceinline__ int window_cmp(const SymU8 &S, u32 p, u32 q, u32 nsym, const uint16_t *lds) {
#pragma unroll 1
  for (u32 k = 0; k < nsym; k += 4) {
    u32 wp, wq;
    __builtin_memcpy(&wp, S.t + p + k, 4);
    __builtin_memcpy(&wq, S.t + q + k, 4);
    if (wp == wq && k + 4 <= nsym && p + k + 4 <= S.m && q + k + 4 <= S.m) continue;
#pragma unroll
    for (u32 b = 0; b < 4; b++) {
      if (k + b < nsym) {
        const u32 cp = (p + k + b < S.m) ? (u32)lds[(wp >> (8 * b)) & 255u] : 0u;
        const u32 cq = (q + k + b < S.m) ? (u32)lds[(wq >> (8 * b)) & 255u] : 0u;
        if (cp != cq) return cp < cq ? -1 : 1;
      }
    }
  }
  return 0;
}
// Key makers: the full sort key of position p.  Key3 = the K–S triple of a level's string; Key9 = 9 bytes of the
// text (three packed byte-triples: exactly the triple key the level-1 string has at the slot of p), used by the
// level-0 whole-text shortcut.
template <class Sym>
struct Key3 {
  Sym S; u32 B;
  static constexpr bool kCodes = false;        // (no code table: stage() is empty)
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ Rec16 make(u32 p, const uint16_t *) const {
    return make_rec(S.get(p), S.get(p + 1), S.get(p + 2), B, p);
  }
  __device__ __forceinline__ Rec8 image(u32 p, const uint16_t *lds, const HiMap &hm) const { return hyb_rec(make(p, lds), hm); }
  __device__ __forceinline__ u64 image_hi(u32 p, const uint16_t *lds, const HiMap &hm) const { return hyb_hi(make(p, lds), hm); }
};
struct Key9 {
  static constexpr bool kCodes = true;
  SymU8 S; u32 B /* sigma+1 */, B3 /* B^3 */;
  u32 deep = 0;     // tie passes only: compare this many symbols instead of the window's 9 (second attempt, few repeats)
  __device__ __forceinline__ void stage(uint16_t *lds) const { S.stage(lds); }
  __device__ __forceinline__ Rec16 make(u32 p, const uint16_t *lds) const {
    u32 w[3]; __builtin_memcpy(w, S.t + p, 12);        // one unaligned 12-byte load (64 zero bytes pad the text)
    u32 q[9];
#pragma unroll
    for (int k = 0; k < 9; k++) q[k] = (p + k < S.m) ? (u32)lds[(w[k >> 2] >> (8 * (k & 3))) & 255u] : 0u;
    const u32 t0 = (q[0] * B + q[1]) * B + q[2];
    const u32 t1 = (q[3] * B + q[4]) * B + q[5];
    const u32 t2 = (q[6] * B + q[7]) * B + q[8];
    return make_rec(t0, t1, t2, B3, p);
  }
  __device__ __forceinline__ u64 raw_hi(u32 p, const HiMap &hm) const {
    u64 v; __builtin_memcpy(&v, S.t + p, 8);
    return __builtin_bswap64(v) >> (64u - hm.nbits);
  }
  __device__ __forceinline__ Rec8 image(u32 p, const uint16_t *lds, const HiMap &hm) const {
    if (hm.raw) { const u64 w = (raw_hi(p, hm) << hm.pbits) | p; return Rec8{(u32)(w >> 32), (u32)w}; }
    return hyb_rec(make(p, lds), hm);
  }
  __device__ __forceinline__ u64 image_hi(u32 p, const uint16_t *lds, const HiMap &hm) const {
    return hm.raw ? raw_hi(p, hm) : hyb_hi(make(p, lds), hm);
  }
  __device__ __forceinline__ int cmp(u32 p, u32 q, const uint16_t *lds) const { return window_cmp(S, p, q, deep ? deep : 9u, lds); }
  __host__ __device__ u32 window_syms() const { return 9; }
};
// KeyT = 3*L symbols of the text as three limbs of L symbols in base B (limb base BL = B^L < 2^32): the longer window
// a small alphabet needs before windows can be distinct (DNA, B = 5: L = 13, 39 symbols, 90 bits).  L = 3 is Key9's
// key.  Positions past the end read as the sentinel 0 (the text is followed by 64 zero bytes; 3*L <= 60).
// Its sort image is NOT taken from that key: base B = sigma + 1 spends log2(sigma + 1) bits per symbol on an alphabet of
// sigma (the sentinel never occurs inside the text), which at sigma = 4 leaves a 34-bit image 14 symbols — fewer values
// than a 1 GiB text has positions.  The image is v = the first J symbols in base sigma (digit = code - 1, past the end
// = 0: a monotone, not injective, map of the key — exactly what the tie refinement allows) scaled to the image
// width: floor(v * mfix / 2^64), sigma^J in (2^nbits, 2^63), J <= 3L, J <= kKeyTMaxImageSyms.
constexpr u32 kKeyTMaxImageSyms = 48;
struct KeyT {
  static constexpr bool kCodes = true;
  SymU8 S; u32 B, BL /* B^L */, L, sigma, J;
  u32 lg = 0;       // sigma = 2^lg (DNA: 2): v is J*lg bits put together by shifts and the image its top nbits; 0 = scale by mfix
  u32 deep = 0;     // as Key9::deep
  __device__ __forceinline__ void stage(uint16_t *lds) const { S.stage(lds); }
  __device__ __forceinline__ Rec16 make(u32 p, const uint16_t *lds) const {
    u32 limb[3];
    u32 k = 0, w = 0;
#pragma unroll 1
    for (int j = 0; j < 3; j++) {
      u32 v = 0;
#pragma unroll 1
      for (u32 i = 0; i < L; i++, k++) {
        if ((k & 3u) == 0) __builtin_memcpy(&w, S.t + p + k, 4);
        const u32 q = (p + k < S.m) ? (u32)lds[(w >> (8 * (k & 3u))) & 255u] : 0u;
        v = v * B + q;
      }
      limb[j] = v;
    }
    return make_rec(limb[0], limb[1], limb[2], BL, p);
  }
  __device__ __forceinline__ int cmp(u32 p, u32 q, const uint16_t *lds) const { return window_cmp(S, p, q, deep ? deep : 3 * L, lds); }
  __host__ __device__ u32 window_syms() const { return 3 * L; }
  __device__ __forceinline__ u64 image_hi(u32 p, const uint16_t *lds, const HiMap &hm) const {
    const u32 nw = (J + 3) / 4;
    u32 w[kKeyTMaxImageSyms / 4];
#pragma unroll
    for (u32 i = 0; i < kKeyTMaxImageSyms / 4; i++)
      if (i < nw) __builtin_memcpy(&w[i], S.t + p + 4 * i, 4); else w[i] = 0;
    u64 v = 0;
#pragma unroll
    for (u32 k = 0; k < kKeyTMaxImageSyms; k++) {
      if (k < J) {
        u32 q = (p + k < S.m) ? (u32)lds[(w[k >> 2] >> (8 * (k & 3u))) & 255u] : 0u;
        q = q ? q - 1 : 0u;
        v = lg ? (v << lg) | q : v * sigma + q;
      }
    }
    return lg ? v >> (J * lg - hm.nbits) : __umul64hi(v, hm.mfix);
  }
  __device__ __forceinline__ Rec8 image(u32 p, const uint16_t *lds, const HiMap &hm) const {
    const u64 word = (image_hi(p, lds, hm) << hm.pbits) | p;
    return Rec8{(u32)(word >> 32), (u32)word};
  }
};
// Key images of the 4 consecutive text positions p0 .. p0+3 (p0 % 4 == 0; positions >= n read the zero padding):
// the bodies of the whole-text pack kernels, shared with the partition pass that packs on the fly (k_msd_part_keys).
// Key9: 12 aligned text bytes -> 12 codes -> 10 byte-triples shared by the 4 keys.
__device__ __forceinline__ void images4(const Key9 &km, const HiMap &hm, u64, u32 p0, u32 n, const uint16_t *lcode, u64 img[4]) {
  const u32 *tw = reinterpret_cast<const u32 *>(km.S.t + p0);
  const u32 w[3] = {tw[0], tw[1], tw[2]};
  if (hm.raw) {                                          // 12 bytes big-endian, the 4 windows are its byte shifts
    const u64 hi = ((u64)__builtin_bswap32(w[0]) << 32) | __builtin_bswap32(w[1]);
    const u32 lo = __builtin_bswap32(w[2]);
    const u32 sh = 64u - hm.nbits;
    img[0] = hi >> sh;
#pragma unroll
    for (int j = 1; j < 4; j++) img[j] = ((hi << (8 * j)) | (lo >> (32 - 8 * j))) >> sh;
    return;
  }
  u32 q[12];
#pragma unroll
  for (int k = 0; k < 12; k++) q[k] = (p0 + k < n) ? (u32)lcode[(w[k >> 2] >> (8 * (k & 3))) & 255u] : 0u;
  u32 u[10];
#pragma unroll
  for (int k = 0; k < 10; k++) u[k] = (q[k] * km.B + q[k + 1]) * km.B + q[k + 2];
#pragma unroll
  for (int j = 0; j < 4; j++) img[j] = hyb_hi(make_rec(u[j], u[j + 3], u[j + 6], km.B3, 0u), hm);
}
// KeyT: J + 3 digits, the first image from scratch and the next three by rolling
// (v' = (v - d_first * sigma^(J-1)) * sigma + d_next); P1 = sigma^(J-1).
__device__ __forceinline__ void images4(const KeyT &km, const HiMap &hm, u64 P1, u32 p0, u32 n, const uint16_t *lcode, u64 img[4]) {
  const u32 J = km.J, sigma = km.sigma, nw = (J + 3 + 3) / 4;
  constexpr u32 kW = (kKeyTMaxImageSyms + 3 + 3) / 4;
  const u32 *tw = reinterpret_cast<const u32 *>(km.S.t + p0);
  if (km.lg) {
    // sigma = 2^lg: the 4*nw symbols as one big-endian number V of 4*nw*lg <= 128 bits, a word (4 symbols) at a time; the
    // image of window j is V's bits below symbol j, nbits of them.  (Digit = code - 1 clamped at 0: the zero padding
    // behind the text — absent byte or smallest symbol — reads as digit 0 without a bounds check.)
    const u32 lg = km.lg;
    unsigned __int128 V = 0;
    for (u32 i = 0; i < nw; i++) {
      const u32 w = tw[i];
      u32 g = 0;
#pragma unroll
      for (u32 b = 0; b < 4; b++) {
        const u32 cd = lcode[(w >> (8 * b)) & 255u];
        g = (g << lg) | (cd ? cd - 1 : 0u);
      }
      V = (V << (4 * lg)) | g;
    }
    const u64 mask = hm.nbits >= 64 ? ~0ull : (1ull << hm.nbits) - 1;
#pragma unroll
    for (u32 j = 0; j < 4; j++) img[j] = (u64)(V >> ((4 * nw - j) * lg - hm.nbits)) & mask;
    return;
  }
  u32 w[kW];
#pragma unroll
  for (u32 i = 0; i < kW; i++) w[i] = i < nw ? tw[i] : 0u;
  u64 v = 0;
  u32 dh[3] = {0, 0, 0}, dt0 = 0, dt1 = 0, dt2 = 0;
#pragma unroll
  for (u32 k = 0; k < kKeyTMaxImageSyms + 3; k++) {
    if (k < J + 3) {
      u32 q = (p0 + k < n) ? (u32)lcode[(w[k >> 2] >> (8 * (k & 3u))) & 255u] : 0u;
      q = q ? q - 1 : 0u;
      if (k < 3) dh[k] = q;
      if (k < J) v = v * sigma + q;
      else if (k == J) dt0 = q;
      else if (k == J + 1) dt1 = q;
      else dt2 = q;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    img[j] = __umul64hi(v, hm.mfix);
    if (j < 3) v = (v - (u64)dh[j] * P1) * sigma + (j == 0 ? dt0 : j == 1 ? dt1 : dt2);
  }
}
// Key3 over a level's names: the six symbols p0 .. p0+5 loaded once (the string has >= 8 zero words behind its end)
__device__ __forceinline__ void images4(const Key3<SymU32> &km, const HiMap &hm, u64, u32 p0, u32 n, const uint16_t *, u64 img[4]) {
  u32 q[6];
  __builtin_memcpy(q, km.S.s + p0, 24);
#pragma unroll
  for (int j = 0; j < 4; j++) img[j] = p0 + j < n ? hyb_hi(make_rec(q[j], q[j + 1], q[j + 2], km.B, 0u), hm) : 0ull;
}
// KeyBits (round 6): the sort image of a power-of-two alphabet (sigma = 2^lg: DNA lg = 2, binary 1, hex 4) read off a
// BIT-PACKED copy of the text — lg bits per symbol, digit = code - 1, most significant bit first, zero bits behind the
// end (k_pack_bits makes it: n lg / 8 bytes, a quarter of a DNA text).  The image of position p is then the stream's bits
// [p lg, p lg + nbits): one unaligned 8-byte load, a byte swap and two shifts — the same "leading bits as they lie in the
// text" that byte alphabets take straight from the text (HiMap::raw), and cheap enough to be made inside partition pass 1
// (k_msd_part_keys<KeyBits, true>).  KeyT's own image took a table look-up and a shift per symbol, 21 symbols per
// position: inside pass 1 it was measured slower twice (round 4), so a pack kernel wrote the images out (8 bytes per
// position written and read back, 3.8 of DNA's 14.4 ms at 1 GiB).  It is the same monotone map of the window (the first
// nbits / lg digits and the top bits of the next), so everything behind pass 1 — tie flags, KeyT's window compares — is
// unchanged.  Needs nbits + 7 + 3 lg <= 64.
struct KeyBits {
  static constexpr bool kCodes = false;
  const uint8_t *bits; u32 lg;
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ u64 image_hi(u32 p, const uint16_t *, const HiMap &hm) const {
    const u64 b = (u64)p * lg;
    u64 v; __builtin_memcpy(&v, bits + (b >> 3), 8);
    return (__builtin_bswap64(v) << (u32)(b & 7u)) >> (64u - hm.nbits);
  }
  __device__ __forceinline__ Rec8 image(u32 p, const uint16_t *lds, const HiMap &hm) const {
    const u64 w = (image_hi(p, lds, hm) << hm.pbits) | p;
    return Rec8{(u32)(w >> 32), (u32)w};
  }
};
__device__ __forceinline__ void images4(const KeyBits &km, const HiMap &hm, u64, u32 p0, u32, const uint16_t *, u64 img[4]) {
  const u64 b = (u64)p0 * km.lg;
  u64 v; __builtin_memcpy(&v, km.bits + (b >> 3), 8);
  const u64 V = __builtin_bswap64(v) << (u32)(b & 7u);
  const u32 sh = 64u - hm.nbits;
#pragma unroll
  for (u32 j = 0; j < 4; j++) img[j] = (V << (j * km.lg)) >> sh;      // (positions >= n read the zero bits behind the end: image 0 or a prefix of the last symbols, never kept)
}
// bits[g lg .. (g + 1) lg) = the 8 symbols 8 g .. 8 g + 7 as 8 lg bits, first symbol in the top bits; groups up to `groups`
// (>= ceil(n / 8) + 8 / lg + 1: the loads above read 8 bytes from the byte a position's bits start in, and everything they
// read behind the last symbol must be zero bits — the image is only a monotone map of the suffix order if it is)
__global__ __launch_bounds__(kBlock) void k_pack_bits(SymU8 S, u32 n, u32 lg, u32 groups, uint8_t *__restrict__ bits) {
  __shared__ uint16_t lcode[256];
  S.stage(lcode);
  for (u32 g = blockIdx.x * kBlock + threadIdx.x; g < groups; g += gridDim.x * kBlock) {
    const u64 p0 = (u64)g * 8u;
    u64 w = 0;
    if (p0 < n) __builtin_memcpy(&w, S.t + p0, 8);            // (64 zero bytes pad the text)
    u64 acc = 0;
#pragma unroll
    for (u32 k = 0; k < 8; k++) {
      const u32 cd = (p0 + k < n) ? (u32)lcode[(w >> (8 * k)) & 255u] : 0u;
      acc = (acc << lg) | (u64)(cd ? cd - 1 : 0u);
    }
    for (u32 i = 0; i < lg; i++) bits[(size_t)g * lg + i] = (uint8_t)(acc >> (8 * (lg - 1 - i)));
  }
}
// The counting pack kernel of KeyBits (the digit table of partition pass 1; the images themselves are made inside the pass):
// a digit is the top nbits - hshift (<= 10) bits of an image, so ONE unaligned 8-byte load serves PER consecutive positions
// ((PER - 1) lg + 7 + 10 <= 64 bits, the host checks) where k_pack_image_all_hist loads and swaps once per position —
// 0.99 -> 0.33 ms of DNA's 11.7 ms at 1 GiB.  The raw image of byte alphabets (HiMap::raw) is the same thing with lg = 8 and
// the text itself as the bit stream (byte-aligned: 6 * 8 + 10 <= 64, PER = 7): 0.47 -> 0.35 ms of random bytes' 10.7.
// Same table as k_pack_image_all_hist<KeyBits, 1024, false> / k_pack_image_text<1024, false>.
template <int PER>
__global__ __launch_bounds__(kBlock) void k_count_image_bits(KeyBits km, u32 n, HiMap hm, u32 chunk, u32 nchunks,
                                                            u32 *__restrict__ table, u32 hshift) {
  constexpr int NB = 1024;
  __shared__ u32 hist[kWaves][NB];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  const u32 sh = 64u - hm.nbits + hshift;
  for (u32 p0 = begin + (u32)PER * threadIdx.x; p0 < end; p0 += (u32)PER * kBlock) {
    const u64 b = (u64)p0 * km.lg;
    u64 v; __builtin_memcpy(&v, km.bits + (b >> 3), 8);
    const u64 V = __builtin_bswap64(v) << (u32)(b & 7u);
#pragma unroll
    for (u32 j = 0; j < (u32)PER; j++)
      if (p0 + j < end) atomicAdd(&myh[(u32)((V << (j * km.lg)) >> sh) & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// any other key maker: position by position
// KeyImg: the images were written out by a pack kernel, one u64 per position and nothing else (the position is the
// index) — how a key maker whose image is too dear to compute inside the partition pass (KeyT) still gets an image
// wider than the word has room for (k_msd_part_keys<KeyImg, true>: pass 1 reads 8 bytes per position as before).
struct KeyImg {
  static constexpr bool kCodes = false;
  const u64 *img;
  __device__ __forceinline__ void stage(uint16_t *) const {}
};
__device__ __forceinline__ void images4(const KeyImg &km, const HiMap &, u64, u32 p0, u32, const uint16_t *, u64 img[4]) {
  const u32x4 *s = reinterpret_cast<const u32x4 *>(km.img + p0);       // (p0 % 4 == 0; the array is padded to a multiple of 4)
  const u32x4 a = s[0], b = s[1];
  img[0] = ((u64)a.x << 32) | a.y; img[1] = ((u64)a.z << 32) | a.w;       // (Rec8 order: high half first)
  img[2] = ((u64)b.x << 32) | b.y; img[3] = ((u64)b.z << 32) | b.w;
}
template <class KM>
__device__ __forceinline__ void images4(const KM &km, const HiMap &hm, u64, u32 p0, u32 n, const uint16_t *lcode, u64 img[4]) {
#pragma unroll
  for (int j = 0; j < 4; j++) img[j] = p0 + j < n ? km.image_hi(p0 + j, lcode, hm) : 0ull;
}
// whole text with KeyT, 4 consecutive positions per thread: J + 3 digits, the first image from scratch and the next
// three by rolling (v' = (v - d_first * sigma^(J-1)) * sigma + d_next); same output, chunking and digit table as
// k_pack_image_text.  P1 = sigma^(J-1).
// kWide: 12-byte records {image, position} (k_pack_image12_all_hist's output) instead of (image << pbits) | position.
// kStore = false: count only (the records are made on the fly by the partition pass that follows, k_msd_part_keys).
// kImageOnly: the 8-byte output is the image alone (KeyImg reads it back).
template <int NB, bool kWide, bool kStore = true, bool kImageOnly = false>
__global__ __launch_bounds__(kBlock) void k_pack_image_textT(KeyT km, u32 n, HiMap hm, u64 P1, void *__restrict__ outv,
                                                            u32 chunk, u32 nchunks, u32 *__restrict__ table, u32 hshift = 0) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[kWaves][NB];
  km.stage(lcode);
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);       // chunk is a multiple of 4
  for (u32 p0 = begin + 4 * threadIdx.x; p0 < end; p0 += 4 * kBlock) {
    u64 img[4];
    images4(km, hm, P1, p0, n, lcode, img);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (p0 + j < end) atomicAdd(&myh[(u32)(img[j] >> hshift) & (NB - 1)], 1u);
    if (!kStore) continue;
    if (kWide) {
      Rec12 *out = static_cast<Rec12 *>(outv);
      if (p0 + 3 < end) {                                    // 48 contiguous, 16-byte aligned bytes (p0 % 4 == 0)
        u32x4 *o = reinterpret_cast<u32x4 *>(out + p0);
        o[0] = u32x4{(u32)img[0], (u32)(img[0] >> 32), p0, (u32)img[1]};
        o[1] = u32x4{(u32)(img[1] >> 32), p0 + 1, (u32)img[2], (u32)(img[2] >> 32)};
        o[2] = u32x4{p0 + 2, (u32)img[3], (u32)(img[3] >> 32), p0 + 3};
      } else {
        for (int j = 0; j < 4; j++) if (p0 + j < end) out[p0 + j] = Rec12{(u32)img[j], (u32)(img[j] >> 32), p0 + j};
      }
    } else {
      Rec8 *out = static_cast<Rec8 *>(outv);
      Rec8 r[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u64 word = kImageOnly ? img[j] : (img[j] << hm.pbits) | (p0 + j);
        r[j] = Rec8{(u32)(word >> 32), (u32)word};
      }
      if (p0 + 3 < end) {
        u32x4 *o = reinterpret_cast<u32x4 *>(out + p0);
        o[0] = u32x4{r[0].key, r[0].val, r[1].key, r[1].val};
        o[1] = u32x4{r[2].key, r[2].val, r[3].key, r[3].val};
      } else {
        for (int j = 0; j < 4; j++) if (p0 + j < end) out[p0 + j] = r[j];
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// (all pack kernels: hshift = which image bits the digit table counts — 0: the lowest digit, for the LSD passes;
//  nbits - d1 with NB = 1024: the top d1 bits, the bucket sizes of the MSD ordering, dc3_msd.hip.hpp)
// whole text, 4 consecutive positions per thread: 12 aligned text bytes -> 12 codes -> 10 byte-triples shared by
// the 4 keys; 32 contiguous output bytes per thread.  Blocks own the chunks of the radix sort that follows and
// also produce its first digit table (the up-sweep of pass 1 never reads the records back): table[d*nchunks + b].
template <int NB, bool kStore = true>
__global__ __launch_bounds__(kBlock) void k_pack_image_text(Key9 km, u32 n, HiMap hm, Rec8 *__restrict__ out, u32 chunk,
                                                           u32 nchunks, u32 *__restrict__ table, u32 hshift = 0) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[kWaves][NB];
  km.stage(lcode);
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);       // chunk is a multiple of 4
  for (u32 p0 = begin + 4 * threadIdx.x; p0 < end; p0 += 4 * kBlock) {
    u64 img[4];
    images4(km, hm, 0ull, p0, n, lcode, img);
    Rec8 r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const u64 word = (img[j] << hm.pbits) | (p0 + j);
      r[j] = Rec8{(u32)(word >> 32), (u32)word};
      if (p0 + j < end) atomicAdd(&myh[(u32)(img[j] >> hshift) & (NB - 1)], 1u);
    }
    if (!kStore) continue;
    if (p0 + 3 < end) {
      u32x4 *o = reinterpret_cast<u32x4 *>(out + p0);
      o[0] = u32x4{r[0].key, r[0].val, r[1].key, r[1].val};
      o[1] = u32x4{r[2].key, r[2].val, r[3].key, r[3].val};
    } else {
      for (int j = 0; j < 4; j++) if (p0 + j < end) out[p0 + j] = r[j];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// records of positions 0, stride, 2*stride, ... (stride 1 = all positions; > 1 = tie-rate predictor sample)
template <class KM>
__global__ __launch_bounds__(kBlock) void k_pack_image_pos(KM km, u32 nout, u32 stride, HiMap hm, Rec8 *out) {
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nout; i += gridDim.x * kBlock)
    out[i] = km.image(i * stride, lcode, hm);
}
// Chunked variants that also produce the digit table of the first radix pass (table[d*nchunks + block]), so the
// sort that follows starts with its down-sweep: all positions of a level, and the samples of a level.
template <class KM, int NB, bool kStore = true>
__global__ __launch_bounds__(kBlock) void k_pack_image_all_hist(KM km, u32 nrec, HiMap hm, Rec8 *__restrict__ out,
                                                               u32 chunk, u32 nchunks, u32 *__restrict__ table, u32 hshift = 0) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[kWaves][NB];
  km.stage(lcode);
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(nrec, begin + chunk);
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) {
    const Rec8 r = km.image(i, lcode, hm);
    if (kStore) out[i] = r;
    atomicAdd(&myh[(u32)(rec8_word(r) >> (hm.pbits + hshift)) & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
template <class Sym, int NB>
__global__ __launch_bounds__(kBlock) void k_pack_image_hist(Sym S, u32 m, u32 m0, u32 m02, u32 b, HiMap hm,
                                                           Rec8 *__restrict__ out, u32 chunk, u32 nchunks,
                                                           u32 *__restrict__ table, u32 hshift = 0) {
  __shared__ u32 hist[kWaves][NB];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(m02, begin + chunk);      // output indices; chunk is even
  for (u32 g = begin / 2 + threadIdx.x; 2 * g < end; g += kBlock) {
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    const Rec8 r0 = hyb_rec(make_rec(s1, s2, s3, b, i), hm);
    out[2 * g] = r0;
    atomicAdd(&myh[(u32)(rec8_word(r0) >> (hm.pbits + hshift)) & (NB - 1)], 1u);
    if (2 * g + 1 < m02) {
      const Rec8 r1 = hyb_rec(make_rec(s2, s3, s4, b, i + 1), hm);
      out[2 * g + 1] = r1;
      atomicAdd(&myh[(u32)(rec8_word(r1) >> (hm.pbits + hshift)) & (NB - 1)], 1u);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// stride > 1 samples every stride-th group (tie-rate predictor); out index = g / stride
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_pack_image(Sym S, u32 m, u32 m0, u32 m02, u32 b, HiMap hm, u32 stride,
                                                     u32 ngroups_out, Rec8 *out) {
  for (u32 go = blockIdx.x * kBlock + threadIdx.x; go < ngroups_out; go += gridDim.x * kBlock) {
    const u32 g = go * stride;
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    out[2 * go] = hyb_rec(make_rec(s1, s2, s3, b, i), hm);
    if (stride > 1 || 2 * g + 1 < m02) {
      // (in sampling mode a possibly non-existent last mod-2 sample only perturbs the estimate)
      if (2 * g + 1 < m02) out[2 * go + 1] = hyb_rec(make_rec(s2, s3, s4, b, i + 1), hm);
      else out[2 * go + 1] = Rec8{0xffffffffu, 0xffffffffu};
    }
  }
}
// all keys distinct: out_sa[k] = pos_k and the (pos_k, k+1) pairs of the rank inversion
// `skip` = 1 when record 0 is the dummy (it always sorts first)
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_emit_sorted(Acc acc, u32 n, u32 skip, u32 *__restrict__ out_sa,
                                                       Rec8 *__restrict__ pairs) {
  for (u32 k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
    const u32 p = acc.pos(k + skip);
    if (out_sa) out_sa[k] = p;
    if (pairs) pairs[k] = Rec8{p, k + 1};
  }
}
// Tie-rate predictor: number of sample records whose key image equals another sample's, by insertion into an
// open-addressing hash table (slot = image + 1, top bit = "seen again"); policy input only, so the count need not
// be order- or schedule-independent beyond what it is: exact.
__global__ __launch_bounds__(kBlock) void k_hash_ties(const Rec8 *__restrict__ a, u32 ns, u32 pbits,
                                                     unsigned long long *table, u32 mask, u32 *ties) {
  constexpr unsigned long long kSeen = 1ull << 63;
  u32 cnt = 0;
  const u32 lane = lane_id();
  for (u32 base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < ns; base += gridDim.x * kBlock) {
    const u32 i = base + lane;
    const bool valid = i < ns;
    const unsigned long long img = valid ? (rec8_word(a[i]) >> pbits) + 1ull : 0ull;
    // lanes that carry the image of the wave's first valid lane travel with it (a text of one repeated byte would
    // otherwise send the whole sample to a single table slot one atomic at a time)
    u64 todo = __ballot(valid);
    if (!todo) continue;
    u32 k = 1;                                                       // records this lane stands for
    bool active = valid;
#pragma unroll 1
    for (int round = 0; round < 4 && todo; round++) {                // a few frequent images per wave are aggregated
      const u32 fl = (u32)__ffsll((long long)todo) - 1;
      const unsigned long long img0 =
          ((unsigned long long)(u32)__shfl((u32)(img >> 32), fl) << 32) | (u32)__shfl((u32)img, fl);
      const u64 same = __ballot(valid && img == img0) & todo;
      if (valid && img == img0 && ((todo >> lane) & 1ull)) { active = lane == fl; k = (u32)__popcll(same); }
      todo &= ~same;
    }
    if (!active) continue;
    u32 hsh = (u32)((img * 0x9E3779B97F4A7C15ull) >> 40) & mask;
    for (;;) {
      const unsigned long long peek = __hip_atomic_load(&table[hsh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (peek == (img | kSeen)) { cnt += k; break; }                // already known to repeat: no atomic needed
      const unsigned long long cur = atomicCAS(&table[hsh], 0ull, img);
      if (cur == 0ull) {                                             // first of its image in the table
        if (k > 1) { atomicOr(&table[hsh], kSeen); cnt += k; }
        break;
      }
      if ((cur & ~kSeen) == img) {
        const unsigned long long old = atomicOr(&table[hsh], kSeen);
        cnt += k + ((old & kSeen) ? 0u : 1u);                        // the first occurrence is counted once, here
        break;
      }
      hsh = (hsh + 1) & mask;
    }
  }
  cnt = wave_reduce(cnt);
  if (lane == 0 && cnt) atomicAdd(ties, cnt);
}
// Not all triples distinct: keep the work.  The samples (pos % 3 != 0, incl. the dummy at pos == m) are
// filtered out of the fully sorted order — they are then in sorted sample order — together with their
// "full name" nf = number of key changes up to them, so equal keys <=> equal nf.
//   k_filter_count : per-chunk number of samples
//   k_filter_write : spos[j] = pos, snf[j] = nf of the j-th sample in sorted order
struct AccFilt {
  const u32 *spos, *snf;
  __device__ __forceinline__ u32 pos(u32 j) const { return spos[j]; }
  __device__ __forceinline__ u32 neq(u32 j) const { return (j == 0 || snf[j] != snf[j - 1]) ? 1u : 0u; }
  __device__ __forceinline__ u32 tail_differs() const { return 1u; }
};
// Which sorted records are samples of the level being named, and under which position:
//   MapSelf : the records are the level's own positions (the dummy, if any, is record 0 at pos == m)
//   MapText : the records are TEXT positions p ordered by 9-byte keys, the level is level 1 (string of the
//             byte-triple names): p = 3g+1 -> j = g, p = 3g+2 -> j = m0 + g (lib.rs:55-60), p = 3g is not in
//             the level-1 string at all.  Two level-1 positions have no text record and are synthesised as
//             the first outputs (npre of them, positions ppos[], nf = 0, 1; real records get nf >= 2):
//             level 1's own dummy (j == m1, all-zero key, when m1 % 3 == 1) and level 0's dummy (j == m0-1 when
//             n % 3 == 1: its name 1 is the smallest and unique, so it follows directly), if it is a sample.
struct MapSelf {
  static constexpr u32 npre = 0, nf_off = 0;
  __device__ __forceinline__ u32 pre_pos(u32) const { return 0; }
  __device__ __forceinline__ bool map(u32 p, u32 &j) const { j = p; return p % 3 != 0; }
};
struct MapText {
  u32 m0, npre, ppos[2];
  static constexpr u32 nf_off = 2;
  __device__ __forceinline__ u32 pre_pos(u32 k) const { return ppos[k]; }
  __device__ __forceinline__ bool map(u32 p, u32 &j) const {
    const u32 g = p / 3, r = p - 3 * g;
    if (r == 0) return false;
    j = (r == 1) ? g : m0 + g;
    return j % 3 != 0;
  }
};
template <class Acc, class Map>
__global__ __launch_bounds__(kBlock) void k_filter_count(Acc acc, Map mp, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) { u32 j; c += mp.map(acc.pos(i), j) ? 1u : 0u; }
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
// name_base[blk] = exclusive prefix of the key-change flags (the scanned k_name_count output for the same
// chunking), samp_base[blk] = exclusive prefix of the sample counts
template <class Acc, class Map>
__global__ __launch_bounds__(kBlock) void k_filter_write(Acc acc, Map mp, u32 n, u32 chunk,
                                                        const u32 *__restrict__ name_base,
                                                        const u32 *__restrict__ samp_base, u32 *__restrict__ spos,
                                                        u32 *__restrict__ snf) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 run_f = name_base[blockIdx.x] + mp.nf_off, run_s = samp_base[blockIdx.x] + mp.npre;
  if (blockIdx.x == 0 && threadIdx.x < mp.npre) { spos[threadIdx.x] = mp.pre_pos(threadIdx.x); snf[threadIdx.x] = threadIdx.x; }
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    u32 fl = 0, j = 0; bool smp = false;
    if (i < end) { fl = acc.neq(i); smp = mp.map(acc.pos(i), j); }
    u32 totf, tots;
    const u32 exf = block_excl_scan<kWaves>(fl, tmp, totf);
    const u32 exs = block_excl_scan<kWaves>(smp ? 1u : 0u, tmp, tots);
    if (smp) { spos[run_s + exs] = j; snf[run_s + exs] = run_f + exf + fl; }
    run_f += totf; run_s += tots;
  }
}
__device__ __forceinline__ bool hyb_tied(const Rec8 *h, u32 i, u32 n, u32 pbits) {
  const u64 a = rec8_word(h[i]) >> pbits;
  return (i > 0 && (rec8_word(h[i - 1]) >> pbits) == a) || (i + 1 < n && (rec8_word(h[i + 1]) >> pbits) == a);
}
__global__ __launch_bounds__(kBlock) void k_tie_count(const Rec8 *__restrict__ h, u32 n, u32 chunk, u32 pbits,
                                                     u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += hyb_tied(h, i, n, pbits) ? 1u : 0u;
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
// compacts the tied elements (order preserving): full-key record rebuilt from S, and the index of
// the slot it came from
template <class KM>
__global__ __launch_bounds__(kBlock) void k_tie_compact(KM km, const Rec8 *__restrict__ h, u32 n, u32 chunk,
                                                       u32 pbits, const u32 *__restrict__ base_excl,
                                                       Rec16 *__restrict__ sub, u32 *__restrict__ tiedidx) {
  // 4 consecutive records per thread and block scan (few ties: most threads only read)
  __shared__ u32 tmp[kWaves];
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  constexpr u32 kIPT = 4, kTile = kBlock * kIPT;
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  const u32 posmask = pbits >= 32 ? 0xffffffffu : ((1u << pbits) - 1u);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 i0 = tile + threadIdx.x * kIPT;
    // image of records i0-1 .. i0+kIPT (neighbours decide "tied")
    u64 img[kIPT + 2];
#pragma unroll
    for (int j = 0; j < (int)kIPT + 2; j++) {
      const u32 i = i0 + j - 1;
      img[j] = (i0 + j >= 1 && i < n) ? (rec8_word(h[i]) >> pbits) : ~0ull - j;   // distinct fillers
    }
    u32 fl[kIPT], local = 0;
#pragma unroll
    for (int j = 0; j < (int)kIPT; j++) {
      const u32 i = i0 + j;
      fl[j] = (i < end && (img[j + 1] == img[j] || img[j + 1] == img[j + 2])) ? 1u : 0u;
      local += fl[j];
    }
    u32 tot;
    u32 o = running + block_excl_scan<kWaves>(local, tmp, tot);
#pragma unroll
    for (int j = 0; j < (int)kIPT; j++) {
      if (fl[j]) {
        const u32 i = i0 + j;
        const u32 p = h[i].val & posmask;
        sub[o] = km.make(p, lcode);
        tiedidx[o] = i;
        o++;
      }
    }
    running += tot;
  }
}
// Tied records form groups (equal key image) that are tiny on high-entropy input (Poisson: almost all of
// size 2-3); groups of at most kTieSmallMax members are settled by one thread each (k_tie_resolve), the
// general path radix-sorts the compacted subset by the full key.
constexpr u32 kTieSmallMax = 16;
__device__ __forceinline__ bool key_less(const Rec16 &a, const Rec16 &b) {
  if (a.k2 != b.k2) return a.k2 < b.k2;
  if (a.k1 != b.k1) return a.k1 < b.k1;
  return a.k0 < b.k0;
}
// A member of a small tied group under key maker KM: the full key in registers, or — KeyT, whose key is long and whose
// ties are settled by its first few symbols past the image — just the position, compared lazily.
template <class KM>
struct TieKey {
  Rec16 r;
  __device__ __forceinline__ void load(const KM &km, u32 p, const uint16_t *lds) { r = km.make(p, lds); }
  __device__ __forceinline__ u32 pos() const { return r.pos; }
  static __device__ __forceinline__ int cmp3(const KM &, const uint16_t *, const TieKey &a, const TieKey &b) {
    return key_less(a.r, b.r) ? -1 : key_neq(a.r, b.r) ? 1 : 0;
  }
};
template <>
struct TieKey<KeyT> {
  u32 p;
  __device__ __forceinline__ void load(const KeyT &, u32 pp, const uint16_t *) { p = pp; }
  __device__ __forceinline__ u32 pos() const { return p; }
  static __device__ __forceinline__ int cmp3(const KeyT &km, const uint16_t *lds, const TieKey &a, const TieKey &b) {
    return km.cmp(a.p, b.p, lds);
  }
};
template <>
struct TieKey<Key9> {
  u32 p;
  __device__ __forceinline__ void load(const Key9 &, u32 pp, const uint16_t *) { p = pp; }
  __device__ __forceinline__ u32 pos() const { return p; }
  static __device__ __forceinline__ int cmp3(const Key9 &km, const uint16_t *lds, const TieKey &a, const TieKey &b) {
    return km.cmp(a.p, b.p, lds);
  }
};
// The common case in one pass over the sorted records: the thread that sees the start of a tied group of at
// most kTieSmallMax members rebuilds the members' full keys (one gather each), orders them (stable insertion
// sort; the LSD passes left them in position order) and rewrites their positions in place, keeping the key
// image so that concurrent neighbour tests see unchanged images.  f[] must be pre-set to 1.  A larger group
// raises *overflow and is left to the general path (k_tie_compact .. k_tie_writeback).
// words[0] = overflow flag, words[1] += tied records, words[2] += records whose full key equals the predecessor's
// (settled groups only).  emit_sa != nullptr: also write the positions in sorted order to emit_sa[i - skip]
// (complete when words[0] == 0; it is the suffix array when words[2] == 0 as well).
// Same: SameRec compares the images of neighbouring records; SameFlag reads the byte the bucket ordering's local sort left
// (the scan then reads 1 byte per record instead of 8).
struct SameRec {
  const Rec8 *h; u32 pbits;
  __device__ __forceinline__ bool operator()(u32 i) const { return i > 0 && (rec8_word(h[i]) >> pbits) == (rec8_word(h[i - 1]) >> pbits); }
};
template <class KM, class Same>
__global__ __launch_bounds__(kBlock) void k_tie_resolve(KM km, Same same, Rec8 *__restrict__ h, u32 n, u32 pbits,
                                                       uint8_t *__restrict__ f, u32 *words,
                                                       u32 *__restrict__ emit_sa, u32 skip) {
  // Group starts are sparse (a few per wave): a block collects the starts of 2048-record tiles in LDS, tile after
  // tile, and works the list in full batches of kBlock groups (one per lane), so the dependent gathers of many groups
  // are in flight together (see k_tie_resolve_split).
  constexpr u32 kIPT = 4, kTile = kBlock * kIPT, kCap = kTile / 2 + kBlock;
  __shared__ uint16_t lcode[256];
  __shared__ u32 starts[kCap];
  __shared__ u32 nstart, ntied, ndup;
  km.stage(lcode);
  const u32 posmask = pbits >= 32 ? 0xffffffffu : ((1u << pbits) - 1u);
  const u32 ntiles = (n + kTile - 1) / kTile;
  if (threadIdx.x == 0) { ntied = 0; ndup = 0; nstart = 0; }
  __syncthreads();
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    u32 tied = 0;
#pragma unroll
    for (u32 j = 0; j < kIPT; j++) {
      const u32 i = tile * kTile + j * kBlock + threadIdx.x;
      if (i < n) {
        const bool eqp = same(i);
        const bool eqn = i + 1 < n && same(i + 1);
        if (eqn && !eqp) starts[atomicAdd(&nstart, 1u)] = i;
        if (eqn || eqp) tied++;
        else if (emit_sa && i >= skip) emit_sa[i - skip] = h[i].val & posmask;
      }
    }
    tied = wave_reduce(tied);
    if (lane_id() == 0 && tied) atomicAdd(&ntied, tied);
    __syncthreads();
    const bool last = tile + gridDim.x >= ntiles;
    u32 dup = 0;
    for (;;) {
      const u32 ns = nstart;
      __syncthreads();                                         // everyone has read it before anyone appends again
      if (ns == 0 || (ns < kBlock && !last)) break;
      const u32 base = ns > kBlock ? ns - kBlock : 0;
      if (base + threadIdx.x < ns) {
        const u32 i = starts[base + threadIdx.x];
        const Rec8 h0 = h[i];
        u32 e = i + 2;
        while (e < n && e - i <= kTieSmallMax && same(e)) e++;
        const u32 len = e - i;
        const u32 lo_img = h0.val & ~posmask;
        Rec8 o; o.key = h0.key;
        if (len > kTieSmallMax) {
          words[0] = 1u;
        } else if (len == 2) {
          TieKey<KM> x, y;
          x.load(km, h0.val & posmask, lcode); y.load(km, h[i + 1].val & posmask, lcode);
          const int c3 = TieKey<KM>::cmp3(km, lcode, y, x);
          if (c3 < 0) { const TieKey<KM> t = x; x = y; y = t; }
          o.val = lo_img | x.pos(); h[i] = o;
          o.val = lo_img | y.pos(); h[i + 1] = o;
          const bool ne = c3 != 0;
          f[i + 1] = ne ? 1 : 0;
          dup += ne ? 0u : 1u;
          if (emit_sa) {
            if (i >= skip) emit_sa[i - skip] = x.pos();
            emit_sa[i + 1 - skip] = y.pos();
          }
        } else {
          TieKey<KM> loc[kTieSmallMax];
          for (u32 x = 0; x < len; x++) {
            TieKey<KM> v;
            v.load(km, h[i + x].val & posmask, lcode);
            u32 y = x;
            while (y > 0 && TieKey<KM>::cmp3(km, lcode, v, loc[y - 1]) < 0) { loc[y] = loc[y - 1]; y--; }
            loc[y] = v;
          }
          for (u32 x = 0; x < len; x++) {
            o.val = lo_img | loc[x].pos();
            h[i + x] = o;
            if (x > 0) {
              const bool ne = TieKey<KM>::cmp3(km, lcode, loc[x], loc[x - 1]) != 0;
              f[i + x] = ne ? 1 : 0;
              dup += ne ? 0u : 1u;
            }
            if (emit_sa && i + x >= skip) emit_sa[i + x - skip] = loc[x].pos();
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) nstart = base;
      __syncthreads();
    }
    if (dup) atomicAdd(&ndup, dup);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ntied) atomicAdd(&words[1], ntied);
    if (ndup) atomicAdd(&words[2], ndup);
  }
}
// Tie pass of the whole-text order when the last radix pass wrote through a SplitSink: img[i] = low 32 bits of the
// key image of the i-th record, sa[i] = its position (already the answer for every untied record).  Same scheme as
// k_tie_resolve, 4 bytes per record to read and nothing to write but the members of tied groups.  (Two neighbours
// whose images differ only above bit 31 look tied here; ordering them by the full key leaves them as they are.)
// words as for k_tie_resolve.
// Same = how the pass learns that record i has the image of record i - 1: SameImg compares the 32 image bits the LSD
// passes' SplitSink left, SameFlag reads the byte the bucket ordering's local sort left (1 byte per record instead of 4).
constexpr u32 kTieTileShift = 12, kTieTile = 1u << kTieTileShift;     // records per tile of k_tie_resolve_split (and per byte of MsdSplitSink::tilef)
struct SameImg {
  const u32 *img;
  __device__ __forceinline__ bool operator()(u32 i) const { return i > 0 && img[i] == img[i - 1]; }
  __device__ __forceinline__ bool tile_clear(u32) const { return false; }
};
struct SameFlag {
  const uint8_t *f;         // 0 / 1 per record (MsdSplitSink), 16-byte aligned, 16 readable bytes behind the last record
  const uint8_t *tilef = nullptr;   // optional: byte per kTieTile records, 0 = no record of the tile is tied or precedes a tied one
  __device__ __forceinline__ bool operator()(u32 i) const { return f[i] != 0; }
  __device__ __forceinline__ bool tile_clear(u32 tile) const { return tilef != nullptr && tilef[tile] == 0; }
};
// bit j = same(i0 + j) for j = 0 .. 16, 0 past the end (i0 a multiple of 16): what a thread of the tie pass needs to know
// about its 16 records and the one behind them.  The flag bytes come as one 16-byte load and one byte — read a byte at a
// time, twice per record, the scan of 2^30 flags was 1.3 ms of a 14 ms build for 0.006 % tied records.
template <class Same>
__device__ __forceinline__ u32 same_mask17(const Same &same, u32 i0, u32 n) {
  u32 m = 0;
#pragma unroll
  for (u32 j = 0; j <= 16; j++) m |= (i0 + j < n && same(i0 + j)) ? 1u << j : 0u;
  return m;
}
template <>
__device__ __forceinline__ u32 same_mask17<SameFlag>(const SameFlag &same, u32 i0, u32 n) {
  const uint4 w = *reinterpret_cast<const uint4 *>(same.f + i0);
  const u32 ws[4] = {w.x, w.y, w.z, w.w};
  u32 m = 0;
#pragma unroll
  for (u32 k = 0; k < 4; k++) {
    const u32 x = ws[k] & 0x01010101u;                               // bytes b0 .. b3, each 0 / 1
    m |= ((x | (x >> 7) | (x >> 14) | (x >> 21)) & 15u) << (4 * k);
  }
  m |= (same.f[i0 + 16] & 1u) << 16;
  const u32 left = n - i0;                                           // >= 1
  return left >= 17 ? m : m & ((1u << left) - 1u);
}
template <class KM, class Same>
__global__ __launch_bounds__(kBlock) void k_tie_resolve_split(KM km, Same same, u32 *__restrict__ sa,
                                                             u32 n, u32 *words) {
  // Group starts are sparse (3 % of the records on random input): a block keeps collecting them tile after tile and
  // works the list only in full batches of kBlock groups (one per lane), so that every wave has 64 dependent gathers
  // in flight instead of a handful.
  constexpr u32 kIPT = 16, kTile = kBlock * kIPT, kCap = kTile / 2 + kBlock;      // a thread scans 16 consecutive records
  static_assert(kTile == kTieTile, "the sink's tile flags (MsdSplitSink::tilef) are per tile of this pass");
  __shared__ uint16_t lcode[256];
  __shared__ u32 starts[kCap];
  __shared__ u32 nstart, ntied, ndup;
  km.stage(lcode);
  const u32 ntiles = (n + kTile - 1) / kTile;
  if (threadIdx.x == 0) { ntied = 0; ndup = 0; nstart = 0; }
  __syncthreads();
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    u32 tied = 0;
    {
      const u32 i0 = tile * kTile + threadIdx.x * kIPT;
      if (i0 < n && !same.tile_clear(tile)) {                        // (a clear tile holds no tied record and no group start)
        const u32 m = same_mask17(same, i0, n);
        const u32 eqp = m & 0xffffu, eqn = (m >> 1) & 0xffffu;       // bit j: record i0 + j equals its predecessor / successor
        u32 st = eqn & ~eqp;                                         // group starts
        tied = (u32)__popc(eqn | eqp);
        while (st) {
          const u32 j = (u32)__builtin_ctz(st);
          st &= st - 1u;
          starts[atomicAdd(&nstart, 1u)] = i0 + j;
        }
      }
    }
    tied = wave_reduce(tied);
    if (lane_id() == 0 && tied) atomicAdd(&ntied, tied);
    __syncthreads();
    const bool last = tile + gridDim.x >= ntiles;
    u32 dup = 0;
    for (;;) {
      const u32 ns = nstart;
      __syncthreads();                                         // everyone has read it before anyone appends again
      if (ns == 0 || (ns < kBlock && !last)) break;
      const u32 base = ns > kBlock ? ns - kBlock : 0;
      if (base + threadIdx.x < ns) {
        const u32 i = starts[base + threadIdx.x];
        u32 e = i + 2;
        while (e < n && e - i <= kTieSmallMax && same(e)) e++;
        const u32 len = e - i;
        if (len > kTieSmallMax) {
          words[0] = 1u;
        } else if (len == 2) {
          TieKey<KM> x, y;
          x.load(km, sa[i], lcode); y.load(km, sa[i + 1], lcode);
          const int c3 = TieKey<KM>::cmp3(km, lcode, y, x);
          if (c3 < 0) { sa[i] = y.pos(); sa[i + 1] = x.pos(); }
          dup += c3 != 0 ? 0u : 1u;
        } else {
          TieKey<KM> loc[kTieSmallMax];
          for (u32 x = 0; x < len; x++) {
            TieKey<KM> v;
            v.load(km, sa[i + x], lcode);
            u32 y = x;
            while (y > 0 && TieKey<KM>::cmp3(km, lcode, v, loc[y - 1]) < 0) { loc[y] = loc[y - 1]; y--; }
            loc[y] = v;
          }
          for (u32 x = 0; x < len; x++) {
            sa[i + x] = loc[x].pos();
            if (x > 0) dup += TieKey<KM>::cmp3(km, lcode, loc[x], loc[x - 1]) != 0 ? 0u : 1u;
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) nstart = base;
      __syncthreads();
    }
    if (dup) atomicAdd(&ndup, dup);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ntied) atomicAdd(&words[1], ntied);
    if (ndup) atomicAdd(&words[2], ndup);
  }
}
__global__ __launch_bounds__(kBlock) void k_tie_writeback(const Rec16 *__restrict__ sub, const u32 *__restrict__ tiedidx,
                                                         u32 t, Rec8 *__restrict__ h, uint8_t *__restrict__ f) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) {
    const Rec16 cur = sub[j];
    const u32 i = tiedidx[j];
    h[i].val = cur.pos;
    bool ne = true;
    if (j > 0) { const Rec16 prev = sub[j - 1]; ne = key_neq(cur, prev); }
    f[i] = ne ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Prefix sort + tie refinement on 12-byte records, for keys wider than 64 bits whose 34-bit image (the Rec8 path
// above) collides everywhere but whose 63-bit prefix does not — low-entropy text two levels down: 81-bit triples of
// 27-bit names, ~97 % of them distinct.  Record = (63-bit prefix of the key in k1:k0, pos): 7 passes of 9-bit digits
// over 12 bytes instead of 9 passes over 16.  Same scheme as above: records whose prefix equals a neighbour's are
// re-ordered by the full key (rebuilt from the level's string), small groups by one thread each, the rest by a
// full-key sort of the tied subset; f[i] = key differs from predecessor.
// ---------------------------------------------------------------------------------------------
constexpr u32 kImg12Bits = 63;
__device__ __forceinline__ u64 img12(const Rec12 &r) { return ((u64)r.k1 << 32) | r.k0; }
// top kImg12Bits bits of the kbits-bit key of r (kbits > 64)
__device__ __forceinline__ Rec12 hyb_rec12(const Rec16 &r, u32 kbits) {
  const u32 sh = kbits - kImg12Bits;                         // 2 .. 33
  const u64 hi = ((u64)r.k2 << 32) | r.k1;
  const u64 img = sh <= 32 ? ((hi << (32 - sh)) | (sh == 32 ? 0u : (r.k0 >> sh)) ) : (hi >> (sh - 32));
  return Rec12{(u32)img, (u32)(img >> 32), r.pos};
}
struct AccHyb12 {
  const Rec12 *h; const uint8_t *f;
  __device__ __forceinline__ u32 pos(u32 i) const { return h[i].pos; }
  __device__ __forceinline__ u32 neq(u32 i) const { return f[i]; }
  __device__ __forceinline__ u32 tail_differs() const { return 1u; }
};
template <class Sym, int NB>
__global__ __launch_bounds__(kBlock) void k_pack_image12_hist(Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits,
                                                             Rec12 *__restrict__ out, u32 chunk, u32 nchunks,
                                                             u32 *__restrict__ table) {
  __shared__ u32 hist[kWaves][NB];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(m02, begin + chunk);      // output indices; chunk is even
  for (u32 g = begin / 2 + threadIdx.x; 2 * g < end; g += kBlock) {
    const u32 i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    const Rec12 r0 = hyb_rec12(make_rec(s1, s2, s3, b, i), kbits);
    out[2 * g] = r0;
    atomicAdd(&myh[r0.k0 & (NB - 1)], 1u);
    if (2 * g + 1 < m02) {
      const Rec12 r1 = hyb_rec12(make_rec(s2, s3, s4, b, i + 1), kbits);
      out[2 * g + 1] = r1;
      atomicAdd(&myh[r1.k0 & (NB - 1)], 1u);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// tie-rate predictor sample: the 63-bit prefixes of every stride-th sample group as Rec8 words (image << 1 | 0), so
// that k_hash_ties (pbits = 1) counts colliding prefixes
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_pack_image12_sample(Sym S, u32 m, u32 m02, u32 b, u32 kbits, u32 stride,
                                                               u32 ngroups_out, Rec8 *out) {
  for (u32 go = blockIdx.x * kBlock + threadIdx.x; go < ngroups_out; go += gridDim.x * kBlock) {
    const u32 g = go * stride, i = 3 * g + 1;
    const u32 s1 = S.get(i), s2 = S.get(i + 1), s3 = S.get(i + 2), s4 = S.get(i + 3);
    const u64 a = img12(hyb_rec12(make_rec(s1, s2, s3, b, i), kbits)) << 1;
    out[2 * go] = Rec8{(u32)(a >> 32), (u32)a};
    const u64 c = (2 * g + 1 < m02) ? (img12(hyb_rec12(make_rec(s2, s3, s4, b, i + 1), kbits)) << 1) : (~0ull >> 2);
    out[2 * go + 1] = Rec8{(u32)(c >> 32), (u32)c};
  }
}
__device__ __forceinline__ bool hyb12_tied(const Rec12 *h, u32 i, u32 n) {
  const u64 a = img12(h[i]);
  return (i > 0 && img12(h[i - 1]) == a) || (i + 1 < n && img12(h[i + 1]) == a);
}
__global__ __launch_bounds__(kBlock) void k_tie_count12(const Rec12 *__restrict__ h, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += hyb12_tied(h, i, n) ? 1u : 0u;
  block_count_store(c, tmp, counts);
}
template <class KM>
__global__ __launch_bounds__(kBlock) void k_tie_compact12(KM km, const Rec12 *__restrict__ h, u32 n, u32 chunk,
                                                         const u32 *__restrict__ base_excl, Rec16 *__restrict__ sub,
                                                         u32 *__restrict__ tiedidx) {
  __shared__ u32 tmp[kWaves];
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    const bool fl = (i < end) && hyb12_tied(h, i, n);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(fl ? 1u : 0u, tmp, tot);
    if (fl) { sub[running + ex] = km.make(h[i].pos, lcode); tiedidx[running + ex] = i; }
    running += tot;
  }
}
// as k_tie_resolve: the thread that sees the start of a tied group of at most kTieSmallMax members orders it by the full
// key in place (positions only; the prefix stays) and sets f; larger groups raise words[0].
// words[1] += tied records, words[2] += records whose full key equals the predecessor's (settled groups only)
template <class KM>
__global__ __launch_bounds__(kBlock) void k_tie_resolve12(KM km, Rec12 *__restrict__ h, u32 n, uint8_t *__restrict__ f,
                                                         u32 *words, u32 *__restrict__ emit_sa) {
  // (group starts are collected tile after tile and worked in full batches of kBlock: see k_tie_resolve_split)
  constexpr u32 kIPT = 4, kTile = kBlock * kIPT, kCap = kTile / 2 + kBlock;
  __shared__ uint16_t lcode[256];
  __shared__ u32 starts[kCap];
  __shared__ u32 nstart, ntied, ndup;
  km.stage(lcode);
  const u32 ntiles = (n + kTile - 1) / kTile;
  if (threadIdx.x == 0) { ntied = 0; ndup = 0; nstart = 0; }
  __syncthreads();
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    u32 tied = 0;
#pragma unroll
    for (u32 j = 0; j < kIPT; j++) {
      const u32 i = tile * kTile + j * kBlock + threadIdx.x;
      if (i < n) {
        const Rec12 r = h[i];
        const u64 a = img12(r);
        const bool eqp = i > 0 && img12(h[i - 1]) == a;
        const bool eqn = i + 1 < n && img12(h[i + 1]) == a;
        if (eqn && !eqp) starts[atomicAdd(&nstart, 1u)] = i;
        if (eqn || eqp) tied++;
        else if (emit_sa) emit_sa[i] = r.pos;       // optimistic suffix array (complete when no group overflowed)
      }
    }
    tied = wave_reduce(tied);
    if (lane_id() == 0 && tied) atomicAdd(&ntied, tied);
    __syncthreads();
    const bool last = tile + gridDim.x >= ntiles;
    u32 dup = 0;
    for (;;) {
      const u32 ns = nstart;
      __syncthreads();                                         // everyone has read it before anyone appends again
      if (ns == 0 || (ns < kBlock && !last)) break;
      const u32 base = ns > kBlock ? ns - kBlock : 0;
      if (base + threadIdx.x < ns) {
        const u32 i = starts[base + threadIdx.x];
        const u64 a = img12(h[i]);
        u32 e = i + 2;
        while (e < n && e - i <= kTieSmallMax && img12(h[e]) == a) e++;
        const u32 len = e - i;
        if (len > kTieSmallMax) {
          words[0] = 1u;
        } else if (len == 2) {
          TieKey<KM> x, y;
          x.load(km, h[i].pos, lcode); y.load(km, h[i + 1].pos, lcode);
          const int c3 = TieKey<KM>::cmp3(km, lcode, y, x);
          if (c3 < 0) { const TieKey<KM> t = x; x = y; y = t; h[i].pos = x.pos(); h[i + 1].pos = y.pos(); }
          if (emit_sa) { emit_sa[i] = x.pos(); emit_sa[i + 1] = y.pos(); }
          f[i + 1] = c3 != 0 ? 1 : 0;
          dup += c3 != 0 ? 0u : 1u;
        } else {
          TieKey<KM> loc[kTieSmallMax];
          for (u32 x = 0; x < len; x++) {
            TieKey<KM> v;
            v.load(km, h[i + x].pos, lcode);
            u32 y = x;
            while (y > 0 && TieKey<KM>::cmp3(km, lcode, v, loc[y - 1]) < 0) { loc[y] = loc[y - 1]; y--; }
            loc[y] = v;
          }
          for (u32 x = 0; x < len; x++) {
            h[i + x].pos = loc[x].pos();
            if (emit_sa) emit_sa[i + x] = loc[x].pos();
            if (x > 0) {
              const bool ne = TieKey<KM>::cmp3(km, lcode, loc[x], loc[x - 1]) != 0;
              f[i + x] = ne ? 1 : 0;
              dup += ne ? 0u : 1u;
            }
          }
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) nstart = base;
      __syncthreads();
    }
    if (dup) atomicAdd(&ndup, dup);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ntied) atomicAdd(&words[1], ntied);
    if (ndup) atomicAdd(&words[2], ndup);
  }
}
__global__ __launch_bounds__(kBlock) void k_tie_writeback12(const Rec16 *__restrict__ sub, const u32 *__restrict__ tiedidx,
                                                           u32 t, Rec12 *__restrict__ h, uint8_t *__restrict__ f) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) {
    const Rec16 cur = sub[j];
    const u32 i = tiedidx[j];
    h[i].pos = cur.pos;
    bool ne = true;
    if (j > 0) { const Rec16 prev = sub[j - 1]; ne = key_neq(cur, prev); }
    f[i] = ne ? 1 : 0;
  }
}

// Whole-text (or whole-level) order on 12-byte records: image of hm.nbits <= 63 bits next to a full 32-bit position —
// for texts beyond 2^31 positions, where the 8-byte record leaves a 32-bit image that ties 39 % of even random
// positions.  Chunking and digit table as k_pack_image_all_hist; k_pack_image12_pos = tie-predictor sample (image << 1).
template <class KM, int NB>
__global__ __launch_bounds__(kBlock) void k_pack_image12_all_hist(KM km, u32 nrec, HiMap hm, Rec12 *__restrict__ out,
                                                                 u32 chunk, u32 nchunks, u32 *__restrict__ table) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[kWaves][NB];
  km.stage(lcode);
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(nrec, begin + chunk);
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) {
    const u64 img = km.image_hi(i, lcode, hm);
    out[i] = Rec12{(u32)img, (u32)(img >> 32), i};
    atomicAdd(&myh[(u32)img & (NB - 1)], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
template <class KM>
__global__ __launch_bounds__(kBlock) void k_pack_image12_pos(KM km, u32 nout, u32 stride, HiMap hm, Rec8 *out) {
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nout; i += gridDim.x * kBlock) {
    const u64 w = km.image_hi(i * stride, lcode, hm) << 1;
    out[i] = Rec8{(u32)(w >> 32), (u32)w};
  }
}

__global__ void k_base1(u32 *out_sa, u32 *out_rank) {
  if (threadIdx.x == 0) { if (out_sa) out_sa[0] = 0; if (out_rank) out_rank[0] = 1; }
}
__global__ void k_zero_tail(u32 *p, u32 from, u32 count) {
  if (threadIdx.x < count) p[from + threadIdx.x] = 0;
}

}  // namespace dc3
