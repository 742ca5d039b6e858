// dc3hip — finishing a whole-text order in which FEW windows repeat, without the recursion.
//
// The whole-text order (dc3_order.hip.hpp) leaves all n positions sorted by their W-symbol windows (W = 9, or 3L on small
// alphabets); if every window is distinct that order is the suffix array.  When only a small fraction of the positions
// share their window with another one (random data with some planted or accidental repeats — however LONG those repeats
// are), handing the order to level 1 and running the whole recursion costs more than not having tried (1 GiB DNA with
// twenty 5 kB duplications: 150 ms against 126 ms).  Instead the tied positions alone are refined by prefix doubling
// (Manber–Myers / Larsson–Sadakane restricted to the unsorted groups): the rank of suffix p + d decides between p's of one
// group, d = W, 2W, 4W, ...  The rank of an UNTIED position is its index in the sorted order, found by binary search with
// window compares (no inverse suffix array is ever built: the tied set is small); the current group of a TIED position is
// kept in a map sorted by position.  Each round sorts the still-tied records by (group, rank of p + d) and splits the
// groups; a record alone in its group is final.  Work per round is proportional to the tied records; the number of rounds
// to log2(longest repeat / W).  dc3hip.hip: doubling_finish().
#pragma once
#include "dc3_common.hip.hpp"

namespace dc3 {

constexpr u32 kDblIPT = 8, kDblTile = kBlock * kDblIPT;
constexpr u32 kNone = 0u;      // "no flagged element yet" in the max-scans below (indices are stored + 1)

// inclusive max-scan over the block of one value per thread; tmp[kWaves]
__device__ __forceinline__ u32 block_incl_max_scan(u32 v, u32 *tmp) {
  const u32 lane = lane_id(), w = wave_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const u32 t = __shfl_up(v, o); if ((int)lane >= o) v = max(v, t); }
  if (lane == 63) tmp[w] = v;
  __syncthreads();
  u32 pre = 0;
  for (u32 i = 0; i < w; i++) pre = max(pre, tmp[i]);
  __syncthreads();
  return max(v, pre);
}

// The whole-text order as its split last pass and tie pass leave it (positions in sa[], 32 image bits in img[]): flags
// f[i] = window of sa[i] differs from that of sa[i-1], so that the doubling can start from there without the records.
struct AccSplit {
  const u32 *sa; const uint8_t *f;
  __device__ __forceinline__ u32 pos(u32 i) const { return sa[i]; }
  __device__ __forceinline__ u32 neq(u32 i) const { return f[i]; }
  __device__ __forceinline__ u32 tail_differs() const { return 1u; }
};
template <class KM, class Same>
__global__ __launch_bounds__(kBlock) void k_split_flags(KM km, Same same, const u32 *__restrict__ sa, u32 n,
                                                       uint8_t *__restrict__ f) {
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    bool ne = true;
    if (same(i)) ne = km.cmp(sa[i - 1], sa[i], lcode) != 0;
    f[i] = ne ? 1 : 0;
  }
}

// tied(i): record i shares its key with a neighbour.  counts[b] = tied records of chunk b
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_dbl_count(Acc acc, u32 n, u32 chunk, u32 *counts) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) c += (!acc.neq(i) || (i + 1 < n && !acc.neq(i + 1))) ? 1u : 0u;
  c = wave_reduce(c);
  if (lane_id() == 0) tmp[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < kWaves; i++) t += tmp[i]; counts[blockIdx.x] = t; }
}
// tied records in slot order: rec[j] = {k0 = 0, k1 = group start flag, k2 = j (map index, filled later), pos}; slot[j] = i
template <class Acc>
__global__ __launch_bounds__(kBlock) void k_dbl_collect(Acc acc, u32 n, u32 chunk, const u32 *__restrict__ base_excl,
                                                       u32 *__restrict__ slot, u32 *__restrict__ pos, u32 *__restrict__ start) {
  __shared__ u32 tmp[kWaves];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    bool t = false; u32 ne = 1;
    if (i < end) { ne = acc.neq(i); t = !ne || (i + 1 < n && !acc.neq(i + 1)); }
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(t ? 1u : 0u, tmp, tot);
    if (t) { slot[running + ex] = i; pos[running + ex] = acc.pos(i); start[running + ex] = ne; }
    running += tot;
  }
}

// gid[j] = slot of the first record of j's group (start[] flags the first record of every group); ONE block
__global__ __launch_bounds__(kBlock) void k_dbl_gid(const u32 *__restrict__ slot, const u32 *__restrict__ start, u32 t,
                                                   u32 *__restrict__ gid) {
  __shared__ u32 tmp[kWaves];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = kNone;
  __syncthreads();
  for (u32 tile = 0; tile < t; tile += kDblTile) {
    const u32 j0 = tile + threadIdx.x * kDblIPT;
    u32 last = kNone;                                     // (index + 1) of the last group start at or before element
    u32 loc[kDblIPT];
#pragma unroll
    for (u32 x = 0; x < kDblIPT; x++) {
      const u32 j = j0 + x;
      if (j < t && start[j]) last = j + 1;
      loc[x] = last;
    }
    const u32 incl = block_incl_max_scan(last, tmp);
    const u32 excl_in = max(__shfl_up(incl, 1), 0u);      // previous thread's inclusive value (lane 0 fixed below)
    __shared__ u32 wave_last[kWaves];
    if (lane_id() == 63) wave_last[wave_id()] = incl;
    __syncthreads();
    u32 before = lane_id() == 0 ? (wave_id() == 0 ? kNone : wave_last[wave_id() - 1]) : excl_in;
    before = max(before, carry_s);
#pragma unroll
    for (u32 x = 0; x < kDblIPT; x++) {
      const u32 j = j0 + x;
      if (j < t) { const u32 s = max(loc[x], before); gid[j] = slot[s - 1]; }
    }
    __syncthreads();
    if (threadIdx.x == kBlock - 1) carry_s = max(incl, carry_s);
    __syncthreads();
  }
}

// map of the tied positions: (pos, j) pairs to be sorted by pos
__global__ __launch_bounds__(kBlock) void k_dbl_map_pairs(const u32 *__restrict__ pos, u32 t, Rec8 *__restrict__ out) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) out[j] = Rec8{pos[j], j};
}
// after the sort by position: map_pos[k], and for record j its map index; map_val[k] = current group of that position
__global__ __launch_bounds__(kBlock) void k_dbl_map_build(const Rec8 *__restrict__ sorted, u32 t, const u32 *__restrict__ gid,
                                                         u32 *__restrict__ map_pos, u32 *__restrict__ map_val,
                                                         u32 *__restrict__ mapidx) {
  for (u32 k = blockIdx.x * kBlock + threadIdx.x; k < t; k += gridDim.x * kBlock) {
    const Rec8 r = sorted[k];
    map_pos[k] = r.key; map_val[k] = gid[r.val]; mapidx[r.val] = k;
  }
}
// active records of round 0: {k0 = key2 (filled per round), k1 = gid, k2 = map index, pos}
__global__ __launch_bounds__(kBlock) void k_dbl_init(const u32 *__restrict__ pos, const u32 *__restrict__ gid,
                                                    const u32 *__restrict__ mapidx, u32 t, Rec16 *__restrict__ act) {
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < t; j += gridDim.x * kBlock) act[j] = Rec16{0u, gid[j], mapidx[j], pos[j]};
}

// key2 of every active record: 1 + rank (or current group) of position pos + d; 0 when the suffix ends exactly there
template <class KM, class Acc>
__global__ __launch_bounds__(kBlock) void k_dbl_key(KM km, Acc acc, u32 n, u32 d, const u32 *__restrict__ map_pos,
                                                   const u32 *__restrict__ map_val, u32 t0, Rec16 *__restrict__ act, u32 a) {
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  for (u32 j = blockIdx.x * kBlock + threadIdx.x; j < a; j += gridDim.x * kBlock) {
    const u64 q64 = (u64)act[j].pos + d;
    u32 key2 = 0;
    if (q64 < n) {
      const u32 q = (u32)q64;
      u32 lo = 0, hi = t0;                                 // tied position?  (lower bound in map_pos)
      while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (map_pos[mid] < q) lo = mid + 1; else hi = mid; }
      if (lo < t0 && map_pos[lo] == q) {
        key2 = map_val[lo] + 1;
      } else {                                             // untied: its rank is its index in the sorted order
        u32 l = 0, h = n;
        while (l < h) {
          const u32 mid = l + (h - l) / 2;
          const u32 pm = acc.pos(mid);
          const int c = pm == q ? 0 : km.cmp(pm, q, lcode);
          if (c < 0) l = mid + 1; else h = mid;
        }
        key2 = l + 1;
      }
    }
    act[j].k0 = key2;
  }
}

// After the sort of the active records by (gid, key2): split the groups.  One block per tile of kDblTile records, in two
// passes around a one-block scan of the tiles' summaries (k_dbl_regroup<false> -> k_dbl_regroup_scan -> k_dbl_regroup<true>):
//   new slot of record j   = gid + (j - first j of its old group)
//   new group of record j  = gid + (first j of its (gid, key2) run - first j of its old group)
// kApply = false: sums[3 * tile + {0, 1, 2}] = (index + 1 of the tile's last group start, of its last run start, number of
//   records it keeps).  kApply = true, with carry[3 * tile + ..] = the same for everything before the tile (max, max, sum):
//   writes out_sa[new slot] = pos, map_val[map index] = new group; records alone in their run are final and dropped, the
//   others are compacted into next[].
template <bool kApply>
__global__ __launch_bounds__(kBlock) void k_dbl_regroup(const Rec16 *__restrict__ act, u32 a, u32 *__restrict__ sums,
                                                       const u32 *__restrict__ carry, u32 *__restrict__ out_sa,
                                                       u32 *__restrict__ map_val, Rec16 *__restrict__ next) {
  __shared__ u32 tmp[kWaves];
  __shared__ u32 wave_last[kWaves];
  const u32 tile = blockIdx.x * kDblTile;
  const u32 j0 = tile + threadIdx.x * kDblIPT;
  u32 lg = kNone, lr = kNone, locg[kDblIPT], locr[kDblIPT];
  Rec16 rec[kDblIPT];
  u32 prev_gid = 0, prev_key = 0;
  if (j0 >= 1 && j0 <= a) { const Rec16 p = act[j0 - 1]; prev_gid = p.k1; prev_key = p.k0; }
#pragma unroll
  for (u32 x = 0; x < kDblIPT; x++) {
    const u32 j = j0 + x;
    if (j < a) {
      rec[x] = act[j];
      const bool fg = j == 0 || rec[x].k1 != prev_gid;
      const bool fr = fg || rec[x].k0 != prev_key;
      if (fg) lg = j + 1;
      if (fr) lr = j + 1;
      prev_gid = rec[x].k1; prev_key = rec[x].k0;
    }
    locg[x] = lg; locr[x] = lr;
  }
  // which records stay (their run has a second member)?
  u32 keep = 0, nkeep = 0;
#pragma unroll
  for (u32 x = 0; x < kDblIPT; x++) {
    const u32 j = j0 + x;
    if (j < a) {
      const bool is_start = locr[x] == j + 1;
      bool next_start = true;                              // does a new run start right behind j?
      if (j + 1 < a) {
        const Rec16 nx = (x + 1 < kDblIPT) ? rec[x + 1 < kDblIPT ? x + 1 : x] : act[j + 1];
        next_start = nx.k1 != rec[x].k1 || nx.k0 != rec[x].k0;
      }
      if (!(is_start && next_start)) { keep |= 1u << x; nkeep++; }
    }
  }
  // block max-scans of the threads' last flagged indices, block sum-scan of the kept counts
  u32 incl = block_incl_max_scan(lg, tmp);
  u32 up = __shfl_up(incl, 1);                             // (executed by every lane: a shuffle inside the ?: below would
  if (lane_id() == 63) wave_last[wave_id()] = incl;        //  read from lanes that do not take part in it)
  __syncthreads();
  u32 before_g = lane_id() == 0 ? (wave_id() == 0 ? kNone : wave_last[wave_id() - 1]) : up;
  const u32 tile_g = incl;                                 // (of thread kBlock - 1: the tile's last group start)
  __syncthreads();
  incl = block_incl_max_scan(lr, tmp);
  up = __shfl_up(incl, 1);
  if (lane_id() == 63) wave_last[wave_id()] = incl;
  __syncthreads();
  u32 before_r = lane_id() == 0 ? (wave_id() == 0 ? kNone : wave_last[wave_id() - 1]) : up;
  const u32 tile_r = incl;
  __syncthreads();
  u32 tot;
  const u32 ex = block_excl_scan<kWaves>(nkeep, tmp, tot);
  if (!kApply) {
    if (threadIdx.x == kBlock - 1) { sums[3 * blockIdx.x] = tile_g; sums[3 * blockIdx.x + 1] = tile_r; sums[3 * blockIdx.x + 2] = tot; }
    return;
  }
  before_g = max(before_g, carry[3 * blockIdx.x]);
  before_r = max(before_r, carry[3 * blockIdx.x + 1]);
  u32 o = carry[3 * blockIdx.x + 2] + ex;
#pragma unroll
  for (u32 x = 0; x < kDblIPT; x++) {
    const u32 j = j0 + x;
    if (j < a) {
      const u32 g0 = max(locg[x], before_g) - 1, r0 = max(locr[x], before_r) - 1;
      const u32 gid = rec[x].k1;
      const u32 new_slot = gid + (j - g0), new_gid = gid + (r0 - g0);
      out_sa[new_slot] = rec[x].pos;
      map_val[rec[x].k2] = new_gid;
      if (keep & (1u << x)) next[o++] = Rec16{0u, new_gid, rec[x].k2, rec[x].pos};
    }
  }
}
// exclusive scan of the tile summaries (max, max, sum); *next_count = records kept in total.  ONE block.
__global__ __launch_bounds__(kBlock) void k_dbl_regroup_scan(const u32 *__restrict__ sums, u32 ntiles, u32 *__restrict__ carry,
                                                            u32 *next_count) {
  __shared__ u32 tmp[kWaves];
  __shared__ u32 wave_last[kWaves];
  __shared__ u32 cg, cr, cs;
  if (threadIdx.x == 0) { cg = kNone; cr = kNone; cs = 0; }
  __syncthreads();
  for (u32 base = 0; base < ntiles; base += kBlock) {
    const u32 b = base + threadIdx.x;
    const u32 g = b < ntiles ? sums[3 * b] : kNone, r = b < ntiles ? sums[3 * b + 1] : kNone, k = b < ntiles ? sums[3 * b + 2] : 0u;
    u32 incl = block_incl_max_scan(g, tmp);
    u32 up = __shfl_up(incl, 1);
    if (lane_id() == 63) wave_last[wave_id()] = incl;
    __syncthreads();
    const u32 eg = max(lane_id() == 0 ? (wave_id() == 0 ? kNone : wave_last[wave_id() - 1]) : up, cg);
    const u32 tg = max(incl, cg);
    __syncthreads();
    incl = block_incl_max_scan(r, tmp);
    up = __shfl_up(incl, 1);
    if (lane_id() == 63) wave_last[wave_id()] = incl;
    __syncthreads();
    const u32 er = max(lane_id() == 0 ? (wave_id() == 0 ? kNone : wave_last[wave_id() - 1]) : up, cr);
    const u32 tr = max(incl, cr);
    __syncthreads();
    u32 tot;
    const u32 es = block_excl_scan<kWaves>(k, tmp, tot) + cs;
    if (b < ntiles) { carry[3 * b] = eg; carry[3 * b + 1] = er; carry[3 * b + 2] = es; }
    __syncthreads();
    if (threadIdx.x == kBlock - 1) { cg = tg; cr = tr; }
    if (threadIdx.x == 0) cs += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *next_count = cs;
}

}  // namespace dc3
