// dc3_global.hip.hpp — device kernels of the GLOBAL multi-GPU mode (one suffix array over all ranks; host side in
// dc3_global_host.hpp, design in DESIGN.md §6).
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); namespace dc3.
//
// Every rank holds the level's string S and, after the rank exchange, the sample ranks rank12 REPLICATED in its own
// HBM (HBM moves 8 TB/s, an xGMI link 0.15: recomputing a key from local memory is cheaper than shipping it).  A rank
// therefore never receives records: it STREAMS over all positions of the level, keeps the ones whose key falls into
// its own key range (order-preserving selection, k_sel_count / k_sel_write with a selector functor) and sorts / merges
// only those.  What crosses xGMI is the rank exchange: (slot, name) and (position, rank) pairs routed to the owner of
// the destination block, and the all-gather of the finished 4-byte blocks.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Order-preserving selection over item indices [0, nitems): sel.pick(i, lds, out) decides whether item i is kept and
// produces its record.  Chunked like the naming kernels: count per chunk -> scan -> write.
// ---------------------------------------------------------------------------------------------
template <class Sel>
__global__ __launch_bounds__(kBlock) void k_sel_count(Sel sel, u32 nitems, u32 chunk, u32 *__restrict__ counts) {
  __shared__ u32 tmp[kWaves];
  __shared__ uint16_t lcode[256];
  sel.stage(lcode);
  const u32 begin = blockIdx.x * chunk, end = min(nitems, begin + chunk);
  u32 c = 0;
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) { typename Sel::Out o; c += sel.pick(i, lcode, o) ? 1u : 0u; }
  block_count_store(c, tmp, counts);
}
template <class Sel>
__global__ __launch_bounds__(kBlock) void k_sel_write(Sel sel, u32 nitems, u32 chunk, const u32 *__restrict__ base_excl,
                                                     typename Sel::Out *__restrict__ out) {
  __shared__ u32 tmp[kWaves];
  __shared__ uint16_t lcode[256];
  sel.stage(lcode);
  const u32 begin = blockIdx.x * chunk, end = min(nitems, begin + chunk);
  u32 running = base_excl[blockIdx.x];
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    typename Sel::Out o;
    const bool f = (i < end) && sel.pick(i, lcode, o);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(f ? 1u : 0u, tmp, tot);
    if (f) out[running + ex] = o;
    running += tot;
  }
}

// One-evaluation form: block b owns chunk b and writes the records it keeps, compacted, at the START of the chunk's
// own region of a staging array (stage[b*chunk ..)); counts[b] = how many.  After the scan of the counts k_sel_copy
// moves the segments next to each other.  pick() runs once per item instead of twice; the extra traffic is one
// read + write of the kept records only.
template <class Sel>
__global__ __launch_bounds__(kBlock) void k_sel_stage(Sel sel, u32 nitems, u32 chunk, typename Sel::Out *__restrict__ stage,
                                                     u32 *__restrict__ counts) {
  __shared__ u32 tmp[kWaves];
  __shared__ uint16_t lcode[256];
  sel.stage(lcode);
  const u32 begin = blockIdx.x * chunk, end = min(nitems, begin + chunk);
  u32 running = 0;
  for (u32 tile = begin; tile < end; tile += kBlock) {
    const u32 i = tile + threadIdx.x;
    typename Sel::Out o;
    const bool f = (i < end) && sel.pick(i, lcode, o);
    u32 tot;
    const u32 ex = block_excl_scan<kWaves>(f ? 1u : 0u, tmp, tot);
    if (f) stage[(size_t)begin + running + ex] = o;
    running += tot;
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = running;
}
template <class Out>
__global__ __launch_bounds__(kBlock) void k_sel_copy(const Out *__restrict__ stage, u32 chunk, u32 nchunks,
                                                    const u32 *__restrict__ base_excl, u32 total, Out *__restrict__ out) {
  const u32 b = blockIdx.x;
  const u32 lo = base_excl[b], hi = (b + 1 < nchunks) ? base_excl[b + 1] : total;
  const Out *src = stage + (size_t)b * chunk;
  for (u32 j = threadIdx.x; j < hi - lo; j += kBlock) out[lo + j] = src[j];
}

// 96-bit key order of Rec16 (k2 most significant)
__device__ __forceinline__ bool key_lt(const Rec16 &a, const Rec16 &b) {
  if (a.k2 != b.k2) return a.k2 < b.k2;
  if (a.k1 != b.k1) return a.k1 < b.k1;
  return a.k0 < b.k0;
}

// --- selectors --------------------------------------------------------------------------------
// whole-level order: item p = position of the level (text position for Key9); record = (image << pbits) | p;
// kept iff lo <= image (< hi unless last)
template <class KM>
struct SelPosImage {
  typedef Rec8 Out;
  KM km; HiMap hm; u64 lo, hi; u32 last;
  __device__ __forceinline__ void stage(uint16_t *lds) const { km.stage(lds); }
  __device__ __forceinline__ bool pick(u32 p, const uint16_t *lds, Rec8 &o) const {
    o = km.image(p, lds, hm);
    const u64 img = rec8_word(o) >> hm.pbits;
    return img >= lo && (last || img < hi);
  }
};
// the selection a selecting partition pass (k_msd_part_keys<.., kSel>) makes, as plain words in position order: the image
// comes from the map `hm` that pass uses (sel.sh bits wider than a word has room for), the word keeps its top bits
template <class KM>
struct SelPosImageW {
  typedef Rec8 Out;
  KM km; HiMap hm; MsdSel sel; u32 pbits;                // pbits: position bits of the word (= hm.pbits + sel.sh)
  __device__ __forceinline__ void stage(uint16_t *lds) const { km.stage(lds); }
  __device__ __forceinline__ bool pick(u32 p, const uint16_t *lds, Rec8 &o) const {
    const u64 img = km.image_hi(p, lds, hm);
    const u64 w = ((img >> sel.sh) << pbits) | p;
    o = Rec8{(u32)(w >> 32), (u32)w};
    return msd_sel_keep(sel, img);
  }
};
// the same on 12-byte records (image of hm.nbits <= 63 bits beside a full 32-bit position): texts beyond 2^31 positions
template <class KM>
struct SelPosImage12 {
  typedef Rec12 Out;
  KM km; HiMap hm; u64 lo, hi; u32 last;
  __device__ __forceinline__ void stage(uint16_t *lds) const { km.stage(lds); }
  __device__ __forceinline__ bool pick(u32 p, const uint16_t *lds, Rec12 &o) const {
    const u64 img = km.image_hi(p, lds, hm);
    o = Rec12{(u32)img, (u32)(img >> 32), p};
    return img >= lo && (last || img < hi);
  }
};
// prefix-sort naming: item q = q-th sample position; record = (image << pbits) | pos; kept by image range
template <class Sym>
struct SelSampleImage {
  typedef Rec8 Out;
  Sym S; u32 B; HiMap hm; u64 lo, hi; u32 last;
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ bool pick(u32 q, const uint16_t *, Rec8 &o) const {
    const u32 g = q >> 1, i = 3 * g + 1 + (q & 1);
    o = hyb_rec(make_rec(S.get(i), S.get(i + 1), S.get(i + 2), B, i), hm);
    const u64 img = rec8_word(o) >> hm.pbits;
    return img >= lo && (last || img < hi);
  }
};
// (key, position) order: equal keys are split between ranks by position, so that a level with few distinct keys
// (one repeated byte, short periods) is still sorted by all ranks; naming then looks at the neighbours' boundary keys
__device__ __forceinline__ bool keypos_lt(const Rec16 &a, const Rec16 &b) {
  if (a.k2 != b.k2) return a.k2 < b.k2;
  if (a.k1 != b.k1) return a.k1 < b.k1;
  if (a.k0 != b.k0) return a.k0 < b.k0;
  return a.pos < b.pos;
}
// sorted naming of a level: item q = q-th sample position in ascending order (3g+1, 3g+2, ...); record = full-key Rec16;
// kept iff klo <= (key, pos) (< khi unless last).  has_lo = 0 for rank 0.
template <class Sym>
struct SelTripleKey {
  typedef Rec16 Out;
  Sym S; u32 B; Rec16 klo, khi; u32 has_lo, last;
  u32 W = 0, sb = 0;                              // W > 3: the records are W-symbol windows (sample_rec)
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ bool pick(u32 q, const uint16_t *, Rec16 &o) const {
    const u32 g = q >> 1, i = 3 * g + 1 + (q & 1);
    o = sample_rec(S, i, B, W, sb);
    return (!has_lo || !keypos_lt(o, klo)) && (last || keypos_lt(o, khi));
  }
};
// merge, sample side: item = slot; kept iff lo <= rank12[slot] < hi; record = (rank - lo, slot): a bijection onto
// [0, hi - lo), so the windowed inversion puts the slots into rank order without a sort
struct SelRankRange {
  typedef Rec8 Out;
  const u32 *rank12; u32 lo, hi;
  __device__ __forceinline__ void stage(uint16_t *) const {}
  __device__ __forceinline__ bool pick(u32 slot, const uint16_t *, Rec8 &o) const {
    const u32 r = rank12[slot];
    o.key = r - lo; o.val = slot;
    return r >= lo && r < hi;
  }
};
// merge, mod-0 side: item g = mod-0 position 3g; record = its merge tuple (lib.rs:118-125 + the symbols/ranks that
// leq2/leq3 read, lib.rs:151-162), all from coalesced reads of the replicated S and rank12.  Owner = number of
// splitter samples that sort before it (the splitters are sample suffixes in ascending rank order).
constexpr int kMaxRanks = 16;
struct Splitters { Tup12 a[kMaxRanks]; u32 n; };     // n = P - 1 splitter tuples
template <class Sym>
struct SelMod0 {
  typedef Tup0 Out;
  Sym S; const u32 *rank12; u32 m, m0; u32 me; Splitters sp;
  __device__ __forceinline__ void stage(uint16_t *lds) const { S.stage(lds); }
  __device__ __forceinline__ bool pick(u32 g, const uint16_t *lds, Tup0 &z) const {
    const u32 j = 3 * g;
    u32 q[4]; S.get4(j, lds, q);
    z.pos = j; z.c0 = q[0]; z.c1 = q[1];
    z.r1 = rank12[g];                               // suffix j+1 (the dummy's rank when j+1 == m)
    z.r2 = (j + 2 < m) ? rank12[m0 + g] : 0u;       // suffix j+2
    u32 owner = 0;
    for (u32 h = 0; h < sp.n; h++) owner += sample_before(sp.a[h], z) ? 1u : 0u;
    return owner == me;
  }
};

// mod-0 tuples as a record type of the radix sort with the 64-bit key ((c0-1) << 32) | r1  (lib.rs:126 sorts by c0 only
// because its input is already in r1 order; a rank's selection is not)
struct Tup0G { u32 pos, c0, c1, r1, r2; };
__device__ __forceinline__ u32 digit_of(const Tup0G &r, KeyDig d) {
  const u64 k = ((u64)(r.c0 - 1u) << 32) | r.r1;
  return (u32)(k >> d.shift) & d.mask;
}

// counts[i] += v (names of this rank continue after the names of the ranks with smaller keys)
__global__ __launch_bounds__(kBlock) void k_add_scalar(u32 *p, u32 n, u32 v) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] += v;
}
// records of the positions [off, off + len) of the level (the block this rank packs before routing them by key range)
template <class KM>
__global__ __launch_bounds__(kBlock) void k_pack_image_range(KM km, u32 off, u32 len, HiMap hm, Rec8 *out) {
  __shared__ uint16_t lcode[256];
  km.stage(lcode);
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < len; i += gridDim.x * kBlock) out[i] = km.image(off + i, lcode, hm);
}
// keys of received pairs relative to the start of this rank's destination block
__global__ __launch_bounds__(kBlock) void k_rebase_keys(Rec8 *p, u32 n, u32 base) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i].key -= base;
}
// out[h] = slot whose sample rank is target[h] (the splitter samples of the merge); rank12 is a bijection onto 1..m02
struct RankTargets { u32 r[kMaxRanks]; u32 n; };
__global__ __launch_bounds__(kBlock) void k_find_ranks(const u32 *__restrict__ rank12, u32 m02, RankTargets t,
                                                      u32 *__restrict__ out) {
  for (u32 s = blockIdx.x * kBlock + threadIdx.x; s < m02; s += gridDim.x * kBlock) {
    const u32 r = rank12[s];
    for (u32 h = 0; h < t.n; h++) if (r == t.r[h]) out[h] = s;
  }
}
// full keys of every stride-th sample position (splitter candidates of the sorted naming)
template <class Sym>
__global__ __launch_bounds__(kBlock) void k_sample_triple_keys(Sym S, u32 B, u32 ns, u32 stride, Rec16 *out, u32 W, u32 sb) {
  for (u32 k = blockIdx.x * kBlock + threadIdx.x; k < ns; k += gridDim.x * kBlock) {
    const u32 q = k * stride, g = q >> 1, i = 3 * g + 1 + (q & 1);
    out[k] = sample_rec(S, i, B, W, sb);
  }
}
// pairs[k].val += add  (local ranks -> global ranks)
__global__ __launch_bounds__(kBlock) void k_add_val(Rec8 *p, u32 n, u32 add) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i].val += add;
}
// (pos, k + 1 + base) pairs of a suffix array slice: rank[sa[k]] = base + k + 1
__global__ __launch_bounds__(kBlock) void k_sa_to_pairs(const u32 *__restrict__ sa, u32 n, u32 base, Rec8 *__restrict__ pairs) {
  for (u32 k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) pairs[k] = Rec8{sa[k], base + k + 1};
}
// k_checksum with global indices: sum over k of mix(first + k, sa[k])
__global__ __launch_bounds__(kBlock) void k_checksum_off(const u32 *__restrict__ sa, u32 n, u64 first, u64 *out) {
  u64 acc = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    acc += splitmix64(((first + i) << 32) | sa[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}
__global__ __launch_bounds__(kBlock) void k_widen_off(const u32 *__restrict__ in, int64_t *__restrict__ out, u32 n) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = (int64_t)in[i];
}

}  // namespace dc3
