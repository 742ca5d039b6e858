// dc3_radix.hip.hpp — stable LSD radix passes (up-sweep / row scan / down-sweep with loaders).
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Stable LSD radix pass (lib.rs:15-39 with the K+1 counters replaced by digits of NB = 256 or 512
// bins; 9-bit digits are used where they save a pass, e.g. 25-27-bit symbols).
//   up-sweep  : per-chunk digit histogram (lib.rs:20-22)          -> table[digit][chunk]
//   scan      : exclusive prefix sums over table (lib.rs:25-32)   (k_scan_rows + k_scan_excl_inplace)
//   down-sweep: stable scatter (lib.rs:35-38)
// A digit is (key >> shift) & mask of the record's sort key.
// ---------------------------------------------------------------------------------------------
struct KeyDig { u32 shift, mask; };
// Rec8 as a 64-bit sort word: key = high half, val = low half (prefix-sort records keep the position
// in the low pbits of val, the rest is the monotone key image)
__device__ __forceinline__ u64 rec8_word(const Rec8 &r) { return ((u64)r.key << 32) | r.val; }
__device__ __forceinline__ u32 digit_of(const Rec8 &r, KeyDig d) { return (u32)(rec8_word(r) >> d.shift) & d.mask; }
__device__ __forceinline__ u32 digit_of(const Rec12 &r, KeyDig d) {
  const u64 k = (u64)r.k0 | ((u64)r.k1 << 32);
  return (u32)(k >> d.shift) & d.mask;
}
__device__ __forceinline__ u32 digit_of(const Rec16 &r, KeyDig d) {     // 96-bit key, shift < 96
  const u32 w = d.shift >> 5, off = d.shift & 31;
  const u32 a = w == 0 ? r.k0 : (w == 1 ? r.k1 : r.k2);
  const u32 b = w == 0 ? r.k1 : (w == 1 ? r.k2 : 0u);
  return (off ? ((a >> off) | (b << (32 - off))) : a) & d.mask;
}
// mod-0 positions are real symbols (c0 >= 1), so the key is c0-1 in [0, K)
__device__ __forceinline__ u32 digit_of(const Tup0 &r, KeyDig d) { return ((r.c0 - 1u) >> d.shift) & d.mask; }

template <class Rec, int NB>
__global__ __launch_bounds__(kBlock) void k_rs_upsweep(const Rec *__restrict__ in, u32 n, u32 chunk, u32 nchunks,
                                                      KeyDig dig, u32 *__restrict__ table) {
  __shared__ u32 hist[kWaves][NB];
  const u32 tid = threadIdx.x;
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = tid; j < NB; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  const u32 begin = blockIdx.x * chunk;
  const u32 end = min(n, begin + chunk);
  u32 *myh = hist[wave_id()];
  for (u32 i = begin + tid; i < end; i += kBlock) {
    const Rec r = in[i];
    atomicAdd(&myh[digit_of(r, dig)], 1u);
  }
  __syncthreads();
  for (int j = tid; j < NB; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}

template <class Rec, int IPT, int NW, int NB>
struct DownsweepSmem {
  static constexpr int kTile = NW * 64 * IPT;
  static constexpr size_t kBytes = sizeof(Rec) * kTile + sizeof(u32) * (NW * NB + NB + NB + 32);
};

// table rows are scanned per digit (k_scan_rows) and the 256 digit totals separately
// (k_scan_excl_inplace on digit_base), so the global base of (digit d, chunk c) is
// digit_base[d] + table[d*nchunks + c].
// NW waves per block (measured on MI355X, profiles/r01_radix_downsweep_variants.txt: 16 waves x 8
// items = 8192-record tiles move 3.5 TB/s vs 2.9 TB/s for 4 waves x 16 items).
// Loaders: where a down-sweep takes its records from.  ArrayLoader = a plain record array.  A loader
// may also DROP elements (load() returns false), which fuses an order-preserving selection into the
// pass (used for Step 2: mod-0 tuples are made from the mod-1 entries of the sorted sample tuples
// and immediately partitioned by their first key byte — lib.rs:118-126 in one pass).
template <class Rec>
struct ArrayLoader {
  const Rec *p;
  __device__ __forceinline__ bool load(u32 i, Rec &r) const { r = p[i]; return true; }
};

// Sinks: where a down-sweep puts record x of global output index g.  RecSink = the record array of the next pass.
// SplitSink (last pass of the whole-text order, 8-byte words (image << pbits) | pos): the position goes straight to
// the suffix-array buffer and the low 32 image bits to a side array — the same 8 bytes, but the tie pass then
// reads 4 bytes per record and leaves the positions of untied records alone.
template <class Rec>
struct RecSink {
  Rec *p;
  __device__ __forceinline__ void store(u32 g, const Rec &x) const { p[g] = x; }
};
struct SplitSink {
  u32 *sa, *img; u32 pbits;
  uint8_t *same = nullptr;     // (not written by the LSD passes: the bucket ordering's local sort leaves "same image as the
                               //  record before" bytes here instead of the image array, see MsdSplitSink)
  uint8_t *tilef = nullptr;    // (optional, zeroed by the host: byte t becomes 1 when a record of tie-pass tile t is tied, see MsdSplitSink)
  __device__ __forceinline__ void store(u32 g, const Rec8 &x) const {
    const u64 w = rec8_word(x);
    sa[g] = (u32)(w & ((1ull << pbits) - 1ull));
    img[g] = (u32)(w >> pbits);
  }
};

// Sinks that read at the destination instead of storing (SinkReads<S>::value): fetch(g) issues the load, check(g, x, v)
// judges it — see the write-out loop of k_rs_downsweep.
template <class S> struct SinkReads { static constexpr bool value = false; };

// How a record is held in registers between its load and the LDS reorder.  hipcc left the array of 20-byte Tup0 structs
// in scratch (112 bytes per thread stored and reloaded, profiles/r05a occupancy report) although every index is a
// constant after unrolling; the same words as one 5-wide vector per record stay in VGPRs.  Other record types as they are.
template <class Rec> struct RegRec {
  typedef Rec T;
  static __device__ __forceinline__ T pack(const Rec &r) { return r; }
  static __device__ __forceinline__ Rec unpack(const T &v) { return v; }
};
typedef u32 u32x5 __attribute__((ext_vector_type(5)));
template <> struct RegRec<Tup0> {
  typedef u32x5 T;
  static __device__ __forceinline__ T pack(const Tup0 &r) { T v; v[0] = r.pos; v[1] = r.c0; v[2] = r.c1; v[3] = r.r1; v[4] = r.r2; return v; }
  static __device__ __forceinline__ Tup0 unpack(const T &v) { return Tup0{v[0], v[1], v[2], v[3], v[4]}; }
};

// PF: prefetch the next tile into registers while the current one is ranked/reordered (pays for
// 8-byte records: 2.3 -> 3.4 TB/s; costs registers and loses for 16/20-byte records, see
// profiles/r01_radix_downsweep_variants_v2.txt).
// kAtomicRank (lab only, not used by the product): the rank inside the wave's digit run comes from ONE returning LDS
// atomic per record instead of 8-9 ballots per round — stable only because gfx950's LDS serves the lanes of one
// instruction that hit the same address in ascending lane order (measured in round 2, not documented).
template <class Rec, int NB, int IPT, int NW, bool PF, class Loader, class Sink = RecSink<Rec>, bool kAtomicRank = false>
__global__ __launch_bounds__(NW * 64) void k_rs_downsweep(Loader in, Sink out, u32 n,
                                                         u32 chunk, u32 nchunks, KeyDig dig,
                                                         const u32 *__restrict__ table,
                                                         const u32 *__restrict__ digit_base, u32 xcd_cpx = 0) {
  // xcd_cpx != 0: block j works chunk (j % 8) * xcd_cpx + j / 8 — the blocks that share an XCD (round-robin placement,
  // a speed assumption only) take CONSECUTIVE chunks at about the same time, so the runs that end up next to each other
  // in the output are written through one L2 within microseconds and their partial lines meet there
  const u32 cid = xcd_cpx ? (blockIdx.x & 7u) * xcd_cpx + (blockIdx.x >> 3) : blockIdx.x;
  if (cid >= nchunks) return;
  constexpr int kB = NW * 64;
  constexpr int kTile = kB * IPT;
  constexpr int kWItems = 64 * IPT;
  constexpr int kBits = NB == 512 ? 9 : 8;
  static_assert(NB == 256 || NB == 512, "digit bins");
  static_assert(NW * 64 >= NB, "one thread per digit");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Rec *srec = reinterpret_cast<Rec *>(smem);
  u32 *wcnt = reinterpret_cast<u32 *>(smem + sizeof(Rec) * kTile);   // [NW][NB]
  u32 *dbase = wcnt + NW * NB;                                       // [NB] running global base
  u32 *texcl = dbase + NB;                                           // [NB] tile-exclusive prefix
  u32 *tmp = texcl + NB;                                             // [NW]
  const u32 tid = threadIdx.x, lane = lane_id(), w = wave_id();
  const u32 begin = cid * chunk;
  const u32 end = min(n, begin + chunk);
  if (tid < NB) dbase[tid] = digit_base[tid] + table[(size_t)tid * nchunks + cid];
  u32 *mycnt = wcnt + w * NB;
  typedef RegRec<Rec> RR;
  // (record types held as they are load straight into their register slot, as before round 5: going through a temporary
  //  cost the 12-byte down-sweep 20 % — 66.5 -> 79.6 ms at 2^31 records — with the same register count)
  constexpr bool kSameRec = std::is_same<typename RR::T, Rec>::value;
  typename RR::T r[IPT], rn[PF ? IPT : 1];
  bool okn[PF ? IPT : 1];
  if (PF) {
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if constexpr (kSameRec) okn[PF ? k : 0] = (begin + t < end) && in.load(begin + t, rn[PF ? k : 0]);
      else { Rec tmp_rec; okn[PF ? k : 0] = (begin + t < end) && in.load(begin + t, tmp_rec); rn[PF ? k : 0] = RR::pack(tmp_rec); }
    }
  }

  for (u32 tile = begin; tile < end; tile += kTile) {
    const u32 nin = min((u32)kTile, end - tile);
#pragma unroll
    for (int j = 0; j < NB / 64; j++) mycnt[lane + 64 * j] = 0;
    u32 d[IPT], rk[IPT];
    bool ok[IPT];
    // wave w owns tile items [w*kWItems, (w+1)*kWItems); round k covers 64 consecutive items
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      const u32 t = w * kWItems + k * 64 + lane;
      if (PF) { r[k] = rn[PF ? k : 0]; ok[k] = okn[PF ? k : 0]; }
      else if constexpr (kSameRec) ok[k] = (t < nin) && in.load(tile + t, r[k]);
      else { Rec tmp_rec; ok[k] = (t < nin) && in.load(tile + t, tmp_rec); r[k] = RR::pack(tmp_rec); }
      if constexpr (kSameRec) d[k] = ok[k] ? digit_of(r[k], dig) : 0u;
      else d[k] = ok[k] ? digit_of(RR::unpack(r[k]), dig) : 0u;
    }
    if (PF) {
      const u32 nt = tile + kTile;
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        const u32 t = w * kWItems + k * 64 + lane;
        if constexpr (kSameRec) okn[PF ? k : 0] = (nt + t < end) && in.load(nt + t, rn[PF ? k : 0]);
        else { Rec tmp_rec; okn[PF ? k : 0] = (nt + t < end) && in.load(nt + t, tmp_rec); rn[PF ? k : 0] = RR::pack(tmp_rec); }
      }
    }
    // stable ranking: items of one wave-round with equal digit are ordered by lane.  The lowest
    // peer lane bumps the wave's digit counter with one LDS atomic per round; the atomics of all
    // rounds are issued back to back (LDS executes them in order, so the returned values are the
    // running prefix) and the bases are broadcast afterwards.  Dropped / out-of-range lanes take
    // no part.
    if (kAtomicRank) {
#pragma unroll
      for (int k = 0; k < IPT; k++)
        rk[k] = ok[k] ? __hip_atomic_fetch_add(&mycnt[d[k]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
    } else {
      u32 below[IPT], old[IPT];
#pragma unroll
      for (int k = 0; k < IPT; k++) {
        // peers = lanes whose digit equals mine: a lane differs from me in bit b iff ballot(b) ^ mybit(b) has its
        // bit set, so peers = ~OR_b (ballot_b ^ splat(mybit_b)), restricted to participating lanes
        const u64 okm = __ballot(ok[k]);
        u32 dlo = 0, dhi = 0;
#pragma unroll
        for (int bit = 0; bit < kBits; bit++) {
          const u32 om = (u32)((int)(d[k] << (31 - bit)) >> 31);      // all ones iff my bit is set
          const u64 mk = __ballot(om != 0u);
          dlo |= (u32)mk ^ om; dhi |= (u32)(mk >> 32) ^ om;
        }
        const u64 peers = ~(((u64)dhi << 32) | dlo) & okm;
        below[k] = mbcnt(peers);
        // every lane reads the wave's running count of its digit, then the lowest peer adds the round's count
        // (no return value needed); DS operations of one wave execute in issue order, so round k+1 reads what
        // round k added
        old[k] = ok[k] ? __hip_atomic_load(&mycnt[d[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) : 0u;
        if (ok[k] && below[k] == 0)
          (void)__hip_atomic_fetch_add(&mycnt[d[k]], (u32)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
#pragma unroll
      for (int k = 0; k < IPT; k++) rk[k] = old[k] + below[k];
    }
    __syncthreads();
    // per digit (thread tid = digit): prefix over waves, tile total, tile-exclusive prefix
    u32 tot = 0;
    if (tid < NB) {
#pragma unroll
      for (int i = 0; i < NW; i++) { const u32 c = wcnt[i * NB + tid]; wcnt[i * NB + tid] = tot; tot += c; }
    }
    u32 nkeep;
    const u32 ex = block_excl_scan<NW>(tid < NB ? tot : 0u, tmp, nkeep);
    if (tid < NB) texcl[tid] = ex;
    __syncthreads();
    // reorder through LDS so every digit run is contiguous
#pragma unroll
    for (int k = 0; k < IPT; k++) {
      if constexpr (kSameRec) { if (ok[k]) srec[texcl[d[k]] + wcnt[w * NB + d[k]] + rk[k]] = r[k]; }
      else { if (ok[k]) srec[texcl[d[k]] + wcnt[w * NB + d[k]] + rk[k]] = RR::unpack(r[k]); }
    }
    __syncthreads();
    if constexpr (SinkReads<Sink>::value) {
      // a sink that READS at the destination (the verifier's comparing sink): four destinations per thread and round, their
      // loads issued together — one dependent load per loop iteration left the pass waiting on memory latency (9 ms per 2^30
      // records against 6 for the storing form)
      for (u32 q0 = tid; q0 < nkeep; q0 += 4 * kB) {
        Rec x[4]; u32 g[4]; typename Sink::Fetched v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const u32 q = min(q0 + (u32)u * kB, nkeep - 1u);
          x[u] = srec[q];
          const u32 dd = digit_of(x[u], dig);
          g[u] = dbase[dd] + (q - texcl[dd]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = out.fetch(g[u]);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (q0 + (u32)u * kB < nkeep) out.check(g[u], x[u], v[u]);
      }
    } else {
      for (u32 q = tid; q < nkeep; q += kB) {
        const Rec x = srec[q];
        const u32 dd = digit_of(x, dig);
        out.store(dbase[dd] + (q - texcl[dd]), x);
      }
    }
    __syncthreads();
    if (tid < NB) dbase[tid] += tot;
    // (the barrier after ranking in the next iteration orders this update before its use)
  }
}

// Row-wise exclusive scan of the [NB][nchunks] digit table: block d scans row d in place and
// writes the row total to totals[d] (then scanned by k_scan_excl_inplace over NB entries).
__global__ __launch_bounds__(kBlock) void k_scan_rows(u32 *__restrict__ table, u32 nchunks, u32 *__restrict__ totals) {
  __shared__ u32 tmp[kWaves];
  u32 *row = table + (size_t)blockIdx.x * nchunks;
  u32 carry = 0;
  for (u32 base = 0; base < nchunks; base += kBlock * 4) {
    const u32 i0 = base + threadIdx.x * 4;
    u32 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (i0 + j < nchunks) ? row[i0 + j] : 0u;
    u32 tot;
    u32 ex = block_excl_scan<kWaves>(v[0] + v[1] + v[2] + v[3], tmp, tot) + carry;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (i0 + j < nchunks) row[i0 + j] = ex; ex += v[j]; }
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

}  // namespace dc3
