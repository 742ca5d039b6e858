// dc3_msd.hip.hpp — bucket (MSD) ordering of packed (image << pbits | pos) words.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); namespace dc3.
//
// The prefix sort of the tie-refine orderings (dc3_order.hip.hpp) sorts 64-bit words by their image bits; the
// words are all distinct (the position is part of the word), so "ascending by the whole word" is the one result every
// correct sort produces — the stable LSD passes of lib.rs:15-39 are one way to get it, and stability is only what
// LSD needs of its own passes.  This file gets the same array with three passes instead of four or five, none of
// which has to be stable:
//   pass 1   k_msd_part<false>  partition by the top d1 image bits          (sizes: digit table of the pack kernel)
//   pass 2   k_msd_part<true>   partition every bucket by the next d2 bits  (sizes: k_msd_hist2 + k_msd_scan2a/c)
//   pass 3   k_msd_local        every sub-bucket (about a thousand words) is ordered inside LDS: counting sort on the
//                               next bits into ~one word per bin, then each word counts the smaller words of its bin
// A partition pass needs no per-(digit, chunk) table and no row scan: a tile ranks its words with one returning LDS
// atomic each and reserves its run in every bucket with one global atomicAdd per (tile, digit).
//
// XCD-aware placement (measured on MI355X, profiles/r03a_radix_lab.jsonl: 2^30 words, 1024 buckets, 7.0 -> 3.9 ms per
// pass).  A tile's run in a bucket is 8-16 words, i.e. mostly PARTIAL 128-byte lines, and the neighbouring run belongs
// to whichever tile reserved next.  The eight XCDs have private L2s: when the neighbours run on different XCDs both
// halves of the shared line leave their L2 as partial writes.  So every bucket is split into eight regions, one per
// group of blocks with equal blockIdx % 8 — the blocks that share an XCD under the round-robin placement the
// dispatcher is observed to use — and a group reserves only inside its own region: neighbouring runs are then written
// through ONE L2 within a microsecond of each other and leave it as full lines.  The groups own contiguous eighths of
// the input (block j works tile (j % 8) * cpx + j / 8), so the per-group bucket sizes come straight from the pack
// kernel's per-chunk digit table.  Placement is a speed assumption only: any block may run anywhere, the result does
// not change.
//
// Sub-bucket sizes depend on the data: k_msd_scan2a reports the largest, and the host falls back to the LSD passes
// when it exceeds the local sort's capacity (skewed images; the orderings that call this are only taken on
// high-entropy input).
#pragma once

namespace dc3 {

// `base` (all kernels): the smallest word of the range the records come from, (image_lo << pbits); digits are taken from
// word - base, so that a caller holding only a slice of the image range (a rank of the global mode) still gets evenly
// filled buckets.  0 for the whole range.
// The records are Rec8 {key = high half, val = low half} in memory (dc3_radix.hip.hpp: rec8_word); as one 8-byte load
// that is the word with its halves exchanged — msd_word() puts them back (a register rename, no instruction).
__device__ __forceinline__ u64 msd_word(u64 mem) { return (mem << 32) | (mem >> 32); }

constexpr int kMsdNW = 16, kMsdIPT = 8, kMsdTile = kMsdNW * 64 * kMsdIPT;      // 8192 words per partition tile
constexpr int kMsdMaxDig = 1024;                                               // d1, d2 <= 10
constexpr size_t kMsdPartSmem = sizeof(u64) * kMsdTile + sizeof(u32) * (2 * kMsdMaxDig + 64);
constexpr u32 kMsdHistTile = 8u * kMsdTile;                                    // words per block of k_msd_hist2
constexpr u32 kMsdGroups = 8;                                                  // XCDs
constexpr u32 kMsdScanSeg = 8192;                                              // entries per block of the size scans
// device words of a sort (plan[] in the kernels below)
enum { kMsdW_T2 = 0, kMsdW_CPX2 = 1, kMsdW_MAXSUB = 2, kMsdW_MAXB1 = 3 /* largest pass-1 bucket */, kMsdW_COUNT = 4 };

// Digit table of the top d1 image bits when the records were not packed for this ordering (callers that build their
// records elsewhere, e.g. the ranks of the global mode): table[d * nchunks + c] as the pack kernels write it.
__global__ __launch_bounds__(kBlock) void k_msd_hist1(const u64 *__restrict__ in, u32 n, u64 base, u32 shift, u32 chunk, u32 nchunks,
                                                     u32 *__restrict__ table /*[1024][nchunks]*/) {
  __shared__ u32 hist[kWaves][kMsdMaxDig];
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < kMsdMaxDig; j += kBlock) hist[w][j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  for (u32 i = begin + threadIdx.x; i < end; i += kBlock) atomicAdd(&myh[(u32)((msd_word(in[i]) - base) >> shift) & (kMsdMaxDig - 1)], 1u);
  __syncthreads();
  for (int j = threadIdx.x; j < kMsdMaxDig; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}

// cntg[d * 8 + g] = words of bucket d inside group g's eighth of the input: the pack kernel's digit table
// table[d * nchunks + c] summed over the chunks of the group (chunk c belongs to group c / cpg).  One block per bucket.
__global__ __launch_bounds__(kBlock) void k_msd_cnt1(const u32 *__restrict__ table, u32 nchunks, u32 cpg,
                                                    u32 *__restrict__ cntg) {
  __shared__ u32 acc[kMsdGroups];
  if (threadIdx.x < kMsdGroups) acc[threadIdx.x] = 0;
  __syncthreads();
  const u32 d = blockIdx.x;
  for (u32 c = threadIdx.x; c < nchunks; c += kBlock) {
    const u32 v = table[(size_t)d * nchunks + c];
    if (v) atomicAdd(&acc[c / cpg], v);
  }
  __syncthreads();
  if (threadIdx.x < kMsdGroups) cntg[d * kMsdGroups + threadIdx.x] = acc[threadIdx.x];
}

// From the per-group bucket sizes (nb1 <= 1024 buckets; thread d owns the 8 entries of bucket d):
//   startg[d * 8 + g]  start of group g's region of bucket d (exclusive prefix in that order), startg[nb1 * 8] = n
//   cur1[g * nb1 + d]  the pass-1 cursors, one plane per group
//   bstart[d]          start of bucket d, bstart[nb1] = n
//   tpre / tpreh       tiles of kMsdTile / kMsdHistTile words per bucket, exclusive prefix (tiles never straddle a
//                      bucket), [nb1] = total
//   plan[T2, CPX2]     pass-2 tiles and tiles per group
// One block of 1024 threads.
__global__ __launch_bounds__(1024) void k_msd_plan1(const u32 *__restrict__ cntg, u32 nb1, u32 n, u32 *__restrict__ startg,
                                                   u32 *__restrict__ cur1, u32 *__restrict__ bstart, u32 *__restrict__ tpre,
                                                   u32 *__restrict__ tpreh, u32 *__restrict__ plan) {
  __shared__ u32 tmp[16];
  const u32 d = threadIdx.x;
  u32 v[kMsdGroups], c = 0;
#pragma unroll
  for (u32 g = 0; g < kMsdGroups; g++) { v[g] = d < nb1 ? cntg[d * kMsdGroups + g] : 0u; c += v[g]; }
  u32 tot;
  u32 ex = block_excl_scan<16>(c, tmp, tot);
  if (d < nb1) {
    bstart[d] = ex;
#pragma unroll
    for (u32 g = 0; g < kMsdGroups; g++) { startg[d * kMsdGroups + g] = ex; cur1[g * nb1 + d] = ex; ex += v[g]; }
  }
  const u32 ext = block_excl_scan<16>((c + kMsdTile - 1) / kMsdTile, tmp, tot);
  if (d < nb1) tpre[d] = ext;
  if (d == 0) { tpre[nb1] = tot; plan[kMsdW_T2] = tot; plan[kMsdW_CPX2] = max(1u, (tot + kMsdGroups - 1) / kMsdGroups); }
  { const u32 mx = wave_reduce_max(c); if (lane_id() == 0 && mx) atomicMax(&plan[kMsdW_MAXB1], mx); }
  const u32 exh = block_excl_scan<16>((c + kMsdHistTile - 1) / kMsdHistTile, tmp, tot);
  if (d < nb1) tpreh[d] = exh;
  if (d == 0) { tpreh[nb1] = tot; bstart[nb1] = n; startg[nb1 * kMsdGroups] = n; }
}

// bucket of tile t: the largest b with tpre[b] <= t (tpre is non-decreasing; buckets without tiles are skipped)
__device__ __forceinline__ u32 msd_find_bucket(const u32 *__restrict__ tpre, u32 nb1, u32 t) {
  u32 lo = 0, hi = nb1;                 // invariant: tpre[lo] <= t < tpre[hi]
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (tpre[mid] <= t) lo = mid; else hi = mid; }
  return lo;
}

// One partition pass; block j belongs to group g = j % 8 and works that group's tile number j / 8.
// kSeg = false (pass 1): group g owns tiles [g * cpx, (g + 1) * cpx) of the input (tile t = words [t * kMsdTile, ...)),
//   digit = word >> shift (the top d1 bits), cursors[g * ndig + digit].
// kSeg = true (pass 2): the tiles are those of the bucket list (tpre / bstart: tile t lies inside one bucket b), group g
//   owns tiles [g * cpx2, (g + 1) * cpx2), digit = (word >> shift) & mask, cursors[g * gstride + (b << dbits) + digit].
// kHi (the host's choice: base == 0 and shift >= 32): the digit is a bit field of the word's upper half — one instruction
//   instead of a 64-bit subtraction and a 64-bit shift, three times per word.
// LDS: the two digit tables first — their reads take a constant offset from the digit's own address — the words behind
//   them.  After the scan gbase[d] holds (start of the tile's run of d in the output) - (its start inside the tile), so
//   the word at tile index q goes to gbase[d] + q.  A full tile (all but the last of a bucket) runs without the per-word
//   guards.  These kernels are bound by instruction issue as much as by HBM (62 VALU instructions per word at first).
// Not stable.
template <bool kHi>
__device__ __forceinline__ u32 msd_digit(u64 x, u64 base, u32 shift, u32 mask) {
  if (kHi) return ((u32)(x >> 32) >> (shift - 32u)) & mask;
  return (u32)((x - base) >> shift) & mask;
}
struct MsdPartLds {
  u32 *hist, *gbase, *tmp; u64 *srec;
  __device__ __forceinline__ explicit MsdPartLds(unsigned char *smem)
      : hist(reinterpret_cast<u32 *>(smem)), gbase(hist + kMsdMaxDig), tmp(gbase + kMsdMaxDig),
        srec(reinterpret_cast<u64 *>(smem + sizeof(u32) * (2 * kMsdMaxDig + 64))) {}
};
// kSlot (pass 2 only; the host's choice when the pass-1 buckets are even and the arena has room): sub-bucket s of the output
//   is a SLOT of slot_cap words at s * slot_cap — no sizes are needed before the pass, so the counting sweep over all words
//   (k_msd_hist2: 8.6 GB read for 2^30 words) and its scans before the pass are not run.  cursors[s] counts the words of s;
//   a run that does not fit goes to the dump area behind the slots (dump_base) and the host — which sees the largest count
//   after the pass — runs the counted form from the untouched input.
template <bool kSeg, bool kHi, bool kSlot = false>
__global__ __launch_bounds__(kMsdNW * 64) void k_msd_part(const u64 *__restrict__ in, u64 *__restrict__ out, u32 n, u64 base, u32 shift,
                                                         u32 dbits, u32 cpx, u32 ntiles, const u32 *__restrict__ tpre,
                                                         const u32 *__restrict__ bstart, u32 nb1, const u32 *__restrict__ plan,
                                                         u32 *__restrict__ cursors, u32 gstride, u32 *__restrict__ xcdmon,
                                                         u32 slot_cap = 0, u32 dump_base = 0) {
  static_assert(!kSlot || kSeg, "slots are sub-buckets");
  constexpr int NT = kMsdNW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const MsdPartLds L(smem);
  const u32 tid = threadIdx.x;
  const u32 ndig = 1u << dbits, mask = ndig - 1u;
  const u32 g = blockIdx.x % kMsdGroups, idx = blockIdx.x / kMsdGroups;
  u32 begin, end, bkt = 0;
  u32 *cur = cursors + (kSlot ? (size_t)0 : (size_t)g * gstride);
  if (kSeg) {
    const u32 cpx2 = plan[kMsdW_CPX2], t2 = plan[kMsdW_T2];
    const u32 tile = g * cpx2 + idx;
    if (idx >= cpx2 || tile >= t2) return;
    const u32 b = msd_find_bucket(tpre, nb1, tile);
    begin = bstart[b] + (tile - tpre[b]) * (u32)kMsdTile;
    end = min(begin + (u32)kMsdTile, bstart[b + 1]);
    if (!kSlot) cur += (size_t)b << dbits;
    bkt = b;
  } else {
    const u32 tile = g * cpx + idx;
    if (idx >= cpx || tile >= ntiles) return;
    begin = tile * (u32)kMsdTile;
    end = min(n, begin + (u32)kMsdTile);
  }
  const u32 nvalid = end - begin;                  // >= 1
  xcd_note(xcdmon, g);
  L.hist[tid] = 0;
  __syncthreads();
  auto body = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;
    u64 r[kMsdIPT];
    u32 rk[kMsdIPT], dg[kMsdIPT];
    // (all loads issued back to back: the index is clamped instead of guarded, a guard would put a wait behind every load)
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++) r[k] = in[begin + (kFull ? (u32)(k * NT) + tid : min((u32)(k * NT) + tid, nvalid - 1u))];
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++) { r[k] = msd_word(r[k]); dg[k] = msd_digit<kHi>(r[k], base, shift, mask); }
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++)
      if (kFull || (u32)(k * NT) + tid < nvalid) rk[k] = atomicAdd(&L.hist[dg[k]], 1u);
    __syncthreads();
    u32 cnt = 0, gb = 0;
    if (tid < ndig) {
      cnt = L.hist[tid];
      if (cnt) {
        if (kSlot) {
          const u32 s = (bkt << dbits) + tid;
          const u32 old = atomicAdd(&cur[s], cnt);
          gb = old + cnt <= slot_cap ? s * slot_cap + old : dump_base;
        } else {
          gb = atomicAdd(&cur[tid], cnt);
        }
      }
    }
    u32 tot;
    const u32 ex = block_excl_scan<kMsdNW>(cnt, L.tmp, tot);
    L.hist[tid] = ex;
    L.gbase[tid] = gb - ex;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++)
      if (kFull || (u32)(k * NT) + tid < nvalid) L.srec[L.hist[dg[k]] + rk[k]] = r[k];
    __syncthreads();
    if (kFull) {
#pragma unroll
      for (int k = 0; k < kMsdIPT; k++) {
        const u32 q = (u32)(k * NT) + tid;
        const u64 x = L.srec[q];
        out[L.gbase[msd_digit<kHi>(x, base, shift, mask)] + q] = msd_word(x);
      }
    } else {
      for (u32 q = tid; q < nvalid; q += NT) {
        const u64 x = L.srec[q];
        out[L.gbase[msd_digit<kHi>(x, base, shift, mask)] + q] = msd_word(x);
      }
    }
  };
  if (nvalid == (u32)kMsdTile) body(std::true_type{}); else body(std::false_type{});
}

// Pass 1 that makes its words on the fly: tile t = the positions [t * kMsdTile, ...) of the text / level, their words
// computed from the key maker (images4: 4 consecutive positions per thread and round, as the pack kernels do) instead of
// being read — the pack kernel then only has to COUNT (k_pack_image_text<…, kStore = false>), and 8 bytes per position
// are neither written nor read back.  Everything after the load is k_msd_part<false>.
// kStrip: `hm` describes an image that is dbits WIDER than the word has room for (hm.pbits = real position bits - dbits).
// The bucket a word lands in already says what its top dbits are, so the stored word keeps only the bits below them:
//     word = (image mod 2^(hm.nbits - dbits)) << (hm.pbits + dbits) | position
// — the order inside every bucket is unchanged, and the image is 2^dbits times finer: at 1 GiB, 44 instead of 34 bits,
// 0.006 % of the words tied instead of 6 % (the tie pass then has next to nothing to gather).  Words that agree across a
// bucket boundary are compared by the tie pass like any tie, consistently with their images.  Inside the tile the
// position field of the LDS copy carries (digit, tile-local index) instead — the position is begin + index — so the
// digit survives the reorder without a second LDS array.  Needs hm.pbits + dbits >= 23.
// kSel (one rank of the global mode, "replicate what is read at random, split the work by key range"): the tile is still
// 8192 consecutive positions of the REPLICATED text / level, but only the words whose image lies in the rank's range are
// partitioned — the buckets are those of the whole image range, a rank simply fills its share of them, and a tile's runs
// are as long as the single device's (a P-th of the words into a P-th of the buckets).
// (MsdSel: dc3_order.hip.hpp)
// the counting pack kernel of a selecting pass 1: table[d * nchunks + chunk] = selected words of bucket d in the chunk
// (hshift = image bits below the bucket's), *total += all of them
template <class KM>
__global__ __launch_bounds__(kBlock) void k_msd_count_sel(KM km, HiMap hm, u64 P1, u32 n, MsdSel sel, u32 chunk, u32 nchunks,
                                                         u32 *__restrict__ table, u32 hshift, u32 *__restrict__ total) {
  __shared__ uint16_t lcode[256];
  __shared__ u32 hist[kWaves][kMsdMaxDig];
  __shared__ u32 bsum;
  km.stage(lcode);
#pragma unroll
  for (int w = 0; w < kWaves; w++)
    for (int j = threadIdx.x; j < (int)kMsdMaxDig; j += kBlock) hist[w][j] = 0;
  if (threadIdx.x == 0) bsum = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);       // chunk is a multiple of 4
  for (u32 p0 = begin + 4 * threadIdx.x; p0 < end; p0 += 4 * kBlock) {
    u64 img[4];
    images4(km, hm, P1, p0, n, lcode, img);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (p0 + j < end && msd_sel_keep(sel, img[j])) atomicAdd(&myh[(u32)(img[j] >> hshift) & (kMsdMaxDig - 1)], 1u);
  }
  __syncthreads();
  u32 mine = 0;
  for (int j = threadIdx.x; j < (int)kMsdMaxDig; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
    mine += sum;
  }
  if (mine) atomicAdd(&bsum, mine);
  __syncthreads();
  if (threadIdx.x == 0 && bsum) atomicAdd(total, bsum);
}
template <class KM, bool kStrip, bool kSel = false>
__global__ __launch_bounds__(kMsdNW * 64, 8) void k_msd_part_keys(KM km, HiMap hm, u64 P1, u64 *__restrict__ out, u32 n, u64 base, u32 shift,
                                                              u32 dbits, u32 cpx, u32 ntiles, const u32 *__restrict__ plan,
                                                              u32 *__restrict__ cursors, u32 gstride, u32 *__restrict__ xcdmon,
                                                              MsdSel sel = MsdSel{0, 0, 1, 0}) {
  constexpr int NT = kMsdNW * 64;
  static_assert(kMsdIPT == 8, "two rounds of 4 positions per thread");
  static_assert(kMsdTile == 8192 && kMsdMaxDig == 1024, "kStrip keeps (digit, index) in 10 + 13 bits");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const MsdPartLds L(smem);
  __shared__ uint16_t lcode[256];
  const u32 tid = threadIdx.x;
  const u32 ndig = 1u << dbits, mask = ndig - 1u;
  const u32 g = blockIdx.x % kMsdGroups, idx = blockIdx.x / kMsdGroups;
  const u32 tile = g * cpx + idx;
  if (idx >= cpx || tile >= ntiles) return;
  u32 *cur = cursors + (size_t)g * gstride;
  const u32 begin = tile * (u32)kMsdTile, end = min(n, begin + (u32)kMsdTile);
  const u32 nvalid = end - begin;
  xcd_note(xcdmon, g);
  // (the code table only where the images need it — the raw image of byte alphabets and the key makers without a table do
  //  not, and the tile then starts with its text loads instead of a table load and a barrier; the barrier in front of the
  //  ranking orders the zeroed counters)
  if (KM::kCodes && !hm.raw) { km.stage(lcode); __syncthreads(); }
  L.hist[tid] = 0;
  const u32 pb = hm.pbits + (kStrip ? dbits : 0u);                 // position bits of the stored word
  auto body = [&](auto full_tag) {
    constexpr bool kFull = decltype(full_tag)::value;               // a whole tile, every word kept: no per-word guards
    u64 r[kMsdIPT];
    u32 rk[kMsdIPT], dg[kMsdIPT];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const u32 p0 = begin + (u32)(k * NT + tid) * 4u;
      u64 img[4] = {0, 0, 0, 0};
      if (kFull || p0 < end) images4(km, hm, P1, p0, n, lcode, img);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u32 t = (u32)(k * NT + tid) * 4u + (u32)j;
        if (kStrip) {
          // hm.nbits + hm.pbits = 64 (the host makes it so): with the image at the top of 64 bits the digit is the top of
          // the upper half and the kept bits are one shift away; (digit, index) sit in the lower half's bits 0 .. 22
          const u64 y = img[j] << hm.pbits;
          dg[k * 4 + j] = (u32)(y >> 32) >> (32u - dbits);
          r[k * 4 + j] = (y << dbits) | (u64)((dg[k * 4 + j] << 13) | t);
        } else {
          r[k * 4 + j] = (img[j] << hm.pbits) | (u64)(p0 + j);
          dg[k * 4 + j] = (u32)((r[k * 4 + j] - base) >> shift) & mask;
        }
        // rk = ~0: not a word of this tile (past the end, or — kSel — not in this rank's image range)
        if (!kFull) rk[k * 4 + j] = (t < nvalid && (!kSel || msd_sel_keep(sel, img[j]))) ? 0u : ~0u;
      }
    }
    // word k * 4 + j of thread tid is tile element t = (k * NT + tid) * 4 + j
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++)
      if (kFull || rk[k] != ~0u) rk[k] = atomicAdd(&L.hist[dg[k]], 1u);
    __syncthreads();
    u32 cnt = 0, gb = 0;
    if (tid < ndig) {
      cnt = L.hist[tid];
      if (cnt) gb = atomicAdd(&cur[tid], cnt);
    }
    u32 tot;
    const u32 ex = block_excl_scan<kMsdNW>(cnt, L.tmp, tot);
    L.hist[tid] = ex;
    L.gbase[tid] = gb - ex;                      // word at tile index q of digit d goes to gbase[d] + q
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kMsdIPT; k++)
      if (kFull || rk[k] != ~0u) L.srec[L.hist[dg[k]] + rk[k]] = r[k];
    __syncthreads();
    auto emit = [&](u32 q) {
      const u64 x = L.srec[q];
      if (kStrip) {
        const u32 dd = ((u32)x >> 13) & mask;
        const u64 w = (x & ~((1ull << pb) - 1ull)) | (u64)(begin + ((u32)x & (kMsdTile - 1u)));
        out[L.gbase[dd] + q] = msd_word(w);
      } else {
        const u32 dd = (u32)((x - base) >> shift) & mask;
        out[L.gbase[dd] + q] = msd_word(x);
      }
    };
    if (kFull) {
#pragma unroll
      for (int k = 0; k < kMsdIPT; k++) emit((u32)(k * NT) + tid);
    } else {
      const u32 nout = kSel ? tot : nvalid;              // (tot = the tile's selected words)
      for (u32 q = tid; q < nout; q += NT) emit(q);
    }
  };
  if (!kSel && nvalid == (u32)kMsdTile) body(std::true_type{}); else body(std::false_type{});
}

// Sizes of the sub-buckets per group: block h counts the d2-digits of its piece (8 pass-2 tiles) of bucket b in LDS and
// adds them to cnt2g[((b << d2) + digit) * 8 + g], g = the group that will work the tile in pass 2 (tile / cpx2).
// No barrier inside the tile loop (hipcc once left out the LDS wait in front of such a barrier, DESIGN.md appendix "ISA
// barrier scan"): the piece's tiles belong to at most 8 consecutive groups, and each group has its own LDS plane.
__global__ __launch_bounds__(1024) void k_msd_hist2(const u64 *__restrict__ in, u64 base, u32 shift, u32 dbits,
                                                   const u32 *__restrict__ tpre, const u32 *__restrict__ tpreh,
                                                   const u32 *__restrict__ bstart, u32 nb1, const u32 *__restrict__ plan,
                                                   u32 *__restrict__ cnt2g) {
  __shared__ u32 hist[kMsdGroups][kMsdMaxDig];
  if (blockIdx.x >= tpreh[nb1]) return;
  const u32 tid = threadIdx.x;
  const u32 ndig = 1u << dbits, mask = ndig - 1u;
  const u32 cpx2 = plan[kMsdW_CPX2];
  const u32 b = msd_find_bucket(tpreh, nb1, blockIdx.x);
  const u32 hh = blockIdx.x - tpreh[b];
  const u32 begin = bstart[b] + hh * kMsdHistTile;
  const u32 end = min(begin + kMsdHistTile, bstart[b + 1]);
  const u32 tile0 = tpre[b] + hh * (kMsdHistTile / kMsdTile);
  const u32 g0 = tile0 / cpx2;
#pragma unroll
  for (u32 p = 0; p < kMsdGroups; p++) hist[p][tid] = 0;
  __syncthreads();
  u32 gl = g0;
  for (u32 pt = 0; pt < kMsdHistTile / kMsdTile; pt++) {
    const u32 pb = begin + pt * (u32)kMsdTile;
    if (pb >= end) break;
    const u32 pe = min(pb + (u32)kMsdTile, end);
    gl = (tile0 + pt) / cpx2;                         // (block-uniform; gl - g0 <= 7: the groups are 0..7)
    u32 *h = hist[gl - g0];
    u64 w[kMsdTile / 1024];
#pragma unroll
    for (u32 k = 0; k < kMsdTile / 1024; k++) w[k] = in[min(pb + k * 1024u + tid, pe - 1u)];
#pragma unroll
    for (u32 k = 0; k < kMsdTile / 1024; k++)
      if (pb + k * 1024u + tid < pe) atomicAdd(&h[(u32)((msd_word(w[k]) - base) >> shift) & mask], 1u);
  }
  __syncthreads();
  if (tid < ndig)
    for (u32 p = 0; p <= gl - g0; p++) {
      const u32 v = hist[p][tid];
      if (v) atomicAdd(&cnt2g[(((size_t)b << dbits) + tid) * kMsdGroups + g0 + p], v);
    }
}

// Exclusive prefix of cnt[0..N) (N = sub-buckets * 8, in that order), in place, in two launches of ceil(N / 8192) blocks
// of 1024 threads (thread t owns 8 consecutive entries = the 8 groups of one sub-bucket):
//   k_msd_scan2a: segsum[block] = sum of the block's entries; plan[MAXSUB] = largest sub-bucket
//   k_msd_scan2c: cnt[i] <- exclusive prefix (the block re-adds the sums of the blocks before it: N / 8192 <= 1024 of
//                 them), cnt[N] = total, and cur2[g * n2 + s] = the same values as one plane per group (pass-2 cursors)
__global__ __launch_bounds__(1024) void k_msd_scan2a(const u32 *__restrict__ cnt, u32 N, u32 *__restrict__ segsum,
                                                    u32 *__restrict__ plan) {
  __shared__ u32 tmp[16];
  const u32 i0 = blockIdx.x * kMsdScanSeg + threadIdx.x * kMsdGroups;
  u32 s = 0;
#pragma unroll
  for (u32 g = 0; g < kMsdGroups; g++) s += (i0 + g < N) ? cnt[i0 + g] : 0u;
  const u32 mx = wave_reduce_max(s);
  u32 tot;
  (void)block_excl_scan<16>(s, tmp, tot);
  if (threadIdx.x == 0) segsum[blockIdx.x] = tot;
  if (lane_id() == 0 && mx) atomicMax(&plan[kMsdW_MAXSUB], mx);
}
__global__ __launch_bounds__(1024) void k_msd_scan2c(u32 *__restrict__ cnt, u32 N, u32 n2, const u32 *__restrict__ segsum,
                                                    u32 *__restrict__ cur2) {
  __shared__ u32 tmp[16];
  // sum of the segments before this one
  u32 before = (threadIdx.x < blockIdx.x) ? segsum[threadIdx.x] : 0u;     // gridDim.x <= 1024
  u32 tot;
  (void)block_excl_scan<16>(before, tmp, tot);
  const u32 base = tot;
  const u32 i0 = blockIdx.x * kMsdScanSeg + threadIdx.x * kMsdGroups;
  u32 v[kMsdGroups], s = 0;
#pragma unroll
  for (u32 g = 0; g < kMsdGroups; g++) { v[g] = (i0 + g < N) ? cnt[i0 + g] : 0u; s += v[g]; }
  u32 ex = block_excl_scan<16>(s, tmp, tot) + base;
  const u32 sub = i0 / kMsdGroups;
#pragma unroll
  for (u32 g = 0; g < kMsdGroups; g++) {
    if (i0 + g < N) { cnt[i0 + g] = ex; cur2[(size_t)g * n2 + sub] = ex; }
    ex += v[g];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) cnt[N] = base + tot;
}

// The same two launches for the slot cursors of k_msd_part<.., kSlot>: cnt[0..N) = words per sub-bucket (one entry each) ->
// exclusive prefix in place, cnt[N] = total, plan[MAXSUB] = the largest entry.
__global__ __launch_bounds__(1024) void k_msd_slot_scan_a(const u32 *__restrict__ cnt, u32 N, u32 *__restrict__ segsum, u32 *__restrict__ plan) {
  __shared__ u32 tmp[16];
  const u32 i0 = blockIdx.x * kMsdScanSeg + threadIdx.x * 8u;
  u32 s = 0, mx = 0;
#pragma unroll
  for (u32 g = 0; g < 8u; g++) { const u32 v = (i0 + g < N) ? cnt[i0 + g] : 0u; s += v; mx = max(mx, v); }
  mx = wave_reduce_max(mx);
  u32 tot;
  (void)block_excl_scan<16>(s, tmp, tot);
  if (threadIdx.x == 0) segsum[blockIdx.x] = tot;
  if (lane_id() == 0 && mx) atomicMax(&plan[kMsdW_MAXSUB], mx);
}
__global__ __launch_bounds__(1024) void k_msd_slot_scan_c(u32 *__restrict__ cnt, u32 N, const u32 *__restrict__ segsum) {
  __shared__ u32 tmp[16];
  u32 before = (threadIdx.x < blockIdx.x) ? segsum[threadIdx.x] : 0u;     // gridDim.x <= 1024
  u32 tot;
  (void)block_excl_scan<16>(before, tmp, tot);
  const u32 base = tot;
  const u32 i0 = blockIdx.x * kMsdScanSeg + threadIdx.x * 8u;
  u32 v[8], s = 0;
#pragma unroll
  for (u32 g = 0; g < 8u; g++) { v[g] = (i0 + g < N) ? cnt[i0 + g] : 0u; s += v[g]; }
  u32 ex = block_excl_scan<16>(s, tmp, tot) + base;
#pragma unroll
  for (u32 g = 0; g < 8u; g++) { if (i0 + g < N) cnt[i0 + g] = ex; ex += v[g]; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 1023) cnt[N] = base + tot;
}

// Sinks of the local sort: where word x of global output index g goes (cf. RecSink / SplitSink of the LSD passes).
// kSame: the sink also wants to know whether the word's image equals its predecessor's in the sorted order (= some
// smaller word of its bin has the same image: equal images share every image bit, hence the sub-bucket and the bin).
// image_floor(x) = x with its position bits cleared: a smaller word w has x's image iff w >= image_floor(x).
struct MsdRecSink {
  u64 *p;
  static constexpr bool kSame = false;
  __device__ __forceinline__ u64 image_floor(u64 x) const { return x; }
  __device__ __forceinline__ void store(u32 g, u64 x, bool) const { p[g] = msd_word(x); }
};
// the sorted words + the tie pass's byte (the orderings that keep records: whole-level orders inside the recursion)
struct MsdRecSameSink {
  u64 *p; uint8_t *same; u32 pbits;
  static constexpr bool kSame = true;
  __device__ __forceinline__ u64 image_floor(u64 x) const { return x & ~((1ull << pbits) - 1ull); }
  __device__ __forceinline__ void store(u32 g, u64 x, bool sm) const { p[g] = msd_word(x); same[g] = sm ? 1 : 0; }
};
// positions to the suffix-array buffer + ONE BYTE per word for the tie pass: 1 = same image as the word before it.  (The
// LSD passes' SplitSink leaves 32 image bits instead, which the tie pass compares itself: 8 bytes per word written and 4
// read back, against 5 and 1 here.)
// tilef (optional): one byte per tile of the tie pass (kTieTile records, k_tie_resolve_split), set when a record of the tile
// is tied or is the predecessor of a tied record — 0.006 % of random records are, so the pass reads one byte instead of 4 KB
// of flags for the nine tiles in ten that hold no tie (0.21 -> 0.14 ms per GiB; what remains are the gathers of the tied records).
// (kTieTileShift, kTieTile: dc3_order.hip.hpp)
struct MsdSplitSink {
  u32 *sa; uint8_t *same; u32 pbits; uint8_t *tilef;
  static constexpr bool kSame = true;
  __device__ __forceinline__ u64 image_floor(u64 x) const { return x & ~((1ull << pbits) - 1ull); }
  __device__ __forceinline__ void store(u32 g, u64 x, bool sm) const {
    sa[g] = (u32)(x & ((1ull << pbits) - 1ull));
    same[g] = sm ? 1 : 0;
    if (sm && tilef) { tilef[g >> kTieTileShift] = 1; tilef[(g - (g != 0u)) >> kTieTileShift] = 1; }   // (benign race: every writer stores 1)
  }
};

// Pass 3: block s orders sub-bucket s = words [start[8 s], start[8 (s + 1)]) (at most CAP of them; larger ones were
// refused on the host) and writes it through the sink at the same indices.  (start = the exclusive prefix of the sub-bucket
// sizes either way: counted before pass 2, or read off the slot cursors after it.)
//   1. bin = BB bits below the sub-bucket bits; one returning LDS atomic per word gives its arrival number in the bin
//   2. exclusive scan of the bin counts
//   3. words are placed bin by bin in LDS (arrival order inside a bin)
//   4. every word counts the smaller words of its own bin (about one word per bin): final index = bin start + count
// The words are distinct, so the result is the unique ascending order.
// kHi (the host's choice: base == 0, shb >= 32): the bin is a bit field of the word's upper half.
// LDS holds kMsdLocPad words of all ones behind the sub-bucket: step 4 reads its bin two words at a time without a bound —
// what follows a bin are the later bins, all of them larger words.
constexpr int kMsdLocPad = 2;
template <int NT, int CAP, int BB, class Sink, bool kHi>
__global__ __launch_bounds__(NT) void k_msd_local(const u64 *__restrict__ in, const u32 *__restrict__ start, u64 base, u32 shb, Sink out,
                                                  u32 slot_cap) {
  constexpr int IPT = CAP / NT, NBIN = 1 << BB, BPT = NBIN / NT;
  static_assert(CAP % NT == 0 && NBIN % NT == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // CAP + kMsdLocPad words
  u64 *srec = reinterpret_cast<u64 *>(smem);
  __shared__ u32 cnt[NBIN + 1];
  __shared__ u32 tmp[NT / 64];
  const u32 tid = threadIdx.x;
  // (start: one entry per (sub-bucket, group) in the counted form, one per sub-bucket with slots)
  const u32 sstr = slot_cap ? 1u : kMsdGroups;
  const u32 begin = start[(size_t)blockIdx.x * sstr], end = start[(size_t)(blockIdx.x + 1) * sstr];
  const u32 m = end - begin;
  if (m == 0) return;
#pragma unroll
  for (int j = 0; j < BPT; j++) cnt[j * NT + tid] = 0;
  if (tid < (u32)kMsdLocPad) srec[m + tid] = ~0ull;
  __syncthreads();
  u64 r[IPT];
  u32 rk[IPT], bn[IPT];
  // (clamped, not guarded: all loads in flight at once)
  // (slot_cap != 0: the sub-bucket's words lie in its slot, k_msd_part<.., kSlot>; they go out at `begin` all the same)
  const size_t src0 = slot_cap ? (size_t)blockIdx.x * slot_cap : (size_t)begin;
#pragma unroll
  for (int k = 0; k < IPT; k++) r[k] = in[src0 + min((u32)(k * NT) + tid, m - 1u)];
#pragma unroll
  for (int k = 0; k < IPT; k++) { r[k] = msd_word(r[k]); bn[k] = msd_digit<kHi>(r[k], base, shb, NBIN - 1); }
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < m) rk[k] = atomicAdd(&cnt[bn[k]], 1u);
  }
  __syncthreads();
  // exclusive scan of the bins: thread tid owns bins [tid * BPT, (tid + 1) * BPT)
  u32 c[BPT], s = 0;
#pragma unroll
  for (int j = 0; j < BPT; j++) { c[j] = cnt[tid * BPT + j]; s += c[j]; }
  u32 tot;
  u32 ex = block_excl_scan<NT / 64>(s, tmp, tot);
#pragma unroll
  for (int j = 0; j < BPT; j++) { cnt[tid * BPT + j] = ex; ex += c[j]; }
  if (tid == NT - 1) cnt[NBIN] = ex;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IPT; k++) {
    const u32 t = k * NT + tid;
    if (t < m) srec[cnt[bn[k]] + rk[k]] = r[k];
  }
  __syncthreads();
  for (u32 q = tid; q < m; q += NT) {
    const u64 x = srec[q];
    const u64 xf = out.image_floor(x);
    const u32 bin = msd_digit<kHi>(x, base, shb, NBIN - 1);
    const u32 lo = cnt[bin], hi = cnt[bin + 1];
    u32 less = 0, lessf = 0;              // words of the bin below x / below x's image
    for (u32 j = lo; j < hi; j += 2) {
      const u64 w0 = srec[j], w1 = srec[j + 1];
      less += (w0 < x ? 1u : 0u) + (w1 < x ? 1u : 0u);
      if (Sink::kSame) lessf += (w0 < xf ? 1u : 0u) + (w1 < xf ? 1u : 0u);
    }
    out.store(begin + lo + less, x, less != lessf);
  }
}

}  // namespace dc3
