// dc3_host_level.hpp — unwinding of a level (tuples, mod-0 order, merge), the level driver, the whole-text orders, the build
// Host side of libdc3hip (single translation unit: included by dc3hip.hip in this order; everything here is static).
#pragma once

// keys64: compact tuples whose symbols fit 15 bits (the caller knows K): the 64-bit-key image of k_merge
template <int NT, int VT, class TA, class TB>
static int launch_merge(dc3hip_ctx *c, u32 ntiles, const TA *A, u32 nA, const TB *B, u32 nB, const u32 *part,
                        u32 *out_sa, Rec8 *out_pairs, u32 rank_base = 0, bool keys64 = false) {
  const size_t smem = MergeSmem<NT, VT>::kBytes;
  if constexpr (std::is_same<TA, TupC>::value) {
    if (keys64) {
      auto kern = k_merge<NT, VT, TA, TB, true>;
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), smem, c->stream, A, nA, B, nB, part, out_sa, out_pairs, rank_base);
      return E_OK;
    }
  }
  auto kern = k_merge<NT, VT, TA, TB, false>;
  HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(NT), smem, c->stream, A, nA, B, nB, part, out_sa, out_pairs, rank_base);
  return E_OK;
}

// DC3HIP_TRACE=1 (stage-level parity, the counterpart of the reference's crosscheck! macro,
// crates/divsufsort/src/crosscheck.rs:17-84): order-sensitive checksums of a level's three canonical arrays — the
// sorted samples SA12 (as text positions of the level, the dummy included), the sorted mod-0 suffixes SA0 and the
// level's suffix array — which do not depend on HOW names were made (dense by sorting or packed directly), so the
// CPU restatement the tests check against emits the same words and can be compared level by level.
enum { TR_SA12 = 0, TR_SA0 = 1, TR_SA = 2 };
static int trace_sum(dc3hip_ctx *c, int which, int depth, const void *arr, u32 n, int kind /*0 u32 positions, 1 slots, 2 Tup0, 3 Tup0C*/, u32 m0) {
  if (!c->trace || depth >= DC3HIP_MAX_LEVELS || n == 0) return E_OK;
  u64 *acc = c->d_trace + (size_t)which * DC3HIP_MAX_LEVELS + depth;
  hipLaunchKernelGGL(k_trace_sum, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, arr, n, kind, m0, acc);
  KCHECK();
  return E_OK;
}

// Sample tuples of slots sa12l[0..cnt) in that order -> t12 (lib.rs:136-162's reads, gathered once): the slot table is
// built by streaming (8-byte entries when the level's symbols fit 16 bits, else 16-byte) and gathered.  table0
// ([256][chunks of cnt]) receives the digit table of the fused mod-0 selection pass.  The slot table lives above the
// caller's arena mark and is released here.
// Sample tuples in SA12 order by scattering instead of gathering (dc3_merge.hip.hpp, "WITHOUT the random gather"), as
// COMPACT tuples (TupC, 12 bytes).  Level 0 (bytes) moves 8-byte records through the two partition passes and reads the
// first symbol off a table of cumulative counts; deeper levels whose symbols fit 16 bits move 12-byte records.
// *done = false when the level is too small or the arena too short for the two record arrays (the caller gathers).
static inline u32 nb_of(u32 m02) { return ((m02 - 1) >> kTupSh1) + 1; }
template <class Sym, class Out, bool kDerive>
static int scatter_tuples_run(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, const u32 *rank12, const u32 *sa12, const Chunking &ckc,
                              TupC *t12, u32 *table0, bool *done) {
  typedef typename Out::Rec Rec;
  *done = false;
  // (level 0, 8-byte words: pass 1 in tiles of 6144 slots on 1024 threads, k_tup8_part1, unless DC3HIP_TUP_BIGTILE=0)
  const bool big = kDerive;
  // one cursor per bucket at its analytic base (k_tup_cur_init): level 0's 8-byte records only — the 12-byte records of the
  // deeper levels lost 1.4 ms per 477 M without the split between the XCD groups (measured, round 5)
  const bool analytic = kDerive;
  const u32 gstr = analytic ? 0u : nb_of(m02);
  const u32 tile1 = big ? (u32)kTup8Tile : (u32)kTupTile;
  const u32 ntiles = (m02 + tile1 - 1) / tile1;
  const u32 tpc = std::max<u32>(1, (ntiles + 2047) / 2048);
  const u32 cpg = ((ntiles + 7) / 8 + tpc - 1) / tpc, cpx = cpg * tpc;
  const u32 chunk = tpc * tile1, nchunks = (m02 + chunk - 1) / chunk;
  const u32 nb = nb_of(m02);                                     // buckets of 2^22 destinations (<= 1024)
  if (c->arena_bytes - c->arena_off < (size_t)m02 * 2 * sizeof(Rec) + (size_t)1024 * nchunks * 4 + ((size_t)nb << 11) + (16u << 20)) return E_OK;
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_tup_part1<Sym, Out>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTupPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_tup_part2<Out>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTupPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_tup_local<Out, kDerive>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * kTupWin * 4)));
    attr_set[c->device & 15] = true;
  }
  const ArenaMark mk = arena_mark(c);
  Rec *ra = nullptr, *rb = nullptr;
  u32 *table1 = nullptr, *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *cur2 = nullptr;
  u32 *cum = nullptr;
  const u32 nsym = 258;                                          // level 0: codes 0..sigma <= 256 (+ slack)
  RC(arena_alloc(c, (size_t)m02, &ra)); RC(arena_alloc(c, (size_t)m02, &rb));
  RC(arena_alloc(c, (size_t)1024 * nchunks, &table1));
  RC(arena_alloc(c, (size_t)nb * 8 + 16, &cntg)); RC(arena_alloc(c, (size_t)nb * 8 + 16, &startg)); RC(arena_alloc(c, (size_t)nb * 8 + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb + 16, &bstart)); RC(arena_alloc(c, (size_t)nb + 16, &tpre)); RC(arena_alloc(c, (size_t)nb + 16, &tpreh));
  RC(arena_alloc(c, (size_t)16, &plan)); RC(arena_alloc(c, (size_t)nb * 512, &cur2));
  RC(arena_alloc(c, (size_t)nsym + 16, &cum));
  const u32 rbits = bits_of(m02);                                // r <= m02
  Out oa, ob;
  oa.p = ra; ob.p = rb;
  if constexpr (kDerive) { oa.rb = rbits; ob.rb = rbits; }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
    if (analytic) {
      hipLaunchKernelGGL(k_tup_cur_init, dim3((nb + 255) / 256), dim3(256), 0, c->stream, cur1, nb);
      KCHECK();
    } else {
      hipLaunchKernelGGL(k_tup_hist1, dim3(nchunks), dim3(kBlock), 0, c->stream, rank12, m02, chunk, nchunks, table1);
      KCHECK();
      hipLaunchKernelGGL(k_msd_cnt1, dim3(nb), dim3(kBlock), 0, c->stream, (const u32 *)table1, nchunks, cpg, cntg);
      KCHECK();
      hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb, m02, startg, cur1, bstart, tpre, tpreh, plan);
      KCHECK();
    }
    HIPC(hipMemsetAsync(cur2, 0, (size_t)nb * 512 * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(table0, 0, (size_t)256 * ckc.nchunks * sizeof(u32), c->stream));
    if (kDerive) {
      HIPC(hipMemsetAsync(cum, 0, (size_t)(nsym + 1) * sizeof(u32), c->stream));
      hipLaunchKernelGGL((k_sample_sym_hist<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, nsym, cum);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cum, nsym + 1, (u32 *)nullptr);
      KCHECK();
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02, 3);
    if constexpr (kDerive) {
      if (big) {
        static std::atomic<bool> attr8[16];
        if (!attr8[c->device & 15]) {
          HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_tup8_part1<Sym>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTup8PartSmem));
          attr8[c->device & 15] = true;
        }
        hipLaunchKernelGGL((k_tup8_part1<Sym>), dim3(8 * cpx), dim3(kTup8NT), kTup8PartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa, gstr);
      } else {
        hipLaunchKernelGGL((k_tup_part1<Sym, Out>), dim3(8 * cpx), dim3(kTupNT), kTupPartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa, gstr);
      }
    } else {
      hipLaunchKernelGGL((k_tup_part1<Sym, Out>), dim3(8 * cpx), dim3(kTupNT), kTupPartSmem, c->stream, S, m, m0, m02, rank12, cpx, ntiles, nb, cur1, oa, gstr);
    }
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02, 3);
    const u32 tpb = (1u << kTupSh1) / kTupTile;
    hipLaunchKernelGGL((k_tup_part2<Out>), dim3(8 * ((nb + 7) / 8) * tpb), dim3(kTupNT), kTupPartSmem, c->stream, (const Rec *)ra, m02, nb, cur2, ob, rbits);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
    hipLaunchKernelGGL((k_tup_local<Out, kDerive>), dim3((m02 + kTupWin - 1) / kTupWin), dim3(1024), 2 * kTupWin * 4, c->stream, (const Rec *)rb, rbits, sa12, m02, m0,
                       ckc.chunk, ckc.nchunks, (const u32 *)cum, nsym, t12, table0);
    KCHECK();
  }
  arena_release(c, mk);
  *done = true;
  return E_OK;
}
template <class Sym>
static int scatter_tuples(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, const u32 *rank12, const u32 *sa12, const Chunking &ckc,
                          TupC *t12, u32 *table0, bool *done) {
  *done = false;
  if (m02 < c->tup_scatter_min || m02 < 2 || ckc.chunk < kTupWin) return E_OK;
  if constexpr (std::is_same<Sym, SymU8>::value) {
    return scatter_tuples_run<Sym, TupOut8, true>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done);
  }
  return scatter_tuples_run<Sym, TupOut12, false>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done);
}

template <class Sym>
static int build_gather_tuples(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u64 K, const u32 *rank12, const u32 *sa12l,
                               u32 cnt, const Chunking &ckc, Tup12 *t12, u32 *table0) {
  const ArenaMark mk = arena_mark(c);
  PhaseScope ps(c, DC3HIP_PH_TUPLES, m02);
  if (K < 65536) {
    TupS8 *ts = nullptr;
    RC(arena_alloc(c, (size_t)m02, &ts));
    hipLaunchKernelGGL((k_build_tuples8<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, rank12, ts);
    KCHECK();
    if (cnt) {
      PhaseScope pg(c, DC3HIP_PH_OTHER, cnt, 4);   // timed separately as kernel class 4 (gather)
      hipLaunchKernelGGL(k_gather_tuples8, dim3(ckc.nchunks), dim3(kBlock), 0, c->stream, ts, sa12l, cnt, m0, ckc.chunk,
                         ckc.nchunks, t12, table0);
      KCHECK();
    }
  } else {
    Tup12 *ts = nullptr;
    RC(arena_alloc(c, (size_t)m02, &ts));
    hipLaunchKernelGGL((k_build_tuples<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, rank12, ts);
    KCHECK();
    if (cnt) {
      PhaseScope pg(c, DC3HIP_PH_OTHER, cnt, 4);
      hipLaunchKernelGGL(k_gather_tuples, dim3(ckc.nchunks), dim3(kBlock), 0, c->stream, ts, sa12l, cnt, ckc.chunk, ckc.nchunks,
                         t12, table0);
      KCHECK();
    }
  }
  arena_release(c, mk);
  return E_OK;
}

template <int kMergeNT, int kMergeVT, class TA, class TB>
static int merge_lists_shape(dc3hip_ctx *c, const TA *A, u32 nA, const TB *B, u32 nB, u32 *out_sa, Rec8 *out_pairs,
                             u32 rank_base, bool keys64) {
  const u32 total = nA + nB;
  if (total == 0) return E_OK;
  const u32 tile = (u32)kMergeNT * kMergeVT;
  const u32 ntiles = (total + tile - 1) / tile;
  const ArenaMark mk = arena_mark(c);
  u32 *part = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &part));
  {
    PhaseScope ps(c, DC3HIP_PH_MERGE, total);
    // coarse split of every 16th tile boundary first, then the bounded per-tile searches
    constexpr u32 kRatio = 16;
    const u32 nco = (ntiles + kRatio - 1) / kRatio;              // coarse tiles of kRatio*tile outputs
    u32 *coarse = nullptr;
    RC(arena_alloc(c, (size_t)nco + 16, &coarse));
    hipLaunchKernelGGL((k_merge_partition<TA, TB>), dim3((nco + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, A, nA, B, nB,
                       nco, tile * kRatio, (const u32 *)nullptr, 1u, coarse);
    KCHECK();
    hipLaunchKernelGGL((k_merge_partition<TA, TB>), dim3((ntiles + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, A, nA, B, nB,
                       ntiles, tile, (const u32 *)coarse, kRatio, part);
    KCHECK();
    RC((launch_merge<kMergeNT, kMergeVT, TA, TB>(c, ntiles, A, nA, B, nB, part, out_sa, out_pairs, rank_base, keys64)));
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}
// Step 3 (lib.rs:131-192): merge-path merge of the sorted sample tuples A and the sorted mod-0 tuples B into
// out_sa[0 .. nA+nB) (and, when out_pairs != nullptr, the (pos, rank_base + k + 1) pairs of the rank inversion).
template <class TA, class TB>
static int merge_lists(dc3hip_ctx *c, const TA *A, u32 nA, const TB *B, u32 nB, u32 *out_sa, Rec8 *out_pairs,
                       u32 rank_base, bool keys64 = false) {
  // 1024 threads x 2 outputs: re-measured in round 4 on the compact tuples against 512 x 4, 1024 x 4, 256 x 8, 512 x 8
  // (merge of 1.07 G suffixes: 6.2 / 7.0 / 8.0 / 10.2 / 10.7 ms, profiles/r04g_lab_shapes.jsonl)
  return merge_lists_shape<1024, 2, TA, TB>(c, A, nA, B, nB, out_sa, out_pairs, rank_base, keys64);
}

// Steps 2 + 3 of a level (lib.rs:118-192) on compact tuples: sample tuples scattered into SA12 order (TupC), mod-0
// tuples (Tup0C) selected and ordered by the fused radix pass(es), merge.  *done = false: nothing happened, the caller
// runs the general form.
template <class Sym>
static int unwind_compact(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m1, u32 m02, u64 K, const u32 *rank12, const u32 *sa12,
                          u32 *out_sa, u32 *out_rank, int depth, bool *done) {
  *done = false;
  const ArenaMark mk = arena_mark(c);
  TupC *t12 = nullptr;
  RC(arena_alloc(c, (size_t)m02, &t12));
  constexpr u32 kTup0Tile = SortCfg<Tup0C, 256>::NW * 64 * SortCfg<Tup0C, 256>::IPT;
  const Chunking ckc = make_chunks(c, m02, kTup0Tile);
  u32 *table0 = nullptr, *dbase0 = nullptr;
  RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
  RC(arena_alloc(c, (size_t)256, &dbase0));
  RC((scatter_tuples<Sym>(c, S, m, m0, m02, rank12, sa12, ckc, t12, table0, done)));
  if (!*done) { arena_release(c, mk); return E_OK; }
  Tup0C *z0 = nullptr, *z1 = nullptr, *zs = nullptr;
  RC(arena_alloc(c, (size_t)m0, &z0));
  RC(arena_alloc(c, (size_t)m0, &z1));
  {
    // pass 0 of the mod-0 sort reads the sample tuples directly (selection fused in the loader)
    RC(scan_digit_table(c, table0, ckc.nchunks, dbase0, 256, DC3HIP_PH_COMPACT));
    Mod0LoaderC ld; ld.t = t12;
    KeyDig dig; dig.shift = 0; dig.mask = 255;
    // (tiles of 8192 slots; 6144 and 4096 — two blocks per CU — were measured at 5.6 and 6.1 ms against 5.2 for 716 M slots)
    RC((launch_downsweep<Tup0C, 256, Mod0LoaderC>(c, ld, z0, m02, ckc, dig, table0, dbase0, DC3HIP_PH_COMPACT)));
  }
  RC(radix_sort<Tup0C>(c, z0, z1, m0, 8, bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  RC(trace_sum(c, TR_SA0, depth, zs, m0, 3, m0));
  {
    const u32 dskip = m0 - m1;                  // lib.rs:133: skip the dummy, which sorts first
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    RC(merge_lists(c, t12 + dskip, m02 - dskip, zs, m0, out_sa, pa, 0u, K < 32768));     // (symbols of 15 bits: 64-bit merge keys)
    if (out_sa) RC(trace_sum(c, TR_SA, depth, out_sa, m, 0, m0));
    if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
  }
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// one DC3 level (lib.rs:44-193) on the device.
//   S: symbols in 1..K with zero tail, m >= 2
//   out_sa  : [m]      k-th smallest suffix -> position   (may be null)
//   out_rank: [m+3..]  position -> 1-based rank, caller zeroes the tail (may be null)
// ---------------------------------------------------------------------------------------------
template <class Sym>
static int dc3_level(dc3hip_ctx *c, Sym S, u32 m, u64 K, u32 *out_sa, u32 *out_rank, int depth, const Presort *pre) {
  if (depth >= DC3HIP_MAX_LEVELS) { set_err("recursion deeper than %d levels", DC3HIP_MAX_LEVELS); return E_HIP; }
  if (m == 1) {   // single suffix (only reachable as the child of a 2- or 3-symbol level)
    c->stats.level_n[depth] = 1; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
    hipLaunchKernelGGL(k_base1, dim3(1), dim3(64), 0, c->stream, out_sa, out_rank);
    KCHECK();
    return E_OK;
  }
  const u32 m0 = (m + 2) / 3, m1 = (m + 1) / 3, m2 = m / 3, m02 = m0 + m2;   // lib.rs:45-48
  struct DepthScope { dc3hip_ctx *c; int was; DepthScope(dc3hip_ctx *x, int d) : c(x), was(x->cur_depth) { c->cur_depth = d; } ~DepthScope() { c->cur_depth = was; } } depth_scope(c, depth);
  c->stats.level_n[depth] = m; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
  const ArenaMark mk0 = arena_mark(c);

  u32 *rank12 = nullptr, *sa12 = nullptr, *R = nullptr;
  RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
  RC(arena_alloc(c, (size_t)m02 + 16, &sa12));
  RC(arena_alloc(c, (size_t)m02 + 16, &R));

  const u64 B = K + 1;
  // (level 1 takes its sample order from the whole-text order when there is one, whatever its alphabet)
  const bool direct = (B * B * B) <= 0x7fffffffull && !(pre && depth == 1);
  c->stats.level_sorted[depth] = direct ? 0 : 1;   // 2 = prefix-sort + tie-refine
  if (direct) {
    // names = the K–S triple packed in base B (order-preserving); always recurse (distinctness unknown)
    // (packing more symbols per name is order-isomorphic too but was measured slower)
    const u32 w = 3; const u64 Bw = B * B * B;       // B^w
    c->stats.level_name_width[depth] = (int32_t)w;
    {
      PhaseScope ps(c, DC3HIP_PH_NAME_DIRECT, m02);
      hipLaunchKernelGGL((k_name_direct<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         (u32)B, w, (u32)(Bw / B), R);
      KCHECK();
    }
    SymU32 RS; RS.s = R; RS.m = m02;
    RC(dc3_level<SymU32>(c, RS, m02, Bw, sa12, rank12, depth + 1, pre));
  } else {
    const u32 b = (u32)B;                          // packing base of make_rec (K < 2^31)
    u32 kbits = 0;                                 // bit width of B^3 - 1; > 32 here (else direct path)
    { unsigned __int128 mx = (unsigned __int128)B * B * B - 1; while (mx) { kbits++; mx >>= 1; } }
    u32 *sslot = nullptr;                           // sorted slots, only used by the discarding recursion
    RC(arena_alloc(c, (size_t)m02 + 16, &sslot));
    const ArenaMark mk1 = arena_mark(c);
    u32 names = 0;
    int mode = 0;
    bool done = false;
    if (pre && depth == 1) {
      // the whole-text sort of level 0 found duplicate keys; its order, filtered down to this level's samples,
      // is the sorted sample order: name it and continue as usual
      c->stats.level_sorted[depth] = 2;
      AccFilt acc; acc.spos = pre->spos; acc.snf = pre->snf;
      RC(name_and_rank<AccFilt>(c, acc, m02, m0, sa12, rank12, R, sslot, &names, &mode));
      done = true;
    }
    // ---- prefix-sort + tie-refine ordering when the N-bit key image separates most samples ------
    if (!done && m02 >= kHybridMinSamples && !c->no_hybrid && !c->no_hybrid8) {
      double pred = 1.0;
      RC(predict_tie_fraction<Sym>(c, S, m, m0, m02, b, make_himap(B, kbits, m), &pred));
      c->stats.level_tie_pred[depth] = pred;
      // (the whole-level order holds 17 B per position + the filtered samples; skipped when the arena is short)
      if (pred < kFullSortMaxPredicted && !c->no_fullsort &&
          c->arena_bytes - c->arena_off >= (size_t)(m + 1) * 17 + (size_t)m02 * 8 + (64u << 20)) {
        // high entropy: try to finish the whole level by sorting all of its positions
        u32 *spos = nullptr, *snf = nullptr;
        RC(arena_alloc(c, (size_t)m02 + 16, &spos));
        RC(arena_alloc(c, (size_t)m02 + 16, &snf));
        int state = 0;
        Key3<Sym> km; km.S = S; km.B = b;
        RC((order_all_positions<Key3<Sym>, MapSelf>(c, km, MapSelf{}, m, kbits, make_himap(B, kbits, m),
                                                    (m % 3 == 1) ? 1u : 0u, out_sa, out_rank, spos, snf, &state,
                                                    depth)));
        if (state == 1) {
          c->stats.level_sorted[depth] = 5;
          arena_release(c, mk0);
          return E_OK;
        }
        if (state == 2) {      // sorted sample order is already there: name it and continue as usual
          c->stats.level_sorted[depth] = 2;
          AccFilt acc; acc.spos = spos; acc.snf = snf;
          RC(name_and_rank<AccFilt>(c, acc, m02, m0, sa12, rank12, R, sslot, &names, &mode));
          done = true;
        }
      }
      if (!done && pred < c->hybrid_max_pred) {
        bool ok = false;
        RC(order_hybrid<Sym>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode, &ok, depth));
        done = ok;
        if (!ok) arena_release(c, mk1);
      }
    }
    // (with the splitter ordering the full 96-bit key costs three passes: no prefix + tie rounds then)
    if (!done && kbits > 64 && m02 >= c->hybrid12_min && !c->no_hybrid && !ssort_applies(c, m02, kbits)) {
      // wide keys whose 34-bit image collides everywhere: try the 63-bit prefix on 12-byte records
      bool ok = false;
      RC(order_hybrid12<Sym>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode, &ok, depth));
      done = ok;
      if (!ok) arena_release(c, mk1);
    }
    if (!done) {
      const u32 W = wide_window_syms(c, m02, K);
      if (W > 3) {
        c->stats.level_sorted[depth] = 1;
        c->stats.level_name_width[depth] = (int32_t)W;
        RC((order_wide<Sym>(c, S, m, m0, m02, bits_of(K), W, sa12, rank12, R, sslot, &names, &mode)));
        done = true;
      }
    }
    if (!done) {
      c->stats.level_sorted[depth] = 1;
      if (kbits <= 64)
        RC((order_straight<Sym, Rec12>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode)));
      else
        RC((order_straight<Sym, Rec16>(c, S, m, m0, m02, b, kbits, sa12, rank12, R, sslot, &names, &mode)));
    }
    arena_release(c, mk1);
    c->stats.trace_names[depth] = (int64_t)names;
    if (mode == 1) {
      SymU32 RS; RS.s = R; RS.m = m02;
      RC(dc3_level<SymU32>(c, RS, m02, names, sa12, rank12, depth + 1));   // lib.rs:104
    } else if (mode == 2) {
      c->stats.level_sorted[depth] += 2;                                    // 3 / 4 = straight / prefix-sort + discarding
      RC(discard_recurse(c, R, sslot, m02, names, sa12, rank12, depth));
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_OTHER);
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, rank12, m02, 8u);
    KCHECK();
  }
  RC(trace_sum(c, TR_SA12, depth, sa12, m02, 1, m0));

  // ---- Step 2 + 3: tuples, mod-0 order, merge -------------------------------------------------
  // t12 = sample tuples in SA12 order.  The gather also produces the digit table of the fused
  // "select mod-0 + first radix pass" (Step 2, lib.rs:118-126).
  // Levels whose symbols fit 16 bits and that are large enough for the scatter: compact tuples (12 / 16 bytes).
  if (K < 65536) {
    bool done = false;
    RC((unwind_compact<Sym>(c, S, m, m0, m1, m02, K, rank12, sa12, out_sa, out_rank, depth, &done)));
    if (done) { arena_release(c, mk0); return E_OK; }
  }
  Tup12 *t12 = nullptr;
  RC(arena_alloc(c, (size_t)m02, &t12));
  constexpr u32 kTup0Tile = SortCfg<Tup0, 256>::NW * 64 * SortCfg<Tup0, 256>::IPT;
  const Chunking ckc = make_chunks(c, m02, kTup0Tile);
  u32 *table0 = nullptr, *dbase0 = nullptr;
  RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
  RC(arena_alloc(c, (size_t)256, &dbase0));
  RC((build_gather_tuples<Sym>(c, S, m, m0, m02, K, rank12, sa12, m02, ckc, t12, table0)));   // slot table released inside
  Tup0 *z0 = nullptr, *z1 = nullptr, *zs = nullptr;
  RC(arena_alloc(c, (size_t)m0, &z0));
  RC(arena_alloc(c, (size_t)m0, &z1));
  {
    // pass 0 of the mod-0 sort reads the sample tuples directly (selection fused in the loader)
    RC(scan_digit_table(c, table0, ckc.nchunks, dbase0, 256, DC3HIP_PH_COMPACT));
    Mod0Loader ld; ld.t = t12;
    KeyDig dig; dig.shift = 0; dig.mask = 255;
    RC((launch_downsweep<Tup0, 256, Mod0Loader>(c, ld, z0, m02, ckc, dig, table0, dbase0, DC3HIP_PH_COMPACT)));
  }
  RC(radix_sort<Tup0>(c, z0, z1, m0, 8, bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  RC(trace_sum(c, TR_SA0, depth, zs, m0, 2, m0));
  {
    const u32 dskip = m0 - m1;                  // lib.rs:133: skip the dummy, which sorts first
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    RC(merge_lists(c, t12 + dskip, m02 - dskip, zs, m0, out_sa, pa, 0u));
    if (out_sa) RC(trace_sum(c, TR_SA, depth, out_sa, m, 0, m0));
    if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
  }
  arena_release(c, mk0);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// build: level 0 = bytes through the dense code table
// ---------------------------------------------------------------------------------------------
// prologue / epilogue shared by the single-device build and the global (multi-rank) build
static int build_begin(dc3hip_ctx *c) {
  c->built = false;
  c->arena_off = 0; c->arena_peak = 0;
  c->ev_used = 0; c->marks.clear();
  memset(&c->stats, 0, sizeof(c->stats));
  c->stats.struct_size = (int32_t)sizeof(dc3hip_stats);
  c->stats.arena_bytes = (int64_t)c->arena_bytes;
  if (c->n < 0) return E_ARGS;
  HIPC(dc3_set_device(c->device));
  for (int l = 0; l < DC3HIP_MAX_LEVELS; l++) c->stats.trace_names[l] = -1;
  if (c->trace) HIPC(hipMemsetAsync(c->d_trace, 0, 3 * DC3HIP_MAX_LEVELS * sizeof(u64), c->stream));
  HIPC(hipMemsetAsync(c->d_xcdmon, 0, 64 * sizeof(u32), c->stream));
  if (c->profile) HIPC(hipEventRecord(c->ev_build_a, c->stream));
  return E_OK;
}
static int build_end(dc3hip_ctx *c) {
  if (c->profile) HIPC(hipEventRecord(c->ev_build_b, c->stream));
  c->stats.trace_on = c->trace ? 1 : 0;
  if (c->trace) {
    static_assert(sizeof(c->stats.trace_sa12[0]) == sizeof(u64), "trace words");
    HIPC(hipMemcpyAsync(c->stats.trace_sa12, c->d_trace, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->stats.trace_sa0, c->d_trace + DC3HIP_MAX_LEVELS, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(c->stats.trace_sa, c->d_trace + 2 * DC3HIP_MAX_LEVELS, DC3HIP_MAX_LEVELS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  }
  void *monp = nullptr;
  RC(stage_d2h(c, c->d_xcdmon, 64 * sizeof(u32), &monp));
  const u32 *mon = static_cast<const u32 *>(monp);
  {
    // where the XCD-grouped partition blocks of this build really ran: share of them on their group's majority XCD
    u64 all = 0, hit = 0;
    for (int g = 0; g < 8; g++) { u32 mx = 0; for (int x = 0; x < 8; x++) { all += mon[g * 8 + x]; mx = std::max(mx, mon[g * 8 + x]); } hit += mx; }
    c->stats.xcd_blocks = (int64_t)all;
    c->stats.xcd_group_hit = all ? (double)hit / (double)all : 0.0;
    c->stats.xcd_round_robin = c->xcd_rr;
  }
  c->stats.arena_peak = (int64_t)c->arena_peak;
  c->stats.arena_bytes = (int64_t)c->arena_bytes;
  if (c->profile) {
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, c->ev_build_a, c->ev_build_b));
    c->stats.build_ms = ms;
    double by_level[DC3HIP_MAX_LEVELS][DC3HIP_PH_COUNT] = {};
    for (const PhaseMark &m : c->marks) {
      float t = 0;
      if (hipEventElapsedTime(&t, m.a, m.b) != hipSuccess) continue;
      if (m.kclass != 4 && m.depth >= 0 && m.depth < DC3HIP_MAX_LEVELS) by_level[m.depth][m.phase] += t;
      if (m.kclass != 4) {   // class 4 is nested inside the TUPLES phase mark
        c->stats.phase_ms[m.phase] += t;
        c->stats.phase_launches[m.phase] += 1;
      }
      if (m.kclass == 4) { c->stats.gather_ms += t; c->stats.gather_launches += 1; c->stats.gather_elems += m.elems; continue; }
      if (m.kclass == 3) { c->stats.partition_ms += t; c->stats.partition_launches += 1; c->stats.partition_elems += m.elems; }
      if (m.kclass == 5) { c->stats.msd_part_ms += t; c->stats.msd_part_launches += 1; c->stats.msd_part_elems += m.elems; }
      if (m.kclass == 6) { c->stats.msd_local_ms += t; c->stats.msd_local_launches += 1; c->stats.msd_local_elems += m.elems; }
      if (m.kclass == 9) { c->stats.msd_part_keys_ms += t; c->stats.msd_part_keys_launches += 1; c->stats.msd_part_keys_elems += m.elems; }
      if (m.kclass == 7) { c->stats.ssort_part_ms += t; c->stats.ssort_part_launches += 1; c->stats.ssort_part_elems += m.elems; }
      if (m.kclass == 8) { c->stats.ssort_local_ms += t; c->stats.ssort_local_launches += 1; c->stats.ssort_local_elems += m.elems; }
      if (m.kclass >= 0 && m.kclass < 3) {
        c->stats.downsweep_ms[m.kclass] += t; c->stats.downsweep_launches[m.kclass] += 1;
        c->stats.downsweep_elems[m.kclass] += m.elems;
      }
    }
    if (c->level_report) {      // DC3HIP_LEVEL_PHASES=1: the phase times level by level, on stderr (a tuning aid)
      for (int l = 0; l < c->stats.levels && l < DC3HIP_MAX_LEVELS; l++) {
        double sum = 0;
        for (int p = 0; p < DC3HIP_PH_COUNT; p++) sum += by_level[l][p];
        std::fprintf(stderr, "dc3hip level %d n=%lld K=%lld mode=%d total=%.2f ms:", l, (long long)c->stats.level_n[l], (long long)c->stats.level_K[l],
                     c->stats.level_sorted[l], sum);
        for (int p = 0; p < DC3HIP_PH_COUNT; p++) if (by_level[l][p] > 0.005) std::fprintf(stderr, " p%d=%.2f", p, by_level[l][p]);
        std::fprintf(stderr, "\n");
      }
    }
  }
  c->built = true;
  c->builds_done++;
  c->sa_trusted = true;
  c->parts_trusted = 0;
  return E_OK;
}
// level-0 alphabet: dense order-preserving codes 1..sigma of the bytes that occur
// The set of byte values only grows with the bytes looked at: once a PREFIX shows all 256, the rest of the text cannot change
// the code table, and the pass over it (1 GiB: 0.24 ms of an 11 ms build) is left out — exactly, no sampling argument.
// Texts of fewer symbols pay one more launch and stream synchronisation (~0.03 ms) and then scan the rest as before.
static constexpr int64_t kAlphabetPrefix = (int64_t)1 << 20, kAlphabetPrefixMinN = (int64_t)16 << 20;
static int build_alphabet(dc3hip_ctx *c, u32 *sigma_out) {
  const int64_t n = c->n;
  HIPC(hipMemsetAsync(c->d_present, 0, 256 * sizeof(u32), c->stream));
  int64_t from = 0, to = n >= kAlphabetPrefixMinN ? kAlphabetPrefix : n;       // (the prefix is a multiple of the kernel's 16-byte loads)
  u32 sigma = 0;
  for (;;) {
    {
      PhaseScope ps(c, DC3HIP_PH_ALPHABET, to - from);
      hipLaunchKernelGGL(k_byte_presence, dim3(grid_for(c, (u64)(to - from) / 16 + 1)), dim3(kBlock), 0, c->stream, c->d_text + from,
                         (u32)(to - from), c->d_present);
      KCHECK();
      hipLaunchKernelGGL(k_make_codes, dim3(1), dim3(kBlock), 0, c->stream, c->d_present, c->d_code, c->d_words + 1);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 1, c->d_words + 1, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    sigma = c->h_words[1];
    if (to == n || sigma == 256) break;
    from = to; to = n;
  }
  if (sigma < 1 || sigma > 256) { set_err("internal: alphabet size %u", sigma); return E_HIP; }
  *sigma_out = sigma;
  return E_OK;
}

// Whole-text shortcut with key maker KM (three limbs of base BL: Key9's 9 symbols or KeyT's 3L): predicted ties
// permitting, order all n positions by their windows.  All windows distinct: that order is the suffix array
// (*whole_text).  Otherwise the order, filtered down to level 1's samples with the dense ranks of the windows as
// their names, still serves level 1 (*pre): a name built from a window LONGER than the K-S triple orders the samples
// consistently and equal names still imply equal triples, which is all lib.rs:78-104 needs of a name.
// km_pred / hm_pred (beyond 2^31 positions): the key maker and map of the image pass 1 really sorts by — d1 bits wider than
// the 32 image bits of the word — so that the tie prediction is that of the words the tie pass will meet.
template <class KM>
static int try_text_order(dc3hip_ctx *c, KM km, u64 BL, const HiMap &hm, u32 sigma, bool *whole_text, Presort *pre,
                          const KM *km_pred = nullptr, const HiMap *hm_pred = nullptr) {
  const int64_t n = c->n;
  u32 kbits = 0;                          // of the full key (limb base BL)
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  double pred = 1.0;
  if (km_pred && hm_pred) RC(predict_tie_fraction_img<KM>(c, *km_pred, (u32)n, *hm_pred, &pred));
  else RC(predict_tie_fraction_pos<KM>(c, km, (u32)n, hm, &pred));
  c->stats.level_tie_pred[0] = pred;
  if (!text_order_worth_trying(pred, (u64)n, hm_pred ? hm_pred->nbits : hm.nbits)) return E_OK;
  const u32 m0 = (u32)((n + 2) / 3), m1 = m0 + (u32)(n / 3);            // level 1 = string of m1 names
  const u32 m02_1 = (m1 + 2) / 3 + m1 / 3;                              // its samples (incl. the dummy)
  // The filtered order (2 * m02_1 words < n) lives in the output buffer: the optimistic SA written there
  // by the tie pass is void when keys repeat, and nothing else writes d_sa before the final merge.
  u32 *spos = c->d_sa, *snf = c->d_sa + m02_1 + 16;
  MapText mp; mp.m0 = m0; mp.npre = 0; mp.ppos[0] = mp.ppos[1] = 0;
  if (m1 % 3 == 1) mp.ppos[mp.npre++] = m1;                              // level 1's dummy sample
  if (n % 3 == 1 && (m0 - 1) % 3 != 0) mp.ppos[mp.npre++] = m0 - 1;      // level 0's dummy, a level-1 position
  int state = 0;
  RC((order_all_positions<KM, MapText>(c, km, mp, (u32)n, kbits, hm, 0u, c->d_sa, nullptr, spos, snf, &state, 0)));
  c->stats.text_sort_state = state == 1 ? 1 : state == 2 ? 2 : 3;
  if (state == 1) {
    *whole_text = true;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1;
    if (c->stats.level_sorted[0] != 6) c->stats.level_sorted[0] = 5;      // 6 = finished by prefix doubling of the tied positions
  } else if (state == 2) {
    pre->spos = spos; pre->snf = snf;      // duplicates: the order still serves level 1
  }
  return E_OK;
}

template <class KM>
static int launch_pack12_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec12 *out, int nb, const Chunking &ck, u32 *table) {
  if (nb == 512)
    hipLaunchKernelGGL((k_pack_image12_all_hist<KM, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out,
                       ck.chunk, ck.nchunks, table);
  else
    hipLaunchKernelGGL((k_pack_image12_all_hist<KM, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out,
                       ck.chunk, ck.nchunks, table);
  KCHECK();
  return E_OK;
}
template <>
int launch_pack12_all<KeyT>(dc3hip_ctx *c, KeyT km, u32 nrec, const HiMap &hm, Rec12 *out, int nb, const Chunking &ck, u32 *table) {
  u64 P1 = 1;
  for (u32 i = 0; i + 1 < km.J; i++) P1 *= km.sigma;
  if (nb == 512)
    hipLaunchKernelGGL((k_pack_image_textT<512, true>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1,
                       (void *)out, ck.chunk, ck.nchunks, table, 0u);
  else
    hipLaunchKernelGGL((k_pack_image_textT<256, true>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1,
                       (void *)out, ck.chunk, ck.nchunks, table, 0u);
  KCHECK();
  return E_OK;
}
// The same shortcut on 12-byte records (image of ibits <= 63 bits NEXT TO the position instead of sharing a 64-bit
// word with it): beyond 2^31 positions the 8-byte record has 32 image bits left and ties 39 % of even random
// positions; here the image is as wide as the text needs (log2 n + 4.2 bits rounded up to whole 9-bit digits: 36 bits =
// 4 passes of 24 B per record up to 3.5 GiB, 3-5 % ties; 45 bits above).  hm = map of KM's image to ibits.
template <class KM>
static int try_text_order12(dc3hip_ctx *c, KM km, u64 BL, const HiMap &hm, u32 sigma, bool *whole_text, Presort *pre) {
  const int64_t n = c->n;
  u32 kbits = 0;
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  const size_t need = (size_t)n * 26 + ((size_t)256 << 20);
  if (c->arena_bytes - c->arena_off < need) {
    if (c->arena_fixed || c->arena_off != 0) return E_OK;
    if (ensure_arena(c, need) != E_OK) return E_OK;
  }
  const ArenaMark mk = arena_mark(c);
  {
    const u32 stride = std::max<u32>(1, (u32)n >> 20);
    const u32 ns = ((u32)n - 1) / stride + 1;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    u32 ts = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, ns);
      hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
      KCHECK();
    }
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    const double ratio = (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1);
    const double pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
    c->stats.level_tie_pred[0] = pred;
    arena_release(c, mk);
    if (!(pred < kTextSortMaxPredicted)) return E_OK;
  }
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)n, &ha));
  RC(arena_alloc(c, (size_t)n, &hb));
  RC(arena_alloc(c, (size_t)n + 16, &f));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, n);
    int nb = 0; Chunking ck;
    radix_plan<Rec12>(c, (u32)n, hm.nbits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    RC(launch_pack12_all(c, km, (u32)n, hm, ha, nb, ck, first_table));
  }
  RC(radix_sort<Rec12>(c, ha, hb, (u32)n, 0, hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN,
                       first_table));
  bool refined = false, distinct = false, deep_flags = false;
  RC((hybrid12_refine<KM>(c, km, kbits, h, (u32)n, f, &refined, 0, c->d_sa, &distinct, &deep_flags)));
  int state = 0;
  const u32 m0 = (u32)((n + 2) / 3), m1 = m0 + (u32)(n / 3);
  const u32 m02_1 = (m1 + 2) / 3 + m1 / 3;
  u32 *spos = c->d_sa, *snf = c->d_sa + m02_1 + 16;                        // (as in try_text_order)
  bool doubled = false;
  if (refined && !distinct) {
    AccHyb12 acc; acc.h = h; acc.f = f;
    // the rank look-ups of the doubling (binary searches with km.cmp) must compare as deep as the flags were made:
    // after the second tie pass the groups of f[] agree on kDeepSyms symbols, and a search with the window alone
    // would return the lower bound of the whole window-equal run for an untied position behind such a group
    KM kd = km;
    if (deep_flags) kd.deep = kDeepSyms;
    RC((doubling_finish<KM, AccHyb12>(c, kd, acc, (u32)n, deep_flags ? kDeepSyms : km.window_syms(), c->d_sa, &doubled)));
  }
  if (refined && (distinct || doubled)) {
    state = 1;                             // the tie pass (or the doubling rounds) already wrote the suffix array
  } else if (refined) {
    MapText mp; mp.m0 = m0; mp.npre = 0; mp.ppos[0] = mp.ppos[1] = 0;
    if (m1 % 3 == 1) mp.ppos[mp.npre++] = m1;
    if (n % 3 == 1 && (m0 - 1) % 3 != 0) mp.ppos[mp.npre++] = m0 - 1;
    AccHyb12 acc; acc.h = h; acc.f = f;
    RC((finish_position_order<AccHyb12, MapText>(c, acc, mp, (u32)n, (u32)n, 0u, c->d_sa, nullptr, spos, snf, &state)));
  }
  arena_release(c, mk);
  c->stats.text_sort_state = state == 1 ? 1 : state == 2 ? 2 : 3;
  if (state == 1) {
    *whole_text = true;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = doubled ? 6 : 5;
  } else if (state == 2) {
    pre->spos = spos; pre->snf = snf;
  }
  return E_OK;
}

// largest text ordered as 8-byte words: its 2^20 sub-buckets average n / 2^20 words and the local sort holds 4096
// (Poisson: mean 3530 + 5.5 sigma = 3857)
static constexpr u64 kText8MaxN = 3700000000ull;
// the device-resident build proper: SA of c->d_text[0..n) into c->d_sa
static int build_core(dc3hip_ctx *c) {
  const int64_t n = c->n;
  if (n == 1) {
    HIPC(hipMemsetAsync(c->d_sa, 0, 4, c->stream));
  } else if (n >= 2) {
    u32 sigma = 0;
    RC(build_alphabet(c, &sigma));
    SymU8 S; S.t = c->d_text; S.code = c->d_code; S.m = (u32)n;
    bool whole_text = false;
    Presort pre{nullptr, nullptr};
    const u64 Bq = (u64)sigma + 1, B3 = Bq * Bq * Bq;
    // (even uniformly random symbols repeat a w-symbol window once sigma^w is not well above n^2/2: skip then)
    const double need_bits = 2.0 * log2((double)n) + 2.0, sym_bits = log2((double)sigma);
    if ((u64)n >= kHybridMinSamples && !c->no_hybrid && !c->no_fullsort && !c->no_text_shortcut &&
        c->arena_bytes - c->arena_off >= (size_t)n * 22 + (64u << 20)) {
      // whole-text shortcut: if all w-symbol windows of a high-entropy text are distinct, sorting all positions by
      // them is the suffix array (the same test level 1 would make on its triples, without building level 1)
      // 12-byte records (image beside the position) once positions take all 32 bits; DC3HIP_TEXT_ORDER12=1/0 forces
      // / forbids them (tests).  Image width: log2 n + 4.2 bits, rounded up to whole 9-bit digits.
      // Round 6: 8-byte words up to kText8MaxN positions.  Pass 1 of the bucket ordering makes its words from an image d1 = 10
      // bits wider than the word has room for and drops the bucket's own bits (MsdPass1Keys::strip), so a word with a 32-bit
      // position still orders by a 42-bit image: 2^19 tied pairs expected among 2^31 random windows, not 39 % of them.
      // The 12-byte records remain for what that needs and does not have (no bucket ordering on this device, switched off
      // by a test) and for texts whose mean sub-bucket (n / 2^20) would pass the local sort's capacity.
      const bool big = bits_of((u64)n - 1) >= 32;
      const bool strip8 = big && !c->no_msd && !c->no_pack_strip && (u64)n >= c->msd_min && (u64)n <= kText8MaxN;
      const bool wide = c->text_order12 >= 0 ? c->text_order12 == 1 : (big && !strip8);
      constexpr u32 kStripBits = 10;                 // msd_geometry's d1 for 2^29 words and more
      const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2((double)n) + 4.2) / 9.0));
      if (9.0 * sym_bits >= need_bits && B3 * B3 * B3 > 0x7fffffffull) {
        Key9 km; km.S = S; km.B = (u32)Bq; km.B3 = (u32)B3;
        u32 kbits = 0;
        { unsigned __int128 mx = (unsigned __int128)B3 * B3 * B3 - 1; while (mx) { kbits++; mx >>= 1; } }
        HiMap hm = make_himap(B3, kbits, (u32)n, wide ? 64 - ibits : bits_of((u64)n - 1));
        // alphabets past half the byte values: the image is the window's leading bits as they lie in the text
        // (HiMap::raw) — under one bit per symbol given away against the scaled key, and none of its arithmetic
        hm.raw = sigma > 128 && !hm.exact ? 1u : 0u;
        if (wide) RC(try_text_order12<Key9>(c, km, B3, hm, sigma, &whole_text, &pre));
        else if (strip8 && !hm.exact && kbits >= hm.nbits + kStripBits) {
          HiMap hp = make_himap(B3, kbits, (u32)n, hm.pbits - kStripBits);
          hp.raw = hm.raw;
          RC(try_text_order<Key9>(c, km, B3, hm, sigma, &whole_text, &pre, &km, &hp));
          if (!whole_text && !pre.spos && c->stats.msd_fallbacks > 0) {
            // the bucket ordering gave up (a sub-bucket beyond the local sort: a long run, a crowded corner of the image
            // range): the same order on 12-byte records, whose image is as wide as the text needs without the strip
            HiMap h12 = make_himap(B3, kbits, (u32)n, 64 - ibits);
            h12.raw = hm.raw;
            RC(try_text_order12<Key9>(c, km, B3, h12, sigma, &whole_text, &pre));
          }
        } else RC(try_text_order<Key9>(c, km, B3, hm, sigma, &whole_text, &pre));
      } else if (!c->no_long_keys) {
        // small alphabets: limbs of L > 3 symbols (as many as fit 32 bits), 3L-symbol windows
        u32 L = 1; u64 BL = Bq;
        while (L < 20 && BL * Bq <= 0xffffffffull) { BL *= Bq; L++; }
        KeyT km; HiMap hm;
        if (L > 3 && 3.0 * L * sym_bits >= need_bits && make_keyt(S, sigma, L, BL, (u32)n, &km, &hm, wide ? ibits : 0u)) {
          KeyT kp; HiMap hp;
          if (wide) RC(try_text_order12<KeyT>(c, km, BL, hm, sigma, &whole_text, &pre));
          else if (strip8 && make_keyt(S, sigma, L, BL, (u32)n, &kp, &hp, hm.nbits + kStripBits)) {
            RC(try_text_order<KeyT>(c, km, BL, hm, sigma, &whole_text, &pre, &kp, &hp));
            KeyT k12; HiMap h12;
            if (!whole_text && !pre.spos && c->stats.msd_fallbacks > 0 && make_keyt(S, sigma, L, BL, (u32)n, &k12, &h12, ibits))
              RC(try_text_order12<KeyT>(c, k12, BL, h12, sigma, &whole_text, &pre));          // (as above)
          }
          else RC(try_text_order<KeyT>(c, km, BL, hm, sigma, &whole_text, &pre));
        }
      }
    }
    if (!whole_text) {
      RC(ensure_arena(c, arena_requirement(n)));          // (the arena is empty here: the filtered order lives in d_sa)
      RC(dc3_level<SymU8>(c, S, (u32)n, sigma, c->d_sa, nullptr, 0, pre.spos ? &pre : nullptr));
    }
  }
  return E_OK;
}

static int ctx_build_once(dc3hip_ctx *c) {
  RC(build_begin(c));
  RC(build_core(c));
  return build_end(c);
}
// arena_requirement() is a model of the paths' peaks, not a proof: if the bump allocator (not hipMalloc) runs out, the
// arena is grown by half and the build — deterministic, nothing was returned yet — is repeated once.
static int ctx_build(dc3hip_ctx *c) {
  c->arena_exhausted = false;
  int rc = ctx_build_once(c);
  if (rc == E_ALLOC && c->arena_exhausted && !c->arena_fixed) {
    (void)hipStreamSynchronize(c->stream);
    c->arena_off = 0;
    if (ensure_arena(c, c->arena_bytes + c->arena_bytes / 2 + ((size_t)64 << 20)) == E_OK) {
      c->arena_exhausted = false;
      rc = ctx_build_once(c);
    }
  }
  return rc;
}

