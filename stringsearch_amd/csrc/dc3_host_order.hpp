// dc3_host_order.hpp — windowed inversions, naming, discarding recursion and the orderings of a level's samples / of all positions
// Host side of libdc3hip (single translation unit: included by dc3hip.hip in this order; everything here is static).
#pragma once

// ---------------------------------------------------------------------------------------------
// out[key] = val for pairs whose keys are a bijection onto [0,n)  (R[SA12[i]] = i+1, lib.rs:106-108;
// SA12[R[i]-1] = i, lib.rs:111-113).  Two partition passes by the high key bits, then windows of
// 16384 destinations are assembled in LDS and stored with full lines.
// ---------------------------------------------------------------------------------------------
// `first`: the source of the first partition pass (PairArray of `a`, or pairs made on the fly — then the pass writes
// into `a` and `a`'s contents on entry do not matter).  Needs n > 2^14 when `first` is not `a` itself.
template <class Src>
static int inverse_permute_from(dc3hip_ctx *c, Src first, bool first_is_a, Rec8 *a, Rec8 *b, u32 n, u32 *out, int phase) {
  static std::atomic<bool> attr_set[16];   // (per function and device, process-wide; a double set is harmless)
  if (!attr_set[c->device & 15]) {
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_invperm_local),
                             hipFuncAttributeMaxDynamicSharedMemorySize, kInvWindow * 4));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_part_msd<Src>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_part_msd<PairArray>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)kPartSmem));
    attr_set[c->device & 15] = true;
  }
  const u32 kb = bits_of(n > 0 ? n - 1 : 0);
  const ArenaMark mk = arena_mark(c);
  const u32 ntiles = (n + kPartTile - 1) / kPartTile;
  // (a pass fed by `first` writes into a when first is not a itself, else into b)
  Rec8 *src = a, *dst = first_is_a ? b : a;
  bool at_first = true;
  if (kb > 22) {                       // pass 1: top digit = key >> 22 (<= 1024 values for n < 2^32)
    const u32 ndig = ((n - 1) >> 22) + 1;
    u32 *cur = nullptr;
    RC(arena_alloc(c, (size_t)1024, &cur));
    PhaseScope ps(c, phase, n, 3);
    HIPC(hipMemsetAsync(cur, 0, 1024 * sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_part_msd<Src>), dim3(ntiles), dim3(kPartNW * 64), kPartSmem, c->stream, first, dst, n, 22u, 32u,
                       ndig, cur, 0u);
    KCHECK();
    src = dst; dst = (src == a) ? b : a;
    at_first = false;
  }
  if (kb > (u32)kInvWindowBits) {      // pass 2: bits [14,22) inside every 2^22-pair segment
    const u32 nseg = kb > 22 ? ((n - 1) >> 22) + 1 : 1;
    u32 *cur = nullptr;
    RC(arena_alloc(c, (size_t)nseg * 256, &cur));
    PhaseScope ps(c, phase, n, 3);
    HIPC(hipMemsetAsync(cur, 0, (size_t)nseg * 256 * sizeof(u32), c->stream));
    // (with more than one 2^22-pair segment: segment s on the XCD group s % 8, see k_part_msd)
    const u32 tps = (1u << 22) / kPartTile;
    const bool xcd = kb > 22;
    const u32 grid = xcd ? 8u * ((nseg + 7) / 8) * tps : ntiles;
    if (at_first)
      hipLaunchKernelGGL((k_part_msd<Src>), dim3(grid), dim3(kPartNW * 64), kPartSmem, c->stream, first, dst, n,
                         (u32)kInvWindowBits, kb > 22 ? 22u : 32u, 256u, cur, xcd ? tps : 0u);
    else {
      PairArray pa; pa.p = src;
      hipLaunchKernelGGL((k_part_msd<PairArray>), dim3(grid), dim3(kPartNW * 64), kPartSmem, c->stream, pa, dst, n,
                         (u32)kInvWindowBits, kb > 22 ? 22u : 32u, 256u, cur, xcd ? tps : 0u);
    }
    KCHECK();
    src = dst; dst = (src == a) ? b : a;
    at_first = false;
  }
  if (at_first && !first_is_a) { set_err("inverse_permute_from: %u pairs are too few for an on-the-fly source", n); return E_ARGS; }
  {
    PhaseScope ps(c, phase, n);
    hipLaunchKernelGGL(k_invperm_local, dim3((n + kInvWindow - 1) / kInvWindow), dim3(1024), kInvWindow * 4,
                       c->stream, src, n, out);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}
static int inverse_permute(dc3hip_ctx *c, Rec8 *a, Rec8 *b, u32 n, u32 *out, int phase) {
  PairArray pa; pa.p = a;
  return inverse_permute_from<PairArray>(c, pa, true, a, b, n, out, phase);
}

// ---------------------------------------------------------------------------------------------
// naming + rank/name placement shared by both ordering paths (lib.rs:80-113).
//   unique names  -> sa12[i] = slot(pos_i), rank12 = inverse            (lib.rs:109-113)
//   otherwise     -> R[slot(pos_i)] = name_i (+ zero tail), caller recurses (lib.rs:93-104)
// ---------------------------------------------------------------------------------------------
static constexpr double kDiscardMinDropInv = 6.0;  // discard when ~1/6 of the slots would leave the recursion

struct Presort { const u32 *spos, *snf; };   // level-1 samples in sorted order + full names (whole-text sort)
template <class Sym>
static int dc3_level(dc3hip_ctx *c, Sym S, u32 m, u64 K, u32 *out_sa, u32 *out_rank, int depth,
                     const Presort *pre = nullptr);

// mode: 0 = names unique, sa12/rank12 complete; 1 = R holds the names, caller recurses on R (lib.rs:104);
//       2 = R holds name | unique<<31 and sslot the sorted slots: caller runs discard_recurse()
template <class Acc>
static int name_and_rank(dc3hip_ctx *c, Acc acc, u32 m02, u32 m0, u32 *sa12, u32 *rank12, u32 *R, u32 *sslot,
                         u32 *names_out, int *mode) {
  const ArenaMark mk = arena_mark(c);
  // fused form (levels beyond one inversion window): the names are made inside the first partition pass of their
  // inversion (PairsOfNames) from per-tile counts, so the counting kernel works in the partition's tiles
  const bool fused = m02 > (1u << kInvWindowBits);
  Chunking ck = make_chunks(c, m02, kBlock * kNameIPT);
  if (fused) { ck.chunk = kPartTile; ck.nchunks = (m02 + kPartTile - 1) / kPartTile; }
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_NAMING, m02);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, m02, ck.chunk, counts,
                       c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, 5 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));     // the lib.rs:103 decision needs the name count
  const u32 names = c->h_words[0], uniq = c->h_words[4];
  *names_out = names;
  Rec8 *pa = nullptr, *pb = nullptr;
  RC(arena_alloc(c, (size_t)m02, &pa));
  RC(arena_alloc(c, (size_t)m02, &pb));
  if (names == m02) {
    *mode = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_RANKS, m02);
      hipLaunchKernelGGL((k_assign_unique<Acc>), dim3(grid_for(c, m02)), dim3(kBlock), 0, c->stream, acc, m02, m0,
                         sa12, pa);
      KCHECK();
    }
    RC(inverse_permute(c, pa, pb, m02, rank12, DC3HIP_PH_RANKS));
  } else {
    // discard unique names from the recursion when enough slots would leave it to pay for the bookkeeping:
    // a unique slot is dropped iff its predecessor is unique too, so about uniq^2/m02 slots go
    const double drop_est = (double)uniq * (double)uniq / (double)m02;
    const bool discard = sslot && !c->no_discard && m02 < 0x7fffffffu && drop_est * kDiscardMinDropInv >= (double)m02 &&
                         c->arena_bytes - c->arena_off >= (size_t)m02 * 16 + (64u << 20);
    *mode = discard ? 2 : 1;
    if (fused) {
      PairsOfNames<Acc> src; src.acc = acc; src.n = m02; src.m0 = m0; src.base_excl = counts; src.sslot = discard ? sslot : nullptr;
      RC((inverse_permute_from<PairsOfNames<Acc>>(c, src, false, pa, pb, m02, R, DC3HIP_PH_NAMING)));
    } else {
      {
        PhaseScope ps(c, DC3HIP_PH_NAMING, m02);
        hipLaunchKernelGGL((k_name_assign<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, m02, ck.chunk,
                           counts, m0, pa, discard ? sslot : (u32 *)nullptr);
        KCHECK();
      }
      RC(inverse_permute(c, pa, pb, m02, R, DC3HIP_PH_NAMING));
    }
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, R, m02, 8u);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}

// Discarding recursion: see dc3_kernels.hip.hpp.  RU[p] = name | unique<<31 (slot order), sslot[i] =
// slot | unique<<31 (sorted order).  Recurses on the reduced string only; fills sa12 and rank12.
static int discard_recurse(dc3hip_ctx *c, const u32 *RU, const u32 *sslot, u32 m02, u32 names, u32 *sa12,
                           u32 *rank12, int depth) {
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, m02, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 mp = 0;
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 5);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 5, c->d_words + 5, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  mp = c->h_words[5];
  c->stats.level_kept[depth] = mp;
  if (mp == 0) { set_err("internal: discarding kept no slot"); return E_HIP; }
  u32 *Rp = nullptr, *kept = nullptr, *sap = nullptr;
  RC(arena_alloc(c, (size_t)mp + 16, &Rp));
  RC(arena_alloc(c, (size_t)mp + 16, &kept));
  RC(arena_alloc(c, (size_t)mp + 16, &sap));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_write, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts, Rp, kept);
    KCHECK();
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, Rp, mp, 8u);
    KCHECK();
  }
  SymU32 RS; RS.s = Rp; RS.m = mp;
  RC(dc3_level<SymU32>(c, RS, mp, names, sap, nullptr, depth + 1));   // m == 1 is the child's base case
  u32 *x = nullptr, *pt = nullptr;
  RC(arena_alloc(c, (size_t)mp + 16, &x));
  RC(arena_alloc(c, (size_t)mp + 16, &pt));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, mp);
    const Chunking ckp = make_chunks(c, mp, kBlock);
    u32 *cnt2 = nullptr;
    RC(arena_alloc(c, (size_t)ckp.nchunks + 16, &cnt2));
    hipLaunchKernelGGL(k_discard_gather, dim3(grid_for(c, mp)), dim3(kBlock), 0, c->stream, sap, mp, kept, x);
    KCHECK();
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ckp.nchunks), dim3(kBlock), 0, c->stream, x, mp, ckp.chunk, cnt2);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cnt2, ckp.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL(k_nonuniq_write, dim3(ckp.nchunks), dim3(kBlock), 0, c->stream, x, mp, ckp.chunk, cnt2, pt);
    KCHECK();
  }
  Rec8 *pa = nullptr, *pb = nullptr;
  RC(arena_alloc(c, (size_t)m02, &pa));
  RC(arena_alloc(c, (size_t)m02, &pb));
  if (m02 > (1u << kInvWindowBits)) {
    // the final order is put together inside the first partition pass of the rank inversion (PairsOfFinal)
    const u32 ntile = (m02 + kPartTile - 1) / kPartTile;
    u32 *tc = nullptr;
    RC(arena_alloc(c, (size_t)ntile + 16, &tc));
    {
      PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
      hipLaunchKernelGGL(k_nonuniq_count, dim3(ntile), dim3(kBlock), 0, c->stream, sslot, m02, (u32)kPartTile, tc);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, tc, ntile, (u32 *)nullptr);
      KCHECK();
    }
    PairsOfFinal src; src.sslot = sslot; src.pt = pt; src.base_excl = tc; src.sa12 = sa12;
    RC((inverse_permute_from<PairsOfFinal>(c, src, false, pa, pb, m02, rank12, DC3HIP_PH_RANKS)));
    arena_release(c, mk);
    return E_OK;
  }
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sslot, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL(k_final_assign, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sslot, m02, ck.chunk, counts, pt,
                       sa12, pa);
    KCHECK();
  }
  RC(inverse_permute(c, pa, pb, m02, rank12, DC3HIP_PH_RANKS));
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// prefix-sort + tie-refine ordering (see dc3_kernels.hip.hpp).  Policy:
//   * a strided sample of ~2^20 triples predicts the fraction of samples whose N-bit key image
//     collide; the path is taken when the prediction is below kHybridMaxPredicted,
//   * and abandoned (falling back to the straight 16-byte LSD sort) if the measured fraction turns
//     out above kHybridMaxMeasured.  Correctness never depends on the policy.
// ---------------------------------------------------------------------------------------------
static constexpr u32 kHybridMinSamples = 1u << 22;
// (kHybridMaxPredicted = dc3hip_ctx::hybrid_max_pred = 0.50)
static constexpr double kHybridMaxMeasured = 0.60;
static constexpr double kFullSortMaxPredicted = 0.10;   // whole-level shortcut only for very few predicted ties
static constexpr double kTextSortMaxPredicted = 0.30;   // whole-text shortcut (33-bit images at 2^30 bytes tie ~12 %)
static constexpr double kTextSortMaxBirthday = 0.55;    // ... or more, if the image width alone explains the ties
// Whole-text shortcut: go when few image ties are predicted, or when the predicted ties are no more than what a
// uniformly random text has at this image width (1 - exp(-n / 2^nbits): the 32-bit images of 2^31 positions tie 39 %
// and the tie pass still costs far less than the recursion), which says the text itself is not repetitive.
static bool text_order_worth_trying(double pred, u64 n, u32 nbits) {
  if (pred < kTextSortMaxPredicted) return true;
  const double birthday = 1.0 - exp(-(double)n / ldexp(1.0, (int)nbits));
  return pred < kTextSortMaxBirthday && pred <= 1.25 * birthday + 0.02;
}
// hi = floor(X * mfix / 2^64) in N = min(64 - pbits, kbits) bits; X = key >> shx; see HiMap
static HiMap make_himap(u64 B, u32 kbits, u32 m, u32 pbits = 0) {
  const unsigned __int128 mx = (unsigned __int128)B * B * B - 1;      // largest key
  HiMap hm;
  hm.pbits = pbits ? pbits : bits_of((u64)m + 2);
  hm.nbits = std::min<u32>(64 - hm.pbits, kbits);
  hm.exact = kbits <= hm.nbits ? 1u : 0u;
  hm.shx = kbits > 64 ? kbits - 64 : 0;
  hm.mfix = 0; hm.raw = 0;
  if (!hm.exact) {
    const unsigned __int128 xmax1 = (mx >> hm.shx) + 1;                // > 2^nbits
    const unsigned __int128 num = (((unsigned __int128)1) << (64 + hm.nbits)) - 1;
    hm.mfix = (u64)(num / xmax1);
  }
  return hm;
}

static int count_ties(dc3hip_ctx *c, const Rec8 *h, u32 n, u32 pbits, u32 *counts, const Chunking &ck, u32 *total) {
  hipLaunchKernelGGL(k_tie_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, h, n, ck.chunk, pbits, counts);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  *total = c->h_words[2];
  return E_OK;
}

// sample records whose key image equals another sample's (hash table in the arena; see k_hash_ties)
static int sample_ties(dc3hip_ctx *c, const Rec8 *a, u32 ns, u32 pbits, u32 *ts) {
  u32 slots = 1; while (slots < 2 * ns) slots <<= 1;
  unsigned long long *table = nullptr;
  RC(arena_alloc(c, (size_t)slots, &table));
  HIPC(hipMemsetAsync(table, 0, (size_t)slots * sizeof(unsigned long long), c->stream));
  HIPC(hipMemsetAsync(c->d_words + 2, 0, sizeof(u32), c->stream));
  hipLaunchKernelGGL(k_hash_ties, dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, a, ns, pbits, table, slots - 1,
                     c->d_words + 2);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  *ts = c->h_words[2];
  return E_OK;
}

template <class Sym>
static int predict_tie_fraction(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, HiMap sh, double *pred) {
  const ArenaMark mk = arena_mark(c);
  const u32 stride = std::max<u32>(1, m0 >> 19);
  const u32 ng = (m0 - 1) / stride + 1;      // sampled groups, 2 records each
  const u32 ns = 2 * ng;
  Rec8 *a = nullptr;
  RC(arena_alloc(c, (size_t)ns, &a));
  PhaseScope ps(c, DC3HIP_PH_PACK, ns);
  hipLaunchKernelGGL((k_pack_image<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, sh,
                     stride, ng, a);
  KCHECK();
  u32 ts = 0;
  RC(sample_ties(c, a, ns, sh.pbits, &ts));
  const double fs = (double)ts / (double)ns;
  const double ratio = (double)(m02 - 1) / (double)(ns > 1 ? ns - 1 : 1);
  *pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
  arena_release(c, mk);
  return E_OK;
}

// straight ordering: full-key records (12 bytes when the key fits 64 bits, else 16), LSD over all key bits
template <class Sym, class Rec>
static int order_straight(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12, u32 *rank12,
                          u32 *R, u32 *sslot, u32 *names, int *mode) {
  Rec *recA = nullptr, *recB = nullptr, *sorted = nullptr;
  RC(arena_alloc(c, (size_t)m02, &recA));
  RC(arena_alloc(c, (size_t)m02, &recB));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck;
    radix_plan<Rec>(c, m02, kbits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    if (nb == 512)
      hipLaunchKernelGGL((k_pack_triples_hist<Sym, Rec, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         b, recA, ck.chunk, ck.nchunks, first_table);
    else
      hipLaunchKernelGGL((k_pack_triples_hist<Sym, Rec, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02,
                         b, recA, ck.chunk, ck.nchunks, first_table);
    KCHECK();
  }
  bool by_splitters = false;
  RC(ssort<Rec>(c, recA, recB, m02, kbits, &sorted, &by_splitters));
  if (!by_splitters)
    RC(radix_sort<Rec>(c, recA, recB, m02, 0, kbits, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                       DC3HIP_PH_SORT12_DOWN, first_table));
  AccRec<Rec> acc; acc.s = sorted;
  return name_and_rank<AccRec<Rec>>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode);
}

// Straight ordering with a wider window than the triple (dc3_ssort.hip.hpp): W = floor(96 / sb) symbols (4..7) in
// 16-byte records, ordered by the splitter ordering whose cost does not depend on the key width.  For the levels whose
// triples repeat everywhere (text: the 3-symbol names of level 0 make a level-1 string whose triples are 9 characters):
// their names would send a string of the same length down the recursion; W symbols settle most samples here.
static u32 wide_window_syms(const dc3hip_ctx *c, u32 m02, u64 K) {
  const u32 sb = bits_of(K);
  if (c->no_wide_window || c->no_hybrid || !ssort_applies(c, m02, 96) || sb > 24) return 0;
  return std::min<u32>(7, 96 / sb);             // (the zero tail behind a level's string is 8 symbols)
}
template <class Sym>
struct WideProducer : SsProducer {
  Sym S; u32 sb, W;
  template <int WW> int sample_w(dc3hip_ctx *c, u32 n, u32 Sn, Rec16 *out) {
    hipLaunchKernelGGL((k_ss_sample_window<Sym, WW>), dim3((Sn + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, S, sb, n, Sn, out);
    KCHECK();
    return E_OK;
  }
  template <int WW> int pack_w(dc3hip_ctx *c, Rec16 *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 grid,
                               u32 *cntg, uint16_t *dig) {
    hipLaunchKernelGGL((k_ss_pack_count1<Sym, WW>), dim3(grid), dim3(kSsNT), 0, c->stream, S, sb, a, n, coarse, nb1, tile, cpx, ntiles, tpb, cntg, dig);
    KCHECK();
    return E_OK;
  }
  int sample(dc3hip_ctx *c, u32 n, u32 Sn, void *out) override {
    Rec16 *o = static_cast<Rec16 *>(out);
    switch (W) { case 4: return sample_w<4>(c, n, Sn, o); case 5: return sample_w<5>(c, n, Sn, o); case 6: return sample_w<6>(c, n, Sn, o); default: return sample_w<7>(c, n, Sn, o); }
  }
  int pack_count(dc3hip_ctx *c, void *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb, u32 grid, u32 *cntg,
                 uint16_t *dig) override {
    Rec16 *r = static_cast<Rec16 *>(a);
    switch (W) {
      case 4: return pack_w<4>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      case 5: return pack_w<5>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      case 6: return pack_w<6>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
      default: return pack_w<7>(c, r, n, coarse, nb1, tile, cpx, ntiles, tpb, grid, cntg, dig);
    }
  }
};
template <class Sym>
static int order_wide(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 sb, u32 W, u32 *sa12, u32 *rank12, u32 *R,
                      u32 *sslot, u32 *names, int *mode) {
  Rec16 *recA = nullptr, *recB = nullptr, *sorted = nullptr;
  RC(arena_alloc(c, (size_t)m02, &recA));
  RC(arena_alloc(c, (size_t)m02, &recB));
  bool by_splitters = false;
  SsGeom geo;
  if (ssort_geometry(c, m02, W * sb, &geo)) {
    // the records are made by the kernel that counts the coarse buckets (written once, not read back for the count)
    WideProducer<Sym> prod; prod.S = S; prod.sb = sb; prod.W = W;
    RC(ssort<Rec16>(c, recA, recB, m02, W * sb, &sorted, &by_splitters, &prod));
  } else {
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, m02);
      const dim3 grid((m02 / 2 + kBlock) / kBlock);
      switch (W) {
        case 4: hipLaunchKernelGGL((k_pack_window16<Sym, 4>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        case 5: hipLaunchKernelGGL((k_pack_window16<Sym, 5>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        case 6: hipLaunchKernelGGL((k_pack_window16<Sym, 6>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
        default: hipLaunchKernelGGL((k_pack_window16<Sym, 7>), grid, dim3(kBlock), 0, c->stream, S, m, m02, sb, recA); break;
      }
      KCHECK();
    }
    RC(ssort<Rec16>(c, recA, recB, m02, W * sb, &sorted, &by_splitters));
  }
  if (!by_splitters)
    RC(radix_sort<Rec16>(c, recA, recB, m02, 0, W * sb, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
  AccRec<Rec16> acc; acc.s = sorted;
  return name_and_rank<AccRec<Rec16>>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode);
}

static constexpr u32 kDeepSyms = 2048;                     // symbols compared by the second tie pass of a whole-text order
template <class KM, class Acc>
static int doubling_finish(dc3hip_ctx *c, KM km, Acc acc, u32 n, u32 W, u32 *out_sa, bool *done, bool order_in_place = false);
// key makers of the whole-text order (they know their window; Key3 is a level's triple)
template <class KM> struct IsTextKey { static constexpr bool value = false; };
template <> struct IsTextKey<Key9> { static constexpr bool value = true; };
template <> struct IsTextKey<KeyT> { static constexpr bool value = true; };

// Core of the prefix-sort + tie-refine ordering: `ha` holds nrec packed (image << pbits | pos) records of the
// positions to order; on return (ok) h = records sorted by the full key, f[i] = key differs from predecessor.
template <class KM>
static int hybrid_sort_core(dc3hip_ctx *c, KM km, u32 kbits, const HiMap &hm, Rec8 *ha, Rec8 *hb, u32 nrec,
                            Rec8 **h_out, uint8_t *f, bool *ok, int depth, u32 *emit_sa = nullptr, u32 skip = 0,
                            bool *emitted_distinct = nullptr, u32 *first_table = nullptr, bool whole_text = false,
                            const MsdGeom *mg = nullptr, u64 img_lo = 0, u64 img_span = 0, MsdPass1 *p1 = nullptr,
                            bool *keys_distinct = nullptr, bool slots_ok = false) {
  // slots_ok: the bucket ordering may write its second pass into slots where the arena has (or can commit) the room — a single
  // device's ordering, and since round 6 a rank's ordering of its key range too (its arena is sized for a whole level: 16 bytes
  // per word of a P-th of the words fit)
  // keys_distinct (record form): set when the tie pass settled every tied group and found no two equal keys — the caller
  // then knows that all nrec keys are distinct without counting the flags
  // p1 (only with mg->on): the records of `ha` were NOT written — the pack kernel only counted, pass 1 of the bucket
  // ordering makes them on the fly
  // img_lo / img_span (only with mg == nullptr): the records hold the images of [img_lo, img_lo + img_span) only
  // mg (and mg->on): the records were packed for the bucket ordering — first_table is then the digit table of the TOP
  // image bits in mg's chunking, and the sort runs msd_sort(); should that give up, the LSD passes start from scratch.
  // mg == nullptr (callers that build their records elsewhere): the bucket ordering counts its top digit itself.
  MsdGeom mg_self;
  if (!mg) {
    mg_self = msd_geometry(c, nrec, hm, img_lo, img_span);
    if (mg_self.on) { mg = &mg_self; first_table = nullptr; }
  }
  // whole_text: the records are ALL positions of the text (single device): few repeated windows may be settled here by
  // prefix doubling.  (A rank of the global mode orders only its image range and must not: ranks are global.)
  *ok = false;
  if (emitted_distinct) *emitted_distinct = false;
  if (keys_distinct) *keys_distinct = false;
  Rec8 *h = nullptr;
  bool msd_ok = false;                 // (record form) the bucket ordering delivered, with its same-image bytes in same_rec
  uint8_t *same_rec = nullptr;
  if (emit_sa && emitted_distinct && skip == 0 && !c->no_small_ties && hm.pbits <= 32) {
    // optimistic end of the whole-text order: the last pass writes positions to the SA buffer and 32 image bits to a
    // side array; the tie pass settles the tied groups in place.  Complete unless a key repeats or a group is large.
    // (what the tie pass reads: a "same image as the record before" byte from the bucket ordering, or the 32 image bits
    //  the LSD passes leave when that ordering does not apply or gave up; the image array is only touched in that case)
    u32 *img = nullptr;                      // (4 bytes per record that only the LSD passes write: allocated when they run)
    uint8_t *same = nullptr;
    RC(arena_alloc(c, (size_t)nrec + 16, &same));
    // (one byte per tile of the tie pass, set by the local sort where a tile holds a tie: MsdSplitSink::tilef)
    uint8_t *tilef = nullptr;
    RC(arena_alloc(c, (size_t)nrec / kTieTile + 16, &tilef));
    HIPC(hipMemsetAsync(tilef, 0, (size_t)nrec / kTieTile + 16, c->stream));
    SplitSink sink; sink.sa = emit_sa; sink.img = nullptr; sink.same = same; sink.pbits = hm.pbits; sink.tilef = tilef;
    LastPass lp;
    MsdRedo mredo; bool msd_ok = false;
    if (mg && mg->on) {
      Rec8 *where = ha;
      RC(msd_sort(c, ha, hb, nrec, hm, *mg, first_table, &sink, &h, &mredo, &msd_ok, &where, p1, nullptr, slots_ok));
      if (msd_ok) lp.src = const_cast<u64 *>(mredo.src);               // (non-null = "the order lives in the sink")
      else if (hm.pbits >= 32 && p1) {
        // Beyond 2^31 positions the word itself holds 32 image bits — it only orders by 42 through pass 1's strip — and the LSD
        // passes over plain words would tie 39-58 % of even random windows (then the recursion: 2.9 s at 3.7e9 bytes in the
        // round-6 soak, for one run of a symbol).  The caller takes the 12-byte records instead (build_core).
        return E_OK;                                                   // (*ok stays false, stats.msd_fallbacks says why)
      }
      else { first_table = nullptr; if (p1) RC(p1->repack(c, ha, nrec, &first_table)); }      // from scratch: `ha` in position order
    }
    if (!msd_ok) {
      RC(arena_alloc(c, (size_t)nrec + 16, &img));
      sink.img = img;
      RC(radix_sort<Rec8>(c, ha, hb, nrec, hm.pbits, hm.pbits + hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                          DC3HIP_PH_SORT8_DOWN, first_table, &sink, &lp));
    }
    if (lp.src) {
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
        if (msd_ok)
          hipLaunchKernelGGL((k_tie_resolve_split<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km,
                             SameFlag{same, tilef}, emit_sa, nrec, c->d_words + 10);
        else
          hipLaunchKernelGGL((k_tie_resolve_split<KM, SameImg>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km,
                             SameImg{img}, emit_sa, nrec, c->d_words + 10);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      c->stats.level_tied[depth] = c->h_words[11];
      if ((double)c->h_words[11] > kHybridMaxMeasured * (double)nrec) return E_OK;   // *ok stays false -> straight LSD
      if (c->h_words[10] == 0 && c->h_words[12] == 0) { *emitted_distinct = true; *h_out = nullptr; *ok = true; return E_OK; }
      if constexpr (IsTextKey<KM>::value) {
        // few windows repeat: a second tie pass that compares kDeepSyms symbols instead of the window settles the repeats
        // shorter than that (the compare is lazy: the depth only costs where windows really agree that far) — single
        // device and global mode alike
        if (c->h_words[10] == 0 && c->h_words[12] <= nrec / 4096 + 16 && !c->no_doubling) {
          KM kd = km; kd.deep = kDeepSyms;
          {
            PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
            HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
            if (msd_ok)
              hipLaunchKernelGGL((k_tie_resolve_split<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, kd,
                                 SameFlag{same, tilef}, emit_sa, nrec, c->d_words + 10);
            else
              hipLaunchKernelGGL((k_tie_resolve_split<KM, SameImg>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, kd,
                                 SameImg{img}, emit_sa, nrec, c->d_words + 10);
            KCHECK();
            HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
          }
          HIPC(hipStreamSynchronize(c->stream));
          if (c->h_words[10] == 0 && c->h_words[12] == 0) { *emitted_distinct = true; *h_out = nullptr; *ok = true; return E_OK; }
        }
        // few windows repeat and no group was too large for the tie pass: the positions are in window order in the SA
        // buffer; flag the window changes and let the prefix doubling finish from there (no records needed)
        if (whole_text && c->h_words[10] == 0 && c->h_words[12] <= nrec / 128 && !c->no_doubling && depth == 0) {
          {
            PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
            if (msd_ok)
              hipLaunchKernelGGL((k_split_flags<KM, SameFlag>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, km, SameFlag{same},
                                 (const u32 *)emit_sa, nrec, f);
            else
              hipLaunchKernelGGL((k_split_flags<KM, SameImg>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, km, SameImg{img},
                                 (const u32 *)emit_sa, nrec, f);
            KCHECK();
          }
          // (the doubling reads the order from the very buffer whose tied slots it rewrites: a slot of a tied group always
          //  holds SOME member of that group, whose window — all the binary searches look at — is the group's)
          AccSplit acc; acc.sa = emit_sa; acc.f = f;
          bool finished = false;
          RC((doubling_finish<KM, AccSplit>(c, km, acc, nrec, km.window_syms(), emit_sa, &finished, true)));
          if (finished) { *emitted_distinct = true; *h_out = nullptr; *ok = true; c->stats.level_sorted[0] = 6; return E_OK; }
        }
      }
      // keys repeat (or a large group): the records are needed after all
      if (msd_ok) RC(msd_redo(c, mredo, nrec, &h));
      else RC(radix_redo_last(c, lp, nrec, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT8_DOWN));
    }
    emit_sa = nullptr;                       // from here on: the record path, positions are emitted by the caller
  } else {
    if (mg && mg->on) {
      MsdRedo mredo; Rec8 *where = ha;
      if (!c->no_small_ties) RC(arena_alloc(c, (size_t)nrec + 16, &same_rec));
      RC(msd_sort(c, ha, hb, nrec, hm, *mg, first_table, nullptr, &h, &mredo, &msd_ok, &where, p1, same_rec, slots_ok));
      if (!msd_ok) { first_table = nullptr; if (p1) RC(p1->repack(c, ha, nrec, &first_table)); }
    }
    if (!msd_ok)
      RC(radix_sort<Rec8>(c, ha, hb, nrec, hm.pbits, hm.pbits + hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                          DC3HIP_PH_SORT8_DOWN, first_table));
  }
  const Chunking ck = make_chunks(c, nrec, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 tied = 0;
  bool general = false;
  HIPC(hipMemsetAsync(f, 1, (size_t)nrec, c->stream));
  if (!c->no_small_ties) {
    // one in-place pass counts the tied records and settles every tied group of at most kTieSmallMax members
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
      HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
      if (msd_ok && same_rec)
        hipLaunchKernelGGL((k_tie_resolve<KM, SameFlag>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km, SameFlag{same_rec}, h, nrec,
                           hm.pbits, f, c->d_words + 10, emit_sa, skip);
      else
        hipLaunchKernelGGL((k_tie_resolve<KM, SameRec>), dim3(grid_for(c, nrec / 4 + 1)), dim3(kBlock), 0, c->stream, km, SameRec{h, hm.pbits}, h, nrec,
                           hm.pbits, f, c->d_words + 10, emit_sa, skip);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    tied = c->h_words[11];
    c->stats.level_tied[depth] = tied;
    if ((double)tied > kHybridMaxMeasured * (double)nrec) return E_OK;   // *ok stays false -> straight LSD
    general = c->h_words[10] != 0;       // some group is larger: redo the ties with the general path
    if (!general && emit_sa && emitted_distinct && c->h_words[12] == 0) *emitted_distinct = true;
    if (!general && keys_distinct && c->h_words[12] == 0) *keys_distinct = true;
  }
  if (general || c->no_small_ties) {
    PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
    RC(count_ties(c, h, nrec, hm.pbits, counts, ck, &tied));
    c->stats.level_tied[depth] = tied;
    if ((double)tied > kHybridMaxMeasured * (double)nrec) return E_OK;
    general = tied > 0;
  }
  if (general) {
    // the tied subset is re-sorted as 16-byte records (2 x 16 B + index + radix tables): when the arena cannot hold
    // that on top of what the caller holds, give the ordering up (*ok stays false -> the caller's next ordering
    // runs instead); arena_requirement() only bounds the straight ordering
    if (!arena_grow_in_use(c, c->arena_off + (size_t)tied * 36 + (32u << 20))) return E_OK;      // (a reserved arena raises its limit where it lies)
    const ArenaMark mk_general = arena_mark(c);     // the tied subset is dead after the write-back: released there
    Rec16 *sa = nullptr, *sb = nullptr, *ss = nullptr;
    u32 *tiedidx = nullptr;
    RC(arena_alloc(c, (size_t)tied, &sa));
    RC(arena_alloc(c, (size_t)tied, &sb));
    RC(arena_alloc(c, (size_t)tied, &tiedidx));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL((k_tie_compact<KM>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, h, nrec, ck.chunk,
                         hm.pbits, counts, sa, tiedidx);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, sa, sb, tied, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN,
                         DC3HIP_PH_SORT12_DOWN));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL(k_tie_writeback, dim3(grid_for(c, tied)), dim3(kBlock), 0, c->stream, ss, tiedidx, tied, h, f);
      KCHECK();
    }
    arena_release(c, mk_general);
  }
  *h_out = h;
  *ok = true;
  return E_OK;
}

// Tie refinement of records sorted by their 63-bit key prefix (see order_hybrid12): f[i] = full key differs from the
// predecessor's, tied groups ordered by the full key.  *ok = false: too many ties, or no room for the general path.
template <class KM>
static int hybrid12_refine(dc3hip_ctx *c, KM km, u32 kbits, Rec12 *h, u32 n, uint8_t *f, bool *ok, int depth,
                           u32 *emit_sa = nullptr, bool *distinct = nullptr, bool *deep_flags = nullptr) {
  // *deep_flags: on return f[] (and the order inside tied groups) reflects equality over kDeepSyms symbols, not over
  // the window — whoever continues from f[] (doubling_finish) must compare at the same depth
  *ok = false;
  if (distinct) *distinct = false;
  if (deep_flags) *deep_flags = false;
  bool deep_ran = false;
  HIPC(hipMemsetAsync(f, 1, (size_t)n, c->stream));
  u32 tied = 0;
  bool general = false;
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, n);
    HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_tie_resolve12<KM>), dim3(grid_for(c, n / 4 + 1)), dim3(kBlock), 0, c->stream, km, h, n, f,
                       c->d_words + 10, emit_sa);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  tied = c->h_words[11];
  c->stats.level_tied[depth] = tied;
  general = c->h_words[10] != 0;
  if constexpr (IsTextKey<KM>::value) {
    // whole-text order: a few windows agree completely -> the tie pass once more, comparing kDeepSyms symbols (see
    // hybrid_sort_core); settles the repeats shorter than that
    if (emit_sa && !general && c->h_words[12] > 0 && c->h_words[12] <= n / 4096 + 16 && !c->no_doubling) {
      KM kd = km; kd.deep = kDeepSyms;
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, n);
        HIPC(hipMemsetAsync(f, 1, (size_t)n, c->stream));
        HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
        hipLaunchKernelGGL((k_tie_resolve12<KM>), dim3(grid_for(c, n / 4 + 1)), dim3(kBlock), 0, c->stream, kd, h, n, f,
                           c->d_words + 10, emit_sa);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      tied = c->h_words[11];
      general = c->h_words[10] != 0;
      deep_ran = true;
    }
  }
  // (the general path below re-sorts every tied record by the window's full key and rewrites f[] from it)
  if (deep_flags) *deep_flags = deep_ran && !general;
  // no group overflowed and no full key repeats: the positions the tie pass wrote to emit_sa are the sorted order
  if (distinct) *distinct = emit_sa && !general && c->h_words[12] == 0;
  if ((double)tied > std::max(kHybridMaxMeasured, c->hybrid12_max_pred + 0.1) * (double)n) return E_OK;
  if (general) {
    // some tied group is larger than kTieSmallMax: re-sort ALL tied records by the full key (the small groups that were
    // already settled are re-done consistently)
    if (!arena_grow_in_use(c, c->arena_off + (size_t)tied * 36 + (32u << 20))) return E_OK;      // (a reserved arena raises its limit where it lies)
    const ArenaMark mk_general = arena_mark(c);     // the tied subset is dead after the write-back: released there
    const Chunking ck = make_chunks(c, n, kBlock);
    u32 *counts = nullptr, *tiedidx = nullptr;
    Rec16 *sa = nullptr, *sb = nullptr, *ss = nullptr;
    RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, n);
      hipLaunchKernelGGL(k_tie_count12, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, h, n, ck.chunk, counts);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    tied = c->h_words[2];
    RC(arena_alloc(c, (size_t)tied, &sa));
    RC(arena_alloc(c, (size_t)tied, &sb));
    RC(arena_alloc(c, (size_t)tied, &tiedidx));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL((k_tie_compact12<KM>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, h, n, ck.chunk, counts,
                         sa, tiedidx);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, sa, sb, tied, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, tied);
      hipLaunchKernelGGL(k_tie_writeback12, dim3(grid_for(c, tied)), dim3(kBlock), 0, c->stream, ss, tiedidx, tied, h, f);
      KCHECK();
    }
    arena_release(c, mk_general);
  }
  *ok = true;
  return E_OK;
}

// Prefix sort + tie refinement on 12-byte records (kernels: "Prefix sort ... on 12-byte records" in dc3_order.hip.hpp):
// for keys wider than 64 bits.  *ok = false: too many ties (predicted or measured), nothing was produced.
template <class Sym>
static int order_hybrid12(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12, u32 *rank12, u32 *R,
                          u32 *sslot, u32 *names, int *mode, bool *ok, int depth) {
  *ok = false;
  const ArenaMark mk = arena_mark(c);
  Key3<Sym> km; km.S = S; km.B = b;
  // predicted fraction of samples whose 63-bit prefix collides with another sample's
  {
    const u32 stride = std::max<u32>(1, m0 >> 19);
    const u32 ng = (m0 - 1) / stride + 1, ns = 2 * ng;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    u32 ts = 0;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, ns);
      hipLaunchKernelGGL((k_pack_image12_sample<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m02, b, kbits,
                         stride, ng, a);
      KCHECK();
    }
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    const double ratio = (double)(m02 - 1) / (double)(ns > 1 ? ns - 1 : 1);
    const double pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
    c->stats.level_tie_pred[depth] = pred;
    arena_release(c, mk);
    if (!(pred < c->hybrid12_max_pred)) return E_OK;
  }
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)m02, &ha));
  RC(arena_alloc(c, (size_t)m02, &hb));
  RC(arena_alloc(c, (size_t)m02 + 16, &f));
  u32 *first_table = nullptr;
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck;
    radix_plan<Rec12>(c, m02, kImg12Bits, &nb, &ck);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
    if (nb == 512)
      hipLaunchKernelGGL((k_pack_image12_hist<Sym, 512>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, kbits,
                         ha, ck.chunk, ck.nchunks, first_table);
    else
      hipLaunchKernelGGL((k_pack_image12_hist<Sym, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, kbits,
                         ha, ck.chunk, ck.nchunks, first_table);
    KCHECK();
  }
  RC(radix_sort<Rec12>(c, ha, hb, m02, 0, kImg12Bits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN,
                       first_table));
  {
    bool refined = false;
    RC((hybrid12_refine<Key3<Sym>>(c, km, kbits, h, m02, f, &refined, depth)));
    if (!refined) { arena_release(c, mk); return E_OK; }
  }
  c->stats.level_sorted[depth] = 2;
  AccHyb12 acc; acc.h = h; acc.f = f;
  RC(name_and_rank<AccHyb12>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode));
  *ok = true;
  arena_release(c, mk);
  return E_OK;
}

// tie-rate predictor over all positions (stride sample) for the whole-text shortcut of level 0
template <class KM>
static int predict_tie_fraction_pos(dc3hip_ctx *c, KM km, u32 n, const HiMap &hm, double *pred) {
  const ArenaMark mk = arena_mark(c);
  const u32 stride = std::max<u32>(1, n >> 20);
  const u32 ns = (n - 1) / stride + 1;
  Rec8 *a = nullptr;
  RC(arena_alloc(c, (size_t)ns, &a));
  PhaseScope ps(c, DC3HIP_PH_PACK, ns);
  hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
  KCHECK();
  u32 ts = 0;
  RC(sample_ties(c, a, ns, hm.pbits, &ts));
  const double fs = (double)ts / (double)ns;
  const double ratio = (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1);
  *pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
  arena_release(c, mk);
  return E_OK;
}

// The same prediction from the images alone (hm.nbits <= 63 bits, no position beside them): what the whole-text order beyond
// 2^31 positions uses — its words carry a 32-bit position and 32 image bits, but pass 1 of the bucket ordering strips the
// bucket's d1 bits off an image that much wider (MsdPass1Keys::strip), and the ties that matter are those of the wider image.
template <class KM>
static int predict_tie_fraction_img(dc3hip_ctx *c, KM km, u32 n, const HiMap &hm, double *pred) {
  const ArenaMark mk = arena_mark(c);
  const u32 stride = std::max<u32>(1, n >> 20);
  const u32 ns = (n - 1) / stride + 1;
  Rec8 *a = nullptr;
  RC(arena_alloc(c, (size_t)ns, &a));
  PhaseScope ps(c, DC3HIP_PH_PACK, ns);
  hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
  KCHECK();
  u32 ts = 0;
  RC(sample_ties(c, a, ns, 1u, &ts));
  const double fs = (double)ts / (double)ns;
  const double ratio = (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1);
  *pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, ratio);
  arena_release(c, mk);
  return E_OK;
}

template <class Sym>
static int order_hybrid(dc3hip_ctx *c, Sym S, u32 m, u32 m0, u32 m02, u32 b, u32 kbits, u32 *sa12,
                        u32 *rank12, u32 *R, u32 *sslot, u32 *names, int *mode, bool *ok, int depth) {
  *ok = false;
  const HiMap hm = make_himap((u64)b, kbits, m);
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)m02, &ha));
  RC(arena_alloc(c, (size_t)m02, &hb));
  RC(arena_alloc(c, (size_t)m02 + 16, &f));
  u32 *first_table = nullptr;
  const MsdGeom mg = msd_geometry(c, m02, hm);
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, m02);
    int nb = 0; Chunking ck; u32 hshift = 0;
    pack_plan(c, m02, hm, &mg, &nb, &ck, &hshift);
    RC(arena_alloc(c, (size_t)nb * ck.nchunks, &first_table));
#define K_(NB) (k_pack_image_hist<Sym, NB>)
    DC3_PACK_LAUNCH(K_, S, m, m0, m02, b, hm, ha, ck.chunk, ck.nchunks, first_table, hshift);
#undef K_
  }
  bool sorted_ok = false;
  Key3<Sym> km; km.S = S; km.B = b;
  RC((hybrid_sort_core<Key3<Sym>>(c, km, kbits, hm, ha, hb, m02, &h, f, &sorted_ok, depth, nullptr, 0, nullptr,
                                  first_table, false, &mg)));
  if (!sorted_ok) return E_OK;
  c->stats.level_sorted[depth] = 2;
  AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
  RC(name_and_rank<AccHyb>(c, acc, m02, m0, sa12, rank12, R, sslot, names, mode));
  *ok = true;
  return E_OK;
}

// Whole-level shortcut for high-entropy levels: order ALL m positions (plus the dummy sample) by their triple.
//   state 1: every triple distinct -> the result is the suffix array of the level (suffixes differ within 3
//            symbols): sampling, tuples and the merge (lib.rs:62-192) are skipped altogether;
//   state 2: duplicates exist -> the samples are filtered out of the sorted order (spos/snf), so the usual
//            naming continues from there and the sort is not repeated;
//   state 0: too many collisions in the key image, nothing was produced.
// spos/snf (m02 entries each) must be allocated by the caller below this function's arena mark.
// Packs the records of all positions; *first_table != nullptr on return when the kernel also produced the digit
// table of the first radix pass (whole text: k_pack_image_text).
// store = false (only with mg->on): count only — pass 1 of the bucket ordering makes the records on the fly (MsdPass1Keys).
template <class KM>
static int launch_pack_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table,
                           const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  if (!store) {
    hipLaunchKernelGGL((k_pack_image_all_hist<KM, 1024, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out, ck.chunk,
                       ck.nchunks, table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_all_hist<KM, NB>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
template <>
int launch_pack_all<Key9>(dc3hip_ctx *c, Key9 km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  if (!store && hm.raw && hshift < hm.nbits && hm.nbits - hshift <= 10u) {
    // (the raw image is KeyBits' with 8 bits per symbol and the text as the stream: dc3_order.hip.hpp)
    hipLaunchKernelGGL(k_count_image_bits<7>, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, KeyBits{km.S.t, 8u}, nrec, hm, ck.chunk, ck.nchunks,
                       table, hshift);
    KCHECK();
  } else if (!store) {
    hipLaunchKernelGGL((k_pack_image_text<1024, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out, ck.chunk, ck.nchunks,
                       table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_text<NB>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
template <>
int launch_pack_all<KeyBits>(dc3hip_ctx *c, KeyBits km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  if (!store && hshift < hm.nbits && 7u * km.lg + 7u + (hm.nbits - hshift) <= 64u && hm.nbits - hshift <= 10u) {      // (at least one digit bit: the kernel shifts by 64 - that)
    hipLaunchKernelGGL(k_count_image_bits<8>, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, ck.chunk, ck.nchunks, table, hshift);
    KCHECK();
  } else if (!store) {
    hipLaunchKernelGGL((k_pack_image_all_hist<KeyBits, 1024, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, out, ck.chunk,
                       ck.nchunks, table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_all_hist<KeyBits, NB>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
static u64 keyt_p1(const KeyT &km) {
  u64 P1 = 1;
  for (u32 i = 0; i + 1 < km.J; i++) P1 *= km.sigma;
  return P1;
}
template <>
int launch_pack_all<KeyT>(dc3hip_ctx *c, KeyT km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg, bool store) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  const u64 P1 = keyt_p1(km);
  if (!store) {
    hipLaunchKernelGGL((k_pack_image_textT<1024, false, false>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, P1, (void *)out,
                       ck.chunk, ck.nchunks, table, hshift);
    KCHECK();
  } else {
#define K_(NB) (k_pack_image_textT<NB, false>)
    DC3_PACK_LAUNCH(K_, km, nrec, hm, P1, (void *)out, ck.chunk, ck.nchunks, table, hshift);
#undef K_
  }
  *first_table = table;
  return E_OK;
}
// KeyT and the map of its image (see the struct): J = fewest symbols whose base-sigma value exceeds the image width by
// two bits, within 63 bits, the key's 3L symbols and kKeyTMaxImageSyms.  false = no such J (the caller skips the path).
static bool make_keyt(SymU8 S, u32 sigma, u32 L, u64 BL, u32 n, KeyT *km, HiMap *hm, u32 image_bits = 0) {
  if (sigma < 2) return false;
  hm->pbits = image_bits ? 64 - image_bits : bits_of((u64)n - 1);        // positions 0..n-1 only
  hm->nbits = 64 - hm->pbits;
  hm->shx = 0; hm->exact = 0; hm->raw = 0;
  u32 J = 1; u64 SJ = sigma;                                             // sigma^J
  const u32 jmax = std::min<u32>(3 * L, kKeyTMaxImageSyms);
  while (J < jmax && (SJ >> std::min<u32>(hm->nbits + 2, 62)) == 0 && SJ * sigma < (1ull << 63)) { SJ *= sigma; J++; }
  if ((SJ >> hm->nbits) == 0) return false;                              // the image must be a proper scaling
  hm->mfix = (u64)(((((unsigned __int128)1) << (64 + hm->nbits)) - 1) / SJ);
  km->S = S; km->B = sigma + 1; km->BL = (u32)BL; km->L = L; km->sigma = sigma; km->J = J;
  km->lg = (sigma & (sigma - 1)) == 0 ? bits_of((u64)sigma - 1) : 0;
  return true;
}

// KeyT under the bucket ordering: the pack kernel writes the images alone, from a map d1 bits wider than the words' (a
// second KeyT with more symbols), and pass 1 strips the bucket's own bits while it partitions (KeyImg) — the tie pass
// then meets 2^-d1 of the ties, as it does for the key makers that compute their image inside pass 1.
struct MsdPass1Img : MsdPass1 {
  KeyImg ki; HiMap hm;                       // the wider map (hm.pbits = position bits - d1)
  KeyT km_plain; HiMap hm_plain;             // the words' own key maker and layout (repack)
  int repack(dc3hip_ctx *c, Rec8 *out, u32 nrec, u32 **first_table) override {
    PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
    return launch_pack_all<KeyT>(c, km_plain, nrec, hm_plain, out, first_table, nullptr, true);
  }
  int launch(dc3hip_ctx *c, u64 *out, u32 n, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KeyImg, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      attr_set[c->device & 15] = true;
    }
    hipLaunchKernelGGL((k_msd_part_keys<KeyImg, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, ki, hm, 0ull, out, n, base,
                       sh1, g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon);
    KCHECK();
    return E_OK;
  }
};
static int launch_pack_images_keyt(dc3hip_ctx *c, KeyT km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table, const MsdGeom *mg) {
  int nb = 0; Chunking ck; u32 hshift = 0;
  pack_plan(c, nrec, hm, mg, &nb, &ck, &hshift);
  u32 *table = nullptr;
  RC(arena_alloc(c, (size_t)nb * ck.nchunks, &table));
  hipLaunchKernelGGL((k_pack_image_textT<1024, false, true, true>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, km, nrec, hm, keyt_p1(km), (void *)out,
                     ck.chunk, ck.nchunks, table, hshift);
  KCHECK();
  *first_table = table;
  return E_OK;
}
template <class KM> static u64 pass1_p1(const KM &) { return 0; }
template <> u64 pass1_p1<KeyT>(const KeyT &km) { return keyt_p1(km); }
// The records of all m (+dummy) positions are in key order behind accessor `acc` (pos, neq): all keys distinct -> the
// order is the suffix array (*state = 1; out_sa / out_rank written); else, with spos/snf given, the samples are
// filtered out with their full names (*state = 2); else *state stays 0.
template <class Acc, class Map>
static int finish_position_order(dc3hip_ctx *c, Acc acc, Map mp, u32 nrec, u32 m, u32 dummy, u32 *out_sa, u32 *out_rank,
                                 u32 *spos, u32 *snf, int *state, bool known_distinct = false) {
  // known_distinct: the tie pass already established that no two keys are equal (no counting pass, no host round trip)
  const Chunking ck = make_chunks(c, nrec, kBlock);
  u32 *counts = nullptr, *scounts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &scounts));
  if (known_distinct) {
    c->h_words[0] = nrec;
  } else {
    PhaseScope ps(c, DC3HIP_PH_NAMING, nrec);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, nrec, ck.chunk,
                       counts, c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
  }
  if (c->h_words[0] == nrec) {          // every key distinct: the sorted order is the suffix array
    Rec8 *pa = nullptr, *pb = nullptr;
    if (out_rank) {
      RC(arena_alloc(c, (size_t)m, &pa));
      RC(arena_alloc(c, (size_t)m, &pb));
    }
    if (out_rank && m > (1u << kInvWindowBits)) {
      // the pairs (pos_k, k + 1) are made by the first partition pass itself, which also leaves out_sa[k] = pos_k
      PairsOfOrder<Acc> po; po.acc = acc; po.skip = dummy; po.out_sa = out_sa;
      RC((inverse_permute_from<PairsOfOrder<Acc>>(c, po, false, pa, pb, m, out_rank, DC3HIP_PH_RANKS)));
    } else {
      {
        PhaseScope ps(c, DC3HIP_PH_RANKS, m);
        hipLaunchKernelGGL((k_emit_sorted<Acc>), dim3(grid_for(c, m)), dim3(kBlock), 0, c->stream, acc, m, dummy,
                           out_sa, pa);
        KCHECK();
      }
      if (out_rank) RC(inverse_permute(c, pa, pb, m, out_rank, DC3HIP_PH_RANKS));
    }
    *state = 1;
  } else if (spos && snf) {             // keep the sort: filter the samples with their full names
    PhaseScope ps(c, DC3HIP_PH_NAMING, nrec);
    hipLaunchKernelGGL((k_filter_count<Acc, Map>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, mp, nrec,
                       ck.chunk, scounts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, scounts, ck.nchunks, (u32 *)nullptr);
    KCHECK();
    hipLaunchKernelGGL((k_filter_write<Acc, Map>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, mp, nrec,
                       ck.chunk, counts, scounts, spos, snf);
    KCHECK();
    *state = 2;
  }
  return E_OK;
}

// Few windows repeat (dc3_doubling.hip.hpp): refine the tied positions alone by prefix doubling and finish the suffix
// array at level 0.  acc = the n positions in window order with their "differs from predecessor" flags; W = symbols per
// window.  *done = false (nothing lost: out_sa is scratch until a caller declares it the result) when too many
// positions are tied, the arena is short, or the rounds do not converge.
static constexpr u32 kDoublingMaxTied = 4u << 20;          // records; and at most 1/64 of the positions
template <class KM, class Acc>
static int doubling_finish(dc3hip_ctx *c, KM km, Acc acc, u32 n, u32 W, u32 *out_sa, bool *done, bool order_in_place) {
  *done = false;
  if (c->no_doubling || !out_sa || n < 2) return E_OK;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, n, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, n);
    hipLaunchKernelGGL((k_dbl_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, n, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 2);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 t = c->h_words[2];
  if (t == 0 || t > kDoublingMaxTied || t > n / 64 || !arena_grow_in_use(c, c->arena_off + (size_t)t * 128 + (32u << 20))) {
    arena_release(c, mk);
    return E_OK;
  }
  u32 *slot = nullptr, *pos = nullptr, *start = nullptr, *gid = nullptr, *mapidx = nullptr, *map_pos = nullptr, *map_val = nullptr;
  Rec8 *pa = nullptr, *pb = nullptr, *ps_sorted = nullptr;
  Rec16 *act = nullptr, *tmp = nullptr, *next = nullptr;
  RC(arena_alloc(c, (size_t)t + 16, &slot)); RC(arena_alloc(c, (size_t)t + 16, &pos)); RC(arena_alloc(c, (size_t)t + 16, &start));
  RC(arena_alloc(c, (size_t)t + 16, &gid)); RC(arena_alloc(c, (size_t)t + 16, &mapidx));
  RC(arena_alloc(c, (size_t)t + 16, &map_pos)); RC(arena_alloc(c, (size_t)t + 16, &map_val));
  RC(arena_alloc(c, (size_t)t + 16, &pa)); RC(arena_alloc(c, (size_t)t + 16, &pb));
  RC(arena_alloc(c, (size_t)t + 16, &act)); RC(arena_alloc(c, (size_t)t + 16, &tmp)); RC(arena_alloc(c, (size_t)t + 16, &next));
  u32 *sums = nullptr, *carry = nullptr;                      // per-tile summaries of the regrouping
  RC(arena_alloc(c, (size_t)3 * (t / kDblTile + 2), &sums)); RC(arena_alloc(c, (size_t)3 * (t / kDblTile + 2), &carry));
  const u32 kb = bits_of((u64)n);                            // ranks + 1 and slots are below 2^kb
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, t);
    // the order as it stands (final for every untied position) — unless acc already reads it from out_sa
    if (!order_in_place) {
      hipLaunchKernelGGL((k_emit_sorted<Acc>), dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, acc, n, 0u, out_sa, (Rec8 *)nullptr);
      KCHECK();
    }
    hipLaunchKernelGGL((k_dbl_collect<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, n, ck.chunk, (const u32 *)counts, slot,
                       pos, start);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_gid, dim3(1), dim3(kBlock), 0, c->stream, (const u32 *)slot, (const u32 *)start, t, gid);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_map_pairs, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const u32 *)pos, t, pa);
    KCHECK();
  }
  RC(radix_sort<Rec8>(c, pa, pb, t, 32, 32 + bits_of((u64)n - 1), &ps_sorted, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
  {
    PhaseScope ps(c, DC3HIP_PH_TIES, t);
    hipLaunchKernelGGL(k_dbl_map_build, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const Rec8 *)ps_sorted, t, (const u32 *)gid,
                       map_pos, map_val, mapidx);
    KCHECK();
    hipLaunchKernelGGL(k_dbl_init, dim3(grid_for(c, t)), dim3(kBlock), 0, c->stream, (const u32 *)pos, (const u32 *)gid,
                       (const u32 *)mapidx, t, act);
    KCHECK();
  }
  u32 a = t;
  u64 d = W;
  int rounds = 0;
  Rec16 *X = act, *Y = tmp, *Z = next;                       // X: this round's records, Y: sort scratch, Z: next round's records
  for (; a > 0 && rounds < 40 && d < (u64)n * 2; rounds++, d *= 2) {
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, a);
      hipLaunchKernelGGL((k_dbl_key<KM, Acc>), dim3(grid_for(c, a)), dim3(kBlock), 0, c->stream, km, acc, n,
                         (u32)std::min<u64>(d, 0xffffffffull), (const u32 *)map_pos, (const u32 *)map_val, t, X, a);
      KCHECK();
    }
    // by (group, rank of p + d): LSD, the rank first
    Rec16 *s1 = nullptr, *s2 = nullptr;
    RC(radix_sort<Rec16>(c, X, Y, a, 0, kb, &s1, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
    RC(radix_sort<Rec16>(c, s1, s1 == X ? Y : X, a, 32, 32 + kb, &s2, DC3HIP_PH_TIES, DC3HIP_PH_TIES, DC3HIP_PH_TIES));
    {
      PhaseScope ps(c, DC3HIP_PH_TIES, a);
      const u32 ntiles = (a + kDblTile - 1) / kDblTile;
      hipLaunchKernelGGL((k_dbl_regroup<false>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const Rec16 *)s2, a, sums, (const u32 *)nullptr,
                         (u32 *)nullptr, (u32 *)nullptr, (Rec16 *)nullptr);
      KCHECK();
      hipLaunchKernelGGL(k_dbl_regroup_scan, dim3(1), dim3(kBlock), 0, c->stream, (const u32 *)sums, ntiles, carry, c->d_words + 2);
      KCHECK();
      hipLaunchKernelGGL((k_dbl_regroup<true>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const Rec16 *)s2, a, (u32 *)nullptr,
                         (const u32 *)carry, out_sa, map_val, Z);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 2, c->d_words + 2, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    a = c->h_words[2];
    Rec16 *nx = Z; Z = Y; Y = X; X = nx;
  }
  c->stats.level_tied[0] = t;
  c->stats.level_kept[0] = rounds;                           // (rounds of prefix doubling over the tied positions)
  arena_release(c, mk);
  *done = a == 0;
  return E_OK;
}

template <class KM, class Map>
static int order_all_positions(dc3hip_ctx *c, KM km, Map mp, u32 m, u32 kbits, const HiMap &hm, u32 dummy, u32 *out_sa,
                               u32 *out_rank, u32 *spos, u32 *snf, int *state, int depth) {
  *state = 0;
  const ArenaMark mk = arena_mark(c);
  const u32 nrec = m + dummy;              // dummy = 1: include the dummy sample at position m (lib.rs:61-64)
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr;
  uint8_t *f = nullptr;
  RC(arena_alloc(c, (size_t)nrec, &ha));
  RC(arena_alloc(c, (size_t)nrec, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  u32 *first_table = nullptr;
  const MsdGeom mg = msd_geometry(c, nrec, hm);
  // bucket ordering: the pack kernel only counts, partition pass 1 makes the records on the fly (8 bytes per position
  // neither written nor read back)
  // — opt-in (DC3HIP_PACK_FUSE=1), Key9 only.  Measured at 1 GiB: bytes: pack 3.3 -> 1.7 ms counting only, pass 1
  // 3.7 -> 4.4 ms (it becomes VALU-bound: 9 bytes moved per word instead of 16, but the key arithmetic on top of the
  // ranking), build 20.4 -> 19.5 ms; DNA (KeyT): the rolling image inside the partition pass costs more than the bytes
  // save, 22.8 -> 27.9 ms.  A 5 % gain on one input class against a second variant of the dominant kernel: off by default.
  // (byte windows at level 0, name triples at the levels below; the small-alphabet windows KeyT keep a pack kernel
  //  that writes: their rolling image inside the partition pass was measured slower, 22.8 -> 27.9 ms at 1 GiB DNA)
  constexpr bool kKeyT = std::is_same<KM, KeyT>::value;
  constexpr bool kFusable = std::is_same<KM, Key9>::value || std::is_same<KM, Key3<SymU32>>::value;
  const bool fuse = mg.on && kFusable;
  MsdPass1Keys<KM> p1; p1.km = km; p1.hm = hm; p1.P1 = pass1_p1<KM>(km);
  MsdGeom mgx = mg;
  if constexpr (kFusable) {
    // ... and since the words are made inside pass 1, they can come from an image d1 bits wider than a word has room
    // for (k_msd_part_keys<.., true>): the tie pass then finds next to nothing tied
    if (fuse && !c->no_pack_strip && hm.pbits >= 23 && !hm.exact && kbits >= hm.nbits + mg.d1) {
      u64 limb = 0;                                  // base of the key's three limbs (make_himap's B)
      if constexpr (std::is_same<KM, Key9>::value) limb = km.B3; else limb = km.B;
      p1.strip = true; p1.hm_plain = hm;
      p1.hm = make_himap(limb, kbits, m, hm.pbits - mg.d1);
      p1.hm.raw = hm.raw;
      mgx.ebits = p1.hm.nbits;                       // (= hm.nbits + d1: the shifts of passes 2 and 3 follow from it)
    }
  }
  MsdPass1Img pimg;
  MsdPass1Keys<KeyBits> pbits;
  MsdPass1 *pass1 = fuse ? &p1 : nullptr;
  bool packed = false;
  if constexpr (kKeyT) {
    // power-of-two alphabets (DNA): the image comes off a bit-packed copy of the text, inside pass 1 (KeyBits) — no image
    // array is written or read back; the pack kernel only counts the top digit
    if (mg.on && !c->no_pack_strip && dummy == 0 && hm.pbits >= 23 && km.lg != 0 && hm.nbits + mg.d1 + 7 + 3 * km.lg <= 64 &&
        (u64)3 * km.L * km.lg >= hm.nbits + mg.d1) {        // (the image spans no more symbols than the window the ties are compared by)
      // (+ 10 groups of zero digits: an image is read with ONE 8-byte load from the byte its first bit lies in, i.e. up to 8 bytes
      //  behind the last symbol's — with lg = 1 a group is one byte.  Two groups were not enough: the soak of round 6 found the
      //  last few suffixes of binary and DNA texts misplaced whenever the arena behind the stream was not zero.)
      const u32 groups = (nrec + 7) / 8 + 10;
      uint8_t *bits = nullptr;
      RC(arena_alloc(c, (size_t)groups * km.lg + 64, &bits));
      HiMap hw; hw.mfix = 0; hw.shx = 0; hw.exact = 0; hw.raw = 0; hw.pbits = hm.pbits - mg.d1; hw.nbits = hm.nbits + mg.d1;
      HiMap hpl = hw; hpl.pbits = hm.pbits; hpl.nbits = hm.nbits;
      pbits.km.bits = bits; pbits.km.lg = km.lg; pbits.hm = hw; pbits.P1 = 0; pbits.strip = true; pbits.hm_plain = hpl;
      mgx.ebits = hw.nbits;
      PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
      hipLaunchKernelGGL(k_pack_bits, dim3(grid_for(c, groups)), dim3(kBlock), 0, c->stream, km.S, m, km.lg, groups, bits);
      KCHECK();
      RC(launch_pack_all<KeyBits>(c, pbits.km, nrec, hw, ha, &first_table, &mgx, false));
      pass1 = &pbits; packed = true;
    }
  }
  if constexpr (kKeyT) {
    // other small alphabets: the image stays in a pack kernel (inside pass 1 it was measured slower), which writes it d1 bits wider
    KeyT kw; HiMap hw;
    if (!packed && mg.on && !c->no_pack_strip && dummy == 0 && hm.pbits >= 23 && hm.nbits + mg.d1 <= 62 &&
        make_keyt(km.S, km.sigma, km.L, km.BL, m, &kw, &hw, hm.nbits + mg.d1)) {
      pimg.ki.img = reinterpret_cast<const u64 *>(ha); pimg.hm = hw; pimg.km_plain = km; pimg.hm_plain = hm;
      mgx.ebits = hw.nbits;
      PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
      RC(launch_pack_images_keyt(c, kw, nrec, hw, ha, &first_table, &mgx));
      pass1 = &pimg; packed = true;
    }
  }
  if (!packed) {
    PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
    RC(launch_pack_all<KM>(c, km, nrec, p1.strip ? p1.hm : hm, ha, &first_table, &mgx, !fuse));
  }
  bool sorted_ok = false, distinct = false, all_distinct = false;
  RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &sorted_ok, depth, out_rank ? nullptr : out_sa, dummy,
                           &distinct, first_table, std::is_same<Map, MapText>::value && dummy == 0 && !out_rank, &mgx, 0, 0,
                           pass1, &all_distinct, true)));
  if (sorted_ok && distinct) {
    *state = 1;                            // the tie pass already wrote the suffix array
  } else if (sorted_ok) {
    AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
    bool finished = false;
    if constexpr (std::is_same<Map, MapText>::value) {       // whole text: few repeated windows are settled right here
      if (dummy == 0 && out_sa && !out_rank) RC((doubling_finish<KM, AccHyb>(c, km, acc, nrec, km.window_syms(), out_sa, &finished)));
    }
    if (finished) { *state = 1; c->stats.level_sorted[0] = 6; }
    else RC((finish_position_order<AccHyb, Map>(c, acc, mp, nrec, m, dummy, out_sa, out_rank, spos, snf, state, all_distinct)));
  }
  arena_release(c, mk);
  return E_OK;
}

