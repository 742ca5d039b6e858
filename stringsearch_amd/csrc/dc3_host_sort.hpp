// dc3_host_sort.hpp — the sorts: stable LSD radix passes, bucket (MSD) ordering, splitter ordering, pack plans
// Host side of libdc3hip (single translation unit: included by dc3hip.hip in this order; everything here is static).
#pragma once

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort over a bit range of the key (lib.rs:15-39 per digit)
// ---------------------------------------------------------------------------------------------
// tile shapes per record type and digit width (NB bins); LDS = records + NW*NB counters (<= 160 KiB)
template <class Rec, int NB> struct SortCfg;
template <> struct SortCfg<Rec8, 256>  { static constexpr int IPT = 12, NW = 16; static constexpr bool PF = true; };
template <> struct SortCfg<Rec8, 512>  { static constexpr int IPT = 12, NW = 16; static constexpr bool PF = true; };
template <> struct SortCfg<Rec12, 256> { static constexpr int IPT = 10, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec12, 512> { static constexpr int IPT = 10, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec16, 256> { static constexpr int IPT = 8, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Rec16, 512> { static constexpr int IPT = 7, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0, 256>  { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0, 512>  { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0C, 256> { static constexpr int IPT = 8, NW = 8; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0C, 512> { static constexpr int IPT = 7, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0G, 256> { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <> struct SortCfg<Tup0G, 512> { static constexpr int IPT = 6, NW = 16; static constexpr bool PF = false; };
template <class Rec> struct RecClass;      // index into dc3hip_stats.downsweep_*
template <> struct RecClass<Tup0G> { static constexpr int k = 2; };
template <> struct RecClass<Rec8>  { static constexpr int k = 0; };
template <> struct RecClass<Rec12> { static constexpr int k = 1; };
template <> struct RecClass<Rec16> { static constexpr int k = 1; };
template <> struct RecClass<Tup0>  { static constexpr int k = 2; };
template <> struct RecClass<Tup0C> { static constexpr int k = 2; };

template <class Rec, int NB, class Loader, class Sink>
static int launch_downsweep_to(dc3hip_ctx *c, Loader in, Sink dst, u32 n, const Chunking &ck, KeyDig dig,
                               const u32 *table, const u32 *digit_base, int phase) {
  constexpr int IPT = SortCfg<Rec, NB>::IPT, NW = SortCfg<Rec, NB>::NW;
  constexpr bool PF = SortCfg<Rec, NB>::PF && std::is_same<Loader, ArrayLoader<Rec>>::value;
  const size_t smem = DownsweepSmem<Rec, IPT, NW, NB>::kBytes;
  auto kern = k_rs_downsweep<Rec, NB, IPT, NW, PF, Loader, Sink>;
  static std::atomic<bool> attr_set[16];   // (per function and device, process-wide; a double set is harmless)
  if (!attr_set[c->device & 15]) {
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)smem));
    attr_set[c->device & 15] = true;
  }
  PhaseScope ps(c, phase, n, RecClass<Rec>::k);
  hipLaunchKernelGGL(kern, dim3(ck.nchunks), dim3(NW * 64), smem, c->stream, in, dst, n, ck.chunk, ck.nchunks, dig,
                     table, digit_base, 0u);
  KCHECK();
  return E_OK;
}
template <class Rec, int NB, class Loader>
static int launch_downsweep(dc3hip_ctx *c, Loader in, Rec *dst, u32 n, const Chunking &ck, KeyDig dig,
                            const u32 *table, const u32 *digit_base, int phase) {
  RecSink<Rec> sink; sink.p = dst;
  return launch_downsweep_to<Rec, NB, Loader, RecSink<Rec>>(c, in, sink, n, ck, dig, table, digit_base, phase);
}
static int scan_digit_table(dc3hip_ctx *c, u32 *table, u32 nchunks, u32 *digit_base, u32 nb, int phase) {
  PhaseScope ps(c, phase, nb * nchunks);
  hipLaunchKernelGGL(k_scan_rows, dim3(nb), dim3(kBlock), 0, c->stream, table, nchunks, digit_base);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, digit_base, nb, (u32 *)nullptr);
  KCHECK();
  return E_OK;
}

// Stable LSD sort of key bits [bit_lo, bit_hi) of the records in `a` (ping-pong with `b`).
// Digit width: 9 bits where that saves a pass over 8-bit digits, else 8.
// first_table: digit table of the first pass already produced by whoever wrote the records (k_pack_image_text);
// it must have been made for radix_plan()'s chunking and bin count.
// final_sink (Rec8 only): the LAST pass writes through it instead of into the other record buffer; *last then
// describes that pass (source buffer, destination buffer, digit) so that it can be repeated into records
// (radix_redo_last) if the caller turns out to need them after all.
struct LastPass { void *src = nullptr, *dst = nullptr; u32 lo = 0; int nb = 0; };
template <class Rec, int NB>
static int radix_passes(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 bit_lo, u32 bit_hi, Rec **result, int ph_up,
                        int ph_scan, int ph_down, u32 *first_table = nullptr, const SplitSink *final_sink = nullptr,
                        LastPass *last = nullptr) {
  constexpr u32 kBits = NB == 512 ? 9 : 8;
  constexpr int kTile = SortCfg<Rec, NB>::NW * 64 * SortCfg<Rec, NB>::IPT;
  const Chunking ck = make_chunks(c, n, kTile);
  const ArenaMark mk = arena_mark(c);
  u32 *table = first_table, *digit_base = nullptr;
  if (!table) RC(arena_alloc(c, (size_t)NB * ck.nchunks, &table));
  RC(arena_alloc(c, (size_t)NB, &digit_base));
  Rec *src = a, *dst = b;
  for (u32 lo = bit_lo; lo < bit_hi; lo += kBits) {
    KeyDig dig; dig.shift = lo; dig.mask = NB - 1;
    if (!(first_table && lo == bit_lo)) {
      PhaseScope ps(c, ph_up, n);
      hipLaunchKernelGGL((k_rs_upsweep<Rec, NB>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, src, n, ck.chunk,
                         ck.nchunks, dig, table);
      KCHECK();
    }
    RC(scan_digit_table(c, table, ck.nchunks, digit_base, NB, ph_scan));
    ArrayLoader<Rec> ld; ld.p = src;
    if constexpr (std::is_same<Rec, Rec8>::value) {
      if (final_sink && lo + kBits >= bit_hi) {
        RC((launch_downsweep_to<Rec, NB, ArrayLoader<Rec>, SplitSink>(c, ld, *final_sink, n, ck, dig, table, digit_base,
                                                                      ph_down)));
        if (last) { last->src = src; last->dst = dst; last->lo = lo; last->nb = NB; }
        arena_release(c, mk);
        *result = nullptr;                 // the order lives in the sink
        return E_OK;
      }
    }
    RC((launch_downsweep<Rec, NB, ArrayLoader<Rec>>(c, ld, dst, n, ck, dig, table, digit_base, ph_down)));
    std::swap(src, dst);
  }
  arena_release(c, mk);
  *result = src;
  return E_OK;
}
// repeat the last pass of a sort that ended in a SplitSink, this time into records
template <int NB>
static int radix_redo_last_nb(dc3hip_ctx *c, const LastPass &lp, u32 n, Rec8 **result, int ph_up, int ph_scan, int ph_down) {
  constexpr int kTile = SortCfg<Rec8, NB>::NW * 64 * SortCfg<Rec8, NB>::IPT;
  const Chunking ck = make_chunks(c, n, kTile);
  const ArenaMark mk = arena_mark(c);
  u32 *table = nullptr, *digit_base = nullptr;
  RC(arena_alloc(c, (size_t)NB * ck.nchunks, &table));
  RC(arena_alloc(c, (size_t)NB, &digit_base));
  KeyDig dig; dig.shift = lp.lo; dig.mask = NB - 1;
  Rec8 *src = static_cast<Rec8 *>(lp.src), *dst = static_cast<Rec8 *>(lp.dst);
  {
    PhaseScope ps(c, ph_up, n);
    hipLaunchKernelGGL((k_rs_upsweep<Rec8, NB>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, src, n, ck.chunk, ck.nchunks,
                       dig, table);
    KCHECK();
  }
  RC(scan_digit_table(c, table, ck.nchunks, digit_base, NB, ph_scan));
  ArrayLoader<Rec8> ld; ld.p = src;
  RC((launch_downsweep<Rec8, NB, ArrayLoader<Rec8>>(c, ld, dst, n, ck, dig, table, digit_base, ph_down)));
  arena_release(c, mk);
  *result = dst;
  return E_OK;
}
static int radix_redo_last(dc3hip_ctx *c, const LastPass &lp, u32 n, Rec8 **result, int ph_up, int ph_scan, int ph_down) {
  return lp.nb == 512 ? radix_redo_last_nb<512>(c, lp, n, result, ph_up, ph_scan, ph_down)
                      : radix_redo_last_nb<256>(c, lp, n, result, ph_up, ph_scan, ph_down);
}
static bool radix_nine(const dc3hip_ctx *, u32 bits) { return (bits + 8) / 9 < (bits + 7) / 8; }
template <class Rec>
static int radix_sort(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 bit_lo, u32 bit_hi, Rec **result, int ph_up,
                      int ph_scan, int ph_down, u32 *first_table = nullptr, const SplitSink *final_sink = nullptr,
                      LastPass *last = nullptr) {
  const u32 bits = bit_hi > bit_lo ? bit_hi - bit_lo : 0;
  if (bits == 0) { *result = a; return E_OK; }
  if (radix_nine(c, bits))
    return radix_passes<Rec, 512>(c, a, b, n, bit_lo, bit_hi, result, ph_up, ph_scan, ph_down, first_table, final_sink, last);
  return radix_passes<Rec, 256>(c, a, b, n, bit_lo, bit_hi, result, ph_up, ph_scan, ph_down, first_table, final_sink, last);
}
// bins and chunking radix_sort<Rec> will use for n records and `bits` key bits
template <class Rec>
static void radix_plan(dc3hip_ctx *c, u32 n, u32 bits, int *nb, Chunking *ck) {
  const bool nine = radix_nine(c, bits);
  *nb = nine ? 512 : 256;
  const int tile = nine ? SortCfg<Rec, 512>::NW * 64 * SortCfg<Rec, 512>::IPT : SortCfg<Rec, 256>::NW * 64 * SortCfg<Rec, 256>::IPT;
  *ck = make_chunks(c, n, (u32)tile);
}

// ---------------------------------------------------------------------------------------------
// Bucket (MSD) ordering of the prefix-sort words (dc3_msd.hip.hpp): the same array the stable LSD passes over the image
// bits produce, in two non-stable partition passes + an in-LDS order of the sub-buckets.
// ---------------------------------------------------------------------------------------------
struct MsdGeom {
  bool on = false;
  u32 d1 = 0, d2 = 0;                            // digit widths of the two partition passes (d2 = 0: one pass)
  u32 ntiles1 = 0, tpc = 0, cpg = 0, cpx1 = 0;   // pass-1 tiles; tiles per pack chunk, chunks and tiles per group
  Chunking ck{0, 0};                             // chunking of the pack kernel that produces the bucket sizes
  u64 img_lo = 0;                                // the records' images lie in [img_lo, img_lo + 2^ebits): digits come from
  u32 ebits = 0;                                 // image - img_lo, ebits wide (the whole range: 0, hm.nbits)
};
static constexpr u32 kMsdCapSmall = 2048, kMsdCapLarge = 4096;     // sub-bucket capacities of the two local-sort shapes
// Geometry for nrec words with hm's layout, or .on = false when the bucket ordering does not apply (switched off, too
// few records, or too few image bits below the bucket bits for the local sort's bins).
// img_lo / img_span: the records hold only the images in [img_lo, img_lo + img_span) (0 = the whole range).
static MsdGeom msd_geometry(const dc3hip_ctx *c, u32 nrec, const HiMap &hm, u64 img_lo = 0, u64 img_span = 0) {
  MsdGeom g;
  if (c->no_msd || nrec < c->msd_min || nrec < 4096 || hm.pbits > 32 || hm.pbits + hm.nbits > 64) return g;
  g.img_lo = img_span ? img_lo : 0;
  g.ebits = img_span ? std::min<u32>(hm.nbits, bits_of(img_span - 1)) : hm.nbits;
  const u32 lg = bits_of((u64)nrec - 1);                       // ceil(log2 nrec)
  u32 tb = lg > 10 ? lg - 10 : 1;                              // sub-buckets of 512..1024 words on uniform images
  if (tb > 20) tb = 20;
  if (g.ebits < tb + 4) return g;
  if (tb <= 10) { g.d1 = tb; g.d2 = 0; } else { g.d1 = (tb + 1) / 2; g.d2 = tb - g.d1; }
  g.ntiles1 = (nrec + kMsdTile - 1) / kMsdTile;
  g.tpc = std::max<u32>(1, (g.ntiles1 + 2047) / 2048);
  g.cpg = ((g.ntiles1 + kMsdGroups - 1) / kMsdGroups + g.tpc - 1) / g.tpc;
  g.cpx1 = g.cpg * g.tpc;
  g.ck.chunk = g.tpc * (u32)kMsdTile;
  g.ck.nchunks = (nrec + g.ck.chunk - 1) / g.ck.chunk;
  g.on = true;
  return g;
}
// Pass 1 of a sort whose words are made on the fly from a key maker (k_msd_part_keys) instead of being read from `ha`:
// the pack kernel then only counted.  launch() = that kernel with the sort's geometry.
struct MsdPass1 {
  virtual ~MsdPass1() {}
  virtual int launch(dc3hip_ctx *c, u64 *out, u32 n, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) = 0;
  // a caller that has to give the bucket ordering up has no words to continue from (the pack kernel only counted):
  // repack() writes the plain words of all positions, in position order, with the LSD passes' first digit table
  virtual int repack(dc3hip_ctx *, Rec8 *, u32, u32 **) { set_err("internal: this pass 1 cannot repack"); return E_HIP; }
};
template <class KM> static int launch_pack_all(dc3hip_ctx *c, KM km, u32 nrec, const HiMap &hm, Rec8 *out, u32 **first_table,
                                               const MsdGeom *mg = nullptr, bool store = true);
template <class KM>
struct MsdPass1Keys : MsdPass1 {
  KM km; HiMap hm; u64 P1 = 0;
  bool strip = false; HiMap hm_plain{};      // strip: hm is the WIDER image (hm.pbits = position bits - d1); hm_plain the words' own layout
  int repack(dc3hip_ctx *c, Rec8 *out, u32 nrec, u32 **first_table) override {
    PhaseScope ps(c, DC3HIP_PH_PACK, nrec);
    return launch_pack_all<KM>(c, km, nrec, hm_plain, out, first_table, nullptr, true);
  }
  int launch(dc3hip_ctx *c, u64 *out, u32 n, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      attr_set[c->device & 15] = true;
    }
    if (strip && (hm.nbits + hm.pbits != 64 || g.d1 == 0)) { set_err("internal: a stripped image must fill the word"); return E_HIP; }
    if (strip)
      hipLaunchKernelGGL((k_msd_part_keys<KM, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, km, hm, P1, out, n, base, sh1,
                         g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon);
    else
      hipLaunchKernelGGL((k_msd_part_keys<KM, false>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, km, hm, P1, out, n, base, sh1,
                         g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon);
    KCHECK();
    return E_OK;
  }
};
// what a finished sort leaves behind so that its last pass can be repeated into records (cf. LastPass)
struct MsdRedo { const u64 *src = nullptr; u64 *dst = nullptr; const u32 *start = nullptr; u32 nsub = 0, shb = 0; bool large = false; u64 base = 0;
                 u32 slot_cap = 0; /* != 0: sub-bucket s lies in src[s * slot_cap ..) (k_msd_part<.., kSlot>) */ };
template <class Sink>
static int msd_launch_local(dc3hip_ctx *c, const MsdRedo &r, u32 n, Sink sink) {
  PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, 6);
  const bool hi = r.base == 0 && r.shb >= 32;          // the bin is a bit field of the word's upper half
  auto go = [&](auto kern, u32 nt, size_t cap) {
    hipLaunchKernelGGL(kern, dim3(r.nsub), dim3(nt), (cap + kMsdLocPad) * 8, c->stream, r.src, r.start, r.base, r.shb, sink, r.slot_cap);
  };
  if (r.large) { if (hi) go(k_msd_local<512, (int)kMsdCapLarge, 12, Sink, true>, 512, kMsdCapLarge); else go(k_msd_local<512, (int)kMsdCapLarge, 12, Sink, false>, 512, kMsdCapLarge); }
  else { if (hi) go(k_msd_local<256, (int)kMsdCapSmall, 10, Sink, true>, 256, kMsdCapSmall); else go(k_msd_local<256, (int)kMsdCapSmall, 10, Sink, false>, 256, kMsdCapSmall); }
  KCHECK();
  return E_OK;
}
// Sort the n words of `ha` (scratch `hb`) by image bits [pbits, pbits + nbits).  table = the pack kernel's digit table of
// the top g.d1 image bits ([1024][g.ck.nchunks]).  split != nullptr: the last pass writes positions + 32 image bits
// through it (as the LSD passes do with a SplitSink) and *result = nullptr; else *result = the sorted records.
// *ok = false: a sub-bucket was too large for the local sort — seen BEFORE pass 2 is launched, so `ha` still holds the
// caller's words in their original (position) order (*where = ha; with p1 they were never written: the caller repacks)
// and the stable LSD passes start from there, exactly as if the bucket ordering had not been tried: the order of equal
// images the tie pass meets does not depend on which way the sort went.  The small tables stay allocated in the arena
// until the caller releases its mark (redo reads them).
static int msd_sort(dc3hip_ctx *c, Rec8 *ha, Rec8 *hb, u32 n, const HiMap &hm, const MsdGeom &g, const u32 *table,
                    const SplitSink *split, Rec8 **result, MsdRedo *redo, bool *ok, Rec8 **where, MsdPass1 *p1 = nullptr,
                    uint8_t *same_out = nullptr, bool allow_slots = false) {
  // same_out (record form only): same_out[i] = 1 iff sorted record i has the image of record i - 1
  // allow_slots (single device only: a rank of the global mode keeps to its memory plan and the counted form): pass 2 may take
  // 16 bytes per word of the arena for its slots.  With a split sink they stay until the caller releases its mark (the last
  // pass may be repeated into records: msd_redo), so the caller must not need that room — the whole-text order does not.  In
  // the record form nothing reads them after the local pass: they are released here, before the caller's rank inversion
  // takes its two record arrays (kept, they made the first build of a process run out of arena and be retried).
  // p1 != nullptr: the words do not exist yet — pass 1 makes them from the key maker (`ha` is then only the scratch of
  // pass 2); needs `table` (the counting pack kernel's)
  *ok = false; *where = ha; *result = nullptr;
  if (p1 && !table) { set_err("internal: on-the-fly pass 1 without a digit table"); return E_HIP; }
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
    attr_set[c->device & 15] = true;
  }
  const u32 nb1 = 1u << g.d1, tb = g.d1 + g.d2, n2 = 1u << tb;
  const u32 sh1 = hm.pbits + g.ebits - g.d1, sh2 = sh1 - g.d2, rb = g.ebits - tb;
  const u64 base = g.img_lo << hm.pbits;
  u32 *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *segsum = nullptr;
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cntg));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &startg));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb1 + 16, &bstart)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpre)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpreh));
  RC(arena_alloc(c, (size_t)kMsdW_COUNT + 12, &plan)); RC(arena_alloc(c, (size_t)1024 + 16, &segsum));
  u64 *wa = reinterpret_cast<u64 *>(ha), *wb = reinterpret_cast<u64 *>(hb);
  if (!table) {                              // records packed elsewhere: count the top digit here (one read of the records)
    u32 *t = nullptr;
    RC(arena_alloc(c, (size_t)kMsdMaxDig * g.ck.nchunks, &t));
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL(k_msd_hist1, dim3(g.ck.nchunks), dim3(kBlock), 0, c->stream, (const u64 *)wa, n, base, sh1, g.ck.chunk, g.ck.nchunks, t);
    KCHECK();
    table = t;
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
    HIPC(hipMemsetAsync(plan, 0, (kMsdW_COUNT + 12) * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_msd_cnt1, dim3(nb1), dim3(kBlock), 0, c->stream, table, g.ck.nchunks, g.cpg, cntg);
    KCHECK();
    hipLaunchKernelGGL(k_msd_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb1, n, startg, cur1, bstart, tpre, tpreh, plan);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, p1 ? 9 : 5);     // (class 9: pass 1 that also makes the words, timed on its own)
    if (p1) {
      RC(p1->launch(c, wb, n, base, sh1, g, nb1, plan, cur1));
    } else {
      // (base == 0 and the digit inside the word's upper half: the digit is one bit-field instruction, k_msd_part<.., kHi>)
      auto kern = (base == 0 && sh1 >= 32) ? k_msd_part<false, true> : k_msd_part<false, false>;
      hipLaunchKernelGGL(kern, dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, (const u64 *)wa, wb, n,
                         base, sh1, g.d1, g.cpx1, g.ntiles1, (const u32 *)nullptr, (const u32 *)nullptr, nb1, (const u32 *)plan, cur1, nb1, c->d_xcdmon, 0u, 0u);
      KCHECK();
    }
  }
  MsdRedo r;
  r.base = base;
  // Pass 2 into SLOTS (k_msd_part<.., kSlot>): when the pass-1 buckets are even (largest <= 1.25 x the mean: the sub-buckets of
  // such inputs are Poisson-sized around n / n2 <= 1024) and the arena has 16 bytes per word to spare, every sub-bucket gets
  // a slot of twice the mean and pass 2 needs no sizes in advance: k_msd_hist2 (a read of all words) and its scans before the
  // pass are not run, the sizes are read off the slot cursors afterwards.  A slot that overflows (seen in the same maximum the
  // counted form checks) sends the sort through the counted form below, from the untouched output of pass 1.
  bool slot_done = false;
  const ArenaMark slot_mk = arena_mark(c);
  if (g.d2 > 0 && allow_slots) {
    const u32 mean = (u32)(((u64)n + n2 - 1) / n2);
    // (slots of twice the mean for the small local shape; where the mean passes what that shape holds with room to spare — texts
    //  beyond 1.7e9 positions, whose 2^20 sub-buckets average more than 1638 words — slots of 1.5 x the mean for the large
    //  shape, up to the 4032 words at which 2^20 slots still index with 32 bits: Poisson sizes around 2049 stay below 2300)
    const bool small_fits = (u64)mean * 5 <= (u64)kMsdCapSmall * 4;
    const u32 slot_cap = c->msd_slot_cap ? std::min<u32>(kMsdCapSmall, c->msd_slot_cap)
                         : small_fits ? std::min<u32>(kMsdCapSmall, (2 * mean + 63) & ~63u)
                                      : std::min<u32>(kMsdCapLarge - 64u, (mean + mean / 2 + 63) & ~63u);
    const u64 slot_words = (u64)n2 * slot_cap + kMsdTile;
    const size_t N = n2;                                                 // one cursor per sub-bucket
    const size_t need = align_up(slot_words * 8, 256) + align_up((N + 16) * 4, 256) + (split ? (size_t)n * 2 + (64u << 20) : (size_t)(1u << 20));   // (+ what the tie pass takes afterwards)
    // what the host knows by itself comes first: slots that index with 32 bits, that hold the mean sub-bucket with room to
    // spare (beyond 2^31 words the mean passes the local sort's small shape: counted form), an arena with room — one that
    // can grow where it lies commits the 16 bytes per word now (arena_grow_in_use; first sort of a context only).  Only then
    // is the largest pass-1 bucket read back (a pipeline drain that a sort without slots must not pay).
    // The FIRST build of a one-shot call's context keeps to the counted form unless the room is there already: the slots cost
    // 16 bytes per word of device memory, and memory somebody freed shortly before is handed out at about 30 ms per GiB (the
    // driver wipes it) — more than one call gains; when the cached context builds again it commits them.  A context the
    // caller created (dc3hip_ctx_create: repeated builds) takes them at once.
    const bool may_grow = !c->one_shot || c->builds_done > 0 || c->msd_slot_cap != 0;
    const bool host_ok = slot_words < (1ull << 32) && (c->msd_slot_cap || (u64)mean * 5 <= (u64)slot_cap * 4) &&
                         (c->arena_off + need <= c->arena_bytes || (may_grow && arena_grow_in_use(c, c->arena_off + need)));
    u32 maxb1 = 0;
    if (host_ok) {
      HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      maxb1 = c->h_words[20 + kMsdW_MAXB1];
    }
    if (host_ok && (u64)maxb1 * nb1 * 4 <= (u64)n * 5) {
      u64 *slots = nullptr; u32 *scnt = nullptr;
      RC(arena_alloc(c, (size_t)slot_words, &slots));
      RC(arena_alloc(c, N + 16, &scnt));
      const u32 nseg = (u32)((N + kMsdScanSeg - 1) / kMsdScanSeg);       // <= 256
      {
        PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, 5);
        HIPC(hipMemsetAsync(scnt, 0, (N + 1) * sizeof(u32), c->stream));
        const u32 grid2 = kMsdGroups * ((n / kMsdTile + nb1 + 1 + kMsdGroups - 1) / kMsdGroups);
        auto kern = (base == 0 && sh2 >= 32) ? k_msd_part<true, true, true> : k_msd_part<true, false, true>;
        hipLaunchKernelGGL(kern, dim3(grid2), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, (const u64 *)wb, slots, n, base, sh2, g.d2,
                           0u, 0u, (const u32 *)tpre, (const u32 *)bstart, nb1, (const u32 *)plan, scnt, n2, c->d_xcdmon, slot_cap,
                           (u32)(slot_words - kMsdTile));
        KCHECK();
      }
      {
        PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, N);
        hipLaunchKernelGGL(k_msd_slot_scan_a, dim3(nseg), dim3(1024), 0, c->stream, (const u32 *)scnt, (u32)N, segsum, plan);
        KCHECK();
        hipLaunchKernelGGL(k_msd_slot_scan_c, dim3(nseg), dim3(1024), 0, c->stream, scnt, (u32)N, (const u32 *)segsum);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      if (c->h_words[20 + kMsdW_MAXSUB] <= slot_cap) {
        r.src = slots; r.dst = wb; r.start = scnt; r.nsub = n2; r.slot_cap = slot_cap;
        slot_done = true;
        c->stats.msd_slot_sorts++;
      } else {
        HIPC(hipMemsetAsync(plan + kMsdW_MAXSUB, 0, sizeof(u32), c->stream));      // (the counted form takes its own maximum)
        arena_release(c, slot_mk);
      }
    }
  }
  if (slot_done) {
  } else if (g.d2 > 0) {
    const size_t N = (size_t)n2 * kMsdGroups;
    u32 *cnt2g = nullptr, *cur2 = nullptr;
    RC(arena_alloc(c, N + 16, &cnt2g));
    RC(arena_alloc(c, N + 16, &cur2));
    const u32 nseg = (u32)((N + kMsdScanSeg - 1) / kMsdScanSeg);       // <= 1024
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
      HIPC(hipMemsetAsync(cnt2g, 0, (N + 1) * sizeof(u32), c->stream));
      hipLaunchKernelGGL(k_msd_hist2, dim3(n / kMsdHistTile + nb1 + 1), dim3(1024), 0, c->stream, (const u64 *)wb, base, sh2, g.d2,
                         (const u32 *)tpre, (const u32 *)tpreh, (const u32 *)bstart, nb1, (const u32 *)plan, cnt2g);
      KCHECK();
    }
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, N);
      hipLaunchKernelGGL(k_msd_scan2a, dim3(nseg), dim3(1024), 0, c->stream, (const u32 *)cnt2g, (u32)N, segsum, plan);
      KCHECK();
      hipLaunchKernelGGL(k_msd_scan2c, dim3(nseg), dim3(1024), 0, c->stream, cnt2g, (u32)N, n2, (const u32 *)segsum, cur2);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));                  // (pass 2 overwrites `ha`: decide first)
    if (c->h_words[20 + kMsdW_MAXSUB] > kMsdCapLarge) { c->stats.msd_max_subbucket = c->h_words[20 + kMsdW_MAXSUB]; c->stats.msd_fallbacks++; return E_OK; }
    {
      PhaseScope ps(c, DC3HIP_PH_SORT8_DOWN, n, 5);
      const u32 grid2 = kMsdGroups * ((n / kMsdTile + nb1 + 1 + kMsdGroups - 1) / kMsdGroups);
      auto kern = (base == 0 && sh2 >= 32) ? k_msd_part<true, true> : k_msd_part<true, false>;
      hipLaunchKernelGGL(kern, dim3(grid2), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, (const u64 *)wb, wa, n, base, sh2, g.d2,
                         0u, 0u, (const u32 *)tpre, (const u32 *)bstart, nb1, (const u32 *)plan, cur2, n2, c->d_xcdmon, 0u, 0u);
      KCHECK();
    }
    r.src = wa; r.dst = wb; r.start = cnt2g; r.nsub = n2;
  } else {
    {
      PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
      hipLaunchKernelGGL(k_msd_scan2a, dim3((nb1 * kMsdGroups + kMsdScanSeg - 1) / kMsdScanSeg), dim3(1024), 0, c->stream, (const u32 *)cntg,
                         nb1 * kMsdGroups, segsum, plan);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    r.src = wb; r.dst = wa; r.start = startg; r.nsub = nb1;
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 maxsub = c->h_words[20 + kMsdW_MAXSUB];
  c->stats.msd_max_subbucket = maxsub;
  if (maxsub > kMsdCapLarge) { c->stats.msd_fallbacks++; return E_OK; }      // (d2 == 0: only `hb` was written)
  r.large = maxsub > kMsdCapSmall;
  r.shb = sh2 - std::min<u32>(r.large ? 12u : 10u, rb);
  if (split) {
    MsdSplitSink sk; sk.sa = split->sa; sk.same = split->same; sk.pbits = split->pbits; sk.tilef = split->tilef;
    RC(msd_launch_local(c, r, n, sk));
  } else if (same_out) {
    MsdRecSameSink sk; sk.p = r.dst; sk.same = same_out; sk.pbits = hm.pbits;
    RC(msd_launch_local(c, r, n, sk));
    *result = reinterpret_cast<Rec8 *>(r.dst);
  } else {
    MsdRecSink sk; sk.p = r.dst;
    RC(msd_launch_local(c, r, n, sk));
    *result = reinterpret_cast<Rec8 *>(r.dst);
  }
  c->stats.msd_sorts++;
  if (slot_done && !split) { arena_release(c, slot_mk); r.src = nullptr; r.start = nullptr; }   // (stream order: later kernels run behind the local pass)
  *redo = r;
  *ok = true;
  return E_OK;
}
// repeat the last pass of an MSD sort that ended in a split sink, this time into records
static int msd_redo(dc3hip_ctx *c, const MsdRedo &r, u32 n, Rec8 **result) {
  MsdRecSink sk; sk.p = r.dst;
  RC(msd_launch_local(c, r, n, sk));
  *result = reinterpret_cast<Rec8 *>(r.dst);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// Splitter ordering of the sample-triple records (dc3_ssort.hip.hpp): the array radix_sort<Rec>(a, b, n, 0, kbits) makes
// from records in position order, in two partition passes over sampled splitters + an in-LDS order of the sub-buckets.
// *ok = false: not applied (too few records, switched off, or a sub-bucket beyond the local capacity — `a` is untouched
// in every such case and the caller runs the LSD passes).
// ---------------------------------------------------------------------------------------------
static constexpr u32 kSsCap = 4096;
// Measured on MI355X (1 GiB text): 318 M 16-byte records with 81-bit keys, 21 ms against 32 ms for the 63-bit
// prefix + tie rounds (and 9 LSD passes for the straight order); 477 M 12-byte records with 45-bit keys, 28 ms against
// 21 ms for the 5 LSD passes — so the keys of at most 64 bits stay with the LSD passes (DC3HIP_SSORT_REC12=1: tests).
static bool ssort_applies(const dc3hip_ctx *c, u32 n, u32 kbits) {
  return n >= c->ssort_min && n >= 8192 && kbits > 64;
}
// A caller whose records do not exist yet hands in a producer: sample() computes S of them (ascending index),
// pack_count() makes all of them into `a` while it counts the coarse buckets (k_ss_count1's arguments).
struct SsProducer {
  virtual ~SsProducer() {}
  virtual int sample(dc3hip_ctx *c, u32 n, u32 S, void *out) = 0;
  virtual int pack_count(dc3hip_ctx *c, void *a, u32 n, const SsVal *coarse, u32 nb1, u32 tile, u32 cpx, u32 ntiles, u32 tpb,
                         u32 grid, u32 *cntg, uint16_t *dig) = 0;
};
// nb1 coarse buckets x F2 sub-buckets of about ssort_mean records, S sample values; false: the ordering does not apply
struct SsGeom { u32 nb1, F2, n2, S; };
static bool ssort_geometry(const dc3hip_ctx *c, u32 n, u32 kbits, SsGeom *g, size_t rec_bytes = 16) {
  if (!ssort_applies(c, n, kbits)) return false;
  const u64 want = ((u64)n + c->ssort_mean - 1) / c->ssort_mean;           // sub-buckets
  u32 nb1 = kSsMaxDig, F2 = (u32)((want + nb1 - 1) / nb1);
  if (F2 < 2) { F2 = 2; nb1 = (u32)std::max<u64>(2, (want + 1) / 2); }
  if (F2 > kSsMaxDig) return false;                                        // (beyond 1.4e9 records)
  g->nb1 = nb1; g->F2 = F2; g->n2 = nb1 * F2; g->S = g->n2 * c->ssort_over;
  if ((u64)g->S * 4 > n) return false;
  // scratch on top of the caller's two record arrays: the sample twice, a digit per record, splitters and size tables;
  // when the arena cannot hold it the LSD passes run (arena_requirement() models those)
  const size_t need = 2 * (size_t)g->S * rec_bytes + (size_t)n * 2 + (size_t)g->n2 * (16 + 2 * 8 * 4) + ((size_t)48 << 20);
  return c->arena_bytes - c->arena_off >= need;
}
template <class Rec>
static int ssort(dc3hip_ctx *c, Rec *a, Rec *b, u32 n, u32 kbits, Rec **result, bool *ok, SsProducer *prod = nullptr) {
  // prod != nullptr: only when ssort_geometry() holds (the caller checked); on return `a` holds the records either way
  *ok = false; *result = nullptr;
  SsGeom geo;
  if (!ssort_geometry(c, n, kbits, &geo, sizeof(Rec))) {
    if (prod) { set_err("internal: splitter ordering with a producer outside its range"); return E_HIP; }
    return E_OK;
  }
  constexpr int IPT = SsCfg<Rec>::IPT;
  constexpr u32 tile = (u32)kSsNT * IPT, htile = tile * kSsHistTiles;
  constexpr int kLocNT = 1024, kLocIPT = (int)(kSsCap / kLocNT);
  constexpr size_t part_smem = ss_part_smem<Rec>(), loc_smem = sizeof(Rec) * (kSsCap + kSsLocPad) + kSsCap;
  static std::atomic<bool> attr_set[16];
  if (!attr_set[c->device & 15]) {
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_ss_part<Rec, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_smem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_ss_part<Rec, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)part_smem));
    HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_ss_local<Rec, kLocNT, kLocIPT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)loc_smem));
    attr_set[c->device & 15] = true;
  }
  const u32 nb1 = geo.nb1, F2 = geo.F2, n2 = geo.n2, S = geo.S;
  const u32 ntiles1 = (n + tile - 1) / tile, cpx1 = (ntiles1 + kSsGroups - 1) / kSsGroups;
  const u32 tpb = std::max<u32>(1, (cpx1 + 255) / 256);
  const ArenaMark mk = arena_mark(c);
  Rec *sa = nullptr, *sb = nullptr, *ss = nullptr;
  SsVal *fine = nullptr, *coarse = nullptr;
  u32 *cntg = nullptr, *startg = nullptr, *cur1 = nullptr, *bstart = nullptr, *tpre = nullptr, *tpreh = nullptr, *plan = nullptr, *segsum = nullptr;
  u32 *cnt2g = nullptr, *cur2 = nullptr;
  uint16_t *dig = nullptr;
  const size_t N2 = (size_t)n2 * kSsGroups;
  RC(arena_alloc(c, (size_t)S, &sa)); RC(arena_alloc(c, (size_t)S, &sb));
  RC(arena_alloc(c, (size_t)n2 + 16, &fine)); RC(arena_alloc(c, (size_t)kSsMaxDig + 16, &coarse));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &cntg));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &startg));
  RC(arena_alloc(c, (size_t)nb1 * kSsGroups + 16, &cur1));
  RC(arena_alloc(c, (size_t)nb1 + 16, &bstart)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpre)); RC(arena_alloc(c, (size_t)nb1 + 16, &tpreh));
  RC(arena_alloc(c, (size_t)kMsdW_COUNT + 12, &plan)); RC(arena_alloc(c, (size_t)1024 + 16, &segsum));
  RC(arena_alloc(c, N2 + 16, &cnt2g)); RC(arena_alloc(c, N2 + 16, &cur2));
  RC(arena_alloc(c, (size_t)n + 16, &dig));
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, S);
    if (prod) RC(prod->sample(c, n, S, sa));
    else {
      hipLaunchKernelGGL((k_ss_sample<Rec>), dim3((S + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (const Rec *)a, n, S, sa);
      KCHECK();
    }
  }
  RC(radix_sort<Rec>(c, sa, sb, S, 0, kbits, &ss, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL((k_ss_splitters<Rec>), dim3((n2 + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, (const Rec *)ss, n2, F2, c->ssort_over, fine, coarse);
    KCHECK();
    HIPC(hipMemsetAsync(cntg, 0, ((size_t)nb1 * kSsGroups + 16) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(plan, 0, (kMsdW_COUNT + 12) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(cnt2g, 0, (N2 + 1) * sizeof(u32), c->stream));
    const u32 grid1 = kSsGroups * ((cpx1 + tpb - 1) / tpb);
    if (prod) RC(prod->pack_count(c, a, n, coarse, nb1, tile, cpx1, ntiles1, tpb, grid1, cntg, dig));
    else {
      hipLaunchKernelGGL((k_ss_count1<Rec>), dim3(grid1), dim3(kSsNT), 0, c->stream, (const Rec *)a, n,
                         (const SsVal *)coarse, nb1, tile, cpx1, ntiles1, tpb, cntg, dig);
      KCHECK();
    }
  }
  unsigned long long *vsum = nullptr;
  if (c->ssort_verify) {
    RC(arena_alloc(c, (size_t)8, &vsum));
    HIPC(hipMemsetAsync(vsum, 0, 8 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)a, n, vsum);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, nb1);
    hipLaunchKernelGGL(k_ss_plan1, dim3(1), dim3(1024), 0, c->stream, (const u32 *)cntg, nb1, n, tile, htile, startg, cur1, bstart, tpre, tpreh, plan);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 7);
    hipLaunchKernelGGL((k_ss_part<Rec, false>), dim3(kSsGroups * cpx1), dim3(kSsNT), part_smem, c->stream, (const Rec *)a, b, n,
                       (const uint16_t *)dig, F2, cpx1, ntiles1, (const u32 *)nullptr, (const u32 *)nullptr, nb1, (const u32 *)plan, cur1, nb1);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)b, n, vsum + 4);
    KCHECK();
  }
  const u32 nseg = (u32)((N2 + kMsdScanSeg - 1) / kMsdScanSeg);              // <= 1024
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_UP, n);
    hipLaunchKernelGGL((k_ss_hist2<Rec>), dim3(n / htile + nb1 + 1), dim3(kSsNT), 0, c->stream, (const Rec *)b, (const SsVal *)fine, F2, tile,
                       (const u32 *)tpre, (const u32 *)tpreh, (const u32 *)bstart, nb1, (const u32 *)plan, cnt2g, dig);
    KCHECK();
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_SCAN, N2);
    hipLaunchKernelGGL(k_msd_scan2a, dim3(nseg), dim3(1024), 0, c->stream, (const u32 *)cnt2g, (u32)N2, segsum, plan);
    KCHECK();
    hipLaunchKernelGGL(k_msd_scan2c, dim3(nseg), dim3(1024), 0, c->stream, cnt2g, (u32)N2, n2, (const u32 *)segsum, cur2);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 20, plan, kMsdW_COUNT * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 maxsub = c->h_words[20 + kMsdW_MAXSUB];
  c->stats.ssort_max_subbucket = maxsub;
  if (maxsub > kSsCap) { c->stats.ssort_fallbacks++; arena_release(c, mk); return E_OK; }     // (a is still the input)
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 7);
    const u32 grid2 = kSsGroups * ((n / tile + nb1 + 1 + kSsGroups - 1) / kSsGroups);
    hipLaunchKernelGGL((k_ss_part<Rec, true>), dim3(grid2), dim3(kSsNT), part_smem, c->stream, (const Rec *)b, a, n, (const uint16_t *)dig, F2, 0u, 0u,
                       (const u32 *)tpre, (const u32 *)bstart, nb1, (const u32 *)plan, cur2, n2);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)a, n, vsum + 6);
    KCHECK();
    unsigned long long *vc = nullptr;
    RC(arena_alloc(c, (size_t)8, &vc));
    HIPC(hipMemsetAsync(vc, 0, 8 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(k_ss_verify_cursors, dim3((u32)((N2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, (const u32 *)cnt2g, (const u32 *)cur2, n2, vc);
    KCHECK();
    unsigned long long hc[4];
    HIPC(hipMemcpyAsync(hc, vc, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (hc[0]) {
      set_err("DC3HIP_SSORT_VERIFY: after pass 2 %llu regions are off; first (sub-bucket %llu, group %llu): cursor %llu, expected %llu (n=%u F2=%u)",
              hc[0], hc[1] / 8, hc[1] % 8, hc[2], hc[3], n, F2);
      return E_HIP;
    }
  }
  {
    PhaseScope ps(c, DC3HIP_PH_SORT12_DOWN, n, 8);
    hipLaunchKernelGGL((k_ss_local<Rec, kLocNT, kLocIPT>), dim3(n2), dim3(kLocNT), loc_smem, c->stream, (const Rec *)a, (const u32 *)cnt2g, b);
    KCHECK();
  }
  if (c->ssort_verify) {
    hipLaunchKernelGGL((k_ss_verify<Rec>), dim3(2048), dim3(kBlock), 0, c->stream, (const Rec *)b, n, vsum + 2);
    KCHECK();
    unsigned long long h[8];
    HIPC(hipMemcpyAsync(h, vsum, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (h[0] != h[2] || h[3] != 0) {
      set_err("DC3HIP_SSORT_VERIFY: n=%u rec=%zu nb1=%u F2=%u S=%u checksums in=%llx pass1=%llx pass2=%llx out=%llx, %llu descents, largest sub-bucket %u",
              n, sizeof(Rec), nb1, F2, S, h[0], h[4], h[6], h[2], h[3], maxsub);
      return E_HIP;
    }
  }
  c->stats.ssort_sorts++;
  arena_release(c, mk);        // (the stream orders the kernels above before whatever reuses the scratch)
  *result = b;
  *ok = true;
  return E_OK;
}

// Digit table of a pack kernel (k_pack_image_*): bins, chunking and which image bits it counts.
// mg (bucket ordering, msd_geometry): the table counts the TOP mg->d1 image bits in mg's chunking instead (1024 rows).
static void pack_plan(dc3hip_ctx *c, u32 nrec, const HiMap &hm, const MsdGeom *mg, int *nb, Chunking *ck, u32 *hshift) {
  if (mg && mg->on) { *nb = 1024; *ck = mg->ck; *hshift = hm.nbits - mg->d1; }
  else { radix_plan<Rec8>(c, nrec, hm.nbits, nb, ck); *hshift = 0; }
}
#define DC3_PACK_LAUNCH(KERNEL_NB, ...)                                                                              \
  do {                                                                                                               \
    if (nb == 1024) hipLaunchKernelGGL(KERNEL_NB(1024), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__);   \
    else if (nb == 512) hipLaunchKernelGGL(KERNEL_NB(512), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL_NB(256), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, __VA_ARGS__);               \
    KCHECK();                                                                                                        \
  } while (0)

