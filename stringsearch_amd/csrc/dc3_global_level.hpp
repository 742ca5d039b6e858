// dc3_global_level.hpp — one rank of a global build: the rank exchange, the orderings split by key range (selecting /
// routed whole-level and whole-text orders, sampled naming, discarding) and the distributed level driver glevel —
// crates/dc3/src/lib.rs:44-193 with the two rules of dc3_global_host.hpp.  Included by dc3_global_host.hpp.
#pragma once

// ---------------------------------------------------------------------------------------------
// one rank of a global build
// ---------------------------------------------------------------------------------------------
struct dc3hip_gctx {
  bool no_wide_msd = false;    // DC3HIP_NO_WIDE_MSD=1: wide mode always sorts 16-byte records with the LSD passes
  u64 wide_msd_min = 1ull << 22; // DC3HIP_WIDE_MSD_MIN (tests): fewest positions per rank for the wide bucket ordering
  bool wide_msd_forced = false;  // ... given explicitly: texts below 2^32 take the unrouted order at every rank count
  u32 w_depth = 0;             // wide mode: symbols the last tie pass of the last build compared (the verifier compares at least as deep)
  bool no_select = false;      // DC3HIP_GLOBAL_NO_SELECT=1 (tests): no selecting partition pass (MsdPass1KeysSel); the routed / scanned forms as before
  bool route = true;           // DC3HIP_GLOBAL_NO_ROUTE=1: every rank evaluates all positions and keeps its key range (the round-2 form)
  dc3hip_ctx *c = nullptr;
  GComm *comm = nullptr;
  int64_t max_total = 0, total_n = 0;
  bool text_set = false, built = false;
  int64_t shard_first = 0, shard_count = 0;     // this rank holds SA[shard_first .. shard_first + shard_count)
  const u32 *shard_ptr = nullptr;               // device
  u32 local_max = 1u << 22;                     // levels up to this length are finished on every rank redundantly
  bool no_text_order = false;
  bool force_dist = false;                      // run the distributed path even with one rank (transport tests)
  // wide mode (texts beyond DC3HIP_MAX_N, or DC3HIP_GLOBAL_FORCE_WIDE=1): 64-bit positions, whole-text order only
  bool wide = false;
  uint8_t *w_text = nullptr;                    // max_total + 64 bytes (the context's own text buffer is not used)
  // wide mode: records of this rank's image range (w_ra / w_rb: pack / partition / sort buffers) and its shard of 64-bit
  // positions; every array with its own capacity (records / words)
  Rec16 *w_ra = nullptr, *w_rb = nullptr; u64 *w_shard = nullptr;
  uint8_t *w_same = nullptr;            // one byte per word of the bucket ordering: same image as the word before
  size_t w_cap_a = 0, w_cap_b = 0, w_cap_s = 0, w_cap_same = 0;
  // wide mode, deepening by rank look-ups (wide_deepen): the whole order, its equal-window flags and its inverse on every rank
  // (w_sa_all: one rank's shard at a time while the inverse is built — the whole order is never held, round 5)
  u64 *w_sa_all = nullptr, *w_isa = nullptr; uint8_t *w_eq_all = nullptr, *w_eq2 = nullptr;
  size_t w_cap_sa = 0, w_cap_isa = 0, w_cap_eq = 0, w_cap_eq2 = 0;
  // groups beyond kWideTieBig members (a run of one symbol, a short period): group starts of the shard (w_aux, 4 bytes per
  // entry) and the compacted members with their sort records (w_aux2, 48 bytes per member of such a group)
  unsigned char *w_aux = nullptr, *w_aux2 = nullptr;
  size_t w_cap_aux = 0, w_cap_aux2 = 0;
  bool w_isa_valid = false;             // the last build ended with w_isa = the exact inverse of the order (the verifier uses it)
  bool no_wide_deepen = false;          // DC3HIP_NO_WIDE_DEEPEN=1 (tests): windows that repeat beyond the symbol compares' budget are refused, as before round 4
  dc3hip_gstats gs;
  char err[512] = "";
  std::vector<dc3hip_gctx *> group;             // loopback: all ranks of the group (rank 0 owns the list)
};

// SELECT (every rank walks all positions of the replicated string and keeps its key range: nothing is routed) or ROUTE (every
// rank packs its own block and sends each 8-byte record to its owner) — by the per-rank cost of the two forms on P GPUs, in
// ms per GiB of the level's string (MI355X, profiles/r04*): the selecting count + partition pass 1 walk ALL positions,
// 4.46; packing and partitioning a rank's own block costs 4.5 / P, and the all-to-all puts 8 / P^2 bytes per position on each
// link.  On xGMI (153 GB/s per link) that is select up to 4 ranks and route beyond (8 ranks: 4.46 against 0.56 + 0.88);
// ranks that share one device (loopback) have no link to pay and select.  Both forms give the same array and are tested.
static bool gselect_pays(const dc3hip_gctx *G, int P) {
  const double link = G->comm->link_GBps();
  if (link <= 0 || P <= 1) return true;
  const double walk = 4.46, pack = 4.5;
  const double xfer = 8.0 * 1073741824.0 / (link * 1e9) * 1e3;       // ms for 8 bytes per position of one GiB over one link
  return walk <= pack / P + xfer / ((double)P * P);
}

static void block_of(int64_t n, int P, int r, int64_t *off, int64_t *len) {
  const int64_t S = n / P + 1;                  // sacapart/src/lib.rs:43
  const int64_t o = std::min<int64_t>(n, (int64_t)r * S);
  *off = o; *len = std::min<int64_t>(S, n - o);
}

// counts of `bytes` per rank -> this rank's prefix and the total
static int gather_counts(GComm *cm, uint64_t mine, uint64_t *prefix, uint64_t *total, uint64_t *all = nullptr) {
  uint64_t buf[kMaxRanks];
  RC(cm->all_gather_host(&mine, buf, sizeof(uint64_t)));
  uint64_t pre = 0, tot = 0;
  for (int r = 0; r < cm->nranks; r++) { if (r < cm->rank) pre += buf[r]; tot += buf[r]; if (all) all[r] = buf[r]; }
  *prefix = pre; *total = tot;
  return E_OK;
}

// order-preserving selection; *out is allocated from the arena.  One evaluation of the selector per item when the
// arena has room for the chunk-local staging array (k_sel_stage / scan / k_sel_copy), else count / scan / write.
template <class Sel>
static int select_records(dc3hip_ctx *c, const Sel &sel, u32 nitems, typename Sel::Out **out, u32 *count, int phase) {
  typedef typename Sel::Out Out;
  const Chunking ck = make_chunks(c, nitems, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  const size_t stage_bytes = align_up((size_t)nitems * sizeof(Out), 256);
  // staged form needs the staging array ABOVE the result (it is released afterwards), so the result is placed first with
  // its worst-case size only when that is affordable; otherwise the two-evaluation form
  const bool staged = c->arena_bytes - c->arena_off >= 2 * stage_bytes + (64u << 20);
  if (staged) {
    Out *res = nullptr, *stage = nullptr;
    RC(arena_alloc(c, (size_t)nitems + 16, &res));         // shrunk to the real count below
    const ArenaMark mk_stage = arena_mark(c);
    RC(arena_alloc(c, (size_t)nitems, &stage));
    {
      PhaseScope ps(c, phase, nitems);
      hipLaunchKernelGGL((k_sel_stage<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, stage, counts);
      KCHECK();
      hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 32);
      KCHECK();
      HIPC(hipMemcpyAsync(c->h_words + 32, c->d_words + 32, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    }
    HIPC(hipStreamSynchronize(c->stream));
    *count = c->h_words[32];
    if (*count) {
      PhaseScope ps(c, phase, *count);
      hipLaunchKernelGGL((k_sel_copy<Out>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, stage, ck.chunk, ck.nchunks, counts,
                         *count, res);
      KCHECK();
    }
    // give back the staging array and the unused tail of the result
    arena_release(c, mk_stage);
    c->arena_off = (size_t)(reinterpret_cast<unsigned char *>(res) - c->arena) + align_up(((size_t)*count + 16) * sizeof(Out), 256);
    *out = res;
    return E_OK;
  }
  {
    PhaseScope ps(c, phase, nitems);
    hipLaunchKernelGGL((k_sel_count<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 32);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 32, c->d_words + 32, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  *count = c->h_words[32];
  RC(arena_alloc(c, (size_t)*count + 16, out));
  if (*count) {
    PhaseScope ps(c, phase, nitems);
    hipLaunchKernelGGL((k_sel_write<Sel>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, sel, nitems, ck.chunk, counts, *out);
    KCHECK();
  }
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// RANK EXCHANGE: every rank holds `cnt` (destination, value) pairs; all destinations together are a bijection onto
// [0, M).  On return out[0..M) is complete on EVERY rank.
//   1. local partition of the pairs by destination digit (<= 256 digits of >= 2^14 destinations; rank h owns a
//      contiguous digit range) — one stable radix pass, its digit table gives the send offsets;
//   2. all-to-all: pairs to the owner of their destination            (8 B x cnt x (P-1)/P per rank over xGMI)
//   3. the owner builds its block by the windowed inversion (inverse_permute)
//   4. all-gather of the blocks                                        (4 B x M x (P-1)/P per rank over xGMI)
// ---------------------------------------------------------------------------------------------
static int rank_exchange(dc3hip_gctx *G, Rec8 *pairs, u32 cnt, u32 M, u32 *out, int phase) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const ArenaMark mk = arena_mark(c);
  u32 sh = (u32)kInvWindowBits;
  while ((((u64)M - 1) >> sh) + 1 > 256) sh++;
  const u32 nd = (u32)((((u64)M - 1) >> sh) + 1);
  auto dlo = [&](int h) { return (u32)(((u64)h * nd + P - 1) / P); };       // first digit of rank h
  auto dest_lo = [&](int h) { return (u64)std::min<u64>((u64)M, (u64)dlo(h) << sh); };
  // 1. partition by digit
  u32 hdb[257];
  Rec8 *sorted = pairs;
  for (u32 d = 0; d <= 256; d++) hdb[d] = 0;
  if (cnt) {
    constexpr int kTile = SortCfg<Rec8, 256>::NW * 64 * SortCfg<Rec8, 256>::IPT;
    const Chunking ck = make_chunks(c, cnt, kTile);
    u32 *table = nullptr, *digit_base = nullptr;
    Rec8 *pb = nullptr;
    RC(arena_alloc(c, (size_t)256 * ck.nchunks, &table));
    RC(arena_alloc(c, (size_t)256, &digit_base));
    RC(arena_alloc(c, (size_t)cnt, &pb));
    KeyDig dig; dig.shift = 32 + sh; dig.mask = 255;
    {
      PhaseScope ps(c, phase, cnt);
      hipLaunchKernelGGL((k_rs_upsweep<Rec8, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, pairs, cnt, ck.chunk,
                         ck.nchunks, dig, table);
      KCHECK();
    }
    RC(scan_digit_table(c, table, ck.nchunks, digit_base, 256, phase));
    void *tmpp = nullptr;
    RC(stage_d2h_async(c, digit_base, 256 * sizeof(u32), &tmpp));
    const u32 *tmp = static_cast<const u32 *>(tmpp);
    ArrayLoader<Rec8> ld; ld.p = pairs;
    RC((launch_downsweep<Rec8, 256, ArrayLoader<Rec8>>(c, ld, pb, cnt, ck, dig, table, digit_base, phase)));
    HIPC(hipStreamSynchronize(c->stream));
    for (u32 d = 0; d < 256; d++) hdb[d] = tmp[d];
    hdb[256] = cnt;
    for (u32 d = nd; d < 256; d++) hdb[d] = cnt;
    sorted = pb;
  }
  // 2. all-to-all
  size_t soff[kMaxRanks], sbytes[kMaxRanks], roff[kMaxRanks], rbytes[kMaxRanks];
  uint64_t scount[kMaxRanks], mat[kMaxRanks * kMaxRanks];
  for (int h = 0; h < P; h++) {
    const u32 a = hdb[std::min<u32>(dlo(h), 256)], b = hdb[std::min<u32>(dlo(h + 1), 256)];
    soff[h] = (size_t)a * sizeof(Rec8); sbytes[h] = (size_t)(b - a) * sizeof(Rec8); scount[h] = b - a;
  }
  RC(cm->all_gather_host(scount, mat, sizeof(uint64_t) * (size_t)P));
  const u64 base = dest_lo(me), myblk = dest_lo(me + 1) - base;
  u64 got = 0;
  for (int r = 0; r < P; r++) { roff[r] = (size_t)got * sizeof(Rec8); rbytes[r] = (size_t)mat[(size_t)r * P + me] * sizeof(Rec8); got += mat[(size_t)r * P + me]; }
  if (got != myblk) { set_err("rank exchange: block of rank %d expects %llu pairs, received %llu (destinations are not a bijection)", me, (unsigned long long)myblk, (unsigned long long)got); return E_HIP; }
  Rec8 *rb = nullptr, *rt = nullptr;
  RC(arena_alloc(c, (size_t)myblk + 16, &rb));
  RC(arena_alloc(c, (size_t)myblk + 16, &rt));
  RC(cm->all_to_all_v(sorted, soff, sbytes, rb, roff, rbytes, c->stream));
  // 3. my block
  if (myblk) {
    if (base) {
      PhaseScope ps(c, phase, myblk);
      hipLaunchKernelGGL(k_rebase_keys, dim3(grid_for(c, myblk)), dim3(kBlock), 0, c->stream, rb, (u32)myblk, (u32)base);
      KCHECK();
    }
    RC(inverse_permute(c, rb, rt, (u32)myblk, out + base, phase));
  }
  // 4. all-gather of the blocks, in place
  size_t goff[kMaxRanks], gbytes[kMaxRanks];
  for (int r = 0; r < P; r++) { goff[r] = (size_t)dest_lo(r) * 4; gbytes[r] = (size_t)(dest_lo(r + 1) - dest_lo(r)) * 4; }
  RC(cm->all_gather_v(out + base, (size_t)myblk * 4, out, goff, gbytes, c->stream));
  G->gs.exchanges += 1;
  G->gs.exchange_pairs += cnt;
  arena_release(c, mk);
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// where a level's result goes
// ---------------------------------------------------------------------------------------------
enum GOut {
  G_TOP = 0,    // level 0: this rank's slice of the suffix array stays in c->d_sa (G->shard_*)
  G_RANK = 1,   // out[pos] = 1-based rank of suffix pos, complete on every rank (the parent's rank12)
  G_SA = 2      // out[k] = position of the k-th smallest suffix, complete on every rank (discarding parent)
};
// slice[0..cnt) = this rank's part of the level's suffix array, starting at global index `pre`
static int deliver(dc3hip_gctx *G, const u32 *slice, u32 cnt, u64 pre, const uint64_t *all, u32 m, u32 *out, GOut mode) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  if (mode == G_TOP) {
    G->shard_first = (int64_t)pre; G->shard_count = cnt; G->shard_ptr = slice;
    return E_OK;
  }
  if (mode == G_RANK) {
    const ArenaMark mk = arena_mark(c);
    Rec8 *pp = nullptr;
    RC(arena_alloc(c, (size_t)cnt + 16, &pp));
    if (cnt) {
      PhaseScope ps(c, DC3HIP_PH_RANKS, cnt);
      hipLaunchKernelGGL(k_sa_to_pairs, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, slice, cnt, (u32)pre, pp);
      KCHECK();
    }
    RC(rank_exchange(G, pp, cnt, m, out, DC3HIP_PH_RANKS));     // rank[pos] = global index + 1, everywhere
    arena_release(c, mk);
    return E_OK;
  }
  size_t roff[kMaxRanks], rbytes[kMaxRanks];
  u64 o = 0;
  for (int r = 0; r < cm->nranks; r++) { roff[r] = (size_t)o * 4; rbytes[r] = (size_t)all[r] * 4; o += all[r]; }
  return cm->all_gather_v(slice, (size_t)cnt * 4, out, roff, rbytes, c->stream);
}

// splitters of a 64-bit image order from ns sampled records ((image << pbits) | pos): every rank computes the same
static int image_splitters(dc3hip_ctx *c, const Rec8 *d_sample, u32 ns, u32 pbits, int P, int me, u64 *lo, u64 *hi) {
  void *hsp = nullptr;
  RC(stage_d2h(c, d_sample, (size_t)ns * sizeof(Rec8), &hsp));
  const Rec8 *hs = static_cast<const Rec8 *>(hsp);
  // a few thousand candidates per rank are plenty (a host sort of the whole 2^20-record predictor sample cost 60 ms)
  const u32 step = std::max<u32>(1, ns / (u32)(4096 * P));
  std::vector<u64> img;
  img.reserve(ns / step + 1);
  for (u32 i = 0; i < ns; i += step) img.push_back(((((u64)hs[i].key) << 32) | hs[i].val) >> pbits);
  std::sort(img.begin(), img.end());
  const size_t k = img.size();
  *lo = 0; *hi = ~0ull;
  if (me > 0) *lo = img[(size_t)((u64)me * k / P)];
  if (me + 1 < P) *hi = img[(size_t)((u64)(me + 1) * k / P)];
  return E_OK;
}

// Pass 1 of the bucket ordering that SELECTS (k_msd_part_keys<.., kSel>): the rank walks the replicated text / level
// string, makes every position's image (as the single device's pass 1 does) and partitions the words of its image
// range only.  Everything behind it — bucket sizes, pass 2, local order, tie pass — is the single device's code on a
// P-th of the words.  m = positions walked.
template <class KM>
struct MsdPass1KeysSel : MsdPass1Keys<KM> {
  MsdSel sel{0, 0, 1, 0}; u32 m = 0;
  // (the ordering gave up before pass 2: the plain words of the selection, in position order, for the LSD passes)
  int repack(dc3hip_ctx *c, Rec8 *out, u32 nrec, u32 **first_table) override {
    SelPosImageW<KM> s; s.km = this->km; s.hm = this->hm; s.sel = sel; s.pbits = this->hm.pbits + sel.sh;
    Rec8 *tmp = nullptr; u32 cnt = 0;
    RC(select_records(c, s, m, &tmp, &cnt, DC3HIP_PH_PACK));
    if (cnt != nrec) { set_err("internal: the selection repacked %u words of %u", cnt, nrec); return E_HIP; }
    HIPC(hipMemcpyAsync(out, tmp, (size_t)cnt * sizeof(Rec8), hipMemcpyDeviceToDevice, c->stream));
    *first_table = nullptr;
    return E_OK;
  }
  int launch(dc3hip_ctx *c, u64 *out, u32, u64 base, u32 sh1, const MsdGeom &g, u32 nb1, const u32 *plan, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_msd_part_keys<KM, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMsdPartSmem));
      attr_set[c->device & 15] = true;
    }
    if (this->strip && (this->hm.nbits + this->hm.pbits != 64 || g.d1 == 0)) { set_err("internal: a stripped image must fill the word"); return E_HIP; }
    if (this->strip)
      hipLaunchKernelGGL((k_msd_part_keys<KM, true, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, this->km, this->hm,
                         this->P1, out, m, base, sh1, g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon, sel);
    else
      hipLaunchKernelGGL((k_msd_part_keys<KM, false, true>), dim3(kMsdGroups * g.cpx1), dim3(kMsdNW * 64), kMsdPartSmem, c->stream, this->km, this->hm,
                         this->P1, out, m, base, sh1, g.d1, g.cpx1, g.ntiles1, plan, cur1, nb1, c->d_xcdmon, sel);
    KCHECK();
    return E_OK;
  }
};

// ---------------------------------------------------------------------------------------------
// Whole-level order, split by key range (the distributed form of order_all_positions): every rank orders the positions
// p in [0, m) whose key image falls into its range by prefix sort + tie refinement.  If every key on every rank is
// distinct the concatenated slices ARE the level's suffix array (suffixes differ inside the key: 9 bytes of text for
// Key9 at level 0, a K-S triple for Key3 below), *done = true and the result has been delivered; otherwise nothing
// was produced.  emit: where this rank's slice goes (c->d_sa at the top; nullptr = arena, valid until the caller's mark
// is released).
// ---------------------------------------------------------------------------------------------
template <class KM>
static int gorder_positions(dc3hip_gctx *G, KM km, u32 m, u32 kbits, const HiMap &hm, int depth, u32 *out, GOut mode,
                            bool *done) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  *done = false;
  const ArenaMark mk = arena_mark(c);
  // the slice first (worst case: every position in my range), so that the sort's temporaries can be released before
  // the result is delivered (the rank exchange needs the room)
  u32 *slice = (mode == G_TOP) ? c->d_sa : nullptr;
  if (!slice) RC(arena_alloc(c, (size_t)m + 16, &slice));
  const ArenaMark mk_tmp = arena_mark(c);
  Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
  u32 nrec = 0;
  u64 img_lo = 0, img_span = 0;
  // SELECTED (default up to 16 ranks, key makers whose image pass 1 of the bucket ordering can make itself): no records
  // are built or sent at all — see MsdPass1KeysSel.  A rank reads the m positions twice (count, partition) and orders
  // m / P words; on one GPU shared by P loopback ranks that is the least total work of the three forms, and on P GPUs
  // the walk (HBM rate) costs less than routing 8 m / P bytes over xGMI.
  constexpr bool kFusable = std::is_same<KM, Key9>::value || std::is_same<KM, Key3<SymU32>>::value;
  typename std::conditional<kFusable, MsdPass1KeysSel<KM>, MsdPass1Keys<KM>>::type psel;     // (the selecting kernels only where they are used)
  MsdGeom mgx;
  u32 *sel_table = nullptr;
  bool selected = false;
  if constexpr (kFusable) {
    const MsdGeom mg = msd_geometry(c, m, hm);
    if (!G->no_select && mg.on && gselect_pays(G, P)) {
      u64 lo = 0, hi = ~0ull;
      {
        u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
        const u32 stride = std::max<u32>(1, m / ns);
        ns = (m - 1) / stride + 1;
        Rec8 *smp = nullptr;
        RC(arena_alloc(c, (size_t)ns, &smp));
        hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
        KCHECK();
        RC(image_splitters(c, smp, ns, hm.pbits, P, me, &lo, &hi));
      }
      mgx = mg;
      psel.km = km; psel.hm = hm; psel.P1 = 0; psel.m = m;
      if (!c->no_pack_strip && hm.pbits >= 23 && !hm.exact && kbits >= hm.nbits + mg.d1) {      // (as order_all_positions)
        u64 limb = 0;
        if constexpr (std::is_same<KM, Key9>::value) limb = km.B3; else limb = km.B;
        psel.strip = true; psel.hm_plain = hm;
        psel.hm = make_himap(limb, kbits, m, hm.pbits - mg.d1);
        psel.hm.raw = hm.raw;
        mgx.ebits = psel.hm.nbits;
      }
      psel.sel = MsdSel{lo, hi, (me + 1 == P) ? 1u : 0u, psel.strip ? mg.d1 : 0u};
      RC(arena_alloc(c, (size_t)kMsdMaxDig * mg.ck.nchunks, &sel_table));
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, m);
        HIPC(hipMemsetAsync(c->d_words + 33, 0, sizeof(u32), c->stream));
        hipLaunchKernelGGL((k_msd_count_sel<KM>), dim3(mg.ck.nchunks), dim3(kBlock), 0, c->stream, km, psel.hm, 0ull, m, psel.sel, mg.ck.chunk,
                           mg.ck.nchunks, sel_table, psel.hm.nbits - mg.d1, c->d_words + 33);
        KCHECK();
        HIPC(hipMemcpyAsync(c->h_words + 33, c->d_words + 33, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      }
      HIPC(hipStreamSynchronize(c->stream));
      nrec = c->h_words[33];
      RC(arena_alloc(c, (size_t)nrec + 16, &ha));            // (pass 1 makes the words: scratch of pass 2)
      selected = true;
      G->gs.select_p1 += 1;
    }
  }
  if (selected) {
  } else if (G->route && hm.nbits >= 8) {
    // ROUTED (default): every rank packs the records of ITS block of positions only (m / P of them), partitions them by
    // the top 8 image bits — rank h owns a contiguous digit range, chosen from a replicated sample so that the ranges hold
    // about m / P records each — and sends every record to its owner: one all-to-all of 8-byte records
    // (8 m (P-1) / P^2 bytes out per rank).  Work per rank is O(m / P); SURVEY.md §8(e) step 2.
    u32 dlo[kMaxRanks + 1];
    {
      u32 ns = (u32)std::min<u64>(m, (u64)4096 * P);
      const u32 stride = std::max<u32>(1, m / ns);
      ns = (m - 1) / stride + 1;
      Rec8 *smp = nullptr;
      RC(arena_alloc(c, (size_t)ns, &smp));
      hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
      KCHECK();
      void *hsp = nullptr;
      RC(stage_d2h(c, smp, (size_t)ns * sizeof(Rec8), &hsp));
      const Rec8 *hs = static_cast<const Rec8 *>(hsp);
      u32 cnt256[257] = {0};
      for (u32 i = 0; i < ns; i++) cnt256[(u32)((((((u64)hs[i].key) << 32) | hs[i].val) >> (hm.pbits + hm.nbits - 8)) & 255u)]++;
      // boundaries: rank h starts at the first digit whose prefix count reaches h * ns / P (identical on all ranks)
      dlo[0] = 0; dlo[P] = 256;
      u32 acc = 0, hnext = 1;
      for (u32 d = 0; d < 256 && hnext < (u32)P; d++) {
        while (hnext < (u32)P && (u64)acc * P >= (u64)hnext * ns) dlo[hnext++] = d;
        acc += cnt256[d];
      }
      while (hnext < (u32)P) dlo[hnext++] = 256;
      for (int r = 1; r <= P; r++) dlo[r] = std::max(dlo[r], dlo[r - 1]);
    }
    const u32 boff = (u32)((u64)m * me / P), blen = (u32)((u64)m * (me + 1) / P) - boff;
    Rec8 *mine = nullptr, *sorted = nullptr;
    RC(arena_alloc(c, (size_t)blen + 16, &mine));
    u32 hdb[257];
    for (u32 d = 0; d <= 256; d++) hdb[d] = 0;
    if (blen) {
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, blen);
        hipLaunchKernelGGL((k_pack_image_range<KM>), dim3(grid_for(c, blen)), dim3(kBlock), 0, c->stream, km, boff, blen, hm, mine);
        KCHECK();
      }
      constexpr int kTile = SortCfg<Rec8, 256>::NW * 64 * SortCfg<Rec8, 256>::IPT;
      const Chunking ck = make_chunks(c, blen, kTile);
      u32 *table = nullptr, *digit_base = nullptr;
      RC(arena_alloc(c, (size_t)256 * ck.nchunks, &table));
      RC(arena_alloc(c, (size_t)256, &digit_base));
      RC(arena_alloc(c, (size_t)blen + 16, &sorted));
      KeyDig dig; dig.shift = hm.pbits + hm.nbits - 8; dig.mask = 255;
      {
        PhaseScope ps(c, DC3HIP_PH_PACK, blen);
        hipLaunchKernelGGL((k_rs_upsweep<Rec8, 256>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, mine, blen, ck.chunk, ck.nchunks, dig, table);
        KCHECK();
      }
      RC(scan_digit_table(c, table, ck.nchunks, digit_base, 256, DC3HIP_PH_PACK));
      void *tmpp = nullptr;
      RC(stage_d2h_async(c, digit_base, 256 * sizeof(u32), &tmpp));
      const u32 *tmp = static_cast<const u32 *>(tmpp);
      ArrayLoader<Rec8> ld; ld.p = mine;
      RC((launch_downsweep<Rec8, 256, ArrayLoader<Rec8>>(c, ld, sorted, blen, ck, dig, table, digit_base, DC3HIP_PH_PACK)));
      HIPC(hipStreamSynchronize(c->stream));
      for (u32 d = 0; d < 256; d++) hdb[d] = tmp[d];
      hdb[256] = blen;
    }
    size_t soff[kMaxRanks], sbytes[kMaxRanks], roff[kMaxRanks], rbytes[kMaxRanks];
    uint64_t scount[kMaxRanks], mat[kMaxRanks * kMaxRanks];
    for (int r = 0; r < P; r++) {
      const u32 a0 = hdb[dlo[r]], b0 = hdb[dlo[r + 1]];
      soff[r] = (size_t)a0 * sizeof(Rec8); sbytes[r] = (size_t)(b0 - a0) * sizeof(Rec8); scount[r] = b0 - a0;
    }
    RC(cm->all_gather_host(scount, mat, sizeof(uint64_t) * (size_t)P));
    u64 got = 0;
    for (int r = 0; r < P; r++) { roff[r] = (size_t)got * sizeof(Rec8); rbytes[r] = (size_t)mat[(size_t)r * P + me] * sizeof(Rec8); got += mat[(size_t)r * P + me]; }
    if (got > (u64)m) { set_err("global order: %llu records routed to rank %d of a level of %u", (unsigned long long)got, me, m); return E_HIP; }
    nrec = (u32)got;
    RC(arena_alloc(c, (size_t)nrec + 16, &ha));
    RC(cm->all_to_all_v(sorted ? sorted : mine, soff, sbytes, ha, roff, rbytes, c->stream));
    G->gs.exchanges += 1;
    img_lo = (u64)dlo[me] << (hm.nbits - 8);
    img_span = (u64)(dlo[me + 1] - dlo[me]) << (hm.nbits - 8);
  } else {
    u64 lo = 0, hi = ~0ull;
    {
      u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
      const u32 stride = std::max<u32>(1, m / ns);
      ns = (m - 1) / stride + 1;
      Rec8 *smp = nullptr;
      RC(arena_alloc(c, (size_t)ns, &smp));
      hipLaunchKernelGGL((k_pack_image_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
      KCHECK();
      RC(image_splitters(c, smp, ns, hm.pbits, P, me, &lo, &hi));
    }
    SelPosImage<KM> sel; sel.km = km; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
    RC(select_records(c, sel, m, &ha, &nrec, DC3HIP_PH_PACK));
  }
  RC(arena_alloc(c, (size_t)nrec + 16, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  bool ok = true, distinct = true;
  if (nrec && selected)
    RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &ok, depth, slice, 0, &distinct, sel_table, false, &mgx, 0, 0, &psel, nullptr, true)));
  else if (nrec)
    RC((hybrid_sort_core<KM>(c, km, kbits, hm, ha, hb, nrec, &h, f, &ok, depth, slice, 0, &distinct, nullptr, false, nullptr,
                             img_lo, img_span, nullptr, nullptr, true)));      // (slots: the routed records fill their image range evenly)
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, (ok && distinct) ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != m) { set_err("global order: %llu of %u positions selected", (unsigned long long)tot, m); return E_HIP; }
  arena_release(c, mk_tmp);
  if (ngood == (uint64_t)P) {
    *done = true;
    RC(deliver(G, slice, nrec, pre, all, m, out, mode));
  }
  arena_release(c, mk);
  return E_OK;
}

// The whole-text order on 12-byte records (try_text_order12's distributed form; top level only): hm maps KM's image to
// hm.nbits <= 63 bits, sorted in full; the tie pass writes the slice.
template <class KM>
static int gorder_positions12(dc3hip_gctx *G, KM km, u32 m, u32 kbits, const HiMap &hm, bool *done) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  *done = false;
  const ArenaMark mk = arena_mark(c);
  u32 *slice = c->d_sa;
  u64 lo = 0, hi = ~0ull;
  {
    u32 ns = (u32)std::min<u64>(m, (u64)2048 * P);
    const u32 stride = std::max<u32>(1, m / ns);
    ns = (m - 1) / stride + 1;
    Rec8 *smp = nullptr;
    RC(arena_alloc(c, (size_t)ns, &smp));
    hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, smp);
    KCHECK();
    RC(image_splitters(c, smp, ns, 1u, P, me, &lo, &hi));
  }
  SelPosImage12<KM> sel; sel.km = km; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
  Rec12 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
  u32 nrec = 0;
  RC(select_records(c, sel, m, &ha, &nrec, DC3HIP_PH_PACK));
  RC(arena_alloc(c, (size_t)nrec + 16, &hb));
  RC(arena_alloc(c, (size_t)nrec + 16, &f));
  bool ok = true, distinct = true;
  if (nrec) {
    RC(radix_sort<Rec12>(c, ha, hb, nrec, 0, hm.nbits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    RC((hybrid12_refine<KM>(c, km, kbits, h, nrec, f, &ok, 0, slice, &distinct)));
  }
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, (ok && distinct) ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != m) { set_err("global order: %llu of %u positions selected", (unsigned long long)tot, m); return E_HIP; }
  arena_release(c, mk);
  if (ngood == (uint64_t)P) {
    *done = true;
    RC(deliver(G, slice, nrec, pre, all, m, nullptr, G_TOP));
  }
  return E_OK;
}

// ---------------------------------------------------------------------------------------------
// one level (lib.rs:44-193) on replicated S; the result goes where `mode` says (GOut).
// ---------------------------------------------------------------------------------------------
template <class Sym>
static int glevel(dc3hip_gctx *G, Sym S, u32 m, u64 K, int depth, u32 *out, GOut mode);

// Sorted naming of this rank's key range (lib.rs:80-100), generic over the accessor of the sorted order.
//   counts of distinct / unique names go around (all-gather of two words), names continue after those of the smaller
//   key ranges (equal keys never straddle ranks); the (slot, name [| unique << 31]) pairs land in the caller's buffer
//   pa (m02 entries), which the caller exchanges into R AFTER releasing its sort buffers.
template <class Acc0>
static int gname_pairs(dc3hip_gctx *G, Acc0 acc0, u32 cnt, u32 m0, u32 m02, Rec8 *pa, u32 *sslot, uint64_t *names_total,
                       uint64_t *uniq_total, uint64_t *cnt_pre, bool *discard, bool first_eq = false, bool last_eq_next = false) {
  typedef AccBound<Acc0> Acc;
  Acc acc; acc.a = acc0; acc.first_eq = first_eq ? 1u : 0u; acc.last_eq_next = last_eq_next ? 1u : 0u;
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, std::max<u32>(cnt, 1), kBlock * kNameIPT);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  u32 distinct = 0, uniq = 0;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_NAMING, cnt);
    HIPC(hipMemsetAsync(c->d_words + 4, 0, sizeof(u32), c->stream));
    hipLaunchKernelGGL((k_name_count<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, cnt, ck.chunk, counts, c->d_words + 4);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words, c->d_words, 5 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    distinct = c->h_words[0]; uniq = c->h_words[4];
  }
  uint64_t name_off = 0, upre = 0, cnt_total = 0;
  RC(gather_counts(cm, distinct, &name_off, names_total));
  RC(gather_counts(cm, uniq, &upre, uniq_total));
  RC(gather_counts(cm, cnt, cnt_pre, &cnt_total));
  if (cnt_total != m02) { set_err("global naming: %llu of %u samples selected", (unsigned long long)cnt_total, m02); return E_HIP; }
  // discarding (see discard_recurse): worth it when ~1/6 of the slots would leave the recursion
  const double drop_est = (double)*uniq_total * (double)*uniq_total / (double)m02;
  *discard = sslot && *names_total != m02 && !c->no_discard && m02 < 0x7fffffffu && drop_est * kDiscardMinDropInv >= (double)m02;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_NAMING, cnt);
    if (name_off) {
      hipLaunchKernelGGL(k_add_scalar, dim3(grid_for(c, ck.nchunks)), dim3(kBlock), 0, c->stream, counts, ck.nchunks, (u32)name_off);
      KCHECK();
    }
    hipLaunchKernelGGL((k_name_assign<Acc>), dim3(ck.nchunks), dim3(kBlock), 0, c->stream, acc, cnt, ck.chunk, counts, m0, pa,
                       *discard ? sslot : (u32 *)nullptr);
    KCHECK();
  }
  arena_release(c, mk);
  return E_OK;
}

// Discarding recursion, distributed (the scheme of discard_recurse).  RU[p] = name | unique << 31 is replicated, so the
// reduced string R' and the kept-slot list are built by every rank (streaming); the child returns its suffix array
// REPLICATED (all-gather of slices: 4 B per kept slot — cheaper than a rank exchange); every rank derives the order of
// the non-unique slots (pt) from it and rewrites ITS range of the sorted array; one rank exchange gives rank12.
static int gdiscard(dc3hip_gctx *G, const u32 *RU, const u32 *sslot, u32 cnt, u64 cnt_pre, u32 m02, u64 names, u32 *rank12,
                    int depth) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const ArenaMark mk = arena_mark(c);
  const Chunking ck = make_chunks(c, m02, kBlock);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ck.nchunks + 16, &counts));
  {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, m02);
    hipLaunchKernelGGL(k_keep_count, dim3(ck.nchunks), dim3(kBlock), 0, c->stream, RU, m02, ck.chunk, counts);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ck.nchunks, c->d_words + 5);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 5, c->d_words + 5, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(hipStreamSynchronize(c->stream));
  const u32 mp = c->h_words[5];
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
  if (mp == 1) { hipLaunchKernelGGL(k_base1, dim3(1), dim3(64), 0, c->stream, sap, (u32 *)nullptr); KCHECK(); }
  else RC(glevel<SymU32>(G, RS, mp, names, depth + 1, sap, G_SA));
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
  // my range [cnt_pre, cnt_pre + cnt) of the level's sorted array: unique entries keep their place, the t-th
  // non-unique entry (t counted over all ranks) receives pt[t]
  Rec8 *pa = nullptr;
  RC(arena_alloc(c, (size_t)cnt + 16, &pa));
  const Chunking ckl = make_chunks(c, std::max<u32>(cnt, 1), kBlock);
  u32 *cl = nullptr;
  RC(arena_alloc(c, (size_t)ckl.nchunks + 16, &cl));
  u32 nu_local = 0;
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, cnt);
    hipLaunchKernelGGL(k_nonuniq_count, dim3(ckl.nchunks), dim3(kBlock), 0, c->stream, sslot, cnt, ckl.chunk, cl);
    KCHECK();
    hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, cl, ckl.nchunks, c->d_words + 6);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 6, c->d_words + 6, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    nu_local = c->h_words[6];
  }
  uint64_t nu_pre = 0, nu_tot = 0;
  RC(gather_counts(cm, nu_local, &nu_pre, &nu_tot));
  if (cnt) {
    PhaseScope ps(c, DC3HIP_PH_DISCARD, cnt);
    if (nu_pre) {
      hipLaunchKernelGGL(k_add_scalar, dim3(grid_for(c, ckl.nchunks)), dim3(kBlock), 0, c->stream, cl, ckl.nchunks, (u32)nu_pre);
      KCHECK();
    }
    hipLaunchKernelGGL(k_final_assign, dim3(ckl.nchunks), dim3(kBlock), 0, c->stream, sslot, cnt, ckl.chunk, cl, pt, (u32 *)nullptr, pa);
    KCHECK();
    if (cnt_pre) {
      hipLaunchKernelGGL(k_add_val, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, pa, cnt, (u32)cnt_pre);
      KCHECK();
    }
  }
  RC(rank_exchange(G, pa, cnt, m02, rank12, DC3HIP_PH_RANKS));
  arena_release(c, mk);
  return E_OK;
}

template <class Sym>
static int glevel(dc3hip_gctx *G, Sym S, u32 m, u64 K, int depth, u32 *out, GOut mode) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  if (depth >= DC3HIP_MAX_LEVELS) { set_err("recursion deeper than %d levels", DC3HIP_MAX_LEVELS); return E_HIP; }
  if (m <= G->local_max || m < 64) {
    // small level: every rank finishes the recursion on its own copy (no communication below this point)
    if (G->gs.local_from_level < 0) G->gs.local_from_level = depth;
    if (mode == G_TOP) {
      RC(dc3_level<Sym>(c, S, m, K, c->d_sa, nullptr, depth));
      int64_t off, len; block_of(m, P, me, &off, &len);
      G->shard_first = off; G->shard_count = len; G->shard_ptr = c->d_sa + off;
    } else if (mode == G_RANK) {
      RC(dc3_level<Sym>(c, S, m, K, nullptr, out, depth));
    } else {
      RC(dc3_level<Sym>(c, S, m, K, out, nullptr, depth));
    }
    return E_OK;
  }
  const u32 m0 = (m + 2) / 3, m1 = (m + 1) / 3, m2 = m / 3, m02 = m0 + m2;   // lib.rs:45-48
  c->stats.level_n[depth] = m; c->stats.level_K[depth] = (int64_t)K; c->stats.levels = depth + 1;
  const ArenaMark mk0 = arena_mark(c);
  u32 *R = nullptr, *rank12 = nullptr;
  RC(arena_alloc(c, (size_t)m02 + 16, &R));
  const u64 B = K + 1;
  const bool direct = (B * B * B) <= 0x7fffffffull;
  if (direct) {
    // order-preserving packed-triple names, computed by every rank for the whole level (one streaming pass)
    c->stats.level_sorted[depth] = 0; c->stats.level_name_width[depth] = 3;
    {
      PhaseScope ps(c, DC3HIP_PH_NAME_DIRECT, m02);
      hipLaunchKernelGGL((k_name_direct<Sym>), dim3(grid_for(c, m0)), dim3(kBlock), 0, c->stream, S, m, m0, m02, (u32)B, 3u,
                         (u32)(B * B), R);
      KCHECK();
    }
    RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
    SymU32 RS; RS.s = R; RS.m = m02;
    RC(glevel<SymU32>(G, RS, m02, B * B * B, depth + 1, rank12, G_RANK));
  } else {
    // ---- sorted naming, split by key range (lib.rs:62-100) ---------------------------------------------------
    c->stats.level_sorted[depth] = 1;
    const u32 b = (u32)B;
    u32 kbits = 0;
    { unsigned __int128 mx = (unsigned __int128)B * B * B - 1; while (mx) { kbits++; mx >>= 1; } }
    u32 *sslot = nullptr;                          // my range of the sorted slots (| unique << 31), for the discarding
    RC(arena_alloc(c, (size_t)m02 + 16, &sslot));
    Rec8 *pa = nullptr;                            // my (slot, name) pairs: below the sort buffers, which go before the exchange
    RC(arena_alloc(c, (size_t)m02 + 16, &pa));
    const ArenaMark mk1 = arena_mark(c);
    const HiMap hm = make_himap(B, kbits, m);
    double pred = 1.0;
    const bool try_hybrid = m02 >= kHybridMinSamples / 4 && !c->no_hybrid && !c->no_hybrid8;
    if (try_hybrid) {
      RC(predict_tie_fraction<Sym>(c, S, m, m0, m02, b, hm, &pred));
      c->stats.level_tie_pred[depth] = pred;
    }
    Key3<Sym> km; km.S = S; km.B = b;
    if (try_hybrid && pred < kFullSortMaxPredicted && !c->no_fullsort) {
      // high-entropy level: if all its triples are distinct, ordering all its positions finishes it
      bool done = false;
      RC((gorder_positions<Key3<Sym>>(G, km, m, kbits, hm, depth, out, mode, &done)));
      if (done) { c->stats.level_sorted[depth] = 5; arena_release(c, mk0); return E_OK; }
      arena_release(c, mk1);
    }
    uint64_t names_total = 0, uniq_total = 0, cnt_pre = 0;
    bool discard = false, named = false;
    u32 cnt = 0;
    if (try_hybrid && pred < c->hybrid_max_pred) {
      // prefix sort + tie refinement of my IMAGE range (equal keys have equal images, so they stay on one rank)
      u64 lo = 0, hi = ~0ull;
      {
        const u32 stride = std::max<u32>(1, m0 / (u32)std::min<u64>(m0, (u64)1024 * P));
        const u32 ng = (m0 - 1) / stride + 1;
        Rec8 *smp = nullptr;
        RC(arena_alloc(c, (size_t)2 * ng, &smp));
        HIPC(hipMemsetAsync(smp, 0xff, (size_t)2 * ng * sizeof(Rec8), c->stream));   // (a missing last mod-2 sample stays a filler)
        hipLaunchKernelGGL((k_pack_image<Sym>), dim3(grid_for(c, ng)), dim3(kBlock), 0, c->stream, S, m, m0, m02, b, hm, stride, ng, smp);
        KCHECK();
        RC(image_splitters(c, smp, 2 * ng, hm.pbits, P, me, &lo, &hi));
      }
      SelSampleImage<Sym> sel; sel.S = S; sel.B = b; sel.hm = hm; sel.lo = lo; sel.hi = hi; sel.last = (me + 1 == P) ? 1u : 0u;
      Rec8 *ha = nullptr, *hb = nullptr, *h = nullptr; uint8_t *f = nullptr;
      RC(select_records(c, sel, m02, &ha, &cnt, DC3HIP_PH_PACK));
      RC(arena_alloc(c, (size_t)cnt + 16, &hb));
      RC(arena_alloc(c, (size_t)cnt + 16, &f));
      bool ok = true;
      if (cnt) RC((hybrid_sort_core<Key3<Sym>>(c, km, kbits, hm, ha, hb, cnt, &h, f, &ok, depth)));
      uint64_t g0 = 0, ngood = 0;
      RC(gather_counts(cm, ok ? 1 : 0, &g0, &ngood));
      if (ngood == (uint64_t)P) {
        c->stats.level_sorted[depth] = 2;
        AccHyb acc; acc.h = h; acc.f = f; acc.posmask = hm.pbits >= 32 ? 0xffffffffu : ((1u << hm.pbits) - 1u);
        RC(gname_pairs<AccHyb>(G, acc, cnt, m0, m02, pa, sslot, &names_total, &uniq_total, &cnt_pre, &discard));
        named = true;
      } else {
        arena_release(c, mk1);                       // too many ties somewhere: every rank takes the straight sort
      }
    }
    if (!named) {
      // straight sort of my KEY range: splitters = every rank sorts the same deterministic sample of keys.
      // Where a rank's share is large enough for the splitter ordering (dc3_ssort.hip.hpp), whose cost does not depend on
      // the key width, the records are W-symbol windows instead of triples (order_wide of the single-device build): more
      // names are distinct, the discarding recursion keeps less, the distributed levels below shrink.
      const u32 Ww = wide_window_syms(c, m02 / (u32)P, K), wsb = bits_of(K);
      const u32 W = Ww > 3 ? Ww : 0u, sort_bits = W ? W * wsb : kbits;
      if (W) c->stats.level_name_width[depth] = (int32_t)W;
      Rec16 klo{0, 0, 0, 0}, khi{0, 0, 0, 0};
      {
        u32 ns = (u32)std::min<u64>(m02, (u64)1024 * P);
        const u32 stride = std::max<u32>(1, m02 / ns);
        ns = (m02 - 1) / stride + 1;
        Rec16 *smp = nullptr;
        RC(arena_alloc(c, (size_t)ns, &smp));
        hipLaunchKernelGGL((k_sample_triple_keys<Sym>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, S, b, ns, stride, smp, W, wsb);
        KCHECK();
        void *hsp = nullptr;
        RC(stage_d2h(c, smp, (size_t)ns * sizeof(Rec16), &hsp));
        std::vector<Rec16> hs(static_cast<const Rec16 *>(hsp), static_cast<const Rec16 *>(hsp) + ns);
        std::sort(hs.begin(), hs.end(), [](const Rec16 &x, const Rec16 &y) {
          if (x.k2 != y.k2) return x.k2 < y.k2;
          if (x.k1 != y.k1) return x.k1 < y.k1;
          if (x.k0 != y.k0) return x.k0 < y.k0;
          return x.pos < y.pos;                      // equal keys are split by position (see keypos_lt)
        });
        if (me > 0) klo = hs[(size_t)((u64)me * ns / P)];
        if (me + 1 < P) khi = hs[(size_t)((u64)(me + 1) * ns / P)];
      }
      SelTripleKey<Sym> sel; sel.S = S; sel.B = b; sel.klo = klo; sel.khi = khi; sel.has_lo = me > 0 ? 1u : 0u; sel.last = (me + 1 == P) ? 1u : 0u;
      sel.W = W; sel.sb = wsb;
      Rec16 *recA = nullptr, *recB = nullptr, *sorted = nullptr;
      RC(select_records(c, sel, m02, &recA, &cnt, DC3HIP_PH_PACK));
      RC(arena_alloc(c, (size_t)cnt + 16, &recB));
      sorted = recA;
      if (cnt) {
        bool by_splitters = false;
        RC(ssort<Rec16>(c, recA, recB, cnt, sort_bits, &sorted, &by_splitters));
        if (!by_splitters)
          RC(radix_sort<Rec16>(c, recA, recB, cnt, 0, sort_bits, &sorted, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
      }
      // equal keys may straddle ranks: every rank learns its neighbours' boundary keys (first / last record of each
      // rank's sorted range, one small all-gather) and names continue across the boundary where they are equal
      struct Edge { u32 f[3], l[3], has, pad; } mine, all_e[kMaxRanks];
      memset(&mine, 0, sizeof(mine));
      if (cnt) {
        Rec16 fl[2];
        void *fp = nullptr;
        RC(stage_d2h(c, sorted, sizeof(Rec16), &fp)); fl[0] = *static_cast<const Rec16 *>(fp);
        RC(stage_d2h(c, sorted + (cnt - 1), sizeof(Rec16), &fp)); fl[1] = *static_cast<const Rec16 *>(fp);
        mine.f[0] = fl[0].k0; mine.f[1] = fl[0].k1; mine.f[2] = fl[0].k2;
        mine.l[0] = fl[1].k0; mine.l[1] = fl[1].k1; mine.l[2] = fl[1].k2; mine.has = 1;
      }
      RC(cm->all_gather_host(&mine, all_e, sizeof(Edge)));
      bool first_eq = false, last_eq_next = false;
      if (cnt) {
        for (int h = me - 1; h >= 0; h--) if (all_e[h].has) { first_eq = memcmp(all_e[h].l, mine.f, 12) == 0; break; }
        for (int h = me + 1; h < P; h++) if (all_e[h].has) { last_eq_next = memcmp(all_e[h].f, mine.l, 12) == 0; break; }
      }
      AccRec<Rec16> acc; acc.s = sorted;
      RC(gname_pairs<AccRec<Rec16>>(G, acc, cnt, m0, m02, pa, sslot, &names_total, &uniq_total, &cnt_pre, &discard, first_eq,
                                    last_eq_next));
    }
    arena_release(c, mk1);
    RC(rank_exchange(G, pa, cnt, m02, R, DC3HIP_PH_NAMING));      // R[slot] = name (| unique << 31), everywhere
    hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, R, m02, 8u);
    KCHECK();
    if (names_total == m02) {
      rank12 = R;                                                 // all names distinct: the names are the ranks (lib.rs:109-113)
    } else if (discard) {
      c->stats.level_sorted[depth] += 2;
      RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
      RC(gdiscard(G, R, sslot, cnt, cnt_pre, m02, names_total, rank12, depth));
    } else {
      RC(arena_alloc(c, (size_t)m02 + 16, &rank12));
      SymU32 RS; RS.s = R; RS.m = m02;
      RC(glevel<SymU32>(G, RS, m02, names_total, depth + 1, rank12, G_RANK));   // lib.rs:104
    }
  }
  hipLaunchKernelGGL(k_zero_tail, dim3(1), dim3(64), 0, c->stream, rank12, m02, 8u);
  KCHECK();

  // ---- Step 2 + 3, split by rank range (lib.rs:118-192) -------------------------------------------------------
  // my slice of the level's suffix array comes first on the stack (worst case: everything), so that the tuples can be
  // released before it is delivered
  u32 *slice = (mode == G_TOP) ? c->d_sa : nullptr;
  if (!slice) RC(arena_alloc(c, (size_t)m + 16, &slice));
  const ArenaMark mk_merge = arena_mark(c);
  // rank g owns the output between splitter samples g and g+1; the splitters are the samples of rank bound[g]
  const u32 dskip = m0 - m1;                      // lib.rs:133: the dummy has sample rank 1 and is not a suffix
  const u32 first_rank = 1 + dskip, nAtot = m02 - dskip;
  u32 bound[kMaxRanks + 1];
  for (int h = 0; h <= P; h++) bound[h] = first_rank + (u32)((u64)nAtot * h / P);
  const u32 nsp = (u32)(P - 1);
  if (P > 1) {
    RankTargets t; memset(&t, 0, sizeof(t)); t.n = nsp;
    for (int h = 1; h < P; h++) t.r[h - 1] = bound[h];
    HIPC(hipMemsetAsync(c->d_words + 40, 0xff, kMaxRanks * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_find_ranks, dim3(grid_for(c, m02)), dim3(kBlock), 0, c->stream, rank12, m02, t, c->d_words + 40);
    KCHECK();
    HIPC(hipMemcpyAsync(c->h_words + 40, c->d_words + 40, kMaxRanks * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int h = 1; h < P; h++)
      if (c->h_words[40 + h - 1] >= m02) { set_err("global merge: sample rank %u not found (rank12 is not a bijection)", bound[h]); return E_HIP; }
  }
  const u32 lo = bound[me], hi = bound[me + 1], nA = hi - lo;
  // A: my samples in rank order = windowed inversion of (rank - lo, slot), then the tuple gather (slot table built by
  // every rank, streaming).  The P-1 splitter samples ride along behind my nA slots; their tuples come back to the host.
  Tup12 *A = nullptr;
  RC(arena_alloc(c, (size_t)nA + nsp + 16, &A));
  Splitters sp; memset(&sp, 0, sizeof(sp)); sp.n = nsp;
  {
    const ArenaMark mkA = arena_mark(c);
    SelRankRange sr; sr.rank12 = rank12; sr.lo = lo; sr.hi = hi;
    Rec8 *pr = nullptr, *pt = nullptr; u32 got = 0;
    RC(select_records(c, sr, m02, &pr, &got, DC3HIP_PH_RANKS));
    if (got != nA) { set_err("global merge: %u samples in rank range [%u,%u), expected %u", got, lo, hi, nA); return E_HIP; }
    u32 *sa12l = nullptr;
    RC(arena_alloc(c, (size_t)nA + 16, &pt));
    RC(arena_alloc(c, (size_t)nA + nsp + 16, &sa12l));
    if (nA) RC(inverse_permute(c, pr, pt, nA, sa12l, DC3HIP_PH_RANKS));
    if (nsp) HIPC(hipMemcpyAsync(sa12l + nA, c->d_words + 40, nsp * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    const u32 cnt = nA + nsp;
    if (cnt) {
      constexpr u32 kTup0Tile = SortCfg<Tup0, 256>::NW * 64 * SortCfg<Tup0, 256>::IPT;
      const Chunking ckc = make_chunks(c, cnt, kTup0Tile);
      u32 *table0 = nullptr;
      RC(arena_alloc(c, (size_t)256 * ckc.nchunks, &table0));
      RC((build_gather_tuples<Sym>(c, S, m, m0, m02, K, rank12, sa12l, cnt, ckc, A, table0)));
    }
    if (nsp) {
      void *spp = nullptr;
      RC(stage_d2h(c, A + nA, nsp * sizeof(Tup12), &spp));
      memcpy(sp.a, spp, nsp * sizeof(Tup12));
    }
    arena_release(c, mkA);
  }
  // B: my mod-0 tuples, sorted by (first symbol, rank of the suffix behind it)
  SelMod0<Sym> sm; sm.S = S; sm.rank12 = rank12; sm.m = m; sm.m0 = m0; sm.me = (u32)me; sm.sp = sp;
  Tup0 *z0 = nullptr; u32 nB = 0;
  RC(select_records(c, sm, m0, &z0, &nB, DC3HIP_PH_COMPACT));
  Tup0G *zs = reinterpret_cast<Tup0G *>(z0);
  if (nB) {
    Tup0G *z1 = nullptr;
    RC(arena_alloc(c, (size_t)nB + 16, &z1));
    Tup0G *t1 = nullptr;
    RC(radix_sort<Tup0G>(c, reinterpret_cast<Tup0G *>(z0), z1, nB, 0, bits_of((u64)m02), &t1, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
    Tup0G *other = (t1 == z1) ? reinterpret_cast<Tup0G *>(z0) : z1;
    RC(radix_sort<Tup0G>(c, t1, other, nB, 32, 32 + bits_of(K - 1), &zs, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0, DC3HIP_PH_SORT0));
  }
  // my slice of the level's suffix array
  const u32 total = nA + nB;
  uint64_t pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, total, &pre, &tot, all));
  if (tot != m) { set_err("global merge: slices hold %llu of %u suffixes", (unsigned long long)tot, m); return E_HIP; }
  RC(merge_lists(c, A, nA, reinterpret_cast<const Tup0 *>(zs), nB, slice, nullptr, 0u));
  arena_release(c, mk_merge);
  RC(deliver(G, slice, total, pre, all, m, out, mode));
  arena_release(c, mk0);
  return E_OK;
}

// level 0 shortcut: the whole-text order by 9-symbol (Key9) or, on small alphabets, 3L-symbol windows (KeyT), split by
// key range (conditions as in build_core; no reuse of the order when windows repeat: the recursion decides then)
static int gorder_text_msd(dc3hip_gctx *G, u32 sigma, bool *done, bool *tried, bool have_select);     // (defined behind the wide mode's pieces)
template <class KM>
static int gtext_order_with(dc3hip_gctx *G, KM km, u64 BL, const HiMap &hm, u32 sigma, bool wide, bool *done) {
  dc3hip_ctx *c = G->c;
  const u32 n = (u32)G->total_n;
  u32 kbits = 0;
  { unsigned __int128 mx = (unsigned __int128)BL * BL * BL - 1; while (mx) { kbits++; mx >>= 1; } }
  double pred = 1.0;
  if (wide) {                               // hm: image of hm.nbits bits for 12-byte records (see try_text_order12)
    const ArenaMark mk = arena_mark(c);
    const u32 stride = std::max<u32>(1, n >> 20);
    const u32 ns = (n - 1) / stride + 1;
    Rec8 *a = nullptr;
    RC(arena_alloc(c, (size_t)ns, &a));
    hipLaunchKernelGGL((k_pack_image12_pos<KM>), dim3(grid_for(c, ns)), dim3(kBlock), 0, c->stream, km, ns, stride, hm, a);
    KCHECK();
    u32 ts = 0;
    RC(sample_ties(c, a, ns, 1u, &ts));
    const double fs = (double)ts / (double)ns;
    pred = fs >= 1.0 ? 1.0 : 1.0 - pow(1.0 - fs, (double)(n - 1) / (double)(ns > 1 ? ns - 1 : 1));
    arena_release(c, mk);
    c->stats.level_tie_pred[0] = pred;
    if (!(pred < kTextSortMaxPredicted)) return E_OK;
    bool tried = false;
    RC(gorder_text_msd(G, sigma, done, &tried, false));
    if (!tried) RC((gorder_positions12<KM>(G, km, n, kbits, hm, done)));
  } else {
    RC(predict_tie_fraction_pos<KM>(c, km, n, hm, &pred));
    c->stats.level_tie_pred[0] = pred;
    if (!text_order_worth_trying(pred, (u64)n, hm.nbits)) return E_OK;
    bool tried = false;
    // (byte windows: gorder_positions has the selecting pass 1 of the single device's own kernels, cheaper still)
    const bool have_select = std::is_same<KM, Key9>::value && !G->no_select && gselect_pays(G, G->comm->nranks) &&
                             msd_geometry(c, n, hm).on;
    RC(gorder_text_msd(G, sigma, done, &tried, have_select));
    if (!tried) RC((gorder_positions<KM>(G, km, n, kbits, hm, 0, nullptr, G_TOP, done)));
  }
  if (*done) {
    c->stats.text_sort_state = 1;
    c->stats.level_n[0] = n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = 5;
  } else {
    c->stats.text_sort_state = 3;       // some window repeats somewhere: the recursion decides
  }
  return E_OK;
}
static int gtext_order(dc3hip_gctx *G, SymU8 S, u32 sigma, bool *done) {
  dc3hip_ctx *c = G->c;
  const u32 n = (u32)G->total_n;
  *done = false;
  const u64 Bq = (u64)sigma + 1, B3 = Bq * Bq * Bq;
  const double need_bits = 2.0 * log2((double)n) + 2.0, sym_bits = log2((double)sigma);
  if (!(n >= kHybridMinSamples / 4 && !c->no_hybrid && !c->no_fullsort && !c->no_text_shortcut && !G->no_text_order)) return E_OK;
  const bool wide = c->text_order12 >= 0 ? c->text_order12 == 1 : bits_of((u64)n - 1) >= 32;
  const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2((double)n) + 4.2) / 9.0));
  if (9.0 * sym_bits >= need_bits && B3 * B3 * B3 > 0x7fffffffull) {
    u32 kbits = 0;
    { unsigned __int128 mx = (unsigned __int128)B3 * B3 * B3 - 1; while (mx) { kbits++; mx >>= 1; } }
    Key9 km; km.S = S; km.B = (u32)Bq; km.B3 = (u32)B3;
    HiMap hm = make_himap(B3, kbits, n, wide ? 64 - ibits : bits_of((u64)n - 1));
    hm.raw = sigma > 128 && !hm.exact ? 1u : 0u;       // (as build_core: byte alphabets)
    return gtext_order_with<Key9>(G, km, B3, hm, sigma, wide, done);
  }
  if (!c->no_long_keys) {
    u32 L = 1; u64 BL = Bq;
    while (L < 20 && BL * Bq <= 0xffffffffull) { BL *= Bq; L++; }
    KeyT km; HiMap hm;
    if (L > 3 && 3.0 * L * sym_bits >= need_bits && make_keyt(S, sigma, L, BL, n, &km, &hm, wide ? ibits : 0u))
      return gtext_order_with<KeyT>(G, km, BL, hm, sigma, wide, done);
  }
  return E_OK;
}

