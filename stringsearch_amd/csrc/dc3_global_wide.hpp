// dc3_global_wide.hpp — the global mode beyond 2^32 bytes (64-bit positions): the distributed whole-text
// order on 8-byte words, the tie rounds, deepening by rank look-ups, groups of any size, and gbuild_wide.  Kernels:
// dc3_wide.hip.hpp, dc3_wide_msd.hip.hpp.  Included by dc3_global_host.hpp.
#pragma once

// ---------------------------------------------------------------------------------------------
// wide mode: texts of 2^32 bytes and more (kernels and the scope in dc3_wide.hip.hpp)
// ---------------------------------------------------------------------------------------------
static uint8_t *gtext(dc3hip_gctx *G) { return G->wide ? G->w_text : G->c->d_text; }

static int wide_key(dc3hip_gctx *G, u32 sigma, WideKey *k, u32 *ibits_out) {
  const double n = (double)G->total_n;
  // (a text over one symbol: every image is 0 and every window repeats — one group that the deepening orders; the image
  //  arithmetic runs as for two symbols)
  if (sigma < 2) { if (G->no_wide_deepen) { set_err("wide global mode: a text over one symbol has no distinct windows"); return E_TOOBIG; } sigma = 2; }
  const u32 ibits = std::min<u32>(63, 9 * (u32)ceil((log2(n) + 4.2) / 9.0));
  u32 J = 1; u64 SJ = sigma;
  while (J < kWideMaxImageSyms && (SJ >> std::min<u32>(ibits + 2, 62)) == 0 && SJ * sigma < (1ull << 63)) { SJ *= sigma; J++; }
  if ((SJ >> ibits) == 0) { set_err("wide global mode: alphabet of %u symbols cannot fill a %u-bit image", sigma, ibits); return E_TOOBIG; }
  k->t = gtext(G); k->code = G->c->d_code; k->n = (u64)G->total_n; k->sigma = sigma; k->J = J; k->W = kWideWindow;
  k->mfix = (u64)(((((unsigned __int128)1) << (64 + ibits)) - 1) / SJ);
  k->P1 = SJ / sigma;
  k->lg = 0; k->sh = 0;
  if ((sigma & (sigma - 1)) == 0) {          // power of two: sigma^J = 2^(lg J), image = v >> (lg J - ibits) exactly
    const u32 lg = bits_of((u64)sigma) - 1, sh = lg * J - ibits;
    if (lg >= 1 && lg * J > ibits && sh < 64) { k->lg = lg; k->sh = sh; k->mfix = 1ull << (64 - sh); }
  }
  *ibits_out = ibits;
  return E_OK;
}

// grow one of the wide mode's arrays to at least `need` elements (never while it holds live data)
template <class T>
static int wide_ensure(dc3hip_ctx *c, T **p, size_t *cap, size_t need) {
  if (need <= *cap) return E_OK;
  HIPC(hipStreamSynchronize(c->stream));
  if (*p) (void)hipFree(*p);
  *p = nullptr; *cap = 0;
  const size_t want = need + need / 16 + 1024;
  if (hipMalloc(p, want * sizeof(T)) != hipSuccess) {
    (void)hipGetLastError();
    set_err("wide global mode: no device memory for %zu elements of %zu bytes", want, sizeof(T));
    return E_ALLOC;
  }
  *cap = want;
  return E_OK;
}
// ---- groups of any size (kernels: dc3_wide.hip.hpp, "Groups of any size") -------------------------------------------
// f(i, start of i's group) for the n entries whose run structure `same` describes (same[0] = 0)
template <class F>
static int wide_seg_apply(dc3hip_ctx *c, const uint8_t *same, u32 n, F f) {
  if (n == 0) return E_OK;
  const u32 ntiles = (n + kSegTile - 1) / kSegTile;
  const ArenaMark mk = arena_mark(c);
  u32 *tiles = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &tiles));
  hipLaunchKernelGGL(k_seg_last, dim3(ntiles), dim3(kBlock), 0, c->stream, same, n, tiles);
  KCHECK();
  hipLaunchKernelGGL(k_seg_carry, dim3(1), dim3(1024), 0, c->stream, tiles, ntiles);
  KCHECK();
  hipLaunchKernelGGL((k_seg_apply<F>), dim3(ntiles), dim3(kBlock), 0, c->stream, same, n, (const u32 *)tiles, f);
  KCHECK();
  arena_release(c, mk);      // (the stream orders the launches before whatever reuses the table)
  return E_OK;
}
// the members of this rank's groups of more than kWideTieBig entries, compacted in index order
struct WideBig { u32 nb = 0; u32 *cslot = nullptr, *gid = nullptr; u64 *cpos = nullptr; Rec16 *ra = nullptr, *rb = nullptr; };
template <class Pos>
static int wide_big_collect(dc3hip_gctx *G, const uint8_t *same, u32 nrec, Pos pos, WideBig *bg) {
  dc3hip_ctx *c = G->c;
  bg->nb = 0;
  if (nrec == 0) return E_OK;
  RC(wide_ensure(c, &G->w_aux, &G->w_cap_aux, ((size_t)nrec + 16) * 4));
  u32 *gstart = reinterpret_cast<u32 *>(G->w_aux);
  SegStore st; st.gstart = gstart;
  RC(wide_seg_apply(c, same, nrec, st));
  const u32 ntiles = (nrec + kSegTile - 1) / kSegTile;
  const ArenaMark mk = arena_mark(c);
  u32 *counts = nullptr;
  RC(arena_alloc(c, (size_t)ntiles + 16, &counts));
  hipLaunchKernelGGL(k_big_count, dim3(ntiles), dim3(kBlock), 0, c->stream, (const u32 *)gstart, nrec, kWideTieBig, counts);
  KCHECK();
  hipLaunchKernelGGL(k_scan_excl_inplace, dim3(1), dim3(1024), 0, c->stream, counts, ntiles, c->d_words + 34);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 34, c->d_words + 34, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  const u32 nb = c->h_words[34];
  if (nb) {
    const size_t per = 4 + 4 + 8 + 16 + 16;
    const int rc = wide_ensure(c, &G->w_aux2, &G->w_cap_aux2, ((size_t)nb + 16) * per);
    if (rc != E_OK) { arena_release(c, mk); return rc; }
    unsigned char *b = G->w_aux2;
    bg->ra = reinterpret_cast<Rec16 *>(b); b += ((size_t)nb + 16) * 16;
    bg->rb = reinterpret_cast<Rec16 *>(b); b += ((size_t)nb + 16) * 16;
    bg->cpos = reinterpret_cast<u64 *>(b); b += ((size_t)nb + 16) * 8;
    bg->cslot = reinterpret_cast<u32 *>(b); b += ((size_t)nb + 16) * 4;
    bg->gid = reinterpret_cast<u32 *>(b);
    hipLaunchKernelGGL((k_big_write<Pos>), dim3(ntiles), dim3(kBlock), 0, c->stream, (const u32 *)gstart, nrec, kWideTieBig, (const u32 *)counts, pos,
                       bg->cslot, bg->gid, bg->cpos);
    KCHECK();
  }
  bg->nb = nb;
  arena_release(c, mk);
  return E_OK;
}
// Segmented sort of the collected members: `ncomp` key components of `bits` bits, most significant first, made by
// make(component, order so far, records out); the stable LSD passes order them last component first, the group's start
// index last.  *sorted = the records in final order (pos = index of the member in the compacted list).
template <class Make>
static int wide_big_sort(dc3hip_gctx *G, const WideBig &bg, u32 nrec, int ncomp, u32 bits, Make make, const Rec16 **sorted) {
  dc3hip_ctx *c = G->c;
  const Rec16 *prev = nullptr;
  for (int comp = ncomp - 1; comp >= -1; comp--) {
    // (the records of a component are made IN PLACE over the order so far: place j reads and writes element j only)
    Rec16 *in = prev ? const_cast<Rec16 *>(prev) : bg.ra, *other = (in == bg.ra) ? bg.rb : bg.ra, *res = nullptr;
    if (comp >= 0) RC(make(comp, prev, in));
    else {
      hipLaunchKernelGGL(k_seg_recs_gid, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u32 *)bg.gid, in);
      KCHECK();
    }
    RC(radix_sort<Rec16>(c, in, other, bg.nb, 0, comp >= 0 ? bits : bits_of((u64)nrec), &res, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    prev = res;
  }
  *sorted = prev;
  return E_OK;
}
// Big groups of the symbol tie pass (entries with one sort image): ordered by their first kWideBigSyms symbols, exactly.
// pos: positions of the sorted records; same: same-image flags.  The shard then is in order kWideBigSyms symbols deep
// wherever such a group stood (and deeper elsewhere): the caller lowers its depth to that.
constexpr u32 kWideBigSyms = 56;           // 8 components of 7 symbols (a multiple of 4: wide_cmp's depth)
template <class Pos>
static int wide_big_syms(dc3hip_gctx *G, const uint8_t *same, u32 nrec, Pos pos, const WideKey &k, u32 *nb_out) {
  dc3hip_ctx *c = G->c;
  WideBig bg;
  RC(wide_big_collect(G, same, nrec, pos, &bg));
  *nb_out = bg.nb;
  if (!bg.nb) return E_OK;
  PhaseScope ps(c, DC3HIP_PH_TIES, bg.nb);
  const Rec16 *sorted = nullptr;
  RC(wide_big_sort(G, bg, nrec, (int)(kWideBigSyms / 7), 63u, [&](int comp, const Rec16 *prev, Rec16 *out) -> int {
    SegKeySyms key; key.k = k; key.off = (u32)comp * 7u;
    hipLaunchKernelGGL((k_seg_recs<SegKeySyms>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u64 *)bg.cpos, key, k.code, out);
    KCHECK();
    return E_OK;
  }, &sorted));
  hipLaunchKernelGGL(k_seg_writeback, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u64 *)bg.cpos, G->w_shard);
  KCHECK();
  return E_OK;
}
// Big groups of a deepening round (entries that agree on D symbols): ordered by the W rank look-ups isa[p + j D], j = 1..W,
// the new flags written for all their members (words[2] of c->d_words + 10 counts those that still agree).
static int wide_big_isa(dc3hip_gctx *G, const uint8_t *eq, u32 nrec, u64 n, u64 D, u32 W, uint8_t *neweq) {
  dc3hip_ctx *c = G->c;
  WideBig bg;
  PosShard ps; ps.s = G->w_shard;
  RC(wide_big_collect(G, eq, nrec, ps, &bg));
  if (!bg.nb) return E_OK;
  PhaseScope pss(c, DC3HIP_PH_TIES, bg.nb);
  const Rec16 *sorted = nullptr;
  RC(wide_big_sort(G, bg, nrec, (int)W, bits_of(n), [&](int comp, const Rec16 *prev, Rec16 *out) -> int {
    SegKeyIsa key; key.isa = G->w_isa; key.n = n; key.add = (u64)(comp + 1) * D;
    hipLaunchKernelGGL((k_seg_recs<SegKeyIsa>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, prev, bg.nb, (const u64 *)bg.cpos, key, (const uint16_t *)nullptr, out);
    KCHECK();
    return E_OK;
  }, &sorted));
  hipLaunchKernelGGL(k_seg_writeback, dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u64 *)bg.cpos, G->w_shard);
  KCHECK();
  SegCmpIsa cmp; cmp.isa = G->w_isa; cmp.n = n; cmp.D = D; cmp.W = W;
  hipLaunchKernelGGL((k_seg_neweq<SegCmpIsa>), dim3(grid_for(c, bg.nb)), dim3(kBlock), 0, c->stream, sorted, bg.nb, (const u32 *)bg.cslot, (const u32 *)bg.gid,
                     (const u64 *)bg.cpos, cmp, neweq, c->d_words + 10);
  KCHECK();
  return E_OK;
}

// The tie rounds of a wide build over the sorted records h[0..nrec): positions to G->w_shard, statistics in
// c->h_words[10..12] (oversized group, tied records, windows that still agree after the last round).
template <class Launch>
static int wide_tie_rounds_with(dc3hip_gctx *G, u32 nrec, WideKey k, Launch launch) {
  dc3hip_ctx *c = G->c;
    // tie pass; while a few windows still agree completely it is repeated with a deeper compare: kWideWindow symbols, then
    // kWideWindowDeep, then 16 times deeper per round for as long as (windows that still agree) x (next depth) stays inside
    // a work budget — the compare is lazy, so the depth only costs where windows really agree that far.  This settles
    // repeats of any length a few of which exist (two copies of a 100 kB block: 10^5 tied pairs x 10^5 symbols); what
    // the budget does not cover is refused (there is no recursion with 64-bit positions).
    u32 depth = kWideWindow;
    for (int round = 0;; round++) {
      k.W = depth;
      G->w_depth = depth;
      HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
      if (nrec) {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        launch(k);
        KCHECK();
      }
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      if (c->h_words[10] != 0 || c->h_words[12] == 0) break;
      if (round == 0) { if (c->h_words[12] > (1u << 20)) break; depth = kWideWindowDeep; continue; }
      const u64 next = (u64)depth * 16;
      // (with the deepening by rank look-ups behind it, a symbol round is only worth its reads while they stay below what
      //  one exchange of the shards moves)
      const u64 budget = G->no_wide_deepen ? kWideTieBudget : std::max<u64>(1ull << 28, 4 * (u64)G->total_n);
      if (next > kWideMaxDepth || (u64)c->h_words[12] * next > budget) break;
      depth = (u32)next;
    }
    return E_OK;
}

static int wide_tie_rounds(dc3hip_gctx *G, const Rec16 *h, u32 nrec, WideKey k) {
  dc3hip_ctx *c = G->c;
  return wide_tie_rounds_with(G, nrec, k, [&](const WideKey &kk) {
    hipLaunchKernelGGL(k_wide_ties, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, h, nrec, kk, G->w_shard, c->d_words + 10);
  });
}

// Pass 1 of the bucket ordering of a wide rank: selection + x' + partition by its top d1 bits, straight from the text
struct WidePass1 : MsdPass1 {
  WideKey k; WideRange rg; u64 chunk = 0; u32 nchunks = 0, cpg = 0;
  int launch(dc3hip_ctx *c, u64 *out, u32, u64, u32, const MsdGeom &, u32, const u32 *, u32 *cur1) override {
    static std::atomic<bool> attr_set[16];
    if (!attr_set[c->device & 15]) {
#define DC3_WIDE_ATTR(JM, PW) HIPC(dc3_func_set_attribute(reinterpret_cast<const void *>(k_wide_part1<JM, PW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWidePartSmem))
      DC3_WIDE_ATTR(8, false); DC3_WIDE_ATTR(16, false); DC3_WIDE_ATTR(24, false); DC3_WIDE_ATTR(kWideMaxImageSyms, false);
      DC3_WIDE_ATTR(8, true); DC3_WIDE_ATTR(16, true); DC3_WIDE_ATTR(24, true); DC3_WIDE_ATTR(kWideMaxImageSyms, true);
#undef DC3_WIDE_ATTR
      attr_set[c->device & 15] = true;
    }
#define DC3_WIDE_P1(JM, PW) hipLaunchKernelGGL((k_wide_part1<JM, PW>), dim3(kMsdGroups * cpg), dim3(kWideNT), kWidePartSmem, c->stream, k, rg, chunk, nchunks, cpg, cur1, out, c->d_xcdmon)
    if (k.lg) switch (wide_jmax(k.J)) { case 8: DC3_WIDE_P1(8, true); break; case 16: DC3_WIDE_P1(16, true); break; case 24: DC3_WIDE_P1(24, true); break; default: DC3_WIDE_P1(kWideMaxImageSyms, true); }
    else switch (wide_jmax(k.J)) { case 8: DC3_WIDE_P1(8, false); break; case 16: DC3_WIDE_P1(16, false); break; case 24: DC3_WIDE_P1(24, false); break; default: DC3_WIDE_P1(kWideMaxImageSyms, false); }
#undef DC3_WIDE_P1
    KCHECK();
    return E_OK;
  }
};

// this rank's image range [lo, hi) from a strided sample of the replicated text (every rank computes the same sorted
// sample `img`; rank r takes the r-th P-quantile as its lower bound)
static int wide_splitters(dc3hip_gctx *G, const WideKey &k, std::vector<u64> *img, u64 *lo, u64 *hi) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = k.n;
  const ArenaMark mk = arena_mark(c);
  const u32 ns = (u32)std::min<u64>(n, (u64)4096 * P);
  const u64 stride = std::max<u64>(1, n / ns);
  const u32 cnt = (u32)((n - 1) / stride + 1);
  u64 *d_img = nullptr;
  RC(arena_alloc(c, (size_t)cnt, &d_img));
  hipLaunchKernelGGL(k_wide_sample, dim3(grid_for(c, cnt)), dim3(kBlock), 0, c->stream, k, stride, cnt, d_img);
  KCHECK();
  void *ip = nullptr;
  RC(stage_d2h(c, d_img, (size_t)cnt * 8, &ip));
  img->assign(static_cast<const u64 *>(ip), static_cast<const u64 *>(ip) + cnt);
  arena_release(c, mk);
  std::sort(img->begin(), img->end());
  *lo = 0; *hi = ~0ull;
  if (me > 0) *lo = (*img)[(size_t)((u64)me * cnt / P)];
  if (me + 1 < P) *hi = (*img)[(size_t)((u64)(me + 1) * cnt / P)];
  return E_OK;
}

// whether a wide build of n bytes over P ranks uses the bucket ordering on 8-byte words: the same answer on every rank
static bool wide_msd_applies(const dc3hip_gctx *G, u64 n, int P, u32 ibits) {
  const u64 est = n / (u64)P;
  if (G->c->no_msd || G->no_wide_msd || est < G->wide_msd_min || est < 8192) return false;
  const u32 pb = bits_of(n - 1), lg = bits_of(est - 1);
  const u32 tb = std::min<u32>(20, lg > 10 ? lg - 10 : 1);
  // a rank's span is about 2^ibits / P: x' keeps min(bits of the span, 64 - pb + d1) bits and needs tb + 4 of them
  const u32 eb_typ = ibits > bits_of((u64)P) ? ibits - bits_of((u64)P) : 0;
  return pb < 54 && std::min<u32>(eb_typ, 64 - pb + (tb <= 10 ? tb : (tb + 1) / 2)) >= tb + 6;
}

// The order of this rank's image range [lo, hi) by the bucket ordering on 8-byte words (dc3_wide_msd.hip.hpp): counting
// pass over the text, partition pass 1 with selection, the 8-byte passes 2 and 3 of dc3_msd.hip.hpp, tie rounds.
// *done = false: does not apply (switched off, too few positions, too few image bits) or a sub-bucket outgrew the local
// sort — nothing was delivered and the caller runs the 16-byte LSD form.  *nrec_out = the rank's record count.
// OutT / bufs: where the words and the positions live — bufs(nrec, &wa, &wb, &out) hands out two arrays of nrec 8-byte words
// and the array of nrec positions (wide contexts: their own device buffers, 64-bit positions; texts below 2^32: the arena
// and the suffix-array buffer, 32-bit positions).
template <class OutT, class Bufs>
static int wide_msd_order(dc3hip_gctx *G, const WideKey &k, u32 ibits, u64 lo, u64 hi, bool last, u32 *nrec_out, bool *done, Bufs bufs) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks;
  const u64 n = (u64)G->total_n;
  *done = false;
  const u64 est = n / (u64)P;
  const u64 top = ibits >= 64 ? ~0ull : (1ull << ibits);
  const u64 span = (last ? top : hi) - lo;
  if (span < 2) return E_OK;
  const u32 eb = bits_of(span - 1), pb = bits_of(n - 1);
  const u32 lg = bits_of(est - 1);
  u32 tb = lg > 10 ? lg - 10 : 1;
  if (tb > 20) tb = 20;
  u32 d1, d2;
  if (tb <= 10) { d1 = tb; d2 = 0; } else { d1 = (tb + 1) / 2; d2 = tb - d1; }
  const u32 E = std::min<u32>(std::min<u32>(eb, 63u), 64u - pb + d1);
  if (pb >= 64 || E < tb + 4) return E_OK;
  WidePass1 p1;
  p1.k = k;
  p1.rg.lo = lo; p1.rg.hi = hi; p1.rg.last = last ? 1u : 0u; p1.rg.eb = eb; p1.rg.E = E; p1.rg.d1 = d1; p1.rg.pb = pb;
  p1.rg.M = (u64)((((unsigned __int128)1) << (63 + eb)) / span);
  p1.chunk = ((n + 2047) / 2048 + kWideRound - 1) / kWideRound * kWideRound;
  p1.nchunks = (u32)((n + p1.chunk - 1) / p1.chunk);
  p1.cpg = (p1.nchunks + kMsdGroups - 1) / kMsdGroups;
  const u32 nb1 = 1u << d1;
  const ArenaMark mk = arena_mark(c);
  u32 *table = nullptr, *cntg = nullptr;
  RC(arena_alloc(c, (size_t)1024 * p1.nchunks, &table));
  RC(arena_alloc(c, (size_t)nb1 * kMsdGroups + 16, &cntg));
  {
    PhaseScope ps(c, DC3HIP_PH_PACK, (int64_t)n);
#define DC3_WIDE_C1(JM, PW) hipLaunchKernelGGL((k_wide_count1<JM, PW>), dim3(p1.nchunks), dim3(kWideNT), 0, c->stream, k, p1.rg, p1.chunk, p1.nchunks, table)
    if (k.lg) switch (wide_jmax(k.J)) { case 8: DC3_WIDE_C1(8, true); break; case 16: DC3_WIDE_C1(16, true); break; case 24: DC3_WIDE_C1(24, true); break; default: DC3_WIDE_C1(kWideMaxImageSyms, true); }
    else switch (wide_jmax(k.J)) { case 8: DC3_WIDE_C1(8, false); break; case 16: DC3_WIDE_C1(16, false); break; case 24: DC3_WIDE_C1(24, false); break; default: DC3_WIDE_C1(kWideMaxImageSyms, false); }
#undef DC3_WIDE_C1
    KCHECK();
    hipLaunchKernelGGL(k_msd_cnt1, dim3(nb1), dim3(kBlock), 0, c->stream, (const u32 *)table, p1.nchunks, p1.cpg, cntg);
    KCHECK();
  }
  void *hcp = nullptr;
  RC(stage_d2h(c, cntg, (size_t)nb1 * kMsdGroups * 4, &hcp));
  u64 nrec64 = 0;
  for (size_t i = 0; i < (size_t)nb1 * kMsdGroups; i++) nrec64 += static_cast<const u32 *>(hcp)[i];
  if (nrec64 > (u64)DC3HIP_MAX_N) { set_err("wide global mode: rank %d would hold %llu suffixes (more ranks needed)", cm->rank, (unsigned long long)nrec64); return E_TOOBIG; }
  const u32 nrec = (u32)nrec64;
  *nrec_out = nrec;
  if (nrec < 4096) { arena_release(c, mk); return E_OK; }
  u64 *wa = nullptr, *wb = nullptr; OutT *shard = nullptr; uint8_t *same = nullptr;
  RC(bufs(nrec, &wa, &wb, &shard, &same));
  MsdGeom g;
  g.on = true; g.d1 = d1; g.d2 = d2; g.cpg = p1.cpg; g.ck.nchunks = p1.nchunks; g.ck.chunk = 0; g.img_lo = 0; g.ebits = E;
  HiMap hm; hm.mfix = 0; hm.shx = 0; hm.pbits = pb; hm.nbits = E; hm.exact = 0; hm.raw = 0;
  Rec8 *res = nullptr, *where = nullptr; MsdRedo redo; bool ok = false;
  RC(msd_sort(c, reinterpret_cast<Rec8 *>(wa), reinterpret_cast<Rec8 *>(wb), nrec, hm, g, table, nullptr, &res, &redo, &ok, &where, &p1, same));
  if (!ok) { arena_release(c, mk); return E_OK; }
  const u64 *h = reinterpret_cast<const u64 *>(res);
  RC(wide_tie_rounds_with(G, nrec, k, [&](const WideKey &kk) {
    hipLaunchKernelGGL((k_wide_ties8<OutT>), dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, h, (const uint8_t *)same, nrec, pb, kk, shard, c->d_words + 10);
  }));
  if constexpr (sizeof(OutT) == 8) {
    if (c->h_words[10] && G->wide && !G->no_wide_deepen) {      // (groups beyond kWideTieBig records: see gbuild_wide)
      PosWord8 ph; ph.h = h; ph.pmask = (1ull << pb) - 1ull;
      u32 nbig = 0;
      RC(wide_big_syms(G, same, nrec, ph, k, &nbig));
      G->w_depth = std::min<u32>(G->w_depth, kWideBigSyms);
      c->h_words[10] = 0; c->h_words[12] = std::max<u32>(c->h_words[12], 1u);
    }
  }
  arena_release(c, mk);
  G->gs.wide_msd = 1;
  *done = true;
  return E_OK;
}

// The whole-text order of a text below 2^32 bytes in the form the wide contexts use (gbuild_wide / wide_msd_order): every
// rank takes the images of its range straight from its replica of the text — counting pass, partition pass 1 with
// selection, 8-byte passes 2 and 3, tie rounds with lazily compared windows — and nothing but the text blocks has crossed
// the transport.  Replaces the routed order (pack own block, partition by owner, all-to-all of 8-byte records, count the
// top digit again) where the bucket ordering applies: numbers in DESIGN.md §6.  *done = false: some rank's windows repeat (or the ordering does not apply): the caller goes on as before.
static int gorder_text_msd(dc3hip_gctx *G, u32 sigma, bool *done, bool *tried, bool have_select) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = (u64)G->total_n;
  *done = false; *tried = false;
  if (sigma < 2) return E_OK;
  WideKey k; u32 ibits = 0;
  { char keep[sizeof(g_err)]; snprintf(keep, sizeof(keep), "%s", g_err); if (wide_key(G, sigma, &k, &ibits) != E_OK) { set_err("%s", keep); return E_OK; } }
  if (!wide_msd_applies(G, n, P, ibits)) return E_OK;
  if ((double)k.W * log2((double)sigma) < 2.0 * log2((double)n) + 2.0) return E_OK;
  // Where it pays (total work of P loopback ranks on one GPU, 256 MiB random bytes: routed 8.1 / 9.0 ms for P = 2 / 4,
  // unrouted 8.3 / 11.5 — every rank evaluates all n positions twice): from 2^31 positions on, where the routed order
  // would sort 12-byte records with LSD passes, and for two ranks.  DC3HIP_WIDE_MSD_MIN set explicitly (tests) forces it.
  if (!(G->wide_msd_forced || (P <= 2 && !have_select) || bits_of(n - 1) >= 32)) return E_OK;
  *tried = true;
  const ArenaMark mk = arena_mark(c);
  u64 lo = 0, hi = ~0ull;
  std::vector<u64> img;
  RC(wide_splitters(G, k, &img, &lo, &hi));
  u32 nrec = 0; bool ordered = false;
  c->h_words[10] = c->h_words[11] = c->h_words[12] = 0;
  const int rc = wide_msd_order<u32>(G, k, ibits, lo, hi, me + 1 == P, &nrec, &ordered, [&](u32 cnt, u64 **wa, u64 **wb, u32 **out, uint8_t **same) -> int {
    RC(arena_alloc(c, (size_t)cnt + 16, wa));
    RC(arena_alloc(c, (size_t)cnt + 16, wb));
    RC(arena_alloc(c, (size_t)cnt + 16, same));
    *out = c->d_sa;
    return E_OK;
  });
  if (rc != E_OK && rc != E_TOOBIG && rc != E_ALLOC) return rc;
  const bool mine_ok = rc == E_OK && ordered && c->h_words[10] == 0 && c->h_words[12] == 0;
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(gather_counts(cm, mine_ok ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, mine_ok ? nrec : 0, &pre, &tot, all));
  arena_release(c, mk);
  if (ngood == (uint64_t)P) {
    if (tot != n) { set_err("global order: %llu of %llu positions selected", (unsigned long long)tot, (unsigned long long)n); return E_HIP; }
    c->stats.level_tied[0] = c->h_words[11];
    *done = true;
    RC(deliver(G, c->d_sa, nrec, pre, all, (u32)n, nullptr, G_TOP));
  }
  return E_OK;
}

template <class T> static void wide_release(T **p, size_t *cap) { if (*p) (void)hipFree(*p); *p = nullptr; *cap = 0; }
// Deepening by rank look-ups (kernels and the idea: dc3_wide.hip.hpp): collective; entered when some rank's windows still
// agree after the last symbol compare (depth G->w_depth) and no rank met an oversized group.  Every round all ranks
// exchange their shards and equal-window flags (9 bytes per suffix of the text), build the inverse, and order their
// groups by kWideDeepenW + 1 rank look-ups per compare.  *ok = every window of every rank is distinct now; the shards are
// in suffix order and G->w_isa is the exact inverse (kept for the verifier).  *ok = false: no memory, or an oversized group.
constexpr u32 kWideDeepenW = 16;
static int wide_deepen(dc3hip_gctx *G, WideKey k, u32 nrec, u64 pre, const uint64_t *all, bool *ok) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = k.n;
  *ok = false;
  HIPC(hipStreamSynchronize(c->stream));
  wide_release(&G->w_ra, &G->w_cap_a);                     // the sort's buffers are done with: room for the inverse
  wide_release(&G->w_rb, &G->w_cap_b);
  u64 maxshard = 0;
  for (int r = 0; r < P; r++) maxshard = std::max<u64>(maxshard, all[r]);
  // the inverse is built from one rank's shard at a time (w_sa_all = the largest shard), never from the whole order
  int rc_alloc = wide_ensure(c, &G->w_sa_all, &G->w_cap_sa, (size_t)maxshard + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_isa, &G->w_cap_isa, (size_t)n + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_eq_all, &G->w_cap_eq, (size_t)n + 16);
  if (rc_alloc == E_OK) rc_alloc = wide_ensure(c, &G->w_eq2, &G->w_cap_eq2, (size_t)nrec + 16);
  if (rc_alloc != E_OK && rc_alloc != E_ALLOC) return rc_alloc;
  uint64_t badp = 0, nbad = 0;
  RC(gather_counts(cm, rc_alloc != E_OK ? 1u : 0u, &badp, &nbad));
  if (nbad) return E_OK;                                   // (every rank returns here: the caller refuses the text as before)
  size_t roff1[kMaxRanks], rb1[kMaxRanks];
  u64 first[kMaxRanks];
  { u64 acc = 0; for (int r = 0; r < P; r++) { first[r] = acc; roff1[r] = (size_t)acc; rb1[r] = (size_t)all[r]; acc += all[r]; } }
  // the depth the look-ups start from: what EVERY rank's symbol compares reached (a rank stops deepening them by its own
  // count of agreeing windows; its shard is in order at least that deep)
  u64 D = G->w_depth;
  { uint64_t mine = G->w_depth, depths[kMaxRanks]; RC(cm->all_gather_host(&mine, depths, sizeof(uint64_t))); for (int r = 0; r < P; r++) D = std::min<u64>(D, depths[r]); }
  k.W = (u32)D;
  if (nrec) {
    PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
    hipLaunchKernelGGL(k_wide_eq, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, (const u64 *)G->w_shard, nrec, k, G->w_eq_all + pre);
    KCHECK();
  }
  bool final_round = false;
  for (int round = 0; round < 64; round++) {
    RC(cm->all_gather_v(G->w_eq_all + pre, (size_t)nrec, G->w_eq_all, roff1, rb1, c->stream));
    G->gs.exchanges += 1;
    // isa[p] = 1 + index of the first entry of p's group, rank by rank: every rank receives rank r's shard (an all-gather
    // in which only r contributes) and scatters the group starts of that range; a rank's range begins with a new group
    HIPC(hipMemsetAsync(G->w_isa + n, 0, 8, c->stream));
    for (int r = 0; r < P; r++) {
      if (all[r] == 0) continue;
      size_t ro[kMaxRanks], rbz[kMaxRanks];
      for (int q = 0; q < P; q++) { ro[q] = 0; rbz[q] = 0; }
      rbz[r] = (size_t)all[r] * 8;
      RC(cm->all_gather_v(r == me ? (const void *)G->w_shard : (const void *)G->w_sa_all, r == me ? (size_t)nrec * 8 : 0, G->w_sa_all, ro, rbz, c->stream));
      PhaseScope ps(c, DC3HIP_PH_RANKS, (int64_t)all[r]);
      SegIsa f; f.sa = G->w_sa_all; f.isa = G->w_isa; f.base = first[r];
      RC(wide_seg_apply(c, G->w_eq_all + first[r], (u32)all[r], f));
    }
    if (final_round) { *ok = true; break; }
    HIPC(hipMemsetAsync(c->d_words + 10, 0, 3 * sizeof(u32), c->stream));
    int rc_big = E_OK;
    if (nrec) {
      {
        PhaseScope ps(c, DC3HIP_PH_TIES, nrec);
        hipLaunchKernelGGL(k_wide_ties_isa, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, G->w_shard, (const uint8_t *)(G->w_eq_all + pre), nrec,
                           (const u64 *)G->w_isa, n, D, kWideDeepenW, G->w_eq2, c->d_words + 10);
        KCHECK();
      }
      HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
      HIPC(hipStreamSynchronize(c->stream));
      if (c->h_words[10]) {
        // groups beyond kWideTieBig members: a segmented sort by the same look-ups (rank-local; its failure is agreed on below)
        rc_big = wide_big_isa(G, G->w_eq_all + pre, nrec, n, D, kWideDeepenW, G->w_eq2);
        if (rc_big != E_OK && rc_big != E_ALLOC) return rc_big;
      }
      HIPC(hipMemcpyAsync(G->w_eq_all + pre, G->w_eq2, (size_t)nrec, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPC(hipMemcpyAsync(c->h_words + 10, c->d_words + 10, 3 * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->h_words[10] = 0;
    G->gs.wide_deepen_rounds += 1;
    uint64_t p0 = 0, nfail = 0, ntied = 0;
    RC(gather_counts(cm, rc_big != E_OK ? 1u : 0u, &p0, &nfail));
    RC(gather_counts(cm, c->h_words[12] ? 1u : 0u, &p0, &ntied));
    if (nfail) { if (rc_big == E_OK) set_err("wide global mode: another rank has no memory for its groups of tied suffixes"); break; }
    D *= (u64)kWideDeepenW + 1;
    if (!ntied) final_round = true;                        // (one more exchange: the inverse of the finished order)
    else if (D > 2 * n * ((u64)kWideDeepenW + 1)) { set_err("internal: suffixes still tied %llu symbols deep", (unsigned long long)D); return E_HIP; }
  }
  if (*ok) { G->w_isa_valid = true; c->h_words[10] = 0; c->h_words[12] = 0; G->w_depth = (u32)std::min<u64>(D, 1u << 30); }
  return E_OK;
}

// The placement probe of context creation (xcd_rr -> no_msd) is a per-device observation, but no_msd decides which COLLECTIVE
// schedule a rank runs (selected pass 1 without an all-to-all, or the routed form; whether the wide bucket ordering is
// tried): ranks on different devices — or a probe disturbed on one of them — must not disagree.  One host all-gather per
// build: the bucket ordering is used only if every rank may use it.
static int gagree_placement(dc3hip_gctx *G) {
  GComm *cm = G->comm;
  if (cm->nranks == 1) return E_OK;
  uint64_t mine = G->c->no_msd ? 1u : 0u, all[kMaxRanks];
  RC(cm->all_gather_host(&mine, all, sizeof(uint64_t)));
  for (int r = 0; r < cm->nranks; r++) if (all[r]) G->c->no_msd = true;
  return E_OK;
}

static int gbuild_wide(dc3hip_gctx *G) {
  dc3hip_ctx *c = G->c; GComm *cm = G->comm;
  const int P = cm->nranks, me = cm->rank;
  const u64 n = (u64)G->total_n;
  c->n = 0;
  c->arena_off = 0;
  RC(gagree_placement(G));
  G->w_isa_valid = false;
  if (G->w_sa_all || G->w_isa) {             // (a deepened build's whole-order arrays: the sort needs the room again)
    HIPC(hipStreamSynchronize(c->stream));
    wide_release(&G->w_sa_all, &G->w_cap_sa); wide_release(&G->w_isa, &G->w_cap_isa); wide_release(&G->w_eq_all, &G->w_cap_eq); wide_release(&G->w_eq2, &G->w_cap_eq2);
  }
  RC(ensure_arena(c, (size_t)256 << 20));   // the sorts' tables: digit table 8 MB, 2 x 2^20 sub-buckets x 8 groups x 4 bytes, counts
  RC(build_begin(c));
  {
    size_t roff[kMaxRanks], rbytes[kMaxRanks];
    for (int r = 0; r < P; r++) { int64_t o, l; block_of((int64_t)n, P, r, &o, &l); roff[r] = (size_t)o; rbytes[r] = (size_t)l; }
    RC(cm->all_gather_v(G->w_text + roff[me], rbytes[me], G->w_text, roff, rbytes, c->stream));
    HIPC(hipMemsetAsync(G->w_text + n, 0, 64, c->stream));
  }
  // alphabet (the presence kernel counts in 32 bits: pieces of 2^30 bytes)
  HIPC(hipMemsetAsync(c->d_present, 0, 256 * sizeof(u32), c->stream));
  for (u64 off = 0; off < n; off += (u64)1 << 30) {
    const u32 len = (u32)std::min<u64>((u64)1 << 30, n - off);
    hipLaunchKernelGGL(k_byte_presence, dim3(grid_for(c, (u64)len / 16 + 1)), dim3(kBlock), 0, c->stream, G->w_text + off, len, c->d_present);
    KCHECK();
  }
  hipLaunchKernelGGL(k_make_codes, dim3(1), dim3(kBlock), 0, c->stream, c->d_present, c->d_code, c->d_words + 1);
  KCHECK();
  HIPC(hipMemcpyAsync(c->h_words + 1, c->d_words + 1, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
  HIPC(hipStreamSynchronize(c->stream));
  const u32 sigma = c->h_words[1];
  WideKey k; u32 ibits = 0;
  RC(wide_key(G, sigma, &k, &ibits));
  if (G->no_wide_deepen && (double)k.W * log2((double)sigma) < 2.0 * log2((double)n) + 2.0) {
    set_err("wide global mode: %u-symbol windows over %u symbols cannot all be distinct in %llu bytes", k.W, sigma, (unsigned long long)n);
    return E_TOOBIG;
  }
  const ArenaMark mk = arena_mark(c);
  // splitters from a strided sample (every rank computes the same ones from the replicated text)
  u64 lo = 0, hi = ~0ull;
  std::vector<u64> img;
  RC(wide_splitters(G, k, &img, &lo, &hi));
  // The selection, the sort and the tie pass of this rank.  A refusal that depends on the data and on the rank (its share
  // exceeds 2^32 - 2^24 suffixes, no device memory for the records) must not leave the other ranks waiting in the
  // collectives below: the status is agreed on there and every rank returns the same error.
  u32 nrec = 0;
  c->h_words[10] = c->h_words[11] = c->h_words[12] = 0;
  // agreement on a refusal (see above): 0, or the code every rank returns
  auto agree = [&](int rc_local) -> int {
    uint64_t refp = 0, refused = 0;
    char local_err[sizeof(g_err)];
    snprintf(local_err, sizeof(local_err), "%s", g_err);
    RC(gather_counts(cm, rc_local == E_TOOBIG ? 1u : rc_local == E_ALLOC ? (1u << 20) : 0u, &refp, &refused));
    if (!refused) return E_OK;
    const int rc_all = (refused >> 20) ? E_ALLOC : E_TOOBIG;
    if (rc_local != E_OK) set_err("%s", local_err);
    else set_err("wide global mode: another rank refused its share (%s)", rc_all == E_ALLOC ? "no device memory for its records" : "more ranks needed");
    return rc_all;
  };
  // (no record is routed between ranks: every rank selects straight from its replica of the text, so what one rank cannot
  //  order by the bucket ordering it orders by the 16-byte LSD form on its own, and no collective sits in between)
  const bool msd_static = wide_msd_applies(G, n, P, ibits);
  const int local_rc = [&]() -> int {
    // bucket ordering on 8-byte words where it applies (the 16-byte LSD form below otherwise)
    if (msd_static) {
      bool msd_done = false;
      // (two arrays of 8-byte words inside the record buffers, and the shard)
      RC((wide_msd_order<u64>(G, k, ibits, lo, hi, me + 1 == P, &nrec, &msd_done, [&](u32 cnt, u64 **wa, u64 **wb, u64 **out, uint8_t **same) -> int {
        RC(wide_ensure(c, &G->w_ra, &G->w_cap_a, (size_t)cnt / 2 + 16));
        RC(wide_ensure(c, &G->w_rb, &G->w_cap_b, (size_t)cnt / 2 + 16));
        RC(wide_ensure(c, &G->w_shard, &G->w_cap_s, (size_t)cnt + 16));
        RC(wide_ensure(c, &G->w_same, &G->w_cap_same, (size_t)cnt + 16));
        *wa = reinterpret_cast<u64 *>(G->w_ra); *wb = reinterpret_cast<u64 *>(G->w_rb); *out = G->w_shard; *same = G->w_same;
        return E_OK;
      })));
      if (msd_done) return E_OK;
    }
    // count, allocate, write
    const u64 chunk = (u64)1 << 20;
    const u64 nblocks64 = (n + chunk - 1) / chunk;
    if (nblocks64 > 0x7fffffffull) { set_err("wide global mode: text too long"); return E_TOOBIG; }
    const u32 nblocks = (u32)nblocks64;
    u32 *counts = nullptr;
    RC(arena_alloc(c, (size_t)nblocks + 16, &counts));
    const u32 last = (me + 1 == P) ? 1u : 0u;
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, n);
      hipLaunchKernelGGL((k_wide_select<false>), dim3(nblocks), dim3(kBlock), 0, c->stream, k, chunk, lo, hi, last, counts,
                         (const u32 *)nullptr, (Rec16 *)nullptr);
      KCHECK();
    }
    // (the per-block counts are summed in 64 bits on the host: a rank's share must stay below 2^32 - 2^24 records)
    void *hcp = nullptr;
    RC(stage_d2h(c, counts, (size_t)nblocks * 4, &hcp));
    std::vector<u32> hc(static_cast<const u32 *>(hcp), static_cast<const u32 *>(hcp) + nblocks);
    u64 nrec64 = 0;
    for (u32 b = 0; b < nblocks; b++) { const u32 v = hc[b]; hc[b] = (u32)nrec64; nrec64 += v; }
    if (nrec64 > (u64)DC3HIP_MAX_N) { set_err("wide global mode: rank %d would hold %llu suffixes (more ranks needed)", me, (unsigned long long)nrec64); return E_TOOBIG; }
    nrec = (u32)nrec64;
    HIPC(hipMemcpyAsync(counts, hc.data(), (size_t)nblocks * 4, hipMemcpyHostToDevice, c->stream));
    RC(wide_ensure(c, &G->w_ra, &G->w_cap_a, (size_t)nrec + 16));
    RC(wide_ensure(c, &G->w_rb, &G->w_cap_b, (size_t)nrec + 16));
    RC(wide_ensure(c, &G->w_shard, &G->w_cap_s, (size_t)nrec + 16));
    {
      PhaseScope ps(c, DC3HIP_PH_PACK, n);
      hipLaunchKernelGGL((k_wide_select<true>), dim3(nblocks), dim3(kBlock), 0, c->stream, k, chunk, lo, hi, last, (u32 *)nullptr,
                         (const u32 *)counts, G->w_ra);
      KCHECK();
    }
    Rec16 *h = G->w_ra;
    if (nrec) RC(radix_sort<Rec16>(c, G->w_ra, G->w_rb, nrec, 0, ibits, &h, DC3HIP_PH_SORT12_UP, DC3HIP_PH_SORT12_SCAN, DC3HIP_PH_SORT12_DOWN));
    RC(wide_tie_rounds(G, h, nrec, k));
    if (c->h_words[10] && !G->no_wide_deepen) {
      // images shared by more than kWideTieBig records (a run of one symbol, a short period): those groups are ordered by
      // their first kWideBigSyms symbols here and go on through the deepening like every other repeat
      RC(wide_ensure(c, &G->w_same, &G->w_cap_same, (size_t)nrec + 16));
      hipLaunchKernelGGL(k_wide_same16, dim3(grid_for(c, nrec)), dim3(kBlock), 0, c->stream, (const Rec16 *)h, nrec, G->w_same);
      KCHECK();
      PosRec16 ph; ph.h = h;
      u32 nbig = 0;
      RC(wide_big_syms(G, G->w_same, nrec, ph, k, &nbig));
      G->w_depth = std::min<u32>(G->w_depth, kWideBigSyms);
      c->h_words[10] = 0; c->h_words[12] = std::max<u32>(c->h_words[12], 1u);
    }
    return E_OK;
  }();
  if (local_rc != E_OK && local_rc != E_TOOBIG && local_rc != E_ALLOC) return local_rc;     // HIP / transport faults: as before
  arena_release(c, mk);
  c->stats.level_tied[0] = c->h_words[11];
  const bool mine_ok = local_rc == E_OK && c->h_words[10] == 0 && c->h_words[12] == 0;
  uint64_t good = 0, ngood = 0, pre = 0, tot = 0, all[kMaxRanks];
  RC(agree(local_rc));
  RC(gather_counts(cm, mine_ok ? 1 : 0, &good, &ngood));
  RC(gather_counts(cm, nrec, &pre, &tot, all));
  if (tot != n) { set_err("wide global order: %llu of %llu positions selected", (unsigned long long)tot, (unsigned long long)n); return E_HIP; }
  if (ngood != (uint64_t)P && !G->no_wide_deepen) {
    // windows repeat beyond what the symbol compares settle: rank look-ups (wide_deepen)
    bool deep_ok = false;
    RC(wide_deepen(G, k, nrec, pre, all, &deep_ok));
    if (deep_ok) ngood = (uint64_t)P;
  }
  if (ngood != (uint64_t)P) {
    set_err("wide global mode: some %u-symbol window of the text repeats; texts of 2^32 bytes and more are only built when all windows "
            "are distinct (no recursion with 64-bit positions) [rank %d: %u records, %u tied, %u equal windows, oversized group %u]",
            k.W, me, nrec, c->h_words[11], c->h_words[12], c->h_words[10]);
    return E_TOOBIG;
  }
  G->shard_first = (int64_t)pre; G->shard_count = (int64_t)nrec; G->shard_ptr = nullptr;
  long long corrupt = 0;
  if (dbg_num("wide_corrupt", &corrupt)) {
    // test hook for the verifier: 1 = swap two neighbours of the last rank's shard, 2 = put one position out of range
    const char e[2] = {(char)('0' + corrupt), 0};
    if (me == P - 1 && nrec >= 2 && (e[0] == '1' || e[0] == '2')) {
      u64 two[2];
      HIPC(hipMemcpy(two, G->w_shard + nrec / 2, 16, hipMemcpyDeviceToHost));
      if (e[0] == '1') std::swap(two[0], two[1]); else two[0] = n;
      HIPC(hipMemcpy(G->w_shard + nrec / 2, two, 16, hipMemcpyHostToDevice));
    }
  }
  c->stats.text_sort_state = 1;
  c->stats.level_n[0] = (int64_t)n; c->stats.level_K[0] = sigma; c->stats.levels = 1; c->stats.level_sorted[0] = 5;
  G->gs.local_from_level = -1;
  RC(build_end(c));
  return E_OK;
}

