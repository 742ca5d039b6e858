// dc3_aux.hip.hpp — synthetic text generator, GPU sufcheck, checksum, BWT, batched search.
// Part of the gfx950 kernel set of libdc3hip (see dc3_kernels.hip.hpp for the overview); all files share
// namespace dc3 and are included in this order by dc3_kernels.hip.hpp.
#pragma once

namespace dc3 {

// ---------------------------------------------------------------------------------------------
// Synthetic text generator: byte i = byte (i&7) of splitmix64(seed + (i>>3)).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// kind 2: low-entropy text with deep LCPs (position-addressable): 16-byte
// cells "word1 word2.." from a 4096-word vocabulary with a skewed id distribution, newline every
// 80 bytes, and every 64 KiB block starts with probability 1/2 with a 1-8 KiB copy of an earlier span.
__device__ __forceinline__ uint8_t text_base(u64 i, u64 seed) {
  const u64 cell = i >> 4; const u32 off = (u32)(i & 15);
  if (cell % 5 == 4 && off == 15) return '\n';
  const u64 hc = splitmix64(seed + cell * 0x9E3779B97F4A7C15ull);
  const u64 a = hc & 0xFFFF, b = (hc >> 16) & 0xFFFF, c = (hc >> 32) & 0xFFFF;
  const u64 wid = (((a * b) >> 16) * c) >> 20;
  const u64 hw = splitmix64(0x5EEDull ^ (wid << 1));
  const u32 wlen = 2 + (u32)(hw % 11);
  if (off < wlen) return (uint8_t)('a' + ((hw >> (8 + 4 * off)) % 26));
  if (off == wlen) return ' ';
  const u64 wid2 = (((hc >> 48) & 0xFFF) * ((hc >> 40) & 0xFF)) >> 8;
  const u64 hw2 = splitmix64(0x5EEDull ^ (wid2 << 1));
  const u32 wlen2 = 2 + (u32)(hw2 % 11), o2 = off - wlen - 1;
  if (o2 < wlen2) return (uint8_t)('a' + ((hw2 >> (8 + 4 * o2)) % 26));
  return ' ';
}
__device__ __forceinline__ uint8_t text_byte(u64 i, u64 seed) {
  const u64 block = i >> 16, within = i & 0xFFFF;
  if (block > 0) {
    const u64 hb = splitmix64((seed ^ 0xB10Cull) + block * 0xD1B54A32D192ED03ull);
    if (hb & 1) {
      const u64 len = 1024 + ((hb >> 8) % 7169);
      if (within < len) {
        const u64 sb = (hb >> 24) % block, so = (hb >> 44) % (65536 - 8192);
        return text_base(sb * 65536 + so + within, seed);
      }
    }
  }
  return text_base(i, seed);
}
__device__ __forceinline__ uint8_t gen_byte(u64 gi, u64 seed, int kind) {
  if (kind == 2) return text_byte(gi, seed);
  if (kind == 0) return (uint8_t)(splitmix64(seed + (gi >> 3)) >> (8 * (gi & 7)));
  const u32 code = (u32)(splitmix64(seed + (gi >> 5)) >> (2 * (gi & 31))) & 3u;
  return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
}
// t[i] = byte (off+i) of the stream; one thread per 8 output bytes, 8-byte stores
__global__ __launch_bounds__(kBlock) void k_generate(uint8_t *t, u64 n, u64 seed, int kind, u64 off) {
  const u64 nw = (n + 7) / 8;
  for (u64 wi = blockIdx.x * (u64)kBlock + threadIdx.x; wi < nw; wi += (u64)gridDim.x * kBlock) {
    u64 v = 0;
    if (kind == 0 && (off & 7) == 0) {
      v = splitmix64(seed + ((off + wi * 8) >> 3));
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) v |= (u64)gen_byte(off + wi * 8 + j, seed, kind) << (8 * j);
    }
    if (wi * 8 + 8 <= n) *reinterpret_cast<u64 *>(t + wi * 8) = v;
    else for (u64 j = wi * 8; j < n; j++) t[j] = (uint8_t)(v >> (8 * (j & 7)));
  }
}

// ---------------------------------------------------------------------------------------------
// GPU verifier = sufcheck() of crates/cdivsufsort/c-sources/utils.c:160-241 (equivalently sacabase::verify,
// sacabase/src/lib.rs:127-149), its three loops as they are:
//   (1) range: every SA[i] in [0, n)                                              utils.c:179-188   (-2)
//   (2) first characters non-decreasing                                           utils.c:191-201   (-3)
//   (3) the scan of utils.c:213-238: for i = 0, 1, ...: the suffix SA[i] - 1, whose first character is c = T[SA[i] - 1],
//       must stand in the next free slot of c's bucket — C[c]++, buckets from the text's histogram; the slot
//       q = (start of T[n-1]'s bucket) is reserved for the suffix n - 1, which the entry SA[i] = 0 claims        (-4)
// Round 6: (3) is what one STABLE counting pass computes — partition the records (c, SA[i] - 1), i ascending, by c: the
// record that arrives at slot t (the gap at q skipped) must equal SA[t].  So the verifier is one gather kernel (the only
// random access: the two text bytes at SA[i] - 1, SA[i]; it also checks (1), (2), leaves the bytes c in rank order and
// the pass's digit table), the text's byte histogram, and one down-sweep of the product's radix pass whose sink COMPARES
// instead of storing (nothing is written but n bytes of c).  An array passes iff the reference's loops accept it: the
// slots the pass assigns are the reference's C[c]++ as long as no bucket overflows, and bucket sizes that differ from the
// text's histogram (an in-range array that is not a permutation) fail here (-4) as they make the reference run a cursor
// out of its bucket.  Rounds 1-5 built the inverse array by a random scatter and gathered it twice per entry: 98 ms per
// 2^30 entries against 11 ms for the build they verified.
// err receives the largest failing value seen: 3 -> -2, 2 -> -3, 1 -> -4 (as in sufcheck, -2 before -3 before -4).
// ---------------------------------------------------------------------------------------------
constexpr int kCheckIPT = 4;      // ranks per lane and round of k_check_gather (independent gathers in flight)
__global__ __launch_bounds__(kBlock) void k_check_text_hist(const uint8_t *__restrict__ t, u32 n, u32 *__restrict__ hist /*[256], zeroed*/) {
  __shared__ u32 h[kWaves][256];
  for (int j = threadIdx.x; j < kWaves * 256; j += kBlock) (&h[0][0])[j] = 0;
  __syncthreads();
  u32 *myh = h[wave_id()];
  const u32 nw = n / 16;                        // 16-byte pieces (the text buffer is 16-byte aligned or this is a partition: see below)
  const bool aligned = (reinterpret_cast<uintptr_t>(t) & 15) == 0;
  if (aligned) {
    const uint4 *t4 = reinterpret_cast<const uint4 *>(t);
    for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < nw; i += gridDim.x * kBlock) {
      const uint4 v = t4[i];
      const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        atomicAdd(&myh[w[k] & 255u], 1u); atomicAdd(&myh[(w[k] >> 8) & 255u], 1u);
        atomicAdd(&myh[(w[k] >> 16) & 255u], 1u); atomicAdd(&myh[w[k] >> 24], 1u);
      }
    }
    for (u32 i = nw * 16 + blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) atomicAdd(&myh[t[i]], 1u);
  } else {
    for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) atomicAdd(&myh[t[i]], 1u);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 256; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += h[w][j];
    if (sum) atomicAdd(&hist[j], sum);
  }
}
// Block b works chunk b of the ranks (the chunking of the radix pass that follows: table[c * nchunks + b] = its records
// with first byte c).  bw[i] = T[SA[i] - 1] for 0 < SA[i] < n (0 otherwise: such entries are not records).
__global__ __launch_bounds__(kBlock) void k_check_gather(const uint8_t *__restrict__ t, const u32 *__restrict__ sa, u32 n, u32 chunk,
                                                        u32 nchunks, uint8_t *__restrict__ bw, u32 *__restrict__ table, int *err) {
  __shared__ u32 hist[kWaves][256];
  for (int j = threadIdx.x; j < kWaves * 256; j += kBlock) (&hist[0][0])[j] = 0;
  __syncthreads();
  u32 *myh = hist[wave_id()];
  const u32 lane = lane_id();
  const u32 begin = blockIdx.x * chunk, end = min(n, begin + chunk);
  int bad = 0;
  // a wave takes 64 * kCheckIPT consecutive ranks per round: rank (k, lane) = base + 64 k + lane
  for (u32 base = begin + wave_id() * (64u * kCheckIPT); base < end; base += kBlock * kCheckIPT) {
    u32 p[kCheckIPT], cf[kCheckIPT], cp[kCheckIPT], in[kCheckIPT];
#pragma unroll
    for (int k = 0; k < kCheckIPT; k++) {
      const u32 i = base + (u32)k * 64u + lane;
      p[k] = i < end ? sa[i] : 0xffffffffu;
      in[k] = p[k] < n ? 1u : 0u;
    }
#pragma unroll
    for (int k = 0; k < kCheckIPT; k++) {
      // T[p - 1], T[p] in one access where p > 0
      uint16_t v = 0;
      if (in[k]) { if (p[k] > 0) __builtin_memcpy(&v, t + p[k] - 1, 2); else v = (uint16_t)((u32)t[0] << 8); }
      cp[k] = v & 255u; cf[k] = v >> 8;
    }
    // the successor's first byte: the next lane's, or the first lane's of the wave's next 64 ranks (all lanes take part in
    // the shuffles; what they mean is sorted out below)
    u32 cq[kCheckIPT], qin[kCheckIPT];
#pragma unroll
    for (int k = 0; k < kCheckIPT; k++) {
      cq[k] = __shfl_down(cf[k], 1); qin[k] = __shfl_down(in[k], 1);
      if (k + 1 < kCheckIPT) {
        const u32 c0 = __shfl(cf[k + 1 < kCheckIPT ? k + 1 : k], 0), i0 = __shfl(in[k + 1 < kCheckIPT ? k + 1 : k], 0);
        if (lane == 63) { cq[k] = c0; qin[k] = i0; }
      }
    }
#pragma unroll
    for (int k = 0; k < kCheckIPT; k++) {
      const u32 i = base + (u32)k * 64u + lane;
      if (i >= end) continue;
      if (!in[k]) { bad = max(bad, 3); bw[i] = 0; continue; }          // utils.c:179-188
      bw[i] = p[k] > 0 ? (uint8_t)cp[k] : (uint8_t)0;
      if (p[k] > 0) atomicAdd(&myh[cp[k]], 1u);
      // (the last rank of the wave's run, and of the chunk: the successor belongs to another wave or block)
      const bool own_fetch = (lane == 63 && k == kCheckIPT - 1) || i + 1 >= end;
      if (!own_fetch) {
        if (qin[k] && cf[k] > cq[k]) bad = max(bad, 2);                 // utils.c:191-201
      } else if (i + 1 < n) {
        const u32 q = sa[i + 1];
        if (q < n && cf[k] > (u32)t[q]) bad = max(bad, 2);
      }
    }
  }
  if (bad) atomicMax(err, bad);
  __syncthreads();
  for (int j = threadIdx.x; j < 256; j += kBlock) {
    u32 sum = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) sum += hist[w][j];
    table[(size_t)j * nchunks + blockIdx.x] = sum;
  }
}
// One block of 256 threads: the bucket starts the pass found (start[c], exclusive prefix of its digit totals; *total = all
// records) against the text's histogram — bucket c of utils.c:204-211 minus the slot of the suffix n - 1 in its own
// bucket — and *q_out = that slot.  SA[q] must be n - 1 (the entry 0's claim, utils.c:219-221).
__global__ __launch_bounds__(256) void k_check_counts(const u32 *__restrict__ hist, const u32 *__restrict__ start, const u32 *__restrict__ total,
                                                     const uint8_t *__restrict__ t, const u32 *__restrict__ sa, u32 n, u32 *__restrict__ q_out, int *err) {
  __shared__ u32 tmp[4];
  const u32 c = threadIdx.x, last = t[n - 1];
  u32 tot;
  const u32 ex = block_excl_scan<4>(hist[c], tmp, tot);           // C[c] of utils.c:207-211
  const u32 want = ex - (c > last ? 1u : 0u);                     // ... in the dense array of the n - 1 records
  if (start[c] != want) atomicMax(err, 1);
  if (c == last) {
    *q_out = ex;
    if (*total != n - 1 || tot != n || sa[ex] != n - 1) atomicMax(err, 1);
  }
}
// the pass's records and where they must be found
struct CheckLoader {
  const u32 *sa; const uint8_t *bw; u32 n;
  __device__ __forceinline__ bool load(u32 i, Rec8 &r) const {
    const u32 p = sa[i];
    if (p == 0 || p >= n) return false;
    r.key = bw[i]; r.val = p - 1;
    return true;
  }
};
struct CheckSink {
  const u32 *sa; const u32 *q; u32 n; int *err;
  typedef u32 Fetched;
  __device__ __forceinline__ u32 slot(u32 g) const { return g + (g >= *q ? 1u : 0u); }     // utils.c:212-213: slot q is skipped
  __device__ __forceinline__ u32 fetch(u32 g) const { const u32 t = slot(g); return t < n ? sa[t] : 0xffffffffu; }
  __device__ __forceinline__ void check(u32 g, const Rec8 &x, u32 v) const {
    if (slot(g) >= n || v != x.val) atomicMax(err, 1);             // utils.c:222 "p != SA[t]"
  }
  __device__ __forceinline__ void store(u32 g, const Rec8 &x) const { check(g, x, fetch(g)); }
};
template <> struct SinkReads<CheckSink> { static constexpr bool value = true; };
__global__ __launch_bounds__(kBlock) void k_checksum(const u32 *__restrict__ sa, u32 n, u64 *out) {
  u64 acc = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    acc += splitmix64(((u64)i << 32) | sa[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}
// DC3HIP_TRACE: sum over k of mix(k, value_k); value = arr[k] (kind 0: positions), the level position of slot arr[k]
// (kind 1: 3s+1 for s < m0, else 3(s-m0)+2 — lib.rs:136-144), or the pos field of a mod-0 tuple (kind 2: Tup0, kind 3: Tup0C)
__global__ __launch_bounds__(kBlock) void k_trace_sum(const void *arr, u32 n, int kind, u32 m0, u64 *out) {
  u64 acc = 0;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    u32 v;
    if (kind == 2) v = static_cast<const Tup0 *>(arr)[i].pos;
    else if (kind == 3) v = static_cast<const u32 *>(arr)[(size_t)i * 4];          // compact mod-0 tuple (Tup0C): pos is word 0 of 4
    else {
      v = static_cast<const u32 *>(arr)[i];
      if (kind == 1) v = v < m0 ? 3 * v + 1 : 3 * (v - m0) + 2;
    }
    acc += splitmix64(((u64)i << 32) | v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}
// ---------------------------------------------------------------------------------------------
// By-products of the suffix array ("next" rows of the scope table).
// BWT: bw_transform()/divbwt() of crates/cdivsufsort/c-sources (utils.c:53-108, divsufsort.c:372-405):
//   U[0] = T[n-1]; then T[SA[i]-1] for every i with SA[i] != 0, in order; primary index = (i: SA[i]==0) + 1
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_find_zero(const u32 *__restrict__ sa, u32 n, u32 *zpos) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) if (sa[i] == 0) *zpos = i;
}
__global__ __launch_bounds__(kBlock) void k_bwt(const uint8_t *__restrict__ t, const u32 *__restrict__ sa, u32 n,
                                               const u32 *__restrict__ zpos, uint8_t *__restrict__ u) {
  const u32 z = *zpos;
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    if (i == z) continue;
    const uint8_t ch = t[sa[i] - 1];
    u[i < z ? i + 1 : i] = ch;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) u[0] = t[n - 1];
}

// Batched search = sacabase::longest_substring_match (crates/sacabase/src/lib.rs:39-99), one thread
// per needle, the reference's own narrowing loop (mid = len/2; needle > suffix(mid) ? right : left
// inclusive; 1-2 survivors compared by common prefix) so that (start, len) are identical.
__device__ __forceinline__ u32 d_common_prefix(const uint8_t *a, u64 la, const uint8_t *b, u64 lb) {
  const u64 l = la < lb ? la : lb;
  u64 i = 0;
  while (i < l && a[i] == b[i]) i++;
  return (u32)i;
}
__global__ __launch_bounds__(kBlock) void k_search(const uint8_t *__restrict__ t, u32 n, const u32 *__restrict__ sa,
                                                  const uint8_t *__restrict__ needles,
                                                  const int64_t *__restrict__ off, u32 q,
                                                  int64_t *__restrict__ out_start, int64_t *__restrict__ out_len) {
  const u32 id = blockIdx.x * kBlock + threadIdx.x;
  if (id >= q) return;
  const uint8_t *nd = needles + off[id];
  const u64 nl = (u64)(off[id + 1] - off[id]);
  u32 lo = 0, len = n;
  for (;;) {
    if (len == 1) {
      const u32 s = sa[lo];
      out_start[id] = s; out_len[id] = d_common_prefix(t + s, n - s, nd, nl);
      return;
    }
    if (len == 2) {
      const u32 s0 = sa[lo], s1 = sa[lo + 1];
      const u32 x = d_common_prefix(t + s0, n - s0, nd, nl), y = d_common_prefix(t + s1, n - s1, nd, nl);
      if (x > y) { out_start[id] = s0; out_len[id] = x; } else { out_start[id] = s1; out_len[id] = y; }
      return;
    }
    const u32 mid = len / 2;
    const u32 s = sa[lo + mid];
    const u64 sl = n - s;
    const u32 c = d_common_prefix(t + s, sl, nd, nl);
    // needle > suffix: first differing byte larger, or suffix is a proper prefix of the needle
    const bool gt = (c < nl && c < sl) ? (nd[c] > t[s + c]) : (nl > sl);
    if (gt) { lo += mid; len -= mid; } else { len = mid + 1; }
  }
}

// The narrowing loop of k_search on one (text, SA) pair: the reference's (start, len)
__device__ __forceinline__ void d_search_one(const uint8_t *__restrict__ t, u32 n, const u32 *__restrict__ sa, const uint8_t *nd, u64 nl,
                                             u32 &start, u32 &mlen) {
  u32 lo = 0, len = n;
  for (;;) {
    if (len == 1) { const u32 s = sa[lo]; start = s; mlen = d_common_prefix(t + s, n - s, nd, nl); return; }
    if (len == 2) {
      const u32 s0 = sa[lo], s1 = sa[lo + 1];
      const u32 x = d_common_prefix(t + s0, n - s0, nd, nl), y = d_common_prefix(t + s1, n - s1, nd, nl);
      if (x > y) { start = s0; mlen = x; } else { start = s1; mlen = y; }
      return;
    }
    const u32 mid = len / 2;
    const u32 s = sa[lo + mid];
    const u64 sl = n - s;
    const u32 c = d_common_prefix(t + s, sl, nd, nl);
    const bool gt = (c < nl && c < sl) ? (nd[c] > t[s + c]) : (nl > sl);
    if (gt) { lo += mid; len -= mid; } else { len = mid + 1; }
  }
}
// Batched sacapart::PartitionedSuffixArray::longest_substring_match (crates/sacapart/src/lib.rs:69-97), one thread per
// needle: sa holds the partition arrays back to back (chunk c = t[c*S .. min(n, (c+1)*S)), local indices).  Every
// partition is searched with the narrowing loop above; a match that reaches its partition's end is re-extended over the
// whole text (:77-84); the strictly longer match wins, so of equally long ones the first partition's stays (:86-92).
__global__ __launch_bounds__(kBlock) void k_search_partitioned(const uint8_t *__restrict__ t, u32 n, const u32 *__restrict__ sa, u32 S,
                                                              const uint8_t *__restrict__ needles, const int64_t *__restrict__ off, u32 q,
                                                              int64_t *__restrict__ out_start, int64_t *__restrict__ out_len) {
  const u32 id = blockIdx.x * kBlock + threadIdx.x;
  if (id >= q) return;
  const uint8_t *nd = needles + off[id];
  const u64 nl = (u64)(off[id + 1] - off[id]);
  bool have = false;
  u64 best_start = 0; u32 best_len = 0;
  for (u32 base = 0; base < n; base += S) {
    const u32 cn = min(S, n - base);
    u32 st, ln;
    d_search_one(t + base, cn, sa + base, nd, nl, st, ln);
    const bool may_extend = st + ln == cn;                                   // :77
    const u64 abs_start = (u64)base + st;                                    // :80
    if (may_extend) ln = d_common_prefix(t + abs_start, n - abs_start, nd, nl);   // :82-84
    if (!have || ln > best_len) { best_start = abs_start; best_len = ln; have = true; }
  }
  out_start[id] = (int64_t)best_start; out_len[id] = (int64_t)best_len;
}

// ---------------------------------------------------------------------------------------------
// LCP array of the resident SA ("next" row f.4): LCP[i] = lcp(suffix SA[i-1], suffix SA[i]), LCP[0] = 0.
// Kasai's argument in position order, PLCP[p] >= PLCP[p-1] - 1 with PLCP[p] = lcp(p, Phi[p]) and
// Phi[SA[i]] = SA[i-1], made parallel in two grains so that no thread ever restarts a long match from zero more
// than once per 2^16 positions:
//   k_plcp_coarse : one thread per 2^16 positions walks the positions s, s+L, s+2L, ... (L = 64) carrying
//                   h - L (PLCP[p+L] >= PLCP[p] - L);
//   k_plcp_fine   : one thread per L positions starts from the coarse value and carries h - 1;
//   k_lcp_gather  : LCP[i] = PLCP[SA[i]].
// Matches are extended 8 bytes at a time (the text buffer is padded; lengths are clamped to the text end).
// ---------------------------------------------------------------------------------------------
constexpr u32 kLcpFine = 64, kLcpCoarse = 1u << 16, kPhiNone = 0xffffffffu;
__global__ __launch_bounds__(kBlock) void k_phi_pairs(const u32 *__restrict__ sa, u32 n, Rec8 *__restrict__ pairs) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    pairs[i] = Rec8{sa[i], i > 0 ? sa[i - 1] : kPhiNone};          // (destination, value): Phi[sa[i]] = sa[i-1]
}
__global__ __launch_bounds__(kBlock) void k_phi_scatter(const u32 *__restrict__ sa, u32 n, u32 *__restrict__ phi) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 p = sa[i];
    if (p < n) phi[p] = (i > 0 && sa[i - 1] < n) ? sa[i - 1] : kPhiNone;   // tolerant of a foreign (set_sa) array
  }
}
__device__ __forceinline__ u32 lcp_extend(const uint8_t *__restrict__ t, u32 n, u32 p, u32 q, u32 h) {
  const u32 lim = n - max(p, q);                       // longest possible common prefix
  while (h + 8 <= lim) {
    u64 a, b;
    __builtin_memcpy(&a, t + p + h, 8); __builtin_memcpy(&b, t + q + h, 8);
    const u64 x = a ^ b;
    if (x) return h + (u32)(__ffsll((long long)x) - 1) / 8;
    h += 8;
  }
  while (h < lim && t[p + h] == t[q + h]) h++;
  return h;
}
// one WAVE per 2^16 positions: the 64 lanes extend a match 512 bytes per step, so even a from-scratch match of
// megabytes (runs, long repeats) is a short loop; the walk over the 1024 coarse positions is sequential
__global__ __launch_bounds__(kBlock) void k_plcp_coarse(const uint8_t *__restrict__ t, const u32 *__restrict__ phi, u32 n,
                                                       u32 *__restrict__ plcp) {
  const u32 g = blockIdx.x * kWaves + wave_id();
  const u32 lane = lane_id();
  const u64 first = (u64)g * kLcpCoarse;
  if (first >= n) return;
  u32 h = 0;
  for (u64 p64 = first; p64 < min((u64)n, first + kLcpCoarse); p64 += kLcpFine) {
    const u32 p = (u32)p64, q = phi[p];
    if (q == kPhiNone) h = 0;
    else {
      const u32 lim = n - max(p, q);
      h = min(h, lim);
      for (;;) {                                               // wave-uniform loop
        const u32 o = h + 8 * lane;                             // this lane's 8 bytes (zero padding past the text)
        u32 m = 8;                                              // matching bytes in my window, clamped to lim
        if (o >= lim) m = 0;
        else {
          u64 a, b;
          __builtin_memcpy(&a, t + p + o, 8); __builtin_memcpy(&b, t + q + o, 8);
          const u64 x = a ^ b;
          if (x) m = (u32)(__ffsll((long long)x) - 1) / 8;
          m = min(m, lim - o);
        }
        const u64 part = __ballot(m < 8);                       // lanes where the match ends
        if (part) {
          const u32 l = (u32)__ffsll((long long)part) - 1;
          h += 8 * l + (u32)__shfl(m, l);
          break;
        }
        h += 512;
      }
    }
    if (lane == 0) plcp[p] = h;
    h = h > kLcpFine ? h - kLcpFine : 0u;
  }
}
__global__ __launch_bounds__(kBlock) void k_plcp_fine(const uint8_t *__restrict__ t, const u32 *__restrict__ phi, u32 n,
                                                     u32 *__restrict__ plcp) {
  const u32 nseg = (n + kLcpFine - 1) / kLcpFine;
  for (u32 s = blockIdx.x * kBlock + threadIdx.x; s < nseg; s += gridDim.x * kBlock) {
    const u32 p0 = s * kLcpFine;
    u32 h = plcp[p0];
    const u32 pend = (u32)min((u64)n, (u64)p0 + kLcpFine);
    for (u32 p = p0 + 1; p < pend; p++) {
      h = h > 0 ? h - 1 : 0u;
      const u32 q = phi[p];
      h = q == kPhiNone ? 0u : lcp_extend(t, n, p, q, min(h, n - max(p, q)));
      plcp[p] = h;
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_lcp_gather(const u32 *__restrict__ sa, const u32 *__restrict__ plcp, u32 n,
                                                      int32_t *__restrict__ out) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 p = sa[i];
    out[i] = (i == 0 || p >= n) ? 0 : (int32_t)plcp[p];
  }
}

__global__ __launch_bounds__(kBlock) void k_widen(const u32 *__restrict__ in, int64_t *__restrict__ out, u32 n) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = (int64_t)in[i];
}

}  // namespace dc3
