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
// GPU verifier = sufcheck() of crates/cdivsufsort/c-sources/utils.c:160-241 restated as parallel
// passes (equivalently sacabase::verify, sacabase/src/lib.rs:127-149).  With ISA = inverse of SA:
//   (1) range: every SA[i] in [0,n) (-2); ISA a bijection (a duplicate entry reports -4 like the reference's scan)
//   (2) first characters non-decreasing                                          (-3)
//   (3) for T[SA[i]] == T[SA[i+1]]: rank of suffix SA[i]+1 < rank of suffix SA[i+1]+1, the end
//       of text ranking lowest                                                   (-4)
// (1)-(3) hold iff SA is the suffix array.  err receives the smallest failing code seen
// (as in sufcheck, -2 is reported before -3 before -4).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_check_fill(const u32 *__restrict__ sa, u32 n, u32 *__restrict__ isa,
                                                      int *err) {
  for (u32 i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const u32 p = sa[i];
    if (p >= n) { atomicMax(err, 3); continue; }   // code = -(5 - v): 3 -> -2
    isa[p] = i + 1;
  }
}
__global__ __launch_bounds__(kBlock) void k_check_order(const uint8_t *__restrict__ t, const u32 *__restrict__ sa,
                                                       const u32 *__restrict__ isa, u32 n, int *err) {
  // a wave takes 64 consecutive ranks; the successor's position and first byte come from the next lane
  // (one random text gather per rank instead of two), lane 63 fetches its own
  const u32 lane = lane_id();
  for (u32 base = blockIdx.x * kBlock + (threadIdx.x & ~63u); base < n; base += gridDim.x * kBlock) {
    const u32 i = base + lane;
    const bool valid = i < n;
    const u32 p = valid ? sa[i] : 0xffffffffu;
    const bool pin = valid && p < n;
    const u32 cp = pin ? (u32)t[p] : 0u;
    u32 q = __shfl_down(p, 1), cq = __shfl_down(cp, 1);
    if (lane == 63) {
      q = (i + 1 < n) ? sa[i + 1] : 0xffffffffu;
      cq = (i + 1 < n && q < n) ? (u32)t[q] : 0u;
    }
    if (!pin) continue;                                       // out of range: reported by k_check_fill
    // an in-range array that is not a permutation (duplicate entries): the reference has no permutation test and
    // fails such arrays in its psi scan (-4, utils.c:213-238) unless the first characters already disagree (-3)
    if (isa[p] != i + 1) { atomicMax(err, 1); continue; }
    if (i + 1 >= n || q >= n) continue;
    if (cp > cq) { atomicMax(err, 2); continue; }           // -3
    if (cp == cq) {
      const u32 rp = (p + 1 < n) ? isa[p + 1] : 0u;
      const u32 rq = (q + 1 < n) ? isa[q + 1] : 0u;
      if (!(rp < rq)) atomicMax(err, 1);                    // -4
    }
  }
}
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
