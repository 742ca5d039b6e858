/*
 * dc3_oracle.c — CPU restatement of the reference's DC3/Skew suffix-array path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (stringsearch_amd/csrc)
 * never links, loads or calls anything in oracle/.
 *
 * What it restates (all citations relative to /root/reference):
 *   crates/dc3/src/lib.rs:3-5     leq2
 *   crates/dc3/src/lib.rs:9-11    leq3   (argument order FIXED to K–S order, see below)
 *   crates/dc3/src/lib.rs:15-39   radix_pass  (stable counting sort, K+1 counters)
 *   crates/dc3/src/lib.rs:44-193  suffix_array (steps 0..3, recursion)
 *   crates/sacabase/src/lib.rs:127-149  verify (adjacent suffixes strictly increasing)
 *   crates/sacabase/src/lib.rs:26-35,39-99  common_prefix_len / longest_substring_match
 *   crates/sacapart/src/lib.rs:39-58,69-97  partition sizes / partitioned search
 *   crates/cdivsufsort/c-sources/divsufsort.c:346-349  n in {0,1,2} + error codes
 *
 * Deliberate deviations from crates/dc3 as shipped (it is wrong as shipped):
 *   1. leq3 is declared (a1,a2,b1,b2,a3,b3) at lib.rs:9 but called as (a1,a2,a3,b1,b2,b3)
 *      at lib.rs:154-161.  We use the Kärkkäinen–Sanders order the call site intends.
 *   2. DC3 needs symbols in 1..K with three 0 sentinels (lib.rs:41-42).  The byte-level
 *      entry points shift bytes by +1 (alphabet 1..256) and pad internally, so callers pass
 *      exactly n bytes, like divsufsort().
 *   3. n in {0,1,2} (excluded by lib.rs:42 "n >= 2" / degenerate) follow divsufsort.c:346-349.
 *
 * Parity pinning: this restatement is checked in tests/test_oracle.py against
 *   (a) the reference's own corpus (the 11 files of crates/divsufsort/src/testdata) and the
 *       known-answer strings of the reference tests, with expected SAs produced by the
 *       reference's C libdivsufsort built from /root/reference (oracle/_ref, see Makefile)
 *       and committed under tests/golden/ by tests/golden/make_golden.py;
 *   (b) a naive O(n^2 log n) suffix sort on exhaustive/random small strings.
 *
 * Index type: OIDX (int32_t or int64_t), selected by the two instantiations at the bottom.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* Generic (64-bit working arrays, like the reference's usize) skew implementation.      */
/* ------------------------------------------------------------------------------------ */

typedef uint64_t usz; /* the reference uses usize for everything (lib.rs:50-57) */

/* lib.rs:3-5 */
static inline int leq2(usz a1, usz a2, usz b1, usz b2) {
  return (a1 < b1) || (a1 == b1 && a2 <= b2);
}
/* lib.rs:9-11, with the K–S parameter order (a1,a2,a3, b1,b2,b3) that lib.rs:154-161 passes */
static inline int leq3(usz a1, usz a2, usz a3, usz b1, usz b2, usz b3) {
  return (a1 < b1) || (a1 == b1 && leq2(a2, a3, b2, b3));
}

/* lib.rs:15-39 — stable counting sort of a[0..n) into b[0..n) by key r[a[i]], keys in 0..K */
static int radix_pass(const usz *a, usz *b, const usz *r, usz n, usz K) {
  usz *c = (usz *)calloc((size_t)K + 1, sizeof(usz)); /* :17 */
  if (!c) return -2;
  for (usz i = 0; i < n; i++) c[r[a[i]]]++;            /* :20-22 */
  usz sum = 0;                                          /* :25-32 */
  for (usz i = 0; i <= K; i++) { usz t = c[i]; c[i] = sum; sum += t; }
  for (usz i = 0; i < n; i++) b[c[r[a[i]]]++] = a[i];   /* :35-38 */
  free(c);
  return 0;
}

typedef struct {
  int     depth;      /* number of suffix_array invocations so far */
  int     cap;        /* capacity of the trace arrays (may be 0) */
  int64_t *n_at;      /* n of each invocation */
  int64_t *K_at;      /* K of each invocation */
  /* optional stage words (dc3_oracle_trace_ex), the same words libdc3hip reports with DC3HIP_TRACE=1: */
  int64_t  *names_at; /* `name` after the naming loop (lib.rs:80-100) */
  uint64_t *h_sa12;   /* sum_i mix(i, position of the i-th smallest sample), dummy included (after lib.rs:103-113) */
  uint64_t *h_sa0;    /* sum_p mix(p, SA0[p])   (after lib.rs:126) */
  uint64_t *h_sa;     /* sum_k mix(k, SA[k])    (after lib.rs:192) */
} dc3_trace;

static inline uint64_t trace_mix(uint64_t i, uint64_t v) {   /* splitmix64((i << 32) | v), as k_trace_sum */
  uint64_t x = (i << 32) | v;
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* lib.rs:44-193.  T has n+3 entries, T[n..n+3)=0, symbols in 1..K, n>=2. */
static int skew(const usz *T, usz *SA, usz n, usz K, dc3_trace *tr) {
  int lvl = -1;                                   /* this invocation's slot in the trace arrays */
  if (tr) {
    if (tr->depth < tr->cap) { lvl = tr->depth; tr->n_at[tr->depth] = (int64_t)n; tr->K_at[tr->depth] = (int64_t)K; }
    tr->depth++;
  }
  const usz n0 = (n + 2) / 3, n1 = (n + 1) / 3, n2 = n / 3, n02 = n0 + n2; /* :45-48 */
  int err = 0;
  usz *R    = (usz *)calloc((size_t)n02 + 3, sizeof(usz)); /* :50-53 (zero tail = next sentinel) */
  usz *SA12 = (usz *)calloc((size_t)n02 + 3, sizeof(usz)); /* :55 */
  usz *R0   = (usz *)calloc((size_t)n0 + 1, sizeof(usz));  /* :56 */
  usz *SA0  = (usz *)calloc((size_t)n0 + 1, sizeof(usz));  /* :57 */
  if (!R || !SA12 || !R0 || !SA0) { err = -2; goto done; }

  /* Step 0 (:62-70): positions of mod-1 and mod-2 suffixes; "+(n0-n1)" adds the dummy
   * mod-1 suffix at position n when n%3==1. */
  {
    usz j = 0;
    for (usz i = 0; i < n + (n0 - n1); i++) if (i % 3 != 0) R[j++] = i;
  }

  /* Step 1 (:74-76): LSD radix sort of the triples. */
  if ((err = radix_pass(R, SA12, T + 2, n02, K))) goto done;
  if ((err = radix_pass(SA12, R, T + 1, n02, K))) goto done;
  if ((err = radix_pass(R, SA12, T, n02, K))) goto done;

  /* Naming (:80-100). */
  usz name = 0;
  {
    usz c0 = 0, c1 = 0, c2 = 0; int first = 1;
    for (usz i = 0; i < n02; i++) {
      usz p = SA12[i];
      if (first || T[p] != c0 || T[p + 1] != c1 || T[p + 2] != c2) {
        first = 0; name++; c0 = T[p]; c1 = T[p + 1]; c2 = T[p + 2];
      }
      if (p % 3 == 1) R[p / 3] = name;        /* :93-95 left half  */
      else            R[p / 3 + n0] = name;   /* :96-98 right half */
    }
  }

  if (name < n02) {                              /* :103-108 */
    if ((err = skew(R, SA12, n02, name, tr))) goto done;
    for (usz i = 0; i < n02; i++) R[SA12[i]] = i + 1;
  } else {                                       /* :109-113 */
    for (usz i = 0; i < n02; i++) SA12[R[i] - 1] = i;
  }

  if (lvl >= 0 && tr->h_sa12) {
    uint64_t h = 0;
    for (usz i = 0; i < n02; i++) h += trace_mix(i, SA12[i] < n0 ? SA12[i] * 3 + 1 : (SA12[i] - n0) * 3 + 2);
    tr->h_sa12[lvl] = h; tr->names_at[lvl] = (int64_t)name;
  }

  /* Step 2 (:118-126): mod-0 suffixes, ordered by rank of suffix i+1, then by T[i]. */
  {
    usz j = 0;
    for (usz i = 0; i < n02; i++) if (SA12[i] < n0) R0[j++] = 3 * SA12[i];
    if ((err = radix_pass(R0, SA0, T, n0, K))) goto done;
  }
  if (lvl >= 0 && tr->h_sa0) {
    uint64_t h = 0;
    for (usz p = 0; p < n0; p++) h += trace_mix(p, SA0[p]);
    tr->h_sa0[lvl] = h;
  }

  /* Step 3 (:131-192): merge. t starts at n0-n1 to skip the dummy. */
  {
    usz p = 0, t = n0 - n1, k = 0;
#define GET_I() (SA12[t] < n0 ? SA12[t] * 3 + 1 : (SA12[t] - n0) * 3 + 2) /* :136-144 */
    while (k < n) {
      usz i = GET_I();
      usz j = SA0[p];
      int sa12_smaller =
          (SA12[t] < n0)
              ? leq2(T[i], R[SA12[t] + n0], T[j], R[j / 3])                                   /* :151-152 */
              : leq3(T[i], T[i + 1], R[SA12[t] - n0 + 1], T[j], T[j + 1], R[j / 3 + n0]);     /* :153-162 */
      if (sa12_smaller) {
        SA[k] = i; t++;
        if (t == n02) { for (k++; p < n0; p++, k++) SA[k] = SA0[p]; }                         /* :167-175 */
      } else {
        SA[k] = j; p++;
        if (p == n0) { for (k++; t < n02; t++, k++) SA[k] = GET_I(); }                        /* :180-188 */
      }
      k++;
    }
#undef GET_I
  }
  if (lvl >= 0 && tr->h_sa) {
    uint64_t h = 0;
    for (usz k = 0; k < n; k++) h += trace_mix(k, SA[k]);
    tr->h_sa[lvl] = h;
  }
done:
  free(R); free(SA12); free(R0); free(SA0);
  return err;
}

/* Byte-level driver shared by the i32/i64 entry points. Fills SAout via callback-free copy. */
static int dc3_bytes(const uint8_t *T, int64_t n, usz *SAw /* n entries */, dc3_trace *tr) {
  usz *W = (usz *)malloc(((size_t)n + 3) * sizeof(usz));
  if (!W) return -2;
  usz K = 0;
  for (int64_t i = 0; i < n; i++) { W[i] = (usz)T[i] + 1; if (W[i] > K) K = W[i]; } /* +1 shift, fact 4 */
  W[n] = W[n + 1] = W[n + 2] = 0;                                                    /* lib.rs:41-42 */
  int err = skew(W, SAw, (usz)n, K, tr);
  free(W);
  return err;
}

#define DEFINE_ENTRY(NAME, OIDX)                                                              \
  ORACLE_API int NAME(const uint8_t *T, OIDX *SA, OIDX n) {                                   \
    if (T == NULL || SA == NULL || n < 0) return -1;          /* divsufsort.c:346 */          \
    if (n == 0) return 0;                                     /* :347 */                      \
    if (n == 1) { SA[0] = 0; return 0; }                      /* :348 */                      \
    if (n == 2) { int m = (T[0] < T[1]); SA[m ^ 1] = 0; SA[m] = 1; return 0; } /* :349 */     \
    usz *SAw = (usz *)malloc((size_t)n * sizeof(usz));                                        \
    if (!SAw) return -2;                                                                      \
    int err = dc3_bytes(T, (int64_t)n, SAw, NULL);                                            \
    if (!err) for (OIDX i = 0; i < n; i++) SA[i] = (OIDX)SAw[i];                              \
    free(SAw);                                                                                \
    return err;                                                                               \
  }

DEFINE_ENTRY(dc3_oracle_sufsort_i32, int32_t)
DEFINE_ENTRY(dc3_oracle_sufsort_i64, int64_t)

/* Per-level trace: fills n_at/K_at (up to cap entries) with the arguments of every
 * suffix_array invocation; returns the number of invocations (levels) or <0 on error. */
ORACLE_API int dc3_oracle_trace(const uint8_t *T, int64_t n, int64_t *n_at, int64_t *K_at, int cap) {
  if (T == NULL || n < 3) return -1;
  usz *SAw = (usz *)malloc((size_t)n * sizeof(usz));
  if (!SAw) return -2;
  dc3_trace tr = {0, cap, n_at, K_at, NULL, NULL, NULL, NULL};
  int err = dc3_bytes(T, n, SAw, &tr);
  free(SAw);
  return err ? err : tr.depth;
}

/* The same with the stage words of every level (see dc3_trace): what libdc3hip reports with DC3HIP_TRACE=1. */
ORACLE_API int dc3_oracle_trace_ex(const uint8_t *T, int64_t n, int cap, int64_t *n_at, int64_t *K_at, int64_t *names_at,
                                   uint64_t *h_sa12, uint64_t *h_sa0, uint64_t *h_sa) {
  if (T == NULL || n < 3) return -1;
  usz *SAw = (usz *)malloc((size_t)n * sizeof(usz));
  if (!SAw) return -2;
  dc3_trace tr = {0, cap, n_at, K_at, names_at, h_sa12, h_sa0, h_sa};
  int err = dc3_bytes(T, n, SAw, &tr);
  free(SAw);
  return err ? err : tr.depth;
}

/* Symbol-level entry (the reference's own signature, lib.rs:44): T has n+3 u64 symbols in
 * 1..K with a zero tail.  Used to pin the restatement on integer alphabets. */
ORACLE_API int dc3_oracle_suffix_array_u64(const uint64_t *T, uint64_t *SA, uint64_t n, uint64_t K) {
  if (!T || !SA || n < 2) return -1;
  return skew(T, SA, n, K, NULL);
}

/* The reference's radix_pass, exposed for kernel-level parity tests (lib.rs:15-39). */
ORACLE_API int dc3_oracle_radix_pass_u64(const uint64_t *a, uint64_t *b, const uint64_t *r,
                                         uint64_t n, uint64_t K) {
  return radix_pass(a, b, r, n, K);
}

/* ------------------------------------------------------------------------------------ */
/* sacabase::verify restated (crates/sacabase/src/lib.rs:127-149):                       */
/*   for i in 0..len-1: require suffix(SA[i]) < suffix(SA[i+1]) (byte-wise, shorter      */
/*   prefix sorts first).  Returns -1 if sorted, else the first failing i.               */
/*   Also rejects out-of-range entries (Rust would panic on the slice index).            */
/* ------------------------------------------------------------------------------------ */
static int suffix_less(const uint8_t *T, int64_t n, int64_t a, int64_t b) {
  int64_t la = n - a, lb = n - b, l = la < lb ? la : lb;
  int c = memcmp(T + a, T + b, (size_t)l);
  if (c != 0) return c < 0;
  return la < lb;
}
#define DEFINE_VERIFY(NAME, OIDX)                                                             \
  ORACLE_API int64_t NAME(const uint8_t *T, const OIDX *SA, int64_t n) {                      \
    for (int64_t i = 0; i < n; i++) if (SA[i] < 0 || (int64_t)SA[i] >= n) return i;           \
    for (int64_t i = 0; i + 1 < n; i++)                                                       \
      if (!suffix_less(T, n, (int64_t)SA[i], (int64_t)SA[i + 1])) return i;                   \
    return -1;                                                                                \
  }
DEFINE_VERIFY(oracle_verify_i32, int32_t)
DEFINE_VERIFY(oracle_verify_i64, int64_t)

/* ------------------------------------------------------------------------------------ */
/* sacabase search restated (crates/sacabase/src/lib.rs:26-35, 39-99) — the "next" row    */
/* §8f-3; used to pin the host-side SuffixArray / PartitionedSuffixArray mirrors.         */
/* ------------------------------------------------------------------------------------ */
static int64_t common_prefix_len(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb) {
  int64_t n = la < lb ? la : lb;                       /* :29 */
  for (int64_t i = 0; i < n; i++) if (a[i] != b[i]) return i;
  return n;
}
/* needle > suffix, Rust slice ordering (lexicographic, shorter prefix is smaller) */
static int slice_gt(const uint8_t *a, int64_t la, const uint8_t *b, int64_t lb) {
  int64_t l = la < lb ? la : lb;
  int c = memcmp(a, b, (size_t)l);
  if (c != 0) return c > 0;
  return la > lb;
}
/* Returns start in *start, len in *len.  sa must be non-empty (the Rust loops forever on
 * an empty slice; we return -1). */
ORACLE_API int oracle_longest_substring_match_i32(const uint8_t *T, int64_t n, const int32_t *sa,
                                                  int64_t sa_len, const uint8_t *needle,
                                                  int64_t needle_len, int64_t *start, int64_t *len) {
  if (sa_len <= 0) return -1;
  const int32_t *s = sa; int64_t m = sa_len;
  for (;;) {
    if (m == 1) {                                       /* :75-77 */
      *start = s[0]; *len = common_prefix_len(T + s[0], n - s[0], needle, needle_len); return 0;
    } else if (m == 2) {                                /* :78-86 */
      int64_t x = common_prefix_len(T + s[0], n - s[0], needle, needle_len);
      int64_t y = common_prefix_len(T + s[1], n - s[1], needle, needle_len);
      if (x > y) { *start = s[0]; *len = x; } else { *start = s[1]; *len = y; }
      return 0;
    } else {                                            /* :87-95 */
      int64_t mid = m / 2;
      if (slice_gt(needle, needle_len, T + s[mid], n - s[mid])) { s += mid; m -= mid; }
      else { m = mid + 1; }
    }
  }
}

/* sacapart::PartitionedSuffixArray::longest_substring_match restated
 * (crates/sacapart/src/lib.rs:69-97).  sas[p] is the local SA of chunk p; chunk size
 * S = n/P + 1 (lib.rs:43); number of chunks = ceil(n/S). */
ORACLE_API int oracle_partitioned_match_i32(const uint8_t *T, int64_t n, const int32_t *const *sas,
                                            int64_t num_chunks, int64_t partition_size,
                                            const uint8_t *needle, int64_t needle_len,
                                            int64_t *start, int64_t *len) {
  int have = 0; int64_t bs = 0, bl = 0;
  for (int64_t p = 0; p < num_chunks; p++) {
    int64_t off = p * partition_size;
    int64_t clen = n - off < partition_size ? n - off : partition_size;
    int64_t st, ln;
    if (oracle_longest_substring_match_i32(T + off, clen, sas[p], clen, needle, needle_len, &st, &ln))
      return -1;
    int may_extend = (st + ln == clen);                 /* :77 */
    st += off;                                          /* :80 */
    if (may_extend) ln = common_prefix_len(T + st, n - st, needle, needle_len); /* :82-84 */
    if (!have || ln > bl) { have = 1; bs = st; bl = ln; } /* :86-92 strictly longer wins */
  }
  if (!have) return -1;                                 /* :93-95 expect() */
  *start = bs; *len = bl;
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* LCP array by Kasai et al. (checker for the "BWT/LCP by-products" row; the   */
/* reference has no LCP routine): lcp[0] = 0, lcp[i] = lcp(suffix sa[i-1], suffix sa[i]).    */
/* ------------------------------------------------------------------------------------ */
ORACLE_API int oracle_lcp_kasai_i32(const uint8_t *T, int64_t n, const int32_t *sa, int32_t *lcp) {
  if (n <= 0) return 0;
  int32_t *rank = (int32_t *)malloc((size_t)n * sizeof(int32_t));
  if (!rank) return -2;
  for (int64_t i = 0; i < n; i++) rank[sa[i]] = (int32_t)i;
  int64_t h = 0;
  lcp[0] = 0;
  for (int64_t p = 0; p < n; p++) {
    const int64_t r = rank[p];
    if (r == 0) { h = 0; continue; }
    const int64_t q = sa[r - 1];
    while (p + h < n && q + h < n && T[p + h] == T[q + h]) h++;
    lcp[r] = (int32_t)h;
    if (h > 0) h--;
  }
  free(rank);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Deterministic synthetic inputs: byte i = byte (i&7) of               */
/* splitmix64(seed + (i>>3)).  Host-side twin of the device generator, so the GPU and the */
/* CPU baseline see the same buffer without shipping files.                              */
/* ------------------------------------------------------------------------------------ */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* kind 2: low-entropy text with deep LCPs (made position-addressable):
 *   16-byte cells "word1 word2..." with words drawn from a 4096-word vocabulary by a skewed
 *   (product-of-uniforms) distribution, '\n' every 80 bytes; every 64 KiB block starts, with
 *   probability 1/2, with a verbatim 1-8 KiB copy of an earlier span. */
static uint8_t text_base(uint64_t i, uint64_t seed) {
  const uint64_t cell = i >> 4; const unsigned off = (unsigned)(i & 15);
  if (cell % 5 == 4 && off == 15) return '\n';
  const uint64_t hc = splitmix64(seed + cell * 0x9E3779B97F4A7C15ull);
  const uint64_t a = hc & 0xFFFF, b = (hc >> 16) & 0xFFFF, c = (hc >> 32) & 0xFFFF;
  const uint64_t wid = (((a * b) >> 16) * c) >> 20;                 /* 0..4095, skewed to small ids */
  const uint64_t hw = splitmix64(0x5EEDull ^ (wid << 1));
  const unsigned wlen = 2 + (unsigned)(hw % 11);                     /* 2..12 letters */
  if (off < wlen) return (uint8_t)('a' + ((hw >> (8 + 4 * off)) % 26));
  if (off == wlen) return ' ';
  const uint64_t wid2 = (((hc >> 48) & 0xFFF) * ((hc >> 40) & 0xFF)) >> 8;
  const uint64_t hw2 = splitmix64(0x5EEDull ^ (wid2 << 1));
  const unsigned wlen2 = 2 + (unsigned)(hw2 % 11), o2 = off - wlen - 1;
  if (o2 < wlen2) return (uint8_t)('a' + ((hw2 >> (8 + 4 * o2)) % 26));
  return ' ';
}
static uint8_t text_byte(uint64_t i, uint64_t seed) {
  const uint64_t block = i >> 16, within = i & 0xFFFF;
  if (block > 0) {
    const uint64_t hb = splitmix64((seed ^ 0xB10Cull) + block * 0xD1B54A32D192ED03ull);
    if (hb & 1) {
      const uint64_t len = 1024 + ((hb >> 8) % 7169);
      if (within < len) {
        const uint64_t sb = (hb >> 24) % block, so = (hb >> 44) % (65536 - 8192);
        return text_base(sb * 65536 + so + within, seed);
      }
    }
  }
  return text_base(i, seed);
}
/* kind 0: random bytes; kind 1: DNA "ACGT"[x&3] (2 bits per base); kind 2: text (above) */
ORACLE_API void oracle_gen_bytes(uint8_t *out, int64_t n, uint64_t seed, int kind) {
  static const char acgt[4] = {'A', 'C', 'G', 'T'};
  if (kind == 2) { for (int64_t i = 0; i < n; i++) out[i] = text_byte((uint64_t)i, seed); return; }
  if (kind == 0) {
    for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)(splitmix64(seed + (uint64_t)(i >> 3)) >> (8 * (i & 7)));
  } else {
    for (int64_t i = 0; i < n; i++) out[i] = (uint8_t)acgt[(splitmix64(seed + (uint64_t)(i >> 5)) >> (2 * (i & 31))) & 3];
  }
}
