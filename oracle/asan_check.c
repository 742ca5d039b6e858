/* asan_check.c — builds the oracle with AddressSanitizer + UBSan and drives it over adversarial sizes
 * (make -C oracle asan && oracle/asan_check).  Test infrastructure only. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int dc3_oracle_sufsort_i32(const uint8_t *T, int32_t *SA, int32_t n);
int dc3_oracle_sufsort_i64(const uint8_t *T, int64_t *SA, int64_t n);
int64_t oracle_verify_i32(const uint8_t *T, const int32_t *SA, int64_t n);
void oracle_gen_bytes(uint8_t *out, int64_t n, uint64_t seed, int kind);
int oracle_longest_substring_match_i32(const uint8_t *T, int64_t n, const int32_t *sa, int64_t sa_len, const uint8_t *needle,
                                       int64_t needle_len, int64_t *start, int64_t *len);
int main(void) {
  int bad = 0;
  for (int kind = 0; kind < 3; kind++) {
    for (int n = 0; n <= 700; n += (n < 40 ? 1 : 37)) {
      /* exact-size heap buffers so any over-read trips ASan */
      uint8_t *t = (uint8_t *)malloc(n ? n : 1);
      int32_t *sa = (int32_t *)malloc(sizeof(int32_t) * (n ? n : 1));
      oracle_gen_bytes(t, n, 17 + n, kind);
      if (kind == 0) for (int i = 0; i < n; i++) t[i] &= (n % 3 == 0) ? 0x01 : 0xff;   /* tiny alphabets too */
      if (dc3_oracle_sufsort_i32(t, sa, n) != 0) bad++;
      if (n > 1 && oracle_verify_i32(t, sa, n) != -1) bad++;
      if (n > 0) {
        int64_t st, ln;
        if (oracle_longest_substring_match_i32(t, n, sa, n, t + n / 2, n - n / 2, &st, &ln) != 0) bad++;
        if (ln != n - n / 2) bad++;
      }
      int64_t *sa64 = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
      if (dc3_oracle_sufsort_i64(t, sa64, n) != 0) bad++;
      for (int i = 0; i < n; i++) if (sa64[i] != sa[i]) bad++;
      free(sa64); free(sa); free(t);
    }
  }
  printf("asan_check: %s\n", bad ? "FAILED" : "ok");
  return bad != 0;
}
