// Host-only lifecycle fuzz of libdc3hip's C ABI on the mock runtime (hip_mock.cpp), run under ASan and TSan in the CPU
// container (tools/hostmock/Makefile).  Kernels do not run, so every build beyond n = 2 fails loudly at its first
// device-to-host read; what is exercised for real is everything that owns host memory and threads: loopback groups
// (LoopWorld, LoopComm, the persistent rank threads), contexts, the pinned-buffer pool, the per-thread one-shot cache,
// failed collectives (abort / leave / reset), teardown in any order, and several host threads doing all of that at once
// (sacapart's callers are rayon workers, crates/sacapart/src/lib.rs:45-49).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>
#include "../../include/dc3hip.h"

#include <atomic>
static std::atomic<int> g_groups{0}, g_ctxs{0}, g_builds_failed{0}, g_builds_ok{0};

static void one_group(std::mt19937_64 &rng) {
  const int P = 1 + (int)(rng() % 16);
  const int64_t cap = 1 + (int64_t)(rng() % 400000);
  std::vector<dc3hip_gctx *> ranks((size_t)P, nullptr);
  if (dc3hip_global_loopback_create(ranks.data(), P, 0, cap) != 0) { std::fprintf(stderr, "create failed: %s\n", dc3hip_last_error()); std::abort(); }
  g_groups++;
  const int rounds = 1 + (int)(rng() % 3);
  for (int it = 0; it < rounds; it++) {
    const int64_t n = (rng() % 4 == 0) ? (int64_t)(rng() % 3) : 1 + (int64_t)(rng() % (uint64_t)cap);
    std::vector<uint8_t> text((size_t)n);
    for (auto &b : text) b = (uint8_t)(rng() % 4);
    for (int r = 0; r < P; r++) {
      int64_t off = 0, len = 0;
      if (dc3hip_global_block(ranks[(size_t)r], n, &off, &len) != 0) std::abort();
      if (dc3hip_global_set_text_block(ranks[(size_t)r], len ? text.data() + off : nullptr, n) != 0) { std::fprintf(stderr, "set_text: %s\n", dc3hip_last_error()); std::abort(); }
    }
    const int rc = dc3hip_global_loopback_build(ranks.data(), P);
    if (rc == 0) g_builds_ok++; else { g_builds_failed++; (void)dc3hip_global_last_error(ranks[0]); }
    int64_t first = 0, cnt = 0;
    for (int r = 0; r < P; r++) {
      if (dc3hip_global_shard(ranks[(size_t)r], &first, &cnt) == 0 && cnt > 0) {
        std::vector<int64_t> out((size_t)cnt);
        (void)dc3hip_global_get_shard_i64(ranks[(size_t)r], out.data());
      }
      dc3hip_gstats gs; dc3hip_stats st;
      (void)dc3hip_global_stats(ranks[(size_t)r], &gs, &st);
    }
  }
  // teardown in a random order
  for (int i = P - 1; i > 0; i--) std::swap(ranks[(size_t)i], ranks[(size_t)(rng() % (uint64_t)(i + 1))]);
  for (auto *g : ranks) dc3hip_global_destroy(g);
}

static void one_ctx(std::mt19937_64 &rng) {
  dc3hip_ctx *c = nullptr;
  const int64_t cap = 1 + (int64_t)(rng() % 300000);
  if (dc3hip_ctx_create(&c, 0, cap) != 0) { std::fprintf(stderr, "ctx_create failed: %s\n", dc3hip_last_error()); std::abort(); }
  g_ctxs++;
  const int64_t n = (rng() % 3 == 0) ? (int64_t)(rng() % 3) : 1 + (int64_t)(rng() % (uint64_t)cap);
  std::vector<uint8_t> text((size_t)n, 7);
  if (dc3hip_ctx_set_text(c, text.data(), n) != 0) std::abort();
  if (dc3hip_ctx_build(c) == 0) g_builds_ok++; else g_builds_failed++;
  dc3hip_stats st;
  (void)dc3hip_ctx_stats(c, &st);
  dc3hip_ctx_destroy(c);
  // the one-shot entry point with its per-thread context cache
  std::vector<int32_t> sa((size_t)n);
  (void)dc3hip_sufsort_i32(text.data(), sa.data(), (int32_t)n);
  if (rng() % 4 == 0) dc3hip_release_cache();
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int workers = argc > 2 ? atoi(argv[2]) : 4;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  std::vector<std::thread> th;
  for (int w = 0; w < workers; w++)
    th.emplace_back([=] {
      std::mt19937_64 rng(seed * 1000 + (uint64_t)w);
      for (int i = 0; i < iters; i++) { if (w == 0 || rng() % 3 == 0) one_group(rng); else one_ctx(rng); }
      dc3hip_release_cache();
    });
  for (auto &t : th) t.join();
  std::printf("{\"lifecycle_iterations\": %d, \"workers\": %d, \"groups\": %d, \"contexts\": %d, \"builds_failed_as_expected\": %d, \"builds_ok\": %d}\n",
              iters, workers, g_groups.load(), g_ctxs.load(), g_builds_failed.load(), g_builds_ok.load());
  return 0;
}
