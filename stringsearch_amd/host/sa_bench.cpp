// sa_bench.cpp — the reference's harness (crates/divsuftest/src/main.rs) restated in C++ for the
// MI355X path:   sa_bench bench|run|verify INPUT [LENGTH]        (main.rs:77-80; `crosscheck` becomes
// `verify`: build, then sacabase::verify + GPU sufcheck).  INPUT is a file, or gen:random:SIZE:SEED /
// gen:dna:SIZE:SEED (the deterministic generator of BASELINE.md §3).  LENGTH takes k/m suffixes
// (main.rs:192-208).  `bench` prints the table of main.rs:168-188: one un-warmed timed call per
// algorithm, SA allocation included (main.rs:145-151), speed = len / secs in binary units.
// Rows: dc3-hip (the FFI entry point), dc3-hip-resident (context API, text/SA in HBM), with
// --ref /path/to/libdivsufsort.so c-divsufsort (dlopen'ed; never linked), and with --global-ranks P
// dc3-hip-global(P): ONE suffix array over P loopback ranks (incl. creating the ranks and the comparison
// with the one-shot result).
#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <fstream>
#include <sys/stat.h>
#include <functional>
#include <iostream>
#include <string>
#include <vector>
#include "dc3hip.hpp"
#include "sacapart.hpp"

static void usage() { std::printf("Usage: sa_bench bench|run|verify INPUT [LENGTH] [--ref LIBDIVSUFSORT.so] [--partitions P [--all-devices]] [--global-ranks P]\n"); std::exit(1); }

// main.rs:192-208
static size_t parse_size(std::string s) {
  for (auto &ch : s) ch = (char)std::tolower(ch);
  size_t factor = 1;
  if (!s.empty() && s.back() == 'k') { factor = 1024; s.pop_back(); }
  else if (!s.empty() && s.back() == 'm') { factor = 1024 * 1024; s.pop_back(); }
  return (size_t)std::stoull(s) * factor;
}
static std::string fmt_binary(double v) {   // SizeFormatterBinary
  const char *u[] = {"", "Ki", "Mi", "Gi", "Ti"}; int k = 0;
  while (v >= 1024.0 && k < 4) { v /= 1024.0; k++; }
  char buf[64]; std::snprintf(buf, sizeof buf, k ? "%.1f%s" : "%.0f%s", v, u[k]); return buf;
}
static uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

// limit: at most this many bytes are read (LENGTH of the command line; ~0 = the whole file).  A stream without an end
// (/dev/urandom, a pipe) must come with a LENGTH: reading "all of it" exhausts the host's memory.
static std::vector<uint8_t> load_input(const std::string &spec, size_t limit) {
  if (spec.rfind("gen:", 0) == 0) {
    const size_t a = spec.find(':', 4), b = spec.find(':', a + 1);
    if (a == std::string::npos || b == std::string::npos) usage();
    const std::string kind = spec.substr(4, a - 4);
    const size_t n = parse_size(spec.substr(a + 1, b - a - 1));
    const uint64_t seed = std::stoull(spec.substr(b + 1));
    std::vector<uint8_t> t(n);
    if (kind == "random") for (size_t i = 0; i < n; i++) t[i] = (uint8_t)(splitmix64(seed + (i >> 3)) >> (8 * (i & 7)));
    else if (kind == "dna") for (size_t i = 0; i < n; i++) t[i] = (uint8_t)"ACGT"[(splitmix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3];
    else usage();
    return t;
  }
  std::ifstream f(spec, std::ios::binary);
  if (!f) { std::fprintf(stderr, "cannot read %s\n", spec.c_str()); std::exit(1); }
  struct stat st;
  const bool regular = stat(spec.c_str(), &st) == 0 && S_ISREG(st.st_mode);
  if (!regular && limit == ~(size_t)0) { std::fprintf(stderr, "%s is not a regular file: give a LENGTH\n", spec.c_str()); std::exit(1); }
  const size_t want = regular ? std::min<size_t>(limit, (size_t)st.st_size) : limit;
  std::vector<uint8_t> t(want);
  f.read(reinterpret_cast<char *>(t.data()), (std::streamsize)want);
  t.resize((size_t)f.gcount());
  return t;
}

int main(int argc, char **argv) {
  std::vector<std::string> free_args; std::string ref_path; size_t partitions = 0; bool all_devices = false; int global_ranks = 0;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--ref" && i + 1 < argc) ref_path = argv[++i];
    else if (a == "--partitions" && i + 1 < argc) partitions = (size_t)std::atoll(argv[++i]);
    else if (a == "--all-devices") all_devices = true;
    else if (a == "--global-ranks" && i + 1 < argc) global_ranks = std::atoi(argv[++i]);   // bench: extra row, ONE SA over P loopback ranks
    else free_args.push_back(a);
  }
  if (free_args.size() < 2) usage();
  const std::string cmd = free_args[0];
  if (cmd != "bench" && cmd != "run" && cmd != "verify") { std::printf("Command should be one of bench, run or verify\n"); return 1; }
  const size_t limit = free_args.size() > 2 ? parse_size(free_args[2]) : ~(size_t)0;
  std::vector<uint8_t> full = load_input(free_args[1], limit);
  const size_t len = free_args.size() > 2 ? limit : full.size();
  if (len > full.size()) { std::fprintf(stderr, "LENGTH exceeds input\n"); return 1; }
  sacabase::Bytes input(full.data(), len);
  std::printf("Input is size %sB\n", fmt_binary((double)len).c_str());            // main.rs:52-55

  using clk = std::chrono::steady_clock;
  try {
    if (cmd == "run") {                                                           // main.rs:115-121
      const auto t0 = clk::now();
      if (partitions > 1 && all_devices)      // one library call, the node's GPUs share the chunks
        sacapart::PartitionedSuffixArray<int32_t> p(input, partitions, dc3hip::sort_partitions(input, partitions, true));
      else if (partitions > 1) sacapart::PartitionedSuffixArray<int32_t> p(input, partitions, dc3hip::sort);
      else dc3hip::sort(input);
      std::printf("Done in %.6fs\n", std::chrono::duration<double>(clk::now() - t0).count());
      return 0;
    }
    if (cmd == "verify") {                                                        // main.rs:82-113 (verify part)
      std::printf("Running dc3-hip...\n");
      auto sa = dc3hip::sort(input);
      std::printf("Verifying result (sacabase::verify)...\n");
      sa.verify();
      const int32_t rc = dc3hip_sufcheck_i32(input.ptr, sa.sa().data(), (int32_t)len);
      std::printf("GPU sufcheck: %d\n", rc);
      return rc == 0 ? 0 : 2;
    }
    // bench: main.rs:123-190
    struct Row { std::string name; double secs; };
    std::vector<Row> rows;
    auto measure = [&](const char *name, const std::function<void()> &f) {
      std::printf("."); std::fflush(stdout);
      const auto t0 = clk::now(); f();
      rows.push_back({name, std::chrono::duration<double>(clk::now() - t0).count()});
    };
    std::printf("measuring"); std::fflush(stdout);
    if (!ref_path.empty()) {
      void *h = dlopen(ref_path.c_str(), RTLD_NOW);
      if (!h) { std::fprintf(stderr, "\ncannot dlopen %s: %s\n", ref_path.c_str(), dlerror()); return 1; }
      auto fn = reinterpret_cast<int32_t (*)(const uint8_t *, int32_t *, int32_t)>(dlsym(h, "divsufsort"));
      if (!fn) { std::fprintf(stderr, "\nno divsufsort symbol in %s\n", ref_path.c_str()); return 1; }
      measure("c-divsufsort", [&] { sacabase::ZVec<int32_t> sa(len); if (fn(input.ptr, sa.data(), (int32_t)len) != 0) std::abort(); });
    }
    measure("dc3-hip", [&] { dc3hip::sort(input); });
    double resident_ms = 0;
    std::string global_note;
    measure("dc3-hip-resident", [&] {
      dc3hip_ctx *c = nullptr;
      if (dc3hip_ctx_create(&c, -1, (int64_t)len) || dc3hip_ctx_set_text(c, input.ptr, (int64_t)len) || dc3hip_ctx_build(c))
        throw dc3hip::Error(-3, dc3hip_last_error());
      dc3hip_stats st; dc3hip_ctx_stats(c, &st); resident_ms = st.build_ms;
      dc3hip_ctx_destroy(c);
    });
    if (global_ranks > 1) {
      // the global mode (one suffix array over P ranks) with the P ranks as loopback ranks on this GPU; checked against
      // the one-shot result
      bool same = false;
      double pred = 0, pw = 0, pl = 0;
      measure(("dc3-hip-global(" + std::to_string(global_ranks) + ")").c_str(), [&] {
        dc3hip::GlobalLoopback grp(global_ranks, (int64_t)len);
        auto all = grp.sort(input);
        pred = grp.predicted_wall_ms(&pw, &pl);
        auto one = dc3hip::sort_i64(input);
        same = all.sa() == one.sa();
      });
      if (!same) { std::fprintf(stderr, "\nglobal-mode result differs from the single-device result\n"); return 2; }
      {
        // the prediction for P GPUs, from a second (untimed) build in which ranks that share this device work one at a time
        // (device token) and the select-or-route policy sees the xGMI link rate: a rank's work_ms is then its OWN work
        const char *old_dbg = std::getenv("DC3HIP_DEBUG");
        const std::string keep = old_dbg ? old_dbg : "";
        setenv("DC3HIP_DEBUG", (keep.empty() ? std::string() : keep + ",").append("global_device_token,global_link_gbps=153").c_str(), 1);
        {
          dc3hip::GlobalLoopback grp(global_ranks, (int64_t)len);
          (void)grp.sort(input);                 // (the first build of a group also allocates: not a rank's steady work)
          (void)grp.sort(input);
          pred = grp.predicted_wall_ms(&pw, &pl);
        }
        if (old_dbg) setenv("DC3HIP_DEBUG", keep.c_str(), 1); else unsetenv("DC3HIP_DEBUG");
      }
      global_note = "dc3-hip-global(" + std::to_string(global_ranks) + ") on " + std::to_string(global_ranks) + " GPUs, predicted: " + std::to_string(pred) +
                    " ms = slowest rank's own work " + std::to_string(pw) + " ms + transport " + std::to_string(pl) + " ms at 153 GB/s per xGMI link (the row above: ranks time-sharing this GPU, incl. host transfers)";
    }
    std::printf("done!\n");
    std::printf("%-20s %-14s %s\n", "Algorithm", "Time", "Average speed");
    for (auto &r : rows) std::printf("%-20s %-14s %sB/s\n", r.name.c_str(), (std::to_string(r.secs) + "s").c_str(), fmt_binary((double)len / r.secs).c_str());
    if (!global_note.empty()) std::printf("%s\n", global_note.c_str());
    std::printf("%-20s %-14s %sB/s   (HIP-event time of the device-resident build only)\n", "dc3-hip-kernels", (std::to_string(resident_ms / 1e3) + "s").c_str(),
                fmt_binary((double)len / (resident_ms / 1e3)).c_str());
  } catch (const std::exception &e) {
    std::fprintf(stderr, "\nerror: %s\n", e.what());
    return 1;
  }
  return 0;
}
