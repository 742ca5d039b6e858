// host_test.cpp — the reference's own unit tests for the composition layer, restated against the
// C++ mirror with dc3hip::sort plugged in as the SACA (needs a GPU):
//   sacapart/src/lib.rs:105-128 worse_test, :130-165 equivalent_test, divsufsort/src/lib.rs:83-91 shruggy
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include "dc3hip.hpp"
#include "sacapart.hpp"
using sacabase::Bytes;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  {  // worse_test
    std::string input = "totor";
    auto sa_full = dc3hip::sort(Bytes(input));
    sacapart::PartitionedSuffixArray<int32_t> sa_part(Bytes(input), 2, dc3hip::sort);
    std::string needle = "tor";
    CHECK(sa_full.longest_substring_match(Bytes(needle)).as_bytes() == Bytes(needle));
    CHECK(sa_part.longest_substring_match(Bytes(needle)).as_bytes() == Bytes(std::string("to")));
    needle = "otor";
    CHECK(sa_full.longest_substring_match(Bytes(needle)).as_bytes() == Bytes(needle));
    CHECK(sa_part.longest_substring_match(Bytes(needle)).as_bytes() == Bytes(needle));
  }
  {  // equivalent_test
    std::string input = "This is a rather long text. We can probably find matches that span two partitions. Oh yes.";
    auto sa_full = dc3hip::sort(Bytes(input));
    for (size_t partitions : {1, 2, 3}) {
      for (std::string needle : {"rather long", "text. We can", "We can probably find matches that span"}) {
        sacapart::PartitionedSuffixArray<int32_t> sa_part(Bytes(input), partitions, dc3hip::sort);
        auto f = sa_full.longest_substring_match(Bytes(needle)), p = sa_part.longest_substring_match(Bytes(needle));
        CHECK(f.as_bytes() == p.as_bytes()); CHECK(f.start == p.start); CHECK(f.len == p.len);
      }
    }
  }
  {  // shruggy + verify
    const uint8_t sh[] = {0xc2, 0xaf, 0x5c, 0x5f, 0x28, 0xe3, 0x83, 0x84, 0x29, 0x5f, 0x2f, 0xc2, 0xaf};
    auto sa = dc3hip::sort(Bytes(sh, sizeof sh));
    sa.verify();
    const int32_t want[] = {4, 8, 10, 2, 3, 9, 6, 7, 12, 1, 11, 0, 5};
    for (size_t i = 0; i < sizeof sh; i++) CHECK(sa.sa()[i] == want[i]);
    auto sa64 = dc3hip::sort_i64(Bytes(sh, sizeof sh));
    for (size_t i = 0; i < sizeof sh; i++) CHECK(sa64.sa()[i] == want[i]);
  }
  {  // device-resident index: GPU batched search == CPU sacabase search, BWT of "banana"
    std::string input = "This is a rather long text. We can probably find matches that span two partitions. Oh yes.";
    dc3hip::DeviceIndex dev{Bytes(input)};
    CHECK(dev.sufcheck() == 0);
    auto host = dev.to_host();
    std::vector<std::string> needles = {"rather long", "text. We can", "zzz", "Oh yes.!", "T", ""};
    std::vector<Bytes> nb; for (auto &n : needles) nb.push_back(Bytes(n));
    auto got = dev.search(nb);
    for (size_t i = 0; i < needles.size(); i++) {
      auto want = host.longest_substring_match(nb[i]);
      CHECK(got[i].start == want.start); CHECK(got[i].len == want.len);
    }
    std::string b = "banana";
    dc3hip::DeviceIndex bi{Bytes(b)};
    std::vector<uint8_t> u; const int64_t pidx = bi.bwt(u);
    CHECK(std::string(u.begin(), u.end()) == "annbaa"); CHECK(pidx == 4);
    CHECK((bi.lcp() == std::vector<int32_t>{0, 1, 3, 0, 0, 2}));
  }
  {  // all partitions from one library call (DC3HIP_F_ALL_DEVICES) == one sort per chunk
    std::string input = "This is a rather long text. We can probably find matches that span two partitions. Oh yes.";
    for (size_t partitions = 1; partitions < 6; partitions++) {
      sacapart::PartitionedSuffixArray<int32_t> a(Bytes(input), partitions, dc3hip::sort);
      sacapart::PartitionedSuffixArray<int32_t> b(Bytes(input), partitions, dc3hip::sort_partitions(Bytes(input), partitions, true));
      CHECK(a.num_partitions() == b.num_partitions());
      for (size_t i = 0; i < a.num_partitions(); i++) CHECK(a.partitions()[i].sa() == b.partitions()[i].sa());
      CHECK(b.longest_substring_match(Bytes(std::string("find matches that span"))).len ==
            a.longest_substring_match(Bytes(std::string("find matches that span"))).len);
    }
  }
  {  // the partitioned index resident on the device: the same partitions and the same matches as the host-side mirror
    std::string input = "This is a rather long text. We can probably find matches that span two partitions. Oh yes.";
    std::vector<std::string> needles = {"rather long", "text. We can", "We can probably find matches that span", "zzz", "Oh yes.!", "s.", ""};
    std::vector<Bytes> nb; for (auto &n : needles) nb.push_back(Bytes(n));
    for (size_t partitions : {1, 2, 3, 7}) {
      sacapart::PartitionedSuffixArray<int32_t> host(Bytes(input), partitions, dc3hip::sort);
      dc3hip::DevicePartitionedIndex dev(Bytes(input), partitions);
      CHECK(dev.num_partitions() == host.num_partitions());
      sacapart::PartitionedSuffixArray<int32_t> again(Bytes(input), partitions, dev.flat());
      for (size_t i = 0; i < host.num_partitions(); i++) CHECK(host.partitions()[i].sa() == again.partitions()[i].sa());
      auto got = dev.search(nb);
      for (size_t i = 0; i < nb.size(); i++) {
        auto want = host.longest_substring_match(nb[i]);
        CHECK(got[i].start == want.start); CHECK(got[i].len == want.len);
      }
    }
    std::string t = "totor";
    dc3hip::DevicePartitionedIndex worse(Bytes(t), 2);
    CHECK(worse.longest_substring_match(Bytes(std::string("tor"))).as_bytes() == Bytes(std::string("to")));    // lib.rs:105-128
    CHECK(worse.longest_substring_match(Bytes(std::string("otor"))).as_bytes() == Bytes(std::string("otor")));
  }
  {  // global mode: one suffix array over 3 loopback ranks == the single-device array (and it verifies)
    std::string input;
    for (int i = 0; i < 4000; i++) input += "This is a rather long text. We can probably find matches that span two partitions. Oh yes. ";
    input += "tail";
    auto one = dc3hip::sort_i64(Bytes(input));
    setenv("DC3HIP_GLOBAL_LOCAL_MAX", "100", 1);
    dc3hip::GlobalLoopback grp(3, (int64_t)input.size());
    unsetenv("DC3HIP_GLOBAL_LOCAL_MAX");
    auto all = grp.sort(Bytes(input));
    CHECK(all.sa() == one.sa());
    CHECK(all.longest_substring_match(Bytes(std::string("matches that span two"))).len == 21);
  }
  {  // error behaviour: len mismatch throws like the Rust assert
    std::vector<int32_t> sa(2);
    bool threw = false;
    try { dc3hip::sort_in_place(Bytes(std::string("abc")), sa.data(), sa.size()); } catch (const std::invalid_argument &) { threw = true; }
    CHECK(threw);
  }
  std::printf("host_test ok\n");
  return 0;
}
