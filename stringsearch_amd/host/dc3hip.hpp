// dc3hip.hpp — the SACA plug-in surface of the reference for the MI355X DC3 path, in C++:
//   dc3hip::sort_in_place(text, sa)   == cdivsufsort::sort_in_place  (crates/cdivsufsort/src/lib.rs:9-23)
//   dc3hip::sort(text)                == cdivsufsort::sort           (crates/cdivsufsort/src/lib.rs:26-30)
// A failing FFI call throws (the Rust asserts `ret == 0`, i.e. panics).  Links against libdc3hip.so.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/dc3hip.h"
#include "sacabase.hpp"

namespace dc3hip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char *msg) : std::runtime_error("dc3hip error " + std::to_string(c) + ": " + msg), code(c) {}
};

inline void sort_in_place(sacabase::Bytes text, int32_t *sa, size_t sa_len) {
  if (text.len != sa_len) throw std::invalid_argument("text and suffix array should have same len");        // :10-14
  if (text.len >= (size_t)INT32_MAX) throw std::invalid_argument("text too large, should not exceed 2147483646 bytes");  // :15-19
  static const uint8_t zero = 0; static int32_t dummy = 0;
  const int32_t ret = dc3hip_sufsort_i32(text.len ? text.ptr : &zero, sa_len ? sa : &dummy, (int32_t)text.len);
  if (ret != 0) throw Error(ret, dc3hip_last_error());                                                         // assert_eq!(0, ret), :22
}

inline sacabase::SuffixArray<int32_t> sort(sacabase::Bytes text) {
  sacabase::ZVec<int32_t> sa(text.len);                   // vec![0; text.len()], :27 (calloc: sacabase.hpp)
  sort_in_place(text, sa.data(), sa.size());
  return sacabase::SuffixArray<int32_t>(text, std::move(sa));
}

inline sacabase::SuffixArray<int64_t> sort_i64(sacabase::Bytes text) {
  sacabase::ZVec<int64_t> sa(text.len);
  static const uint8_t zero = 0; static int64_t dummy = 0;
  const int32_t ret = dc3hip_sufsort_i64(text.len ? text.ptr : &zero, sa.size() ? sa.data() : &dummy, (int64_t)text.len);
  if (ret != 0) throw Error(ret, dc3hip_last_error());
  return sacabase::SuffixArray<int64_t>(text, std::move(sa));
}

// The local SAs of a sacapart partitioning (chunk size len/P + 1, sacapart/src/lib.rs:43-46) from ONE library call,
// back to back; with all_devices the node's GPUs share the chunks (one host worker per GPU) — the par_chunks of
// lib.rs:45-49.  Feed the result to sacapart::PartitionedSuffixArray(text, P, flat).
inline std::vector<int32_t> sort_partitions(sacabase::Bytes text, size_t num_partitions, bool all_devices = true) {
  std::vector<int32_t> sa(text.len, 0);
  if (text.len == 0) return sa;
  dc3hip_opts o; o.struct_size = (int32_t)sizeof(o); o.index_bits = 32; o.device = -1;
  o.num_partitions = (int32_t)num_partitions; o.flags = all_devices ? DC3HIP_F_ALL_DEVICES : 0;
  const int32_t ret = dc3hip_sufsort_ex(text.ptr, sa.data(), (int64_t)text.len, &o);
  if (ret != 0) throw Error(ret, dc3hip_last_error());
  return sa;
}

// Device-resident index: text and SA stay in HBM; batched searches and the BWT run on the GPU.
// search() returns exactly what sacabase::longest_substring_match would (start, len) per needle.
class DeviceIndex : public sacabase::StringIndex {
  dc3hip_ctx *ctx_ = nullptr;
  sacabase::Bytes text_;
  static void check(int rc) { if (rc != 0) throw Error(rc, dc3hip_last_error()); }

 public:
  explicit DeviceIndex(sacabase::Bytes text, int device = -1) : text_(text) {
    check(dc3hip_ctx_create(&ctx_, device, (int64_t)text.len));
    try {
      check(dc3hip_ctx_set_text(ctx_, text.ptr, (int64_t)text.len));
      check(dc3hip_ctx_build(ctx_));
    } catch (...) { dc3hip_ctx_destroy(ctx_); throw; }
  }
  DeviceIndex(const DeviceIndex &) = delete;
  DeviceIndex &operator=(const DeviceIndex &) = delete;
  ~DeviceIndex() { dc3hip_ctx_destroy(ctx_); }

  sacabase::SuffixArray<int32_t> to_host() const {
    sacabase::ZVec<int32_t> sa(text_.len);
    check(dc3hip_ctx_get_sa_i32(ctx_, sa.data()));
    return sacabase::SuffixArray<int32_t>(text_, std::move(sa));
  }
  int32_t sufcheck() const { return dc3hip_ctx_sufcheck(ctx_); }
  std::vector<sacabase::LongestCommonSubstring> search(const std::vector<sacabase::Bytes> &needles) const {
    std::vector<int64_t> off(needles.size() + 1, 0);
    for (size_t i = 0; i < needles.size(); i++) off[i + 1] = off[i] + (int64_t)needles[i].len;
    std::vector<uint8_t> cat((size_t)off.back() + 1);
    for (size_t i = 0; i < needles.size(); i++) if (needles[i].len) std::memcpy(cat.data() + off[i], needles[i].ptr, needles[i].len);
    std::vector<int64_t> st(needles.size()), ln(needles.size());
    check(dc3hip_ctx_search(ctx_, cat.data(), off.data(), (int32_t)needles.size(), st.data(), ln.data()));
    std::vector<sacabase::LongestCommonSubstring> out;
    for (size_t i = 0; i < needles.size(); i++) out.push_back(sacabase::LongestCommonSubstring{text_, (size_t)st[i], (size_t)ln[i]});
    return out;
  }
  sacabase::LongestCommonSubstring longest_substring_match(sacabase::Bytes needle) const override {
    return search({needle})[0];
  }
  // bw_transform (utils.c:53-108): returns the primary index
  int64_t bwt(std::vector<uint8_t> &u) const {
    u.resize(text_.len);
    int64_t idx = 0;
    check(dc3hip_ctx_bwt(ctx_, u.data(), &idx));
    return idx;
  }
  // LCP array: lcp[0] = 0, lcp[i] = longest common prefix of the suffixes sa[i-1] and sa[i]
  std::vector<int32_t> lcp() const {
    std::vector<int32_t> out(text_.len);
    static int32_t dummy = 0;
    check(dc3hip_ctx_lcp_i32(ctx_, out.empty() ? &dummy : out.data()));
    return out;
  }
};

// sacapart::PartitionedSuffixArray (crates/sacapart/src/lib.rs:26-97) resident on one device: the P partition arrays are
// built there and every needle of a batch is searched in every partition by one kernel (re-extension over partition
// ends and the strictly-longer rule included) — the same (start, len) the host-side sacapart mirror returns.
class DevicePartitionedIndex : public sacabase::StringIndex {
  dc3hip_ctx *ctx_ = nullptr;
  sacabase::Bytes text_;
  int32_t parts_;
  static void check(int rc) { if (rc != 0) throw Error(rc, dc3hip_last_error()); }

 public:
  DevicePartitionedIndex(sacabase::Bytes text, size_t num_partitions, int device = -1) : text_(text), parts_((int32_t)num_partitions) {
    check(dc3hip_ctx_create(&ctx_, device, (int64_t)text.len));
    try {
      check(dc3hip_ctx_set_text(ctx_, text.ptr, (int64_t)text.len));
      check(dc3hip_ctx_build_partitions(ctx_, parts_));
    } catch (...) { dc3hip_ctx_destroy(ctx_); throw; }
  }
  DevicePartitionedIndex(const DevicePartitionedIndex &) = delete;
  DevicePartitionedIndex &operator=(const DevicePartitionedIndex &) = delete;
  ~DevicePartitionedIndex() { dc3hip_ctx_destroy(ctx_); }
  size_t partition_size() const { return text_.len / (size_t)parts_ + 1; }                      // lib.rs:43
  size_t num_partitions() const { return (text_.len + partition_size() - 1) / partition_size(); }   // :60
  // the partition arrays, back to back (what dc3hip_sufsort_ex(num_partitions = P) writes)
  std::vector<int32_t> flat() const {
    std::vector<int32_t> sa(text_.len);
    static int32_t dummy = 0;
    check(dc3hip_ctx_get_sa_i32(ctx_, sa.empty() ? &dummy : sa.data()));
    return sa;
  }
  std::vector<sacabase::LongestCommonSubstring> search(const std::vector<sacabase::Bytes> &needles) const {
    std::vector<int64_t> off(needles.size() + 1, 0);
    for (size_t i = 0; i < needles.size(); i++) off[i + 1] = off[i] + (int64_t)needles[i].len;
    std::vector<uint8_t> cat((size_t)off.back() + 1);
    for (size_t i = 0; i < needles.size(); i++) if (needles[i].len) std::memcpy(cat.data() + off[i], needles[i].ptr, needles[i].len);
    std::vector<int64_t> st(needles.size()), ln(needles.size());
    check(dc3hip_ctx_search_partitioned(ctx_, parts_, cat.data(), off.data(), (int32_t)needles.size(), st.data(), ln.data()));
    std::vector<sacabase::LongestCommonSubstring> out;
    for (size_t i = 0; i < needles.size(); i++) out.push_back(sacabase::LongestCommonSubstring{text_, (size_t)st[i], (size_t)ln[i]});
    return out;
  }
  sacabase::LongestCommonSubstring longest_substring_match(sacabase::Bytes needle) const override {
    return search({needle})[0];
  }
};

// GLOBAL mode (include/dc3hip.h): ONE suffix array of a text over P ranks.  GlobalLoopback = P ranks on one device in
// this process (tests, single-GPU boxes); a multi-process host builds one GlobalRank per GPU from a 128-byte RCCL id it
// distributes itself (MPI_Bcast, a file, ...).  Shards in rank order concatenate to what dc3hip::sort_i64 returns.
class GlobalRank {
  dc3hip_gctx *g_ = nullptr;
  friend class GlobalLoopback;
  explicit GlobalRank(dc3hip_gctx *g) : g_(g) {}

 public:
  static std::vector<uint8_t> rccl_unique_id() {
    std::vector<uint8_t> id(128);
    const int rc = dc3hip_rccl_unique_id(id.data());
    if (rc != 0) throw Error(rc, dc3hip_last_error());
    return id;
  }
  GlobalRank(const std::vector<uint8_t> &id128, int rank, int nranks, int device, int64_t max_total_n) {
    const int rc = dc3hip_global_rccl_create(&g_, id128.data(), rank, nranks, device, max_total_n);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
  }
  GlobalRank(GlobalRank &&o) noexcept : g_(o.g_) { o.g_ = nullptr; }
  GlobalRank(const GlobalRank &) = delete;
  ~GlobalRank() { if (g_) dc3hip_global_destroy(g_); }
  dc3hip_gctx *handle() const { return g_; }
  // this rank's bytes of a text of total_n bytes: [offset, offset + length)
  void block(int64_t total_n, int64_t *offset, int64_t *length) const {
    const int rc = dc3hip_global_block(g_, total_n, offset, length);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
  }
  void set_text_block(const uint8_t *block_bytes, int64_t total_n) {
    const int rc = dc3hip_global_set_text_block(g_, block_bytes, total_n);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
  }
  void build() {                                               // collective
    const int rc = dc3hip_global_build(g_);
    if (rc != 0) throw Error(rc, dc3hip_global_last_error(g_));
  }
  // wide contexts (texts of 2^32 bytes and more): the collective verifier; 0 = the shards are the suffix array
  int sufcheck() {
    const int rc = dc3hip_global_sufcheck(g_);
    if (rc <= -11) throw Error(rc + 10, dc3hip_global_last_error(g_));
    return rc;
  }
  // SA[first .. first + shard.size())
  std::vector<int64_t> shard(int64_t *first) const {
    int64_t cnt = 0;
    int rc = dc3hip_global_shard(g_, first, &cnt);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
    std::vector<int64_t> out((size_t)cnt);
    static int64_t dummy = 0;
    rc = dc3hip_global_get_shard_i64(g_, cnt ? out.data() : &dummy);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
    return out;
  }
};

class GlobalLoopback {
  std::vector<dc3hip_gctx *> h_;

 public:
  GlobalLoopback(int nranks, int64_t max_total_n, int device = -1) : h_((size_t)nranks, nullptr) {
    const int rc = dc3hip_global_loopback_create(h_.data(), nranks, device, max_total_n);
    if (rc != 0) throw Error(rc, dc3hip_last_error());
  }
  GlobalLoopback(const GlobalLoopback &) = delete;
  ~GlobalLoopback() { for (auto *g : h_) if (g) dc3hip_global_destroy(g); }
  // What the last build would take on P GPUs (the ranks here may share one): the slowest rank's own work + the transport
  // priced per collective at the most bytes a rank exchanges with one peer over a 153 GB/s xGMI link (dc3hip_gstats).
  // (Own work of ranks that share a device: create the group under DC3HIP_DEBUG=global_device_token,global_link_gbps=153
  //  and ask after a second build — the first one of a group also allocates; sa_bench --global-ranks does both.)
  double predicted_wall_ms(double *max_work_ms = nullptr, double *max_link_ms = nullptr) const {
    double w = 0, l = 0;
    for (auto *g : h_) {
      dc3hip_gstats gs;
      if (dc3hip_global_stats(g, &gs, nullptr) != 0) throw Error(-3, dc3hip_last_error());
      if (gs.work_ms > w) w = gs.work_ms;
      if (gs.link_ms > l) l = gs.link_ms;
    }
    if (max_work_ms) *max_work_ms = w;
    if (max_link_ms) *max_link_ms = l;
    return w + l;
  }
  // the whole text lives on this host: every rank takes its block
  sacabase::SuffixArray<int64_t> sort(sacabase::Bytes text) {
    for (auto *g : h_) {
      int64_t off = 0, len = 0;
      int rc = dc3hip_global_block(g, (int64_t)text.len, &off, &len);
      if (rc == 0) rc = dc3hip_global_set_text_block(g, text.ptr + off, (int64_t)text.len);
      if (rc != 0) throw Error(rc, dc3hip_last_error());
    }
    const int rc = dc3hip_global_loopback_build(h_.data(), (int32_t)h_.size());
    if (rc != 0) throw Error(rc, dc3hip_last_error());
    sacabase::ZVec<int64_t> sa;
    sa.reserve(text.len);
    for (auto *g : h_) {
      GlobalRank r(g);
      int64_t first = 0;
      std::vector<int64_t> part = r.shard(&first);
      r.g_ = nullptr;                                          // (borrowed handle)
      if ((size_t)first != sa.size()) throw std::runtime_error("global mode: shards do not tile the suffix array");
      sa.insert(sa.end(), part.begin(), part.end());
    }
    return sacabase::SuffixArray<int64_t>(text, std::move(sa));
  }
};

}  // namespace dc3hip
