// sacapart.hpp — C++ mirror of crates/sacapart/src/lib.rs: split the text into P chunks, build an
// independent suffix array per chunk CONCURRENTLY (the reference uses rayon's par_chunks, lib.rs:45-49;
// here one std::thread per chunk, so `f` must be re-entrant exactly as `F: Sync` demands), search all
// partitions and stitch matches that touch a partition end.
#pragma once
#include <exception>
#include <functional>
#include <thread>
#include "sacabase.hpp"

namespace sacapart {

template <class Index>
class PartitionedSuffixArray : public sacabase::StringIndex {
  size_t partition_size_;
  sacabase::Bytes text_;
  std::vector<sacabase::SuffixArray<Index>> sas_;

 public:
  using SortFn = std::function<sacabase::SuffixArray<Index>(sacabase::Bytes)>;

  // lib.rs:39-58
  PartitionedSuffixArray(sacabase::Bytes text, size_t num_partitions, SortFn f)
      : partition_size_(text.len / num_partitions + 1), text_(text) {
    std::vector<sacabase::Bytes> chunks;
    for (size_t off = 0; off < text.len; off += partition_size_)
      chunks.push_back(text.slice(off, std::min(text.len, off + partition_size_)));
    std::vector<sacabase::ZVec<Index>> out(chunks.size());
    std::vector<std::exception_ptr> err(chunks.size());
    std::vector<std::thread> th;
    for (size_t i = 0; i < chunks.size(); i++)
      th.emplace_back([&, i] {
        try { out[i] = std::move(f(chunks[i])).into_parts().second; } catch (...) { err[i] = std::current_exception(); }
      });
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
    for (size_t i = 0; i < chunks.size(); i++) sas_.emplace_back(chunks[i], std::move(out[i]));   // chunk order, :50-51
  }

  // Same object from partition SAs that were built elsewhere, back to back in `flat` (chunk c at offset c*S, local
  // indices) — what dc3hip_sufsort_ex(num_partitions = P) writes.
  PartitionedSuffixArray(sacabase::Bytes text, size_t num_partitions, const std::vector<Index> &flat)
      : partition_size_(text.len / num_partitions + 1), text_(text) {
    if (flat.size() != text.len) throw std::invalid_argument("flat partition SAs should have the text's length");
    for (size_t off = 0; off < text.len; off += partition_size_) {
      const size_t end = std::min(text.len, off + partition_size_);
      sas_.emplace_back(text.slice(off, end), std::vector<Index>(flat.begin() + off, flat.begin() + end));
    }
  }

  size_t num_partitions() const { return sas_.size(); }   // :60
  size_t partition_size() const { return partition_size_; }
  const std::vector<sacabase::SuffixArray<Index>> &partitions() const { return sas_; }

  // lib.rs:69-97
  sacabase::LongestCommonSubstring longest_substring_match(sacabase::Bytes needle) const override {
    bool have = false;
    sacabase::LongestCommonSubstring best;
    for (size_t i = 0; i < sas_.size(); i++) {
      sacabase::LongestCommonSubstring lcs = sas_[i].longest_substring_match(needle);
      const size_t offset = i * partition_size_;
      const bool may_extend = lcs.start + lcs.len == sas_[i].text().len;    // :77
      lcs.start += offset;                                                  // :80
      lcs.text = text_;
      if (may_extend) lcs.len = sacabase::common_prefix_len(text_.slice_from(lcs.start), needle);   // :82-84
      if (!have || lcs.len > best.len) { best = lcs; have = true; }         // :86-92: strictly longer wins
    }
    if (!have) throw std::runtime_error("partitioned suffix arrays should always find at least one longest common substring");
    return best;
  }
};

}  // namespace sacapart
