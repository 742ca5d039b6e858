// dc3hip.hpp — the SACA plug-in surface of the reference for the MI355X DC3 path, in C++:
//   dc3hip::sort_in_place(text, sa)   == cdivsufsort::sort_in_place  (crates/cdivsufsort/src/lib.rs:9-23)
//   dc3hip::sort(text)                == cdivsufsort::sort           (crates/cdivsufsort/src/lib.rs:26-30)
// A failing FFI call throws (the Rust asserts `ret == 0`, i.e. panics).  Links against libdc3hip.so.
#pragma once
#include <stdexcept>
#include <string>
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
  std::vector<int32_t> sa(text.len, 0);                   // vec![0; text.len()], :27
  sort_in_place(text, sa.data(), sa.size());
  return sacabase::SuffixArray<int32_t>(text, std::move(sa));
}

inline sacabase::SuffixArray<int64_t> sort_i64(sacabase::Bytes text) {
  std::vector<int64_t> sa(text.len, 0);
  static const uint8_t zero = 0; static int64_t dummy = 0;
  const int32_t ret = dc3hip_sufsort_i64(text.len ? text.ptr : &zero, sa.size() ? sa.data() : &dummy, (int64_t)text.len);
  if (ret != 0) throw Error(ret, dc3hip_last_error());
  return sacabase::SuffixArray<int64_t>(text, std::move(sa));
}

}  // namespace dc3hip
