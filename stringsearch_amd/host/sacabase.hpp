// sacabase.hpp — C++ mirror of crates/sacabase/src/lib.rs (base types every SACA plug-in returns +
// suffix-array search + verifier).  Same names, argument meaning and error behaviour; Rust panics
// become C++ exceptions.  Header-only, no device code.
#pragma once
#include <cstdlib>
#include <new>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace sacabase {

struct Bytes {                       // &[u8]
  const uint8_t *ptr = nullptr; size_t len = 0;
  Bytes() {}
  Bytes(const uint8_t *p, size_t n) : ptr(p), len(n) {}
  Bytes(const std::string &s) : ptr(reinterpret_cast<const uint8_t *>(s.data())), len(s.size()) {}
  Bytes slice_from(size_t a) const { return Bytes(ptr + a, len - a); }
  Bytes slice(size_t a, size_t b) const { return Bytes(ptr + a, b - a); }
  // Rust slice ordering: lexicographic, a proper prefix is smaller
  int cmp(const Bytes &o) const {
    const size_t l = std::min(len, o.len);
    const int c = l ? std::memcmp(ptr, o.ptr, l) : 0;
    if (c) return c;
    return len < o.len ? -1 : (len > o.len ? 1 : 0);
  }
  bool operator==(const Bytes &o) const { return len == o.len && (len == 0 || std::memcmp(ptr, o.ptr, len) == 0); }
};

// lib.rs:4-21
struct LongestCommonSubstring {
  Bytes text; size_t start = 0, len = 0;
  Bytes as_bytes() const { return text.slice(start, start + len); }
};

// lib.rs:26-35
inline size_t common_prefix_len(Bytes a, Bytes b) {
  const size_t n = std::min(a.len, b.len);
  for (size_t i = 0; i < n; i++) if (a.ptr[i] != b.ptr[i]) return i;
  return n;
}

// lib.rs:39-99
template <class Index>
LongestCommonSubstring longest_substring_match(Bytes text, const Index *sa, size_t sa_len, Bytes needle) {
  auto suff = [&](size_t x) { return text.slice_from((size_t)sa[x]); };
  auto clen = [&](size_t x) { return common_prefix_len(suff(x), needle); };
  for (;;) {
    if (sa_len == 1) return LongestCommonSubstring{text, (size_t)sa[0], clen(0)};
    if (sa_len == 2) {
      const size_t x = clen(0), y = clen(1);
      return x > y ? LongestCommonSubstring{text, (size_t)sa[0], x} : LongestCommonSubstring{text, (size_t)sa[1], y};
    }
    if (sa_len == 0) throw std::out_of_range("index out of bounds: the len is 0 but the index is 0");
    const size_t mid = sa_len / 2;
    if (needle.cmp(suff(mid)) > 0) { sa += mid; sa_len -= mid; } else { sa_len = mid + 1; }
  }
}

// lib.rs:102-123
struct NotSorted : std::runtime_error {
  size_t i, j;
  NotSorted(size_t i_, size_t j_)
      : std::runtime_error("invariant doesn't hold: suf(SA(" + std::to_string(i_) + ")) < suf(SA(" + std::to_string(j_) + "))"),
        i(i_), j(j_) {}
};

// lib.rs:127-149 — throws NotSorted{i, i+1} at the first adjacent pair out of order
template <class Index>
void verify(Bytes input, const Index *sa) {
  for (size_t i = 0; i + 1 < input.len; i++) {
    if (!(input.slice_from((size_t)sa[i]).cmp(input.slice_from((size_t)sa[i + 1])) < 0)) throw NotSorted(i, i + 1);
  }
}

// lib.rs:160-163
struct StringIndex {
  virtual LongestCommonSubstring longest_substring_match(Bytes needle) const = 0;
  virtual ~StringIndex() {}
};

// `vec![0; n]` (cdivsufsort/src/lib.rs:27) is alloc_zeroed, i.e. calloc: zero pages the OS hands out when they are first
// written.  std::vector<T>(n, 0) writes every page first — a second for the 4 GiB array of a 1 GiB text, inside the one
// un-warmed call divsuftest times (main.rs:145-151) and three times the whole GPU call.  ZVec<T>(n) is the Rust
// semantics: calloc'ed storage whose value-initialisation is a no-op.
template <class T>
struct ZeroedAlloc {
  typedef T value_type;
  ZeroedAlloc() = default;
  template <class U> ZeroedAlloc(const ZeroedAlloc<U> &) {}
  T *allocate(size_t n) { void *p = std::calloc(n ? n : 1, sizeof(T)); if (!p) throw std::bad_alloc(); return static_cast<T *>(p); }
  void deallocate(T *p, size_t) { std::free(p); }
  template <class U> void construct(U *) noexcept {}                        // value-initialisation: the storage is zero already
  template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...); }
  template <class U> bool operator==(const ZeroedAlloc<U> &) const { return true; }
  template <class U> bool operator!=(const ZeroedAlloc<U> &) const { return false; }
};
template <class T> using ZVec = std::vector<T, ZeroedAlloc<T>>;

// lib.rs:152-197 — owns `sa`, borrows `text`
template <class Index>
class SuffixArray : public StringIndex {
  ZVec<Index> sa_;
  Bytes text_;

 public:
  SuffixArray(Bytes text, ZVec<Index> sa) : sa_(std::move(sa)), text_(text) {}                  // new, :170
  SuffixArray(Bytes text, const std::vector<Index> &sa) : sa_(sa.begin(), sa.end()), text_(text) {}
  std::pair<Bytes, ZVec<Index>> into_parts() && { return {text_, std::move(sa_)}; }             // :175
  void verify() const { sacabase::verify(text_, sa_.data()); }                                  // :180
  Bytes text() const { return text_; }                                                          // :185
  const ZVec<Index> &sa() const { return sa_; }
  LongestCommonSubstring longest_substring_match(Bytes needle) const override {                 // :190-196
    return sacabase::longest_substring_match(text_, sa_.data(), sa_.size(), needle);
  }
};

}  // namespace sacabase
