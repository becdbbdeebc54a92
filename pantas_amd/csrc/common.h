// Shared host-side helpers for libpantas_gpu: errors, mmap, ASCII tokenising with the
// whitespace rules of Python's str.strip()/str.split() (the reference tokenises every
// GFA/GAF line that way: scripts/alignments_augmentation_from_gaf.py:125,142,381,420 and
// scripts/call.py:146-172), and a minimal parallel_for.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <string>
#include <string_view>
#include <stdexcept>
#include <thread>
#include <vector>
#include <exception>
#include <functional>
#include <algorithm>

namespace pantas {

// Error raised where the reference would die with a Python exception (kind = its class
// name) or where this library refuses an input.  Converted to a status code + message at
// the C-ABI; never crosses it.
struct Error : std::runtime_error {
  std::string kind;
  Error(std::string k, const std::string& msg) : std::runtime_error(k + ": " + msg), kind(std::move(k)) {}
};

constexpr uint32_t NONE32 = 0xFFFFFFFFu;

// Python's ASCII whitespace for str.split()/strip(): \t \n \v \f \r, 0x1c-0x1f, space.
inline bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

struct MappedFile {
  const char* data = nullptr;
  size_t size = 0;
  int fd = -1;
  bool borrowed = false;
  MappedFile() = default;
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
  ~MappedFile();
  void open(const char* path);   // throws Error("FileNotFoundError"/"OSError")
  void attach(const char* p, size_t n) { close(); data = p; size = n; borrowed = true; }   // caller keeps p alive
  void close();
};

// A half-open byte range of a mapped text file.
struct Span {
  const char* b;
  const char* e;
  size_t size() const { return size_t(e - b); }
  std::string_view sv() const { return std::string_view(b, size_t(e - b)); }
};

inline Span strip(Span s) {
  while (s.b < s.e && is_ws((unsigned char)*s.b)) ++s.b;
  while (s.e > s.b && is_ws((unsigned char)s.e[-1])) --s.e;
  return s;
}

// Python reads text files with universal newlines: a '\r' that is not followed by '\n' ends a line too.  This engine splits
// at '\n' only ('\r' is whitespace to it), so a text with such a byte would silently be read differently: every reader
// stops at it (an Error whose text holds LONE_CR_MARK), and the callers that own the text read a COPY with every such byte turned
// into '\n' instead (universal_newlines_copy: same size, same offsets, the reference's lines) -- a slow path nothing a mapper
// or vg writes ever takes.  [b, e) = one line without its '\n' (a '\r' as its last byte is the CR of a CRLF).  Returns the
// offending byte or nullptr.
inline const char* lone_cr(const char* b, const char* e) {
  const char* p = b;
  while (p < e) {
    const char* q = static_cast<const char*>(memchr(p, '\r', size_t(e - p)));
    if (!q) return nullptr;
    if (q + 1 < e && q[1] != '\n') return q;
    p = q + 1;
  }
  return nullptr;
}

constexpr const char* LONE_CR_MARK = "lone carriage return";
bool is_lone_cr_error(const std::exception& e);
bool has_lone_cr(const char* p, size_t n);

// Iterates whitespace-separated tokens of a (stripped or not) line.
struct Tokens {
  const char* p;
  const char* e;
  explicit Tokens(Span s) : p(s.b), e(s.e) {}
  bool next(Span& t) {
    while (p < e && is_ws((unsigned char)*p)) ++p;
    if (p >= e) return false;
    t.b = p;
    while (p < e && !is_ws((unsigned char)*p)) ++p;
    t.e = p;
    return true;
  }
};

// Canonical non-negative decimal (no sign, no leading zeros except "0", <= 18 digits).
inline bool parse_canonical_u64(Span t, uint64_t& v) {
  size_t n = t.size();
  if (n == 0 || n > 18) return false;
  if (n > 1 && t.b[0] == '0') return false;
  uint64_t x = 0;
  for (const char* c = t.b; c < t.e; ++c) {
    if (*c < '0' || *c > '9') return false;
    x = x * 10 + uint64_t(*c - '0');
  }
  v = x;
  return true;
}

// Python int(str) for the ASCII subset: optional sign, digits with single '_' separators.  Values beyond 2^60 are
// refused, or with `saturate` become +-2^61 (for fields that are only compared or range-checked afterwards).
// ---- Python's str.strip() / float(str) for the native forms of quantify.py and remap.py (shared: ADVICE r3) -------------------
inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
inline Span py_strip(Span s) {
  while (s.b < s.e && py_space(static_cast<unsigned char>(*s.b))) ++s.b;
  while (s.e > s.b && py_space(static_cast<unsigned char>(s.e[-1]))) --s.e;
  return s;
}
// float(s): decimal literals with '_' between digits, inf / infinity / nan in any case, surrounding whitespace; nothing else
// (strtod alone would take hex floats and "nan(...)" and refuse "1_0").  false = Python raises ValueError.
inline bool parse_py_float(Span s0, double& out) {
  const Span s = py_strip(s0);
  if (s.b == s.e) return false;
  const char* p = s.b;
  bool neg = false;
  if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
  auto word = [&](const char* w) {
    const size_t n = strlen(w);
    if (size_t(s.e - p) != n) return false;
    for (size_t i = 0; i < n; ++i) if (tolower(static_cast<unsigned char>(p[i])) != w[i]) return false;
    return true;
  };
  if (word("inf") || word("infinity")) { out = neg ? -INFINITY : INFINITY; return true; }
  if (word("nan")) { out = neg ? -NAN : NAN; return true; }
  // [digits][.digits][(e|E)[sign]digits], at least one digit in the mantissa; '_' only between two digits
  std::string t;
  if (neg) t += '-';
  bool any = false, prev_digit = false;
  auto digits = [&]() {
    bool got = false;
    while (p < s.e) {
      if (*p >= '0' && *p <= '9') { t += *p; got = true; prev_digit = true; ++p; }
      else if (*p == '_' && prev_digit && p + 1 < s.e && p[1] >= '0' && p[1] <= '9') { prev_digit = false; ++p; }
      else break;
    }
    prev_digit = false;
    return got;
  };
  any |= digits();
  if (p < s.e && *p == '.') { t += '.'; ++p; any |= digits(); }
  if (!any) return false;
  if (p < s.e && (*p == 'e' || *p == 'E')) {
    t += 'e'; ++p;
    if (p < s.e && (*p == '+' || *p == '-')) { t += *p; ++p; }
    if (!digits()) return false;
  }
  if (p != s.e) return false;
  out = strtod(t.c_str(), nullptr);
  return true;
}

inline bool parse_py_int(Span t, int64_t& v, bool saturate = false) {
  const char* c = t.b;
  if (c >= t.e) return false;
  bool neg = false;
  if (*c == '+' || *c == '-') { neg = (*c == '-'); ++c; }
  if (c >= t.e || *c < '0' || *c > '9') return false;
  uint64_t x = 0;
  bool prev_us = false, big = false;
  for (; c < t.e; ++c) {
    if (*c == '_') { if (prev_us) return false; prev_us = true; continue; }
    if (*c < '0' || *c > '9') return false;
    prev_us = false;
    if (x > (uint64_t(1) << 60)) { if (!saturate) return false; big = true; }
    if (!big) x = x * 10 + uint64_t(*c - '0');
  }
  if (prev_us) return false;
  if (big) x = uint64_t(1) << 61;
  v = neg ? -int64_t(x) : int64_t(x);
  return true;
}

unsigned hw_threads();

// std::vector whose resize() leaves trivially constructible elements uninitialised: for buffers of gigabytes that are
// about to be overwritten (a device-to-host copy, a parallel fill), where the value-initialising fill would be a serial pass
// over fresh pages.  Blocks of 8 MB and more are 2 MB-aligned and marked MADV_HUGEPAGE: the first touch of 4 GB of fresh
// 4 KB pages costs 0.3 s on 64 threads, of transparent huge pages 0.014 s (profiles/r2_hostio_probe.txt) -- the graph
// arrays of a whole-genome GFA are 14 GB, the loader's temporaries another 20.
void* big_alloc(size_t bytes);
void big_free(void* p, size_t bytes);
template <class T>
struct NoInitAlloc {
  using value_type = T;
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(big_alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t n) { big_free(p, n * sizeof(T)); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
    else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
  template <class U> bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T> using RawVector = std::vector<T, NoInitAlloc<T>>;
// out = [p, p + n) with every '\r' that no '\n' follows replaced by '\n' (universal newlines, see lone_cr above); many threads
void universal_newlines_copy(const char* p, size_t n, RawVector<char>& out);
// threads of the two output writers (PANTAS_WRITER_THREADS, default hw_threads()): they format pieces and queue for the output
// file's lock, so the best count is not the parsers'
unsigned writer_threads();

// Giving tens of GB of temporaries back to the kernel takes seconds (50 - 85 ms per GB on the GPU box, huge pages or not) and holds
// the address-space lock in long stretches: a free on a thread of its own right after the parse made the NEXT call that maps memory
// -- the first hipMalloc / stream / pinned buffer -- wait for it (0.9 s at whole-genome scale).  So the big frees are parked
// (defer_free) and started (release_deferred: one detached thread) when the run turns to writing its output file, which waits on
// the file system, not on memory.  With less than four times `bytes` of memory available the free happens at once.
void defer_free(size_t bytes, std::function<void()> fn);
void release_deferred();

// touches every page of [p, p + bytes) from several threads (first-touch page faults are the cost of a fresh buffer)
void prefault(void* p, size_t bytes);

// Runs fn(chunk_index, begin, end) over [0, n) split into `chunks` contiguous ranges.
void parallel_chunks(size_t n, unsigned chunks, const std::function<void(unsigned, size_t, size_t)>& fn);

// std::sort by several threads: chunks sorted side by side, then merged pairwise (rounds of std::inplace_merge, each round's merges
// side by side); the same result as std::sort for a strict weak order whose ties are equal elements
template <class It, class Cmp>
void par_sort(It b, It e, Cmp cmp, unsigned threads = 0) {
  const size_t n = size_t(e - b);
  if (threads == 0) threads = hw_threads();
  unsigned c = 1;
  while (c * 2 <= threads && c < 16 && n / (c * 2) >= (size_t(1) << 16)) c *= 2;
  if (c == 1) { std::sort(b, e, cmp); return; }
  std::vector<size_t> cut(c + 1);
  for (unsigned i = 0; i <= c; ++i) cut[i] = n * i / c;
  parallel_chunks(c, c, [&](unsigned, size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) std::sort(b + long(cut[i]), b + long(cut[i + 1]), cmp); });
  for (unsigned w = 1; w < c; w *= 2) {
    const unsigned pairs = c / (2 * w);
    parallel_chunks(pairs, pairs, [&](unsigned, size_t lo, size_t hi) {
      for (size_t p = lo; p < hi; ++p) {
        const size_t a = cut[p * 2 * w], m = cut[p * 2 * w + w], z = cut[p * 2 * w + 2 * w];
        std::inplace_merge(b + long(a), b + long(m), b + long(z), cmp);
      }
    });
  }
}

// v = n copies of value, written by many threads (RawVector::assign would be one thread's pass over fresh pages)
template <class T>
void par_assign(RawVector<T>& v, size_t n, T value, unsigned threads = 0) {
  v.resize(n);
  T* p = v.data();
  parallel_chunks(n, n < (size_t(1) << 20) ? 1u : (threads ? threads : hw_threads()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) p[i] = value;
  });
}

// Splits a text buffer into `want` byte ranges that end on '\n' boundaries.
std::vector<std::pair<size_t, size_t>> split_at_newlines(const char* data, size_t size, unsigned want);

// stable parallel LSD radix sort of (key, idx) pairs by key (16-bit digits)
void radix_sort_pairs(RawVector<uint32_t>& key, RawVector<uint32_t>& idx, unsigned threads);
void radix_sort_pairs(std::vector<uint32_t>& key, std::vector<uint32_t>& idx, unsigned threads);

double now_s();

// Output of pieces that are produced in parallel but must land in order (the augmented GFA, the events CSV).  Piece p may
// be put once piece p - 1 has been put (its end is then known); what happens to the bytes depends on the descriptor:
//   a regular file: pwrite at the piece's offset, from the thread that made the piece.  (The kernel takes the inode lock
//     for every write: 64 threads writing 23 GB into tmpfs measured 4.4 GB/s, the speed of ONE thread, and that is what
//     `pantas augment` spends 5 of its 13 s on at whole-genome scale -- profiles/r2_e2e_s4_cli_output_cost.json.  With
//     PANTAS_OUTPUT_MMAP=1 the file is extended (sparse), mapped, and the threads copy into the mapping instead: measured
//     four times slower on the same tmpfs, kept as a switch for filesystems where it is not.);
//   anything else (pipe, terminal): write(), one piece after the other.
// finish() sets the file size and position behind the last piece.  One put per piece index, any thread.
struct OrderedOutput {
  OrderedOutput(int fd, size_t n_pieces, uint64_t expected_bytes, uint64_t bound_bytes);
  ~OrderedOutput();
  OrderedOutput(const OrderedOutput&) = delete;
  OrderedOutput& operator=(const OrderedOutput&) = delete;
  void put(size_t piece, const char* data, size_t n);    // throws Error("OSError") on a failed write
  void finish();
  const char* mode_name() const;
 private:
  struct Impl;
  Impl* im;
};

// phase timing to stderr when PANTAS_TIMING is set
struct PhaseTimer {
  const char* what;
  double t0, last;
  bool on;
  explicit PhaseTimer(const char* w);
  void lap(const char* phase);
};

}  // namespace pantas
