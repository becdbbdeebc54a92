// CPU oracle for the AS-event calling path in plain C++.  TEST INFRASTRUCTURE ONLY.
//
// An independent restatement of the reference script scripts/call.py (main at :132-1105; bracketed numbers below are its
// lines) with the four fixed rules of oracle/call_oracle.py for what the reference leaves to Python's hash seed:
//   R1  rows in (pass, junction in file order, emission order);
//   R2  "any exon of the set" (:484-486) = the EX tag that appears first in the GFA;
//   R3  set.pop() of the intronic search (:743, :796) level by level: the members the set held when pvisitl was last set, in
//       S-line order, before anything added since; "any hit" (:752, :805) = the segment whose S line comes first;
//   R4  '|'-joined set contents (:876, :1072-1077) in first-appearance order.
// Written for speed -- whole-genome graphs, 10^8 rows -- so that `pantas call` can be checked at BASELINE.json sizes and
// bench.py's `cpu_baseline` leg can time "call on the host cores" (SURVEY.md 8-d rows 2a / 2b).  Only tests/,
// __graft_entry__.smoke() and that leg of bench.py may build or run it; nothing in pantas_amd/ does, and it shares no
// code with the product (names are strings looked up in a hash table, sets are small vectors of string views, the
// junctions sharing a head or tail are found through the by-head / by-tail rows of the link lists instead of the
// reference's linear scans :280-282, :945-947 -- same sets, not the same cost).
//
// Parity status: PINNED -- tests/test_oracle_cpp.py holds it to the outputs of the real reference under tests/golden/
// (18 flag sets on six cases as row multisets with the reference's own seed-dependent alternatives, the call crash
// inputs by exception class, the edge inputs) and to oracle/call_oracle.py on generated workloads, with one thread and
// with several.
//
//   call_oracle_cpp [--threads T] [--rca N] [--novel] [--no-annotated] [--events A B ..] [--isw N] [--minintronsize N]
//                   [--junction a-b] [--no-out] [--digest] GFA GTF > events.csv
//   call_oracle_cpp --digest-csv FILE      (row-multiset digest of any events CSV, header excluded)
//
// --digest: the last stderr line is a JSON object with the number of rows and an order-independent digest of the
// NORMALISED rows (on `IR,novel` rows the '|' parts of columns 6 and 7 sorted) -- two CSVs are equal as
// row multisets iff their digests are (up to a 2^-128 collision).  Rows are produced junction block by junction block
// on T threads and written in block order, so the output does not depend on T.
// Deviations from Python's text handling, none of which the goldens or the synthetic inputs contain: integers are
// [+-]?[0-9]+, lengths count bytes, whitespace is ASCII.  Universal newlines are applied when a file is mapped.
// Where the reference's answer depends on the hash seed beyond R1-R4 (the `else` branch of :681-685 reads a loop
// variable left over from an earlier junction of a set iteration) the oracle stops with `Unsupported`.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <string_view>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace {

using sv = std::string_view;
using u32 = uint32_t;
using u64 = uint64_t;
using i64 = int64_t;

struct Crash {
  std::string kind, msg;
};
[[noreturn]] void crash(const char* kind, const std::string& msg) { throw Crash{kind, msg}; }

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
void parallel(int T, F f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back([&f, t] { f(t); });
  for (auto& x : th) x.join();
}

// ---------------------------------------------------------------- text
struct Mapped {
  char* p = nullptr;
  size_t n = 0;
  void open(const char* path, int T) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { perror(path); exit(2); }
    struct stat st;
    fstat(fd, &st);
    n = size_t(st.st_size);
    if (n) {
      // private + writable: a '\r' that no '\n' follows ends its line (Python's universal newlines), fixed in place
      p = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0));
      if (p == MAP_FAILED) { perror("mmap"); exit(2); }
      parallel(T, [&](int t) {
        char* b = p + n * size_t(t) / size_t(T);
        char* e = p + n * size_t(t + 1) / size_t(T);
        for (char* q = b; q < e && (q = static_cast<char*>(memchr(q, '\r', size_t(e - q)))) != nullptr; ++q)
          if (q + 1 < p + n && q[1] != '\n') *q = '\n';
      });
    }
    close(fd);
  }
};

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r') || (c >= 0x1c && c <= 0x1f); }
inline sv strip(sv s) {
  while (!s.empty() && is_space(s.front())) s.remove_prefix(1);
  while (!s.empty() && is_space(s.back())) s.remove_suffix(1);
  return s;
}
// str.split(): the tokens of s, at most `cap` of them stored, all of them counted
inline size_t split_ws(sv s, sv* out, size_t cap) {
  size_t k = 0, i = 0;
  while (i < s.size()) {
    while (i < s.size() && is_space(s[i])) ++i;
    if (i == s.size()) break;
    size_t j = i;
    while (j < s.size() && !is_space(s[j])) ++j;
    if (k < cap) out[k] = s.substr(i, j - i);
    ++k;
    i = j;
  }
  return k;
}
inline bool parse_int(sv s, i64& v) {
  size_t i = 0;
  bool neg = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
  if (i == s.size()) return false;
  i64 x = 0;
  // int() takes single '_' between digits ("1_0" == 10), not in front, at the end or doubled
  if (s[i] == '_' || s.back() == '_') return false;
  for (; i < s.size(); ++i) {
    if (s[i] == '_') { if (s[i - 1] == '_') return false; continue; }
    if (s[i] < '0' || s[i] > '9') return false;
    if (x > (INT64_MAX - 9) / 10) crash("Unsupported", "integer beyond 64 bits");
    x = x * 10 + (s[i] - '0');
  }
  v = neg ? -x : x;
  return true;
}
inline sv before_last(sv s, char c) {        // c.join(s.split(c)[:-1])
  const size_t p = s.rfind(c);
  return p == sv::npos ? sv() : s.substr(0, p);
}
inline sv tx_of_exon_tag(sv e) { return before_last(before_last(e, '.'), '_'); }                       // [:107-121]
inline sv tx_of_junction_tag(sv t) { return before_last(before_last(before_last(t, '.'), '.'), '_'); }  // [:98-104]
inline sv tx_for_filter(sv t) { return before_last(t, '_'); }                                          // [:356, :364, :595, :603]
inline sv after_last(sv s, char c) {         // s.split(c)[-1]
  const size_t p = s.rfind(c);
  return p == sv::npos ? s : s.substr(p + 1);
}

// small sets of strings: unique, in insertion order
using Set = std::vector<sv>;
inline bool has(const Set& s, sv x) {
  for (sv y : s) if (y == x) return true;
  return false;
}
inline void add(Set& s, sv x) { if (!has(s, x)) s.push_back(x); }
inline void split_list(sv value, std::vector<sv>& out) {     // value.split(",")
  out.clear();
  size_t i = 0;
  for (;;) {
    const size_t j = value.find(',', i);
    if (j == sv::npos) { out.push_back(value.substr(i)); return; }
    out.push_back(value.substr(i, j - i));
    i = j + 1;
  }
}
inline Set set_of_list(sv value) {
  std::vector<sv> l;
  split_list(value, l);
  Set s;
  for (sv x : l) add(s, x);
  return s;
}
inline Set inter(const Set& a, const Set& b) {
  Set r;
  for (sv x : a) if (has(b, x)) r.push_back(x);
  return r;
}
inline bool any_outside(const Set& a, const Set& b) {        // len(a - b) > 0
  for (sv x : a) if (!has(b, x)) return true;
  return false;
}

u64 hash_bytes(const char* p, size_t n, u64 seed) {
  u64 h = seed ^ (n * 0x9E3779B97F4A7C15ull);
  while (n >= 8) {
    u64 w;
    memcpy(&w, p, 8);
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    h ^= h >> 32;
    p += 8;
    n -= 8;
  }
  u64 w = 0;
  memcpy(&w, p, n);
  h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 29;
  h *= 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  return h;
}

// ---------------------------------------------------------------- the graph as call.py holds it
constexpr u32 NONE = 0xffffffffu;

struct Seg {                  // gfaS[name] as the LAST S line of that name left it [:147-154]
  const char* name;
  u32 name_len;
  u32 ln;                     // len(seq) [:150]
  const char* ex;             // value of the EX tag, or nullptr
  u32 ex_len;
  bool has_nc;
  i64 nc;
  u64 off;                    // where this S line starts (adjacency lists restart here [:153-154])
  u64 first_off;              // where the first S line of the name starts (a link in front of it: KeyError [:174])
  sv nm() const { return sv(name, name_len); }
};
struct RawLink {              // an L line before its names are looked up
  sv u, v, jn;
  bool has_jn, has_id, has_rc;
  i64 rc;
  u64 off;
};
struct Link {
  u32 u, v;
  const char* jn;
  u32 jn_len;
  bool has_jn, has_id, has_rc;    // of THIS line
  bool grp_jn, grp_id;            // (u, v) in junctions / noveljunctions: some line of the pair carries the tag [:178-181]
  bool first_jn, first_id;        // the line that put the pair into the set (R1: the set's order)
  u32 canon;                      // the last line of the pair: its attributes are gfaL[(u, v)] [:173]
  i64 rc;
  u64 off;
};

struct Err {
  u64 off;
  std::string kind, msg;
};

// an array of graph size: not value-initialised (the threads that fill it take its page faults)
template <class X>
struct Buf {
  X* p = nullptr;
  size_t n = 0;
  void alloc(size_t k) {
    free(p);
    n = k;
    p = static_cast<X*>(malloc(std::max<size_t>(k, 1) * sizeof(X)));
    if (!p) { perror("malloc"); exit(2); }
  }
  X& operator[](size_t i) { return p[i]; }
  const X& operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
};

struct Graph {
  Buf<Seg> seg;
  Buf<Link> link;
  std::vector<std::atomic<u32>> table;     // open addressing: name -> index of its last S line + 1
  u64 mask = 0;
  std::vector<u64> out_ptr, in_ptr;        // rows of `link` by u / by v, file order
  Buf<u32> out_idx, in_idx;
  std::vector<u32> junctions, novel;       // links with first_jn / first_id, file order
  std::unordered_map<sv, u64> exon_rank;   // EX tag -> (S line offset, position in its list) of its first appearance

  u32 find(sv name) const {
    if (table.empty()) return NONE;
    for (u64 h = hash_bytes(name.data(), name.size(), 1) & mask;; h = (h + 1) & mask) {
      const u32 s = table[h].load(std::memory_order_relaxed);
      if (!s) return NONE;
      if (seg[s - 1].nm() == name) return s - 1;
    }
  }
  // gfaL[(a, b)]: the record whose attributes count, or NONE
  u32 find_link(u32 a, u32 b) const {
    for (u64 i = out_ptr[a]; i < out_ptr[a + 1]; ++i)
      if (link[out_idx[i]].v == b) return link[out_idx[i]].canon;
    return NONE;
  }
  template <class F>
  void outs(u32 n, F f) const {            // gfaS[n]["O"] [:174]
    for (u64 i = out_ptr[n]; i < out_ptr[n + 1]; ++i) {
      const Link& l = link[out_idx[i]];
      if (l.off > seg[n].off) f(l.v);
    }
  }
  template <class F>
  void ins(u32 n, F f) const {             // gfaS[n]["I"] [:175]
    for (u64 i = in_ptr[n]; i < in_ptr[n + 1]; ++i) {
      const Link& l = link[in_idx[i]];
      if (l.off > seg[n].off) f(l.u);
    }
  }
  std::vector<u32> out_list(u32 n) const { std::vector<u32> r; outs(n, [&](u32 x) { r.push_back(x); }); return r; }
  std::vector<u32> in_list(u32 n) const { std::vector<u32> r; ins(n, [&](u32 x) { r.push_back(x); }); return r; }
  bool has_out(u32 n) const { bool r = false; outs(n, [&](u32) { r = true; }); return r; }
};

// one IL / OL value through build_attrs [:33-66]: nothing of it is read later, but it can raise
void check_positions(sv value, u64 off) {
  std::vector<sv> ent, part;
  std::vector<std::vector<i64>> v;
  split_list(value, ent);
  for (sv e : ent) {
    v.emplace_back();
    size_t i = 0;
    for (;;) {
      const size_t j = e.find('.', i);
      const sv p = j == sv::npos ? e.substr(i) : e.substr(i, j - i);
      i64 x;
      if (!parse_int(p, x)) crash("ValueError", "IL/OL entry at byte " + std::to_string(off));
      v.back().push_back(x);
      if (j == sv::npos) break;
      i = j + 1;
    }
  }
  auto collapse = [&](const std::vector<const std::vector<i64>*>& lc) {
    i64 count = 0;
    for (auto* x : lc) {
      if (x->size() < 2) crash("IndexError", "IL/OL entry without a count at byte " + std::to_string(off));
      count += (*x)[1];
    }
    if (count == 0) crash("ZeroDivisionError", "IL/OL counts sum to zero at byte " + std::to_string(off));
  };
  if (v.size() >= 2) {
    size_t mn = 0, mx = 0;
    for (size_t i = 1; i < v.size(); ++i) {
      if (v[i][0] < v[mn][0]) mn = i;
      if (v[i][0] > v[mx][0]) mx = i;
    }
    std::vector<const std::vector<i64>*> k1, k2;
    const i64 spread = v[mn][0] > v[mx][0] ? v[mn][0] - v[mx][0] : v[mx][0] - v[mn][0];
    if (spread < 3) {
      for (auto& x : v) k1.push_back(&x);
      collapse(k1);
    } else {
      k1.push_back(&v[mn]);
      k2.push_back(&v[mx]);
      for (auto& x : v) {
        if (x == v[mn] || x == v[mx]) continue;
        const i64 dmin = std::llabs(x[0] - v[mn][0]), dmax = std::llabs(x[0] - v[mx][0]);
        (dmin < dmax ? k1 : k2).push_back(&x);
      }
      collapse(k1);
      collapse(k2);
    }
  } else if (v[0].size() < 2) {
    crash("IndexError", "IL/OL entry without a count at byte " + std::to_string(off));     // max(_v, key=x[1]) [:66]
  }
}

struct Attrs {
  sv ex, jn;
  bool has_ex = false, has_jn = false, has_id = false, has_rc = false, has_nc = false;
  i64 rc = 0, nc = 0;
};
// build_attrs [:25-69] over the tokens behind the fixed columns
void build_attrs(const sv* f, size_t n, u64 off, Attrs& a) {
  for (size_t i = 0; i < n; ++i) {
    const sv t = f[i];
    const size_t c1 = t.find(':');
    const size_t c2 = c1 == sv::npos ? sv::npos : t.find(':', c1 + 1);
    if (c2 == sv::npos || t.find(':', c2 + 1) != sv::npos)
      crash("ValueError", "tag '" + std::string(t) + "' does not split in 3 on ':' (byte " + std::to_string(off) + ")");
    const sv name = t.substr(0, c1), value = t.substr(c2 + 1);
    if (name == "LN" || name == "RC" || name == "NC") {
      i64 x;
      if (!parse_int(value, x)) crash("ValueError", "invalid literal for int(): '" + std::string(value) + "'");
      if (name == "RC") { a.has_rc = true; a.rc = x; }
      if (name == "NC") { a.has_nc = true; a.nc = x; }
    } else if (name == "IL" || name == "OL") {
      check_positions(value, off);
    } else if (name == "EX") { a.has_ex = true; a.ex = value; }
    else if (name == "JN") { a.has_jn = true; a.jn = value; }
    else if (name == "ID") { a.has_id = true; }
  }
}

void load_gfa(const Mapped& m, int T, Graph& g) {
  const bool timing = getenv("CALL_ORACLE_TIMING") != nullptr;
  double tp = now();
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[load_gfa] %-12s %.3f s\n", what, now() - tp);
    tp = now();
  };
  // ---- cut at line starts
  std::vector<size_t> cut(size_t(T) + 1, m.n);
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    size_t c = m.n * size_t(t) / size_t(T);
    if (c < cut[size_t(t) - 1]) c = cut[size_t(t) - 1];
    if (c > 0 && c < m.n) {
      const void* q = memchr(m.p + c - 1, '\n', m.n - c + 1);
      c = q ? size_t(static_cast<const char*>(q) - m.p) + 1 : m.n;
    }
    cut[size_t(t)] = c;
  }
  std::vector<std::vector<Seg>> segs(static_cast<size_t>(T));
  std::vector<std::vector<RawLink>> raws(static_cast<size_t>(T));
  std::vector<std::unordered_map<sv, u64>> ranks(static_cast<size_t>(T));
  std::vector<Err> errs;
  std::mutex mu;
  parallel(T, [&](int t) {
    auto& S = segs[size_t(t)];
    auto& L = raws[size_t(t)];
    auto& R = ranks[size_t(t)];
    std::vector<sv> tok(64), exl;
    size_t pos = cut[size_t(t)];
    const size_t end = cut[size_t(t) + 1];
    u64 off = pos;
    try {
      while (pos < end) {
        const void* q = memchr(m.p + pos, '\n', m.n - pos);
        const size_t stop = q ? size_t(static_cast<const char*>(q) - m.p) : m.n;
        const sv line = strip(sv(m.p + pos, stop - pos));
        off = pos;
        pos = stop + 1;
        if (line.empty()) continue;
        if (line[0] == 'S') {                                                  // [:147-154]
          size_t k = split_ws(line, tok.data(), tok.size());
          if (k > tok.size()) { tok.resize(k); k = split_ws(line, tok.data(), tok.size()); }
          if (k < 3) crash("ValueError", "not enough values to unpack (S line at byte " + std::to_string(off) + ")");
          Attrs a;
          build_attrs(tok.data() + 3, k - 3, off, a);
          Seg s;
          s.name = tok[1].data();
          s.name_len = u32(tok[1].size());
          if (tok[2].size() > 0xffffffffull) crash("Unsupported", "segment of 4 GB");
          s.ln = u32(tok[2].size());
          s.ex = a.has_ex ? a.ex.data() : nullptr;
          s.ex_len = u32(a.ex.size());
          s.has_nc = a.has_nc;
          s.nc = a.nc;
          s.off = s.first_off = off;
          S.push_back(s);
          if (a.has_ex) {
            split_list(a.ex, exl);
            for (size_t i = 0; i < exl.size(); ++i) {
              const u64 r = (off << 20) | std::min<u64>(i, 0xfffff);
              auto it = R.find(exl[i]);
              if (it == R.end()) R.emplace(exl[i], r);
            }
          }
        } else if (line[0] == 'P') {                                           // [:155-162]
          if (split_ws(line, tok.data(), 0) != 4) crash("ValueError", "P line needs 4 fields (byte " + std::to_string(off) + ")");
        } else if (line[0] == 'L') {                                           // [:163-181]
          size_t k = split_ws(line, tok.data(), tok.size());
          if (k > tok.size()) { tok.resize(k); k = split_ws(line, tok.data(), tok.size()); }
          if (k < 6) crash("ValueError", "not enough values to unpack (L line at byte " + std::to_string(off) + ")");
          Attrs a;
          build_attrs(tok.data() + 6, k - 6, off, a);
          L.push_back(RawLink{tok[1], tok[3], a.jn, a.has_jn, a.has_id, a.has_rc, a.rc, off});
        }
      }
    } catch (Crash& c) {
      std::lock_guard<std::mutex> lk(mu);
      errs.push_back(Err{off, c.kind, c.msg});
    }
  });
  lap("scan");
  // ---- one array of S lines in file order, the name table
  std::vector<size_t> sbase(size_t(T) + 1, 0), lbase(size_t(T) + 1, 0);
  for (int t = 0; t < T; ++t) {
    sbase[size_t(t) + 1] = sbase[size_t(t)] + segs[size_t(t)].size();
    lbase[size_t(t) + 1] = lbase[size_t(t)] + raws[size_t(t)].size();
  }
  const size_t NS = sbase[size_t(T)], NL = lbase[size_t(T)];
  if (NS >= 0xfffffff0ull || NL >= 0xfffffff0ull) crash("Unsupported", "more than 2^32 S or L lines");
  g.seg.alloc(NS);
  parallel(T, [&](int t) {
    std::copy(segs[size_t(t)].begin(), segs[size_t(t)].end(), g.seg.p + sbase[size_t(t)]);
    std::vector<Seg>().swap(segs[size_t(t)]);
  });
  lap("gather S");
  u64 cap = 16;
  while (cap < 2 * NS + 2) cap <<= 1;
  g.table = std::vector<std::atomic<u32>>(cap);
  g.mask = cap - 1;
  std::vector<std::pair<u32, u32>> dups;
  parallel(T, [&](int t) {
    const size_t b = NS * size_t(t) / size_t(T), e = NS * size_t(t + 1) / size_t(T);
    for (size_t i = b; i < e; ++i) {
      const sv name = g.seg[i].nm();
      for (u64 h = hash_bytes(name.data(), name.size(), 1) & g.mask;; h = (h + 1) & g.mask) {
        u32 s = g.table[h].load(std::memory_order_acquire);
        if (!s) {
          if (g.table[h].compare_exchange_strong(s, u32(i) + 1, std::memory_order_acq_rel)) break;
        }
        if (g.seg[s - 1].nm() == name) {
          // a second S line of the name: the later line is the one that stays [:149]
          { std::lock_guard<std::mutex> lk(mu); dups.emplace_back(s - 1, u32(i)); }
          while (s < u32(i) + 1 && !g.table[h].compare_exchange_weak(s, u32(i) + 1, std::memory_order_acq_rel)) {}
          break;
        }
      }
    }
  });
  for (auto& d : dups) {
    const u32 w = g.find(g.seg[d.first].nm());
    g.seg[w].first_off = std::min({g.seg[w].first_off, g.seg[d.first].off, g.seg[d.second].off});
  }
  lap("name table");
  // ---- EX tags: first appearance
  for (int t = 0; t < T; ++t) {
    for (auto& kv : ranks[size_t(t)]) {
      auto it = g.exon_rank.find(kv.first);
      if (it == g.exon_rank.end()) g.exon_rank.emplace(kv.first, kv.second);
      else if (kv.second < it->second) it->second = kv.second;
    }
    std::unordered_map<sv, u64>().swap(ranks[size_t(t)]);
  }
  lap("EX ranks");
  // ---- links: names -> segments
  g.link.alloc(NL);
  parallel(T, [&](int t) {
    const auto& L = raws[size_t(t)];
    for (size_t i = 0; i < L.size(); ++i) {
      const RawLink& r = L[i];
      Link& l = g.link[lbase[size_t(t)] + i];
      l.u = g.find(r.u);
      l.v = l.u == NONE ? NONE : g.find(r.v);
      if (l.u == NONE || l.v == NONE || g.seg[l.u].first_off > r.off || g.seg[l.v].first_off > r.off) {
        std::lock_guard<std::mutex> lk(mu);
        errs.push_back(Err{r.off, "KeyError", "link end without an S line in front (byte " + std::to_string(r.off) + ")"});
        l.u = l.v = NONE;
        break;
      }
      l.jn = r.has_jn ? r.jn.data() : nullptr;
      l.jn_len = u32(r.jn.size());
      l.has_jn = r.has_jn;
      l.has_id = r.has_id;
      l.has_rc = r.has_rc;
      l.grp_jn = l.grp_id = l.first_jn = l.first_id = false;
      l.canon = NONE;
      l.rc = r.rc;
      l.off = r.off;
    }
    std::vector<RawLink>().swap(raws[size_t(t)]);
  });
  if (!errs.empty()) {        // the reference dies at the first offending line
    const Err* first = &errs[0];
    for (auto& e : errs) if (e.off < first->off) first = &e;
    crash(first->kind.c_str(), first->msg);
  }
  lap("resolve L");
  // ---- rows by u and by v, file order (a stable counting sort each)
  auto rows = [&](bool by_u, std::vector<u64>& ptr, Buf<u32>& idx) {
    ptr.assign(NS + 1, 0);
    for (size_t i = 0; i < NL; ++i) ++ptr[(by_u ? g.link[i].u : g.link[i].v) + 1];
    for (size_t i = 0; i < NS; ++i) ptr[i + 1] += ptr[i];
    idx.alloc(NL);
    std::vector<u64> cur(ptr.begin(), ptr.end() - 1);
    for (size_t i = 0; i < NL; ++i) idx[cur[by_u ? g.link[i].u : g.link[i].v]++] = u32(i);
  };
  {
    std::thread a([&] { rows(true, g.out_ptr, g.out_idx); });
    rows(false, g.in_ptr, g.in_idx);
    a.join();
  }
  lap("rows");
  // ---- gfaL keeps ONE entry per (u, v): the last line's attributes, set membership from any line
  parallel(T, [&](int t) {
    const size_t b = NS * size_t(t) / size_t(T), e = NS * size_t(t + 1) / size_t(T);
    for (size_t n = b; n < e; ++n) {
      const u64 lo = g.out_ptr[n], hi = g.out_ptr[n + 1];
      for (u64 i = lo; i < hi; ++i) {
        Link& l = g.link[g.out_idx[i]];
        if (l.canon != NONE) continue;
        bool jn = false, id = false;
        u32 last = g.out_idx[i];
        for (u64 k = i; k < hi; ++k) {
          Link& o = g.link[g.out_idx[k]];
          if (o.v != l.v) continue;
          if (o.has_jn && !jn) { jn = true; o.first_jn = true; }
          if (o.has_id && !id) { id = true; o.first_id = true; }
          last = g.out_idx[k];
        }
        for (u64 k = i; k < hi; ++k) {
          Link& o = g.link[g.out_idx[k]];
          if (o.v != l.v) continue;
          o.canon = last;
          o.grp_jn = jn;
          o.grp_id = id;
        }
      }
    }
  });
  lap("pairs");
  for (size_t i = 0; i < NL; ++i) {
    if (g.link[i].first_jn) g.junctions.push_back(u32(i));
    if (g.link[i].first_id) g.novel.push_back(u32(i));
  }
  lap("lists");
}

// ---------------------------------------------------------------- GTF [:188-216]
struct Gene {
  sv name, strand, chrom;
};
struct Annotation {
  std::unordered_map<sv, u32> tx2gene, gene_ix;
  std::vector<Gene> genes;
};
bool id_char(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_'; }
// re.search(key + ' "[A-Za-z0-9_]+";', s): leftmost match, the id between the quotes
bool search_id(sv s, sv key, sv& id) {
  for (size_t p = s.find(key); p != sv::npos; p = s.find(key, p + 1)) {
    size_t i = p + key.size();
    if (i + 1 < s.size() && s[i] == ' ' && s[i + 1] == '"') {
      size_t j = i + 2;
      while (j < s.size() && id_char(s[j])) ++j;
      if (j > i + 2 && j + 1 < s.size() && s[j] == '"' && s[j + 1] == ';') { id = s.substr(i + 2, j - i - 2); return true; }
    }
  }
  return false;
}
void load_gtf(const Mapped& m, Annotation& a) {
  static const sv feats[] = {"mRNA", "transcript", "miRNA", "ncRNA", "pre_miRNA", "snoRNA", "pseudogene"};
  size_t pos = 0, lineno = 0;
  std::vector<sv> col;
  while (pos < m.n) {
    const void* q = memchr(m.p + pos, '\n', m.n - pos);
    const size_t stop = q ? size_t(static_cast<const char*>(q) - m.p) : m.n;
    const sv line(m.p + pos, stop - pos);
    pos = stop + 1;
    ++lineno;
    if (!line.empty() && line[0] == '#') continue;
    col.clear();
    for (size_t i = 0;;) {
      const size_t j = line.find('\t', i);
      if (j == sv::npos) { col.push_back(line.substr(i)); break; }
      col.push_back(line.substr(i, j - i));
      i = j + 1;
    }
    if (col.size() < 3) crash("IndexError", "GTF line " + std::to_string(lineno) + ": list index out of range");
    if (std::find(std::begin(feats), std::end(feats), col[2]) == std::end(feats)) continue;
    sv gid, tid;
    if (!search_id(col.back(), "gene_id", gid) || !search_id(col.back(), "transcript_id", tid))
      crash("AttributeError", "GTF line " + std::to_string(lineno) + ": 'NoneType' object has no attribute 'group'");
    if (col.size() < 7) crash("IndexError", "GTF line " + std::to_string(lineno) + ": list index out of range");
    auto it = a.gene_ix.find(gid);
    u32 gi;
    if (it == a.gene_ix.end()) {
      gi = u32(a.genes.size());
      a.gene_ix.emplace(gid, gi);
      a.genes.push_back(Gene{gid, col[6], col[0]});
    } else {
      gi = it->second;
      a.genes[gi].strand = col[6];
      a.genes[gi].chrom = col[0];
    }
    a.tx2gene[tid] = gi;
  }
}

// ---------------------------------------------------------------- the three passes
struct Opts {
  i64 rca = -1, isw = 5, minintron = 100;
  bool novel = false, no_annotated = false, es = false, ss = false, ir = false;
  bool has_junction = false;
  std::string junction;
};

struct Digest {
  u64 rows = 0, s1 = 0, s2 = 0, x = 0;
  void add(const Digest& o) { rows += o.rows; s1 += o.s1; s2 += o.s2; x ^= o.x; }
};
// one row into the digest, normalised as SURVEY.md 8-P #2 says
void digest_row(sv row, Digest& d, std::string& tmp) {
  if (row.size() > 9 && row.substr(0, 9) == "IR,novel,") {
    std::vector<sv> c;
    for (size_t i = 0;;) {
      const size_t j = row.find(',', i);
      if (j == sv::npos) { c.push_back(row.substr(i)); break; }
      c.push_back(row.substr(i, j - i));
      i = j + 1;
    }
    if (c.size() == 14) {
      tmp.clear();
      for (size_t k = 0; k < 14; ++k) {
        if (k) tmp += ',';
        if (k == 5 || k == 6) {
          std::vector<sv> parts;
          for (size_t i = 0;;) {
            const size_t j = c[k].find('|', i);
            if (j == sv::npos) { parts.push_back(c[k].substr(i)); break; }
            parts.push_back(c[k].substr(i, j - i));
            i = j + 1;
          }
          std::sort(parts.begin(), parts.end());
          for (size_t i = 0; i < parts.size(); ++i) {
            if (i) tmp += '|';
            tmp.append(parts[i]);
          }
        } else tmp.append(c[k]);
      }
      row = tmp;
    }
  }
  const u64 a = hash_bytes(row.data(), row.size(), 0x1234567), b = hash_bytes(row.data(), row.size(), 0x9abcdef01ull);
  ++d.rows;
  d.s1 += a;
  d.s2 += b;
  d.x ^= a * 3 + b;
}

struct Out {
  std::string rows, err;
};

struct Caller {
  const Graph& g;
  const Annotation& an;
  const Opts& o;
  Caller(const Graph& g_, const Annotation& a_, const Opts& o_) : g(g_), an(a_), o(o_) {}

  sv name(u32 n) const { return g.seg[n].nm(); }
  Set ex(u32 n) const { return g.seg[n].ex ? set_of_list(sv(g.seg[n].ex, g.seg[n].ex_len)) : Set(); }   // get_set_exons [:86-87]
  u32 gene_of(sv tx) const {
    auto it = an.tx2gene.find(tx);
    if (it == an.tx2gene.end()) crash("KeyError", "'" + std::string(tx) + "': transcript not in the GTF");
    return it->second;
  }
  i64 rc(u32 l) const {
    if (!g.link[l].has_rc) crash("KeyError", "'RC'");
    return g.link[l].rc;
  }
  i64 nc(u32 n) const {
    if (!g.seg[n].has_nc) crash("KeyError", "'NC'");
    return g.seg[n].nc;
  }
  sv jn_value(u32 l) const {
    if (!g.link[l].has_jn) crash("KeyError", "'JN'");
    return sv(g.link[l].jn, g.link[l].jn_len);
  }
  std::vector<sv> jn(u32 l) const { std::vector<sv> r; split_list(jn_value(l), r); return r; }
  Set jn_txs(u32 l) const {                 // get_haplotranscripts_from_junction(...).keys()
    Set s;
    for (sv t : jn(l)) add(s, tx_of_junction_tag(t));
    return s;
  }
  Set ex_txs(const Set& exons) const {
    Set s;
    for (sv e : exons) add(s, tx_of_exon_tag(e));
    return s;
  }
  std::vector<u32> genes_of(const Set& txs) const {
    std::vector<u32> r;
    for (sv t : txs) {
      const u32 x = gene_of(t);
      if (std::find(r.begin(), r.end(), x) == r.end()) r.push_back(x);
    }
    return r;
  }
  bool is_junction(u32 a, u32 b, u32& l) const {
    l = g.find_link(a, b);
    return l != NONE && g.link[l].grp_jn;
  }
  u64 rank_of(sv e) const { return g.exon_rank.at(e); }
  Set by_rank(Set s) const {
    std::sort(s.begin(), s.end(), [&](sv a, sv b) { return rank_of(a) < rank_of(b); });
    return s;
  }
  Set same_gene_exons(u32 n, u32 gene) const {          // [:258-275]
    Set r;
    for (sv e : ex(n)) if (gene_of(tx_of_exon_tag(e)) == gene) r.push_back(e);
    return r;
  }
  i64 mean_nc(const std::vector<u32>& path) const {     // ceil(sum / len) [:517-520]
    i64 s = 0;
    for (u32 x : path) s += nc(x);
    const i64 n = i64(path.size());
    return s >= 0 ? (s + n - 1) / n : -((-s) / n);
  }

  static void num(std::string& s, i64 v) {
    char b[24];
    const int n = snprintf(b, sizeof b, "%lld", static_cast<long long>(v));
    s.append(b, size_t(n));
  }
  static void join(std::string& s, const std::vector<sv>& l, char c) {
    for (size_t i = 0; i < l.size(); ++i) {
      if (i) s += c;
      s.append(l[i]);
    }
  }
  void gene_cols(std::string& s, const char* type, const char* kind, u32 gene) const {
    const Gene& G = an.genes[gene];
    s.append(type).append(",").append(kind).append(",").append(G.chrom).append(",").append(G.name).append(",").append(G.strand).append(",");
  }
  void pair(std::string& s, u32 a, u32 b) const { s.append(name(a)).append(">").append(name(b)); }
  void path_col(std::string& s, const std::vector<u32>& p) const {
    for (size_t i = 0; i < p.size(); ++i) {
      if (i) s += '>';
      s.append(name(p[i]));
    }
  }
  void jn_kept(std::string& s, u32 l, const Set& inc) const {        // JN entries whose transcript is in `inc`
    bool first = true;
    for (sv t : jn(l)) {
      if (!has(inc, tx_for_filter(t))) continue;
      if (!first) s += '|';
      s.append(t);
      first = false;
    }
  }
  bool plus(u32 gene) const { return an.genes[gene].strand == "+"; }

  // [:488-499, :843-852]: the first out-neighbour that sorts <= goal as a string and passes `ok`; the reference's loop
  // variable stays on the LAST out-neighbour when none does
  template <class OK>
  std::vector<u32> walk_exon(u32 start, u32 goal, OK ok, const char* what) const {
    std::vector<u32> path{start};
    size_t guard = g.seg.size() + 2;
    while (path.back() != goal) {
      u32 nxt = NONE;
      bool hit = false;
      g.outs(path.back(), [&](u32 nn) {
        if (hit) return;
        nxt = nn;
        if (name(nn) <= name(goal) && ok(nn)) hit = true;
      });
      if (nxt == NONE) crash("AssertionError", std::string("Error while reconstrucing ") + what + " path");
      path.push_back(nxt);
      if (--guard == 0) crash("Hang", std::string(what) + " walk does not terminate");
    }
    return path;
  }

  bool wanted_literal() const { return !o.has_junction || o.junction == "f{_j[0]}-{_j[1]}"; }      // [:241] sic
  bool wanted(u32 u, u32 v) const {
    if (!o.has_junction) return true;
    return o.junction == std::string(name(u)) + "-" + std::string(name(v));
  }

  // ---- annotated junctions [:239-524]
  void annotated(u32 jl, Out& out) const {
    const Link& J = g.link[jl];
    const u32 u = J.u, v = J.v, c = J.canon;
    if (!wanted_literal()) return;
    if (rc(c) < o.rca) return;
    const Set txs = jn_txs(c);
    const std::vector<u32> genes = genes_of(txs);
    if (genes.size() > 1) return;
    const u32 gene = genes[0];
    const Set ex_u = same_gene_exons(u, gene), ex_v = same_gene_exons(v, gene);
    if (ex_u.empty() || ex_v.empty()) crash("AssertionError", "junction without an exon of its gene on an end");
    std::vector<u32> heads, tails;           // Js1, Js2 [:280-314]
    auto same_gene = [&](u32 l) {
      for (u32 x : genes_of(jn_txs(l))) if (x == gene) return true;
      return false;
    };
    for (u64 i = g.out_ptr[u]; i < g.out_ptr[u + 1]; ++i) {
      const Link& l = g.link[g.out_idx[i]];
      if (l.first_jn && l.v != v && same_gene(l.canon)) heads.push_back(g.out_idx[i]);
    }
    for (u64 i = g.in_ptr[v]; i < g.in_ptr[v + 1]; ++i) {
      const Link& l = g.link[g.in_idx[i]];
      if (l.first_jn && l.u != u && same_gene(l.canon)) tails.push_back(g.in_idx[i]);
    }
    const std::vector<sv> jn_j = jn(c);
    std::string& s = out.rows;
    if (o.es) {                                                                     // [:316-375]
      for (u32 h : heads) {
        const Link& j1 = g.link[h];
        const Set t1 = jn_txs(j1.canon);
        for (u32 t : tails) {
          const Link& j2 = g.link[t];
          Set inc;
          for (sv x : inter(t1, jn_txs(j2.canon))) if (!has(txs, x)) inc.push_back(x);
          if (inc.empty()) continue;
          Set mid = ex(j1.v);
          for (sv e : ex(j2.u)) add(mid, e);
          bool fresh = false;
          for (sv e : mid) if (!has(ex_u, e) && !has(ex_v, e)) { fresh = true; break; }
          if (!fresh) continue;
          gene_cols(s, "ES", "annotated", gene);
          join(s, jn_j, '|'); s += ',';
          jn_kept(s, j1.canon, inc); s += ',';
          jn_kept(s, j2.canon, inc); s += ',';
          pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
          pair(s, j1.u, j1.v); s += ','; num(s, rc(j1.canon)); s += ',';
          pair(s, j2.u, j2.v); s += ','; num(s, rc(j2.canon)); s += '\n';
        }
      }
    }
    if (o.ss) {                                                                     // [:377-463]
      auto side = [&](bool first) {
        const std::vector<u32>& js = first ? tails : heads;
        if (js.empty()) return;
        const std::vector<u32> nbrs = first ? g.out_list(u) : g.in_list(v);
        for (u32 n : nbrs) {
          if (n == (first ? v : u)) continue;
          const Set cont = inter(ex(n), first ? ex_u : ex_v);
          if (cont.empty()) continue;
          for (u32 x : js) {
            const Link& j2 = g.link[x];
            // (j2[0] == _j[0] / j1[1] == _j[1] cannot hold: the pair would be _j itself, which the sets exclude)
            const u32 far = first ? j2.u : j2.v;
            const Set shared = inter(ex(far), cont);
            if (shared.empty()) continue;
            if (!any_outside(ex_txs(shared), txs)) continue;
            if (first) {
              const std::vector<sv> other = jn(j2.canon);
              for (sv a : jn_j) for (sv b : other) if (a == b) crash("AssertionError", "two junctions share a JN tag");   // [:402]
            }
            const bool p = plus(gene);
            gene_cols(s, first ? (p ? "A5" : "A3") : (p ? "A3" : "A5"), "annotated", gene);
            join(s, jn_j, '|'); s += ',';
            join(s, jn(j2.canon), '|'); s += ",.,";
            pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
            pair(s, j2.u, j2.v); s += ','; num(s, rc(j2.canon)); s += ",.,.\n";
          }
        }
      };
      side(true);
      side(false);
    }
    if (o.ir) {                                                                     // [:465-524]
      const Set both = inter(ex_u, ex_v);
      const std::vector<u32> nnext = g.out_list(u), nprev = g.in_list(v);
      std::vector<u32> found;
      for (u32 n1 : nnext) {
        for (u32 n2 : nprev) {
          if (n1 == v || n2 == u) continue;
          if (name(n2) < name(n1)) continue;
          const Set cand = inter(inter(ex(n1), ex(n2)), both);
          if (cand.empty()) continue;
          const sv e = by_rank(cand)[0];                                            // rule R2
          found = walk_exon(n1, n2, [&](u32 nn) { return has(ex(nn), e); }, "IR");
          break;
        }
        if (!found.empty()) break;
      }
      if (!found.empty()) {
        gene_cols(s, "IR", "annotated", gene);
        join(s, jn_j, '|'); s += ",.,.,";
        pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
        path_col(s, found); s += ','; num(s, mean_nc(found)); s += ",.,.\n";
      }
    }
  }

  // the level-bounded frontier search [:734-750, :787-803] with rule R3 for pop() and for the hit: `visit` is the reference's set,
  // `level_nodes` the pvisitl members it held when the level began, by S-line position -- popped before anything added since
  u32 search(u32 start, const std::vector<u32>& targets, bool forward) const {
    std::vector<u32> visit{start}, level_nodes{start};
    size_t next = 0;
    auto first_of = [&](u32 a, u32 b) { return g.seg[a].first_off < g.seg[b].first_off; };
    auto hits = [&] {
      u32 best = NONE;
      for (u32 x : visit)
        if (std::find(targets.begin(), targets.end(), x) != targets.end() && (best == NONE || first_of(x, best))) best = x;
      return best;
    };
    i64 left = 1, level = 0;
    while (hits() == NONE && level < o.isw) {
      if (visit.empty()) return NONE;
      const u32 n = level_nodes[next++];
      visit.erase(std::find(visit.begin(), visit.end(), n));
      --left;
      auto push = [&](u32 x) { if (std::find(visit.begin(), visit.end(), x) == visit.end()) visit.push_back(x); };
      if (forward) g.outs(n, push); else g.ins(n, push);
      if (left == 0) {
        ++level;
        left = i64(visit.size());
        level_nodes = visit;
        std::sort(level_nodes.begin(), level_nodes.end(), first_of);
        next = 0;
      }
    }
    return level < o.isw ? hits() : NONE;
  }

  // ---- links the reads added [:531-893]
  void novel_link(u32 jl, Out& out) const {
    const Link& J = g.link[jl];
    const u32 u = J.u, v = J.v, c = J.canon;
    if (!wanted(u, v)) return;
    if (rc(c) < o.rca) return;
    const Set ex0 = ex(u), ex1 = ex(v);
    if (ex0.empty() && ex1.empty()) return;
    const Set cap = inter(ex_txs(ex0), ex_txs(ex1));
    const std::vector<u32> nxt = g.out_list(u), prv = g.in_list(v);
    std::string& s = out.rows;
    if (o.es) {                                                                     // [:550-614]
      if (!ex0.empty() && !ex1.empty() && inter(ex0, ex1).size() != ex0.size()) {
        std::vector<u32> n1, n2;
        u32 l;
        for (u32 n : nxt) if (is_junction(u, n, l)) n1.push_back(l);
        for (u32 p : prv) if (is_junction(p, v, l)) n2.push_back(l);
        for (u32 l1 : n1) {
          for (u32 l2 : n2) {
            const Set inc = inter(jn_txs(l1), jn_txs(l2));
            if (inc.empty()) continue;
            const std::vector<u32> genes = genes_of(inc);
            if (genes.size() > 1) { out.err += "Skipping ES due to multiple genes\n"; continue; }
            gene_cols(s, "ES", "novel", genes[0]);
            s += "?,";
            jn_kept(s, l1, inc); s += ',';
            jn_kept(s, l2, inc); s += ',';
            pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
            pair(s, g.link[l1].u, g.link[l1].v); s += ','; num(s, rc(l1)); s += ',';
            pair(s, g.link[l2].u, g.link[l2].v); s += ','; num(s, rc(l2)); s += '\n';
          }
        }
      }
    }
    if (o.ss) {
      if (!cap.empty()) {                                                           // [:617-726]
        for (int side = 0; side < 2; ++side) {
          Set spliced;
          if (side == 0) {
            if (std::find(prv.begin(), prv.end(), u) != prv.end()) spliced = ex1;   // [:623-626]
            else for (u32 p : prv) spliced = inter(ex(p), ex1);                     // [:629-631]
          } else {
            if (std::find(nxt.begin(), nxt.end(), v) != nxt.end()) spliced = ex0;   // [:677-680]
            else crash("Unsupported", "call.py:685 reads a loop variable left over from an earlier junction of a set iteration");
          }
          if (spliced.empty()) continue;
          std::vector<std::pair<u32, Set>> hits;
          for (u32 n : (side == 0 ? nxt : prv)) {
            const Set shared = inter(ex(n), spliced);
            if (shared.empty()) continue;
            const Set keep = inter(ex_txs(shared), cap);
            u32 l;
            if ((side == 0 ? is_junction(u, n, l) : is_junction(n, v, l)) && !keep.empty()) hits.emplace_back(l, keep);
          }
          for (auto& h : hits) {
            const std::vector<u32> genes = genes_of(h.second);
            if (genes.size() > 1) { out.err += "Skipping SS due to multiple genes\n"; continue; }
            const bool p = plus(genes[0]);
            gene_cols(s, side == 0 ? (p ? "A3" : "A5") : (p ? "A5" : "A3"), "novel", genes[0]);
            s += "?,";
            join(s, jn(h.first), '|'); s += ",.,";
            pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
            pair(s, g.link[h.first].u, g.link[h.first].v); s += ','; num(s, rc(h.first)); s += ",.,.\n";
          }
        }
      }
      u32 j1 = NONE;                                                                // [:730-835]
      if (!ex0.empty() && ex1.empty()) {
        std::vector<u32> targets;
        u32 l;
        for (u32 n : nxt) if (is_junction(u, n, l)) targets.push_back(n);
        const u32 hit = search(v, targets, true);
        if (hit != NONE) j1 = g.find_link(u, hit);
      } else if (ex0.empty() && !ex1.empty()) {
        std::vector<u32> targets;
        u32 l;
        for (u32 p : prv) if (is_junction(p, v, l)) targets.push_back(p);
        const u32 hit = search(u, targets, false);
        if (hit != NONE) j1 = g.find_link(hit, v);
      }
      if (j1 != NONE) {
        const std::vector<u32> genes = genes_of(jn_txs(j1));
        if (genes.size() > 1) out.err += "Skipping SS due to multiple genes\n";
        else {
          gene_cols(s, plus(genes[0]) ? "A3" : "A5", "novel", genes[0]);
          join(s, jn(j1), '|'); s += ",?,.,";
          pair(s, g.link[j1].u, g.link[j1].v); s += ','; num(s, rc(j1)); s += ',';
          pair(s, u, v); s += ','; num(s, rc(c)); s += ",.,.\n";
        }
      }
    }
    if (o.ir && !cap.empty()) {                                                     // [:837-893]
      const Set inside = inter(ex0, ex1);
      if (!inside.empty()) {
        const std::vector<u32> path = walk_exon(u, v, [&](u32 nn) { return !inter(inside, ex(nn)).empty(); }, "novel IR");
        i64 len = 0;
        for (u32 x : path) len += g.seg[x].ln;
        if (len >= o.minintron) {
          const std::vector<u32> genes = genes_of(ex_txs(inside));
          if (genes.size() > 1) out.err += "Skipping novel IR due to multiple genes\n";
          else {
            gene_cols(s, "IR", "novel", genes[0]);
            s += "?,";
            join(s, by_rank(inside), '|'); s += ",.,";                              // rule R4
            pair(s, u, v); s += ','; num(s, rc(c)); s += ',';
            path_col(s, path); s += ','; num(s, mean_nc(path)); s += ",.,.\n";
          }
        }
      }
    }
  }

  // ---- annotated junctions again: cassette exons, read-through introns [:895-1102]
  void around(u32 jl, Out& out) const {
    const Link& J = g.link[jl];
    const u32 u = J.u, v = J.v, c = J.canon;
    if (!wanted(u, v)) return;
    if (rc(c) < o.rca) return;
    const std::vector<u32> genes = genes_of(jn_txs(c));
    if (genes.size() > 1) return;
    const u32 gene = genes[0];
    const Set ex0 = same_gene_exons(u, gene), ex1 = same_gene_exons(v, gene);
    if (ex0.empty() || ex1.empty()) crash("AssertionError", "junction without an exon of its gene on an end");
    std::string& s = out.rows;
    if (o.es) {                                                                     // [:954-984]
      std::vector<u32> h, t;
      for (u64 i = g.out_ptr[u]; i < g.out_ptr[u + 1]; ++i) {
        const Link& l = g.link[g.out_idx[i]];
        if (l.first_id && l.v != v) h.push_back(l.canon);
      }
      for (u64 i = g.in_ptr[v]; i < g.in_ptr[v + 1]; ++i) {
        const Link& l = g.link[g.in_idx[i]];
        if (l.first_id && l.u != u) t.push_back(l.canon);
      }
      if (!h.empty() && !t.empty()) {
        std::string head;
        gene_cols(head, "CE", "novel", gene);
        join(head, jn(c), '|'); head += ",?,?,";
        pair(head, u, v); head += ','; num(head, rc(c)); head += ',';
        std::vector<std::string> tail(t.size());
        for (size_t k = 0; k < t.size(); ++k) {
          pair(tail[k], g.link[t[k]].u, v); tail[k] += ','; num(tail[k], rc(t[k])); tail[k] += '\n';
        }
        std::string mid;
        for (u32 l1 : h) {
          const sv es = name(g.link[l1].v);
          mid.clear();
          pair(mid, u, g.link[l1].v); mid += ','; num(mid, rc(l1)); mid += ',';
          // (two novel links out of u differ in their targets, two into v in their sources: no pair (es, ee) comes twice)
          for (size_t k = 0; k < t.size(); ++k) {
            if (!(es <= name(g.link[t[k]].u))) continue;                            // [:961] compared as strings
            s.append(head).append(mid).append(tail[k]);
          }
        }
      }
    }
    if (o.ir) {                                                                     // [:985-1102]
      auto probe = [&](bool left) {
        i64 base;
        if (!parse_int(name(left ? u : v), base)) crash("ValueError", "invalid literal for int() with base 10: '" + std::string(name(left ? u : v)) + "'");
        bool any = false;
        for (i64 x = left ? base + 1 : base - 3; x < (left ? base + 4 : base); ++x) {
          const u32 n = g.find(std::to_string(x));
          if (n == NONE) continue;
          const u32 l = left ? g.find_link(u, n) : g.find_link(n, v);
          if (l == NONE || g.link[l].grp_jn) continue;
          if (rc(l) >= o.rca) any = true;          // (a list comprehension: every candidate's RC is read)
        }
        return any;
      };
      if (!probe(true) || !probe(false)) return;
      const std::vector<u32> outs_u = g.out_list(u), ins_v = g.in_list(v);
      Set ends, starts;
      for (sv e : by_rank(ex0)) {
        bool on = false;
        for (u32 x : outs_u) if (has(ex(x), e)) { on = true; break; }
        if (!on) ends.push_back(e);
      }
      for (sv e : by_rank(ex1)) {
        bool on = false;
        for (u32 x : ins_v) if (has(ex(x), e)) { on = true; break; }
        if (!on) starts.push_back(e);
      }
      std::vector<std::pair<sv, sv>> pairs;
      for (sv a : ends) for (sv b : starts) if (tx_of_exon_tag(a) == tx_of_exon_tag(b)) pairs.emplace_back(a, b);
      if (pairs.empty()) return;
      std::vector<u32> path{u};
      size_t guard = g.seg.size() + 2;
      while (path.back() != v) {
        std::vector<u32> outs;
        g.outs(path.back(), [&](u32 x) { if (name(x) <= name(v)) outs.push_back(x); });
        if (path.size() == 1) {
          auto it = std::find(outs.begin(), outs.end(), v);
          if (it == outs.end()) crash("ValueError", "list.remove(x): x not in list");
          outs.erase(it);
        }
        if (std::find(outs.begin(), outs.end(), v) != outs.end()) { path.push_back(v); continue; }
        u32 best = NONE;
        for (u32 x : outs) {
          if (!g.has_out(x)) continue;
          if (best == NONE || nc(x) > nc(best)) best = x;      // max(): the first maximum [:1049]
          nc(x);
        }
        if (best == NONE) crash("ValueError", "max() arg is an empty sequence");
        path.push_back(best);
        if (--guard == 0) crash("Hang", "novel IR (2) walk does not terminate");
      }
      i64 len = 0;
      for (size_t i = 1; i + 1 < path.size(); ++i) len += g.seg[path[i]].ln;
      if (len < o.minintron) return;
      Set firsts;
      for (auto& p : pairs) add(firsts, p.first);
      const std::vector<u32> genes2 = genes_of(ex_txs(firsts));
      if (genes2.size() > 1) { out.err += "Skipping novel IR due to multiple genes\n"; return; }
      gene_cols(s, "IR", "novel", genes2[0]);
      for (size_t i = 0; i < pairs.size(); ++i) {
        if (i) s += '|';
        s.append(pairs[i].first).append(".").append(after_last(pairs[i].second, '.'));
      }
      s += ",?,.,";
      path_col(s, path); s += ','; num(s, mean_nc(path)); s += ',';
      pair(s, u, v); s += ','; num(s, rc(c)); s += ",.,.\n";
    }
  }
};

const char* HEADER = "event_type,annotated/novel,chrom,gene,strand,transcripts1,transcripts2,transcripts3,"
                     "nodes1,coverage1,nodes2,coverage2,nodes3,coverage3\n";

void write_all(int fd, const char* p, size_t n) {
  while (n) {
    const ssize_t w = ::write(fd, p, n);
    if (w < 0) { perror("write"); exit(2); }
    p += w;
    n -= size_t(w);
  }
}

Digest digest_text(sv text, int T) {
  std::vector<Digest> part(static_cast<size_t>(T));
  std::vector<size_t> cut(size_t(T) + 1, text.size());
  cut[0] = 0;
  for (int t = 1; t < T; ++t) {
    size_t c = std::max(cut[size_t(t) - 1], text.size() * size_t(t) / size_t(T));
    if (c > 0 && c < text.size()) {
      const size_t q = text.find('\n', c - 1);
      c = q == sv::npos ? text.size() : q + 1;
    }
    cut[size_t(t)] = c;
  }
  parallel(T, [&](int t) {
    std::string tmp;
    size_t pos = cut[size_t(t)];
    while (pos < cut[size_t(t) + 1]) {
      size_t q = text.find('\n', pos);
      if (q == sv::npos) q = text.size();
      digest_row(text.substr(pos, q - pos), part[size_t(t)], tmp);
      pos = q + 1;
    }
  });
  Digest d;
  for (auto& p : part) d.add(p);
  return d;
}

void print_digest(FILE* f, const Digest& d, const char* extra) {
  fprintf(f, "{\"rows\": %llu, \"s1\": \"%016llx\", \"s2\": \"%016llx\", \"x\": \"%016llx\"%s}\n", static_cast<unsigned long long>(d.rows),
          static_cast<unsigned long long>(d.s1), static_cast<unsigned long long>(d.s2), static_cast<unsigned long long>(d.x), extra);
}

}  // namespace

int main(int argc, char** argv) {
  Opts o;
  int T = 1;
  bool no_out = false, want_digest = false;
  std::vector<std::string> events{"ES", "SS", "IR"}, pos;
  const char* digest_csv = nullptr;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* {
      if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", a.c_str()); exit(2); }
      return argv[++i];
    };
    if (a == "--threads") T = std::max(1, atoi(val()));
    else if (a == "--rca") o.rca = atoll(val());
    else if (a == "--isw") o.isw = atoll(val());
    else if (a == "--minintronsize") o.minintron = atoll(val());
    else if (a == "--novel") o.novel = true;
    else if (a == "--no-annotated") o.no_annotated = true;
    else if (a == "--no-out") no_out = true;
    else if (a == "--digest") want_digest = true;
    else if (a == "--digest-csv") digest_csv = val();
    else if (a == "--junction") { o.has_junction = true; o.junction = val(); }
    else if (a == "--events") {
      events.clear();
      while (i + 1 < argc && strncmp(argv[i + 1], "--", 2) != 0) events.push_back(argv[++i]);
    } else pos.push_back(a);
  }
  if (digest_csv) {
    Mapped m;
    m.open(digest_csv, T);
    sv text(m.p, m.n);
    const size_t nl = text.find('\n');
    text = nl == sv::npos ? sv() : text.substr(nl + 1);
    print_digest(stdout, digest_text(text, T), "");
    return 0;
  }
  if (events.size() >= 2 && pos.size() < 2) {       // argparse: --events takes every word up to the next option
    while (pos.size() < 2 && !events.empty()) { pos.insert(pos.begin(), events.back()); events.pop_back(); }
  }
  if (pos.size() != 2) { fprintf(stderr, "usage: call_oracle_cpp [options] GFA GTF\n"); return 2; }
  for (auto& e : events) {
    if (e == "ES") o.es = true;
    if (e == "SS") o.ss = true;
    if (e == "IR") o.ir = true;
  }
  const double t0 = now();
  Graph g;
  Annotation an;
  Mapped gfa, gtf;
  try {
    gfa.open(pos[0].c_str(), T);
    load_gfa(gfa, T, g);
  } catch (Crash& c) {
    fprintf(stderr, "%s: %s\n", c.kind.c_str(), c.msg.c_str());
    return 1;
  }
  const double t1 = now();
  try {
    gtf.open(pos[1].c_str(), 1);
    load_gtf(gtf, an);
  } catch (Crash& c) {
    fprintf(stderr, "%s: %s\n", c.kind.c_str(), c.msg.c_str());
    return 1;
  }
  const double t2 = now();
  if (!no_out) write_all(1, HEADER, strlen(HEADER));

  // ---- the passes, junction block by junction block; blocks leave in order (rule R1)
  struct Pass {
    int kind;
    const std::vector<u32>* items;
  };
  std::vector<Pass> passes;
  if (!o.no_annotated) passes.push_back({0, &g.junctions});
  if (o.novel) {
    passes.push_back({1, &g.novel});
    if (o.ir || o.es) passes.push_back({2, &g.junctions});
  }
  constexpr size_t BLOCK = 256;
  std::vector<std::pair<int, size_t>> blocks;        // (pass, first item)
  for (size_t p = 0; p < passes.size(); ++p)
    for (size_t b = 0; b < passes[p].items->size(); b += BLOCK) blocks.emplace_back(int(p), b);
  Caller caller(g, an, o);
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::condition_variable cv;
  size_t turn = 0;
  bool failed = false;
  Crash first_crash;
  size_t crash_block = SIZE_MAX;
  Digest total;
  u64 bytes = 0, skipped = 0;
  parallel(T, [&](int) {
    Out out;
    std::string tmp;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= blocks.size()) break;
      out.rows.clear();
      out.err.clear();
      bool bad = false;
      Crash mine;
      const Pass& P = passes[size_t(blocks[b].first)];
      const size_t lo = blocks[b].second, hi = std::min(lo + BLOCK, P.items->size());
      try {
        for (size_t i = lo; i < hi; ++i) {
          const u32 l = (*P.items)[i];
          if (P.kind == 0) caller.annotated(l, out);
          else if (P.kind == 1) caller.novel_link(l, out);
          else caller.around(l, out);
        }
      } catch (Crash& c) {
        bad = true;
        mine = c;
      }
      Digest d;
      if (want_digest && !bad) d = digest_text(out.rows, 1);
      else d.rows = u64(std::count(out.rows.begin(), out.rows.end(), '\n'));
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return turn == b; });
      if (bad && b < crash_block) { failed = true; first_crash = mine; crash_block = b; }
      if (!failed) {
        if (!no_out) write_all(1, out.rows.data(), out.rows.size());
        if (!out.err.empty()) { fputs(out.err.c_str(), stderr); skipped += u64(std::count(out.err.begin(), out.err.end(), '\n')); }
        total.add(d);
        bytes += out.rows.size();
      }
      ++turn;
      lk.unlock();
      cv.notify_all();
      if (bad) { next.store(blocks.size()); }
    }
    // a thread that stopped early must not leave later turns waiting
  });
  if (failed) {
    fprintf(stderr, "%s: %s\n", first_crash.kind.c_str(), first_crash.msg.c_str());
    return 1;
  }
  const double t3 = now();
  char extra[512];
  snprintf(extra, sizeof extra,
           ", \"bytes\": %llu, \"threads\": %d, \"segments\": %zu, \"links\": %zu, \"junctions\": %zu, \"novel_links\": %zu, "
           "\"gfa_s\": %.3f, \"gtf_s\": %.3f, \"passes_s\": %.3f, \"total_s\": %.3f",
           static_cast<unsigned long long>(bytes), T, g.seg.size(), g.link.size(), g.junctions.size(), g.novel.size(), t1 - t0, t2 - t1, t3 - t2, t3 - t0);
  print_digest(stderr, total, extra);
  fflush(stdout);
  _exit(0);       // (tens of GB of tables: no point in freeing them one by one)
}
