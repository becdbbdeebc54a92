// CPU oracle for the GAF-coverage augmentation path in plain C++.  TEST INFRASTRUCTURE ONLY.
//
// The same restatement as oracle/augment_oracle.py (which follows the reference script
// scripts/alignments_augmentation_from_gaf.py, main at :110-427; bracketed numbers below are its lines), written
// for speed so that bench.py's `cpu_baseline` leg can time "the algorithm on the host cores" (SURVEY.md 8-d rows
// (2a) C++ single thread and (2b) C++ all cores) beside the GPU.  Only tests/, __graft_entry__.smoke() and that leg of
// bench.py may build or run it; nothing in pantas_amd/ does.
//
// Parity status: PINNED -- tests/test_oracle_cpp.py compares its output byte for byte with the outputs of the real
// reference captured under tests/golden/ (every case of tests/golden/cases.json that the reference finishes, and the
// exception class where it dies), with one thread and with several.
//
//   augment_oracle_cpp [--threads T] [--thr N] [--no-out] [--digest] GAF GFA  > augmented.gfa
//   augment_oracle_cpp [--threads T] --digest-file FILE
//
// --digest: nothing is written; the last stderr line carries {"bytes", "h1", "h2"} of the output the run would have written
// (an order-dependent polynomial digest computed piecewise); --digest-file prints the same of any file.  Round 5: with
// T > 1 every phase runs on T threads (name table in T hash maps, merge by key shards, pass 2 in pieces of the text) -- the
// whole-genome check fell from ten minutes to what fits the default GPU test run.
//
// --threads T > 1: the GAF is cut into T byte ranges at line starts; every thread feeds its lines into tables of its
// own that also remember WHEN a key first appeared (line number, event number inside the line); the merge adds the
// counts and keeps the earliest appearance, so the output is the single-thread output.  What cannot be cut that way is
// the reference's `curr_cigar_len` leaking from one line into the next (a '~' as the first op a line reaches): with
// T > 1 such a line is an error.
// Deviations from Python's text handling, none of which the goldens or the synthetic inputs contain: integers are
// [+-]?[0-9]+, lengths count bytes.  (Universal newlines are applied when a file is mapped.)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
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

struct Crash {
  std::string kind, msg;
};
[[noreturn]] void crash(const char* kind, const std::string& msg) { throw Crash{kind, msg}; }

struct Mapped {
  const char* p = nullptr;
  size_t n = 0;
  void open(const char* path) {
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { perror(path); exit(2); }
    struct stat st;
    fstat(fd, &st);
    n = size_t(st.st_size);
    if (n) {
      // a private, writable mapping: Python's universal newlines are applied in place -- a '\r' that no '\n' follows ends its line
      char* w = static_cast<char*>(mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd, 0));
      if (w == MAP_FAILED) { perror("mmap"); exit(2); }
      for (char* q = w; (q = static_cast<char*>(memchr(q, '\r', size_t(w + n - q)))) != nullptr; ++q)
        if (q + 1 < w + n && q[1] != '\n') *q = '\n';
      p = w;
    }
    close(fd);
  }
};

inline bool is_space(char c) { return c == ' ' || (c >= '\t' && c <= '\r') || (c >= 0x1c && c <= 0x1f); }   // str.split() on ASCII
inline sv strip(sv s) {
  while (!s.empty() && is_space(s.front())) s.remove_prefix(1);
  while (!s.empty() && is_space(s.back())) s.remove_suffix(1);
  return s;
}
inline void split(sv s, std::vector<sv>& out) {
  out.clear();
  size_t i = 0;
  while (i < s.size()) {
    while (i < s.size() && is_space(s[i])) ++i;
    size_t j = i;
    while (j < s.size() && !is_space(s[j])) ++j;
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j;
  }
}
inline bool to_int(sv t, long long& v) {
  size_t i = 0;
  bool neg = false;
  if (i < t.size() && (t[i] == '+' || t[i] == '-')) { neg = t[i] == '-'; ++i; }
  if (i >= t.size()) return false;
  v = 0;
  for (; i < t.size(); ++i) {
    if (t[i] < '0' || t[i] > '9') return false;
    v = v * 10 + (t[i] - '0');
  }
  if (neg) v = -v;
  return true;
}

struct Op { char op; long long n; };
inline bool is_op(char c) { return c == '=' || c == ':' || c == '*' || c == '+' || c == '-' || c == '~'; }

// [:10-37]
void split_cs(sv cs, std::vector<Op>& out) {
  out.clear();
  size_t i = 0;
  while (i < cs.size() && !is_op(cs[i])) ++i;
  while (i < cs.size()) {
    const char op = cs[i];
    size_t j = i + 1;
    while (j < cs.size() && !is_op(cs[j])) ++j;
    const sv run = cs.substr(i + 1, j - i - 1);
    bool digits = !run.empty();
    for (char c : run) digits = digits && c >= '0' && c <= '9';
    if (op == '*') out.push_back({'*', 1});
    else if (digits) { long long v = 0; for (char c : run) v = v * 10 + (c - '0'); out.push_back({op, v}); }
    else out.push_back({op, (long long)run.size()});
    i = j;
  }
}

// first appearance of a key: (line number, event number inside the line)
struct When {
  uint64_t line = 0;
  uint32_t seq = 0;
  bool operator<(const When& o) const { return line != o.line ? line < o.line : seq < o.seq; }
};
struct Entry { uint64_t key; uint64_t count; When first; };
// insertion-ordered map of 64-bit keys
struct Table {
  std::unordered_map<uint64_t, uint32_t> at;
  std::vector<Entry> e;
  void add(uint64_t key, uint64_t n, const When& w) {
    auto ins = at.emplace(key, uint32_t(e.size()));
    if (ins.second) e.push_back(Entry{key, n, w});
    else { Entry& x = e[ins.first->second]; x.count += n; if (w < x.first) x.first = w; }
  }
};

constexpr uint32_t NO_NODE = 0xFFFFFFFFu;
inline size_t name_hash(sv s) { return std::hash<sv>{}(s); }

// runs f(t) on T threads
template <class F>
void parallel(unsigned T, F f) {
  if (T <= 1) { f(0u); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back([&f, t] { f(t); });
  for (auto& x : th) x.join();
}

// a text cut into pieces at line starts
std::vector<size_t> cut_lines(const char* p, size_t n, size_t pieces) {
  std::vector<size_t> cut{0};
  for (size_t k = 1; k < pieces; ++k) {
    size_t at = n / pieces * k;
    if (at <= cut.back()) continue;
    const char* nl = static_cast<const char*>(memchr(p + at, '\n', n - at));
    if (!nl || size_t(nl - p) + 1 >= n) break;
    cut.push_back(size_t(nl - p) + 1);
  }
  cut.push_back(n);
  return cut;
}

// The segment table [:121-126].  A node's number is the ordinal of the FIRST S line with its name (numbers of repeated names stay
// unused); a repeated S line overwrites the length.  With several threads the text is scanned in pieces and the names go into
// one hash table per thread, chosen by the name's hash: every table is filled by its own thread in file order, so what a
// table holds does not depend on the number of threads.
struct Graph {
  Mapped text;
  std::vector<std::unordered_map<sv, uint32_t>> id;      // S name -> node, by name_hash % id.size()
  std::vector<sv> name;
  std::vector<long long> len;
  uint32_t find(sv nm) const {
    const auto& m = id[id.size() == 1 ? 0 : name_hash(nm) % id.size()];
    const auto it = m.find(nm);
    return it == m.end() ? NO_NODE : it->second;
  }
  void load(const char* path, unsigned threads) {
    text.open(path);
    struct Rec { sv nm; uint32_t len, nth; };        // nth: which S line of its piece
    const std::vector<size_t> cut = cut_lines(text.p, text.n, threads == 1 ? 1 : size_t(threads) * 4);
    const size_t C = cut.size() - 1;
    std::vector<std::vector<std::vector<Rec>>> recs(C, std::vector<std::vector<Rec>>(threads));   // [piece][table]
    std::vector<char> bad(C, 0);
    std::vector<uint64_t> n_s(C + 1, 0);
    std::atomic<size_t> next{0};
    parallel(threads, [&](unsigned) {
      std::vector<sv> tok;
      for (size_t c; (c = next.fetch_add(1)) < C;) {
        uint32_t nth = 0;
        size_t i = cut[c];
        while (i < cut[c + 1]) {
          const char* nl = static_cast<const char*>(memchr(text.p + i, '\n', cut[c + 1] - i));
          const size_t e = nl ? size_t(nl - text.p) : cut[c + 1];
          const sv line(text.p + i, e - i);
          if (!line.empty() && line[0] == 'S') {
            split(strip(line), tok);
            if (tok.size() < 3) { bad[c] = 1; break; }
            recs[c][threads == 1 ? 0 : name_hash(tok[1]) % threads].push_back(Rec{tok[1], uint32_t(tok[2].size()), nth++});
          }
          i = e + 1;
        }
        n_s[c + 1] = nth;
      }
    });
    for (size_t c = 0; c < C; ++c)
      if (bad[c]) crash("IndexError", "S line with fewer than 3 fields");
    for (size_t c = 0; c < C; ++c) n_s[c + 1] += n_s[c];
    if (n_s[C] >= NO_NODE) crash("Unsupported", "more than 2^32 - 2 S lines");
    name.assign(n_s[C], sv());
    len.assign(n_s[C], 0);
    id.resize(threads);
    parallel(threads, [&](unsigned t) {
      size_t mine = 0;
      for (size_t c = 0; c < C; ++c) mine += recs[c][t].size();
      auto& m = id[t];
      m.reserve(mine);
      for (size_t c = 0; c < C; ++c)
        for (const Rec& r : recs[c][t]) {
          auto ins = m.emplace(r.nm, uint32_t(n_s[c]) + r.nth);
          if (ins.second) name[ins.first->second] = r.nm;
          len[ins.first->second] = (long long)r.len;
        }
    });
  }
};

constexpr uint64_t SIDE_IL = 0, SIDE_OL = 1;
inline uint64_t hist_key(uint32_t node, uint64_t side, long long pos) {
  // positions are node offsets (|pos| < 2^31 on anything real); the sign is kept
  return (uint64_t(node) << 33) | (side << 32) | uint64_t(uint32_t(int32_t(pos)));
}

struct Augmenter {
  const Graph& g;
  long long thr;
  bool shared_carry;                 // single thread: the reference's curr_cigar_len lives across lines
  std::unordered_map<uint32_t, uint32_t> nc;   // per thread (a shared dense array has every core fighting for the lines of the
                                               // few highly expressed genes' nodes); added up in the merge
  Table hist, rc;                    // (node, side, pos) -> count; (u << 32 | v) -> count
  uint64_t rejected = 0, used = 0;
  bool have_carry = false;
  long long carry = 0;
  // scratch
  std::vector<sv> tok, names;
  std::vector<Op> ops;
  std::vector<long long> room;
  struct Piece { char op; long long n; };
  std::vector<std::vector<Piece>> pieces;
  std::vector<uint32_t> placed_pos, kept_pos, node_of;

  Augmenter(const Graph& g_, long long thr_, bool sc) : g(g_), thr(thr_), shared_carry(sc) {}

  void feed(sv raw, uint64_t lineno) {
    split(strip(raw), tok);
    if (tok.size() < 12) crash("IndexError", "line " + std::to_string(lineno) + ": fewer than 12 columns");
    long long mapq;
    if (!to_int(tok[11], mapq)) crash("ValueError", "line " + std::to_string(lineno) + ": mapq");
    if (mapq < thr) { ++rejected; return; }                                   // [:143-146]
    if (tok[5] == "*") return;                                                // [:147-148]
    long long path_len, start, end;
    if (!to_int(tok[6], path_len) || !to_int(tok[7], start) || !to_int(tok[8], end))
      crash("ValueError", "line " + std::to_string(lineno) + ": path_len/start/end");
    const long long tail = path_len - end;                                    // [:153]
    // cs: first "cs:" in the tags, up to the end of its token, every "cs:Z:" removed  [:154-160]
    std::string cs_buf;
    sv cs = "*";
    for (size_t i = 12; i < tok.size(); ++i) {
      const size_t at = tok[i].find("cs:");
      if (at == sv::npos) continue;
      sv m = tok[i].substr(at);
      if (m.find("cs:Z:") != sv::npos) {
        cs_buf.clear();
        size_t q = 0;
        while (q < m.size()) {
          if (m.compare(q, 5, "cs:Z:") == 0) q += 5; else cs_buf.push_back(m[q++]);
        }
        cs = cs_buf;
      } else cs = m;
      break;
    }
    split_cs(cs, ops);
    if (ops.size() == 2) {                                                    // [:40-50, :164-167]
      if (ops[0].op == '+' && ops[1].op == ':') { start += ops[0].n; ops.erase(ops.begin()); }
      else if (ops[0].op == ':' && ops[1].op == '+') ops.pop_back();
    }
    // dv:f:(\d+(\.\d+)?) first occurrence in the joined tags  [:172-180]
    bool have_dv = false;
    double dv = 0;
    for (size_t i = 12; i < tok.size() && !have_dv; ++i) {
      size_t at = 0;
      while ((at = tok[i].find("dv:f:", at)) != sv::npos) {
        size_t q = at + 5, b = q;
        while (q < tok[i].size() && tok[i][q] >= '0' && tok[i][q] <= '9') ++q;
        if (q > b) {
          size_t e = q;
          if (q + 1 < tok[i].size() && tok[i][q] == '.' && tok[i][q + 1] >= '0' && tok[i][q + 1] <= '9') {
            e = q + 1;
            while (e < tok[i].size() && tok[i][e] >= '0' && tok[i][e] <= '9') ++e;
          }
          dv = strtod(std::string(tok[i].substr(b, e - b)).c_str(), nullptr);
          have_dv = true;
          break;
        }
        at += 5;
      }
    }
    if (!have_dv) crash("ValueError", "line " + std::to_string(lineno) + ": no dv:f: tag");
    if (dv > 0.1) return;
    const sv p = tok[5];
    const bool rev = p[0] != '>';                                             // [:186-194]
    const char sep = rev ? '<' : '>';
    names.clear();
    {
      size_t i = p.find(sep);
      while (i != sv::npos) {
        size_t j = p.find(sep, i + 1);
        const sv nm = p.substr(i + 1, (j == sv::npos ? p.size() : j) - i - 1);
        if (names.empty() || names.back() != nm) names.push_back(nm);
        i = j;
      }
    }
    if (names.empty()) crash("AssertionError", "line " + std::to_string(lineno) + ": empty path");
    const size_t K = names.size();
    room.resize(K);
    node_of.resize(K);
    for (size_t i = 0; i < K; ++i) {                                          // [:211-219]
      const uint32_t nd = g.find(names[i]);
      if (nd == NO_NODE) crash("KeyError", "line " + std::to_string(lineno) + ": node not in GFA");
      node_of[i] = nd;
      long long r = g.len[nd];
      if (i == 0) r -= start;
      if (i == K - 1) r = r - tail + 1;
      room[i] = r;
    }
    // spread [:209-255]
    if (pieces.size() < K) pieces.resize(K);
    placed_pos.clear();
    size_t head = 0;
    for (size_t pos = 0; pos < K; ++pos) {
      long long r = room[pos];
      bool opened = false;
      while (r > 0) {
        if (head >= ops.size()) crash("IndexError", "line " + std::to_string(lineno) + ": cs ops exhausted before path end");
        const char op = ops[head].op;
        long long n = ops[head].n;
        if (op == '~') {
          if (!have_carry) {
            if (shared_carry) crash("UnboundLocalError", "line " + std::to_string(lineno) + ": '~' as first walked op");
            crash("Unsupported", "line " + std::to_string(lineno) + ": '~' as the first op of a line needs the previous line (single thread only)");
          }
          n = carry;
        } else { carry = n; have_carry = true; }
        if (!opened) { opened = true; pieces[placed_pos.size()].clear(); placed_pos.push_back(uint32_t(pos)); }
        std::vector<Piece>& pc = pieces[placed_pos.size() - 1];
        if (r <= n) {
          pc.push_back({op, r});
          ops[head].n = n - r;
          if (ops[head].n == 0) ++head;
          r = 0;
        } else {
          pc.push_back({op, n});
          r -= n;
          ++head;
        }
        if (head >= ops.size()) break;
      }
    }
    if (!shared_carry) have_carry = false;      // T > 1: nothing leaks between lines
    // tidy + squeeze [:63-107], in place
    kept_pos.clear();
    size_t w = 0;
    for (size_t q = 0; q < placed_pos.size(); ++q) {
      std::vector<Piece>& pc = pieces[q];
      if (pc.size() == 1 && (pc[0].op == '+' || pc[0].op == '-')) continue;
      std::vector<Piece> kept;
      for (size_t idx = 0; idx < pc.size(); ++idx) {
        const Piece x = pc[idx];
        if (idx == 0) { if (x.op != '*') kept.push_back(x); continue; }
        if (kept.empty()) { kept.push_back({x.op, x.n + 1}); continue; }
        if (x.op == kept.back().op || x.op == '*') kept.back().n += x.n;
        else kept.push_back(x);
      }
      if (w != q) std::swap(pieces[w], pieces[q]);
      pieces[w] = std::move(kept);
      kept_pos.push_back(placed_pos[q]);
      ++w;
    }
    const size_t F = kept_pos.size();
    for (size_t i = 0; i < F; ++i) ++nc[node_of[kept_pos[i]]];                // [:263-269]
    uint32_t seq = 0;
    for (size_t i = 0; i < F; ++i) {                                          // [:276-353]
      const uint32_t node = node_of[kept_pos[i]];
      const long long ln = g.len[node];
      const uint64_t enter = rev ? SIDE_OL : SIDE_IL, leave = rev ? SIDE_IL : SIDE_OL;
      const std::vector<Piece>& pc = pieces[i];
      for (size_t j = 0; j < pc.size(); ++j) {
        const char op = pc[j].op;
        const long long n = pc[j].n;
        if (op == '-') {
          if (i != 0 && j == 0) hist.add(hist_key(node, enter, !rev ? n : ln - 1 - n), 1, When{lineno, seq++});
          if (i != F - 1 && j == pc.size() - 1) hist.add(hist_key(node, leave, !rev ? ln - n - 1 : n), 1, When{lineno, seq++});
        } else if (op != '*') {
          if (!rev) {
            if (i != 0) hist.add(hist_key(node, enter, 0), 1, When{lineno, seq++});
            if (i != F - 1) hist.add(hist_key(node, leave, ln), 1, When{lineno, seq++});
          } else {
            if (i != F - 1) hist.add(hist_key(node, leave, 0), 1, When{lineno, seq++});
            if (i != 0) hist.add(hist_key(node, enter, ln), 1, When{lineno, seq++});
          }
        }
      }
    }
    for (size_t i = 0; i + 1 < F; ++i) {                                      // [:357-363]
      const uint32_t a = node_of[kept_pos[i]], b = node_of[kept_pos[i + 1]];
      rc.add(rev ? (uint64_t(b) << 32) | a : (uint64_t(a) << 32) | b, 1, When{lineno, seq++});
    }
    ++used;
  }
};

// ---- an order-DEPENDENT digest of a byte stream that can be computed in pieces: the polynomial sum(c_i * b^(n-1-i)) mod 2^61 - 1
// for two bases, with the length.  digest(A + B) = digest(A) * b^|B| + digest(B), so threads digest pieces and the pieces are
// joined in order; two streams have the same digest iff (up to 2^-100) they are the same bytes in the same order.
struct Digest {
  static constexpr uint64_t P = (uint64_t(1) << 61) - 1;
  static constexpr uint64_t BASE[2] = {0x1F3D5B79A4C6E801ull % P, 0x0B1C2D3E4F506172ull % P};
  uint64_t h[2] = {0, 0};
  uint64_t n = 0;
  static uint64_t red(unsigned __int128 x) {
    uint64_t r = uint64_t(x & P) + uint64_t(x >> 61);
    r = (r & P) + (r >> 61);
    return r >= P ? r - P : r;
  }
  static uint64_t mul(uint64_t a, uint64_t b) { return red((unsigned __int128)a * b); }
  static uint64_t power(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, b = mul(b, b))
      if (e & 1) r = mul(r, b);
    return r;
  }
  void add(const char* p, size_t len) {
    uint64_t pw[2][9];
    for (int k = 0; k < 2; ++k) { pw[k][0] = 1; for (int j = 1; j <= 8; ++j) pw[k][j] = mul(pw[k][j - 1], BASE[k]); }
    size_t i = 0;
    for (; i + 8 <= len; i += 8)
      for (int k = 0; k < 2; ++k) {
        unsigned __int128 in = 0;
        for (int j = 0; j < 8; ++j) in += (unsigned __int128)(unsigned char)p[i + j] * pw[k][7 - j];
        h[k] = red((unsigned __int128)h[k] * pw[k][8] + in);
      }
    for (; i < len; ++i)
      for (int k = 0; k < 2; ++k) h[k] = red((unsigned __int128)h[k] * BASE[k] + (unsigned char)p[i]);
    n += len;
  }
  void append(const Digest& b) {
    for (int k = 0; k < 2; ++k) h[k] = red((unsigned __int128)h[k] * power(BASE[k], b.n) + b.h[k]);
    n += b.n;
  }
  std::string json() const {
    char buf[128];
    snprintf(buf, sizeof buf, "{\"bytes\": %llu, \"h1\": \"%016llx\", \"h2\": \"%016llx\"}", (unsigned long long)n, (unsigned long long)h[0],
             (unsigned long long)h[1]);
    return buf;
  }
};

void write_all(const char* p, size_t n) {
  size_t off = 0;
  while (off < n) {
    const ssize_t k = write(1, p + off, n - off);
    if (k <= 0) { perror("write"); exit(2); }
    off += size_t(k);
  }
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the threads' tables as one: counts added, the earliest appearance kept, entries in order of first appearance -- which is the
// insertion order of the single-thread run.  Keys are dealt to `shards` hash maps (by key), each merged by a thread of its own.
struct Merged {
  std::vector<Entry> e;
  std::vector<std::unordered_map<uint64_t, uint32_t>> at;       // key -> index into e (built on demand)
  static size_t shard_of(uint64_t key, size_t shards) { return shards == 1 ? 0 : size_t((key * 0x9E3779B97F4A7C15ull) >> 32) % shards; }
  uint32_t find(uint64_t key) const {
    const auto& m = at[shard_of(key, at.size())];
    const auto it = m.find(key);
    return it == m.end() ? NO_NODE : it->second;
  }
  void build(std::vector<Table*> tables, unsigned threads, bool index) {
    if (tables.size() == 1) {
      e = std::move(tables[0]->e);
      if (index) { at.resize(1); at[0] = std::move(tables[0]->at); }
      return;
    }
    std::vector<std::vector<Entry>> part(threads);
    parallel(threads, [&](unsigned s) {
      std::unordered_map<uint64_t, uint32_t> where;
      std::vector<Entry>& out = part[s];
      for (Table* t : tables)
        for (const Entry& x : t->e) {
          if (shard_of(x.key, threads) != s) continue;
          auto ins = where.emplace(x.key, uint32_t(out.size()));
          if (ins.second) out.push_back(x);
          else { Entry& y = out[ins.first->second]; y.count += x.count; if (x.first < y.first) y.first = x.first; }
        }
      std::sort(out.begin(), out.end(), [](const Entry& a, const Entry& b) { return a.first < b.first; });
    });
    for (Table* t : tables) { Table().e.swap(t->e); std::unordered_map<uint64_t, uint32_t>().swap(t->at); }
    // the sorted parts into one sorted list: pairwise merges, a level at a time
    std::vector<std::vector<Entry>> level = std::move(part);
    while (level.size() > 1) {
      std::vector<std::vector<Entry>> up((level.size() + 1) / 2);
      parallel(unsigned(up.size()), [&](unsigned k) {
        if (2 * k + 1 == level.size()) { up[k] = std::move(level[2 * k]); return; }
        up[k].resize(level[2 * k].size() + level[2 * k + 1].size());
        std::merge(level[2 * k].begin(), level[2 * k].end(), level[2 * k + 1].begin(), level[2 * k + 1].end(), up[k].begin(),
                   [](const Entry& a, const Entry& b) { return a.first < b.first; });
        std::vector<Entry>().swap(level[2 * k]);
        std::vector<Entry>().swap(level[2 * k + 1]);
      });
      level = std::move(up);
    }
    e = std::move(level[0]);
    if (e.size() >= 0x7FFFFFFFu) crash("Unsupported", "more than 2^31 keys in a table");
    if (index) {
      at.resize(threads);
      parallel(threads, [&](unsigned s) {
        size_t mine = 0;
        for (const Entry& x : e) mine += shard_of(x.key, threads) == s;
        at[s].reserve(mine);
        for (uint32_t i = 0; i < e.size(); ++i)
          if (shard_of(e[i].key, threads) == s) at[s].emplace(e[i].key, i);
      });
    }
  }
};

inline void put_int(std::string& b, long long v) {
  char tmp[24];
  int k = 0;
  const bool neg = v < 0;
  unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { tmp[k++] = char('0' + u % 10); u /= 10; } while (u);
  if (neg) b.push_back('-');
  while (k) b.push_back(tmp[--k]);
}

}  // namespace

int main(int argc, char** argv) {
  unsigned threads = 1;
  long long thr = 20;
  bool no_out = false, want_digest = false;
  const char* digest_file = nullptr;
  std::vector<const char*> pos;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = unsigned(atoi(argv[++i]));
    else if (!strcmp(argv[i], "--thr") && i + 1 < argc) thr = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--no-out")) no_out = true;
    else if (!strcmp(argv[i], "--digest")) want_digest = true;
    else if (!strcmp(argv[i], "--digest-file") && i + 1 < argc) digest_file = argv[++i];
    else pos.push_back(argv[i]);
  }
  if (digest_file && pos.empty() && threads) {
    // the digest of a file somebody else wrote (bytes as they are: no newline translation)
    const int fd = ::open(digest_file, O_RDONLY);
    if (fd < 0) { perror(digest_file); return 2; }
    struct stat st;
    fstat(fd, &st);
    const size_t n = size_t(st.st_size);
    const char* p = n ? static_cast<const char*>(mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0)) : "";
    if (p == MAP_FAILED) { perror("mmap"); return 2; }
    const size_t pieces = std::max<size_t>(threads, (n >> 26) + 1);
    std::vector<Digest> d(pieces);
    std::atomic<size_t> next{0};
    parallel(threads, [&](unsigned) {
      for (size_t c; (c = next.fetch_add(1)) < pieces;) {
        const size_t b = n / pieces * c, e = c + 1 == pieces ? n : n / pieces * (c + 1);
        d[c].add(p + b, e - b);
      }
    });
    Digest all;
    for (const Digest& x : d) all.append(x);
    printf("%s\n", all.json().c_str());
    return 0;
  }
  if (pos.size() != 2 || threads == 0) {
    fprintf(stderr, "usage: augment_oracle_cpp [--threads T] [--thr N] [--no-out] [--digest] GAF GFA\n"
                    "       augment_oracle_cpp [--threads T] --digest-file FILE\n");
    return 2;
  }
  try {
    const double t0 = now();
    Graph g;
    g.load(pos[1], threads);
    const double t1 = now();
    Mapped gaf;
    gaf.open(pos[0]);
    std::vector<uint32_t> nc(g.name.size(), 0u);
    // byte ranges cut at line starts; the line number of a range's first line comes from counting newlines
    std::vector<size_t> cut(threads + 1, gaf.n);
    cut[0] = 0;
    for (unsigned t = 1; t < threads; ++t) {
      size_t at = gaf.n * t / threads;
      if (at < cut[t - 1]) at = cut[t - 1];
      const char* nl = at < gaf.n ? static_cast<const char*>(memchr(gaf.p + at, '\n', gaf.n - at)) : nullptr;
      cut[t] = nl ? size_t(nl - gaf.p) + 1 : gaf.n;
    }
    std::vector<uint64_t> first_line(threads + 1, 0);
    std::vector<Augmenter> aug;
    aug.reserve(threads);
    for (unsigned t = 0; t < threads; ++t) aug.emplace_back(g, thr, threads == 1);
    std::vector<Crash> died(threads);
    std::vector<uint64_t> died_line(threads, UINT64_MAX);
    parallel(threads, [&](unsigned t) {
      uint64_t c = 0;
      for (size_t i = cut[t]; i < cut[t + 1]; ++i) c += gaf.p[i] == '\n';
      if (cut[t + 1] > cut[t] && gaf.p[cut[t + 1] - 1] != '\n') ++c;
      first_line[t + 1] = c;
    });
    for (unsigned t = 0; t < threads; ++t) first_line[t + 1] += first_line[t];
    const double t2 = now();
    parallel(threads, [&](unsigned t) {
      uint64_t lineno = first_line[t] + 1;
      size_t i = cut[t];
      try {
        while (i < cut[t + 1]) {
          const char* nl = static_cast<const char*>(memchr(gaf.p + i, '\n', cut[t + 1] - i));
          const size_t e = nl ? size_t(nl - gaf.p) : cut[t + 1];
          aug[t].feed(sv(gaf.p + i, e - i), lineno);
          ++lineno;
          i = e + 1;
        }
      } catch (const Crash& c) { died[t] = c; died_line[t] = lineno; }
    });
    // the reference dies at the FIRST bad line of the file
    for (unsigned t = 0; t < threads; ++t)
      if (died_line[t] != UINT64_MAX) throw died[t];
    const double t3 = now();
    // merge (T > 1): counts add up, the earliest appearance wins; then insertion order = order of first appearance
    uint64_t rejected = 0, used = 0;
    parallel(threads, [&](unsigned t) {
      for (const auto& kv : aug[t].nc) __atomic_fetch_add(&nc[kv.first], kv.second, __ATOMIC_RELAXED);
      std::unordered_map<uint32_t, uint32_t>().swap(aug[t].nc);
    });
    Merged hist, rc;
    {
      std::vector<Table*> h, r;
      for (unsigned t = 0; t < threads; ++t) { h.push_back(&aug[t].hist); r.push_back(&aug[t].rc); rejected += aug[t].rejected; used += aug[t].used; }
      hist.build(h, threads, false);
      rc.build(r, threads, true);
    }
    const double t4 = now();
    fprintf(stderr, "Rejected alignments: %llu\n", (unsigned long long)rejected);
    // ---- pass 2 [:377-427]
    // per node: its IL / OL entries in insertion order
    std::vector<uint32_t> h_ptr(g.name.size() + 1, 0);
    for (const Entry& e : hist.e) ++h_ptr[(e.key >> 33) + 1];
    for (size_t i = 0; i < g.name.size(); ++i) h_ptr[i + 1] += h_ptr[i];
    std::vector<uint32_t> h_idx(hist.e.size()), fill(h_ptr.begin(), h_ptr.end() - 1);
    for (uint32_t i = 0; i < hist.e.size(); ++i) h_idx[fill[hist.e[i].key >> 33]++] = i;
    std::vector<uint32_t>().swap(fill);
    // The text in pieces cut at line starts.  (A) per piece, in parallel: which table entry every L line names (what `weights.pop`
    // [:421] will be asked for) and whether a line makes the reference raise; (B) one pass over those lists in file order: the
    // FIRST L line of a pair takes the count, later ones get 0; (C) per piece, in parallel: the output bytes -- written in order,
    // or only digested (--digest: 23 GB of whole-genome output are compared through 40 bytes).
    constexpr uint32_t TAKES = 0x80000000u;
    const std::vector<size_t> piece = cut_lines(g.text.p, g.text.n, std::max<size_t>(size_t(threads) * 4, (g.text.n >> 25) + 1));
    const size_t C = piece.size() - 1;
    std::vector<std::vector<uint32_t>> lent(C);
    std::vector<size_t> bad_at(C, SIZE_MAX);
    std::vector<Crash> bad(C);
    std::atomic<size_t> next{0};
    parallel(threads, [&](unsigned) {
      std::vector<sv> tok;
      for (size_t c; (c = next.fetch_add(1)) < C;) {
        size_t i = piece[c];
        while (i < piece[c + 1]) {
          const char* nl = static_cast<const char*>(memchr(g.text.p + i, '\n', piece[c + 1] - i));
          const size_t e = nl ? size_t(nl - g.text.p) : piece[c + 1];
          const sv line = strip(sv(g.text.p + i, e - i));
          const size_t at = i;
          i = e + 1;
          if (line.empty()) continue;
          if (line[0] == 'S') {
            split(line, tok);
            if (tok.size() < 3) continue;
            if (g.find(tok[1]) == NO_NODE) { bad_at[c] = at; bad[c] = Crash{"KeyError", "S line behind leading whitespace: not registered by the first pass"}; break; }
          } else if (line[0] == 'L') {
            if (line.size() == 1) continue;
            split(line, tok);
            if (tok.size() < 4) { bad_at[c] = at; bad[c] = Crash{"IndexError", "L line with fewer than 4 fields"}; break; }
            const uint32_t u = g.find(tok[1]), v = g.find(tok[3]);
            lent[c].push_back(u != NO_NODE && v != NO_NODE ? rc.find((uint64_t(u) << 32) | v) : NO_NODE);
          }
        }
      }
    });
    for (size_t c = 0; c < C; ++c)
      if (bad_at[c] != SIZE_MAX) throw bad[c];
    std::vector<char> popped(rc.e.size(), 0);
    for (size_t c = 0; c < C; ++c)
      for (uint32_t& x : lent[c])
        if (x != NO_NODE) {
          if (popped[x]) x = NO_NODE;
          else { popped[x] = 1; x |= TAKES; }
        }
    Digest total;
    std::vector<std::string> buf(threads);
    std::vector<Digest> dg(threads);
    for (size_t c0 = 0; c0 < C; c0 += threads) {
      const unsigned k_n = unsigned(std::min<size_t>(threads, C - c0));
      parallel(k_n, [&](unsigned k) {
        const size_t c = c0 + k;
        std::string& b = buf[k];
        b.clear();
        b.reserve((piece[c + 1] - piece[c]) * 5 / 4 + 64);
        std::vector<sv> tok;
        size_t l_at = 0;
        size_t i = piece[c];
        while (i < piece[c + 1]) {
          const char* nl = static_cast<const char*>(memchr(g.text.p + i, '\n', piece[c + 1] - i));
          const size_t e = nl ? size_t(nl - g.text.p) : piece[c + 1];
          const sv line = strip(sv(g.text.p + i, e - i));
          i = e + 1;
          if (!line.empty() && line[0] == 'S') {
            split(line, tok);
            if (tok.size() < 3) continue;
            const uint32_t n = g.find(tok[1]);
            b.append(line);
            b.append("\tNC:i:");
            put_int(b, nc[n]);
            for (uint64_t side = 0; side < 2; ++side) {
              bool any = false;
              for (uint32_t q = h_ptr[n]; q < h_ptr[n + 1]; ++q) {
                const Entry& x = hist.e[h_idx[q]];
                if (((x.key >> 32) & 1u) != side) continue;
                b.append(any ? "," : (side == SIDE_IL ? "\tIL:Z:" : "\tOL:Z:"));
                any = true;
                put_int(b, (long long)int32_t(uint32_t(x.key)));
                b.push_back('.');
                put_int(b, (long long)x.count);
              }
            }
            b.push_back('\n');
          } else if (!line.empty() && line[0] == 'L') {
            if (line.size() == 1) continue;
            const uint32_t x = lent[c][l_at++];
            b.append(line);
            b.append("\tRC:i:");
            put_int(b, x != NO_NODE && (x & TAKES) ? (long long)rc.e[x & ~TAKES].count : 0);
            b.push_back('\n');
          } else {
            b.append(line);
            b.push_back('\n');
          }
        }
        if (want_digest) { dg[k] = Digest(); dg[k].add(b.data(), b.size()); }
      });
      for (unsigned k = 0; k < k_n; ++k) {
        if (want_digest) total.append(dg[k]);
        if (!no_out && !want_digest) write_all(buf[k].data(), buf[k].size());
        std::vector<uint32_t>().swap(lent[c0 + k]);
      }
    }
    {
      std::string& b = buf[0];
      b.clear();
      for (uint32_t q = 0; q < rc.e.size(); ++q) {                              // [:426-427]
        if (popped[q]) continue;
        b.append("L\t"); b.append(g.name[rc.e[q].key >> 32]); b.append("\t+\t"); b.append(g.name[uint32_t(rc.e[q].key)]);
        b.append("\t+\t*\tRC:i:"); put_int(b, (long long)rc.e[q].count); b.append("\tID:Z:N\n");
      }
      if (want_digest) { Digest d; d.add(b.data(), b.size()); total.append(d); }
      if (!no_out && !want_digest) write_all(b.data(), b.size());
    }
    const double t5 = now();
    fprintf(stderr, "{\"threads\": %u, \"lines\": %llu, \"used\": %llu, \"rejected\": %llu, \"load_gfa_s\": %.4f, \"index_lines_s\": %.4f, "
                    "\"hot_loop_s\": %.4f, \"merge_s\": %.4f, \"emit_s\": %.4f%s%s}\n",
            threads, (unsigned long long)first_line[threads], (unsigned long long)used, (unsigned long long)rejected, t1 - t0, t2 - t1, t3 - t2,
            t4 - t3, t5 - t4, want_digest ? ", \"digest\": " : "", want_digest ? total.json().c_str() : "");
  } catch (const Crash& c) {
    fprintf(stderr, "%s: %s\n", c.kind.c_str(), c.msg.c_str());
    return 1;
  }
  return 0;
}
