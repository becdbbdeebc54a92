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
//   augment_oracle_cpp [--threads T] [--thr N] [--no-out] GAF GFA  > augmented.gfa
//
// --threads T > 1: the GAF is cut into T byte ranges at line starts; every thread feeds its lines into tables of its
// own that also remember WHEN a key first appeared (line number, event number inside the line); the merge adds the
// counts and keeps the earliest appearance, so the output is the single-thread output.  What cannot be cut that way is
// the reference's `curr_cigar_len` leaking from one line into the next (a '~' as the first op a line reaches): with
// T > 1 such a line is an error.
// Deviations from Python's text handling, none of which the goldens or the synthetic inputs contain: integers are
// [+-]?[0-9]+, lengths count bytes.  (Universal newlines are applied when a file is mapped.)
#include <algorithm>
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

struct Graph {
  Mapped text;
  std::unordered_map<sv, uint32_t> id;      // S name -> node (first appearance)
  std::vector<sv> name;
  std::vector<long long> len;
  void load(const char* path) {                    // [:121-126]
    text.open(path);
    id.reserve(text.n / 64);          // (S and L lines of a spliced pangenome average ~75 bytes: about one segment per 150)
    std::vector<sv> tok;
    size_t i = 0;
    while (i < text.n) {
      const char* nl = static_cast<const char*>(memchr(text.p + i, '\n', text.n - i));
      const size_t e = nl ? size_t(nl - text.p) : text.n;
      const sv line(text.p + i, e - i);
      if (!line.empty() && line[0] == 'S') {
        split(strip(line), tok);
        if (tok.size() < 3) crash("IndexError", "S line with fewer than 3 fields");
        auto ins = id.emplace(tok[1], uint32_t(name.size()));
        if (ins.second) { name.push_back(tok[1]); len.push_back((long long)tok[2].size()); }
        else len[ins.first->second] = (long long)tok[2].size();
      }
      i = e + 1;
    }
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
      auto it = g.id.find(names[i]);
      if (it == g.id.end()) crash("KeyError", "line " + std::to_string(lineno) + ": node not in GFA");
      node_of[i] = it->second;
      long long r = g.len[it->second];
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

struct Out {
  std::string buf;
  bool enabled;
  explicit Out(bool e) : enabled(e) { buf.reserve(1 << 22); }
  void put(sv s) { if (!enabled) return; buf.append(s); if (buf.size() > (1 << 21)) flush(); }
  void put_int(long long v) { if (!enabled) return; buf.append(std::to_string(v)); }
  void flush() {
    size_t off = 0;
    while (off < buf.size()) {
      const ssize_t k = write(1, buf.data() + off, buf.size() - off);
      if (k <= 0) { perror("write"); exit(2); }
      off += size_t(k);
    }
    buf.clear();
  }
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char** argv) {
  unsigned threads = 1;
  long long thr = 20;
  bool no_out = false;
  std::vector<const char*> pos;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = unsigned(atoi(argv[++i]));
    else if (!strcmp(argv[i], "--thr") && i + 1 < argc) thr = atoll(argv[++i]);
    else if (!strcmp(argv[i], "--no-out")) no_out = true;
    else pos.push_back(argv[i]);
  }
  if (pos.size() != 2 || threads == 0) { fprintf(stderr, "usage: augment_oracle_cpp [--threads T] [--thr N] [--no-out] GAF GFA\n"); return 2; }
  try {
    const double t0 = now();
    Graph g;
    g.load(pos[1]);
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
    {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < threads; ++t) th.emplace_back([&, t] {
        uint64_t c = 0;
        for (size_t i = cut[t]; i < cut[t + 1]; ++i) c += gaf.p[i] == '\n';
        if (cut[t + 1] > cut[t] && gaf.p[cut[t + 1] - 1] != '\n') ++c;
        first_line[t + 1] = c;
      });
      for (auto& x : th) x.join();
      for (unsigned t = 0; t < threads; ++t) first_line[t + 1] += first_line[t];
    }
    const double t2 = now();
    {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < threads; ++t) th.emplace_back([&, t] {
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
      for (auto& x : th) x.join();
    }
    // the reference dies at the FIRST bad line of the file
    for (unsigned t = 0; t < threads; ++t)
      if (died_line[t] != UINT64_MAX) throw died[t];
    const double t3 = now();
    // merge (T > 1): counts add up, the earliest appearance wins; then insertion order = order of first appearance
    Table hist = std::move(aug[0].hist), rc = std::move(aug[0].rc);
    uint64_t rejected = aug[0].rejected, used = aug[0].used;
    for (unsigned t = 0; t < threads; ++t)
      for (const auto& kv : aug[t].nc) nc[kv.first] += kv.second;
    for (unsigned t = 1; t < threads; ++t) {
      for (const Entry& e : aug[t].hist.e) hist.add(e.key, e.count, e.first);
      for (const Entry& e : aug[t].rc.e) rc.add(e.key, e.count, e.first);
      rejected += aug[t].rejected;
      used += aug[t].used;
    }
    if (threads > 1) {
      auto by_first = [](const Entry& a, const Entry& b) { return a.first < b.first; };
      std::sort(hist.e.begin(), hist.e.end(), by_first);
      std::sort(rc.e.begin(), rc.e.end(), by_first);
      rc.at.clear();
      for (uint32_t i = 0; i < rc.e.size(); ++i) rc.at[rc.e[i].key] = i;
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
    std::vector<char> popped(rc.e.size(), 0);
    Out out(!no_out);
    std::vector<sv> tok;
    size_t i = 0;
    while (i < g.text.n) {
      const char* nl = static_cast<const char*>(memchr(g.text.p + i, '\n', g.text.n - i));
      const size_t e = nl ? size_t(nl - g.text.p) : g.text.n;
      const sv line = strip(sv(g.text.p + i, e - i));
      i = e + 1;
      if (!line.empty() && line[0] == 'S') {
        split(line, tok);
        if (tok.size() < 3) continue;
        const auto it = g.id.find(tok[1]);
        if (it == g.id.end()) crash("KeyError", "S line behind leading whitespace: not registered by the first pass");
        const uint32_t n = it->second;
        out.put(line);
        out.put("\tNC:i:");
        out.put_int(nc[n]);
        for (uint64_t side = 0; side < 2; ++side) {
          bool any = false;
          for (uint32_t q = h_ptr[n]; q < h_ptr[n + 1]; ++q) {
            const Entry& x = hist.e[h_idx[q]];
            if (((x.key >> 32) & 1u) != side) continue;
            out.put(any ? "," : (side == SIDE_IL ? "\tIL:Z:" : "\tOL:Z:"));
            any = true;
            out.put_int((long long)int32_t(uint32_t(x.key)));
            out.put(".");
            out.put_int((long long)x.count);
          }
        }
        out.put("\n");
      } else if (!line.empty() && line[0] == 'L') {
        if (line.size() == 1) continue;
        split(line, tok);
        if (tok.size() < 4) crash("IndexError", "L line with fewer than 4 fields");
        long long n = 0;
        auto u = g.id.find(tok[1]), v = g.id.find(tok[3]);
        if (u != g.id.end() && v != g.id.end()) {
          auto it = rc.at.find((uint64_t(u->second) << 32) | v->second);
          if (it != rc.at.end() && !popped[it->second]) { popped[it->second] = 1; n = (long long)rc.e[it->second].count; }
        }
        out.put(line);
        out.put("\tRC:i:");
        out.put_int(n);
        out.put("\n");
      } else {
        out.put(line);
        out.put("\n");
      }
    }
    for (uint32_t q = 0; q < rc.e.size(); ++q) {                              // [:426-427]
      if (popped[q]) continue;
      out.put("L\t"); out.put(g.name[rc.e[q].key >> 32]); out.put("\t+\t"); out.put(g.name[uint32_t(rc.e[q].key)]);
      out.put("\t+\t*\tRC:i:"); out.put_int((long long)rc.e[q].count); out.put("\tID:Z:N\n");
    }
    if (!no_out) out.flush();
    const double t5 = now();
    fprintf(stderr, "{\"threads\": %u, \"lines\": %llu, \"used\": %llu, \"rejected\": %llu, \"load_gfa_s\": %.4f, \"index_lines_s\": %.4f, "
                    "\"hot_loop_s\": %.4f, \"merge_s\": %.4f, \"emit_s\": %.4f}\n",
            threads, (unsigned long long)first_line[threads], (unsigned long long)used, (unsigned long long)rejected, t1 - t0, t2 - t1, t3 - t2,
            t4 - t3, t5 - t4);
  } catch (const Crash& c) {
    fprintf(stderr, "%s: %s\n", c.kind.c_str(), c.msg.c_str());
    return 1;
  }
  return 0;
}
