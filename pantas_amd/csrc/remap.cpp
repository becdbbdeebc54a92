// `pantas remap` in native code: scripts/remap.py main() (:29-222; the Python mirror pantas_amd/remap.py restates it line by line
// and is this file's differential partner in tests/test_quant_remap.py).  A quantification CSV of a whole-genome experiment has
// 10^7 - 10^8 rows; the script is linear in them, but interpreted.
//
//   pass 1 (:31-43)    the reference copy (<tid>_R<k>...) of the three junction names of every row names the transcripts wanted;
//   GTF    (:45-63)    their exons, sorted;
//   pass 2 (:105-222)  per row: positions of the three junctions from the exon table, annotated A3 / A5 rows shorter junction
//                      first, annotated IR rows moved to the first slot, reference / haplotype label, novel rows with a region
//                      shorter than -i dropped.
// Rows are independent: both passes run on all threads over byte ranges of the file cut at line ends, output in row order.
// The reference prints rows until it meets one it cannot handle and dies there; here the first such row (in file order, pass 1
// before the GTF before pass 2, as the script would reach them) decides the error and NOTHING is written -- the rule of every
// entry point of this library.  Errors carry the Python exception class the script raises.
#include "remap.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <algorithm>
#include <cerrno>
#include <unistd.h>

#include "common.h"
#include "quant.h"

namespace pantas {

namespace {

struct RowErr {
  uint64_t row = UINT64_MAX;      // global order key of the failing row (smaller = met earlier by the script)
  std::string kind, what;
  void take(uint64_t r, const char* k, std::string w) { if (r < row) { row = r; kind = k; what = std::move(w); } }
};
struct Fail {                      // thrown inside the per-row code, caught per chunk
  const char* kind;
  std::string what;
};

// int(s)
int64_t py_int(Span s) {
  int64_t v;
  if (!parse_py_int(py_strip(s), v)) throw Fail{"ValueError", "invalid literal for int() with base 10: '" + std::string(s.b, s.size()) + "'"};
  return v;
}
// float(s)
double py_float(Span s0) {
  double v;
  if (!parse_py_float(s0, v)) throw Fail{"ValueError", "could not convert string to float: '" + std::string(s0.b, s0.size()) + "'"};
  return v;
}

// first '|'-separated name whose last '_' piece starts with R; empty Span{nullptr} = "?"  (remap.py:21-26)
bool reference_copy(Span names, Span& out) {
  const char* p = names.b;
  for (;;) {
    const char* bar = static_cast<const char*>(memchr(p, '|', size_t(names.e - p)));
    const Span x{p, bar ? bar : names.e};
    const char* u = x.e;
    while (u > x.b && u[-1] != '_') --u;          // x.split("_")[-1] = [u, x.e)
    if (u == x.e) throw Fail{"IndexError", "string index out of range"};
    if (*u == 'R') { out = x; return true; }
    if (!bar) return false;
    p = bar + 1;
  }
}
// "_".join(name.split(".")[0].split("_")[:-1])
Span transcript_of(Span name) {
  const char* dot = static_cast<const char*>(memchr(name.b, '.', name.size()));
  const Span head{name.b, dot ? dot : name.e};
  const char* u = head.e;
  while (u > head.b && u[-1] != '_') --u;
  if (u == head.b) return Span{head.b, head.b};
  return Span{head.b, u - 1};
}

struct Lines {                      // the text as Python's text mode hands it out
  const char* b = nullptr;
  const char* e = nullptr;
  RawVector<char> own;             // when the file holds a '\r': "\r\n" and a lone '\r' become '\n'
  void open(const MappedFile& f) {
    b = f.data; e = f.data + f.size;
    if (f.size && memchr(f.data, '\r', f.size)) {
      own.resize(f.size);
      size_t n = 0;
      for (size_t i = 0; i < f.size; ++i) {
        if (f.data[i] == '\r') { own[n++] = '\n'; if (i + 1 < f.size && f.data[i + 1] == '\n') ++i; }
        else own[n++] = f.data[i];
      }
      b = own.data(); e = b + n;
    }
  }
};

// fields of a data row: line.strip("\n").split(",")
uint32_t split_commas(Span line, Span* out, uint32_t cap) {
  while (line.e > line.b && line.e[-1] == '\n') --line.e;
  while (line.b < line.e && *line.b == '\n') ++line.b;
  uint32_t n = 0;
  const char* p = line.b;
  for (;;) {
    const char* c = static_cast<const char*>(memchr(p, ',', size_t(line.e - p)));
    if (n < cap) out[n] = Span{p, c ? c : line.e};
    ++n;
    if (!c) break;
    p = c + 1;
  }
  return n;
}

using Exons = std::vector<std::pair<int64_t, int64_t>>;
using ExonTable = std::unordered_map<std::string_view, Exons>;

struct Pos {                        // a position string "chrom:s-e[?]" or "?"
  bool known = false;
  std::string text;
};

void position(bool has_name, Span name, Span chrom, const ExonTable& exons_of, Pos& out) {
  out.known = false;
  out.text.clear();
  if (!has_name) return;
  const char* dot = static_cast<const char*>(memchr(name.b, '.', name.size()));
  const Span head{name.b, dot ? dot : name.e};
  const char* u = head.e;
  while (u > head.b && u[-1] != '_') --u;
  if (u == head.e) throw Fail{"IndexError", "string index out of range"};
  const bool imprecise = *u == 'H';
  const Span tid = transcript_of(name);
  auto it = exons_of.find(tid.sv());
  if (it == exons_of.end()) throw Fail{"KeyError", "'" + std::string(tid.b, tid.size()) + "'"};
  const Exons& exons = it->second;
  int64_t idx[3];
  uint32_t n_idx = 0;
  if (dot) {
    const char* p = dot + 1;
    for (;;) {
      const char* d = static_cast<const char*>(memchr(p, '.', size_t(name.e - p)));
      const int64_t v = py_int(Span{p, d ? d : name.e});
      if (n_idx < 3) idx[n_idx] = v;
      ++n_idx;
      if (!d) break;
      p = d + 1;
    }
  }
  if (n_idx > 2) throw Fail{"AssertionError", ""};
  if (n_idx == 0) return;
  auto at = [&](int64_t i) -> const std::pair<int64_t, int64_t>& {      // Python indexing: negative counts from the end
    const int64_t n = int64_t(exons.size());
    if (i < 0) i += n;
    if (i < 0 || i >= n) throw Fail{"IndexError", "list index out of range"};
    return exons[size_t(i)];
  };
  int64_t s, e;
  if (n_idx == 1) { const auto& x = at(idx[0] - 1); s = x.first; e = x.second; }
  else { s = at(idx[0] - 1).second + 1; e = at(idx[1] - 1).first - 1; }
  if (!(s <= e)) throw Fail{"AssertionError", ""};
  out.known = true;
  out.text.assign(chrom.b, chrom.size());
  out.text += ':';
  out.text += std::to_string(s);
  out.text += '-';
  out.text += std::to_string(e);
  if (imprecise) out.text += '?';
}

// (start, end) of "chrom:start-end": region.split(":")[1].split("-"), two ints
void two_ints(Span region, int64_t& s, int64_t& e) {
  const char* c1 = static_cast<const char*>(memchr(region.b, ':', region.size()));
  if (!c1) throw Fail{"IndexError", "list index out of range"};
  const char* c2 = static_cast<const char*>(memchr(c1 + 1, ':', size_t(region.e - c1 - 1)));
  const Span mid{c1 + 1, c2 ? c2 : region.e};
  const char* d1 = static_cast<const char*>(memchr(mid.b, '-', mid.size()));
  if (!d1) throw Fail{"ValueError", "not enough values to unpack (expected 2, got 1)"};
  if (memchr(d1 + 1, '-', size_t(mid.e - d1 - 1))) throw Fail{"ValueError", "too many values to unpack (expected 2)"};
  s = py_int(Span{mid.b, d1});
  e = py_int(Span{d1 + 1, mid.e});
}
void span_of(const std::string& pos, int64_t& s, int64_t& e) {        // remap.py:4-12: a trailing '?' is dropped first
  Span r{pos.data(), pos.data() + pos.size()};
  if (r.size() && r.e[-1] == '?') --r.e;
  two_ints(r, s, e);
}

// one row of pass 2 appended to `out` (nothing when the row is dropped)
void remap_row(Span line, const ExonTable& exons_of, int64_t min_intron, std::string& out) {
  Span f[17];
  const uint32_t n = split_commas(line, f, 17);
  if (n != 16) throw Fail{"ValueError", n < 16 ? "not enough values to unpack (expected 16, got " + std::to_string(n) + ")" : "too many values to unpack (expected 16)"};
  const Span etype = f[0], annotation = f[1], chrom = f[2], gene = f[3], strand = f[4];
  Span name[3];
  bool has[3];
  Pos pos[3];
  Span nodes[3] = {f[8], f[9], f[10]};
  for (int k = 0; k < 3; ++k) has[k] = reference_copy(f[5 + k], name[k]);
  for (int k = 0; k < 3; ++k) position(has[k], name[k], chrom, exons_of, pos[k]);
  Span w1 = f[11], w2 = f[12], psi1 = f[13], psi2 = f[14];
  const Span dpsi_text = f[15];
  bool swapped = false;
  double dpsi = 0;
  static const char DOT[] = ".";
  auto swap_conditions = [&] {
    std::swap(w1, w2);
    std::swap(psi1, psi2);
    dpsi = -(swapped ? dpsi : py_float(dpsi_text));
    swapped = true;
  };
  auto is = [](Span s, const char* lit) { return s.size() == strlen(lit) && memcmp(s.b, lit, s.size()) == 0; };
  const bool annotated = is(annotation, "annotated");
  const bool a_type = etype.size() > 0 ? etype.b[0] == 'A' : false;
  if (annotated) {
    if (etype.size() == 0) throw Fail{"IndexError", "string index out of range"};      // etype[0]
    if (a_type) {
      if (!pos[0].known) {
        // the second junction moves to the first slot -- except its position, which the script overwrites with the slot it has just
        // cleared (remap.py:150-156): both end up '?'
        name[0] = name[1]; has[0] = has[1]; has[1] = false;
        pos[0].known = false; pos[1].known = false;
        nodes[0] = nodes[1]; nodes[1] = Span{DOT, DOT + 1};
        swap_conditions();
      } else if (pos[1].known) {
        int64_t a0, a1, b0, b1;
        span_of(pos[0].text, a0, a1);
        span_of(pos[1].text, b0, b1);
        if (a1 - a0 > b1 - b0) {          // shorter junction first
          std::swap(name[0], name[1]); std::swap(has[0], has[1]);
          std::swap(pos[0], pos[1]);
          std::swap(nodes[0], nodes[1]);
          swap_conditions();
        }
      }
    } else if (is(etype, "IR")) {
      name[0] = name[1]; has[0] = has[1]; has[1] = false;
      pos[0] = pos[1]; pos[1].known = false;
      nodes[0] = nodes[1]; nodes[1] = Span{DOT, DOT + 1};
      swap_conditions();
    }
  }
  const bool u0 = !pos[0].known, u1 = !pos[1].known, u2 = !pos[2].known;
  const bool es = is(etype, "ES"), ir = is(etype, "IR");
  bool hap;
  if (annotated) {
    if (!es && etype.size() == 0) throw Fail{"IndexError", "string index out of range"};
    hap = (es && (u0 || u1 || u2)) || (a_type && (u0 || u1)) || (ir && u0);
  } else {
    if (!(es && u0 && u1 && u2) && etype.size() == 0) throw Fail{"IndexError", "string index out of range"};
    hap = (es && u0 && u1 && u2) || (a_type && u0 && u1) || (ir && u0 && u1);
  }
  if (is(annotation, "novel")) {
    // any([length(p) < min for p in pos if p != "?"]): every known position is measured (a trailing '?' is no int: ValueError, :15-18)
    bool drop = false;
    for (int k = 0; k < 3; ++k) {
      if (!pos[k].known) continue;
      int64_t s, e;
      two_ints(Span{pos[k].text.data(), pos[k].text.data() + pos[k].text.size()}, s, e);
      if (e - s < min_intron) drop = true;
    }
    if (drop) return;
  }
  auto put = [&](Span s) { out.append(s.b, s.size()); out += ','; };
  put(etype); put(annotation);
  out += hap ? "haplotype," : "reference,";
  put(chrom); put(gene); put(strand);
  for (int k = 0; k < 3; ++k) { if (has[k]) put(name[k]); else out += "?,"; }
  for (int k = 0; k < 3; ++k) put(nodes[k]);
  for (int k = 0; k < 3; ++k) { if (pos[k].known) { out += pos[k].text; out += ','; } else out += "?,"; }
  put(w1); put(w2); put(psi1); put(psi2);
  if (swapped) py_float_repr(out, dpsi); else out.append(dpsi_text.b, dpsi_text.size());
  out += '\n';
}

const char HEADER[] =
    "etype,annotation_type,haplotype_type,chrom,gene,strand,junction1_name,junction2_name,junction3_name,"
    "junction1_nodes,junction2_nodes,junction3_nodes,junction1_positions,junction2_positions,junction3_positions,"
    "W1,W2,psi_c1,psi_c2,dpsi\n";

// byte ranges of [b, e) cut behind '\n'
std::vector<std::pair<const char*, const char*>> cut_lines(const char* b, const char* e, unsigned want) {
  std::vector<std::pair<const char*, const char*>> out;
  const size_t size = size_t(e - b);
  const char* at = b;
  for (unsigned c = 1; c <= want && at < e; ++c) {
    const char* stop = c == want ? e : b + size * c / want;
    if (stop < at) stop = at;
    if (stop < e) {
      const char* nl = static_cast<const char*>(memchr(stop, '\n', size_t(e - stop)));
      stop = nl ? nl + 1 : e;
    }
    if (stop > at) out.emplace_back(at, stop);
    at = stop;
  }
  return out;
}
template <class Fn>
void each_line(const char* b, const char* e, Fn fn) {
  while (b < e) {
    const char* nl = static_cast<const char*>(memchr(b, '\n', size_t(e - b)));
    const char* stop = nl ? nl + 1 : e;
    fn(Span{b, stop});
    b = stop;
  }
}
inline bool starts(Span s, const char* lit) { const size_t n = strlen(lit); return s.size() >= n && memcmp(s.b, lit, n) == 0; }

void write_all(int fd, const char* p, size_t n) {
  while (n) {
    const ssize_t w = ::write(fd, p, n);
    if (w < 0) { if (errno == EINTR) continue; throw Error("OSError", std::string("write failed: ") + strerror(errno)); }
    p += w; n -= size_t(w);
  }
}

}  // namespace

void remap_run(const std::string& csv_path, const std::string& gtf_path, int64_t min_intron, int fd_out, unsigned threads) {
  if (threads == 0) threads = hw_threads();
  PhaseTimer pt("remap");
  MappedFile csv_f, gtf_f;
  csv_f.open(csv_path.c_str());
  gtf_f.open(gtf_path.c_str());
  Lines csv, gtf;
  csv.open(csv_f);
  gtf.open(gtf_f);
  const unsigned C = csv_f.size < (size_t(1) << 20) ? 1u : threads;
  const auto chunks = cut_lines(csv.b, csv.e, C);
  std::mutex mu;
  RowErr err;
  std::atomic<uint64_t> first_bad{UINT64_MAX};      // = err.row, readable without the lock
  // ---- pass 1: the transcripts the rows name.  Row order key = byte offset of the row
  ExonTable wanted;
  std::vector<std::vector<std::string_view>> local(chunks.size());
  parallel_chunks(chunks.size(), unsigned(std::max<size_t>(1, chunks.size())), [&](unsigned, size_t cb, size_t ce) {
    for (size_t c = cb; c < ce; ++c) {
      std::unordered_map<std::string_view, char> seen;
      each_line(chunks[c].first, chunks[c].second, [&](Span line) {
        if (starts(line, "etype")) return;
        const uint64_t key = uint64_t(line.b - csv.b);
        if (first_bad.load(std::memory_order_relaxed) < key) return;          // (a row in front of this one has failed already)
        try {
          Span f[9];
          const uint32_t n = split_commas(line, f, 9);
          if (n < 8) throw Fail{"ValueError", "not enough values to unpack (expected 3, got " + std::to_string(n > 5 ? n - 5 : 0) + ")"};
          for (int k = 0; k < 3; ++k) {
            Span t;
            if (reference_copy(f[5 + k], t)) seen.emplace(transcript_of(t).sv(), 0);
          }
        } catch (const Fail& x) {
          std::lock_guard<std::mutex> lk(mu);
          err.take(key, x.kind, x.what);
          first_bad.store(err.row, std::memory_order_relaxed);
        }
      });
      local[c].reserve(seen.size());
      for (auto& kv : seen) local[c].push_back(kv.first);
    }
  });
  if (err.row != UINT64_MAX) throw Error(err.kind, "quantification CSV byte " + std::to_string(err.row) + ": " + err.what);
  for (auto& v : local) for (std::string_view s : v) wanted.emplace(s, Exons{});
  pt.lap("transcripts named by the rows");
  // ---- GTF: exons of those transcripts (sequential in file order: the first failing line decides, and the file is small)
  each_line(gtf.b, gtf.e, [&](Span line) {
    if (err.row != UINT64_MAX) return;
    if (starts(line, "#")) return;
    const uint64_t key = uint64_t(line.b - gtf.b);
    try {
      Span l = line;
      while (l.e > l.b && l.e[-1] == '\n') --l.e;
      while (l.b < l.e && *l.b == '\n') ++l.b;
      std::vector<Span> cols;
      const char* p = l.b;
      for (;;) {
        const char* t = static_cast<const char*>(memchr(p, '\t', size_t(l.e - p)));
        cols.push_back(Span{p, t ? t : l.e});
        if (!t) break;
        p = t + 1;
      }
      if (cols.size() < 3) throw Fail{"IndexError", "list index out of range"};
      if (!(cols[2].size() == 4 && memcmp(cols[2].b, "exon", 4) == 0)) return;
      // re.search('transcript_id "[A-Za-z0-9_]+";', cols[-1])
      const Span last = cols.back();
      static const char KEY[] = "transcript_id \"";
      const size_t KL = sizeof KEY - 1;
      Span tid{nullptr, nullptr};
      for (const char* q = last.b; q + KL <= last.e; ++q) {
        if (memcmp(q, KEY, KL) != 0) continue;
        const char* r = q + KL;
        while (r < last.e && (isalnum(static_cast<unsigned char>(*r)) || *r == '_') && static_cast<unsigned char>(*r) < 128) ++r;
        if (r > q + KL && r + 1 < last.e && r[0] == '"' && r[1] == ';') { tid = Span{q + KL, r}; break; }
      }
      if (!tid.b) throw Fail{"AttributeError", "'NoneType' object has no attribute 'group'"};
      auto it = wanted.find(tid.sv());
      if (it == wanted.end()) return;
      if (cols.size() < 4) throw Fail{"IndexError", "list index out of range"};
      const int64_t s = py_int(cols[3]);
      if (cols.size() < 5) throw Fail{"IndexError", "list index out of range"};
      const int64_t e = py_int(cols[4]);
      it->second.emplace_back(s, e);
    } catch (const Fail& x) {
      err.take(key, x.kind, x.what);
    }
  });
  if (err.row != UINT64_MAX) throw Error(err.kind, "GTF byte " + std::to_string(err.row) + ": " + err.what);
  for (auto& kv : wanted) std::sort(kv.second.begin(), kv.second.end());
  pt.lap("exon table");
  // ---- pass 2, streamed: pieces of ~32 MB of rows, a wave of `threads` pieces at a time, each wave written in row order and dropped
  // (the whole output of a whole-genome quantification is tens of GB).  Pass 1 and the GTF have succeeded, so from here
  // on the output is the reference's also when a row fails: the header and every row in front of the failing one are out, as
  // remap.py has printed them when it dies (:131-195 print row by row).
  const size_t PIECE = size_t(32) << 20;
  const unsigned P = unsigned(std::max<size_t>(C, std::min<size_t>(1u << 20, size_t(csv.e - csv.b) / PIECE + 1)));
  const auto pieces = P == C ? chunks : cut_lines(csv.b, csv.e, P);
  if (fd_out >= 0) write_all(fd_out, HEADER, sizeof HEADER - 1);
  double t_rows = 0, t_write = 0;
  for (size_t w0 = 0; w0 < pieces.size(); w0 += threads) {
    const size_t w1 = std::min(pieces.size(), w0 + threads);
    std::vector<std::string> outs(w1 - w0);
    const double ta = now_s();
    parallel_chunks(w1 - w0, unsigned(w1 - w0), [&](unsigned, size_t cb, size_t ce) {
      for (size_t c = cb; c < ce; ++c) {
        std::string& out = outs[c];
        const auto& pc = pieces[w0 + c];
        out.reserve(size_t(pc.second - pc.first) + (size_t(pc.second - pc.first) >> 2));
        each_line(pc.first, pc.second, [&](Span line) {
          if (starts(line, "etype")) return;
          const uint64_t key = uint64_t(line.b - csv.b);
          if (first_bad.load(std::memory_order_relaxed) < key) return;
          try {
            remap_row(line, wanted, min_intron, out);
          } catch (const Fail& x) {
            std::lock_guard<std::mutex> lk(mu);
            err.take(key, x.kind, x.what);
            first_bad.store(err.row, std::memory_order_relaxed);
          }
        });
      }
    });
    const double tb = now_s();
    t_rows += tb - ta;
    const bool bad = err.row != UINT64_MAX;
    for (size_t c = 0; c < outs.size(); ++c) {
      const auto& pc = pieces[w0 + c];
      if (bad && uint64_t(pc.first - csv.b) > err.row) break;          // pieces behind the failing row: not printed
      std::string& o = outs[c];
      // (the piece of the failing row: a piece is one thread's, walked in row order, so `o` ends in front of that row)
      if (fd_out >= 0) write_all(fd_out, o.data(), o.size());
    }
    t_write += now_s() - tb;
    if (bad) throw Error(err.kind, "quantification CSV byte " + std::to_string(err.row) + ": " + err.what);
  }
  {
    char buf[96];
    snprintf(buf, sizeof buf, "rows %.3f s + writing %.3f s, in waves", t_rows, t_write);
    pt.lap(buf);
  }
}

}  // namespace pantas
