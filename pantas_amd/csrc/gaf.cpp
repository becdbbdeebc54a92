#include "gaf.h"

#include <cstdlib>
#include <memory>

namespace pantas {

namespace {

inline bool is_op_char(char c) { return c == '=' || c == ':' || c == '*' || c == '+' || c == '-' || c == '~'; }

inline uint32_t op_code(char c) {
  switch (c) {
    case ':': return OP_COLON;
    case '*': return OP_STAR;
    case '+': return OP_PLUS;
    case '-': return OP_MINUS;
    case '=': return OP_EQUAL;
    default: return OP_TILDE;
  }
}

struct Chunk {
  GafBatch b;
  std::unique_ptr<Error> err;
  uint64_t err_line = 0;
  size_t keep_hdr = 0, keep_nodes = 0, keep_ops = 0;
  uint64_t keep_del_lines = 0, keep_del_nodes = 0;
  // drop whatever the raising line had appended
  void rollback() {
    b.hdr.resize(keep_hdr);
    b.nodes.resize(keep_nodes);
    b.ops.resize(keep_ops);
    b.node_off.resize(keep_hdr + 1);
    b.op_off.resize(keep_hdr + 1);
    b.del_lines = keep_del_lines;
    b.del_nodes = keep_del_nodes;
  }
};

[[noreturn]] void fail(const char* kind, uint64_t local_line, const std::string& what) {
  // local_line is patched to the global number by the caller through the message prefix
  throw Error(kind, "GAF line @" + std::to_string(local_line) + "@: " + what);
}

// cs ops of `text` (already without "cs:Z:") appended to ops; returns count.  (:10-37)
uint32_t parse_cs(const char* b, const char* e, std::vector<uint32_t>& ops, uint64_t line) {
  uint32_t n = 0;
  const char* p = b;
  while (p < e && !is_op_char(*p)) ++p;   // text before the first op char opens no entry
  while (p < e) {
    const char op = *p++;
    const char* r = p;
    while (p < e && !is_op_char(*p)) ++p;
    uint64_t val;
    if (op == '*') {
      val = 1;
    } else {
      bool digits = p > r;
      for (const char* q = r; q < p && digits; ++q) digits = (*q >= '0' && *q <= '9');
      if (digits) {
        val = 0;
        for (const char* q = r; q < p; ++q) {
          val = val * 10 + uint64_t(*q - '0');
          if (val > OP_MAX_LEN) fail("Unsupported", line, "cs op longer than 2^29-1");
        }
      } else {
        val = uint64_t(p - r);
        if (val > OP_MAX_LEN) fail("Unsupported", line, "cs op longer than 2^29-1");
      }
    }
    ops.push_back(uint32_t(val << 3) | op_code(op));
    ++n;
  }
  return n;
}

void tokenize_range(const Graph& g, const char* data, size_t size, int thr, Chunk& ck) {
  GafBatch& out = ck.b;
  out.node_off.push_back(0);
  out.op_off.push_back(0);
  std::string cs_buf;
  std::vector<Span> tok;
  tok.reserve(32);
  const char* p = data;
  const char* end = data + size;
  uint64_t line = 0;   // local line index
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
    const char* le = nl ? nl : end;
    const char* const p_line = p;
    ck.err_line = line;
    tok.clear();
    {
      Tokens tk(Span{p, le});
      Span t;
      while (tk.next(t)) tok.push_back(t);
    }
    ++out.lines_total;
    p = nl ? nl + 1 : end;
    const uint64_t this_line = line++;
    // state to fall back to if this line raises
    ck.keep_hdr = out.hdr.size(); ck.keep_nodes = out.nodes.size(); ck.keep_ops = out.ops.size();
    ck.keep_del_lines = out.del_lines; ck.keep_del_nodes = out.del_nodes;
    if (lone_cr(p_line, le)) fail("Unsupported", this_line, "lone carriage return (Python's universal newlines end a line there; this engine splits at \\n only)");
    if (tok.size() < 12) fail("IndexError", this_line, "fewer than 12 columns");
    int64_t mapq;
    if (!parse_py_int(tok[11], mapq, true)) fail("ValueError", this_line, "mapq is not an integer");
    if (mapq < thr) { ++out.rejected; continue; }                                  // :143-146
    if (tok[5].size() == 1 && *tok[5].b == '*') { ++out.skipped; continue; }       // :147-148
    int64_t path_len, start, endpos;
    if (!parse_py_int(tok[6], path_len, true) || !parse_py_int(tok[7], start, true) || !parse_py_int(tok[8], endpos, true))
      fail("ValueError", this_line, "path length / start / end is not an integer");
    // ---- cs tag: first "cs:" in any optional field, up to the end of that field (:154-160)
    const size_t ops_before = out.ops.size();
    bool found_cs = false;
    for (size_t i = 12; i < tok.size() && !found_cs; ++i) {
      const char* b = tok[i].b;
      const char* e = tok[i].e;
      for (const char* q = b; q + 3 <= e; ++q) {
        if (q[0] == 'c' && q[1] == 's' && q[2] == ':') {
          // drop every "cs:Z:" inside the matched text
          cs_buf.clear();
          const char* s = q;
          while (s < e) {
            if (e - s >= 5 && memcmp(s, "cs:Z:", 5) == 0) { s += 5; continue; }
            cs_buf.push_back(*s++);
          }
          parse_cs(cs_buf.data(), cs_buf.data() + cs_buf.size(), out.ops, this_line);
          found_cs = true;
          break;
        }
      }
    }
    if (!found_cs) out.ops.push_back((1u << 3) | OP_STAR);   // cigar "*" -> [('*', 1)]
    // ---- two-op clipping (:40-50, :164-167)
    if (out.ops.size() - ops_before == 2) {
      uint32_t a = out.ops[ops_before], b2 = out.ops[ops_before + 1];
      if ((a & 7u) == OP_PLUS && (b2 & 7u) == OP_COLON) {
        start += int64_t(a >> 3);
        out.ops[ops_before] = b2;
        out.ops.pop_back();
      } else if ((a & 7u) == OP_COLON && (b2 & 7u) == OP_PLUS) {
        out.ops.pop_back();
      }
    }
    // ---- dv tag: first "dv:f:" followed by a digit (:172-180)
    bool found_dv = false;
    double dv = 0.0;
    for (size_t i = 12; i < tok.size() && !found_dv; ++i) {
      const char* b = tok[i].b;
      const char* e = tok[i].e;
      for (const char* q = b; q + 6 <= e; ++q) {
        if (memcmp(q, "dv:f:", 5) == 0 && q[5] >= '0' && q[5] <= '9') {
          const char* s = q + 5;
          const char* t = s;
          while (t < e && *t >= '0' && *t <= '9') ++t;
          if (t + 1 < e && *t == '.' && t[1] >= '0' && t[1] <= '9') {
            ++t;
            while (t < e && *t >= '0' && *t <= '9') ++t;
          }
          std::string num(s, t);
          dv = strtod(num.c_str(), nullptr);
          found_dv = true;
          break;
        }
      }
    }
    if (!found_dv) fail("ValueError", this_line, "could not convert string to float: '*' (no dv:f: tag)");
    if (dv > 0.1) { out.ops.resize(ops_before); ++out.skipped; continue; }
    // ---- path (:185-197)
    const size_t nodes_before = out.nodes.size();
    {
      const char* b = tok[5].b;
      const char* e = tok[5].e;
      const bool rev = *b != '>';
      const char sep = rev ? '<' : '>';
      const char* q = static_cast<const char*>(memchr(b, sep, size_t(e - b)));
      if (!q) fail("AssertionError", this_line, "path holds no node");
      while (q < e) {
        const char* s = q + 1;
        const char* t = static_cast<const char*>(memchr(s, sep, size_t(e - s)));
        if (!t) t = e;
        uint32_t idx = g.names.find(Span{s, t});
        if (idx == NONE32)
          fail("KeyError", this_line, "path node '" + std::string(s, t) + "' has no S line");
        if (out.nodes.size() == nodes_before || out.nodes.back() != idx) out.nodes.push_back(idx);
        q = t;
      }
      const int64_t tail = path_len - endpos;
      if (start < INT32_MIN || start > INT32_MAX || tail < INT32_MIN || tail > INT32_MAX)
        fail("Unsupported", this_line, "start / end offsets beyond 32 bits");
      uint32_t flags = rev ? LINE_REV : 0u;
      bool has_del = false;
      uint32_t stars = 0, indels = 0, others = 0;
      uint64_t total = 0;
      for (size_t i = ops_before; i < out.ops.size(); ++i) {
        const uint32_t code = out.ops[i] & 7u;
        has_del |= code == OP_MINUS;
        if ((out.ops[i] >> 3) == 0 || code == OP_TILDE) flags |= LINE_SLOW;
        if (code == OP_STAR) ++stars;
        else if (code == OP_PLUS || code == OP_MINUS) ++indels;
        else if (code != OP_COLON) ++others;
        total += out.ops[i] >> 3;
      }
      flags |= line_class_flags(uint32_t(out.ops.size() - ops_before), stars, indels, others, (flags & LINE_SLOW) != 0, total, start, tail);
      if (has_del) {
        flags |= LINE_HAS_DEL;
        ++out.del_lines;
        out.del_nodes += out.nodes.size() - nodes_before;
      }
      out.hdr.push_back(LineHdr{int32_t(start), int32_t(tail), uint32_t(this_line), flags});
    }
    if (out.nodes.size() >= NONE32 || out.ops.size() >= NONE32)
      fail("Unsupported", this_line, "more than 2^32-2 path nodes or cs ops in one batch");
    out.node_off.push_back(uint32_t(out.nodes.size()));
    out.op_off.push_back(uint32_t(out.ops.size()));
  }
}

}  // namespace

void tokenize_gaf(const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id,
                  unsigned threads, GafBatch& out) {
  PhaseTimer pt("tokenize_gaf");
  if (threads == 0) threads = hw_threads();
  if (size < (1u << 20)) threads = 1;
  auto ranges = split_at_newlines(data, size, threads);
  std::vector<Chunk> cks(ranges.size());
  parallel_chunks(ranges.size(), unsigned(ranges.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      try {
        tokenize_range(g, data + ranges[i].first, ranges[i].second - ranges[i].first, thr, cks[i]);
      } catch (const Error& er) {
        cks[i].err = std::make_unique<Error>(er);
        cks[i].rollback();
      }
    }
  });
  pt.lap("parse lines (parallel)");
  // the reference stops at the first offending line: keep the records before it, remember the error
  uint64_t line_base = first_line_id;
  size_t used = cks.size();
  for (size_t ci = 0; ci < cks.size(); ++ci) {
    Chunk& c = cks[ci];
    if (c.err) {
      std::string msg = c.err->what();
      // patch "@local@" into the 1-based global line number
      size_t a = msg.find('@'), b = msg.find('@', a + 1);
      if (a != std::string::npos && b != std::string::npos) {
        uint64_t local = strtoull(msg.substr(a + 1, b - a - 1).c_str(), nullptr, 10);
        msg = msg.substr(0, a) + std::to_string(line_base + local + 1) + msg.substr(b + 1);
      }
      size_t colon = msg.find(": ");
      out.deferred = std::make_shared<Error>(c.err->kind, colon == std::string::npos ? msg : msg.substr(colon + 2));
      used = ci + 1;
      break;
    }
    line_base += c.b.lines_total;
  }
  cks.resize(used);
  size_t n = 0, nn = 0, no = 0;
  for (auto& c : cks) { n += c.b.hdr.size(); nn += c.b.nodes.size(); no += c.b.ops.size(); }
  if (nn >= NONE32 || no >= NONE32) throw Error("Unsupported", "more than 2^32-2 path nodes or cs ops in one batch");
  out.hdr.resize(n);
  out.node_off.resize(n + 1);
  out.op_off.resize(n + 1);
  out.nodes.resize(nn);
  out.ops.resize(no);
  out.bytes = size;
  std::vector<size_t> hb(cks.size() + 1, 0), nb(cks.size() + 1, 0), ob(cks.size() + 1, 0);
  std::vector<uint64_t> lb(cks.size() + 1, first_line_id);
  for (size_t i = 0; i < cks.size(); ++i) {
    hb[i + 1] = hb[i] + cks[i].b.hdr.size();
    nb[i + 1] = nb[i] + cks[i].b.nodes.size();
    ob[i + 1] = ob[i] + cks[i].b.ops.size();
    lb[i + 1] = lb[i] + cks[i].b.lines_total;
    out.lines_total += cks[i].b.lines_total;
    out.rejected += cks[i].b.rejected;
    out.skipped += cks[i].b.skipped;
    out.del_lines += cks[i].b.del_lines;
    out.del_nodes += cks[i].b.del_nodes;
  }
  if (lb.back() > uint64_t(NONE32)) throw Error("Unsupported", "more than 2^32-1 GAF lines");
  parallel_chunks(cks.size(), unsigned(cks.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      const GafBatch& s = cks[i].b;
      for (size_t k = 0; k < s.hdr.size(); ++k) {
        LineHdr h = s.hdr[k];
        h.line_id = uint32_t(lb[i] + h.line_id);
        out.hdr[hb[i] + k] = h;
        out.node_off[hb[i] + k] = uint32_t(nb[i] + s.node_off[k]);
        out.op_off[hb[i] + k] = uint32_t(ob[i] + s.op_off[k]);
      }
      if (!s.nodes.empty()) memcpy(&out.nodes[nb[i]], s.nodes.data(), s.nodes.size() * 4);
      if (!s.ops.empty()) memcpy(&out.ops[ob[i]], s.ops.data(), s.ops.size() * 4);
    }
  });
  out.node_off[n] = uint32_t(nn);
  out.op_off[n] = uint32_t(no);
  if (n) out.first_record_line = out.hdr[0].line_id;     // records are in file order here (bucket_by_locus comes later)
  pt.lap("merge chunks");
}


void bucket_by_locus(GafBatch& b, unsigned threads) {
  PhaseTimer pt("bucket_by_locus");
  const size_t n = b.size();
  if (n < 2) {
    b.n_short = (n == 1 && short_walk_class(b.hdr[0].flags, b.node_off[1] - b.node_off[0])) ? 1 : 0;
    return;
  }
  if (threads == 0) threads = hw_threads();
  std::vector<uint32_t> key(n), idx(n);
  parallel_chunks(n, n < (1u << 16) ? 1 : threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      // lowest end of the path (reverse-strand reads list their nodes downwards)
      key[i] = b.node_off[i + 1] > b.node_off[i] ? std::min(b.nodes[b.node_off[i]], b.nodes[b.node_off[i + 1] - 1]) : 0u;
      idx[i] = uint32_t(i);
    }
  });
  pt.lap("keys");
  radix_sort_pairs(key, idx, threads);
  pt.lap("radix sort");
  // the two classes apart (stable), then inside every block of PARTITION_BLOCK records of a class by path length:
  // the lanes of a wave then loop alike
  size_t n_short = 0;
  {
    std::vector<uint32_t> part(n);
    for (size_t i = 0; i < n; ++i) {
      const uint32_t s = idx[i];
      n_short += short_walk_class(b.hdr[s].flags, b.node_off[s + 1] - b.node_off[s]) ? 1u : 0u;
    }
    size_t a = 0, g = n_short;
    for (size_t i = 0; i < n; ++i) {
      const uint32_t s = idx[i];
      if (short_walk_class(b.hdr[s].flags, b.node_off[s + 1] - b.node_off[s])) part[a++] = s; else part[g++] = s;
    }
    idx.swap(part);
  }
  b.n_short = n_short;
  {
    constexpr size_t PARTITION_BLOCK = 2048;
    // blocks are counted from the start of each class
    const size_t nb0 = (n_short + PARTITION_BLOCK - 1) / PARTITION_BLOCK, nb1 = (n - n_short + PARTITION_BLOCK - 1) / PARTITION_BLOCK;
    const size_t nb = nb0 + nb1;
    parallel_chunks(nb, nb < 64 ? 1 : threads, [&](unsigned, size_t lo, size_t hi) {
      std::vector<std::pair<uint32_t, uint32_t>> tmp;   // (length rank, record)
      for (size_t blk = lo; blk < hi; ++blk) {
        const size_t a = blk < nb0 ? blk * PARTITION_BLOCK : n_short + (blk - nb0) * PARTITION_BLOCK;
        const size_t e = std::min(blk < nb0 ? n_short : n, a + PARTITION_BLOCK);
        tmp.clear();
        for (size_t i = a; i < e; ++i) {
          const uint32_t s = idx[i];
          tmp.emplace_back(block_order_key(b.hdr[s].flags, b.node_off[s + 1] - b.node_off[s]), s);
        }
        std::stable_sort(tmp.begin(), tmp.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (size_t i = a; i < e; ++i) idx[i] = tmp[i - a].second;
      }
    });
  }
  pt.lap("order blocks");
  std::vector<LineHdr> hdr(n);
  std::vector<uint32_t> node_off(n + 1), op_off(n + 1), nodes(b.nodes.size()), ops(b.ops.size());
  // sizes, then exclusive offsets (sequential scan: 2 adds per record), then a parallel gather
  node_off[0] = 0;
  op_off[0] = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t s = idx[i];
    node_off[i + 1] = node_off[i] + (b.node_off[s + 1] - b.node_off[s]);
    op_off[i + 1] = op_off[i] + (b.op_off[s + 1] - b.op_off[s]);
  }
  parallel_chunks(n, n < (1u << 16) ? 1 : threads, [&](unsigned, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t s = idx[i];
      hdr[i] = b.hdr[s];
      const uint32_t nn = b.node_off[s + 1] - b.node_off[s], no = b.op_off[s + 1] - b.op_off[s];
      if (nn) memcpy(&nodes[node_off[i]], &b.nodes[b.node_off[s]], size_t(nn) * 4);
      if (no) memcpy(&ops[op_off[i]], &b.ops[b.op_off[s]], size_t(no) * 4);
    }
  });
  pt.lap("offsets + gather");
  b.hdr.swap(hdr);
  b.node_off.swap(node_off);
  b.op_off.swap(op_off);
  b.nodes.swap(nodes);
  b.ops.swap(ops);
}

namespace {
// walk_line's sink with nothing behind it: only the lengths of the nodes matter to what a line leaves in `curr_cigar_len`
struct LengthOnlySink {
  const Graph& g;
  uint32_t node_word(uint32_t n) const { return g.seg_len[n]; }
  void nc(uint32_t) const {}
  void common(uint32_t, uint32_t, uint32_t, uint32_t, uint64_t) const {}
  void rare(uint32_t, uint32_t, int32_t, uint64_t) const {}
  void edge(uint32_t, uint32_t, uint64_t) const {}
};
}  // namespace

std::vector<TildeEntry> tilde_carries(const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id) {
  // candidates: lines with a cs tag that starts with '~' (start offset and 0-based line number inside the text)
  struct Cand { size_t off; uint64_t line; };
  std::vector<Cand> cand;
  {
    const unsigned T = size < (size_t(1) << 24) ? 1u : hw_threads();
    auto ranges = split_at_newlines(data, size, T);
    std::vector<std::vector<Cand>> part(ranges.size());
    std::vector<uint64_t> lines_in(ranges.size(), 0);
    parallel_chunks(ranges.size(), unsigned(ranges.size()), [&](unsigned, size_t rb, size_t re) {
      for (size_t r = rb; r < re; ++r) {
        const char* p = data + ranges[r].first;
        const char* end = data + ranges[r].second;
        uint64_t ln = 0;
        while (p < end) {
          const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
          const char* le = nl ? nl : end;
          static const char key[] = "cs:Z:~";
          if (le - p >= 6 && memmem(p, size_t(le - p), key, 6)) part[r].push_back(Cand{size_t(p - data), ln});
          ++ln;
          p = nl ? nl + 1 : end;
        }
        lines_in[r] = ln;
      }
    });
    uint64_t base = 0;
    for (size_t r = 0; r < ranges.size(); ++r) {
      for (const Cand& c : part[r]) cand.push_back(Cand{c.off, base + c.line});
      base += lines_in[r];
    }
  }
  std::vector<TildeEntry> out;
  out.reserve(cand.size());
  const LengthOnlySink sink{g};
  for (const Cand& c : cand) {
    uint32_t carry = TILDE_UNBOUND;
    size_t at = c.off;            // start of the line under the cursor
    uint64_t line = c.line;
    while (at > 0) {
      // the line in front
      const char* prev_end = data + at - 1;                                  // its '\n'
      const char* q = static_cast<const char*>(memrchr(data, '\n', size_t(prev_end - data)));
      const size_t lb = q ? size_t(q - data) + 1 : 0;
      --line;
      GafBatch one;
      tokenize_gaf(g, data + lb, size_t(prev_end - (data + lb)), thr, first_line_id + line, 1, one);
      at = lb;
      if (one.deferred) break;                     // the reference dies there: nothing behind that line is ever read
      if (one.size() == 0) continue;               // rejected, unmapped or beyond the dv threshold: the walk never saw it
      const TildeCarry tc{out.data(), uint32_t(out.size())};
      uint32_t left = TILDE_UNBOUND;
      LengthOnlySink s2 = sink;
      const int st = walk_line(one.hdr[0], one.nodes.data(), uint32_t(one.nodes.size()), one.ops.data(), uint32_t(one.ops.size()), s2, &tc, &left);
      if (st != WALK_OK) break;                    // that line dies in the reference (or is itself unbound)
      if (left != TILDE_UNBOUND) { carry = left; break; }
      // (no op came into hand on that line -- no room on any of its nodes: `curr_cigar_len` is what it was before it)
    }
    if (first_line_id + c.line > 0xFFFFFFFFull) throw Error("Unsupported", "more than 2^32-1 GAF lines in one run");
    out.push_back(TildeEntry{uint32_t(first_line_id + c.line), carry});
  }
  return out;
}

}  // namespace pantas
