#include "graph.h"

namespace pantas {

namespace {

struct SRec { uint64_t line_off, name_off; uint32_t name_len, seq_len; bool defines; };
struct LRec { uint64_t line_off, u_off, v_off; uint32_t u_len, v_len; uint8_t ntok; };

struct ChunkOut {
  std::vector<SRec> s;
  std::vector<LRec> l;
  uint64_t lines = 0, p_lines = 0;
  uint64_t bad_p = UINT64_MAX;
};

void scan_chunk(const char* base, size_t b, size_t e, ChunkOut& out) {
  const char* p = base + b;
  const char* end = base + e;
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
    const char* le = nl ? nl : end;
    Span raw{p, le};
    Span st = strip(raw);
    ++out.lines;
    if (st.size() > 0) {
      char c = *st.b;
      if (c == 'S') {
        Tokens tk(st);
        Span t0, name, seq;
        if (!tk.next(t0) || !tk.next(name) || !tk.next(seq))
          throw Error("IndexError", "GFA byte " + std::to_string(st.b - base) + ": S line with fewer than 3 fields");
        if (name.size() > 65535) throw Error("Unsupported", "segment name longer than 65535 bytes");
        if (seq.size() >= (size_t(1) << 30)) throw Error("Unsupported", "segment longer than 2^30-1 bases");
        out.s.push_back(SRec{uint64_t(st.b - base), uint64_t(name.b - base), uint32_t(name.size()),
                             uint32_t(seq.size()), *raw.b == 'S'});
      } else if (c == 'L') {
        if (st.size() > 1) {
          Tokens tk(st);
          Span t[6];
          uint8_t n = 0;
          while (n < 6 && tk.next(t[n])) ++n;
          if (n < 4)
            throw Error("IndexError", "GFA byte " + std::to_string(st.b - base) + ": L line with fewer than 4 fields");
          out.l.push_back(LRec{uint64_t(st.b - base), uint64_t(t[1].b - base), uint64_t(t[3].b - base),
                               uint32_t(t[1].size()), uint32_t(t[3].size()), n});
        }
      } else if (c == 'P') {
        ++out.p_lines;
        Tokens tk(st);
        Span t;
        int n = 0;
        while (n < 5 && tk.next(t)) ++n;
        if (n != 4 && out.bad_p == UINT64_MAX) out.bad_p = uint64_t(st.b - base);
      }
    }
    p = nl ? nl + 1 : end;
  }
}

}  // namespace

static std::unique_ptr<Graph> parse_graph(std::unique_ptr<Graph> g, unsigned threads);

std::unique_ptr<Graph> load_graph(const char* path, unsigned threads) {
  auto g = std::make_unique<Graph>();
  g->path = path;
  g->text.open(path);
  return parse_graph(std::move(g), threads);
}

std::unique_ptr<Graph> load_graph_mem(const char* data, size_t size, unsigned threads) {
  auto g = std::make_unique<Graph>();
  g->path = "<memory>";
  g->text.attach(data, size);
  return parse_graph(std::move(g), threads);
}

static std::unique_ptr<Graph> parse_graph(std::unique_ptr<Graph> g, unsigned threads) {
  const char* base = g->text.data;
  PhaseTimer pt("load_graph");
  if (threads == 0) threads = hw_threads();
  if (g->text.size < (1u << 20)) threads = 1;
  auto ranges = split_at_newlines(base, g->text.size, threads);
  std::vector<ChunkOut> outs(ranges.size());
  parallel_chunks(ranges.size(), unsigned(ranges.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) scan_chunk(base, ranges[i].first, ranges[i].second, outs[i]);
  });

  pt.lap("scan lines (parallel)");
  size_t ns = 0, nl = 0;
  for (auto& o : outs) {
    ns += o.s.size();
    nl += o.l.size();
    g->n_lines += o.lines;
    g->n_p_lines += o.p_lines;
    if (o.bad_p != UINT64_MAX && g->first_bad_p_off == UINT64_MAX) g->first_bad_p_off = o.bad_p;
  }
  g->n_s_lines = ns;
  g->n_l_lines = nl;
  if (ns >= NONE32) throw Error("Unsupported", "more than 2^32-2 segments");

  // ---- name index ---------------------------------------------------------------------
  bool all_num = ns > 0;
  uint64_t lo = UINT64_MAX, hi = 0;
  for (auto& o : outs) {
    for (auto& s : o.s) {
      uint64_t id;
      if (!parse_canonical_u64(Span{base + s.name_off, base + s.name_off + s.name_len}, id)) { all_num = false; break; }
      lo = std::min(lo, id);
      hi = std::max(hi, id);
    }
    if (!all_num) break;
  }
  NameIndex& ix = g->names;
  if (all_num && hi - lo < std::max<uint64_t>(uint64_t(ns) * 16, 1u << 20)) {
    ix.dense = true;
    ix.base = lo;
    ix.table.assign(size_t(hi - lo + 1), NONE32);
  } else {
    ix.map.reserve(ns * 2);
  }
  g->seg_len.reserve(ns);
  g->seg_line_off.reserve(ns);
  g->seg_name_off.reserve(ns);
  g->seg_name_len.reserve(ns);
  for (auto& o : outs) {
    for (auto& s : o.s) {
      Span nm{base + s.name_off, base + s.name_off + s.name_len};
      uint32_t* slot;
      if (ix.dense) {
        uint64_t id = 0;
        parse_canonical_u64(nm, id);
        slot = &ix.table[size_t(id - lo)];
      } else {
        slot = &ix.map.emplace(nm.sv(), NONE32).first->second;
      }
      if (*slot == NONE32) {
        if (!s.defines)
          throw Error("KeyError", "GFA byte " + std::to_string(s.line_off) +
                                      ": S line behind leading whitespace names a segment no plain S line defines");
        *slot = uint32_t(g->seg_len.size());
        g->seg_len.push_back(s.seq_len);
        g->seg_line_off.push_back(s.line_off);
        g->seg_name_off.push_back(s.name_off);
        g->seg_name_len.push_back(uint16_t(s.name_len));
      } else {
        g->dup_names = true;
        if (s.defines) {
          g->seg_len[*slot] = s.seq_len;
          g->seg_line_off[*slot] = s.line_off;
        }
      }
    }
  }
  pt.lap("name index (sequential)");
  g->V = uint32_t(g->seg_len.size());
  for (uint32_t v : g->seg_len) g->max_seg_len = std::max(g->max_seg_len, v);

  // ---- resolve link ends (parallel) and build the out CSR in file order --------------------
  std::vector<size_t> lbase(outs.size() + 1, 0);
  for (size_t i = 0; i < outs.size(); ++i) lbase[i + 1] = lbase[i] + outs[i].l.size();
  std::vector<uint32_t> lu(nl), lv(nl);
  parallel_chunks(outs.size(), unsigned(outs.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      size_t k = lbase[i];
      for (auto& r : outs[i].l) {
        lu[k] = ix.find(Span{base + r.u_off, base + r.u_off + r.u_len});
        lv[k] = ix.find(Span{base + r.v_off, base + r.v_off + r.v_len});
        ++k;
      }
    }
  });
  pt.lap("resolve link ends (parallel)");
  g->row_ptr.assign(size_t(g->V) + 1, 0);
  g->irow_ptr.assign(size_t(g->V) + 1, 0);
  uint64_t A = 0;
  {
    size_t k = 0;
    for (auto& o : outs)
      for (auto& r : o.l) {
        if (r.ntok < 6 && g->first_short_l_off == UINT64_MAX) g->first_short_l_off = r.line_off;
        if (lu[k] == NONE32 || lv[k] == NONE32) {
          ++g->links_unknown_end;
        } else {
          ++g->row_ptr[lu[k] + 1];
          ++g->irow_ptr[lv[k] + 1];
          ++A;
          if (g->seg_line_off[lu[k]] > r.line_off || g->seg_line_off[lv[k]] > r.line_off) g->link_before_segment = true;
        }
        ++k;
      }
  }
  if (A >= NONE32) throw Error("Unsupported", "more than 2^32-2 links");
  g->A = A;
  for (uint32_t i = 0; i < g->V; ++i) g->row_ptr[i + 1] += g->row_ptr[i];
  for (uint32_t i = 0; i < g->V; ++i) g->irow_ptr[i + 1] += g->irow_ptr[i];
  g->col.resize(A);
  g->slot_off.resize(A);
  g->icol.resize(A);
  g->islot.resize(A);
  {
    std::vector<uint32_t> cur(g->row_ptr.begin(), g->row_ptr.end() - 1);
    std::vector<uint32_t> icur(g->irow_ptr.begin(), g->irow_ptr.end() - 1);
    size_t k = 0;
    for (auto& o : outs)
      for (auto& r : o.l) {
        if (lu[k] != NONE32 && lv[k] != NONE32) {
          uint32_t p = cur[lu[k]]++;
          g->col[p] = lv[k];
          g->slot_off[p] = r.line_off;
          uint32_t q = icur[lv[k]]++;   // in-lists follow L-line FILE order (call.py:175)
          g->icol[q] = lu[k];
          g->islot[q] = p;
        }
        ++k;
      }
  }
  pt.lap("CSR fill (sequential)");
  return g;
}

}  // namespace pantas
