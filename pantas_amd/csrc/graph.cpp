#include "graph.h"

#include <atomic>

namespace pantas {

namespace {

// One record per S / L line between the scan and the arrays of the graph: 2.5e8 + 2.6e8 of them at whole-genome scale, written once,
// read twice and given back -- their size is load time (16 and 24 bytes: field positions as offsets from the line start, 10 GB
// instead of 18).
struct SRec {
  uint64_t line_off;
  uint32_t seq_len_def;                 // len(sequence) (< 2^30) | bit 31: the line starts with 'S' itself (no whitespace before it)
  uint16_t name_delta, name_len;
  uint64_t name_off() const { return line_off + name_delta; }
  uint32_t seq_len() const { return seq_len_def & 0x7FFFFFFFu; }
  bool defines() const { return (seq_len_def >> 31) != 0; }
};
struct LRec {
  uint64_t line_off;
  uint32_t v_delta;
  uint16_t u_delta, u_len, v_len;
  uint8_t ntok;
  uint64_t u_off() const { return line_off + u_delta; }
  uint64_t v_off() const { return line_off + v_delta; }
};
static_assert(sizeof(SRec) == 16 && sizeof(LRec) == 24, "loader records");

struct ChunkOut {
  RawVector<SRec> s;      // (18 GB of records at whole-genome scale: huge pages)
  RawVector<LRec> l;
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
    if (const char* cr = lone_cr(p, le))
      throw Error("Unsupported", "GFA byte " + std::to_string(cr - base) + ": " + LONE_CR_MARK + " (Python's universal newlines end a line there; the loader reads a copy with them applied)");
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
        if (name.b - st.b > 65535) throw Error("Unsupported", "GFA byte " + std::to_string(st.b - base) + ": more than 65535 bytes between the line start and the segment name");
        out.s.push_back(SRec{uint64_t(st.b - base), uint32_t(seq.size()) | (*raw.b == 'S' ? 0x80000000u : 0u), uint16_t(name.b - st.b), uint16_t(name.size())});
      } else if (c == 'L') {
        if (st.size() > 1) {
          Tokens tk(st);
          Span t[6];
          uint8_t n = 0;
          while (n < 6 && tk.next(t[n])) ++n;
          if (n < 4)
            throw Error("IndexError", "GFA byte " + std::to_string(st.b - base) + ": L line with fewer than 4 fields");
          if (t[1].size() > 65535 || t[3].size() > 65535) throw Error("Unsupported", "segment name longer than 65535 bytes");
          if (t[1].b - st.b > 65535 || uint64_t(t[3].b - st.b) > 0xFFFFFFFFull)
            throw Error("Unsupported", "GFA byte " + std::to_string(st.b - base) + ": an L line with its fields too far apart");
          out.l.push_back(LRec{uint64_t(st.b - base), uint32_t(t[3].b - st.b), uint16_t(t[1].b - st.b), uint16_t(t[1].size()), uint16_t(t[3].size()), n});
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

// the graph over a copy of the text in which a '\r' that no '\n' follows ends its line, as it does for the reference
static std::unique_ptr<Graph> parse_with_universal_newlines(const char* path, const char* data, size_t size, unsigned threads) {
  auto g = std::make_unique<Graph>();
  g->path = path;
  universal_newlines_copy(data, size, g->text_owned);
  g->text.attach(g->text_owned.data(), g->text_owned.size());
  return parse_graph(std::move(g), threads);
}

std::unique_ptr<Graph> load_graph(const char* path, unsigned threads) {
  {
    auto g = std::make_unique<Graph>();
    g->path = path;
    g->text.open(path);
    try {
      return parse_graph(std::move(g), threads);
    } catch (const Error& e) {
      if (!is_lone_cr_error(e)) throw;
    }
  }
  MappedFile f;
  f.open(path);
  return parse_with_universal_newlines(path, f.data, f.size, threads);
}

std::unique_ptr<Graph> load_graph_mem(const char* data, size_t size, unsigned threads) {
  {
    auto g = std::make_unique<Graph>();
    g->path = "<memory>";
    g->text.attach(data, size);
    try {
      return parse_graph(std::move(g), threads);
    } catch (const Error& e) {
      if (!is_lone_cr_error(e)) throw;
    }
  }
  return parse_with_universal_newlines("<memory>", data, size, threads);
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
      if (!parse_canonical_u64(Span{base + s.name_off(), base + s.name_off() + s.name_len}, id)) { all_num = false; break; }
      lo = std::min(lo, id);
      hi = std::max(hi, id);
    }
    if (!all_num) break;
  }
  NameIndex& ix = g->names;
  if (all_num && hi - lo < std::max<uint64_t>(uint64_t(ns) * 16, 1u << 20)) {
    ix.dense = true;
    ix.base = lo;
    par_assign(ix.table, size_t(hi - lo + 1), NONE32, threads);
  } else {
    ix.map.reserve(ns * 2);
  }
  // S-record ordinals: chunk bases
  std::vector<size_t> sbase(outs.size() + 1, 0);
  for (size_t i = 0; i < outs.size(); ++i) sbase[i + 1] = sbase[i] + outs[i].s.size();
  bool fast_names = false;
  const bool force_seq = getenv("PANTAS_LOADER_SEQUENTIAL") != nullptr;   // tests compare both paths
  if (ix.dense && ns > (1u << 16) && !force_seq) {
    // optimistic parallel fill: with no duplicate names the node index is the S-line ordinal
    g->seg_len.resize(ns);
    g->seg_line_off.resize(ns);
    g->seg_name_off.resize(ns);
    g->seg_name_len.resize(ns);
    std::atomic<bool> dup{false}, undefined{false};
    parallel_chunks(outs.size(), unsigned(outs.size()), [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        size_t k = sbase[i];
        for (auto& s : outs[i].s) {
          uint64_t id = 0;
          parse_canonical_u64(Span{base + s.name_off(), base + s.name_off() + s.name_len}, id);
          uint32_t expect = NONE32;
          if (!__atomic_compare_exchange_n(&ix.table[size_t(id - lo)], &expect, uint32_t(k), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
            dup.store(true, std::memory_order_relaxed);
          if (!s.defines()) undefined.store(true, std::memory_order_relaxed);
          g->seg_len[k] = s.seq_len();
          g->seg_line_off[k] = s.line_off;
          g->seg_name_off[k] = s.name_off();
          g->seg_name_len[k] = uint16_t(s.name_len);
          ++k;
        }
      }
    });
    if (!dup.load() && !undefined.load()) {
      fast_names = true;
    } else {
      // duplicates (or S lines behind leading whitespace): redo in file order
      par_assign(ix.table, ix.table.size(), NONE32, threads);
      g->seg_len.clear(); g->seg_line_off.clear(); g->seg_name_off.clear(); g->seg_name_len.clear();
    }
  }
  if (!fast_names) {
  g->seg_len.reserve(ns);
  g->seg_line_off.reserve(ns);
  g->seg_name_off.reserve(ns);
  g->seg_name_len.reserve(ns);
  std::vector<uint64_t> first_off;
  std::vector<bool> first_defines;
  first_off.reserve(ns);
  first_defines.reserve(ns);
  for (auto& o : outs) {
    for (auto& s : o.s) {
      Span nm{base + s.name_off(), base + s.name_off() + s.name_len};
      uint32_t* slot;
      if (ix.dense) {
        uint64_t id = 0;
        parse_canonical_u64(nm, id);
        slot = &ix.table[size_t(id - lo)];
      } else {
        slot = &ix.map.emplace(nm.sv(), NONE32).first->second;
      }
      if (*slot == NONE32) {
        // an S line behind leading whitespace that no plain S line backs: call.py strips the line before it looks at its
        // first byte (:146-147) and takes it; the augment script misses it in its first pass (:123) and dies with a
        // KeyError in the second (:387) -- the segment exists, and pantas_augment_reset refuses the graph
        if (!s.defines() && g->ws_only_seg_off == UINT64_MAX) g->ws_only_seg_off = s.line_off;
        *slot = uint32_t(g->seg_len.size());
        g->seg_len.push_back(s.seq_len());
        g->seg_line_off.push_back(s.line_off);
        g->seg_name_off.push_back(s.name_off());
        g->seg_name_len.push_back(uint16_t(s.name_len));
        first_off.push_back(s.line_off);
        first_defines.push_back(s.defines());
      } else {
        g->dup_names = true;
        if (!s.defines() || !first_defines[*slot]) g->dup_ws = true;
        if (s.defines()) {
          g->seg_len[*slot] = s.seq_len();
          g->seg_line_off[*slot] = s.line_off;
        }
      }
    }
  }
  if (g->dup_names) {
    // call.py keeps the LAST S line of a name and restarts the segment's adjacency lists there (:149-154); a link is only
    // an error when it stands in front of the FIRST one (:174)
    g->seg_first_off.resize(first_off.size());
    std::copy(first_off.begin(), first_off.end(), g->seg_first_off.data());
  }
  }
  pt.lap(fast_names ? "name index (parallel)" : "name index (sequential)");
  g->V = uint32_t(g->seg_len.size());
  {
    std::vector<uint32_t> mx(std::max<size_t>(1, outs.size()), 0);
    parallel_chunks(g->V, g->V < (1u << 16) ? 1 : unsigned(mx.size()), [&](unsigned c, size_t b, size_t e) {
      uint32_t m = 0;
      for (size_t i = b; i < e; ++i) m = std::max(m, g->seg_len[i]);
      mx[c] = m;
    });
    for (uint32_t m : mx) g->max_seg_len = std::max(g->max_seg_len, m);
  }

  // ---- resolve link ends (parallel) and build the out CSR in file order --------------------
  std::vector<size_t> lbase(outs.size() + 1, 0);
  for (size_t i = 0; i < outs.size(); ++i) lbase[i + 1] = lbase[i] + outs[i].l.size();
  RawVector<uint32_t> lu(nl), lv(nl);
  parallel_chunks(outs.size(), unsigned(outs.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      size_t k = lbase[i];
      for (auto& r : outs[i].l) {
        lu[k] = ix.find(Span{base + r.u_off(), base + r.u_off() + r.u_len});
        lv[k] = ix.find(Span{base + r.v_off(), base + r.v_off() + r.v_len});
        ++k;
      }
    }
  });
  pt.lap("resolve link ends (parallel)");
  // per-record facts (parallel): validity, sortedness by source, the flags the callers look at
  const unsigned T = unsigned(outs.size());
  std::vector<uint64_t> c_valid(T + 1, 0), c_unknown(T, 0), c_short(T, UINT64_MAX);
  std::vector<uint8_t> c_unsorted(T, 0), c_before(T, 0);
  parallel_chunks(T, T, [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      size_t k = lbase[i];
      uint32_t prev = 0;
      bool have_prev = false;
      for (auto& r : outs[i].l) {
        if (r.ntok < 6 && c_short[i] == UINT64_MAX) c_short[i] = r.line_off;
        if (lu[k] == NONE32 || lv[k] == NONE32) {
          ++c_unknown[i];
        } else {
          ++c_valid[i + 1];
          const RawVector<uint64_t>& first = g->dup_names ? g->seg_first_off : g->seg_line_off;
          if (first[lu[k]] > r.line_off || first[lv[k]] > r.line_off) c_before[i] = 1;
          if (have_prev && lu[k] < prev) c_unsorted[i] = 1;
          prev = lu[k];
          have_prev = true;
        }
        ++k;
      }
    }
  });
  bool sorted_by_source = true;
  {
    // also across chunk borders
    uint32_t prev = 0;
    bool have_prev = false;
    for (unsigned i = 0; i < T; ++i) {
      if (c_unsorted[i]) sorted_by_source = false;
      size_t k = lbase[i];
      for (size_t j = 0; j < outs[i].l.size(); ++j, ++k) {
        if (lu[k] == NONE32 || lv[k] == NONE32) continue;
        if (have_prev && lu[k] < prev) sorted_by_source = false;
        break;
      }
      // last valid source of the chunk
      for (size_t j = outs[i].l.size(), k2 = lbase[i + 1]; j-- > 0;) {
        --k2;
        if (lu[k2] == NONE32 || lv[k2] == NONE32) continue;
        prev = lu[k2];
        have_prev = true;
        break;
      }
      g->links_unknown_end += c_unknown[i];
      if (c_short[i] != UINT64_MAX && g->first_short_l_off == UINT64_MAX) g->first_short_l_off = c_short[i];
      if (c_before[i]) g->link_before_segment = true;
      c_valid[i + 1] += c_valid[i];
    }
  }
  uint64_t A = c_valid[T];
  if (A >= NONE32) throw Error("Unsupported", "more than 2^32-2 links");
  g->A = A;
  par_assign(g->row_ptr, size_t(g->V) + 1, 0u, threads);
  par_assign(g->irow_ptr, size_t(g->V) + 1, 0u, threads);
  g->col.resize(A);
  g->slot_off.resize(A);
  g->icol.resize(A);
  g->islot.resize(A);
  const bool fast_csr = sorted_by_source && A > (1u << 16) && !force_seq;
  bool csr_merged = false;
  if (fast_csr) {
    // links already grouped by source in node order (what vg / the annotator write): slot = rank among valid links
    RawVector<uint32_t> ikey(A), iidx(A);
    parallel_chunks(T, T, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        size_t k = lbase[i];
        uint64_t p = c_valid[i];
        for (auto& r : outs[i].l) {
          if (lu[k] != NONE32 && lv[k] != NONE32) {
            g->col[p] = lv[k];
            g->slot_off[p] = r.line_off;
            ikey[p] = lv[k];
            iidx[p] = uint32_t(p);
            ++p;
          }
          ++k;
        }
      }
    });
    // row_ptr[u] = first slot whose source is >= u.  Sources of slots: recover from the records again.
    RawVector<uint32_t> src(A);
    parallel_chunks(T, T, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        size_t k = lbase[i];
        uint64_t p = c_valid[i];
        for (size_t j = 0; j < outs[i].l.size(); ++j, ++k)
          if (lu[k] != NONE32 && lv[k] != NONE32) src[p++] = lu[k];
      }
    });
    auto fill_ptr = [&](const RawVector<uint32_t>& keys, RawVector<uint32_t>& ptr) {
      // keys are sorted: ptr[u] = lower_bound(keys, u)
      parallel_chunks(A, threads, [&](unsigned, size_t b, size_t e) {
        for (size_t p = b; p < e; ++p) {
          const uint32_t cur = keys[p];
          const uint32_t from = p == 0 ? 0u : keys[p - 1] + 1u;
          if (p == 0 || keys[p - 1] != cur)
            for (uint32_t u = from; u <= cur; ++u) ptr[u] = uint32_t(p);
        }
      });
      const uint32_t last = A ? keys[A - 1] : 0u;
      for (size_t u = size_t(A ? last + 1 : 0); u <= g->V; ++u) ptr[u] = uint32_t(A);
    };
    fill_ptr(src, g->row_ptr);
    // in-lists: stable sort of the slots by target keeps L-line file order inside a list (call.py:175)
    radix_sort_pairs(ikey, iidx, threads);
    parallel_chunks(A, threads, [&](unsigned, size_t b, size_t e) {
      for (size_t q = b; q < e; ++q) {
        g->islot[q] = iidx[q];
        g->icol[q] = src[iidx[q]];
      }
    });
    fill_ptr(ikey, g->irow_ptr);
  } else if (A > (1u << 16) && !force_seq) {
    // Links in any order (an augmented GFA carries the links the reads added behind its P lines): rows are the links of a
    // source in FILE order, i.e. a stable sort of the file-order records by source -- parallel radix sort -- and the in-lists
    // a stable sort by target of the same file-order records (call.py:174-175 appends to both lists line by line).
    RawVector<uint32_t> fsrc(A), fdst(A), okey(A), oidx(A), ikey(A), iidx(A);
    RawVector<uint64_t> foff(A);
    parallel_chunks(T, T, [&](unsigned, size_t b2, size_t e2) {
      for (size_t i = b2; i < e2; ++i) {
        size_t k = lbase[i];
        uint64_t p = c_valid[i];
        for (auto& r : outs[i].l) {
          if (lu[k] != NONE32 && lv[k] != NONE32) {
            fsrc[p] = lu[k]; fdst[p] = lv[k]; foff[p] = r.line_off;
            okey[p] = lu[k]; oidx[p] = uint32_t(p);
            ikey[p] = lv[k]; iidx[p] = uint32_t(p);
            ++p;
          }
          ++k;
        }
      }
    });
    auto fill_ptr = [&](const RawVector<uint32_t>& keys, RawVector<uint32_t>& ptr) {
      parallel_chunks(A, threads, [&](unsigned, size_t b2, size_t e2) {
        for (size_t p = b2; p < e2; ++p) {
          const uint32_t cur = keys[p];
          const uint32_t from = p == 0 ? 0u : keys[p - 1] + 1u;
          if (p == 0 || keys[p - 1] != cur)
            for (uint32_t u = from; u <= cur; ++u) ptr[u] = uint32_t(p);
        }
      });
      const uint32_t last = A ? keys[A - 1] : 0u;
      for (size_t u = size_t(A ? last + 1 : 0); u <= g->V; ++u) ptr[u] = uint32_t(A);
    };
    // The usual case of "any order" is an augmented GFA: the original links, grouped by source as vg wrote them, and behind them
    // the few links the reads added.  Then the stable sort by source is a merge of the sorted head with the (sorted) tail: a head
    // record moves up by the tail records with a SMALLER source, a tail record stands behind every head record with the same or a
    // smaller source -- no second radix sort over 2.6e8 records.
    size_t head = A;
    csr_merged = false;
    {
      std::vector<size_t> first_drop(threads, A);
      parallel_chunks(A, threads, [&](unsigned t, size_t b2, size_t e2) {
        for (size_t p2 = std::max<size_t>(b2, 1); p2 < e2; ++p2)
          if (fsrc[p2] < fsrc[p2 - 1]) { first_drop[t] = p2; break; }
      });
      for (size_t v : first_drop) head = std::min(head, v);
    }
    if (A - head <= A / 8 && !getenv("PANTAS_CSR_NO_MERGE")) {
      csr_merged = true;
      const size_t tn = A - head;
      std::vector<uint32_t> tidx(tn);
      for (size_t j = 0; j < tn; ++j) tidx[j] = uint32_t(head + j);
      std::stable_sort(tidx.begin(), tidx.end(), [&](uint32_t a, uint32_t b2) { return fsrc[a] < fsrc[b2]; });
      std::vector<uint32_t> tsrc(tn);
      for (size_t j = 0; j < tn; ++j) tsrc[j] = fsrc[tidx[j]];
      parallel_chunks(head, threads, [&](unsigned, size_t b2, size_t e2) {
        if (b2 >= e2) return;
        size_t kt = size_t(std::lower_bound(tsrc.begin(), tsrc.end(), fsrc[b2]) - tsrc.begin());
        for (size_t p2 = b2; p2 < e2; ++p2) {
          while (kt < tn && tsrc[kt] < fsrc[p2]) ++kt;
          okey[p2 + kt] = fsrc[p2];
          oidx[p2 + kt] = uint32_t(p2);
        }
      });
      const uint32_t* hb = fsrc.data();
      parallel_chunks(tn, tn < 4096 ? 1 : threads, [&](unsigned, size_t b2, size_t e2) {
        for (size_t j = b2; j < e2; ++j) {
          const size_t q = size_t(std::upper_bound(hb, hb + head, tsrc[j]) - hb) + j;
          okey[q] = tsrc[j];
          oidx[q] = tidx[j];
        }
      });
    } else {
      radix_sort_pairs(okey, oidx, threads);        // slot q holds file-order record oidx[q]
    }
    RawVector<uint32_t> slot_of(A);
    parallel_chunks(A, threads, [&](unsigned, size_t b2, size_t e2) {
      for (size_t q = b2; q < e2; ++q) {
        const uint32_t p = oidx[q];
        g->col[q] = fdst[p];
        g->slot_off[q] = foff[p];
        slot_of[p] = uint32_t(q);
      }
    });
    fill_ptr(okey, g->row_ptr);
    radix_sort_pairs(ikey, iidx, threads);
    parallel_chunks(A, threads, [&](unsigned, size_t b2, size_t e2) {
      for (size_t q = b2; q < e2; ++q) {
        g->islot[q] = slot_of[iidx[q]];
        g->icol[q] = fsrc[iidx[q]];
      }
    });
    fill_ptr(ikey, g->irow_ptr);
  } else {
  {
    size_t k = 0;
    for (auto& o : outs)
      for (size_t j = 0; j < o.l.size(); ++j, ++k) {
        if (lu[k] != NONE32 && lv[k] != NONE32) {
          ++g->row_ptr[lu[k] + 1];
          ++g->irow_ptr[lv[k] + 1];
        }
      }
  }
  for (uint32_t i = 0; i < g->V; ++i) g->row_ptr[i + 1] += g->row_ptr[i];
  for (uint32_t i = 0; i < g->V; ++i) g->irow_ptr[i + 1] += g->irow_ptr[i];
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
  }
  pt.lap(fast_csr ? "CSR build (parallel)" : (A > (1u << 16) && !force_seq) ? (csr_merged ? "CSR build (parallel, sorted links + a tail in any order)" : "CSR build (parallel, links in any order)") : "CSR fill (sequential)");
  if (ns + nl > (size_t(1) << 24)) {
    // the per-line records of a whole-genome graph are ~20 GB: giving them back to the kernel takes a second, which a
    // thread of its own can spend while the caller goes on
    auto* junk = new std::tuple<std::vector<ChunkOut>, RawVector<uint32_t>, RawVector<uint32_t>>(std::move(outs), std::move(lu), std::move(lv));
    defer_free((ns * sizeof(SRec) + nl * (sizeof(LRec) + 8)) * 3 / 2, [junk] { delete junk; });      // (common.h: when the output is being written)
  }
  return g;
}

}  // namespace pantas
