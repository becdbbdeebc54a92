// TEST HARNESS ONLY.  Host-side views for the short walks of walk.h (walk_short, walk_short_rare), so that the CPU harness
// walks every line the way the kernels do: short-walk class through walk_short, the rest through walk_line.
#pragma once
#include <vector>

#include "graph.h"
#include "walk.h"

namespace pantas {

// the graph as one record per node: {length word incl. the flags the rare pass has set, first out-target or NONE}
struct HostInfo {
  const Graph& g;
  const std::vector<uint32_t>& word;
  NodeInfo info(uint32_t n) const {
    const uint32_t lo = g.row_ptr[n], hi = g.row_ptr[n + 1];
    return NodeInfo{word[n], lo < hi ? g.col[lo] : 0xFFFFFFFFu};
  }
  // the 16-byte record of walk_short_w: targets of the first three out-slots
  NodeInfo3 info3(uint32_t n) const {
    const uint32_t lo = g.row_ptr[n], hi = g.row_ptr[n + 1];
    return NodeInfo3{word[n], lo < hi ? g.col[lo] : NO_TARGET, lo + 1 < hi ? g.col[lo + 1] : NO_TARGET, lo + 2 < hi ? g.col[lo + 2] : NO_TARGET};
  }
};

// walk_short_w's sink in terms of a walk_line sink: a slot code is a hit on that out-slot of the node, link_later is looked up at once
template <class GenSink>
struct ShortWFromGeneral {
  GenSink& s;
  const Graph& g;
  void kept(uint32_t node, uint32_t word, uint32_t line_id, uint32_t a_il, uint32_t a_ol, uint32_t code, uint64_t exact_order) const {
    s.nc(node);
    const uint64_t order = exact_order ? exact_order : event_order(line_id, 0, 1);
    if (a_il) s.common(node, word, SIDE_IL, a_il, order);
    if (a_ol) s.common(node, word, SIDE_OL, a_ol, order);
    if (code) s.edge(node, g.col[g.row_ptr[node] + code - 1u], 0);
  }
  uint32_t line_id;
  void later(uint32_t a, uint32_t b, uint32_t idx) const { s.edge(a, b, event_order(line_id, idx, 0)); }
};

// walk_short's sink in terms of a walk_line sink (nc / common / edge)
template <class GenSink>
struct ShortFromGeneral {
  GenSink& s;
  const Graph& g;
  void kept(uint32_t node, uint32_t word, uint32_t line_id, uint32_t a_il, uint32_t a_ol, uint32_t r0, uint64_t exact_order) const {
    s.nc(node);
    const uint64_t order = exact_order ? exact_order : event_order(line_id, 0, 1);
    if (a_il) s.common(node, word, SIDE_IL, a_il, order);
    if (a_ol) s.common(node, word, SIDE_OL, a_ol, order);
    for (uint32_t i = 0; i < r0; ++i) s.edge(node, g.col[g.row_ptr[node]], 0);      // the node's first out-slot
  }
  void link_rest(uint32_t a, uint32_t b, uint64_t order) const { s.edge(a, b, order); }
  // walk_short<.., DEFER>: links off the first out-slot as path indices, looked up after the node loop (as the kernel does)
  mutable std::vector<uint32_t> q;
  void defer(uint32_t ia, uint32_t ib, uint32_t kidx, uint32_t a, uint32_t b, uint32_t line_id) const {
    if (q.size() < 4) q.push_back(ia | (ib << 5) | (kidx << 10));
    else link_rest(a, b, event_order(line_id, kidx, 0));
  }
  void drain(const uint32_t* nodes, uint32_t line_id) const {
    for (uint32_t e : q) link_rest(nodes[e & 31u], nodes[(e >> 5) & 31u], event_order(line_id, e >> 10, 0));
    q.clear();
  }
};

// main pass over one line, as augment.hip dispatches it
template <class GenSink>
int main_walk(const Graph& g, const std::vector<uint32_t>& word, const LineHdr& h, const uint32_t* nodes, uint32_t k, const uint32_t* ops,
              uint32_t c, GenSink& s, const TildeCarry* tc = nullptr) {
  if (!short_walk_class(h.flags, k) || getenv("HOSTSIM_GENERAL_WALK")) return walk_line(h, nodes, k, ops, c, s, tc);
  const HostInfo ga{g, word};
  const ShortFromGeneral<GenSink> ss{s, g};
  uint32_t first_ids[4] = {0, 0, 0, 0};
  for (uint32_t j = 0; j < 4 && k; ++j) first_ids[j] = nodes[j < k - 1 ? j : k - 1];
  // every instance the kernels use takes its turn: gathers one / two nodes ahead, deferred link lookups, the lean
  // instance for lines without an indel
  static const bool wide_only = getenv("HOSTSIM_WIDE_WALK") != nullptr;      // every short-walk line through walk_short_w
  const uint32_t sel = wide_only ? 4u + (h.line_id & 1u) : h.line_id % 6u;
  switch (sel) {
    case 4: case 5: {     // round 3: 16-byte node records, links matched against the first three out-slots where the node is visited
      const ShortWFromGeneral<GenSink> sw{s, g, h.line_id};
      ShortOps so;
      short_ops_scan(ops, c, so);
      return sel == 4 ? walk_short_w<1>(ga, h, nodes, k, sw, first_ids, so) : walk_short_w<2>(ga, h, nodes, k, sw, first_ids, so);
    }
    case 0: return walk_short<1>(ga, h, nodes, k, ops, c, ss, first_ids);
    case 1: {      // with the line's ShortOps handed in (the GPU batch keeps it beside the record) and ops[] out of reach
      ShortOps so;
      short_ops_scan(ops, c, so);
      return walk_short<2>(ga, h, nodes, k, nullptr, c, ss, first_ids, &so);
    }
    case 2: return walk_short<2, true>(ga, h, nodes, k, ops, c, ss, first_ids);
    default:
      if (h.flags & LINE_SIMPLE) return walk_short<2, true, true>(ga, h, nodes, k, ops, c, ss, first_ids);
      return walk_short<1, true>(ga, h, nodes, k, ops, c, ss, first_ids);
  }
}

// rare-key pass over one line with a '-'
template <class GenSink>
int rare_walk(const Graph& g, const std::vector<uint32_t>& word, const LineHdr& h, const uint32_t* nodes, uint32_t k, const uint32_t* ops,
              uint32_t c, GenSink& s, const TildeCarry* tc = nullptr) {
  if (!(h.flags & LINE_ONE_INDEL) || k == 0 || getenv("HOSTSIM_GENERAL_WALK")) return walk_line(h, nodes, k, ops, c, s, tc);
  // the rare sink sets flags in `word` while it walks: the short walk only reads lengths from it
  const HostInfo ga{g, word};
  return walk_short_rare(ga, h, nodes, k, ops, c, s);
}

}  // namespace pantas
