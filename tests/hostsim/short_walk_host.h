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

// Round 4, one lane per path position (walk.h tile_*): the line stands in a tile of 64 lanes, alone or behind / in front of copies
// of itself (the copies' calls are dropped: they are there so that records start at other lanes than 0, masks have several heads and
// neighbours are looked for across record borders), and the 64 lanes are looped over where the kernel does its cross-lane steps --
// the wave scan of the rooms, the two lane reads, the ballot of the kept positions.
template <class GenSink>
int tile_walk_host(const Graph& g, const std::vector<uint32_t>& word, const LineHdr& h, const uint32_t* nodes, uint32_t k, const uint32_t* ops,
                   uint32_t c, GenSink& s) {
  if (k == 0) return WALK_OK;
  const HostInfo ga{g, word};
  ShortOps so;
  short_ops_scan(ops, c, so);
  const PosLine pr = make_pos_rec(h, so).line;
  // copies in front of / behind the line, by line id; greedy packing decides where records start (tile_pack)
  const uint32_t want_front = (h.line_id >> 1) % 3u, want_back = (h.line_id >> 3) % 3u;
  uint32_t fill = 0, n_rec = 0, mine = 0;
  uint32_t start_of[8];
  for (uint32_t r = 0; r < want_front + 1u + want_back; ++r) {
    uint32_t f = fill;
    if (tile_pack(f, k, r == 0) && r != 0) {
      if (r <= want_front) { fill = 0; n_rec = 0; f = k; }      // the tile closes in front of the line: start over with it
      else break;                                              // ... behind it: done
    }
    start_of[n_rec] = f - k;
    if (r == want_front) mine = n_rec;
    ++n_rec;
    fill = f;
  }
  const uint32_t used = fill;
  uint32_t node[64], meta[64], room[64], incl[64], plain[64];
  NodeInfo3 inf[64];
  TileLane tl[64];
  bool keep[64];
  for (uint32_t l = 0; l < 64; ++l) meta[l] = POS_INACTIVE;
  for (uint32_t r = 0; r < n_rec; ++r)
    for (uint32_t j = 0; j < k; ++j) meta[start_of[r] + j] = pos_meta(j, k, r);
  for (uint32_t l = 0; l < 64; ++l) {
    const bool active = meta[l] != POS_INACTIVE;
    tl[l] = tile_lane(meta[l], l);
    node[l] = active ? nodes[meta[l] & 31u] : nodes[0];
    inf[l] = ga.info3(node[l]);
    room[l] = active ? tile_room(inf[l].word, (meta[l] & 31u) == 0u, ((meta[l] >> 5) & 31u) == 0u, pr) : 0u;
    if (active && (l < used) != true) return -1;
    if (active && (pos_ordinal(meta[l]) >= n_rec || tl[l].seg_start != start_of[pos_ordinal(meta[l])])) return -1;
  }
  uint32_t sum = 0;
  for (uint32_t l = 0; l < 64; ++l) { sum += room[l]; incl[l] = sum; }
  uint64_t kept = 0;
  int st = WALK_OK;
  for (uint32_t l = 0; l < 64; ++l) {
    const uint32_t base = incl[(tl[l].seg_start - 1u) & 63u];
    const uint32_t in_front = incl[l] - room[l] - (tl[l].seg_start ? base : 0u);
    bool exhausted;
    tile_keep(pr, so, room[l], in_front, keep[l], plain[l], exhausted);
    if (exhausted) st = WALK_OPS_EXHAUSTED;
    if (keep[l]) kept |= 1ull << l;
  }
  if (st != WALK_OK) return st;      // (walk_short_w has made the calls in front of the refused node by then; a refused line ends the run)
  const ShortWFromGeneral<GenSink> sw{s, g, h.line_id};
  for (uint32_t l = start_of[mine]; l < start_of[mine] + k; ++l) {
    if (!keep[l]) continue;
    const uint64_t le = tile_lanes_le(l);
    const TileNeighbour nb = tile_neighbour(kept, tl[l], l, le, (pr.tw & POS_REV) != 0);
    tile_emit(sw, node[l], inf[l], plain[l], pr, tl[l], nb, l, node[nb.lane], tile_kept_index(kept, tl[l], le));
  }
  return WALK_OK;
}

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
  static const bool tile_only = getenv("HOSTSIM_TILE_WALK") != nullptr;      // ... through the position-parallel walk (round 4, the default kernel)
  const uint32_t sel = tile_only ? 6u : wide_only ? 4u + (h.line_id & 1u) : h.line_id % 8u;
  switch (sel) {
    case 6: case 7: return tile_walk_host(g, word, h, nodes, k, ops, c, s);
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
