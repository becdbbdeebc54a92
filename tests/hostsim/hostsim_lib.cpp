// TEST HARNESS ONLY -- CPU stand-in for the augment kernels with the staged interface the
// sharded driver (pantas_amd/distributed.py) talks to, so that the N>1 exchange logic can be
// exercised over gloo where no GPU exists.  Built by tests/test_distributed.py into
// build/hostsim/libhostsim.so; the product never loads it.
#include <map>
#include <tuple>
#include "augment.h"
#include "walk.h"
#include "short_walk_host.h"

using namespace pantas;

struct HS {
  std::unique_ptr<Graph> g;
  GafBatch b;
  std::vector<uint32_t> word;
  std::vector<uint32_t> counters;   // [NC | IL0 | OLL | RC]
  std::map<std::tuple<uint32_t, uint32_t, int32_t>, std::pair<uint32_t, uint64_t>> rare;
  std::map<std::pair<uint32_t, uint32_t>, uint64_t> common_first;
  std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint64_t>> novel;
  AugmentResult res;
  std::string err;
};

namespace {
struct RareSink {
  HS& t;
  uint32_t node_word(uint32_t n) { return t.word[n]; }
  void nc(uint32_t) {}
  void common(uint32_t, uint32_t, uint32_t, uint32_t, uint64_t) {}
  void rare(uint32_t n, uint32_t side, int32_t pos, uint64_t order) {
    auto& e = t.rare.emplace(std::make_tuple(n, side, pos), std::make_pair(0u, UINT64_MAX)).first->second;
    e.first += 1;
    e.second = std::min(e.second, order);
    t.word[n] |= side == SIDE_IL ? FLAG_IL : FLAG_OL;
  }
  void edge(uint32_t, uint32_t, uint64_t) {}
};
struct MainSink {
  HS& t;
  uint32_t node_word(uint32_t n) { return t.word[n]; }
  void nc(uint32_t n) { ++t.counters[n]; }
  void common(uint32_t n, uint32_t word, uint32_t side, uint32_t times, uint64_t order) {
    const size_t V = t.g->V;
    t.counters[(side == SIDE_IL ? V : 2 * V) + n] += times;
    if (word & (side == SIDE_IL ? FLAG_IL : FLAG_OL)) {
      auto& f = t.common_first.emplace(std::make_pair(n, side), UINT64_MAX).first->second;
      f = std::min(f, order);
    }
  }
  void rare(uint32_t, uint32_t, int32_t, uint64_t) {}
  void edge(uint32_t a, uint32_t b, uint64_t order) {
    const Graph& g = *t.g;
    for (uint32_t s = g.row_ptr[a]; s < g.row_ptr[a + 1]; ++s)
      if (g.col[s] == b) { ++t.counters[3 * size_t(g.V) + s]; return; }
    auto& e = t.novel.emplace(std::make_pair(a, b), std::make_pair(0u, UINT64_MAX)).first->second;
    e.first += 1;
    e.second = std::min(e.second, order);
  }
};

template <class F>
int guard(HS* h, F&& f) {
  try { f(); return 0; } catch (const std::exception& e) { if (h) h->err = e.what(); return -1; }
}

template <class Sink>
void pass(HS& h, bool only_del) {
  for (size_t i = 0; i < h.b.size(); ++i) {
    const LineHdr& hd = h.b.hdr[i];
    if (only_del && !(hd.flags & LINE_HAS_DEL)) continue;
    Sink s{h};
    const uint32_t* nodes = h.b.nodes.data() + h.b.node_off[i];
    const uint32_t* ops = h.b.ops.data() + h.b.op_off[i];
    const uint32_t k = h.b.node_off[i + 1] - h.b.node_off[i], c = h.b.op_off[i + 1] - h.b.op_off[i];
    int st = only_del ? rare_walk(*h.g, h.word, hd, nodes, k, ops, c, s) : main_walk(*h.g, h.word, hd, nodes, k, ops, c, s);
    if (st != WALK_OK) throw Error("IndexError", "cs ops exhausted");
  }
}
}  // namespace

extern "C" {
HS* hs_open(const char* gfa) {
  auto h = std::make_unique<HS>();
  try { h->g = load_graph(gfa, 2); } catch (...) { return nullptr; }
  h->word.assign(h->g->seg_len.begin(), h->g->seg_len.end());
  h->counters.assign(3 * size_t(h->g->V) + h->g->A, 0);
  return h.release();
}
void hs_close(HS* h) { delete h; }
const char* hs_error(HS* h) { return h->err.c_str(); }
int hs_load_shard(HS* h, const char* gaf, int thr, uint64_t begin, uint64_t end, uint64_t first_line, uint64_t* stats3) {
  return guard(h, [&] {
    MappedFile f;
    f.open(gaf);
    if (end == 0 || end > f.size) end = f.size;
    h->b = GafBatch();
    tokenize_gaf(*h->g, f.data + begin, size_t(end - begin), thr, first_line, 2, h->b);
    if (h->b.deferred) throw *h->b.deferred;
    bucket_by_locus(h->b, 2);
    stats3[0] = h->b.lines_total; stats3[1] = h->b.rejected; stats3[2] = h->b.del_nodes;
  });
}
int hs_rare(HS* h) { return guard(h, [&] { pass<RareSink>(*h, true); }); }
int hs_main(HS* h) { return guard(h, [&] { pass<MainSink>(*h, false); }); }
uint64_t hs_flags_export(HS* h, uint32_t* codes, uint64_t cap) {
  uint64_t n = 0;
  for (uint32_t i = 0; i < h->g->V; ++i) {
    if (h->word[i] & FLAG_IL) { if (n < cap) codes[n] = i << 1; ++n; }
    if (h->word[i] & FLAG_OL) { if (n < cap) codes[n] = (i << 1) | 1u; ++n; }
  }
  return n;
}
void hs_flags_import(HS* h, const uint32_t* codes, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) h->word[codes[i] >> 1] |= (codes[i] & 1u) ? FLAG_OL : FLAG_IL;
}
uint32_t* hs_counters(HS* h, uint64_t* words) { *words = h->counters.size(); return h->counters.data(); }
int hs_finish(HS* h) {
  return guard(h, [&] {
    const size_t V = h->g->V;
    AugmentResult& r = h->res;
    r = AugmentResult();
    r.nc.assign(h->counters.begin(), h->counters.begin() + V);
    r.il0.assign(h->counters.begin() + V, h->counters.begin() + 2 * V);
    r.oll.assign(h->counters.begin() + 2 * V, h->counters.begin() + 3 * V);
    r.rc.assign(h->counters.begin() + 3 * V, h->counters.end());
    for (auto& kv : h->rare)
      r.hist.push_back(HistEntry{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), kv.second.first, kv.second.second, false});
    for (auto& kv : h->common_first) r.hist.push_back(HistEntry{kv.first.first, kv.first.second, 0, 0, kv.second, true});
    for (auto& kv : h->novel) r.novel.push_back(NovelLink{kv.first.first, kv.first.second, kv.second.first, kv.second.second});
    finalize_result(r);
  });
}
uint64_t hs_sparse_count(HS* h) { return h->res.hist.size() + h->res.novel.size(); }
void hs_sparse_export(HS* h, uint64_t* p) {
  for (const HistEntry& e : h->res.hist) {
    uint32_t code = e.is_common ? 0x7FFFFFFFu : (uint32_t(e.pos + 1) & 0x7FFFFFFFu);
    *p++ = (uint64_t(e.node) << 32) | (uint64_t(e.side) << 31) | code; *p++ = e.count; *p++ = e.first; *p++ = 0;
  }
  for (const NovelLink& n : h->res.novel) { *p++ = (uint64_t(n.u) << 32) | n.v; *p++ = n.count; *p++ = n.first; *p++ = 1; }
}
int hs_sparse_import(HS* h, const uint64_t* rows, uint64_t n) {
  return guard(h, [&] {
    // same merge rule as the product (capi.cpp pantas_augment_sparse_import): counts add, first-seen = min
    std::map<std::tuple<uint32_t, uint32_t, int, int32_t>, std::pair<uint32_t, uint64_t>> H;
    std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint64_t>> N;
    for (const HistEntry& e : h->res.hist) {
      auto& v = H.emplace(std::make_tuple(e.node, e.side, e.is_common ? 1 : 0, e.pos), std::make_pair(0u, UINT64_MAX)).first->second;
      v.first += e.count; v.second = std::min(v.second, e.first);
    }
    for (const NovelLink& e : h->res.novel) {
      auto& v = N.emplace(std::make_pair(e.u, e.v), std::make_pair(0u, UINT64_MAX)).first->second;
      v.first += e.count; v.second = std::min(v.second, e.first);
    }
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t key = rows[4 * i], cnt = rows[4 * i + 1], first = rows[4 * i + 2], kind = rows[4 * i + 3];
      if (kind == 0) {
        uint32_t code = uint32_t(key) & 0x7FFFFFFFu;
        bool common = code == 0x7FFFFFFFu;
        auto& v = H.emplace(std::make_tuple(uint32_t(key >> 32), uint32_t(key >> 31) & 1u, common ? 1 : 0, common ? 0 : int32_t(code) - 1),
                            std::make_pair(0u, UINT64_MAX)).first->second;
        v.first += uint32_t(cnt); v.second = std::min(v.second, first);
      } else {
        auto& v = N.emplace(std::make_pair(uint32_t(key >> 32), uint32_t(key)), std::make_pair(0u, UINT64_MAX)).first->second;
        v.first += uint32_t(cnt); v.second = std::min(v.second, first);
      }
    }
    h->res.hist.clear();
    h->res.novel.clear();
    for (auto& kv : H) h->res.hist.push_back(HistEntry{std::get<0>(kv.first), std::get<1>(kv.first), std::get<3>(kv.first), kv.second.first, kv.second.second, std::get<2>(kv.first) == 1});
    for (auto& kv : N) h->res.novel.push_back(NovelLink{kv.first.first, kv.first.second, kv.second.first, kv.second.second});
    finalize_result(h->res);
  });
}
int hs_write(HS* h, int fd) { return guard(h, [&] { write_augmented_gfa(*h->g, h->res, fd, 2); }); }
}
