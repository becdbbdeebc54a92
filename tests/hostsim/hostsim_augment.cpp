// TEST HARNESS ONLY -- not part of the product and never loaded by it.
// Compiles the device walk (pantas_amd/csrc/walk.h) together with the product's host-side
// loader, tokeniser and writer for the CPU, under ASan/UBSan, and drives the walk with a
// sink that mimics the two-kernel protocol of augment.hip (rare-key pass that sets the
// node flags, then the main pass; lines of the short-walk class go through walk_short / walk_short_rare
// like on the GPU, HOSTSIM_GENERAL_WALK=1 sends everything through walk_line).  It lets the `-m "not gpu"` suite check the exact
// device-side source against the reference goldens where no GPU exists.
//
// usage: hostsim_augment <gaf> <gfa> [thr] > out.gfa     (stderr: "Rejected alignments: N")
#include <cstdio>
#include <map>
#include <tuple>
#include "augment.h"
#include "walk.h"
#include "short_walk_host.h"

using namespace pantas;

struct Tables {
  std::vector<uint32_t> word, nc, rc, il0, oll;
  std::map<std::tuple<uint32_t, uint32_t, int32_t>, std::pair<uint32_t, uint64_t>> rare;
  std::map<std::pair<uint32_t, uint32_t>, uint64_t> common_first;
  std::map<std::pair<uint32_t, uint32_t>, std::pair<uint32_t, uint64_t>> novel;
};

struct RareSink {   // first pass: only lines with a '-' op
  Tables& t;
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
  Tables& t;
  const Graph& g;
  uint32_t node_word(uint32_t n) { return t.word[n]; }
  void nc(uint32_t n) { ++t.nc[n]; }
  void common(uint32_t n, uint32_t word, uint32_t side, uint32_t times, uint64_t order) {
    (side == SIDE_IL ? t.il0 : t.oll)[n] += times;
    if (word & (side == SIDE_IL ? FLAG_IL : FLAG_OL)) {
      auto& f = t.common_first.emplace(std::make_pair(n, side), UINT64_MAX).first->second;
      f = std::min(f, order);
    }
  }
  void rare(uint32_t, uint32_t, int32_t, uint64_t) {}
  void edge(uint32_t a, uint32_t b, uint64_t order) {
    for (uint32_t s = g.row_ptr[a]; s < g.row_ptr[a + 1]; ++s)
      if (g.col[s] == b) { ++t.rc[s]; return; }
    auto& e = t.novel.emplace(std::make_pair(a, b), std::make_pair(0u, UINT64_MAX)).first->second;
    e.first += 1;
    e.second = std::min(e.second, order);
  }
};

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s gaf gfa [thr]\n", argv[0]); return 2; }
  try {
    int thr = argc > 3 ? atoi(argv[3]) : 20;
    unsigned threads = getenv("HOSTSIM_THREADS") ? unsigned(atoi(getenv("HOSTSIM_THREADS"))) : 3;
    fprintf(stderr, "Read GFA\n");
    std::unique_ptr<Graph> g;
    if (const char* cache = getenv("HOSTSIM_GRAPH_CACHE")) {   // the binary graph cache (graph_cache.cpp) in front of the parser
      g = load_graph_cached(argv[2], cache, threads);
      fprintf(stderr, "graph cache: %s\n", g ? "used" : "parsed");
      if (!g) {
        g = load_graph(argv[2], threads);
        save_graph_cache(*g, cache);
      }
    } else {
      g = load_graph(argv[2], threads);
    }
    require_augmentable(*g);
    fprintf(stderr, "Augmentation by GAF alignments\n");
    MappedFile gaf;
    gaf.open(argv[1]);
    GafBatch b;
    tokenize_gaf(*g, gaf.data, gaf.size, thr, 0, threads, b);
    RawVector<char> universal;
    if (b.deferred && is_lone_cr_error(*b.deferred)) {      // as pantas_augment_stream_rare: the text again, with universal newlines applied
      universal_newlines_copy(gaf.data, gaf.size, universal);
      b = GafBatch();
      tokenize_gaf(*g, universal.data(), universal.size(), thr, 0, threads, b);
    }
    const uint64_t first_record_line = b.first_record_line;
    // lines whose first cs op is '~' inherit a length from the line in front (walk.h TildeCarry): as pantas_augment_run, from the text
    const char* text = universal.empty() ? gaf.data : universal.data();
    const std::vector<TildeEntry> tilde = tilde_carries(*g, text, gaf.size, thr, 0);
    const TildeCarry tc{tilde.data(), uint32_t(tilde.size())};
    if (getenv("HOSTSIM_BUCKET")) bucket_by_locus(b, threads);
    Tables t;
    t.word.assign(g->seg_len.begin(), g->seg_len.end());
    t.nc.assign(g->V, 0);
    t.il0.assign(g->V, 0);
    t.oll.assign(g->V, 0);
    t.rc.assign(g->A, 0);
    for (int pass = 0; pass < 2; ++pass) {
      for (size_t i = 0; i < b.size(); ++i) {
        const LineHdr& h = b.hdr[i];
        if (pass == 0 && !(h.flags & LINE_HAS_DEL)) continue;
        int st;
        const uint32_t* nodes = b.nodes.data() + b.node_off[i];
        const uint32_t* ops = b.ops.data() + b.op_off[i];
        uint32_t k = b.node_off[i + 1] - b.node_off[i], c = b.op_off[i + 1] - b.op_off[i];
        // as the kernels dispatch: short-walk class through walk.h's short walks, the rest through walk_line
        if (pass == 0) { RareSink s{t}; st = rare_walk(*g, t.word, h, nodes, k, ops, c, s, &tc); }
        else { MainSink s{t, *g}; st = main_walk(*g, t.word, h, nodes, k, ops, c, s, &tc); }
        if (st == WALK_OPS_EXHAUSTED) throw Error("IndexError", "GAF line " + std::to_string(h.line_id + 1) + ": cs ops exhausted before the path end");
        if (st == WALK_TILDE_FIRST) {
          (void)first_record_line;
          throw Error("UnboundLocalError", "GAF line " + std::to_string(h.line_id + 1) + ": '~' as the first op the walk reaches and no line before it left a length behind");
        }
      }
    }
    AugmentResult r;
    r.nc.assign(t.nc.begin(), t.nc.end()); r.rc.assign(t.rc.begin(), t.rc.end()); r.il0.assign(t.il0.begin(), t.il0.end()); r.oll.assign(t.oll.begin(), t.oll.end());
    for (auto& kv : t.rare)
      r.hist.push_back(HistEntry{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), kv.second.first, kv.second.second, false});
    for (auto& kv : t.common_first)
      r.hist.push_back(HistEntry{kv.first.first, kv.first.second, 0, 0, kv.second, true});
    for (auto& kv : t.novel) r.novel.push_back(NovelLink{kv.first.first, kv.first.second, kv.second.first, kv.second.second});
    if (b.deferred) throw *b.deferred;      // no earlier line failed in the walk: the tokeniser's error stands
    finalize_result(r);
    fprintf(stderr, "Rejected alignments: %llu\n", (unsigned long long)b.rejected);
    fprintf(stderr, "Annotating GFA\n");
    write_augmented_gfa(*g, r, 1, threads);
    return 0;
  } catch (const Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
