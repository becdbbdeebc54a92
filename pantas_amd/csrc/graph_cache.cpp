// Binary cache of a parsed GFA: the graph is shared by every replicate of an experiment
// (reference exps/*/pantas.smk runs `pantas augment` per sample on one graph), parsing its text is the largest
// host cost of a run (19 GB -> 11 s at the S4 scale), and everything the parse produces is plain arrays.
// The cache holds those arrays; the GFA text itself is still mapped (the second pass echoes its lines) but
// not read.  A cache is only used for the exact file it was made from (size, mtime, a hash of both ends and of 4 KB
// samples spread over the file), and its arrays are range-checked before anything indexes with them: a stale or damaged
// cache is parsed again, never trusted.
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>

#include "graph.h"

namespace pantas {

namespace {

constexpr uint64_t MAGIC = 0x3143475041544E50ull;   // "PNTAPGC1"
constexpr uint64_t CACHE_VERSION = 2;

struct Header {
  uint64_t magic, version;
  uint64_t gfa_size, gfa_mtime_ns, gfa_ends_hash;
  uint64_t V, A, n_lines, n_s_lines, n_l_lines, n_p_lines;
  uint64_t names_base, names_size;
  uint64_t links_unknown_end, first_bad_p_off, first_short_l_off, ws_only_seg_off;
  uint32_t max_seg_len;
  uint8_t dup_names, link_before_segment, pad_[2];
  uint64_t sections;      // number of arrays that follow: each {uint64 bytes; data; pad to 64}
};

uint64_t ends_hash(const char* data, size_t size) {
  uint64_t h = 1469598103934665603ull ^ size;
  auto mix = [&](const char* p, size_t n) {
    for (size_t i = 0; i < n; ++i) { h ^= uint8_t(p[i]); h *= 1099511628211ull; }
  };
  const size_t span = std::min<size_t>(size, 65536);
  mix(data, span);
  mix(data + size - span, span);
  // 4 KB out of every 16 MB in between: an edit in the middle of the file that keeps its size is caught with the
  // probability of hitting a sample -- and whatever slips through still has to pass the range checks of the loader
  constexpr size_t STEP = size_t(16) << 20, SAMPLE = 4096;
  for (size_t at = STEP; at + SAMPLE <= size; at += STEP) mix(data + at, SAMPLE);
  return h;
}

uint64_t mtime_ns(const char* path) {
  struct stat st;
  if (stat(path, &st) != 0) return 0;
  return uint64_t(st.st_mtim.tv_sec) * 1000000000ull + uint64_t(st.st_mtim.tv_nsec);
}

struct Section { const void* p; size_t bytes; };

template <class V> Section sec(const V& v) { return Section{v.data(), v.size() * sizeof(typename V::value_type)}; }

}  // namespace

bool save_graph_cache(const Graph& g, const char* cache_path) {
  if (!g.names.dense || g.text.borrowed || g.path.empty()) return false;   // only file-backed graphs with decimal names
  if (g.dup_names) return false;                                           // (several S lines for one name: parsed every time)
  Header h{};
  h.magic = MAGIC; h.version = CACHE_VERSION;
  h.gfa_size = g.text.size;
  h.gfa_mtime_ns = mtime_ns(g.path.c_str());
  h.gfa_ends_hash = ends_hash(g.text.data, g.text.size);
  h.V = g.V; h.A = g.A; h.n_lines = g.n_lines; h.n_s_lines = g.n_s_lines; h.n_l_lines = g.n_l_lines; h.n_p_lines = g.n_p_lines;
  h.names_base = g.names.base; h.names_size = g.names.table.size();
  h.links_unknown_end = g.links_unknown_end; h.first_bad_p_off = g.first_bad_p_off; h.first_short_l_off = g.first_short_l_off;
  h.ws_only_seg_off = g.ws_only_seg_off;
  h.max_seg_len = g.max_seg_len; h.dup_names = g.dup_names; h.link_before_segment = g.link_before_segment;
  const Section secs[] = {sec(g.names.table), sec(g.seg_len), sec(g.seg_line_off), sec(g.seg_name_off), sec(g.seg_name_len),
                          sec(g.row_ptr), sec(g.col), sec(g.slot_off), sec(g.irow_ptr), sec(g.icol), sec(g.islot)};
  h.sections = sizeof(secs) / sizeof(secs[0]);
  const std::string tmp = std::string(cache_path) + ".tmp" + std::to_string(uint64_t(getpid()));
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) throw Error("OSError", "cannot write graph cache " + tmp);
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  static const char zeros[64] = {0};
  size_t pos = sizeof h;
  auto pad = [&] { const size_t r = (64 - pos % 64) % 64; if (r) { ok = ok && fwrite(zeros, 1, r, f) == r; pos += r; } };
  pad();
  for (const Section& s : secs) {
    const uint64_t n = s.bytes;
    ok = ok && fwrite(&n, 8, 1, f) == 1;
    pos += 8;
    pad();
    if (n) ok = ok && fwrite(s.p, 1, n, f) == n;
    pos += n;
    pad();
  }
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), cache_path) != 0) {
    remove(tmp.c_str());
    throw Error("OSError", std::string("cannot write graph cache ") + cache_path);
  }
  return true;
}

std::unique_ptr<Graph> load_graph_cached(const char* gfa_path, const char* cache_path, unsigned threads) {
  PhaseTimer pt("load_graph_cached");
  if (threads == 0) threads = hw_threads();
  MappedFile c;
  try {
    c.open(cache_path);
  } catch (const Error&) {
    return nullptr;
  }
  if (c.size < sizeof(Header)) return nullptr;
  Header h;
  memcpy(&h, c.data, sizeof h);
  if (h.magic != MAGIC || h.version != CACHE_VERSION || h.sections != 11) return nullptr;
  auto g = std::make_unique<Graph>();
  g->path = gfa_path;
  g->text.open(gfa_path);
  if (g->text.size != h.gfa_size || mtime_ns(gfa_path) != h.gfa_mtime_ns || ends_hash(g->text.data, g->text.size) != h.gfa_ends_hash)
    return nullptr;   // made from another file (or the file changed): parse again
  g->V = uint32_t(h.V); g->A = h.A; g->n_lines = h.n_lines; g->n_s_lines = h.n_s_lines; g->n_l_lines = h.n_l_lines; g->n_p_lines = h.n_p_lines;
  g->names.dense = true; g->names.base = h.names_base;
  g->links_unknown_end = h.links_unknown_end; g->first_bad_p_off = h.first_bad_p_off; g->first_short_l_off = h.first_short_l_off;
  g->ws_only_seg_off = h.ws_only_seg_off;
  g->max_seg_len = h.max_seg_len; g->dup_names = h.dup_names; g->link_before_segment = h.link_before_segment;
  size_t pos = (sizeof h + 63) / 64 * 64;
  bool bad = false;
  auto take = [&](auto& vec, uint64_t expect_elems) {
    using T = typename std::remove_reference_t<decltype(vec)>::value_type;
    if (bad || pos + 8 > c.size) { bad = true; return; }
    uint64_t n;
    memcpy(&n, c.data + pos, 8);
    pos = (pos + 8 + 63) / 64 * 64;
    if (n != expect_elems * sizeof(T) || pos + n > c.size) { bad = true; return; }
    vec.resize(expect_elems);
    const char* src = c.data + pos;
    char* dst = reinterpret_cast<char*>(vec.data());
    parallel_chunks(n, n < (size_t(8) << 20) ? 1u : threads, [&](unsigned, size_t b, size_t e) { memcpy(dst + b, src + b, e - b); });
    pos = (pos + n + 63) / 64 * 64;
  };
  take(g->names.table, h.names_size);
  take(g->seg_len, h.V);
  take(g->seg_line_off, h.V);
  take(g->seg_name_off, h.V);
  take(g->seg_name_len, h.V);
  take(g->row_ptr, h.V + 1);
  take(g->col, h.A);
  take(g->slot_off, h.A);
  take(g->irow_ptr, h.V + 1);
  take(g->icol, h.A);
  take(g->islot, h.A);
  if (bad) return nullptr;
  pt.lap("arrays from cache");
  // ---- range checks: everything that is later used as an index or an offset (a damaged cache must not become an
  // out-of-bounds access on the host or, worse, in a kernel)
  {
    const uint64_t V = h.V, A = h.A, T = g->text.size;
    if (V >= NONE32 || A >= NONE32) return nullptr;
    std::atomic<bool> ok{true};
    auto fail = [&] { ok.store(false, std::memory_order_relaxed); };
    if (g->row_ptr[0] != 0 || g->row_ptr[V] != A || g->irow_ptr[0] != 0 || g->irow_ptr[V] != A) return nullptr;
    parallel_chunks(V, V < (1u << 16) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        if (g->row_ptr[i] > g->row_ptr[i + 1] || g->irow_ptr[i] > g->irow_ptr[i + 1]) fail();
        if (g->seg_line_off[i] >= T || g->seg_name_off[i] + g->seg_name_len[i] > T) fail();
        if ((g->seg_len[i] >> 30) != 0) fail();
      }
    });
    parallel_chunks(A, A < (1u << 16) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i)
        if (g->col[i] >= V || g->icol[i] >= V || g->islot[i] >= A || g->slot_off[i] >= T) fail();
    });
    const size_t NT = g->names.table.size();
    parallel_chunks(NT, NT < (1u << 16) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) { const uint32_t n = g->names.table[i]; if (n != NONE32 && n >= V) fail(); }
    });
    if (!ok.load()) return nullptr;
  }
  pt.lap("range checks");
  return g;
}

}  // namespace pantas
