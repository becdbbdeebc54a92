#include "augment.h"
#include <sys/mman.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <memory>
#include <mutex>
#include <thread>

namespace pantas {

void finalize_result(AugmentResult& r) {
  // (first-seen orders are unique: no ties for the order of equal keys to matter)
  par_sort(r.hist.begin(), r.hist.end(), [](const HistEntry& a, const HistEntry& b) {
    if (a.node != b.node) return a.node < b.node;
    if (a.side != b.side) return a.side < b.side;
    return a.first < b.first;
  });
  par_sort(r.novel.begin(), r.novel.end(), [](const NovelLink& a, const NovelLink& b) { return a.first < b.first; });
}

namespace {

inline void put_u64(std::string& s, uint64_t v) {
  char buf[24];
  int n = 0;
  do { buf[n++] = char('0' + v % 10); v /= 10; } while (v);
  while (n) s.push_back(buf[--n]);
}

inline void put_i64(std::string& s, int64_t v) {
  if (v < 0) { s.push_back('-'); put_u64(s, uint64_t(-v)); } else put_u64(s, uint64_t(v));
}

struct KeyCount { int64_t key; uint64_t count; uint64_t first; };

// one histogram tag (IL or OL) of node `n`; appends nothing when the histogram is empty
void put_hist(std::string& s, const Graph& g, const AugmentResult& r, uint32_t n, uint32_t side,
              const HistEntry* lo, const HistEntry* hi) {
  const uint32_t common = side == SIDE_IL ? r.il0[n] : r.oll[n];
  const int64_t common_key = side == SIDE_IL ? 0 : int64_t(g.seg_len[n]);
  const char* tag = side == SIDE_IL ? "\tIL:Z:" : "\tOL:Z:";
  // entries of this side
  while (lo < hi && lo->side != side) ++lo;
  const HistEntry* e = lo;
  while (e < hi && e->side == side) ++e;
  if (lo == e) {
    if (common) { s += tag; put_i64(s, common_key); s.push_back('.'); put_u64(s, common); }
    return;
  }
  KeyCount kc[64];
  std::vector<KeyCount> big;
  KeyCount* v = kc;
  size_t cnt = 0;
  const size_t need = size_t(e - lo) + 1;
  if (need > 64) { big.resize(need); v = big.data(); }
  bool common_seen = false;
  for (const HistEntry* x = lo; x < e; ++x) {
    int64_t key = x->is_common ? common_key : int64_t(x->pos);
    uint64_t c = x->is_common ? 0 : x->count;
    if (x->is_common) common_seen = true;
    size_t j = 0;
    for (; j < cnt; ++j) if (v[j].key == key) break;
    if (j == cnt) v[cnt++] = KeyCount{key, c, x->first};
    else { v[j].count += c; v[j].first = std::min(v[j].first, x->first); }
  }
  if (common) {
    size_t j = 0;
    for (; j < cnt; ++j) if (v[j].key == common_key) break;
    if (j == cnt) {
      // only reachable if the kernels failed to record the order of the common key
      if (!common_seen) throw Error("InternalError", "IL/OL common key without first-seen order on a flagged node");
      v[cnt++] = KeyCount{common_key, common, 0};
    } else v[j].count += common;
  }
  std::sort(v, v + cnt, [](const KeyCount& a, const KeyCount& b) { return a.first < b.first; });
  bool any = false;
  for (size_t j = 0; j < cnt; ++j) {
    if (v[j].count == 0) continue;
    s += any ? "," : tag;
    any = true;
    put_i64(s, v[j].key);
    s.push_back('.');
    put_u64(s, v[j].count);
  }
}

// has_hist: one bit per node that owns explicit IL/OL entries (a few % of the nodes: the binary search below is for them only)
void format_range(const Graph& g, const AugmentResult& r, const std::vector<uint64_t>& has_hist, size_t b, size_t e, std::string& out) {
  const char* base = g.text.data;
  const char* p = base + b;
  const char* end = base + e;
  out.reserve((e - b) + (e - b) / 4);
  // Node indices follow the order of the S lines and, in a graph written by vg / the annotator, adjacency slots follow the
  // order of the L lines: the node / slot of the NEXT S / L line is then the one after the last -- checked against the
  // line offsets the loader kept, with the name lookup as the fallback.  That turns the three or four dependent random
  // reads per line (name table, row bounds, slot offsets, counters: a cache miss each on a whole-genome graph) into
  // sequential ones.
  size_t n_exp = size_t(std::lower_bound(g.seg_line_off.begin(), g.seg_line_off.end(), uint64_t(b)) - g.seg_line_off.begin());
  size_t s_exp = size_t(std::lower_bound(g.slot_off.begin(), g.slot_off.end(), uint64_t(b)) - g.slot_off.begin());
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
    const char* le = nl ? nl : end;
    Span st = strip(Span{p, le});
    p = nl ? nl + 1 : end;
    if (st.size() && *st.b == 'S') {
      uint32_t n;
      if (n_exp < g.seg_line_off.size() && g.seg_line_off[n_exp] == uint64_t(st.b - base)) {
        n = uint32_t(n_exp++);
      } else {
        Tokens tk(st);
        Span t0, name;
        tk.next(t0);
        tk.next(name);
        n = g.names.find(name);
        if (n == NONE32) throw Error("KeyError", "S line names an unknown segment");
        n_exp = size_t(n) + 1;
      }
      out.append(st.b, st.size());
      out += "\tNC:i:";
      put_u64(out, r.nc[n]);
      // explicit entries of this node, if any
      const HistEntry* lo = nullptr;
      const HistEntry* hi = nullptr;
      if ((has_hist[n >> 6] >> (n & 63u)) & 1u) {
        auto cmp = [](const HistEntry& a, uint32_t node) { return a.node < node; };
        const HistEntry* hb = r.hist.data();
        const HistEntry* he = hb + r.hist.size();
        lo = std::lower_bound(hb, he, n, cmp);
        hi = lo;
        while (hi < he && hi->node == n) ++hi;
      }
      put_hist(out, g, r, n, SIDE_IL, lo, hi);
      put_hist(out, g, r, n, SIDE_OL, lo, hi);
      out.push_back('\n');
    } else if (st.size() && *st.b == 'L') {
      if (st.size() == 1) continue;            // bare "L": dropped (:418-419)
      uint32_t w = 0;
      const uint64_t off = uint64_t(st.b - base);
      if (s_exp < g.slot_off.size() && g.slot_off[s_exp] == off) {
        w = r.rc[s_exp++];
      } else {
        Tokens tk(st);
        Span t[4];
        for (int i = 0; i < 4; ++i) tk.next(t[i]);
        uint32_t u = g.names.find(t[1]), v = g.names.find(t[3]);
        if (u != NONE32 && v != NONE32) {
          for (uint32_t s = g.row_ptr[u]; s < g.row_ptr[u + 1]; ++s)
            if (g.slot_off[s] == off) { w = r.rc[s]; s_exp = size_t(s) + 1; break; }
        }
      }
      out.append(st.b, st.size());
      out += "\tRC:i:";
      put_u64(out, w);
      out.push_back('\n');
    } else {
      out.append(st.b, st.size());
      out.push_back('\n');
    }
  }
}

}  // namespace

// The GFA text is cut into pieces of a few MB at line ends; worker threads claim pieces in file order, format them into
// buffers they keep, and hand the output position down the chain: piece p may be written once piece p - 1 knows where
// it ends.  On a regular file (the `> out.gfa` of the command line) that is known as soon as p - 1 is FORMATTED, and the
// pieces are written concurrently with pwrite; on a pipe the chain advances after each write.  (Formatting 19 GB in
// rounds and writing them from one thread took 10 of the 22 s of `pantas augment` at whole-genome scale.)
void write_augmented_gfa(const Graph& g, const AugmentResult& r, int fd, unsigned threads) {
  PhaseTimer pt("write_augmented_gfa");
  if (threads == 0) threads = writer_threads();
  const size_t size = g.text.size;
  const char* base = g.text.data;
  constexpr size_t PIECE = size_t(4) << 20;
  // piece boundaries: the line end at or behind every multiple of PIECE
  std::vector<size_t> cut;
  cut.push_back(0);
  {
    const size_t n_guess = size / PIECE + 2;
    std::vector<size_t> c(n_guess, size);
    parallel_chunks(n_guess, n_guess < 64 ? 1u : threads, [&](unsigned, size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        const size_t want = (i + 1) * PIECE;
        if (want >= size) { c[i] = size; continue; }
        const void* nl = memchr(base + want, '\n', size - want);
        c[i] = nl ? size_t(static_cast<const char*>(nl) - base) + 1 : size;
      }
    });
    for (size_t x : c) if (x > cut.back()) cut.push_back(x);
    if (cut.back() != size) cut.push_back(size);
  }
  const size_t n_pieces = cut.size() - 1;
  std::vector<uint64_t> has_hist((size_t(g.V) + 63) / 64 + 1, 0);
  for (const HistEntry& h : r.hist) has_hist[h.node >> 6] |= uint64_t(1) << (h.node & 63u);
  // the novel-link tail is the last piece
  std::string tail;
  for (const NovelLink& nl : r.novel) {
    tail += "L\t";
    Span a = g.name_of(nl.u), b = g.name_of(nl.v);
    tail.append(a.b, a.size());
    tail += "\t+\t";
    tail.append(b.b, b.size());
    tail += "\t+\t*\tRC:i:";
    put_u64(tail, nl.count);
    tail += "\tID:Z:N\n";
  }
  // what the lines can grow by at most: NC (16), a common-only IL (18) and OL (27) per S line, RC (16) per L line, 24 per
  // explicit histogram entry -- the (sparse) size the output file is mapped with; about a fifth more than the text is expected
  const uint64_t bound = uint64_t(size) + 61ull * g.n_s_lines + 16ull * g.n_l_lines + 24ull * r.hist.size() + tail.size() + 4096;
  OrderedOutput sink(fd, n_pieces + 1, uint64_t(size) + uint64_t(size) / 4 + tail.size(), bound);
  const unsigned T = unsigned(std::min<size_t>(std::max<size_t>(n_pieces, 1), threads));
  const bool drop_text = g.text.fd >= 0 && !g.text.borrowed && g.text_owned.empty() && size >= (size_t(1) << 30) && !getenv("PANTAS_KEEP_TEXT_MAPPED");
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::shared_ptr<Error> first_err;
  std::mutex err_mu;
  auto worker = [&] {
    std::string buf;
    for (;;) {
      const size_t p = next.fetch_add(1);
      if (p >= n_pieces) break;
      try {
        buf.clear();
        if (!failed.load(std::memory_order_relaxed)) format_range(g, r, has_hist, cut[p], cut[p + 1], buf);
        // this stretch of the mapped text has been read for the last time: its page-table entries go now, by the thread that is about
        // to wait for the writer anyway, instead of all 23 GB of them at exit, where nothing runs beside it (profiles/r2_exit_probe.txt)
        if (drop_text) {
          const uintptr_t page = 4096, b0 = (uintptr_t(g.text.data) + cut[p] + page - 1) / page * page, b1 = (uintptr_t(g.text.data) + cut[p + 1]) / page * page;
          if (b1 > b0) (void)madvise(reinterpret_cast<void*>(b0), b1 - b0, MADV_DONTNEED);
        }
      } catch (const Error& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!first_err) first_err = std::make_shared<Error>(e);
        failed.store(true);
        buf.clear();
      }
      try {
        if (failed.load(std::memory_order_relaxed)) buf.clear();
        sink.put(p, buf.data(), buf.size());      // (an empty piece still hands the position on)
      } catch (const Error& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!first_err) first_err = std::make_shared<Error>(e);
        failed.store(true);
      }
    }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(worker);
    worker();
    for (auto& x : th) x.join();
  }
  if (first_err) throw *first_err;
  // links seen in reads but absent from the GFA, in first-seen order (:426-427)
  sink.put(n_pieces, tail.data(), tail.size());
  sink.finish();
  pt.lap(sink.mode_name());
}

}  // namespace pantas
