#include "augment.h"

#include <unistd.h>
#include <cerrno>

namespace pantas {

void finalize_result(AugmentResult& r) {
  std::sort(r.hist.begin(), r.hist.end(), [](const HistEntry& a, const HistEntry& b) {
    if (a.node != b.node) return a.node < b.node;
    if (a.side != b.side) return a.side < b.side;
    return a.first < b.first;
  });
  std::sort(r.novel.begin(), r.novel.end(), [](const NovelLink& a, const NovelLink& b) { return a.first < b.first; });
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

void write_all(int fd, const char* p, size_t n) {
  while (n) {
    ssize_t w = ::write(fd, p, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      throw Error("OSError", std::string("write failed: ") + strerror(errno));
    }
    p += w;
    n -= size_t(w);
  }
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

void format_range(const Graph& g, const AugmentResult& r, size_t b, size_t e, std::string& out) {
  const char* base = g.text.data;
  const char* p = base + b;
  const char* end = base + e;
  out.reserve((e - b) + (e - b) / 4);
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
    const char* le = nl ? nl : end;
    Span st = strip(Span{p, le});
    p = nl ? nl + 1 : end;
    if (st.size() && *st.b == 'S') {
      Tokens tk(st);
      Span t0, name;
      tk.next(t0);
      tk.next(name);
      uint32_t n = g.names.find(name);
      if (n == NONE32) throw Error("KeyError", "S line names an unknown segment");
      out.append(st.b, st.size());
      out += "\tNC:i:";
      put_u64(out, r.nc[n]);
      // explicit entries of this node, if any
      const HistEntry* lo = nullptr;
      const HistEntry* hi = nullptr;
      if (!r.hist.empty()) {
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
      Tokens tk(st);
      Span t[4];
      for (int i = 0; i < 4; ++i) tk.next(t[i]);
      uint32_t u = g.names.find(t[1]), v = g.names.find(t[3]);
      uint32_t w = 0;
      if (u != NONE32 && v != NONE32) {
        const uint64_t off = uint64_t(st.b - base);
        for (uint32_t s = g.row_ptr[u]; s < g.row_ptr[u + 1]; ++s)
          if (g.slot_off[s] == off) { w = r.rc[s]; break; }
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

void write_augmented_gfa(const Graph& g, const AugmentResult& r, int fd, unsigned threads) {
  if (threads == 0) threads = hw_threads();
  const size_t size = g.text.size;
  const size_t piece = 32u << 20;   // bytes of input per formatting task
  size_t done = 0;
  while (done < size) {
    size_t round_end = std::min(size, done + piece * threads);
    if (round_end < size) {
      const void* nl = memchr(g.text.data + round_end, '\n', size - round_end);
      round_end = nl ? size_t(static_cast<const char*>(nl) - g.text.data) + 1 : size;
    }
    auto ranges = split_at_newlines(g.text.data + done, round_end - done, (round_end - done) < (1u << 20) ? 1 : threads);
    std::vector<std::string> bufs(ranges.size());
    parallel_chunks(ranges.size(), unsigned(ranges.size()), [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) format_range(g, r, done + ranges[i].first, done + ranges[i].second, bufs[i]);
    });
    for (auto& s : bufs) write_all(fd, s.data(), s.size());
    done = round_end;
  }
  // links seen in reads but absent from the GFA, in first-seen order (:426-427)
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
    if (tail.size() > (8u << 20)) { write_all(fd, tail.data(), tail.size()); tail.clear(); }
  }
  write_all(fd, tail.data(), tail.size());
}

}  // namespace pantas
