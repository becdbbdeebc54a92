#include "synth.h"

#include <cmath>
#include <map>

namespace pantas {

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double prob() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
  // integer in [a, b]
  int64_t range(int64_t a, int64_t b) { return b <= a ? a : a + int64_t(next() % uint64_t(b - a + 1)); }
  template <class T> const T& pick(const std::vector<T>& v) { return v[size_t(next() % v.size())]; }
};

struct Slot { uint8_t bubble; uint8_t len; };
using Range = std::pair<uint32_t, uint32_t>;   // inclusive slot range, gene-local

struct Gene {
  std::vector<Slot> slots;
  std::vector<uint32_t> before;     // nodes before slot i (gene-local)
  std::vector<Range> exons;
  std::vector<std::vector<Range>> txs;
  uint64_t node_base = 0, bp_base = 0;
  uint32_t n_nodes = 0, chrom = 0;
  uint64_t bp_len = 0;
  char strand = '+';
};

struct Model {
  SynthParams p;
  std::vector<Gene> genes;          // genes.back() is the trailing intergenic stretch (no exons)
  int id_width = 0;
};

inline uint64_t node_id(const Gene& g, uint32_t slot, int hap) {
  return g.node_base + g.before[slot] + ((g.slots[slot].bubble && hap) ? 1u : 0u);
}
inline uint32_t node_len(const Gene& g, uint32_t slot) { return g.slots[slot].bubble ? 1u : g.slots[slot].len; }

void add_run(Gene& g, Rng& r, uint32_t n, bool bubbles, double rate) {
  for (uint32_t i = 0; i < n; ++i) {
    if (bubbles && i > 0 && i + 1 < n && !g.slots.back().bubble && r.prob() < rate) {
      g.slots.push_back(Slot{1, 1});
    } else {
      uint8_t ln = r.prob() < 0.8 ? 32 : uint8_t(r.range(1, 31));
      g.slots.push_back(Slot{0, ln});
    }
  }
}

void build_gene(const SynthParams& p, uint32_t gi, Gene& g) {
  Rng r(p.seed * 1000003ull + gi * 7919ull + 17);
  g.strand = (gi % 2 == 0) ? '+' : '-';
  add_run(g, r, uint32_t(r.range(p.intergenic_min, p.intergenic_max)), false, 0);
  if (gi < p.n_genes) {
    double x = -5.0 * std::log(1.0 - r.prob());
    uint32_t n_ex = std::min<uint32_t>(p.max_exons, std::max<uint32_t>(3, uint32_t(x) + 3));
    for (uint32_t e = 0; e < n_ex; ++e) {
      uint32_t ex_len = r.prob() < 0.85 ? uint32_t(r.range(2, 6)) : uint32_t(r.range(7, 12));
      uint32_t a = uint32_t(g.slots.size());
      add_run(g, r, ex_len, true, p.bubble_rate);
      g.exons.emplace_back(a, uint32_t(g.slots.size()) - 1);
      if (e + 1 != n_ex) {
        uint32_t in_len = r.prob() < 0.7 ? uint32_t(r.range(3, 9)) : uint32_t(r.range(10, std::max<uint32_t>(10, p.intron_long_max)));
        add_run(g, r, in_len, true, p.bubble_rate);
      }
    }
    const auto& ex = g.exons;
    g.txs.push_back(ex);                                    // full
    {                                                       // exon skip
      uint32_t k = uint32_t(r.range(1, n_ex - 2));
      std::vector<Range> t;
      for (uint32_t i = 0; i < n_ex; ++i) if (i != k) t.push_back(ex[i]);
      g.txs.push_back(t);
    }
    {                                                       // alternative donor: exon k ends early
      std::vector<uint32_t> cand;
      for (uint32_t i = 0; i + 1 < n_ex; ++i) if (ex[i].second - ex[i].first >= 2) cand.push_back(i);
      if (!cand.empty() && r.prob() < 0.8) {
        uint32_t k = r.pick(cand);
        int64_t cut = r.range(ex[k].first, ex[k].second - 1);
        while (cut >= int64_t(ex[k].first) && g.slots[size_t(cut)].bubble) --cut;
        if (cut >= int64_t(ex[k].first)) { auto t = ex; t[k].second = uint32_t(cut); g.txs.push_back(t); }
      }
    }
    {                                                       // alternative acceptor: exon k starts late
      std::vector<uint32_t> cand;
      for (uint32_t i = 1; i < n_ex; ++i) if (ex[i].second - ex[i].first >= 2) cand.push_back(i);
      if (!cand.empty() && r.prob() < 0.8) {
        uint32_t k = r.pick(cand);
        uint32_t cut = uint32_t(r.range(ex[k].first + 1, ex[k].second));
        while (cut <= ex[k].second && g.slots[cut].bubble) ++cut;
        if (cut <= ex[k].second) { auto t = ex; t[k].first = cut; g.txs.push_back(t); }
      }
    }
    if (r.prob() < 0.8) {                                   // intron retention
      uint32_t k = uint32_t(r.range(0, n_ex - 2));
      std::vector<Range> t(ex.begin(), ex.begin() + k);
      t.emplace_back(ex[k].first, ex[k + 1].second);
      t.insert(t.end(), ex.begin() + k + 2, ex.end());
      g.txs.push_back(t);
    }
    if (n_ex >= 5 && r.prob() < 0.5) {                      // double skip
      uint32_t k = uint32_t(r.range(1, n_ex - 3));
      std::vector<Range> t;
      for (uint32_t i = 0; i < n_ex; ++i) if (i != k && i != k + 1) t.push_back(ex[i]);
      g.txs.push_back(t);
    }
  }
  g.before.resize(g.slots.size());
  uint32_t n = 0;
  uint64_t bp = 0;
  for (size_t i = 0; i < g.slots.size(); ++i) {
    g.before[i] = n;
    n += g.slots[i].bubble ? 2u : 1u;
    bp += node_len(g, uint32_t(i));
  }
  g.n_nodes = n;
  g.bp_len = bp;
}

inline void put_u(std::string& s, uint64_t v) {
  char buf[24];
  int n = 0;
  do { buf[n++] = char('0' + v % 10); v /= 10; } while (v);
  while (n) s.push_back(buf[--n]);
}

void tx_name(std::string& s, uint32_t gi, uint32_t ti) { s += 'G'; put_u(s, gi + 1); s += 'T'; put_u(s, ti + 1); }

// node path (ids) of a transcript on haplotype hap, with exon numbers and node lengths
void tx_path(const Gene& g, const std::vector<Range>& exons, int hap, std::vector<uint64_t>& ids,
             std::vector<uint32_t>* exno, std::vector<uint32_t>* lens) {
  ids.clear();
  if (exno) exno->clear();
  if (lens) lens->clear();
  uint32_t k = 1;
  for (const Range& e : exons) {
    for (uint32_t s = e.first; s <= e.second; ++s) {
      ids.push_back(node_id(g, s, hap));
      if (exno) exno->push_back(k);
      if (lens) lens->push_back(node_len(g, s));
    }
    ++k;
  }
}

void emit_gene_gfa(const Model& m, uint32_t gi, std::string& s_out, std::string& l_out, uint64_t& n_links,
                   uint64_t& n_junc) {
  const Gene& g = m.genes[gi];
  Rng r(m.p.seed * 2000003ull + gi * 104729ull + 5);
  std::vector<std::string> ex(g.n_nodes);
  std::map<std::pair<uint64_t, uint64_t>, std::string> jn;
  std::vector<std::pair<uint64_t, uint64_t>> jn_order;
  std::vector<uint64_t> ids;
  std::vector<uint32_t> exno;
  static const char* HAP[2] = {"R1", "H1"};
  for (uint32_t ti = 0; ti < g.txs.size(); ++ti) {
    for (int hap = 0; hap < 2; ++hap) {
      tx_path(g, g.txs[ti], hap, ids, &exno, nullptr);
      std::string tag;
      tx_name(tag, gi, ti);
      tag += '_';
      tag += HAP[hap];
      for (size_t i = 0; i < ids.size(); ++i) {
        std::string& e = ex[size_t(ids[i] - g.node_base)];
        if (!e.empty()) e += ',';
        e += tag;
        e += '.';
        put_u(e, exno[i]);
        if (i + 1 < ids.size() && exno[i] != exno[i + 1]) {
          auto key = std::make_pair(ids[i], ids[i + 1]);
          auto it = jn.find(key);
          if (it == jn.end()) { it = jn.emplace(key, std::string()).first; jn_order.push_back(key); }
          if (!it->second.empty()) it->second += ',';
          it->second += tag;
          it->second += '.';
          put_u(it->second, exno[i]);
          it->second += '.';
          put_u(it->second, exno[i + 1]);
        }
      }
    }
  }
  static const char ACGT[4] = {'A', 'C', 'G', 'T'};
  for (uint32_t s = 0; s < g.slots.size(); ++s) {
    for (int al = 0; al < (g.slots[s].bubble ? 2 : 1); ++al) {
      const uint64_t id = node_id(g, s, al);
      s_out += "S\t";
      put_u(s_out, id);
      s_out += '\t';
      const uint32_t ln = node_len(g, s);
      for (uint32_t i = 0; i < ln; ++i) s_out += ACGT[r.next() & 3];
      const std::string& e = ex[size_t(id - g.node_base)];
      if (!e.empty()) { s_out += "\tEX:Z:"; s_out += e; }
      s_out += '\n';
    }
  }
  // links grouped by source node: backbone first, then junction links in first-appearance order
  std::map<uint64_t, std::vector<size_t>> jn_by_src;
  for (size_t i = 0; i < jn_order.size(); ++i) jn_by_src[jn_order[i].first].push_back(i);
  auto put_link = [&](uint64_t u, uint64_t v, const std::string* tags) {
    l_out += "L\t";
    put_u(l_out, u);
    l_out += "\t+\t";
    put_u(l_out, v);
    l_out += "\t+\t*";
    if (tags) { l_out += "\tJN:Z:"; l_out += *tags; ++n_junc; }
    l_out += '\n';
    ++n_links;
  };
  const bool last_region = gi + 1 == m.genes.size();
  std::vector<uint64_t> done;   // targets already written for the current source node
  for (uint32_t s = 0; s < g.slots.size(); ++s) {
    for (int al = 0; al < (g.slots[s].bubble ? 2 : 1); ++al) {
      const uint64_t u = node_id(g, s, al);
      done.clear();
      if (s + 1 < g.slots.size()) {
        for (int b = 0; b < (g.slots[s + 1].bubble ? 2 : 1); ++b) {
          const uint64_t v = node_id(g, s + 1, b);
          auto it = jn.find({u, v});
          put_link(u, v, it == jn.end() ? nullptr : &it->second);
          done.push_back(v);
        }
      } else if (!last_region) {
        put_link(u, m.genes[gi + 1].node_base, nullptr);
        done.push_back(m.genes[gi + 1].node_base);
      }
      auto jt = jn_by_src.find(u);
      if (jt != jn_by_src.end())
        for (size_t i : jt->second) {
          const uint64_t v = jn_order[i].second;
          if (std::find(done.begin(), done.end(), v) != done.end()) continue;
          put_link(u, v, &jn[jn_order[i]]);
        }
    }
  }
}

void emit_gene_gtf(const Model& m, uint32_t gi, std::string& out) {
  const Gene& g = m.genes[gi];
  if (g.exons.empty()) return;
  std::vector<uint64_t> pos_a(g.slots.size()), pos_b(g.slots.size());
  uint64_t acc = g.bp_base + 1;
  for (size_t i = 0; i < g.slots.size(); ++i) {
    pos_a[i] = acc;
    acc += node_len(g, uint32_t(i));
    pos_b[i] = acc - 1;
  }
  std::string chrom = "chrS";
  put_u(chrom, g.chrom + 1);
  auto row = [&](const char* feat, uint64_t a, uint64_t b) {
    out += chrom;
    out += "\tsynth\t";
    out += feat;
    out += '\t';
    put_u(out, a);
    out += '\t';
    put_u(out, b);
    out += "\t.\t";
    out += g.strand;
    out += "\t.\tgene_id \"G";
    put_u(out, gi + 1);
    out += "\";";
  };
  row("gene", pos_a[g.exons.front().first], pos_b[g.exons.back().second]);
  out += '\n';
  for (uint32_t ti = 0; ti < g.txs.size(); ++ti) {
    const auto& t = g.txs[ti];
    row("mRNA", pos_a[t.front().first], pos_b[t.back().second]);
    out += " transcript_id \"";
    tx_name(out, gi, ti);
    out += "\";\n";
    for (const Range& e : t) {
      row("exon", pos_a[e.first], pos_b[e.second]);
      out += " transcript_id \"";
      tx_name(out, gi, ti);
      out += "\";\n";
    }
  }
}

void cs_for_span(Rng& r, const SynthParams& p, uint32_t span, std::string& cs) {
  cs.clear();
  const double x = r.prob();
  static const char B[4] = {'a', 'c', 'g', 't'};
  if (x >= p.p_sub + p.p_indel || span < 12) { cs += ':'; put_u(cs, span); return; }
  if (x < p.p_sub) {
    int nsub = r.prob() < 0.7 ? 1 : 2;
    uint32_t c1 = uint32_t(r.range(1, span - 2)), c2 = c1;
    if (nsub == 2) { while (c2 == c1) c2 = uint32_t(r.range(1, span - 2)); if (c2 < c1) std::swap(c1, c2); }
    uint32_t prev = 0;
    for (int i = 0; i < nsub; ++i) {
      uint32_t c = i == 0 ? c1 : c2;
      if (c > prev) { cs += ':'; put_u(cs, c - prev); }
      char a = B[r.next() & 3], b = a;
      while (b == a) b = B[r.next() & 3];
      cs += '*'; cs += a; cs += b;
      prev = c + 1;
    }
    if (span > prev) { cs += ':'; put_u(cs, span - prev); }
    return;
  }
  uint32_t c = uint32_t(r.range(3, span - 8));
  uint32_t n = uint32_t(r.range(1, 4));
  std::string ins;
  for (uint32_t i = 0; i < n; ++i) ins += B[r.next() & 3];
  cs += ':'; put_u(cs, c);
  if (r.prob() < 0.5) { cs += '+'; cs += ins; cs += ':'; put_u(cs, span - c); }
  else { cs += '-'; cs += ins; cs += ':'; put_u(cs, span - c - n); }
}

void emit_gaf_line(const Model& m, const std::vector<double>& cdf, const std::vector<uint32_t>& perm, uint64_t li,
                   std::string& out, std::vector<uint64_t>& ids, std::vector<uint32_t>& lens, std::string& cs) {
  const SynthParams& p = m.p;
  Rng r(p.seed * 3000017ull + li * 0x9E3779B1ull + 29);
  auto name = [&] { out += "read"; put_u(out, li / 2); out += '/'; put_u(out, li % 2 + 1); };
  if (r.prob() < p.p_unmapped) {
    name();
    out += '\t'; put_u(out, p.read_len);
    out += "\t0\t0\t+\t*\t0\t0\t0\t0\t0\t";
    out += (r.next() & 1) ? "60" : "0";
    out += '\n';
    return;
  }
  // deletion skew (p_del_hot): decided from a stream of its own so that p_del_hot = 0 leaves every other draw as it was
  bool hot = false, cool_del = false;
  uint32_t hot_id = 0;
  if (p.p_del_hot > 0) {
    Rng h(p.seed * 7919ull + li * 0x9E3779B97F4A7C15ull + 101);
    hot = h.prob() < p.p_indel * 0.5 * p.p_del_hot;
    cool_del = h.prob() < p.p_del_hot;        // a random-position deletion of this read becomes an insertion
    hot_id = uint32_t(h.next() % std::max<uint32_t>(1, p.hot_sites));
  }
  const double x = r.prob() * cdf.back();
  uint32_t gi = perm[size_t(std::lower_bound(cdf.begin(), cdf.end(), x) - cdf.begin())];
  if (hot) gi = perm[hot_id % p.n_genes];          // perm[0], perm[1], ... = the most expressed genes
  const Gene& g = m.genes[gi];
  std::vector<Range> ex = g.txs[hot ? 0 : size_t(r.next() % g.txs.size())];
  const int hap = hot ? 0 : int(r.next() & 1);
  const double kx = hot ? 1.0 : r.prob();
  int kind = 0;   // 0 plain, 1 skip, 2 cassette, 3 intron, 4 splice-site shift
  double acc = 0;
  const double pk[4] = {p.p_novel_skip, p.p_novel_ce, p.p_intron, p.p_novel_ss};
  for (int k = 0; k < 4; ++k) { acc += pk[k]; if (kx < acc) { kind = k + 1; break; } }
  size_t jpos = 0;   // index in the path of the first node after the hop
  auto slots_before = [&](const std::vector<Range>& e, size_t upto) {
    size_t n = 0;
    for (size_t i = 0; i < upto; ++i) n += e[i].second - e[i].first + 1;
    return n;
  };
  if (kind == 1 && ex.size() >= 3) {
    size_t k = size_t(r.range(1, int64_t(ex.size()) - 2));
    ex.erase(ex.begin() + k);
    jpos = slots_before(ex, k);
  } else if (kind == 2 && ex.size() >= 2) {
    size_t k = size_t(r.range(0, int64_t(ex.size()) - 2));
    int64_t ia = int64_t(ex[k].second) + 1, ib = int64_t(ex[k + 1].first) - 1;
    if (ib - ia >= 2) {
      uint32_t a = uint32_t(r.range(ia + 1, ib - 1));
      uint32_t b = uint32_t(std::min<int64_t>(ib - 1, a + r.range(0, 1)));
      ex.insert(ex.begin() + k + 1, Range{a, b});
      jpos = slots_before(ex, k + 1);
    } else kind = 0;
  } else if (kind == 3 && ex.size() >= 2) {
    size_t k = size_t(r.range(0, int64_t(ex.size()) - 2));
    jpos = slots_before(ex, k + 1);
    ex[k].second = ex[k + 1].second;
    ex.erase(ex.begin() + k + 1);
  } else if (kind == 4 && ex.size() >= 2) {
    size_t k = size_t(r.range(0, int64_t(ex.size()) - 2));
    static const int D[6] = {-3, -2, -1, 1, 2, 3};
    int d = D[r.next() % 6];
    if (r.next() & 1) {
      int64_t nb = int64_t(ex[k].second) + d;
      if (int64_t(ex[k].first) <= nb && nb < int64_t(ex[k + 1].first) - 1) ex[k].second = uint32_t(nb);
    } else {
      int64_t na = int64_t(ex[k + 1].first) + d;
      if (int64_t(ex[k].second) + 1 < na && na <= int64_t(ex[k + 1].second)) ex[k + 1].first = uint32_t(na);
    }
    jpos = slots_before(ex, k + 1);
  } else kind = 0;
  tx_path(g, ex, hap, ids, nullptr, &lens);
  uint64_t total = 0;
  for (uint32_t l : lens) total += l;
  const uint32_t span = uint32_t(std::min<uint64_t>(p.read_len, total));
  uint64_t off;
  uint32_t hot_c = 0, hot_n = 0;     // the hot deletion starts hot_c path bases into the window and is hot_n long
  if (hot && span >= 24) {
    hot_n = 1 + (hot_id * 2654435761u >> 30);                                       // 1 .. 4, fixed per site
    // the site = the first base of a path node (fixed per site): the '-' is then the first op the reads lay on that node,
    // which is what makes a deletion an IL/OL key at all (:282, :322) -- every covering read bumps the same key
    uint64_t site = 3;
    {
      const size_t nn = lens.size();
      const size_t j = nn > 2 ? 1 + size_t((uint64_t(hot_id) * 0x9E3779B1ull + gi) % (nn - 1)) : 0;
      uint64_t pre = 0;
      for (size_t i = 0; i < j; ++i) pre += lens[i];
      site = std::min<uint64_t>(std::max<uint64_t>(pre, 3), total > 12 + hot_n ? total - 9 - hot_n : 3);
    }
    // windows that hold the site with the margins cs_for_span keeps (3 before, 8 behind)
    const int64_t lo = std::max<int64_t>(0, int64_t(site) - (int64_t(span) - 8 - hot_n));
    const int64_t hi = std::min<int64_t>(int64_t(total - span), int64_t(site) - 3);
    if (lo <= hi) { off = uint64_t(r.range(lo, hi)); hot_c = uint32_t(site - off); }
    else { hot = false; off = uint64_t(r.range(0, int64_t(total - span))); }
  } else if (hot) { hot = false; off = uint64_t(r.range(0, int64_t(total - span))); }
  else if (kind == 0) off = uint64_t(r.range(0, int64_t(total - span)));
  else {
    uint64_t before = 0;
    for (size_t i = 0; i < jpos && i < lens.size(); ++i) before += lens[i];
    int64_t lo = std::max<int64_t>(0, int64_t(before) - span + 5);
    int64_t hi = std::min<int64_t>(int64_t(total - span), std::max<int64_t>(0, int64_t(before) - 5));
    off = uint64_t(std::min<int64_t>(int64_t(total - span), r.range(std::min(lo, hi), std::max(lo, hi))));
  }
  size_t i0 = 0;
  uint64_t a0 = 0;
  while (a0 + lens[i0] <= off) { a0 += lens[i0]; ++i0; }
  uint64_t start = off - a0;
  size_t i1 = i0;
  uint64_t a1 = a0;
  while (a1 + lens[i1] < off + span) { a1 += lens[i1]; ++i1; }
  uint64_t plen = 0;
  for (size_t i = i0; i <= i1; ++i) plen += lens[i];
  const bool rev = r.next() & 1;
  const int mapq = (r.prob() < p.p_lowmapq && !hot) ? int(r.range(0, 19)) : 60;
  cs_for_span(r, p, span, cs);
  if (p.p_two_indels > 0 && !hot && span >= 40) {
    // (a stream of its own: p_two_indels = 0 leaves every other draw as it was)
    Rng t(p.seed * 104729ull + li * 0x9E3779B97F4A7C15ull + 211);
    if (t.prob() < p.p_two_indels) {
      static const char B[4] = {'a', 'c', 'g', 't'};
      const uint32_t c1 = uint32_t(t.range(3, int64_t(span) / 2 - 8)), n1 = uint32_t(t.range(1, 4));
      const uint32_t c2 = uint32_t(t.range(3, int64_t(span) - c1 - 12)), n2 = uint32_t(t.range(1, 4));
      const bool ins_first = t.next() & 1;
      cs.clear();
      cs += ':'; put_u(cs, c1);
      cs += ins_first ? '+' : '-';
      for (uint32_t i = 0; i < n1; ++i) cs += B[t.next() & 3];
      cs += ':'; put_u(cs, c2);
      cs += ins_first ? '-' : '+';
      for (uint32_t i = 0; i < n2; ++i) cs += B[t.next() & 3];
      // what the two '-' pieces take off the span; an insertion adds nothing to it (its letters are consumed like any op:
      // alignments_augmentation_from_gaf.py:209-255, and the ops may outlast the path)
      const uint32_t del = ins_first ? n2 : n1;
      cs += ':'; put_u(cs, span - c1 - c2 - del);
    }
  }
  if (hot) {
    // the deletion in READ order: a reverse read runs down the path, so the site is met from the other end
    const uint32_t c = rev ? span - hot_c - hot_n : hot_c;
    cs.clear();
    cs += ':'; put_u(cs, c);
    cs += '-';
    for (uint32_t i = 0; i < hot_n; ++i) cs += 'a';
    cs += ':'; put_u(cs, span - c - hot_n);
  } else if (cool_del) {
    const size_t d = cs.find('-');
    if (d != std::string::npos) {       // ":c-xyz:rest" -> ":c+xyz:(rest + n)": the ops must still cover the window
      size_t e = d + 1;
      while (e < cs.size() && cs[e] >= 'a' && cs[e] <= 'z') ++e;
      const uint64_t n = e - d - 1;
      const uint64_t rest = strtoull(cs.c_str() + e + 1, nullptr, 10);
      std::string t = cs.substr(0, d) + "+" + cs.substr(d + 1, n) + ":";
      put_u(t, rest + n);
      cs = t;
    }
  }
  if (!hot && r.prob() < p.p_clip && span > 20) {
    uint32_t n = uint32_t(r.range(1, 6));
    std::string ins;
    static const char B[4] = {'a', 'c', 'g', 't'};
    for (uint32_t i = 0; i < n; ++i) ins += B[r.next() & 3];
    cs.clear();
    if (r.next() & 1) { cs += '+'; cs += ins; cs += ':'; put_u(cs, span); }
    else { cs += ':'; put_u(cs, span); cs += '+'; cs += ins; }
  }
  static const char* DV[4] = {"0", "0.01", "0.03", "0.005"};
  const char* dv = r.prob() < p.p_highdv ? "0.2" : DV[r.next() & 3];
  if (rev) start = plen - (start + span);
  name();
  out += '\t'; put_u(out, p.read_len);
  out += "\t0\t"; put_u(out, p.read_len);
  out += "\t+\t";
  if (rev) for (size_t i = i1 + 1; i-- > i0;) { out += '<'; put_u(out, ids[i]); }
  else for (size_t i = i0; i <= i1; ++i) { out += '>'; put_u(out, ids[i]); }
  out += '\t'; put_u(out, plen);
  out += '\t'; put_u(out, start);
  out += '\t'; put_u(out, start + span);
  out += '\t'; put_u(out, span);
  out += '\t'; put_u(out, span);
  out += '\t'; put_u(out, uint64_t(mapq));
  out += "\tAS:i:"; put_u(out, span);
  out += "\tcs:Z:"; out += cs;
  out += "\tdv:f:"; out += dv;
  out += '\n';
}

}  // namespace

void synth_generate(const SynthParams& p, SynthOutput& out, unsigned threads) {
  if (threads == 0) threads = hw_threads();
  if (p.n_genes == 0) throw Error("ArgumentError", "n_genes must be positive");
  Model m;
  m.p = p;
  const uint32_t R = p.n_genes + 1;   // + trailing intergenic stretch
  m.genes.resize(R);
  const unsigned tg = R < 64 ? 1 : threads;
  parallel_chunks(R, tg, [&](unsigned, size_t b, size_t e) { for (size_t i = b; i < e; ++i) build_gene(p, uint32_t(i), m.genes[i]); });
  uint64_t nb = p.base_id, bp = 0;
  const uint32_t per_chrom = (p.n_genes + p.n_chrom - 1) / std::max<uint32_t>(1, p.n_chrom);
  for (uint32_t i = 0; i < R; ++i) {
    Gene& g = m.genes[i];
    g.node_base = nb;
    g.bp_base = bp;
    g.chrom = std::min<uint32_t>(i / std::max<uint32_t>(1, per_chrom), std::max<uint32_t>(1, p.n_chrom) - 1);
    nb += g.n_nodes;
    bp += g.bp_len;
    out.transcripts += g.txs.size();
  }
  out.nodes = nb - p.base_id;
  {
    // fixed-width ids keep string order == numeric order
    uint64_t a = p.base_id, b = nb - 1;
    int wa = 0, wb = 0;
    while (a) { ++wa; a /= 10; }
    while (b) { ++wb; b /= 10; }
    if (wa != wb) throw Error("ArgumentError", "base_id too small: node ids would change width");
  }
  // ---- GFA ----------------------------------------------------------------------------------
  if (!p.gaf_only) {
  std::vector<std::string> s_parts(R), l_parts(R), gtf_parts(R);
  std::vector<uint64_t> nl(R, 0), nj(R, 0);
  parallel_chunks(R, tg, [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      emit_gene_gfa(m, uint32_t(i), s_parts[i], l_parts[i], nl[i], nj[i]);
      emit_gene_gtf(m, uint32_t(i), gtf_parts[i]);
    }
  });
  // gather: header, every region's S lines, every region's L lines, P lines -- copied in parallel
  std::string paths;
  if (p.with_path) {
    uint32_t i = 0;
    while (i < R) {
      const uint32_t c = m.genes[i].chrom;
      paths += "P\tchrS";
      put_u(paths, c + 1);
      paths += '\t';
      bool first = true;
      for (; i < R && m.genes[i].chrom == c; ++i) {
        const Gene& g = m.genes[i];
        for (uint32_t s = 0; s < g.slots.size(); ++s) {
          if (!first) paths += ',';
          first = false;
          put_u(paths, node_id(g, s, 0));
          paths += '+';
        }
      }
      paths += "\t*\n";
    }
  }
  const std::string header = "H\tVN:Z:1.1\n";
  std::vector<size_t> s_off(R + 1), l_off(R + 1);
  size_t total = header.size();
  for (uint32_t i = 0; i < R; ++i) { s_off[i] = total; total += s_parts[i].size(); }
  for (uint32_t i = 0; i < R; ++i) { l_off[i] = total; total += l_parts[i].size(); out.links += nl[i]; out.junction_links += nj[i]; }
  const size_t p_off = total;
  total += paths.size();
  out.gfa.p.reset(new char[total + 1]);
  out.gfa.n = total;
  char* dst = out.gfa.p.get();
  memcpy(dst, header.data(), header.size());
  parallel_chunks(R, tg, [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      memcpy(dst + s_off[i], s_parts[i].data(), s_parts[i].size());
      memcpy(dst + l_off[i], l_parts[i].data(), l_parts[i].size());
      std::string().swap(s_parts[i]);
      std::string().swap(l_parts[i]);
    }
  });
  memcpy(dst + p_off, paths.data(), paths.size());
  out.gtf = "#!synthetic annotation\n";
  for (auto& s : gtf_parts) out.gtf += s;
  }
  // ---- GAF ----------------------------------------------------------------------------------
  std::vector<uint32_t> perm(p.n_genes);
  for (uint32_t i = 0; i < p.n_genes; ++i) perm[i] = i;
  {
    Rng r(p.seed * 77 + 3);
    for (uint32_t i = p.n_genes; i > 1; --i) std::swap(perm[i - 1], perm[size_t(r.next() % i)]);
  }
  std::vector<double> cdf(p.n_genes);
  double acc = 0;
  for (uint32_t i = 0; i < p.n_genes; ++i) { acc += 1.0 / std::pow(double(i + 1), p.zipf_a); cdf[i] = acc; }
  const unsigned tl = p.n_lines < 4096 ? 1 : threads;
  std::vector<std::string> parts(tl);
  parallel_chunks(size_t(p.n_lines), tl, [&](unsigned c, size_t b, size_t e) {
    std::string& o = parts[c];
    o.reserve((e - b) * 230);
    std::vector<uint64_t> ids;
    std::vector<uint32_t> lens;
    std::string cs;
    for (size_t li = b; li < e; ++li) emit_gaf_line(m, cdf, perm, p.line_offset + li, o, ids, lens, cs);
  });
  std::vector<size_t> g_off(parts.size() + 1, 0);
  for (size_t i = 0; i < parts.size(); ++i) g_off[i + 1] = g_off[i] + parts[i].size();
  out.gaf.p.reset(new char[g_off.back() + 1]);
  out.gaf.n = g_off.back();
  parallel_chunks(parts.size(), unsigned(parts.size()), [&](unsigned, size_t b, size_t e) {
    for (size_t i = b; i < e; ++i) {
      memcpy(out.gaf.p.get() + g_off[i], parts[i].data(), parts[i].size());
      std::string().swap(parts[i]);
    }
  });
  out.gaf_lines = p.n_lines;
}

}  // namespace pantas
