#include "call_host.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cerrno>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace pantas {

const char* const CALL_CSV_HEADER =
    "event_type,annotated/novel,chrom,gene,strand,transcripts1,transcripts2,transcripts3,"
    "nodes1,coverage1,nodes2,coverage2,nodes3,coverage3";

const char* call_error_kind(uint32_t code) {
  switch (code) {
    case CE_KEY_RC: case CE_KEY_TX: case CE_KEY_NC: return "KeyError";
    case CE_ASSERT_EXONS: case CE_ASSERT_JN: case CE_ASSERT_PATH: return "AssertionError";
    case CE_VALUE_INT: case CE_VALUE_MAX: return "ValueError";
    default: return "Unsupported";
  }
}
const char* call_error_text(uint32_t code) {
  switch (code) {
    case CE_KEY_RC: return "'RC' (a link the caller reads has no RC tag: GFA not augmented?)";
    case CE_KEY_TX: return "a transcript named by an EX/JN tag is not in the GTF";
    case CE_KEY_NC: return "'NC' (a segment on a reported path has no NC tag)";
    case CE_ASSERT_EXONS: return "a junction end carries no exon of the junction's gene";
    case CE_ASSERT_JN: return "two junctions compared for an alternative splice site share a JN tag";
    case CE_ASSERT_PATH: return "Error while reconstrucing IR path";
    case CE_VALUE_INT: return "invalid literal for int(): segment name of a junction end";
    case CE_VALUE_MAX: return "max() arg is an empty sequence (novel IR walk ran into a dead end)";
    case CE_HANG: return "intron walk does not terminate (the reference would loop forever)";
    case CE_FRONTIER: return "intronic search frontier beyond the engine's overflow arena (a node with tens of thousands of links near a novel junction)";
    default: return "unknown";
  }
}

namespace {

// sizes below which a step stays on one thread; PANTAS_TEST_SMALL_PARALLEL=1 (tests) makes every step take its parallel path
// on the small inputs the test-suite has
inline size_t par_min(size_t n) { static const bool small = getenv("PANTAS_TEST_SMALL_PARALLEL") != nullptr; return small ? 16 : n; }

// (trivial on purpose: arrays of 2.5e8 of them are sized without a constructor pass and every element is set by the tag scan)
struct SegAttr { uint64_t ex_off; uint32_t ex_len; bool has_ex; };

[[noreturn]] void bad_tag(uint64_t off, Span f) {
  throw Error("ValueError", "GFA byte " + std::to_string(off) + ": tag '" + std::string(f.b, f.size()) +
                                "' does not split into 3 fields on ':'");
}

// name:type:value with exactly two ':' (scripts/call.py:28)
inline bool split_tag(Span f, Span& name, Span& value) {
  const char* c1 = static_cast<const char*>(memchr(f.b, ':', f.size()));
  if (!c1) return false;
  const char* c2 = static_cast<const char*>(memchr(c1 + 1, ':', size_t(f.e - c1 - 1)));
  if (!c2) return false;
  if (memchr(c2 + 1, ':', size_t(f.e - c2 - 1))) return false;
  name = Span{f.b, c1};
  value = Span{c2 + 1, f.e};
  return true;
}

inline bool name_is(Span n, const char* s) { return n.size() == 2 && n.b[0] == s[0] && n.b[1] == s[1]; }

int32_t tag_int(Span v, uint64_t off) {
  int64_t x;
  if (!parse_py_int(v, x)) throw Error("ValueError", "GFA byte " + std::to_string(off) + ": invalid literal for int()");
  if (x < 0 || x > INT32_MAX) throw Error("Unsupported", "GFA byte " + std::to_string(off) + ": count outside [0, 2^31)");
  return int32_t(x);
}

// IL/OL values go through build_attrs' position clustering (scripts/call.py:33-66) although nothing reads the result: what is
// left of it is where it raises.  Entries are int lists split on '.' (extra fields are carried along and only matter for the
// list equality of :49); one entry skips the clustering and meets `max(_v, key=x[1])` (:66, IndexError without a second int);
// two or more are summed in one group (positions within d = 3 of each other, :43-45) or in two around the first minimum / first
// maximum (:47-61) -- `collapse_linkcounts` (:15-21) reads x[1] of every member (IndexError) and divides by the sum of the
// counts (ZeroDivisionError), the low group in front of the high one.
void check_hist_value(Span v, uint64_t off) {
  struct Ent { uint32_t first, n; };            // pieces [first, first + n) of `ints`
  std::vector<int64_t> ints;
  std::vector<Ent> ent;
  const char* p = v.b;
  while (p <= v.e) {
    const char* q = static_cast<const char*>(memchr(p, ',', size_t(v.e - p)));
    if (!q) q = v.e;
    Ent en{uint32_t(ints.size()), 0};
    const char* a = p;
    while (a <= q) {
      const char* d = static_cast<const char*>(memchr(a, '.', size_t(q - a)));
      if (!d) d = q;
      int64_t x;
      if (!parse_py_int(Span{a, d}, x)) {
        int64_t big;
        if (parse_py_int(Span{a, d}, big, true)) throw Error("Unsupported", "GFA byte " + std::to_string(off) + ": IL/OL integer beyond 2^60");
        throw Error("ValueError", "GFA byte " + std::to_string(off) + ": IL/OL value is not 'int.int'");
      }
      ints.push_back(x);
      ++en.n;
      a = d + 1;
    }
    ent.push_back(en);
    p = q + 1;
  }
  auto no_count = [&]() { throw Error("IndexError", "GFA byte " + std::to_string(off) + ": list index out of range (IL/OL entry without a count)"); };
  if (ent.size() == 1) {
    if (ent[0].n < 2) no_count();
    return;
  }
  size_t lo = 0, hi = 0;                          // min() / max() keep the first of equals
  for (size_t i = 1; i < ent.size(); ++i) {
    if (ints[ent[i].first] < ints[ent[lo].first]) lo = i;
    if (ints[ent[i].first] > ints[ent[hi].first]) hi = i;
  }
  auto same_list = [&](const Ent& a, const Ent& b) {
    return a.n == b.n && std::equal(ints.begin() + a.first, ints.begin() + a.first + a.n, ints.begin() + b.first);
  };
  // group of an entry: 0 = low, 1 = high, 2 = in neither (a list equal to the minimum's or the maximum's, :49-50)
  const bool one_group = ints[ent[hi].first] - ints[ent[lo].first] < 3;
  auto group_of = [&](size_t i) -> int {
    if (one_group) return 0;
    if (same_list(ent[i], ent[lo]) || same_list(ent[i], ent[hi])) return 2;
    const int64_t x = ints[ent[i].first];
    return x - ints[ent[lo].first] < ints[ent[hi].first] - x ? 0 : 1;
  };
  for (int grp = 0; grp < (one_group ? 1 : 2); ++grp) {
    __int128 count = 0;
    auto take = [&](const Ent& e) { if (e.n < 2) no_count(); count += ints[e.first + 1]; };
    if (!one_group) take(ent[grp == 0 ? lo : hi]);   // k1 = [minv], k2 = [maxv] (:40-42)
    for (size_t i = 0; i < ent.size(); ++i)
      if (group_of(i) == grp) take(ent[i]);
    if (count == 0) throw Error("ZeroDivisionError", "GFA byte " + std::to_string(off) + ": division by zero (IL/OL counts of a cluster sum to 0)");
  }
}

// 'T_H.k' -> 'T' (:107-121): drop the last '.'-field, then the last '_'-field
inline Span tx_of_exon(Span tag) {
  const char* d = tag.e;
  while (d > tag.b && d[-1] != '.') --d;
  if (d == tag.b) return Span{tag.b, tag.b};
  Span ht{tag.b, d - 1};
  const char* u = ht.e;
  while (u > ht.b && u[-1] != '_') --u;
  if (u == ht.b) return Span{tag.b, tag.b};
  return Span{tag.b, u - 1};
}
// 'T_H.k.k2' -> 'T' (:98-104): drop the last two '.'-fields, then the last '_'-field
inline Span tx_of_junction(Span tag) {
  const char* d = tag.e;
  int dots = 0;
  while (d > tag.b) {
    if (d[-1] == '.') { if (++dots == 2) break; }
    --d;
  }
  if (dots < 2) return Span{tag.b, tag.b};
  Span ht{tag.b, d - 1};
  const char* u = ht.e;
  while (u > ht.b && u[-1] != '_') --u;
  if (u == ht.b) return Span{tag.b, tag.b};
  return Span{tag.b, u - 1};
}
// "_".join(x.split("_")[:-1]) (:356, :364, :595, :603)
inline Span tx_for_filter(Span tag) {
  const char* u = tag.e;
  while (u > tag.b && u[-1] != '_') --u;
  if (u == tag.b) return Span{tag.b, tag.b};
  return Span{tag.b, u - 1};
}

struct Gtf {
  std::unordered_map<std::string, uint32_t> gene_id;    // gene name -> id
  std::unordered_map<std::string, uint32_t> tx_gene;    // transcript name -> gene id
};

bool is_id_char(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_'; }

// re.search('<key> "[A-Za-z0-9_]+";', s): first match
bool find_attr(Span s, const char* key, Span& out) {
  const size_t kl = strlen(key);
  for (const char* p = s.b; p + kl + 4 <= s.e; ++p) {
    if (memcmp(p, key, kl) != 0 || p[kl] != ' ' || p[kl + 1] != '"') continue;
    const char* q = p + kl + 2;
    const char* r = q;
    while (r < s.e && is_id_char(*r)) ++r;
    if (r > q && r + 1 < s.e && r[0] == '"' && r[1] == ';') { out = Span{q, r}; return true; }
  }
  return false;
}

// :188-216.  The lines are split and matched by all threads (a whole-genome annotation is 180 MB: 1 s on one); the rows that
// matter -- one per transcript -- are applied in file order by one, and the first line the reference would die on (in FILE
// order, whatever chunk found it) is the error.
void parse_gtf(const MappedFile& f, Gtf& gtf, CallTables& t, unsigned threads = 0) {
  struct Row { Span gid, tid, strand, chrom; };
  struct Chunk { std::vector<Row> rows; uint64_t lines = 0; std::unique_ptr<Error> err; uint64_t err_line = 0; size_t rows_before_err = 0; };
  if (threads == 0) threads = hw_threads();
  const unsigned C = f.size < par_min(size_t(4) << 20) ? 1u : threads;
  const auto ranges = split_at_newlines(f.data, f.size, C);
  std::vector<Chunk> ck(ranges.size());
  static const char* feats[] = {"mRNA", "transcript", "miRNA", "ncRNA", "pre_miRNA", "snoRNA", "pseudogene"};
  parallel_chunks(ranges.size(), unsigned(ranges.size()), [&](unsigned, size_t cb, size_t ce) {
    for (size_t c = cb; c < ce; ++c) {
      Chunk& k = ck[c];
      const char* p = f.data + ranges[c].first;
      const char* end = f.data + ranges[c].second;
      auto fail = [&](const char* kind, const std::string& what) {
        k.err.reset(new Error(kind, what));
        k.err_line = k.lines;
        k.rows_before_err = k.rows.size();
      };
      std::vector<Span> col;
      while (p < end && !k.err) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', size_t(end - p)));
        const char* le = nl ? nl : end;
        ++k.lines;
        Span line{p, le};
        p = nl ? nl + 1 : end;
        if (lone_cr(line.b, line.e)) { fail("Unsupported", ": lone carriage return (Python's universal newlines end a line there; this engine splits at \\n only)"); break; }
        if (line.size() && *line.b == '#') continue;
        // split on '\t' only (:194)
        col.clear();
        const char* a = line.b;
        for (;;) {
          const char* tb = static_cast<const char*>(memchr(a, '\t', size_t(line.e - a)));
          if (!tb) { col.push_back(Span{a, line.e}); break; }
          col.push_back(Span{a, tb});
          a = tb + 1;
        }
        if (col.size() < 3) { fail("IndexError", ": fewer than 3 columns"); break; }
        bool want = false;
        for (const char* ft : feats) want |= col[2].sv() == ft;
        if (!want) continue;
        Span gid, tid;
        if (!find_attr(col.back(), "gene_id", gid) || !find_attr(col.back(), "transcript_id", tid)) {
          fail("AttributeError", ": 'NoneType' object has no attribute 'group' (gene_id/transcript_id \"[A-Za-z0-9_]+\";)");
          break;
        }
        if (col.size() < 7) { fail("IndexError", ": fewer than 7 columns"); break; }
        k.rows.push_back(Row{gid, tid, col[6], col[0]});
      }
    }
  });
  uint64_t lines_before = 0;
  for (Chunk& k : ck) {
    const size_t n = k.err ? k.rows_before_err : k.rows.size();
    for (size_t i = 0; i < n; ++i) {
      const Row& r = k.rows[i];
      std::string g(r.gid.b, r.gid.size());
      auto it = gtf.gene_id.find(g);
      uint32_t id;
      if (it == gtf.gene_id.end()) {
        id = uint32_t(t.gene_name.size());
        gtf.gene_id.emplace(g, id);
        t.gene_name.push_back(g);
        t.gene_strand.emplace_back();
        t.gene_chrom.emplace_back();
      } else id = it->second;
      t.gene_strand[id].assign(r.strand.b, r.strand.size());
      t.gene_chrom[id].assign(r.chrom.b, r.chrom.size());
      gtf.tx_gene[std::string(r.tid.b, r.tid.size())] = id;
    }
    if (k.err) {
      const std::string msg = k.err->what();      // "Kind: <suffix>"
      const size_t colon = msg.find(": ");
      throw Error(k.err->kind, "GTF line " + std::to_string(lines_before + k.err_line) + (colon == std::string::npos ? msg : msg.substr(colon + 2)));
    }
    lines_before += k.lines;
  }
}

// Strings interned per thread -> ids in order of FIRST APPEARANCE over all of them.  Every chunk lists its distinct strings
// in the order it met them; chunks are in file order.  One pass per distinct string on one thread was 1.3 s for the 4e6 EX
// tags of a whole-genome graph; here the strings are dealt to shards by hash, every shard (one thread each) walks the
// chunks in order and keeps the first place it saw each of its strings, the places are sorted, and the rank is the id.
// 64-bit mix over bytes: "the same GTF" for CallTables::Reusable, the keys of merge_interned
static uint64_t bytes_hash(const char* p, size_t n) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
  }
  for (; i < n; ++i) h = (h ^ uint8_t(p[i])) * 0x100000001B3ull;
  return h;
}

struct InternChunk {
  std::vector<StrRef> span;          // local id -> text (offsets into `base`), in local first-appearance order
  std::vector<uint32_t> to_global;   // filled by merge_interned
};
std::vector<StrRef> merge_interned(const char* base, std::vector<InternChunk>& ck, unsigned threads) {
  // Sort-based (round 3; the first form dealt the strings to per-thread hash maps: 0.9 s for the 2e6 distinct EX tags of a
  // whole-genome graph, most of it cache misses of the maps): every local string as {hash, place}, sorted by hash -- equal
  // hashes by the strings themselves, then by place --, so that the copies of a string stand together, the first of them being its
  // first appearance; the first appearances sorted by place give the ids.
  PhaseTimer mt("merge_interned");
  const size_t C = ck.size();
  std::vector<size_t> cbase(C + 1, 0);
  for (size_t c = 0; c < C; ++c) cbase[c + 1] = cbase[c] + ck[c].span.size();
  const size_t N = cbase[C];
  auto sv_of = [&](const StrRef& r) { return std::string_view(base + r.off, r.len); };
  struct Item { uint64_t hash, place; };      // place = chunk << 32 | local id: the order of first appearance
  RawVector<Item> it(N);
  parallel_chunks(C, unsigned(std::min<size_t>(C, threads ? threads : 1)), [&](unsigned, size_t cb, size_t ce) {
    for (size_t c = cb; c < ce; ++c)
      for (size_t i = 0; i < ck[c].span.size(); ++i) it[cbase[c] + i] = Item{bytes_hash(base + ck[c].span[i].off, ck[c].span[i].len), (uint64_t(c) << 32) | uint64_t(i)};
  });
  mt.lap("hashes");
  auto span_at = [&](uint64_t place) -> const StrRef& { return ck[size_t(place >> 32)].span[size_t(place & 0xFFFFFFFFull)]; };
  par_sort(it.begin(), it.end(), [&](const Item& a, const Item& b) {
    if (a.hash != b.hash) return a.hash < b.hash;
    const std::string_view x = sv_of(span_at(a.place)), y = sv_of(span_at(b.place));
    if (x != y) return x < y;                  // (two strings of one hash: apart)
    return a.place < b.place;
  }, N < par_min(size_t(1) << 16) ? 1u : threads);
  mt.lap("sorted by hash");
  // group heads: the first copy of every distinct string (smallest place of its group)
  RawVector<uint32_t> group(N);              // item -> index of its group among the heads, in sorted order
  std::vector<std::pair<uint64_t, uint32_t>> heads;      // (place of the first appearance, group)
  {
    RawVector<uint8_t> is_head(N);
    parallel_chunks(N, N < par_min(65536) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t k = b; k < e; ++k)
        is_head[k] = k == 0 || it[k].hash != it[k - 1].hash || sv_of(span_at(it[k].place)) != sv_of(span_at(it[k - 1].place));
    });
    size_t d = 0;
    for (size_t k = 0; k < N; ++k) { d += is_head[k]; group[k] = uint32_t(d - 1); }
    heads.resize(d);
    parallel_chunks(N, N < par_min(65536) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t k = b; k < e; ++k) if (is_head[k]) heads[group[k]] = {it[k].place, group[k]};
    });
  }
  const size_t D = heads.size();
  if (D >= size_t(CNONE)) throw Error("Unsupported", "more than 2^32-2 distinct tags");
  par_sort(heads.begin(), heads.end(), [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; },
           D < par_min(size_t(1) << 16) ? 1u : threads);
  mt.lap("order by first place");
  std::vector<StrRef> global(D);
  std::vector<uint32_t> id_of_group(D);
  parallel_chunks(D, D < par_min(65536) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
    for (size_t r = b; r < e; ++r) { id_of_group[heads[r].second] = uint32_t(r); global[r] = span_at(heads[r].first); }
  });
  for (size_t c = 0; c < C; ++c) ck[c].to_global.resize(ck[c].span.size());
  parallel_chunks(N, N < par_min(65536) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
    for (size_t k = b; k < e; ++k) ck[size_t(it[k].place >> 32)].to_global[size_t(it[k].place & 0xFFFFFFFFull)] = id_of_group[group[k]];
  });
  mt.lap("ids back to the chunks");
  return global;
}

template <class T>
void sort_unique(std::vector<T>& v) {
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

}  // namespace

void build_call_tables(const Graph& g, const char* gtf_path, CallTables& t, unsigned threads) {
  MappedFile f;
  f.open(gtf_path);
  build_call_tables_mem(g, f.data, f.size, t, threads);
}

void build_call_tables_mem(const Graph& g, const char* gtf_data, size_t gtf_size, CallTables& t, unsigned threads,
                           const AugmentResult* aug, CallTables* previous) {
  const double t0 = now_s();
  PhaseTimer pt("call_tables");
  release_deferred();      // host-only work from here to the upload: the parse's temporaries are freed beside it
  // a GTF with a '\r' that no '\n' follows: read the way Python does (common.h lone_cr), from a copy
  RawVector<char> gtf_universal;
  if (has_lone_cr(gtf_data, gtf_size)) {
    universal_newlines_copy(gtf_data, gtf_size, gtf_universal);
    gtf_data = gtf_universal.data();
  }
  if (threads == 0) threads = hw_threads();
  const char* base = g.text.data;
  const uint64_t gtf_hash = aug ? bytes_hash(gtf_data, gtf_size) : 0;
  // the graph-and-GTF-only part of an earlier fused build on this graph
  const bool reuse = aug && previous && previous->reuse.valid && previous->reuse.text == static_cast<const void*>(base) &&
                     previous->reuse.V == g.V && previous->reuse.graph_A == g.A && previous->reuse.gtf_size == gtf_size &&
                     previous->reuse.gtf_hash == gtf_hash && !getenv("PANTAS_CALL_NO_REUSE");
  t.reused_previous = reuse;
  if (pt.on && reuse) fprintf(stderr, "[timing] call_tables: graph-and-GTF-only part taken over from the previous build on this handle\n");
  // inputs the reference's GFA loop dies on (call.py:145-181)
  if (g.first_bad_p_off != UINT64_MAX)
    throw Error("ValueError", "GFA byte " + std::to_string(g.first_bad_p_off) + ": P line does not have exactly 4 fields");
  if (g.first_short_l_off != UINT64_MAX)
    throw Error("ValueError", "GFA byte " + std::to_string(g.first_short_l_off) + ": L line with fewer than 6 fields");
  if (g.links_unknown_end || g.link_before_segment)
    throw Error("KeyError", "an L line names a segment with no S line before it");
  if (g.dup_names && g.dup_ws)
    throw Error("Unsupported", "several S lines for one segment name, one of them behind leading whitespace (call)");
  const uint32_t V = g.V;
  if (V >= 0x80000000u) throw Error("Unsupported", "more than 2^31 segments (call: the frontier search keeps a bit beside a node index)");
  // ---- link layout -------------------------------------------------------------------------------
  RawVector<uint32_t> old_slot;     // fused: slot of the graph behind each slot of the layout (NONE32 = added link)
  std::vector<uint32_t> slot_of_added;   // fused: slot of the i-th added link (first-seen order) in the layout
  if (!aug || aug->novel.empty()) {
    t.A = g.A;
    t.row_ptr = g.row_ptr.data(); t.col = g.col.data();
    t.irow_ptr = g.irow_ptr.data(); t.icol = g.icol.data(); t.islot = g.islot.data();
    t.order = g.slot_off.data();
  } else {
    const size_t extra = aug->novel.size();
    const size_t An = g.A + extra;
    if (An >= NONE32) throw Error("Unsupported", "more than 2^32-2 links");
    // The layout = the graph's rows with the links the reads added appended to their rows, in the first-seen order the
    // writer would print them in.  They are few (1.5e6 on 2.6e8 slots): row u moves up by the number of added links with a
    // smaller source, found in a sorted copy of their sources; everything per node / per slot is done by all threads.
    std::vector<uint32_t> by_u(extra), by_v(extra);
    for (size_t i = 0; i < extra; ++i) by_u[i] = by_v[i] = uint32_t(i);
    std::stable_sort(by_u.begin(), by_u.end(), [&](uint32_t a, uint32_t b2) { return aug->novel[a].u < aug->novel[b2].u; });
    std::stable_sort(by_v.begin(), by_v.end(), [&](uint32_t a, uint32_t b2) { return aug->novel[a].v < aug->novel[b2].v; });
    std::vector<uint32_t> src_sorted(extra), dst_sorted(extra);
    for (size_t i = 0; i < extra; ++i) { src_sorted[i] = aug->novel[by_u[i]].u; dst_sorted[i] = aug->novel[by_v[i]].v; }
    t.x_row_ptr.resize(size_t(V) + 1);
    t.x_irow_ptr.resize(size_t(V) + 1);
    t.x_col.resize(An); t.x_order.resize(An); t.x_icol.resize(An); t.x_islot.resize(An);
    old_slot.resize(An);
    RawVector<uint32_t> new_of_old(g.A);
    parallel_chunks(size_t(V) + 1, V < 4096 ? 1 : threads, [&](unsigned, size_t ub, size_t ue) {
      size_t ku = size_t(std::lower_bound(src_sorted.begin(), src_sorted.end(), uint32_t(ub)) - src_sorted.begin());
      size_t kv = size_t(std::lower_bound(dst_sorted.begin(), dst_sorted.end(), uint32_t(ub)) - dst_sorted.begin());
      for (size_t u = ub; u < ue; ++u) {
        while (ku < extra && src_sorted[ku] < u) ++ku;          // added links with a source below u
        while (kv < extra && dst_sorted[kv] < u) ++kv;
        t.x_row_ptr[u] = g.row_ptr[u] + uint32_t(ku);
        t.x_irow_ptr[u] = g.irow_ptr[u] + uint32_t(kv);
      }
    });
    parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t ub, size_t ue) {
      for (size_t u = ub; u < ue; ++u) {
        uint32_t w = t.x_row_ptr[u];
        for (uint32_t s2 = g.row_ptr[u]; s2 < g.row_ptr[u + 1]; ++s2, ++w) {
          t.x_col[w] = g.col[s2];
          t.x_order[w] = g.slot_off[s2];
          old_slot[w] = s2;
          new_of_old[s2] = w;
        }
      }
    });
    // the added links behind the old ones of their row, in first-seen order (by_u is a stable sort of that order)
    for (size_t k = 0; k < extra;) {
      const uint32_t u = src_sorted[k];
      uint32_t w = t.x_row_ptr[u] + (g.row_ptr[u + 1] - g.row_ptr[u]);
      for (; k < extra && src_sorted[k] == u; ++k, ++w) {
        const NovelLink& nl = aug->novel[by_u[k]];
        t.x_col[w] = nl.v;
        t.x_order[w] = uint64_t(g.text.size) + by_u[k];
        old_slot[w] = NONE32;
      }
    }
    parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t vb, size_t ve) {
      for (size_t v = vb; v < ve; ++v) {
        uint32_t w = t.x_irow_ptr[v];
        for (uint32_t q = g.irow_ptr[v]; q < g.irow_ptr[v + 1]; ++q, ++w) {
          t.x_icol[w] = g.icol[q];
          t.x_islot[w] = new_of_old[g.islot[q]];
        }
      }
    });
    {
      // slot of added link i in the out layout: its row's old links, then its rank among the row's added links
      slot_of_added.resize(extra);
      for (size_t k = 0; k < extra;) {
        const uint32_t u = src_sorted[k];
        uint32_t w = t.x_row_ptr[u] + (g.row_ptr[u + 1] - g.row_ptr[u]);
        for (; k < extra && src_sorted[k] == u; ++k, ++w) slot_of_added[by_u[k]] = w;
      }
      for (size_t k = 0; k < extra;) {
        const uint32_t v = dst_sorted[k];
        uint32_t q = t.x_irow_ptr[v] + (g.irow_ptr[v + 1] - g.irow_ptr[v]);
        for (; k < extra && dst_sorted[k] == v; ++k, ++q) {
          t.x_icol[q] = aug->novel[by_v[k]].u;
          t.x_islot[q] = slot_of_added[by_v[k]];
        }
      }
    }
    t.A = An;
    t.row_ptr = t.x_row_ptr.data(); t.col = t.x_col.data();
    t.irow_ptr = t.x_irow_ptr.data(); t.icol = t.x_icol.data(); t.islot = t.x_islot.data();
    t.order = t.x_order.data();
  }
  const size_t A = t.A;
  auto graph_slot = [&](size_t s) -> uint32_t { return old_slot.empty() ? uint32_t(s) : old_slot[s]; };
  t.hidden.clear();
  if (g.dup_names) {
    // A later S line of a name replaces gfaS[name] and with it the segment's "O" / "I" lists (call.py:149-154): links read
    // before it stay in gfaL (and in `junctions`) but are no neighbours any more.  Links the reads added stand behind the text.
    t.hidden.resize(A);
    for (size_t u = 0; u < V; ++u)
      for (uint32_t s = t.row_ptr[u]; s < t.row_ptr[u + 1]; ++s)
        t.hidden[s] = uint8_t((t.order[s] < g.seg_line_off[u] ? 1 : 0) | (t.order[s] < g.seg_line_off[t.col[s]] ? 2 : 0));
  }
  size_t next_novel = 0;
  (void)next_novel;

  Gtf gtf;
  if (reuse) {
    t.gene_name = std::move(previous->gene_name); t.gene_strand = std::move(previous->gene_strand); t.gene_chrom = std::move(previous->gene_chrom);
  } else {
    MappedFile f;
    f.attach(gtf_data, gtf_size);
    parse_gtf(f, gtf, t);
  }

  pt.lap("gtf + link layout");
  // ---- phase A (parallel): locate the tags of every S line and every link slot -----------------
  // (the GPU loader may have read the tags while the text was in HBM: graph.h CallTagScan)
  std::unique_ptr<CallTagScan> ct;
  if (!reuse && g.call_tags && g.call_tags->nc.size() == V && g.call_tags->la.size() == g.A && !getenv("PANTAS_CALL_TAGS_HOST")) ct = std::move(g.call_tags);
  RawVector<SegAttr> sa((reuse || ct) ? 0 : V);
  if (ct) { t.nc = std::move(ct->nc); t.nflags = std::move(ct->nflags); }
  else { t.nc.resize(V); t.nflags.resize(V); }
  if (ct) {
    if (aug) parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t n = b; n < e; ++n) {
        if (aug->nc[n] > uint32_t(INT32_MAX)) throw Error("Unsupported", "node coverage beyond 2^31");
        t.nc[n] = int32_t(aug->nc[n]);
        t.nflags[n] |= N_HAS_NC;
      }
    });
  } else if (reuse) {
    // the S lines were read (and their tags checked) by the build whose tables are taken over: only the counters are new
    parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t n = b; n < e; ++n) {
        if (aug->nc[n] > uint32_t(INT32_MAX)) throw Error("Unsupported", "node coverage beyond 2^31");
        t.nc[n] = int32_t(aug->nc[n]);
        t.nflags[n] = N_HAS_NC;
      }
    });
  } else
  parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
    for (size_t n = b; n < e; ++n) {
      sa[n] = SegAttr{0, 0, false};
      t.nc[n] = 0;
      t.nflags[n] = 0;
      const uint64_t off = g.seg_line_off[n];
      const char* le = static_cast<const char*>(memchr(base + off, '\n', g.text.size - off));
      Span st = strip(Span{base + off, le ? le : base + g.text.size});
      Tokens tk(st);
      Span f;
      tk.next(f); tk.next(f); tk.next(f);
      while (tk.next(f)) {
        Span name, value;
        if (!split_tag(f, name, value)) bad_tag(off, f);
        if (name_is(name, "NC")) { t.nc[n] = tag_int(value, off); t.nflags[n] |= N_HAS_NC; }
        else if (name_is(name, "LN") || name_is(name, "RC")) (void)tag_int(value, off);
        else if (name_is(name, "IL") || name_is(name, "OL")) check_hist_value(value, off);
        else if (name_is(name, "EX")) { sa[n].ex_off = uint64_t(value.b - base); sa[n].ex_len = uint32_t(value.size()); sa[n].has_ex = true; }
      }
      if (aug) {     // the NC tag augment would append overrides any earlier one (dict semantics, call.py:25-32)
        if (aug->nc[n] > uint32_t(INT32_MAX)) throw Error("Unsupported", "node coverage beyond 2^31");
        t.nc[n] = int32_t(aug->nc[n]);
        t.nflags[n] |= N_HAS_NC;
      }
    }
  });
  // tag records of the GRAPH's L lines, by graph slot: read from the text, or taken over from the previous build
  RawVector<LinkAttr> la_graph;
  if (reuse) la_graph = std::move(previous->reuse.la_graph);
  else if (ct) la_graph = std::move(ct->la);
  else {
    la_graph.resize(g.A);
    parallel_chunks(g.A, g.A < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
      for (size_t gs = b; gs < e; ++gs) {
        LinkAttr& a = la_graph[gs];
        a = LinkAttr{0, 0, false, false, false, 0};
        const uint64_t off = g.slot_off[gs];
        const char* le = static_cast<const char*>(memchr(base + off, '\n', g.text.size - off));
        Span st = strip(Span{base + off, le ? le : base + g.text.size});
        Tokens tk(st);
        Span f;
        for (int i = 0; i < 6; ++i) tk.next(f);
        while (tk.next(f)) {
          Span name, value;
          if (!split_tag(f, name, value)) bad_tag(off, f);
          if (name_is(name, "RC")) { a.rc = tag_int(value, off); a.has_rc = true; }
          else if (name_is(name, "LN") || name_is(name, "NC")) (void)tag_int(value, off);
          else if (name_is(name, "IL") || name_is(name, "OL")) check_hist_value(value, off);
          else if (name_is(name, "JN")) { a.jn_off = uint64_t(value.b - base); a.jn_len = uint32_t(value.size()); a.has_jn = true; }
          else if (name_is(name, "ID")) a.has_id = true;
        }
      }
    });
  }
  RawVector<LinkAttr> la;
  t.src.resize(A);
  const bool la_is_graph = !aug;              // the command line on a GFA text: the layout is the graph, nothing to keep
  if (la_is_graph) la = std::move(la_graph); else la.resize(A);
  parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
    for (size_t u = b; u < e; ++u)
      for (uint32_t s = t.row_ptr[u]; s < t.row_ptr[u + 1]; ++s) {
        t.src[s] = uint32_t(u);
        if (la_is_graph) continue;
        const uint32_t gs = graph_slot(s);
        if (gs == NONE32) { la[s] = LinkAttr{0, 0, false, false, false, 0}; continue; }      // a link added by the reads: filled in below
        LinkAttr& a = la[s];
        a = la_graph[gs];
        if (aug) {
          if (aug->rc[gs] > uint32_t(INT32_MAX)) throw Error("Unsupported", "link coverage beyond 2^31");
          a.rc = int32_t(aug->rc[gs]);
          a.has_rc = true;
        }
      }
  });
  if (aug && !old_slot.empty()) {
    // "L u + v + * RC:i:n ID:Z:N" lines the augment writer appends
    for (size_t i = 0; i < aug->novel.size(); ++i) {
      const NovelLink& nl = aug->novel[i];
      LinkAttr& a = la[slot_of_added[i]];
      if (nl.count > uint32_t(INT32_MAX)) throw Error("Unsupported", "link coverage beyond 2^31");
      a.rc = int32_t(nl.count);
      a.has_rc = true;
      a.has_id = true;
    }
  }

  pt.lap("tag scan (parallel)");
  // ---- canonical slots: first slot of each (u, v); attributes of the LAST duplicate (dict overwrite, :173)
  t.first_slot.resize(A);
  par_assign(t.lflags, A, 0u, threads);
  par_assign(t.rc, A, int32_t(0), threads);
  RawVector<uint32_t> attr_slot(A);
  parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t ub, size_t ue) {
    for (size_t u = ub; u < ue; ++u) {
      const uint32_t b = t.row_ptr[u], e = t.row_ptr[u + 1];
      for (uint32_t s = b; s < e; ++s) {
        uint32_t first = s;
        for (uint32_t q = b; q < s; ++q) if (t.col[q] == t.col[s]) { first = q; break; }
        t.first_slot[s] = first;
        attr_slot[first] = s;   // the last duplicate wins
        if (first != s) {
          // Several L lines for one (from, to): gfaL keeps the attributes of the LAST line (dict overwrite, :173) while
          // `junctions` / `noveljunctions` gain the pair whenever ANY of the lines carries JN / ID (:177-180).  la[first]
          // is not read for attributes once a duplicate exists (they come from the last line): its flags collect "any".
          la[first].has_jn = la[first].has_jn || la[s].has_jn;
          la[first].has_id = la[first].has_id || la[s].has_id;
        }
      }
      // a pair that is a junction through an earlier line only: the reference indexes gfaL[pair]["JN"] of the last line
      // and dies with KeyError wherever it evaluates the junction -- refused here at load
      for (uint32_t s = b; s < e; ++s)
        if (t.first_slot[s] == s && attr_slot[s] != s && la[s].has_jn && !la[attr_slot[s]].has_jn)
          throw Error("Unsupported", "duplicate L lines of one link: an earlier line carries a JN tag, the last one does not");
    }
  });
  parallel_chunks(A, A < 4096 ? 1 : threads, [&](unsigned, size_t sb, size_t se) {
    for (size_t s = sb; s < se; ++s) {
      if (t.first_slot[s] != s) continue;
      const LinkAttr& a = la[attr_slot[s]];
      uint32_t fl = L_FIRST;
      if (a.has_jn) fl |= L_JUNC;                       // (an earlier line's JN without the last one's was refused above)
      if (a.has_id || la[s].has_id) fl |= L_NOVEL;      // any line with ID makes the pair a novel junction
      if (a.has_rc) { fl |= L_HAS_RC; t.rc[s] = a.rc; }
      t.lflags[s] = fl;
    }
  });

  pt.lap("canonical slots (parallel)");
  // ---- phase B: intern transcripts, EX tags, JN tags (first-appearance order) -------------------------
  std::unordered_map<std::string_view, uint32_t> tx_ids;
  auto tx_id = [&](Span name) -> uint32_t {
    auto it = tx_ids.find(name.sv());
    if (it != tx_ids.end()) return it->second;
    uint32_t id = uint32_t(t.tx_gene.size());
    tx_ids.emplace(name.sv(), id);
    auto gi = gtf.tx_gene.find(std::string(name.b, name.size()));
    t.tx_gene.push_back(gi == gtf.tx_gene.end() ? CNONE : gi->second);
    return id;
  };
  // EX tags (:86-121), ids in first-appearance order (rule R2 picks "the tag that appears first").  At whole-genome scale
  // there are ~4e7 tag occurrences on 2.5e8 segments but only a few million distinct tags, local to their genes: every
  // thread interns the occurrences of its node range in a map of its own (no locks), the maps are merged in chunk order
  // -- which IS first-appearance order -- touching each distinct tag once, and the per-node lists are written in parallel.
  std::vector<uint32_t> extag_tx;
  if (reuse) {
    // node-indexed and tag-indexed: nothing in them depends on a run
    t.ex_ptr = std::move(previous->ex_ptr); t.ex_tag = std::move(previous->ex_tag); t.ex_tx = std::move(previous->ex_tx);
    t.ex_gene = std::move(previous->ex_gene); t.extag_span = std::move(previous->extag_span);
    t.tx_gene = std::move(previous->tx_gene);         // complete: the JN tags of the earlier build added theirs
  } else {
  par_assign(t.ex_ptr, size_t(V) + 1, 0u, threads);
  {
    struct ExChunk {
      size_t nb = 0, ne = 0;
      std::vector<uint32_t> occ;                            // per node, its distinct local ids (concatenated)
      uint64_t out_base = 0;
    };
    const unsigned C = V < par_min(4096) ? 1u : threads;
    std::vector<ExChunk> ck(C);
    std::vector<InternChunk> ick(C);
    parallel_chunks(V, C, [&](unsigned c, size_t nb, size_t ne) {
      ExChunk& k = ck[c];
      InternChunk& ik = ick[c];
      k.nb = nb; k.ne = ne;
      std::unordered_map<std::string_view, uint32_t> ids;   // tag -> local id (order of first appearance in the chunk)
      std::vector<uint32_t> tmp;
      const char* prev = nullptr;     // the EX value of the chunk's previous tagged node: the segments of one exon (a bubble cuts it
      uint32_t prev_len = 0;          // into dozens) carry the same value, whose ids are then in tmp already
      auto one = [&](size_t n, uint64_t ex_off, uint32_t ex_len) {
        const char* p = base + ex_off;
        const char* e = p + ex_len;
        if (prev && prev_len == ex_len && memcmp(prev, p, prev_len) == 0) {
          t.ex_ptr[n] = uint32_t(tmp.size());
          k.occ.insert(k.occ.end(), tmp.begin(), tmp.end());
          return;
        }
        prev = p; prev_len = ex_len;
        tmp.clear();
        for (;;) {
          const char* cm = static_cast<const char*>(memchr(p, ',', size_t(e - p)));
          Span tag{p, cm ? cm : e};
          auto ins = ids.emplace(tag.sv(), uint32_t(ik.span.size()));
          if (ins.second) ik.span.push_back(StrRef{uint64_t(tag.b - base), uint32_t(tag.size())});
          tmp.push_back(ins.first->second);
          if (!cm) break;
          p = cm + 1;
        }
        sort_unique(tmp);
        t.ex_ptr[n] = uint32_t(tmp.size());        // count for now, offset after the scan below
        k.occ.insert(k.occ.end(), tmp.begin(), tmp.end());
      };
      if (ct) {      // the tagged nodes as a list (ascending)
        const uint32_t* first = ct->ex_node.data();
        const uint32_t* last = first + ct->ex_node.size();
        for (size_t q = size_t(std::lower_bound(first, last, uint32_t(nb)) - first); q < ct->ex_node.size() && ct->ex_node[q] < ne; ++q)
          one(ct->ex_node[q], ct->ex_off[q], ct->ex_len[q]);
      } else {
        for (size_t n = nb; n < ne; ++n) if (sa[n].has_ex) one(n, sa[n].ex_off, sa[n].ex_len);
      }
    });
    pt.lap("intern EX: per-chunk maps (parallel)");
    t.extag_span = merge_interned(base, ick, threads);
    const size_t D = t.extag_span.size();
    // transcripts of the distinct tags, ids in first-appearance order again: the same merge over chunks of the tag list
    {
      const unsigned TC = D < par_min(65536) ? 1u : threads;
      std::vector<InternChunk> tck(TC);
      std::vector<std::vector<uint32_t>> local_tx(TC);
      std::vector<size_t> tb(TC + 1, 0);
      for (unsigned c = 0; c <= TC; ++c) tb[c] = D * c / TC;
      parallel_chunks(TC, TC, [&](unsigned, size_t cb, size_t ce) {
        for (size_t c = cb; c < ce; ++c) {
          std::unordered_map<std::string_view, uint32_t> ids;
          local_tx[c].resize(tb[c + 1] - tb[c]);
          for (size_t i = tb[c]; i < tb[c + 1]; ++i) {
            const StrRef& r = t.extag_span[i];
            const Span tx = tx_of_exon(Span{base + r.off, base + r.off + r.len});
            auto ins = ids.emplace(tx.sv(), uint32_t(tck[c].span.size()));
            if (ins.second) tck[c].span.push_back(StrRef{uint64_t(tx.b - base), uint32_t(tx.size())});
            local_tx[c][i - tb[c]] = ins.first->second;
          }
        }
      });
      const std::vector<StrRef> txs = merge_interned(base, tck, threads);
      // (tx_ids is empty here: EX tags are interned first; JN tags below add the transcripts only they name)
      tx_ids.reserve(txs.size() * 2);
      t.tx_gene.reserve(txs.size());
      for (const StrRef& r : txs) (void)tx_id(Span{base + r.off, base + r.off + r.len});
      extag_tx.resize(D);
      parallel_chunks(TC, TC, [&](unsigned, size_t cb, size_t ce) {
        for (size_t c = cb; c < ce; ++c)
          for (size_t i = tb[c]; i < tb[c + 1]; ++i) extag_tx[i] = tck[c].to_global[local_tx[c][i - tb[c]]];
      });
    }
    uint64_t total = 0;
    for (unsigned c = 0; c < C; ++c) { ck[c].out_base = total; total += ck[c].occ.size(); }
    pt.lap("intern EX: merge of distinct tags and transcripts (sharded)");
    if (total >= uint64_t(CNONE)) throw Error("Unsupported", "more than 2^32-2 EX tag entries");
    // counts -> offsets (per chunk: its nodes' entries are contiguous behind out_base)
    parallel_chunks(C, C, [&](unsigned, size_t cb, size_t ce) {
      for (size_t c = cb; c < ce; ++c) {
        uint64_t at = ck[c].out_base;
        for (size_t n = ck[c].nb; n < ck[c].ne; ++n) { const uint32_t cnt = t.ex_ptr[n]; t.ex_ptr[n] = uint32_t(at); at += cnt; }
      }
    });
    t.ex_ptr[V] = uint32_t(total);
    t.ex_tag.resize(total); t.ex_tx.resize(total); t.ex_gene.resize(total);
    parallel_chunks(C, C, [&](unsigned, size_t cb, size_t ce) {
      std::vector<uint32_t> tmp;
      for (size_t c = cb; c < ce; ++c) {
        const ExChunk& k = ck[c];
        size_t in = 0, prev_in = SIZE_MAX;
        uint32_t prev_cnt = 0;
        for (size_t n = k.nb; n < k.ne; ++n) {
          const uint32_t lo = t.ex_ptr[n], cnt = t.ex_ptr[n + 1] - lo;      // (ex_ptr[n + 1] of the chunk's last node belongs to the next chunk: already final)
          if (!cnt) continue;
          // (the same local ids as the previous tagged node's -- the next segment of the same exon: tmp holds their global form)
          const bool same = prev_in != SIZE_MAX && prev_cnt == cnt && std::equal(k.occ.begin() + long(in), k.occ.begin() + long(in + cnt), k.occ.begin() + long(prev_in));
          if (!same) {
            tmp.assign(k.occ.begin() + long(in), k.occ.begin() + long(in + cnt));
            for (uint32_t& x : tmp) x = ick[c].to_global[x];
            std::sort(tmp.begin(), tmp.end());
          }
          prev_in = in; prev_cnt = cnt;
          in += cnt;
          for (uint32_t j = 0; j < cnt; ++j) {
            const uint32_t id = tmp[j];
            t.ex_tag[lo + j] = id;
            t.ex_tx[lo + j] = extag_tx[id];
            t.ex_gene[lo + j] = t.tx_gene[extag_tx[id]];
          }
        }
      }
    });
  }
  }

  pt.lap("intern EX: per-node lists (parallel)");
  // JN tags: only junction links carry them (5e5 of 2.6e8 slots): the tags are interned in slot order by one thread, the
  // four per-slot offset arrays -- constant between two junction slots -- are filled by all of them
  std::vector<uint32_t> jslot;                      // junction slots (first slot of their pair), ascending
  {
    const unsigned C = A < 65536 ? 1u : threads;
    std::vector<std::vector<uint32_t>> part(C);
    parallel_chunks(A, C, [&](unsigned c, size_t sb, size_t se) {
      for (size_t s2 = sb; s2 < se; ++s2)
        if (t.first_slot[s2] == s2 && (t.lflags[s2] & L_JUNC)) part[c].push_back(uint32_t(s2));
    });
    for (auto& v : part) jslot.insert(jslot.end(), v.begin(), v.end());
  }
  pt.lap("intern JN: junction slots (parallel)");
  const size_t NJ = jslot.size();
  std::vector<uint32_t>& o_jn = t.jn_ptr; std::vector<uint32_t>& o_jt = t.jt_ptr;
  std::vector<uint32_t>& o_jx = t.jx_ptr; std::vector<uint32_t>& o_jg = t.jg_ptr;
  std::vector<uint8_t> j_keyerr(NJ, 0);
  if (reuse) {
    // The junction links are links of the GRAPH (the reads add links with ID, never with JN) and a junction's ordinal is its
    // rank among them in slot order, which the added links do not disturb: everything per junction ordinal carries over.
    if (previous->reuse.n_junctions != NJ) throw Error("InternalError", "junction count changed between two builds on one graph");
    o_jn = std::move(previous->jn_ptr); o_jt = std::move(previous->jt_ptr); o_jx = std::move(previous->jx_ptr); o_jg = std::move(previous->jg_ptr);
    t.jn_ftx = std::move(previous->jn_ftx); t.jn_span = std::move(previous->jn_span);
    t.jt = std::move(previous->jt); t.jx = std::move(previous->jx); t.jg = std::move(previous->jg);
    j_keyerr = std::move(previous->reuse.j_keyerr);
    for (size_t j = 0; j < NJ; ++j) if (j_keyerr[j]) t.lflags[jslot[j]] |= L_GENE_KEYERR;
  } else {
  o_jn.assign(NJ + 1, 0); o_jt.assign(NJ + 1, 0); o_jx.assign(NJ + 1, 0); o_jg.assign(NJ + 1, 0);
  {
    // per thread: the tags of its junctions and the two transcript names each tag yields (:98-104 and :356 / :595), interned
    // locally; merged to first-appearance ids (merge_interned); the transcripts only a JN tag names (none in a graph the
    // annotator wrote) are added behind the EX ones in that order; the per-junction sets are then built by all threads
    struct JEntry { uint32_t tag, tx, ftx; StrRef span; };
    struct JnChunk {
      size_t jb = 0, je = 0;
      std::vector<JEntry> ent;              // entries of all its junctions, file order
      std::vector<uint32_t> n_ent;          // per junction
      std::vector<uint32_t> jt, jx, jg, ftx, cnt;     // outputs: concatenated per junction; cnt = {jt, jx, jg} sizes per junction
      std::vector<StrRef> spans;
      std::vector<uint8_t> keyerr;
    };
    const unsigned JC = NJ < par_min(4096) ? 1u : threads;
    std::vector<JnChunk> jck(JC);
    std::vector<InternChunk> tagck(JC), txck(JC);
    parallel_chunks(NJ, JC, [&](unsigned c, size_t jb, size_t je) {
      JnChunk& k = jck[c];
      k.jb = jb; k.je = je;
      std::unordered_map<std::string_view, uint32_t> tag_ids, txn_ids;
      auto local = [&](std::unordered_map<std::string_view, uint32_t>& m, InternChunk& ic, Span sp) {
        auto ins = m.emplace(sp.sv(), uint32_t(ic.span.size()));
        if (ins.second) ic.span.push_back(StrRef{uint64_t(sp.b - base), uint32_t(sp.size())});
        return ins.first->second;
      };
      for (size_t j = jb; j < je; ++j) {
        const LinkAttr& a = la[attr_slot[jslot[j]]];
        const char* p = base + a.jn_off;
        const char* e = p + a.jn_len;
        uint32_t n = 0;
        for (;;) {
          const char* cm = static_cast<const char*>(memchr(p, ',', size_t(e - p)));
          Span tag{p, cm ? cm : e};
          JEntry en;
          en.tag = local(tag_ids, tagck[c], tag);
          en.tx = local(txn_ids, txck[c], tx_of_junction(tag));        // in this order: the order tx ids were handed out in
          en.ftx = local(txn_ids, txck[c], tx_for_filter(tag));
          en.span = StrRef{uint64_t(tag.b - base), uint32_t(tag.size())};
          k.ent.push_back(en);
          ++n;
          if (!cm) break;
          p = cm + 1;
        }
        k.n_ent.push_back(n);
      }
    });
    (void)merge_interned(base, tagck, threads);      // only the numbering matters (disjointness of JN sets, :402)
    const std::vector<StrRef> jtx = merge_interned(base, txck, threads);
    std::vector<uint32_t> jtx_id(jtx.size());
    for (size_t i = 0; i < jtx.size(); ++i) jtx_id[i] = tx_id(Span{base + jtx[i].off, base + jtx[i].off + jtx[i].len});
    parallel_chunks(JC, JC, [&](unsigned, size_t cb, size_t ce) {
      std::vector<uint32_t> tags, txs, genes;
      for (size_t c = cb; c < ce; ++c) {
        JnChunk& k = jck[c];
        size_t at = 0;
        for (size_t j = k.jb; j < k.je; ++j) {
          const uint32_t n = k.n_ent[j - k.jb];
          tags.clear(); txs.clear(); genes.clear();
          bool keyerr = false;
          for (uint32_t q = 0; q < n; ++q) {
            const JEntry& en = k.ent[at + q];
            tags.push_back(tagck[c].to_global[en.tag]);
            const uint32_t tx = jtx_id[txck[c].to_global[en.tx]];
            txs.push_back(tx);
            if (t.tx_gene[tx] == CNONE) keyerr = true; else genes.push_back(t.tx_gene[tx]);
            k.ftx.push_back(jtx_id[txck[c].to_global[en.ftx]]);
            k.spans.push_back(en.span);
          }
          at += n;
          sort_unique(tags); sort_unique(txs); sort_unique(genes);
          k.jt.insert(k.jt.end(), tags.begin(), tags.end());
          k.jx.insert(k.jx.end(), txs.begin(), txs.end());
          k.jg.insert(k.jg.end(), genes.begin(), genes.end());
          k.cnt.push_back(uint32_t(tags.size())); k.cnt.push_back(uint32_t(txs.size())); k.cnt.push_back(uint32_t(genes.size()));
          k.keyerr.push_back(keyerr ? 1 : 0);
        }
      }
    });
    // chunk outputs one behind the other (they are in junction order)
    std::vector<size_t> b_jn(JC + 1, 0), b_jt(JC + 1, 0), b_jx(JC + 1, 0), b_jg(JC + 1, 0);
    for (unsigned c = 0; c < JC; ++c) {
      b_jn[c + 1] = b_jn[c] + jck[c].ftx.size(); b_jt[c + 1] = b_jt[c] + jck[c].jt.size();
      b_jx[c + 1] = b_jx[c] + jck[c].jx.size(); b_jg[c + 1] = b_jg[c] + jck[c].jg.size();
    }
    if (b_jn[JC] >= size_t(CNONE)) throw Error("Unsupported", "more than 2^32-2 JN tag entries");
    t.jn_ftx.resize(b_jn[JC]); t.jn_span.resize(b_jn[JC]); t.jt.resize(b_jt[JC]); t.jx.resize(b_jx[JC]); t.jg.resize(b_jg[JC]);
    parallel_chunks(JC, JC, [&](unsigned, size_t cb, size_t ce) {
      for (size_t c = cb; c < ce; ++c) {
        const JnChunk& k = jck[c];
        std::copy(k.ftx.begin(), k.ftx.end(), t.jn_ftx.begin() + long(b_jn[c]));
        std::copy(k.spans.begin(), k.spans.end(), t.jn_span.begin() + long(b_jn[c]));
        std::copy(k.jt.begin(), k.jt.end(), t.jt.begin() + long(b_jt[c]));
        std::copy(k.jx.begin(), k.jx.end(), t.jx.begin() + long(b_jx[c]));
        std::copy(k.jg.begin(), k.jg.end(), t.jg.begin() + long(b_jg[c]));
        size_t on = b_jn[c], ot = b_jt[c], ox = b_jx[c], og = b_jg[c];
        for (size_t j = k.jb; j < k.je; ++j) {
          o_jn[j] = uint32_t(on); o_jt[j] = uint32_t(ot); o_jx[j] = uint32_t(ox); o_jg[j] = uint32_t(og);
          on += k.n_ent[j - k.jb];
          ot += k.cnt[3 * (j - k.jb)]; ox += k.cnt[3 * (j - k.jb) + 1]; og += k.cnt[3 * (j - k.jb) + 2];
          if (k.keyerr[j - k.jb]) { t.lflags[jslot[j]] |= L_GENE_KEYERR; j_keyerr[j] = 1; }
        }
      }
    });
    o_jn[NJ] = uint32_t(b_jn[JC]); o_jt[NJ] = uint32_t(b_jt[JC]); o_jx[NJ] = uint32_t(b_jx[JC]); o_jg[NJ] = uint32_t(b_jg[JC]);
  }
  }
  pt.lap("intern JN: tags of the junctions (sharded merge, parallel sets)");
  par_assign(t.jrow, A, uint32_t(CNONE), threads);
  for (size_t j = 0; j < NJ; ++j) t.jrow[jslot[j]] = uint32_t(j);

  pt.lap("intern JN: slot -> junction map");
  // ---- lexicographic rank of segment names ------------------------------------------------------------
  if (reuse) t.rank = std::move(previous->rank);
  else {
  t.rank.resize(V);
  bool fixed_width = g.names.dense && V > 0;
  if (fixed_width) {
    uint16_t w = g.seg_name_len[0];
    for (uint32_t n = 0; n < V && fixed_width; ++n) fixed_width = g.seg_name_len[n] == w;
  }
  if (fixed_width) {
    // equal-length canonical decimals: string order == numeric order, and a name's number is its place in the dense name table
    // (read from there: parsing 2.5e8 names back out of the text cost 0.18 s of cache misses)
    const size_t TS = g.names.table.size();
    parallel_chunks(TS, TS < 4096 ? 1 : threads, [&](unsigned, size_t ib, size_t ie) {
      for (size_t i = ib; i < ie; ++i) {
        const uint32_t n = g.names.table[i];
        if (n != NONE32) t.rank[n] = uint32_t(i);
      }
    });
  } else {
    std::vector<uint32_t> order(V);
    for (uint32_t n = 0; n < V; ++n) order[n] = n;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return g.name_of(a).sv() < g.name_of(b).sv(); });
    for (uint32_t i = 0; i < V; ++i) t.rank[order[i]] = i;
  }
  }

  pt.lap("name ranks");
  // ---- the walks' fast step: inside an intron nearly every segment has exactly one out-neighbour, and the rule "first out-neighbour
  // that ..., else the last" (:488-499, :843-852) then takes it whatever the test says
  t.nxt1.resize(V);
  parallel_chunks(V, V < 4096 ? 1 : threads, [&](unsigned, size_t b, size_t e) {
    for (size_t n = b; n < e; ++n) {
      uint32_t only = CNONE, cnt = 0;
      for (uint32_t s = t.row_ptr[n]; s < t.row_ptr[n + 1] && cnt < 2; ++s) {
        if (!t.hidden.empty() && (t.hidden[s] & 1u)) continue;
        only = t.col[s];
        ++cnt;
      }
      t.nxt1[n] = cnt == 1 ? only : CNONE;
    }
  });
  pt.lap("single out-neighbours");
  // ---- junction / novel lists in file order; IR-2 probes -----------------------------------------------
  {
    const unsigned C = A < 65536 ? 1u : threads;
    std::vector<std::vector<uint32_t>> pj(C), pn(C);
    parallel_chunks(A, C, [&](unsigned c, size_t sb, size_t se) {
      for (size_t s2 = sb; s2 < se; ++s2) {
        if (t.first_slot[s2] != s2) continue;
        if (t.lflags[s2] & L_JUNC) pj[c].push_back(uint32_t(s2));
        if (t.lflags[s2] & L_NOVEL) pn[c].push_back(uint32_t(s2));
      }
    });
    for (unsigned c = 0; c < C; ++c) {
      t.junctions.insert(t.junctions.end(), pj[c].begin(), pj[c].end());
      t.novels.insert(t.novels.end(), pn[c].begin(), pn[c].end());
    }
  }
  auto by_off = [&](uint32_t a, uint32_t b) { return t.order[a] < t.order[b]; };
  std::sort(t.junctions.begin(), t.junctions.end(), by_off);
  std::sort(t.novels.begin(), t.novels.end(), by_off);
  t.probe.assign(t.junctions.size() * 6, CNONE);
  t.probe_ok.assign(t.junctions.size(), 0);
  auto pair_slot = [&](uint32_t u, uint32_t v) -> uint32_t {
    for (uint32_t s = t.row_ptr[u]; s < t.row_ptr[u + 1]; ++s) if (t.col[s] == v) return s;
    return CNONE;
  };
  parallel_chunks(t.junctions.size(), t.junctions.size() < 4096 ? 1 : threads, [&](unsigned, size_t jb, size_t je) {
    for (size_t j = jb; j < je; ++j) {
      const uint32_t s = t.junctions[j];
      const uint32_t u = t.src[s], v = t.col[s];
      int64_t iu, iv;
      char buf[32];
      if (parse_py_int(g.name_of(u), iu)) {
        t.probe_ok[j] |= 1u;
        for (int d = 1; d <= 3; ++d) {
          int n = snprintf(buf, sizeof buf, "%lld", (long long)(iu + d));
          uint32_t x = g.names.find(Span{buf, buf + n});
          if (x != CNONE) t.probe[j * 6 + size_t(d - 1)] = pair_slot(u, x);
        }
      }
      if (parse_py_int(g.name_of(v), iv)) {
        t.probe_ok[j] |= 2u;
        for (int d = 3; d >= 1; --d) {
          int n = snprintf(buf, sizeof buf, "%lld", (long long)(iv - d));
          uint32_t x = g.names.find(Span{buf, buf + n});
          if (x != CNONE) t.probe[j * 6 + 3 + size_t(3 - d)] = pair_slot(x, v);
        }
      }
    }
  });
  pt.lap("lists + probes");
  t.parse_s = now_s() - t0;
  if (previous) previous->reuse.valid = false;      // whatever was taken out of it, it is not a set of tables any more
  if (aug) {
    // what the next build on this graph may take over (the handle keeps these tables until then)
    CallTables::Reusable& r = t.reuse;
    r.gtf_hash = gtf_hash; r.gtf_size = gtf_size; r.graph_A = g.A; r.V = g.V; r.text = base; r.n_junctions = NJ;
    r.la_graph = std::move(la_graph);
    r.j_keyerr = std::move(j_keyerr);
    r.valid = true;
  }
  if (size_t(V) + A > (size_t(1) << 24)) {
    // the per-line tag records of a whole-genome graph are ~13 GB: handed to a thread of its own to give back (a second)
    auto* junk = new std::tuple<RawVector<SegAttr>, RawVector<LinkAttr>, RawVector<uint32_t>>(std::move(sa), std::move(la), std::move(attr_slot));
    defer_free(size_t(V) * sizeof(SegAttr) + A * (sizeof(LinkAttr) + 4), [junk] { delete junk; });      // (common.h: when the output is being written)
  }
}

CallView host_view(const Graph& g, const CallTables& t, const CallOptions& o) {
  CallView c{};
  c.V = g.V;
  c.row_ptr = t.row_ptr; c.col = t.col;
  c.irow_ptr = t.irow_ptr; c.icol = t.icol; c.islot = t.islot;
  c.first_slot = t.first_slot.data(); c.src = t.src.data(); c.lflags = t.lflags.data();
  c.hid = t.hidden.empty() ? nullptr : t.hidden.data();
  c.nxt1 = t.nxt1.data();
  c.run_end_rev = nullptr; c.pl = nullptr; c.pnc = nullptr; c.rank_monotone = 0; c.share_ex = nullptr; c.iflags = nullptr; c.stamps = nullptr; c.j_ht_n = nullptr; c.j_ht = nullptr; c.j_carried = nullptr; c.oj_ptr = c.oj = c.ij_ptr = c.ij = nullptr; c.frontier_lds = 0;      // (made on the device, call.hip build_walk_runs / build_share_bits; the CPU harness makes its own)
  c.rc = t.rc.data(); c.nc = t.nc.data(); c.nflags = t.nflags.data(); c.ln = g.seg_len.data(); c.rank = t.rank.data();
  c.ex_ptr = t.ex_ptr.data(); c.ex_tag = t.ex_tag.data(); c.ex_tx = t.ex_tx.data(); c.ex_gene = t.ex_gene.data();
  c.jrow = t.jrow.data();
  c.jn_ptr = t.jn_ptr.data(); c.jn_ftx = t.jn_ftx.data();
  c.jt_ptr = t.jt_ptr.data(); c.jt = t.jt.data();
  c.jx_ptr = t.jx_ptr.data(); c.jx = t.jx.data();
  c.jg_ptr = t.jg_ptr.data(); c.jg = t.jg.data();
  c.tx_gene = t.tx_gene.data();
  c.probe = t.probe.data(); c.probe_ok = t.probe_ok.data();
  c.rca = o.rca; c.isw = o.isw; c.minintron = o.minintron; c.events = o.events; c.debug = o.debug ? 1u : 0u;
  c.arena = nullptr; c.arena_next = nullptr; c.arena_slots = 0; c.arena_cap = 0;
  return c;
}

void pass_lists(const Graph& g, const CallTables& t, const CallOptions& o, std::vector<uint32_t>& annotated,
                std::vector<uint32_t>& novel, std::vector<uint32_t>& around, std::vector<uint32_t>& around_jidx) {
  annotated.clear(); novel.clear(); around.clear(); around_jidx.clear();
  // Several S lines for one name: without --novel the reference is deterministic (the adjacency lists restart at the last
  // S line, hidden slots above); with it, check_novel reaches the else branches of :623-631 / :677-685, the second of which
  // reads a loop variable left over from an earlier junction of a SET iteration -- nothing to be bit-exact with
  if (!t.hidden.empty() && o.novel)
    throw Error("Unsupported", "several S lines for one segment name with --novel: what the reference prints depends on its hash seed (call.py:681-685)");
  auto matches = [&](uint32_t s) {
    if (!o.has_junction) return true;
    Span a = g.name_of(t.src[s]), b = g.name_of(t.col[s]);
    std::string k(a.b, a.size());
    k += '-';
    k.append(b.b, b.size());
    return k == o.junction;
  };
  if (!o.no_annotated) {
    // :241 compares the LITERAL "f{_j[0]}-{_j[1]}" (missing f-prefix) with --junction
    if (!o.has_junction || o.junction == "f{_j[0]}-{_j[1]}") annotated = t.junctions;
  }
  if (o.novel) {
    for (uint32_t s : t.novels) if (matches(s)) novel.push_back(s);
    for (size_t j = 0; j < t.junctions.size(); ++j)
      if (matches(t.junctions[j])) { around.push_back(t.junctions[j]); around_jidx.push_back(uint32_t(j)); }
  }
  if (o.shard_world > 1) {
    // junctions and novel links are evaluated independently of each other: deal them round-robin
    auto keep = [&](std::vector<uint32_t>& v, std::vector<uint32_t>* w) {
      size_t k = 0;
      for (size_t i = 0; i < v.size(); ++i)
        if (i % o.shard_world == o.shard_rank) { v[k] = v[i]; if (w) (*w)[k] = (*w)[i]; ++k; }
      v.resize(k);
      if (w) w->resize(k);
    };
    keep(annotated, nullptr);
    keep(novel, nullptr);
    keep(around, &around_jidx);
  }
}

void cassette_items(const CallTables& t, const std::vector<uint32_t>& around, std::vector<uint32_t>& junction,
                    std::vector<uint32_t>& head_link) {
  // (a row scan per junction -- 40 ms on one core at S4, where a hub's row is read once per junction that leaves it: chunks of
  // the list on the host threads, joined in list order)
  const unsigned T = around.size() < 50000 ? 1u : hw_threads();
  std::vector<std::vector<uint32_t>> js(T), hs(T);
  parallel_chunks(around.size(), T, [&](unsigned c, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) {
      const uint32_t s = around[i], u = t.src[s];
      for (uint32_t t1 = t.row_ptr[u]; t1 < t.row_ptr[u + 1]; ++t1) {
        if (t.first_slot[t1] != t1 || t1 == s || !(t.lflags[t1] & L_NOVEL)) continue;
        js[c].push_back(s);
        hs[c].push_back(t1);
      }
    }
  });
  size_t n = 0;
  for (const auto& v : js) n += v.size();
  junction.clear();
  head_link.clear();
  junction.reserve(n);
  head_link.reserve(n);
  for (unsigned c = 0; c < T; ++c) {
    junction.insert(junction.end(), js[c].begin(), js[c].end());
    head_link.insert(head_link.end(), hs[c].begin(), hs[c].end());
  }
}

namespace {

inline void put_u(std::string& s, uint64_t v) {
  char buf[24];
  int n = 0;
  do { buf[n++] = char('0' + v % 10); v /= 10; } while (v);
  while (n) s.push_back(buf[--n]);
}

struct Renderer {
  const Graph& g;
  const CallTables& t;
  const char* base;
  std::string& out;
  void name(uint32_t node) { Span n = g.name_of(node); out.append(n.b, n.size()); }
  void link_nodes(uint32_t s) { name(t.src[s]); out.push_back('>'); name(t.col[s]); }
  void link_rc(uint32_t s) { put_u(out, uint64_t(t.rc[s])); }
  // (t.jn_ptr is per junction ordinal: a slot that is no junction has no JN entries)
  uint32_t jn_lo(uint32_t s) const { const uint32_t j = t.jrow[s]; return j == CNONE ? 0u : t.jn_ptr[j]; }
  uint32_t jn_hi(uint32_t s) const { const uint32_t j = t.jrow[s]; return j == CNONE ? 0u : t.jn_ptr[j + 1]; }
  void jn_all(uint32_t s) {
    for (uint32_t i = jn_lo(s); i < jn_hi(s); ++i) {
      if (i != jn_lo(s)) out.push_back('|');
      out.append(base + t.jn_span[i].off, t.jn_span[i].len);
    }
  }
  // payload: count, then indices into the link's JN list
  const uint32_t* jn_some(uint32_t s, const uint32_t* p) {
    uint32_t n = *p++;
    for (uint32_t i = 0; i < n; ++i) {
      if (i) out.push_back('|');
      const StrRef& r = t.jn_span[jn_lo(s) + p[i]];
      out.append(base + r.off, r.len);
    }
    return p + n;
  }
  // n segments: a word is a segment, or PATH_RANGE | first with last behind it (call_device.h: the walks write a run of
  // consecutive segments as its two ends)
  void path(const uint32_t* p, uint32_t n) {
    for (uint32_t done = 0; done < n; ++p) {
      uint32_t a = *p, b = a;
      if (a & PATH_RANGE) { a &= ~PATH_RANGE; b = *++p; }
      for (uint32_t x = a; x <= b; ++x, ++done) { if (done) out.push_back('>'); name(x); }
    }
  }
  void extag(uint32_t id) { out.append(base + t.extag_span[id].off, t.extag_span[id].len); }
  void head(const char* type, const char* kind, uint32_t gene, const char* strand_plus, const char* strand_other) {
    const std::string& st = t.gene_strand[gene];
    out += type ? type : (st == "+" ? strand_plus : strand_other);
    out.push_back(',');
    out += kind;
    out.push_back(',');
    out += t.gene_chrom[gene];
    out.push_back(',');
    out += t.gene_name[gene];
    out.push_back(',');
    out += st;
    out.push_back(',');
  }
};

}  // namespace

static void render_range(const Graph& g, const CallTables& t, const EventRec* recs, size_t lo, size_t hi, const uint32_t* side,
                         const TaskOut* tasks, const uint32_t* paths, std::string& csv, std::string& err);

void render_events(const Graph& g, const CallTables& t, const EventRec* recs, size_t n, const uint32_t* side,
                   const TaskOut* tasks, const uint32_t* paths, std::string& csv, std::string& err) {
  const unsigned T = n < 100000 ? 1 : hw_threads();
  if (T == 1) { render_range(g, t, recs, 0, n, side, tasks, paths, csv, err); return; }
  std::vector<std::string> parts(T), errs(T);
  parallel_chunks(n, T, [&](unsigned c, size_t lo, size_t hi) { render_range(g, t, recs, lo, hi, side, tasks, paths, parts[c], errs[c]); });
  size_t total = csv.size();
  for (auto& s2 : parts) total += s2.size();
  csv.reserve(total);
  for (unsigned c = 0; c < T; ++c) { csv += parts[c]; std::string().swap(parts[c]); err += errs[c]; }
}

void render_events_to_fd(const Graph& g, const CallTables& t, const EventRec* recs, size_t n, const uint32_t* side,
                         const TaskOut* tasks, const uint32_t* paths, int fd_csv, std::string& err) {
  constexpr size_t BLOCK = 16384;
  const size_t nb = (n + BLOCK - 1) / BLOCK;
  const unsigned T = unsigned(std::max<size_t>(1, std::min<size_t>(n < 100000 ? 1 : writer_threads(), nb)));
  // blocks of rows are rendered by a thread pool into buffers the threads keep and land in row order through
  // OrderedOutput (common.h): on a regular file by many threads at once.  The bytes of a row are not known before it is
  // rendered: ~200 per row are expected, and the file is mapped (sparse) with room for 16 times that -- a run that
  // needs more finishes through pwrite.
  OrderedOutput sink(fd_csv, nb, uint64_t(n) * 200, uint64_t(n) * 3200 + (uint64_t(1) << 30));
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  std::mutex mu;                 // first_error
  std::unique_ptr<Error> first_error;
  std::vector<std::string> err_parts(nb);      // "Skipping ..." lines per block, joined in block order at the end
  auto work = [&] {
    std::string buf;
    for (;;) {
      const size_t b = next.fetch_add(1);
      if (b >= nb) return;
      buf.clear();
      std::unique_ptr<Error> mine;
      try {
        if (!failed.load(std::memory_order_relaxed))
          render_range(g, t, recs, b * BLOCK, std::min(n, (b + 1) * BLOCK), side, tasks, paths, buf, err_parts[b]);
      } catch (const Error& x) {
        mine.reset(new Error(x));
      }
      if (mine || failed.load(std::memory_order_relaxed)) buf.clear();
      try {
        sink.put(b, buf.data(), buf.size());
      } catch (const Error& x) {
        if (!mine) mine.reset(new Error(x));
      }
      if (mine) {
        std::lock_guard<std::mutex> lk(mu);
        if (!first_error) first_error = std::move(mine);
        failed.store(true);
      }
    }
  };
  if (T == 1) work();
  else {
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < T; ++i) pool.emplace_back(work);
    for (auto& th : pool) th.join();
  }
  if (first_error) throw *first_error;
  sink.finish();
  for (const std::string& e : err_parts) err += e;
}

static void render_range(const Graph& g, const CallTables& t, const EventRec* recs, size_t lo, size_t hi, const uint32_t* side,
                         const TaskOut* tasks, const uint32_t* paths, std::string& csv, std::string& err) {
  Renderer R{g, t, g.text.data, csv};
  for (size_t i = lo; i < hi; ++i) {
    const EventRec& r = recs[i];
    const uint32_t* p = side + r.side_off;
    std::string& o = csv;
    const TaskOut* tk = nullptr;
    if (r.type == T_IR_ANN || r.type == T_IR1_NOV || r.type == T_IR2_NOV) {
      tk = &tasks[r.j1];
      if (tk->status == 2) { err += "Skipping novel IR due to multiple genes\n"; continue; }
      if (tk->status != 1) continue;
    }
    switch (r.type) {
      case T_ES_ANN:
        R.head("ES", "annotated", r.gene, nullptr, nullptr);
        R.jn_all(r.j0); o.push_back(',');
        p = R.jn_some(r.j1, p); o.push_back(',');
        p = R.jn_some(r.j2, p); o.push_back(',');
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o.push_back(',');
        R.link_nodes(r.j2); o.push_back(','); R.link_rc(r.j2);
        break;
      case T_SS_ANN_A:
      case T_SS_ANN_B:
        R.head(nullptr, "annotated", r.gene, r.type == T_SS_ANN_A ? "A5" : "A3", r.type == T_SS_ANN_A ? "A3" : "A5");
        R.jn_all(r.j0); o.push_back(',');
        R.jn_all(r.j1); o += ",.,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o += ",.,.";
        break;
      case T_IR_ANN:
        R.head("IR", "annotated", r.gene, nullptr, nullptr);
        R.jn_all(r.j0); o += ",.,.,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.path(paths + tk->path_off, tk->np); o.push_back(','); put_u(o, tk->mean); o += ",.,.";
        break;
      case T_ES_NOV:
        R.head("ES", "novel", r.gene, nullptr, nullptr);
        o += "?,";
        p = R.jn_some(r.j1, p); o.push_back(',');
        p = R.jn_some(r.j2, p); o.push_back(',');
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o.push_back(',');
        R.link_nodes(r.j2); o.push_back(','); R.link_rc(r.j2);
        break;
      case T_SS1_NOV_A:
      case T_SS1_NOV_B:
        R.head(nullptr, "novel", r.gene, r.type == T_SS1_NOV_A ? "A3" : "A5", r.type == T_SS1_NOV_A ? "A5" : "A3");
        o += "?,";
        R.jn_all(r.j1); o += ",.,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o += ",.,.";
        break;
      case T_SS2_NOV:
        // j0 = the annotated junction, j1 = the novel link; both branches print "A3" on '+' (:767, :820)
        R.head(nullptr, "novel", r.gene, "A3", "A5");
        R.jn_all(r.j0); o += ",?,.,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o += ",.,.";
        break;
      case T_IR1_NOV: {
        R.head("IR", "novel", r.gene, nullptr, nullptr);
        o += "?,";
        const uint32_t* tags = p;
        for (uint32_t k = 0; k < tags[0]; ++k) { if (k) o.push_back('|'); R.extag(tags[1 + k]); }
        o += ",.,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.path(paths + tk->path_off, tk->np); o.push_back(','); put_u(o, tk->mean); o += ",.,.";
        break;
      }
      case T_CE_NOV:
        R.head("CE", "novel", r.gene, nullptr, nullptr);
        R.jn_all(r.j0); o += ",?,?,";
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o.push_back(',');
        R.link_nodes(r.j1); o.push_back(','); R.link_rc(r.j1); o.push_back(',');
        R.link_nodes(r.j2); o.push_back(','); R.link_rc(r.j2);
        break;
      case T_IR2_NOV: {
        R.head("IR", "novel", r.gene, nullptr, nullptr);
        const uint32_t* pr = p;
        for (uint32_t k = 0; k < pr[0]; ++k) {
          if (k) o.push_back('|');
          R.extag(pr[1 + 2 * k]);
          o.push_back('.');
          // e1.split(".")[-1]
          const StrRef& b = t.extag_span[pr[2 + 2 * k]];
          const char* s = R.base + b.off;
          const char* e = s + b.len;
          const char* d = e;
          while (d > s && d[-1] != '.') --d;
          o.append(d, size_t(e - d));
        }
        o += ",?,.,";
        R.path(paths + tk->path_off, tk->np); o.push_back(','); put_u(o, tk->mean); o.push_back(',');
        R.link_nodes(r.j0); o.push_back(','); R.link_rc(r.j0); o += ",.,.";
        break;
      }
      case T_SKIP_ES: err += "Skipping ES due to multiple genes\n"; continue;
      case T_SKIP_SS: err += "Skipping SS due to multiple genes\n"; continue;
      case T_SKIP_IR: err += "Skipping novel IR due to multiple genes\n"; continue;
      case T_DBG_ANN_J: case T_DBG_NOV_J: {
        err += r.type == T_DBG_ANN_J ? "Checking annotated junction " : "Checking novel junction ";
        Span a = g.name_of(t.src[r.j0]), b = g.name_of(t.col[r.j0]);
        err.append(a.b, a.size()); err += " -> "; err.append(b.b, b.size()); err.push_back('\n');
        continue;
      }
      case T_DBG_ANN_ES: err += "Checking annotated ES\n"; continue;
      case T_DBG_ANN_SS: err += "Checking annotated SS\n"; continue;
      case T_DBG_ANN_IR: err += "Checking annotated IR\n"; continue;
      case T_DBG_NOV_ES: err += "Checking novel ES\n"; continue;
      case T_DBG_NOV_SS1: err += "Checking novel SS (1)\n"; continue;
      case T_DBG_NOV_SS2: err += "Checking novel SS (2)\n"; continue;
      case T_DBG_NOV_IR1: err += "Checking novel IR (1)\n"; continue;
      case T_DBG_NOV_CE: err += "Checking novel CE\n"; continue;
      case T_DBG_NOV_IR2: err += "Checking novel IR (2)\n"; continue;
      default: throw Error("InternalError", "unknown event record type");
    }
    o.push_back('\n');
  }
}

}  // namespace pantas
