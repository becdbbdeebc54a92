// Host side of the call path: tag/GTF parsing and interning (scripts/call.py:25-69,145-216),
// the tables the kernels read, and the CSV renderer (:218-234 and every print(..., sep=",")).
#pragma once
#include "call_device.h"
#include "common.h"
#include "graph.h"
#include "augment.h"

namespace pantas {

struct CallOptions {
  int32_t rca = -1;            // --rca
  int32_t isw = 5;             // --isw
  int32_t minintron = 100;     // --minintronsize
  uint32_t events = EV_ES | EV_SS | EV_IR;   // --events
  bool novel = false;          // --novel
  bool no_annotated = false;   // --no-annotated
  bool debug = false;          // --debug: the reference's "Checking ..." lines on stderr
  bool has_junction = false;   // --junction given
  std::string junction;        // "u-v"
  uint32_t shard_rank = 0, shard_world = 1;   // multi-GPU: this process evaluates every world-th junction / novel link
};

struct StrRef { uint64_t off; uint32_t len; };

// tag record of one L line (call_host.cpp fills it; kept per graph slot between builds on the same graph, see CallTables::Reusable)
using LinkAttr = LinkTagRec;      // (graph.h: the GPU loader can fill them)

struct CallTables {
  // ---- link layout the call path works on: the graph's CSR, or (fused augment -> call) a copy
  //      extended by the links the reads added, which the reference appends after the GFA's
  //      own lines (alignments_augmentation_from_gaf.py:426-427) ------------------------------
  size_t A = 0;
  const uint32_t* row_ptr = nullptr;
  const uint32_t* col = nullptr;
  const uint32_t* irow_ptr = nullptr;
  const uint32_t* icol = nullptr;
  const uint32_t* islot = nullptr;
  const uint64_t* order = nullptr;           // file-order key of a slot (byte offset; novel links after the text)
  RawVector<uint32_t> x_row_ptr, x_col, x_irow_ptr, x_icol, x_islot;
  RawVector<uint64_t> x_order;
  // ---- mirrored to the device -----------------------------------------------------------
  // (RawVector: [V] / [A] arrays, gigabytes at whole-genome scale -- huge pages, filled by many threads)
  RawVector<uint32_t> first_slot, src, lflags;
  RawVector<uint8_t> hidden;     // [A], only for a graph with several S lines for one name: bit 0 -- the slot's L line stands in front
                                 // of the LAST S line of its source (it is in gfaL, not in gfaS[u]["O"]: call.py:153-154); bit 1 -- of its target
  RawVector<int32_t> rc, nc;
  RawVector<uint32_t> nflags, rank;
  RawVector<uint32_t> nxt1;      // [V] the segment's ONE out-neighbour, or CNONE when it has none or several (the intron walks' fast step)
  RawVector<uint32_t> ex_ptr;
  std::vector<uint32_t> ex_tag, ex_tx, ex_gene;
  RawVector<uint32_t> jrow;                                // [A] junction ordinal of a slot or CNONE
  std::vector<uint32_t> jn_ptr, jt_ptr, jx_ptr, jg_ptr;    // per junction ordinal (+ 1)
  std::vector<uint32_t> jn_ftx, jt, jx, jg;
  std::vector<uint32_t> tx_gene;
  std::vector<uint32_t> junctions, novels;   // canonical slots in file order
  std::vector<uint32_t> probe, probe_ok;
  // ---- rendering only ----------------------------------------------------------------------
  std::vector<StrRef> jn_span;               // per JN entry (file order, aligned with jn_ptr)
  std::vector<StrRef> extag_span;            // per EX tag id
  std::vector<std::string> gene_name, gene_strand, gene_chrom;
  double parse_s = 0;
  // ---- what a later build on the SAME graph and GTF takes over instead of computing it again (replicates: the tags of the graph
  //      and the GTF do not change between augment runs; counters, added links and everything indexed by slot do).  Filled by a
  //      fused build (`aug` given); a build that is handed these tables as `previous` MOVES the arrays out of them.
  struct Reusable {
    bool valid = false;
    uint64_t gtf_hash = 0;
    size_t gtf_size = 0, graph_A = 0, n_junctions = 0;
    uint32_t V = 0;
    const void* text = nullptr;           // the graph's mapped text: spans point into it
    RawVector<LinkAttr> la_graph;         // tag record of every slot of the GRAPH (before the counters of a run replace RC)
    std::vector<uint8_t> j_keyerr;        // per junction ordinal: a transcript of its JN tags is missing from the GTF
  } reuse;
  bool reused_previous = false;           // this build took the previous tables' static part
};

// Parses the tags of an (augmented) GFA already loaded as `g` plus the GTF.  Throws Error where
// the reference's parsers raise.
void build_call_tables(const Graph& g, const char* gtf_path, CallTables& t, unsigned threads = 0);
// `aug` (optional): results of an augment run on the same graph.  NC / RC then come from its
// counters and its novel links join the layout, exactly as if the augmented GFA text had been
// written and parsed again.
// `previous` (optional, fused builds): the tables of an earlier fused build on this graph; when its GTF was the same bytes, the
// graph-and-GTF-only part (tag records, EX / JN interning, name ranks, gene table) is moved over from it instead of recomputed.
// `previous` is left unusable either way.
void build_call_tables_mem(const Graph& g, const char* gtf_data, size_t gtf_size, CallTables& t, unsigned threads = 0,
                           const AugmentResult* aug = nullptr, CallTables* previous = nullptr);

// A CallView over the host arrays (used by the CPU test harness; the product builds one over
// device copies in call.hip).
CallView host_view(const Graph& g, const CallTables& t, const CallOptions& o);

// The slots each pass iterates, after the --junction filter (:241, :532, :897)
void pass_lists(const Graph& g, const CallTables& t, const CallOptions& o, std::vector<uint32_t>& annotated,
                std::vector<uint32_t>& novel, std::vector<uint32_t>& around, std::vector<uint32_t>& around_jidx);
// work items of the cassette-exon pass: (junction slot, canonical novel link leaving its head), junction-major
void cassette_items(const CallTables& t, const std::vector<uint32_t>& around, std::vector<uint32_t>& junction,
                    std::vector<uint32_t>& head_link);

// CSV header + rows (stdout of the reference) and the "Skipping ..." lines (stderr)
// tasks[recs[i].j1] / paths hold the outcome of the deferred intron walks of IR records
void render_events(const Graph& g, const CallTables& t, const EventRec* recs, size_t n, const uint32_t* side,
                   const TaskOut* tasks, const uint32_t* paths, std::string& csv, std::string& err);
// The same rows straight to a file descriptor: blocks of rows are rendered by a pool of threads into buffers they keep,
// and written in row order as they become due -- no row ever sits in a 10 GB string (68 M rows at whole-genome scale),
// whose growth and page faults cost more than the formatting.  fd < 0: rendered (errors are found) and dropped.
void render_events_to_fd(const Graph& g, const CallTables& t, const EventRec* recs, size_t n, const uint32_t* side,
                         const TaskOut* tasks, const uint32_t* paths, int fd_csv, std::string& err);
extern const char* const CALL_CSV_HEADER;

const char* call_error_kind(uint32_t code);
const char* call_error_text(uint32_t code);

// emitters shared by call.hip and the CPU harness
struct CountEmit {
  uint32_t recs = 0, words = 0, tasks = 0;
  PANTAS_HD void begin(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { ++recs; }
  PANTAS_HD void task() { ++tasks; }
  PANTAS_HD void push(uint32_t) { ++words; }
  PANTAS_HD void end(uint32_t) {}
};
// Evaluate ONCE (round 3): what an item emits is kept in a small staging area of its own (STAGE_RECS records, STAGE_WORDS payload
// words) while it is counted; the write pass then only moves it to its scanned place.  An item that emits more than fits is
// counted in full all the same (`fits` cleared) and evaluated a second time by the write pass, as every item was before.
constexpr uint32_t STAGE_RECS = 4, STAGE_WORDS = 96;
// Round 5: an item that outgrows its own area moves on to a chunk of a pool -- 64 records / 1 024 words, then 4 096 / 16 384 -- and
// is still evaluated once; only an item that finds no chunk (or outgrows the large one) is evaluated again.  The items that did not
// fit were the expensive ones: the junctions at rows of thousands of links write thousands of rows, and the second evaluation
// was a tail as long as the first.
constexpr uint32_t STAGE_MID_RECS = 64, STAGE_MID_WORDS = 1024, STAGE_BIG_RECS = 4096, STAGE_BIG_WORDS = 16384;
struct StagePool {
  EventRec* mid_rec; uint32_t* mid_side; uint32_t mid_n;       // mid_n chunks of STAGE_MID_RECS / STAGE_MID_WORDS
  EventRec* big_rec; uint32_t* big_side; uint32_t big_n;
  uint32_t* taken;                                             // [2]: chunks handed out (mid, big)
};
// what the stage kernel leaves per item: where its records stand
enum : uint8_t { FITS_NO = 0, FITS_OWN = 1, FITS_MID = 2, FITS_BIG = 3 };
struct StageEmit {
  EventRec* rec;          // cap_r slots
  uint32_t* side;         // cap_w slots
  StagePool pool;
  uint32_t cap_r = STAGE_RECS, cap_w = STAGE_WORDS;
  uint32_t recs = 0, words = 0, tasks = 0;
  uint32_t chunk = 0;     // of the pool, once where == FITS_MID / FITS_BIG
  uint8_t where = FITS_OWN;
  bool fits = true;
  // room for one more record / word: the next chunk up, with what is staged so far (an open record included)
  PANTAS_HD bool grow() {
    if (!fits || where == FITS_BIG || pool.taken == nullptr) return false;
    const bool to_mid = where == FITS_OWN;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t k = atomicAdd(pool.taken + (to_mid ? 0 : 1), 1u);
#else
    const uint32_t k = pool.taken[to_mid ? 0 : 1]++;
#endif
    if (k >= (to_mid ? pool.mid_n : pool.big_n)) return false;
    EventRec* nr = to_mid ? pool.mid_rec + size_t(k) * STAGE_MID_RECS : pool.big_rec + size_t(k) * STAGE_BIG_RECS;
    uint32_t* ns = to_mid ? pool.mid_side + size_t(k) * STAGE_MID_WORDS : pool.big_side + size_t(k) * STAGE_BIG_WORDS;
    const uint32_t nrec = recs < cap_r ? recs + 1u : cap_r;      // (the record that is open, if it has a slot)
    for (uint32_t i = 0; i < nrec; ++i) nr[i] = rec[i];
    const uint32_t nw = words < cap_w ? words : cap_w;
    for (uint32_t i = 0; i < nw; ++i) ns[i] = side[i];
    rec = nr; side = ns;
    cap_r = to_mid ? STAGE_MID_RECS : STAGE_BIG_RECS;
    cap_w = to_mid ? STAGE_MID_WORDS : STAGE_BIG_WORDS;
    where = to_mid ? FITS_MID : FITS_BIG;
    chunk = k;
    return true;
  }
  PANTAS_HD void begin(uint32_t type, uint32_t gene, uint32_t j0, uint32_t j1, uint32_t j2) {
    if (fits && recs >= cap_r && !grow()) fits = false;
    if (!fits) return;
    EventRec& r = rec[recs];
    r.type = type; r.gene = gene; r.j0 = j0; r.j1 = j1; r.j2 = j2; r.side_off = words; r.side_len = 0; r.mean = 0;
  }
  PANTAS_HD void task() {
    if (fits) rec[recs].j1 = tasks;       // the item's own task ordinal; the write pass adds the item's task offset
    ++tasks;
  }
  PANTAS_HD void push(uint32_t w) {
    if (fits && words >= cap_w && !grow()) fits = false;
    if (fits) side[words] = w;
    ++words;
  }
  PANTAS_HD void end(uint32_t mean) {
    if (fits) {
      EventRec& r = rec[recs];
      r.side_len = words - r.side_off;
      r.mean = mean;
    }
    ++recs;
  }
};
PANTAS_HD bool rec_is_task(uint32_t type) { return type == T_IR_ANN || type == T_IR1_NOV || type == T_IR2_NOV; }

struct WriteEmit {
  EventRec* rec;
  uint32_t* side;
  uint32_t rec_i, word_i;
  uint32_t* task_list;   // record index of every walk task
  uint32_t task_i;
  PANTAS_HD void task() {
    task_list[task_i] = rec_i;
    rec[rec_i].j1 = task_i;
    ++task_i;
  }
  PANTAS_HD void begin(uint32_t type, uint32_t gene, uint32_t j0, uint32_t j1, uint32_t j2) {
    EventRec& r = rec[rec_i];
    r.type = type; r.gene = gene; r.j0 = j0; r.j1 = j1; r.j2 = j2; r.side_off = word_i; r.side_len = 0; r.mean = 0;
  }
  PANTAS_HD void push(uint32_t w) { side[word_i++] = w; }
  PANTAS_HD void end(uint32_t mean) {
    EventRec& r = rec[rec_i];
    r.side_len = word_i - r.side_off;
    r.mean = mean;
    ++rec_i;
  }
};

}  // namespace pantas
