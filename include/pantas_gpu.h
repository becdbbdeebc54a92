/* libpantas_gpu -- C ABI of the MI355X augment+call engine.
 *
 * The reference (AlgoLab/pantas) has no in-process plugin interface; its hot path is reached
 * through two command lines (`pantas augment`, `pantas call`; reference `pantas:114-175`) that
 * spawn `scripts/alignments_augmentation_from_gaf.py` and `scripts/call.py`.  This header is
 * the boundary a maintainer would bind (ctypes) to replace the bodies of
 * those two scripts.  Every entry point names the reference code it stands in for.
 *
 * Conventions: plain C types only; every function that can fail returns 0 on success and a
 * negative status otherwise, with the reason available from pantas_last_error() (thread
 * local, library-owned).  PANTAS_E_REFERENCE means "the reference script raises a Python
 * exception on this input" (pantas_last_error_kind() names the exception class); nothing
 * has then been written to the output descriptor, as in the reference.  One host thread
 * drives one handle.  The kernels have NO CPU fallback: without a HIP device every
 * device entry point fails with PANTAS_E_NODEVICE.
 */
#ifndef PANTAS_GPU_H
#define PANTAS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANTAS_OK 0
#define PANTAS_E_REFERENCE (-1) /* the reference raises on this input */
#define PANTAS_E_UNSUPPORTED (-2)
#define PANTAS_E_NODEVICE (-3)
#define PANTAS_E_HIP (-4)
#define PANTAS_E_IO (-5)
#define PANTAS_E_INTERNAL (-6)
#define PANTAS_E_ARG (-7)

typedef struct pantas_graph pantas_graph; /* GFA as CSR, host + HBM copy, + run results */
typedef struct pantas_batch pantas_batch; /* tokenised GAF records resident in HBM */

const char* pantas_last_error(void);
const char* pantas_last_error_kind(void);
const char* pantas_version(void);
int pantas_device_count(void);
/* optional: starts the HIP runtime and the context of `device` on a thread of its own and returns at once -- the second it takes
 * on a cold process then runs beside the parsing of the GFA (the command lines call it first thing) */
void pantas_device_warmup(int device);
/* the same start-up on the calling thread, returning when it is done.  For a caller whose next seconds are a graph parse on every
 * core: the parse's page faults hold the address-space lock the runtime's start-up needs, and the background thread was measured to
 * take longer than the parse (0.9 s left over when the graph went to the device) where the same calls take 0.25 s up front. */
void pantas_device_init(int device);

/* Structs cross this boundary by pointer and have grown between releases.  Every struct below therefore starts with
 * struct_size, which the CALLER sets to the sizeof() of the struct it was compiled (or, for a ctypes binding, declared)
 * with before every call: the library reads / writes that many bytes and never more, so an older, shorter declaration
 * keeps working and a forgotten one is refused (PANTAS_E_ARG) instead of overrunning the caller's buffer.  Fields are only
 * ever appended.  PANTAS_ABI_VERSION counts such appends; pantas_abi_version() is what the loaded library was built with. */
#define PANTAS_ABI_VERSION 8
int pantas_abi_version(void);

/* ---- graph: replaces the GFA readers of both scripts and scripts/gfautils.py:92-128 ------ */
typedef struct {
  uint32_t struct_size, reserved0_;
  uint64_t segments;       /* distinct S names (V) */
  uint64_t link_slots;     /* L lines with both ends known (A) */
  uint64_t lines, s_lines, l_lines, p_lines;
  uint64_t text_bytes;
  uint32_t max_segment_len;
  uint32_t dense_names;    /* 1: names are canonical decimals in a compact range */
  double load_seconds;
} pantas_graph_stats;

/* alignments_augmentation_from_gaf.py:121-126 (first GFA pass) / call.py:145-181 */
int pantas_graph_load(const char* gfa_path, pantas_graph** out);
/* The same graph for a caller that goes on to pantas_call_prepare (ABI 4): when the text is parsed on the GPU (after
 * pantas_device_init), the tags call.py reads off the S / L lines (:25-32, :145-181: NC, RC, EX, JN, ID) are taken in the same
 * visit of the text, and pantas_call_prepare does not walk the lines again.  Results are those of pantas_graph_load. */
int pantas_graph_load_for_call(const char* gfa_path, pantas_graph** out);
/* Binary cache of the parsed graph (the graph is shared by all replicates).  Loads `cache_path` when
 * it was made from exactly this GFA file (size, mtime, hash of both ends); otherwise parses the GFA and, if
 * write_if_missing, writes the cache.  *used_cache (may be NULL) = 1 when the text was not parsed.  Graphs whose
 * segment names are not canonical decimals are parsed every time. */
int pantas_graph_load_cached(const char* gfa_path, const char* cache_path, int write_if_missing, pantas_graph** out,
                             int* used_cache);
void pantas_graph_free(pantas_graph* g);
int pantas_graph_get_stats(const pantas_graph* g, pantas_graph_stats* out);
/* copies node_len / row_ptr / col to the HBM of `device` and allocates the counters */
int pantas_graph_to_device(pantas_graph* g, int device);
/* Gives the handle's device memory back (graph arrays, resident batches, call tables, what the GPU loader left parked) while the
 * host side stays: a finished augment run can still be written (pantas_augment_write_gfa), pantas_graph_to_device brings the
 * graph back.  For a process that has an output file of tens of GB to write after its last kernel: memory freed at exit is
 * scrubbed by the driver while the NEXT process waits for its first large hipMalloc (2 - 3 s after a whole-genome run); freed
 * here, that happens beside the writing.  (ABI 5; the reference has no counterpart.) */
int pantas_graph_release_device(pantas_graph* g);

/* ---- augment: alignments_augmentation_from_gaf.py main() :110-427 ------------------------- */
typedef struct {
  uint32_t struct_size, reserved0_;
  uint64_t lines;          /* GAF lines read */
  uint64_t rejected;       /* mapq < thr            (:143-146, "Rejected alignments: N") */
  uint64_t skipped;        /* '*' path or dv > 0.1  (:147-148, :179-180) */
  uint64_t records;        /* lines that reach the walk */
  uint64_t path_nodes;     /* sum of path lengths of those */
  uint64_t cs_ops;
  uint64_t del_records;    /* records with a '-' op */
  uint64_t del_path_nodes; /* sum of their path lengths (4x this bounds the distinct IL/OL keys they create) */
  uint64_t text_bytes;
  uint64_t alg_bytes;      /* algorithmic bytes of one launch over this batch */
  double tokenize_seconds, upload_seconds;
  /* which tokeniser made the batch (PANTAS_TOKENIZER_HOST / _DEVICE) and, for the device one, its phases:
   * upload_seconds = text -> HBM, then line index, measure pass + scans, write pass, locus bucketing */
  int32_t tokenizer, reserved_;
  double tok_index_seconds, tok_measure_seconds, tok_write_seconds, tok_bucket_seconds;
  /* the short-walk class (records the dominant kernel takes: cs of ':' and '*' only, at most 31 path nodes) */
  uint64_t short_records, short_alg_bytes;
} pantas_batch_stats;

typedef struct {
  uint32_t struct_size, reserved0_;
  float rare_ms;           /* augment_rare_kernel, HIP events on the launch stream */
  float main_ms;           /* all main kernels of the launch */
  float main_short_ms;     /* of which the short-walk kernel (the dominant one); 0 when one kernel took every record */
  float reserved_;
} pantas_launch_stats;

/* Where the GAF text is turned into records (:142-197, :10-50).  AUTO = on the GPU when the graph's segment names are
 * canonical decimals (what vg writes), else on the host; the environment variable PANTAS_TOKENIZER=host|device
 * overrides AUTO.  DEVICE on a graph with other names is PANTAS_E_UNSUPPORTED at load time. */
#define PANTAS_TOKENIZER_AUTO 0
#define PANTAS_TOKENIZER_HOST 1
#define PANTAS_TOKENIZER_DEVICE 2
int pantas_augment_tokenizer(pantas_graph* g, int mode);
/* zero NC/RC/IL/OL and the sparse tables (start of a run) */
int pantas_augment_reset(pantas_graph* g);
/* :142-197 + :10-50: tokenise GAF bytes [begin,end) of the file (end = 0: to EOF; ranges must
 * fall on line starts, see pantas_gaf_split) into binary records and upload them.
 * first_line = index of the range's first line in the whole file (first-seen ordering).
 * A line the reference raises on while splitting it ends the batch; the error is reported by
 * pantas_augment_finish AFTER the lines before it have been walked, because the reference handles the file
 * line by line and an earlier line that only fails in the walk is the one it dies on. */
int pantas_augment_load_gaf(pantas_graph* g, const char* gaf_path, int thr, uint64_t begin, uint64_t end,
                            uint64_t first_line, pantas_batch** out, pantas_batch_stats* stats);
/* byte range of shard `rank` of `world` (cut at line starts) and the number of lines before it */
int pantas_gaf_split(const char* gaf_path, int rank, int world, uint64_t* begin, uint64_t* end,
                     uint64_t* lines_before);
void pantas_batch_free(pantas_batch* b);
/* copies a resident batch back to the host (inspection / tests).  Sizes from its stats: hdr 4 int32 per record
 * {start, tail, line_id, flags}, node_off and op_off records + 1, nodes path_nodes, ops cs_ops words. */
int pantas_batch_download(pantas_graph* g, pantas_batch* b, int32_t* hdr, uint32_t* node_off, uint32_t* op_off,
                          uint32_t* nodes, uint32_t* ops);
/* :209-363 on the GPU: both kernels over a resident batch, accumulating into g's counters.
 * A run over SEVERAL batches must use the staged form below: launch_rare for every batch, then launch_main
 * for every batch (a late rare-key kernel cannot flag nodes for batches already walked; PANTAS_E_ARG).
 * Relaunching a batch that was already launched in this run is allowed (benchmarks). */
int pantas_augment_launch(pantas_graph* g, pantas_batch* b, pantas_launch_stats* out);
/* ABI 7.  The same two kernels enqueued on the graph's stream and NOT waited for, so that a caller who launches batch after batch
 * (or one resident batch again and again) keeps the GPU fed: pantas_augment_launch pays a host round trip -- launch, wake-up, event
 * reads, about 50 microseconds -- per call.  At most 64 launches may be in flight (PANTAS_E_ARG beyond).  launch_collect waits for
 * everything enqueued and returns the number of launches in *n, their kernel times oldest first as three floats per launch
 * {rare_ms, main_ms, main_short_ms} (what pantas_launch_stats holds) in ms[0 .. 3 * min(*n, capacity)).  The staged rule above holds
 * as for pantas_augment_launch; pantas_augment_finish and every download wait for the stream as before. */
int pantas_augment_launch_async(pantas_graph* g, pantas_batch* b);
int pantas_augment_launch_collect(pantas_graph* g, float* ms, uint64_t capacity, uint64_t* n);
/* the two kernels separately, for runs sharded over several GPUs: after the rare-key kernel every
 * rank exports the node flags it set ((node << 1) | side, uint32), the ranks all-gather them, import
 * the union, and only then run the main kernel (it records the first-seen order of the common IL/OL
 * key on flagged nodes).  pantas_augment_launch == launch_rare + launch_main. */
/* sharded runs, after reset and before the first launch: room for `keys` distinct IL/OL keys (the sum
 * over ranks of 4 * del_path_nodes), since every rank also records keys for nodes other ranks flagged */
int pantas_augment_reserve(pantas_graph* g, uint64_t keys);
int pantas_augment_launch_rare(pantas_graph* g, pantas_batch* b, float* ms);
int pantas_augment_launch_main(pantas_graph* g, pantas_batch* b, float* ms);
int pantas_augment_flags_export(pantas_graph* g, uint32_t* codes, uint64_t capacity, uint64_t* n);
int pantas_augment_flags_import(pantas_graph* g, const uint32_t* codes, uint64_t n);
/* device address of the dense counters (uint32) for a caller-run all-reduce (RCCL through torch.distributed):
 * {NC, IL0, OLL, RC of the first out-slot} per node, then RC per adjacency slot; words = 4V + A.  The layout is
 * the library's own (one 16-byte slot per node keeps a node's counters in one HBM line); any elementwise sum over
 * ranks merges it.  Host-side planar copies: pantas_augment_read after pantas_augment_finish. */
int pantas_augment_counters(pantas_graph* g, void** dev_ptr, uint64_t* words);
/* Block-sparse form of that all-reduce (reads cover exons only, so nearly all counters are zero and a
 * ring over point-to-point xGMI links should not carry them).  All pointers are device addresses.
 *   1. pantas_counters_bitmap: one bit per block of block_words (8, 16, 32 or 64) counters that holds a non-zero word;
 *      the caller all-gathers the bitmaps of all ranks (uint32 words, rank-major);
 *   2. pantas_counters_pack: ORs the n_bitmaps gathered bitmaps and copies the union's blocks into a staging buffer
 *      (same layout on every rank); the caller all-reduces staging_words uint32 sums (staging_words may be 0);
 *   3. pantas_counters_unpack: staging -> counters.  Skipping 3 leaves the counters as they were. */
int pantas_counters_bitmap(pantas_graph* g, uint32_t block_words, void** dev_bitmap, uint64_t* bitmap_words);
/* ABI 8.  on != 0: from the next pantas_augment_reset on, the augment kernels keep the bitmap of step 1 themselves (every site that
 * adds to a counter sets its block's bit), and pantas_counters_bitmap / pantas_counters_allreduce only fold it to block_words instead
 * of reading all 4V + A counters (5 GB at whole-genome scale).  For callers that will exchange: a run on one GPU leaves it off (the
 * kernels then carry one never-taken branch per flushed counter).  With PANTAS_EXCHANGE_VERIFY=1 in the environment step 1 also
 * reads the counters and fails (PANTAS_E_INTERNAL) when a block that holds a count is missing from the kept bitmap. */
int pantas_augment_track_touched(pantas_graph* g, int on);
int pantas_counters_pack(pantas_graph* g, const void* dev_bitmaps, uint32_t n_bitmaps, void** dev_staging, uint64_t* staging_words);
int pantas_counters_unpack(pantas_graph* g);
/* THE data-path collective of a read-sharded replicate behind the C ABI (round 3; north_star: "a single RCCL all-reduce over the
 * node/edge coverage arrays when reads are split"; the reference has no counterpart -- one process per replicate,
 * exps/3-homo-real/workflow/rules/pantas.smk:161): steps 1 - 3 above with the two collectives between them run on the caller's
 * RCCL communicator -- bitmap, all-gather of the bitmaps, pack, all-reduce (uint32 sums) of the packed blocks, unpack; the dense
 * all-reduce of all 4V + A counters instead when more than half of the blocks are touched or block_words is 0.  In place, on
 * every rank; returns when the counters hold the sums.
 *   comm:   an ncclComm_t of the ranks that hold the shards (this process's rank of it runs on the graph's device);
 *   stream: a hipStream_t for the collectives, NULL = the graph's own stream;
 *   api:    the RCCL entry points to call.  A process can hold two RCCLs (PyTorch ships its own beside /opt/rocm/lib): the
 *           functions must be those of the library that MADE `comm`.  NULL: looked up in the process (dlsym RTLD_DEFAULT),
 *           then in librccl.so.1 / librccl.so.  The signatures are rccl.h's with void* for ncclComm_t / hipStream_t. */
typedef struct {
  uint32_t struct_size, reserved0_;
  int (*all_reduce)(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, void* stream);   /* ncclAllReduce */
  int (*all_gather)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, void* stream);       /* ncclAllGather */
  int (*comm_count)(const void* comm, int* count);                                                                         /* ncclCommCount */
} pantas_rccl_api;
typedef struct {
  uint32_t struct_size, reserved0_;
  uint32_t mode;              /* 0 dense, 1 block-sparse */
  uint32_t world;             /* ranks of the communicator */
  uint32_t block_words;
  uint32_t reserved1_;
  uint64_t counter_bytes;     /* 4 (4V + A) */
  uint64_t exchanged_bytes;   /* what this rank handed to the collectives: gathered bitmaps + packed blocks, or counter_bytes */
  float bitmap_ms, pack_ms, allreduce_ms, unpack_ms, total_ms;
  float reserved2_;
} pantas_allreduce_stats;
int pantas_counters_allreduce(pantas_graph* g, const pantas_rccl_api* api, void* comm, void* stream, uint32_t block_words,
                              pantas_allreduce_stats* out);
/* sparse side results (links absent from the GFA, rare IL/OL keys): export from one rank /
 * merge into another.  Rows are 4 x uint64: {key, count, first_seen, kind (0 hist, 1 novel)} */
int pantas_augment_sparse_count(pantas_graph* g, uint64_t* rows);
int pantas_augment_sparse_export(pantas_graph* g, uint64_t* rows_out, uint64_t capacity);
int pantas_augment_sparse_import(pantas_graph* g, const uint64_t* rows, uint64_t n);
/* reports lines the kernels flagged (reference IndexError / '~'), downloads the results */
int pantas_augment_finish(pantas_graph* g);
/* host copies of the dense counters after finish (for tests): any pointer may be NULL */
int pantas_augment_read(pantas_graph* g, uint32_t* nc, uint32_t* il0, uint32_t* oll, uint32_t* rc);
/* :377-427 second GFA pass: echo + NC/IL/OL/RC tags + novel links, to fd */
int pantas_augment_write_gfa(pantas_graph* g, int fd);
/* :135-371 over a GAF of ANY size (the reference reads line by line, :135-138): the byte range [begin, end) (end = 0: to
 * EOF; cut at line starts as pantas_gaf_split does) is cut into batches of at most PANTAS_BATCH_BYTES of text (default
 * 2 GiB).  stream_rare tokenises batch after batch (the next batch's text is copied to HBM meanwhile) and runs each one's
 * rare-key kernel; the records stay in HBM while they fit PANTAS_RESIDENT_BYTES (default 60 % of the free HBM), later
 * batches are dropped and tokenised again by stream_main, which runs every batch's main kernel.  Between the two a
 * sharded run exchanges node flags exactly as with launch_rare / launch_main.  One streamed range per run
 * (pantas_augment_reset starts the next); stats (may be NULL) = totals over the batches, *_ms = summed kernel times. */
int pantas_augment_stream_rare(pantas_graph* g, const char* gaf_path, int thr, uint64_t begin, uint64_t end,
                               uint64_t first_line, pantas_batch_stats* stats, float* rare_ms);
int pantas_augment_stream_main(pantas_graph* g, float* main_ms);
/* batches of the streamed run, how many stayed resident, how many were tokenised twice (any pointer may be NULL) */
int pantas_augment_stream_info(const pantas_graph* g, uint64_t* batches, uint64_t* resident, uint64_t* retokenised);
/* whole script: reset, stream_rare + stream_main over the whole file, finish, write (fd_out < 0: no write).
 * stats and launch may be NULL */
int pantas_augment_run(pantas_graph* g, const char* gaf_path, int thr, int fd_out, pantas_batch_stats* stats,
                       pantas_launch_stats* launch);

/* ---- quant: scripts/quantify.py main() :233-406 (host code, no device needed) ----
 * PSI / dPSI of the events of n replicates of condition 1 against n replicates of condition 2 (`pantas call` CSVs), written as CSV
 * to fd_out: the bytes the reference prints (row order, float formatting).  w: -w, both: --both.  Errors carry the class the
 * reference raises (TypeError, ValueError, AttributeError) in pantas_last_error_kind(). */
int pantas_quant_run(const char* const* c1_paths, const char* const* c2_paths, uint32_t n_replicates, int32_t w, int32_t both, int fd_out);

/* ---- remap: scripts/remap.py main() :29-222 (host code, no device needed; ABI 6) ----
 * Reference-genome coordinates for the junctions of a `pantas quant` CSV, written as CSV to fd_out: the bytes the reference prints.
 * min_intron_size: -i (default 100).  Errors carry the class the reference raises (ValueError, IndexError, AttributeError,
 * AssertionError) in pantas_last_error_kind(); nothing is written then. */
int pantas_remap_run(const char* csv_path, const char* gtf_path, int64_t min_intron_size, int fd_out);

/* ---- call: scripts/call.py main() :132-1105 ------------------------------------------------ */
typedef struct {
  uint32_t struct_size, reserved0_;
  int32_t rca;            /* --rca (default -1) */
  int32_t isw;            /* --isw (default 5) */
  int32_t minintronsize;  /* --minintronsize (default 100) */
  uint32_t events;        /* --events: bit 0 ES, bit 1 SS, bit 2 IR (default 7) */
  int32_t novel;          /* --novel */
  int32_t no_annotated;   /* --no-annotated */
  const char* junction;   /* --junction "u-v" or NULL */
  uint32_t shard_rank;    /* multi-GPU: evaluate every shard_world-th junction / novel link, starting at */
  uint32_t shard_world;   /* shard_rank (0 or 1 = everything); rows of all shards together = the full CSV */
  int32_t debug;          /* --debug: the reference's "Checking ..." lines (call.py:254,317,...,1002) to fd_err */
  int32_t release_device; /* nonzero (ABI 5): the handle's device memory is given back as soon as the kernels' results are on the host, before the
                           * rows are rendered and written -- what pantas_graph_release_device does, in the one place a caller cannot reach; the
                           * handle then needs pantas_call_prepare before another pantas_call_run */
} pantas_call_opts;

typedef struct {
  uint32_t struct_size, reserved0_;
  uint64_t junctions;       /* annotated junction links (JN) */
  uint64_t novel_links;     /* links with ID:Z:N */
  uint64_t evaluated;       /* junction evaluations launched over the three passes */
  uint64_t events;          /* CSV rows written */
  uint64_t ex_entries, jn_entries;
  double parse_seconds;     /* tag + GTF parsing and interning (host) */
  double upload_seconds;
  double render_seconds;
  float annotated_ms, novel_ms, around_ms;   /* kernels of each pass, HIP events; the passes run side by side on streams of their own (PANTAS_CALL_SERIAL=1: one after
                                              * the other), so these overlap -- around_ms = the pass around the novel links + the cassette items */
  float passes_ms;                           /* (ABI 8) all passes on the device: first launch to the last pass's end */
} pantas_call_stats;

/* :145-216: parse the tags of the (augmented) GFA behind `g` and the GTF, intern them, copy the
 * tables to the HBM of `device` */
int pantas_call_prepare(pantas_graph* g, const char* gtf_path, int device);
/* :218-1105: header + event rows to fd_csv, "Skipping ..." lines to fd_err (-1: drop) */
int pantas_call_run(pantas_graph* g, const pantas_call_opts* opts, int fd_csv, int fd_err, pantas_call_stats* stats);

/* ---- synthetic workloads (SURVEY.md 8-d recipe) for benchmarks and scale tests ------------------
 * Generates annotated GFA + GTF + GAF TEXT in memory; the *_mem entry points run the product's
 * real parsers over such buffers (or any caller-owned text), which must outlive the handles
 * built from them. */
typedef struct pantas_synth pantas_synth;
typedef struct {
  uint32_t struct_size, reserved0_;
  uint64_t seed;
  uint32_t n_genes;
  uint32_t n_chrom;
  uint64_t n_lines;
  uint64_t base_id;        /* first node id; ids must keep one decimal width */
  double bubble_rate;
  uint32_t max_exons;
  uint32_t intron_long_max;
  uint32_t intergenic_min, intergenic_max;
  uint32_t read_len;
  int32_t with_path;
  double zipf_a;
  double p_lowmapq, p_unmapped, p_sub, p_indel, p_highdv;
  double p_novel_skip, p_novel_ce, p_intron, p_novel_ss, p_clip;
  uint64_t line_offset;    /* lines [line_offset, line_offset + n_lines) of the read stream */
  int32_t gaf_only;        /* 1: no GFA / GTF text (further blocks of the read stream of a graph generated before) */
  uint32_t hot_sites;      /* with p_del_hot > 0: that share of the deletion reads is aimed at hot_sites fixed */
  double p_del_hot;        /* (node, position) sites on the most expressed genes -- the skew of real indel variants */
  double p_two_indels;     /* ABI 8: share of the mapped reads whose cs holds TWO indels (an insertion and a deletion) -- reads outside
                              the short-walk class of the augment kernels, which the plain recipe never writes; 0 leaves every
                              other draw of the stream as it was */
} pantas_synth_params;
typedef struct {
  uint32_t struct_size, reserved0_;
  uint64_t nodes, links, junction_links, transcripts, gaf_lines;
  uint64_t gfa_bytes, gtf_bytes, gaf_bytes;
  double seconds;
} pantas_synth_stats;
void pantas_synth_defaults(pantas_synth_params* p);
int pantas_synth_generate(const pantas_synth_params* p, pantas_synth** out, pantas_synth_stats* stats);
/* which: 0 GFA, 1 GTF, 2 GAF */
int pantas_synth_text(const pantas_synth* s, int which, const char** data, uint64_t* size);
/* writes <prefix>.gfa/.gtf and the first max_gaf_lines lines (0 = all) of the GAF to <prefix>.gaf */
int pantas_synth_write(const pantas_synth* s, const char* prefix, uint64_t max_gaf_lines);
void pantas_synth_free(pantas_synth* s);
int pantas_graph_load_mem(const char* gfa_text, uint64_t size, pantas_graph** out);
int pantas_augment_load_gaf_mem(pantas_graph* g, const char* gaf_text, uint64_t size, int thr, uint64_t first_line,
                                pantas_batch** out, pantas_batch_stats* stats);
int pantas_call_prepare_mem(pantas_graph* g, const char* gtf_text, uint64_t size, int device);
/* augment -> call without the GFA text round trip: like pantas_call_prepare on
 * the GFA this handle's last finished augment run WOULD write -- NC / RC from its counters, its
 * novel links appended after the GFA's own links.  gtf_text == NULL: read gtf_path instead. */
int pantas_call_prepare_fused(pantas_graph* g, const char* gtf_path, const char* gtf_text, uint64_t size, int device);
#ifdef __cplusplus
}
#endif
#endif
