// Augment engine: device state, kernels' host entry points, result assembly and the GFA
// writer.  Reference: scripts/alignments_augmentation_from_gaf.py (hot loop :135-371 on the
// GPU, second GFA pass :377-427 in write_augmented_gfa).
#pragma once
#include "common.h"
#include "gaf.h"
#include "graph.h"

namespace pantas {

struct HistEntry {      // one key of a node's IL or OL histogram that needs explicit ordering
  uint32_t node;
  uint32_t side;        // SIDE_IL / SIDE_OL
  int32_t pos;          // histogram key; for is_common the key is 0 (IL) / len(node) (OL)
  uint32_t count;       // increments (0 for is_common: the dense array holds the count)
  uint64_t first;       // first-seen order (walk.h event_order)
  bool is_common;
};

struct NovelLink {
  uint32_t u, v, count;
  uint64_t first;
};

struct AugmentResult {
  // (RawVector: gigabytes at whole-genome scale, overwritten by a device-to-host copy right after they are sized)
  RawVector<uint32_t> nc;          // [V]
  RawVector<uint32_t> rc;          // [A] per adjacency slot (first slot of a duplicated link gets the count)
  RawVector<uint32_t> il0, oll;    // [V] IL[0] and OL[len] counts
  std::vector<HistEntry> hist;     // sorted by (node, side, first)
  std::vector<NovelLink> novel;    // sorted by first
  uint64_t lines_total = 0, rejected = 0, skipped = 0, used = 0;
};

// Second GFA pass (:377-427): echoes the mapped GFA with NC/IL/OL on S lines, RC on L lines and
// the novel links appended.  Writes to fd.
void write_augmented_gfa(const Graph& g, const AugmentResult& r, int fd, unsigned threads = 0);

// sorts hist / novel into the orders documented above
void finalize_result(AugmentResult& r);

struct AugmentTimes {
  double tokenize_s = 0, h2d_s = 0, kernel_ms = 0, d2h_s = 0;
  double kernel_rare_ms = 0, kernel_main_ms = 0;
};

// Device side (augment.hip).  Opaque to the rest of the host code.
struct DeviceGraph;
struct DeviceBatch;

struct DeviceGraphDeleter { void operator()(DeviceGraph*) const; };
struct DeviceBatchDeleter { void operator()(DeviceBatch*) const; };
using DeviceGraphPtr = std::unique_ptr<DeviceGraph, DeviceGraphDeleter>;
using DeviceBatchPtr = std::unique_ptr<DeviceBatch, DeviceBatchDeleter>;

int device_count();                                   // 0 when no HIP device is usable
// Starts the HIP runtime and the context of `device` on a thread of its own (about a second on a cold process) so that it
// runs beside the parsing of the GFA instead of in front of the first upload.  Harmless without a device.
void set_tilde_table(DeviceGraph& d, const std::vector<TildeEntry>& tab);   // walk.h TildeCarry of the runs to come
void warm_up_device(int device);
void init_device(int device);     // the same on the calling thread
DeviceGraphPtr upload_graph(const Graph& g, int device);
void reset_counters(DeviceGraph& dg);                 // zero NC/RC/IL/OL, clear flags and tables
DeviceBatchPtr upload_batch(DeviceGraph& dg, const GafBatch& b);
// GPU tokeniser (tokenize.hip): text of whole lines -> bucketed records resident on dg's device.  `counts` receives the
// batch counters and a deferred tokeniser error, its arrays stay empty.  Needs a graph with dense (decimal) names.
struct DeviceTokenizeTimes { double h2d_s = 0, index_s = 0, measure_s = 0, write_s = 0, bucket_s = 0, total_s = 0; };
// text of whole lines already in HBM (zero-padded as the tokeniser's kernels need it): what a TextPrefetcher hands out
struct DeviceText { uint8_t* dev = nullptr; size_t size = 0; bool lone_cr = false; };   // lone_cr: see tokenize.hip upload_text
// `resident`: the same bytes as [data, data + size), already on the device (data is still read for error wording)
DeviceBatchPtr tokenize_gaf_device(DeviceGraph& dg, const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id,
                                   GafBatch& counts, DeviceTokenizeTimes* times = nullptr, const DeviceText* resident = nullptr);
// Copies the text of the NEXT batch to HBM (own stream, own pinned staging, a worker thread) while the current batch is
// tokenised and walked.  At most two ranges in flight: start, start, get, (use), start, get, ...; a buffer handed out
// by get() is overwritten by the second start() after it.
struct TextPrefetcher;
struct TextPrefetcherDeleter { void operator()(TextPrefetcher*) const; };
using TextPrefetcherPtr = std::unique_ptr<TextPrefetcher, TextPrefetcherDeleter>;
TextPrefetcherPtr make_text_prefetcher(DeviceGraph& dg, size_t max_bytes);
void prefetch_start(TextPrefetcher& p, const char* data, size_t size);
DeviceText prefetch_get(TextPrefetcher& p);
double prefetch_copy_seconds(const TextPrefetcher& p);
// copies a resident batch back (tests: the two tokenisers must produce the same arrays)
void download_batch(const DeviceGraph& dg, const DeviceBatch& db, GafBatch& out);
// Launches the two augment kernels over a resident batch; accumulates into dg's counters.
// Returns the device time of the two kernels in ms (HIP events on the launch stream).
float launch_augment(DeviceGraph& dg, DeviceBatch& db, float* ms_rare = nullptr, float* ms_main = nullptr);
// The two stages separately (multi-GPU: node flags are exchanged between them).
float launch_rare(DeviceGraph& dg, DeviceBatch& db);
float launch_main(DeviceGraph& dg, DeviceBatch& db);
// launch_augment without the wait (several in flight on the graph's stream); launch_collect waits for all of them and returns how many
// there were, their kernel times oldest first in out[0 .. min(n, capacity))
struct LaunchTimes { float rare_ms = 0, main_ms = 0, main_short_ms = 0; };
void launch_augment_async(DeviceGraph& dg, DeviceBatch& db);
size_t launch_collect(DeviceGraph& dg, LaunchTimes* out, size_t capacity);
void export_flags(DeviceGraph& dg, std::vector<uint32_t>& codes);          // (node << 1 | side), sorted
void import_flags(DeviceGraph& dg, const uint32_t* codes, size_t n);
// Raises Error if a kernel reported a line the reference would have died on (or one this engine refuses).
// first_record_line: line id of the run's first record in file order (walk.h WALK_TILDE_FIRST: only there the reference
// raises UnboundLocalError; on a later line it reuses a length left over from the line before, which is refused here).
void check_device_status(DeviceGraph& dg, uint64_t first_record_line);
void boost_novel_table(DeviceGraph& dg);              // next sizing of the novel-link table is 4x larger
void boost_hist_table(DeviceGraph& dg);               // the same for the IL/OL key table (its bound should never be exceeded)
uint64_t device_free_bytes(DeviceGraph& dg);          // HBM not allocated right now on dg's device
// `b` holds the same records as a batch whose rare-key kernel already ran this run (a streamed run tokenises a dropped
// batch again for its main pass): it may be launched with launch_main
void adopt_rare_done(DeviceGraph& dg, DeviceBatch& b);
void reserve_hist(DeviceGraph& dg, uint64_t keys);   // bound on distinct IL/OL keys of the whole (sharded) run
void download_result(DeviceGraph& dg, const Graph& g, AugmentResult& r);
// device pointers of the dense counters, for collectives run by the caller (RCCL via torch.distributed)
// base[0 .. words): {NC, IL0, OLL, RC of the first out-slot} per node (4 V words), then RC per adjacency slot (A words).
// Any elementwise sum over ranks is a valid merge; the planar arrays only exist on the host (download_result).
struct CounterPtrs { void* base; uint64_t words; uint64_t V, A; };
CounterPtrs counter_ptrs(DeviceGraph& dg);
// Round 5: the kernels can keep the bitmap of touched counter blocks themselves (one bit per block_words counters, set by every
// site that adds to a counter), so that the exchange does not read the whole counter array to find them.  Asked for by a caller
// that will exchange (read-sharded runs); kept from the next zeroing of the counters on.  bits == nullptr: not kept (this run).
void set_track_touched(DeviceGraph& dg, bool on);
struct TouchedBitmap { const uint32_t* bits; uint64_t words; uint32_t block_words; };
TouchedBitmap touched_bitmap_of(DeviceGraph& dg);
// Block-sparse exchange of the counters between ranks (exchange.hip): bitmap of touched blocks -> (caller all-gathers)
// -> pack the union's blocks into a staging buffer -> (caller all-reduces it) -> unpack.  The allocation behind
// counter_ptrs is zero-padded to a multiple of COUNTER_PAD_WORDS so that whole blocks can be read.
constexpr uint64_t COUNTER_PAD_WORDS = 2048;
struct CounterExchange;
struct CounterExchangeDeleter { void operator()(CounterExchange*) const; };
using CounterExchangePtr = std::unique_ptr<CounterExchange, CounterExchangeDeleter>;
void exchange_bitmap(DeviceGraph& dg, CounterExchangePtr& x, uint32_t block_words, void** dev_bitmap, uint64_t* bitmap_words);
void exchange_pack(DeviceGraph& dg, CounterExchange& x, const void* dev_bitmaps, uint32_t n_bitmaps, void** dev_staging, uint64_t* staging_words);
void exchange_unpack(DeviceGraph& dg, CounterExchange& x);
// for pantas_counters_allreduce (capi.cpp is host-only code: streams and device memory go through these)
void exchange_drop(CounterExchange& x);                        // forget bitmap / staging of an exchange that is not finished
void* exchange_stream(DeviceGraph& dg);                        // the graph's stream (a hipStream_t)
void exchange_stream_wait(DeviceGraph& dg, void* stream);      // hipStreamSynchronize
void* exchange_scratch_take(DeviceGraph& dg, size_t bytes);    // device memory from the graph's scratch pool ...
void exchange_scratch_give(DeviceGraph& dg, void* p);          // ... and back
uint64_t batch_lines(const DeviceBatch& db);
uint64_t batch_short_records(const DeviceBatch& db);   // records of the short-walk class (walk.h), first in the batch
float last_short_kernel_ms(const DeviceGraph& dg);      // short-walk kernel of the last launch_main alone
void batch_sizes(const DeviceBatch& db, uint64_t* path_nodes, uint64_t* cs_ops);
uint64_t batch_alg_bytes(const DeviceBatch& db);
uint64_t batch_alg_bytes_short(const DeviceBatch& db);   // the same for the short-walk class alone      // algorithmic bytes of the main kernel

}  // namespace pantas
