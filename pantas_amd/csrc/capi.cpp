// extern "C" boundary of libpantas_gpu (include/pantas_gpu.h).  No C++ exception leaves it.
#include "../../include/pantas_gpu.h"

#include "augment.h"
#include "call.h"
#include "synth.h"
#include "quant.h"
#include "remap.h"

#include <atomic>
#include <dlfcn.h>
#include <fcntl.h>

#include <unistd.h>

#include <string>

using namespace pantas;

struct pantas_graph {
  std::unique_ptr<Graph> g;
  DeviceGraphPtr dev;
  CounterExchangePtr xchg;           // block-sparse counter exchange in flight (its buffers belong to dev's scratch pool)
  AugmentResult res;
  bool have_result = false;
  uint64_t lines = 0, rejected = 0, skipped = 0;
  uint64_t first_record_line = UINT64_MAX;   // of the run, over its batches (walk.h WALK_TILDE_FIRST)
  uint64_t run_first_line = UINT64_MAX;      // smallest `first_line` a batch of the run was loaded with: 0 = the run holds the START of its file
  std::shared_ptr<Error> deferred;   // tokeniser error of a loaded batch, reported by finish (see gaf.h)
  int tokenizer = PANTAS_TOKENIZER_AUTO;
  double load_s = 0;
  std::unique_ptr<struct StreamState> stream;   // a run over a GAF byte range in several batches (pantas_augment_stream_*)
  std::vector<TildeEntry> tilde_tab;   // walk.h TildeCarry of the file pantas_augment_run is on (computed when a run stops at such a line)
  bool tilde_tried = false;
  // call
  std::unique_ptr<CallTables> call_tables;
  DeviceCallPtr call_dev;
  double call_upload_s = 0;
};

struct pantas_synth {
  SynthOutput out;
};

struct pantas_batch {
  DeviceBatchPtr dev;
  pantas_batch_stats st{};
};

// One byte range of the GAF handled as a batch.  After the rare-key pass the batch either stays in HBM (resident) or, when
// the residency budget is spent, is dropped and tokenised again for the main pass.
struct StreamBatch {
  uint64_t begin = 0, end = 0, first_line = 0;
  std::unique_ptr<pantas_batch> resident;
  uint64_t device_bytes = 0;
};
struct StreamState {
  RawVector<char> text_copy;          // universal newlines applied (stream_rare_pass): `file` then points here
  MappedFile file;
  int thr = 20;
  std::vector<StreamBatch> batches;
  uint64_t max_batch_bytes = 0;
  uint64_t resident_bytes = 0, resident_budget = 0;
  uint64_t retokenised = 0, kept_resident = 0;
  bool main_done = false;
  pantas_batch_stats total{};
};

namespace {

thread_local std::string g_err, g_kind;

int fail(const Error& e) {
  g_err = e.what();
  g_kind = e.kind;
  if (e.kind == "Unsupported") return PANTAS_E_UNSUPPORTED;
  if (e.kind == "NoDevice") return PANTAS_E_NODEVICE;
  if (e.kind == "HipError") return PANTAS_E_HIP;
  if (e.kind == "OSError" || e.kind == "FileNotFoundError") return PANTAS_E_IO;
  if (e.kind == "InternalError" || e.kind == "TableOverflow") return PANTAS_E_INTERNAL;
  if (e.kind == "ProtocolError") return PANTAS_E_ARG;
  if (e.kind == "ArgumentError") return PANTAS_E_ARG;
  return PANTAS_E_REFERENCE;
}

template <class F>
int guarded(F&& f) {
  try {
    f();
    return PANTAS_OK;
  } catch (const Error& e) {
    return fail(e);
  } catch (const std::bad_alloc&) {
    return fail(Error("InternalError", "out of host memory"));
  } catch (const std::exception& e) {
    return fail(Error("InternalError", e.what()));
  }
}

void need(bool ok, const char* what) {
  if (!ok) throw Error("ArgumentError", what);
}

void need_device(pantas_graph* g) {
  need(g && g->g, "null graph handle");
  if (!g->dev) throw Error("NoDevice", "graph is not on a device: call pantas_graph_to_device first");
}

// Structs of the ABI start with the size their owner declared (include/pantas_gpu.h): copy out / in that many bytes.
template <class T>
uint32_t declared_size(const T* p, const char* what) {
  const uint32_t n = p->struct_size;
  if (n < 8 || n > 4096 || (n & 3u)) throw Error("ArgumentError", std::string(what) + ".struct_size is not set (set it to sizeof(" + what + ") before the call)");
  return n;
}
template <class T>
void put_out(T* dst, T src, const char* what) {
  if (!dst) return;
  const uint32_t n = declared_size(dst, what);
  src.struct_size = n;
  src.reserved0_ = 0;
  memcpy(dst, &src, std::min<size_t>(n, sizeof(T)));
}
template <class T>
T get_in(const T* src, const char* what) {
  T v;
  memset(&v, 0, sizeof v);
  memcpy(&v, src, std::min<size_t>(declared_size(src, what), sizeof(T)));
  return v;
}

constexpr uint64_t KIND_HIST = 0, KIND_NOVEL = 1;
constexpr uint32_t COMMON_CODE = 0x7FFFFFFFu;

}  // namespace

static void write_fd(int fd, const std::string& s) {
  const char* p = s.data();
  size_t n = s.size();
  while (n) {
    ssize_t w = ::write(fd, p, n);
    if (w < 0) { if (errno == EINTR) continue; throw Error("OSError", "write failed"); }
    p += w;
    n -= size_t(w);
  }
}

extern "C" {

__attribute__((visibility("default"))) const char* pantas_last_error(void) { return g_err.c_str(); }
__attribute__((visibility("default"))) const char* pantas_last_error_kind(void) { return g_kind.c_str(); }
__attribute__((visibility("default"))) const char* pantas_version(void) { return "pantas-mi355x 0.1 (gfx950)"; }
__attribute__((visibility("default"))) int pantas_device_count(void) { return device_count(); }
__attribute__((visibility("default"))) int pantas_abi_version(void) { return PANTAS_ABI_VERSION; }

__attribute__((visibility("default"))) int pantas_quant_run(const char* const* c1_paths, const char* const* c2_paths, uint32_t n_replicates, int32_t w, int32_t both,
                                                            int fd_out) {
  return guarded([&] {
    need(c1_paths && c2_paths && n_replicates >= 1, "bad argument");
    std::vector<std::string> a, b;
    for (uint32_t i = 0; i < n_replicates; ++i) {
      need(c1_paths[i] && c2_paths[i], "null path");
      a.emplace_back(c1_paths[i]);
      b.emplace_back(c2_paths[i]);
    }
    quant_run(a, b, w, both != 0, fd_out);
  });
}
__attribute__((visibility("default"))) int pantas_remap_run(const char* csv_path, const char* gtf_path, int64_t min_intron_size, int fd_out) {
  return guarded([&] {
    need(csv_path && gtf_path, "null argument");
    remap_run(csv_path, gtf_path, min_intron_size, fd_out);
  });
}
__attribute__((visibility("default"))) void pantas_device_warmup(int device) { try { warm_up_device(device); } catch (...) {} }
// The call passes run on four streams beside the graph's own (call.hip run_call).  The HIP runtime spreads a process's streams over
// four hardware queues unless told otherwise, and two lanes that land on one queue run in turn (S4: 31 ms for the passes against
// 24 with eight queues).  The variable is read when the runtime starts, i.e. at the first HIP call: set here, when this library is
// loaded, unless the user has chosen a value.
__attribute__((constructor)) static void ask_for_more_hardware_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

static std::atomic<int> g_init_device{-1};      // the device of the last pantas_device_init: where a GFA is parsed when a GPU is there
__attribute__((visibility("default"))) void pantas_device_init(int device) { try { init_device(device); g_init_device.store(device); } catch (...) {} }

// The GFA parsed on the GPU when the process has one (PANTAS_GFA_LOADER=host keeps it on the host cores, =device insists): the device
// form takes the graphs a graph builder writes and hands everything else back (load_graph_device returns nullptr).  The two loaders
// build the same arrays (tests/test_gpu_graph_loader.py compares them through the binary cache both can write).
static std::unique_ptr<Graph> load_graph_anywhere(const char* gfa_path, bool call_tags = false) {
  const char* how = getenv("PANTAS_GFA_LOADER");
  const bool host_only = how && !strcmp(how, "host");
  const bool insist = how && !strcmp(how, "device");
  if (!host_only) {
    int dev = g_init_device.load();
    if (dev < 0 && insist) dev = getenv("PANTAS_DEVICE") ? atoi(getenv("PANTAS_DEVICE")) : 0;
    if (dev >= 0) {
      std::unique_ptr<Graph> g;
      try {
        g = load_graph_device(gfa_path, dev, call_tags);
      } catch (const Error& e) {
        if (insist || e.kind == "FileNotFoundError") throw;
        g = nullptr;                        // (a HIP error while parsing: the host loader does not need the device)
      }
      if (g) return g;
      if (insist) throw Error("Unsupported", "PANTAS_GFA_LOADER=device: this GFA is not one the device form takes (or there is no device)");
    }
  }
  return load_graph(gfa_path);
}

__attribute__((visibility("default"))) int pantas_graph_load(const char* gfa_path, pantas_graph** out) {
  return guarded([&] {
    need(gfa_path && out, "null argument");
    double t0 = now_s();
    auto h = std::make_unique<pantas_graph>();
    h->g = load_graph_anywhere(gfa_path);
    h->load_s = now_s() - t0;
    *out = h.release();
  });
}

__attribute__((visibility("default"))) int pantas_graph_load_for_call(const char* gfa_path, pantas_graph** out) {
  return guarded([&] {
    need(gfa_path && out, "null argument");
    double t0 = now_s();
    auto h = std::make_unique<pantas_graph>();
    h->g = load_graph_anywhere(gfa_path, true);
    h->load_s = now_s() - t0;
    *out = h.release();
  });
}

__attribute__((visibility("default"))) int pantas_graph_load_cached(const char* gfa_path, const char* cache_path, int write_if_missing,
                                                                     pantas_graph** out, int* used_cache) {
  return guarded([&] {
    need(gfa_path && cache_path && out, "null argument");
    double t0 = now_s();
    auto h = std::make_unique<pantas_graph>();
    h->g = load_graph_cached(gfa_path, cache_path);
    if (used_cache) *used_cache = h->g ? 1 : 0;
    const bool timing = getenv("PANTAS_TIMING") != nullptr;
    if (h->g) {
      if (timing) fprintf(stderr, "[timing] graph cache: loaded %s in %.3f s\n", cache_path, now_s() - t0);
    } else {
      h->g = load_graph_anywhere(gfa_path);
      const bool saved = write_if_missing && save_graph_cache(*h->g, cache_path);
      if (timing) fprintf(stderr, "[timing] graph cache: parsed the GFA text (%.3f s)%s\n", now_s() - t0, saved ? ", cache written" : "");
    }
    h->load_s = now_s() - t0;
    *out = h.release();
  });
}

__attribute__((visibility("default"))) int pantas_graph_release_device(pantas_graph* g) {
  return guarded([&] {
    need(g, "null argument");
    g->stream.reset();
    g->xchg.reset();
    g->call_dev.reset();
    g->dev.reset();
    release_parked_device_memory();
  });
}

__attribute__((visibility("default"))) void pantas_graph_free(pantas_graph* g) {
  release_deferred();      // nothing parked outlives the handle it came from
  delete g;
  release_parked_device_memory();
}

__attribute__((visibility("default"))) int pantas_graph_get_stats(const pantas_graph* g, pantas_graph_stats* out) {
  return guarded([&] {
    need(g && g->g && out, "null argument");
    const Graph& G = *g->g;
    pantas_graph_stats o{};
    o.segments = G.V;
    o.link_slots = G.A;
    o.lines = G.n_lines;
    o.s_lines = G.n_s_lines;
    o.l_lines = G.n_l_lines;
    o.p_lines = G.n_p_lines;
    o.text_bytes = G.text.size;
    o.max_segment_len = G.max_seg_len;
    o.dense_names = G.names.dense ? 1u : 0u;
    o.load_seconds = g->load_s;
    put_out(out, o, "pantas_graph_stats");
  });
}

__attribute__((visibility("default"))) int pantas_graph_to_device(pantas_graph* g, int device) {
  return guarded([&] {
    need(g && g->g, "null graph handle");
    PhaseTimer pt("pantas_graph_to_device");
    if (!g->dev) g->dev = upload_graph(*g->g, device);
    pt.lap("CSR + counters in HBM");
  });
}

static int reset_run(pantas_graph* g, bool keep_tilde);
__attribute__((visibility("default"))) int pantas_augment_reset(pantas_graph* g) { return reset_run(g, false); }

static int reset_run(pantas_graph* g, bool keep_tilde) {
  return guarded([&] {
    need_device(g);
    require_augmentable(*g->g);
    if (!keep_tilde && (g->tilde_tried || !g->tilde_tab.empty())) {
      // the lengths worked out for the lines of one file (walk.h TildeCarry) say nothing about the next
      g->tilde_tab.clear();
      g->tilde_tried = false;
      set_tilde_table(*g->dev, g->tilde_tab);
    }
    reset_counters(*g->dev);
    g->have_result = false;
    g->res = AugmentResult();
    g->lines = g->rejected = g->skipped = 0;
    g->first_record_line = UINT64_MAX;
    g->run_first_line = UINT64_MAX;
    g->deferred.reset();
    g->stream.reset();
  });
}

__attribute__((visibility("default"))) int pantas_gaf_split(const char* gaf_path, int rank, int world, uint64_t* begin, uint64_t* end,
                     uint64_t* lines_before) {
  return guarded([&] {
    need(gaf_path && begin && end && world > 0 && rank >= 0 && rank < world, "bad argument");
    MappedFile f;
    f.open(gaf_path);
    auto cut = [&](int r) -> size_t {
      if (r <= 0) return 0;
      if (r >= world) return f.size;
      size_t p = f.size / size_t(world) * size_t(r);
      if (p == 0) return 0;
      const void* nl = memchr(f.data + p - 1, '\n', f.size - (p - 1));   // p is a line start iff data[p-1] == '\n'
      return nl ? size_t(static_cast<const char*>(nl) - f.data) + 1 : f.size;
    };
    *begin = cut(rank);
    *end = cut(rank + 1);
    if (lines_before) {
      uint64_t n = 0;
      const char* p = f.data;
      const char* e = f.data + *begin;
      while (p < e) {
        const void* nl = memchr(p, '\n', size_t(e - p));
        if (!nl) break;
        ++n;
        p = static_cast<const char*>(nl) + 1;
      }
      *lines_before = n;
    }
  });
}

static int effective_tokenizer(const pantas_graph* g) {
  int mode = g->tokenizer;
  if (mode == PANTAS_TOKENIZER_AUTO) {
    const char* e = getenv("PANTAS_TOKENIZER");
    if (e && !strcmp(e, "host")) mode = PANTAS_TOKENIZER_HOST;
    else if (e && !strcmp(e, "device")) mode = PANTAS_TOKENIZER_DEVICE;
    else if (e && *e && strcmp(e, "auto")) throw Error("ValueError", "PANTAS_TOKENIZER must be host, device or auto");
  }
  if (mode == PANTAS_TOKENIZER_AUTO) mode = g->g->names.dense ? PANTAS_TOKENIZER_DEVICE : PANTAS_TOKENIZER_HOST;
  return mode;
}

static void load_gaf_common(pantas_graph* g, const char* data, size_t size, int thr, uint64_t first_line,
                            pantas_batch** out, pantas_batch_stats* stats, const DeviceText* resident_text = nullptr, bool account = true) {
  GafBatch b;
  auto h = std::make_unique<pantas_batch>();
  pantas_batch_stats& s = h->st;
  const int mode = effective_tokenizer(g);
  s.tokenizer = mode;
  if (mode == PANTAS_TOKENIZER_DEVICE) {
    DeviceTokenizeTimes tm;
    h->dev = tokenize_gaf_device(*g->dev, *g->g, data, size, thr, first_line, b, &tm, resident_text);
    s.records = batch_lines(*h->dev);
    s.tokenize_seconds = tm.total_s - tm.h2d_s;
    s.upload_seconds = tm.h2d_s;
    s.tok_index_seconds = tm.index_s;
    s.tok_measure_seconds = tm.measure_s;
    s.tok_write_seconds = tm.write_s;
    s.tok_bucket_seconds = tm.bucket_s;
    uint64_t pn = 0, ops = 0;
    batch_sizes(*h->dev, &pn, &ops);
    s.path_nodes = pn;
    s.cs_ops = ops;
  } else {
    double t0 = now_s();
    tokenize_gaf(*g->g, data, size, thr, first_line, 0, b);
    if (!getenv("PANTAS_NO_BUCKET")) bucket_by_locus(b);
    double t1 = now_s();
    h->dev = upload_batch(*g->dev, b);
    double t2 = now_s();
    s.records = b.size();
    s.path_nodes = b.nodes.size();
    s.cs_ops = b.ops.size();
    s.tokenize_seconds = t1 - t0;
    s.upload_seconds = t2 - t1;
  }
  s.lines = b.lines_total;
  s.rejected = b.rejected;
  s.skipped = b.skipped;
  s.del_records = b.del_lines;
  s.del_path_nodes = b.del_nodes;
  s.text_bytes = b.bytes;
  s.alg_bytes = batch_alg_bytes(*h->dev);
  s.short_records = batch_short_records(*h->dev);
  s.short_alg_bytes = batch_alg_bytes_short(*h->dev);
  if (account) {   // (a batch tokenised a second time for the main pass of a streamed run was counted the first time)
    g->lines += b.lines_total;
    g->rejected += b.rejected;
    g->skipped += b.skipped;
    g->first_record_line = std::min(g->first_record_line, b.first_record_line);
    g->run_first_line = std::min(g->run_first_line, first_line);
    if (b.deferred && !g->deferred) g->deferred = b.deferred;
  }
  if (stats) put_out(stats, s, "pantas_batch_stats");
  *out = h.release();
}

__attribute__((visibility("default"))) int pantas_augment_tokenizer(pantas_graph* g, int mode) {
  return guarded([&] {
    need(g && g->g, "null graph");
    need(mode == PANTAS_TOKENIZER_AUTO || mode == PANTAS_TOKENIZER_HOST || mode == PANTAS_TOKENIZER_DEVICE, "bad tokenizer mode");
    g->tokenizer = mode;
  });
}

__attribute__((visibility("default"))) int pantas_batch_download(pantas_graph* g, pantas_batch* b, int32_t* hdr, uint32_t* node_off, uint32_t* op_off,
                                                                  uint32_t* nodes, uint32_t* ops) {
  return guarded([&] {
    need_device(g);
    need(b && b->dev && hdr && node_off && op_off && nodes && ops, "null argument");
    GafBatch h;
    download_batch(*g->dev, *b->dev, h);
    static_assert(sizeof(LineHdr) == 16, "LineHdr is 4 words");
    if (!h.hdr.empty()) memcpy(hdr, h.hdr.data(), h.hdr.size() * sizeof(LineHdr));
    memcpy(node_off, h.node_off.data(), h.node_off.size() * 4);
    memcpy(op_off, h.op_off.data(), h.op_off.size() * 4);
    if (!h.nodes.empty()) memcpy(nodes, h.nodes.data(), h.nodes.size() * 4);
    if (!h.ops.empty()) memcpy(ops, h.ops.data(), h.ops.size() * 4);
  });
}

__attribute__((visibility("default"))) int pantas_augment_load_gaf(pantas_graph* g, const char* gaf_path, int thr, uint64_t begin, uint64_t end,
                            uint64_t first_line, pantas_batch** out, pantas_batch_stats* stats) {
  return guarded([&] {
    need_device(g);
    need(gaf_path && out, "null argument");
    MappedFile f;
    f.open(gaf_path);
    if (end == 0 || end > f.size) end = f.size;
    need(begin <= end, "bad byte range");
    load_gaf_common(g, f.data + begin, size_t(end - begin), thr, first_line, out, stats);
  });
}

__attribute__((visibility("default"))) int pantas_augment_load_gaf_mem(pantas_graph* g, const char* gaf_text, uint64_t size, int thr,
                                                                        uint64_t first_line, pantas_batch** out, pantas_batch_stats* stats) {
  return guarded([&] {
    need_device(g);
    need(gaf_text && out, "null argument");
    load_gaf_common(g, gaf_text, size_t(size), thr, first_line, out, stats);
  });
}

__attribute__((visibility("default"))) void pantas_batch_free(pantas_batch* b) { delete b; }

__attribute__((visibility("default"))) int pantas_augment_launch(pantas_graph* g, pantas_batch* b, pantas_launch_stats* out) {
  return guarded([&] {
    need_device(g);
    need(b && b->dev, "null batch");
    float r = 0, m = 0;
    launch_augment(*g->dev, *b->dev, &r, &m);
    pantas_launch_stats o{};
    o.rare_ms = r; o.main_ms = m; o.main_short_ms = last_short_kernel_ms(*g->dev);
    put_out(out, o, "pantas_launch_stats");
  });
}

__attribute__((visibility("default"))) int pantas_augment_launch_async(pantas_graph* g, pantas_batch* b) {
  return guarded([&] {
    need_device(g);
    need(b && b->dev, "null batch");
    launch_augment_async(*g->dev, *b->dev);
  });
}

__attribute__((visibility("default"))) int pantas_augment_launch_collect(pantas_graph* g, float* ms, uint64_t capacity, uint64_t* n) {
  return guarded([&] {
    need_device(g);
    need(n, "null argument");
    need(ms || capacity == 0, "null buffer");
    std::vector<LaunchTimes> t(std::min<uint64_t>(capacity, 64));      // (at most 64 launches are ever in flight)
    const size_t got = launch_collect(*g->dev, t.data(), t.size());
    *n = got;
    for (size_t i = 0; i < std::min<size_t>(got, t.size()); ++i) { ms[3 * i] = t[i].rare_ms; ms[3 * i + 1] = t[i].main_ms; ms[3 * i + 2] = t[i].main_short_ms; }
  });
}

__attribute__((visibility("default"))) int pantas_augment_reserve(pantas_graph* g, uint64_t keys) {
  return guarded([&] {
    need_device(g);
    reserve_hist(*g->dev, keys);
  });
}

__attribute__((visibility("default"))) int pantas_augment_launch_rare(pantas_graph* g, pantas_batch* b, float* ms) {
  return guarded([&] {
    need_device(g);
    need(b && b->dev, "null batch");
    float t = launch_rare(*g->dev, *b->dev);
    if (ms) *ms = t;
  });
}

__attribute__((visibility("default"))) int pantas_augment_launch_main(pantas_graph* g, pantas_batch* b, float* ms) {
  return guarded([&] {
    need_device(g);
    need(b && b->dev, "null batch");
    float t = launch_main(*g->dev, *b->dev);
    if (ms) *ms = t;
  });
}

__attribute__((visibility("default"))) int pantas_augment_flags_export(pantas_graph* g, uint32_t* codes, uint64_t capacity, uint64_t* n) {
  return guarded([&] {
    need_device(g);
    need(n, "null argument");
    std::vector<uint32_t> v;
    export_flags(*g->dev, v);
    *n = v.size();
    if (codes) {
      need(capacity >= v.size(), "buffer too small");
      if (!v.empty()) memcpy(codes, v.data(), v.size() * 4);
    }
  });
}

__attribute__((visibility("default"))) int pantas_augment_flags_import(pantas_graph* g, const uint32_t* codes, uint64_t n) {
  return guarded([&] {
    need_device(g);
    need(codes || n == 0, "null argument");
    import_flags(*g->dev, codes, size_t(n));
  });
}

__attribute__((visibility("default"))) int pantas_augment_counters(pantas_graph* g, void** dev_ptr, uint64_t* words) {
  return guarded([&] {
    need_device(g);
    need(dev_ptr && words, "null argument");
    CounterPtrs p = counter_ptrs(*g->dev);
    *dev_ptr = p.base;
    *words = p.words;
  });
}

__attribute__((visibility("default"))) int pantas_augment_track_touched(pantas_graph* g, int on) {
  return guarded([&] {
    need_device(g);
    set_track_touched(*g->dev, on != 0);
  });
}

__attribute__((visibility("default"))) int pantas_counters_bitmap(pantas_graph* g, uint32_t block_words, void** dev_bitmap, uint64_t* bitmap_words) {
  return guarded([&] {
    need_device(g);
    need(dev_bitmap && bitmap_words, "null argument");
    exchange_bitmap(*g->dev, g->xchg, block_words, dev_bitmap, bitmap_words);
  });
}

__attribute__((visibility("default"))) int pantas_counters_pack(pantas_graph* g, const void* dev_bitmaps, uint32_t n_bitmaps, void** dev_staging,
                                                                uint64_t* staging_words) {
  return guarded([&] {
    need_device(g);
    need(dev_staging && staging_words, "null argument");
    if (!g->xchg) throw Error("ProtocolError", "pantas_counters_pack before pantas_counters_bitmap");
    exchange_pack(*g->dev, *g->xchg, dev_bitmaps, n_bitmaps, dev_staging, staging_words);
  });
}

__attribute__((visibility("default"))) int pantas_counters_unpack(pantas_graph* g) {
  return guarded([&] {
    need_device(g);
    if (!g->xchg) throw Error("ProtocolError", "pantas_counters_unpack before pantas_counters_pack");
    exchange_unpack(*g->dev, *g->xchg);
  });
}

// ---- the collective itself (bitmap all-gather -> pack -> all-reduce -> unpack) on the caller's RCCL communicator ----------------
namespace {
struct RcclFns {
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, void*) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, void*) = nullptr;
  int (*comm_count)(const void*, int*) = nullptr;
};
RcclFns resolve_rccl(const pantas_rccl_api* api) {
  RcclFns f;
  if (api) {
    const pantas_rccl_api a = get_in(api, "pantas_rccl_api");
    f.all_reduce = a.all_reduce; f.all_gather = a.all_gather; f.comm_count = a.comm_count;
  } else {
    void* lib = RTLD_DEFAULT;
    if (!dlsym(lib, "ncclAllReduce")) {
      lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) throw Error("ArgumentError", "pantas_counters_allreduce: no RCCL in the process and librccl.so cannot be loaded; pass a pantas_rccl_api");
    }
    f.all_reduce = reinterpret_cast<decltype(f.all_reduce)>(dlsym(lib, "ncclAllReduce"));
    f.all_gather = reinterpret_cast<decltype(f.all_gather)>(dlsym(lib, "ncclAllGather"));
    f.comm_count = reinterpret_cast<decltype(f.comm_count)>(dlsym(lib, "ncclCommCount"));
  }
  if (!f.all_reduce || !f.all_gather || !f.comm_count) throw Error("ArgumentError", "pantas_counters_allreduce: ncclAllReduce / ncclAllGather / ncclCommCount missing");
  return f;
}
constexpr int RCCL_UINT32 = 3, RCCL_SUM = 0;      // rccl.h: ncclUint32, ncclSum
}  // namespace

__attribute__((visibility("default"))) int pantas_counters_allreduce(pantas_graph* g, const pantas_rccl_api* api, void* comm, void* stream, uint32_t block_words,
                                                                     pantas_allreduce_stats* out) {
  return guarded([&] {
    need_device(g);
    need(comm != nullptr, "null communicator");
    need(block_words == 0 || block_words == 8 || block_words == 16 || block_words == 32 || block_words == 64, "block_words must be 0, 8, 16, 32 or 64");
    const RcclFns f = resolve_rccl(api);
    auto ok = [](int rc, const char* what) { if (rc != 0) throw Error("RcclError", std::string(what) + " failed with ncclResult " + std::to_string(rc)); };
    int world = 0;
    ok(f.comm_count(comm, &world), "ncclCommCount");
    need(world >= 1, "communicator without ranks");
    const CounterPtrs cp = counter_ptrs(*g->dev);
    void* counters = cp.base;
    const uint64_t words = cp.words;
    void* s = stream ? stream : exchange_stream(*g->dev);
    auto stream_wait = [&](void* st2) { exchange_stream_wait(*g->dev, st2); };
    pantas_allreduce_stats st{};
    st.world = uint32_t(world);
    st.block_words = block_words;
    st.counter_bytes = words * 4;
    st.exchanged_bytes = words * 4;
    const double t0 = now_s();
    bool done = false;
    if (block_words && words) {
      void* bm = nullptr;
      uint64_t bw = 0;
      exchange_bitmap(*g->dev, g->xchg, block_words, &bm, &bw);
      void* all = exchange_scratch_take(*g->dev, size_t(world) * bw * 4);
      ok(f.all_gather(bm, all, bw, RCCL_UINT32, comm, s), "ncclAllGather");
      stream_wait(s);
      st.bitmap_ms = float((now_s() - t0) * 1e3);
      double t1 = now_s();
      void* staging = nullptr;
      uint64_t sw = 0;
      exchange_pack(*g->dev, *g->xchg, all, uint32_t(world), &staging, &sw);
      exchange_scratch_give(*g->dev, all);
      st.pack_ms = float((now_s() - t1) * 1e3);
      if (2 * sw <= words) {
        t1 = now_s();
        if (sw) { ok(f.all_reduce(staging, staging, sw, RCCL_UINT32, RCCL_SUM, comm, s), "ncclAllReduce"); stream_wait(s); }
        st.allreduce_ms = float((now_s() - t1) * 1e3);
        t1 = now_s();
        exchange_unpack(*g->dev, *g->xchg);
        st.unpack_ms = float((now_s() - t1) * 1e3);
        st.mode = 1;
        st.exchanged_bytes = sw * 4 + uint64_t(world) * bw * 4;
        done = true;
      } else {
        exchange_drop(*g->xchg);          // too dense: the staging buffer is not used
      }
    }
    if (!done) {
      const double t1 = now_s();
      if (words) { ok(f.all_reduce(counters, counters, words, RCCL_UINT32, RCCL_SUM, comm, s), "ncclAllReduce"); stream_wait(s); }
      st.allreduce_ms = float((now_s() - t1) * 1e3);
    }
    st.total_ms = float((now_s() - t0) * 1e3);
    if (out) put_out(out, st, "pantas_allreduce_stats");
  });
}

__attribute__((visibility("default"))) int pantas_augment_finish(pantas_graph* g) {
  return guarded([&] {
    need_device(g);
    PhaseTimer pt("pantas_augment_finish");
    // ('~' first with nothing left behind is the reference's UnboundLocalError only on the first record of the FILE: the first
    // record of a rank's byte range has lines of another rank in front of it, and what they leave behind is not known here)
    check_device_status(*g->dev, g->run_first_line == 0 ? g->first_record_line : UINT64_MAX);      // a line that fails in the walk, before ...
    if (g->deferred) {                          // ... the line the tokeniser stopped at
      Error e = *g->deferred;
      g->deferred.reset();
      throw e;
    }
    download_result(*g->dev, *g->g, g->res);
    pt.lap("counters + sparse tables to the host");
    g->res.lines_total = g->lines;
    g->res.rejected = g->rejected;
    g->res.skipped = g->skipped;
    g->have_result = true;
    release_deferred();      // the device work of the run is over: what the parse left behind is freed beside whatever comes next
  });
}

__attribute__((visibility("default"))) int pantas_augment_sparse_count(pantas_graph* g, uint64_t* rows) {
  return guarded([&] {
    need(g && g->have_result && rows, "no finished run");
    *rows = g->res.hist.size() + g->res.novel.size();
  });
}

__attribute__((visibility("default"))) int pantas_augment_sparse_export(pantas_graph* g, uint64_t* out, uint64_t capacity) {
  return guarded([&] {
    need(g && g->have_result && out, "no finished run");
    need(capacity >= g->res.hist.size() + g->res.novel.size(), "buffer too small");
    uint64_t* p = out;
    for (const HistEntry& h : g->res.hist) {
      uint32_t code = h.is_common ? COMMON_CODE : (uint32_t(h.pos + 1) & 0x7FFFFFFFu);
      *p++ = (uint64_t(h.node) << 32) | (uint64_t(h.side) << 31) | code;
      *p++ = h.count;
      *p++ = h.first;
      *p++ = KIND_HIST;
    }
    for (const NovelLink& n : g->res.novel) {
      *p++ = (uint64_t(n.u) << 32) | n.v;
      *p++ = n.count;
      *p++ = n.first;
      *p++ = KIND_NOVEL;
    }
  });
}

__attribute__((visibility("default"))) int pantas_augment_sparse_import(pantas_graph* g, const uint64_t* rows, uint64_t n) {
  return guarded([&] {
    need(g && g->have_result && (rows || n == 0), "no finished run");
    // merge by key: counts add, first-seen takes the minimum
    std::vector<HistEntry>& H = g->res.hist;
    std::vector<NovelLink>& N = g->res.novel;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t key = rows[4 * i], cnt = rows[4 * i + 1], first = rows[4 * i + 2], kind = rows[4 * i + 3];
      if (kind == KIND_HIST) {
        uint32_t code = uint32_t(key) & 0x7FFFFFFFu;
        bool common = code == COMMON_CODE;
        H.push_back(HistEntry{uint32_t(key >> 32), uint32_t(key >> 31) & 1u, common ? 0 : int32_t(code) - 1,
                              uint32_t(cnt), first, common});
      } else {
        N.push_back(NovelLink{uint32_t(key >> 32), uint32_t(key), uint32_t(cnt), first});
      }
    }
    auto hkey = [](const HistEntry& h) {
      return std::make_tuple(h.node, h.side, h.is_common ? 1 : 0, h.pos);
    };
    std::sort(H.begin(), H.end(), [&](const HistEntry& a, const HistEntry& b) { return hkey(a) < hkey(b); });
    size_t w = 0;
    for (size_t i = 0; i < H.size(); ++i) {
      if (w && hkey(H[w - 1]) == hkey(H[i])) {
        H[w - 1].count += H[i].count;
        H[w - 1].first = std::min(H[w - 1].first, H[i].first);
      } else H[w++] = H[i];
    }
    H.resize(w);
    std::sort(N.begin(), N.end(), [](const NovelLink& a, const NovelLink& b) {
      return a.u != b.u ? a.u < b.u : a.v < b.v;
    });
    w = 0;
    for (size_t i = 0; i < N.size(); ++i) {
      if (w && N[w - 1].u == N[i].u && N[w - 1].v == N[i].v) {
        N[w - 1].count += N[i].count;
        N[w - 1].first = std::min(N[w - 1].first, N[i].first);
      } else N[w++] = N[i];
    }
    N.resize(w);
    finalize_result(g->res);
  });
}

__attribute__((visibility("default"))) int pantas_augment_read(pantas_graph* g, uint32_t* nc, uint32_t* il0, uint32_t* oll, uint32_t* rc) {
  return guarded([&] {
    need(g && g->have_result, "no finished run");
    const AugmentResult& r = g->res;
    if (nc) memcpy(nc, r.nc.data(), r.nc.size() * 4);
    if (il0) memcpy(il0, r.il0.data(), r.il0.size() * 4);
    if (oll) memcpy(oll, r.oll.data(), r.oll.size() * 4);
    if (rc) memcpy(rc, r.rc.data(), r.rc.size() * 4);
  });
}

__attribute__((visibility("default"))) int pantas_augment_write_gfa(pantas_graph* g, int fd) {
  return guarded([&] {
    need(g && g->have_result, "no finished run");
    write_augmented_gfa(*g->g, g->res, fd);
  });
}

// ---- a run over a GAF of any size (alignments_augmentation_from_gaf.py:135-138 reads line by line, no limit) ---------------------
// The byte range is cut at line starts into batches of at most PANTAS_BATCH_BYTES of text (default 2 GiB; the GPU tokeniser's
// own limits are ~8.6 GB of text and 2^32 path nodes per batch).  Rare-key pass: batch by batch, tokenise -> rare-key kernel,
// with the text of the next batch on its way to HBM meanwhile (TextPrefetcher).  The tokenised records stay in HBM as long as
// they fit the residency budget (PANTAS_RESIDENT_BYTES, default 60 % of the memory free at the start): records are smaller than
// their text and 288 GB hold billions of reads beside a whole-genome graph.  Main pass: resident batches are launched as they
// are, dropped ones are tokenised again first.  The rare-before-main order is what the first-seen order of the common IL/OL
// key needs (augment.hip): every flag must be set before the first main kernel runs.

static uint64_t env_u64(const char* name, uint64_t dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  char* end = nullptr;
  const unsigned long long v = strtoull(e, &end, 10);
  if (end == e || *end) throw Error("ValueError", std::string(name) + " must be a non-negative integer");
  return v;
}

static void add_stats(pantas_batch_stats& t, const pantas_batch_stats& s) {
  t.lines += s.lines; t.rejected += s.rejected; t.skipped += s.skipped; t.records += s.records; t.path_nodes += s.path_nodes;
  t.cs_ops += s.cs_ops; t.del_records += s.del_records; t.del_path_nodes += s.del_path_nodes; t.text_bytes += s.text_bytes;
  t.alg_bytes += s.alg_bytes; t.tokenize_seconds += s.tokenize_seconds; t.upload_seconds += s.upload_seconds;
  t.tokenizer = s.tokenizer;
  t.tok_index_seconds += s.tok_index_seconds; t.tok_measure_seconds += s.tok_measure_seconds; t.tok_write_seconds += s.tok_write_seconds;
  t.tok_bucket_seconds += s.tok_bucket_seconds; t.short_records += s.short_records; t.short_alg_bytes += s.short_alg_bytes;
}

// next cut at or behind `want`: a position that follows a '\n' (or the range end)
static uint64_t cut_at_line_start(const MappedFile& f, uint64_t want, uint64_t end) {
  if (want >= end) return end;
  if (want == 0) return 0;
  const void* nl = memchr(f.data + want - 1, '\n', size_t(end - (want - 1)));
  return nl ? uint64_t(static_cast<const char*>(nl) - f.data) + 1 : end;
}

static uint64_t batch_device_bytes(const pantas_batch_stats& s) {
  // hdr 16 + offsets 8 per record, 4 per path node and cs op, 4 per record with a '-'; the tiles of the position-parallel kernel: a
  // 32-byte line record (with the op summary) per record, node + position word (6 bytes) per lane of a tile.  A record never
  // straddles a tile, so the worst packing is records of 33 positions, one per tile of 64: 64 / 33 * 6 = 11.7 bytes per path node --
  // that bound, not the 5 % of padding the synthetic reads (6 nodes) show, is what is budgeted: reads of 17 .. 31 nodes leave 13 .. 20
  // lanes of a tile empty.  (rec, 48 bytes per record, only exists for a batch walked by a record-per-lane kernel; it is counted:
  // the bound holds for the fallback too.)
  return 72 * s.records + 4 * (s.path_nodes + s.cs_ops) + 4 * s.del_records + 32 * s.records + 12 * s.path_nodes + 4096;
}

static void stream_rare_pass(pantas_graph* g, const char* gaf_path, int thr, uint64_t begin, uint64_t end, uint64_t first_line,
                             pantas_batch_stats* stats, float* rare_ms, bool universal_newlines = false) {
  auto st = std::make_unique<StreamState>();
  st->file.open(gaf_path);
  if (universal_newlines) {
    // a '\r' that no '\n' follows ends its line, as it does for the reference (common.h lone_cr): the run reads a copy of the text
    // with those bytes turned into '\n' -- cut into batches, counted in lines and tokenised like any other text
    universal_newlines_copy(st->file.data, st->file.size, st->text_copy);
    st->file.attach(st->text_copy.data(), st->text_copy.size());
  }
  st->thr = thr;
  if (end == 0 || end > st->file.size) end = st->file.size;
  need(begin <= end, "bad byte range");
  const uint64_t cap = std::max<uint64_t>(1, env_u64("PANTAS_BATCH_BYTES", uint64_t(2) << 30));
  if (cap > (uint64_t(8) << 30)) throw Error("ValueError", "PANTAS_BATCH_BYTES above 8 GiB (the tokeniser's limit per batch)");
  for (uint64_t b = begin; b < end;) {
    const uint64_t e = cut_at_line_start(st->file, std::min(end, b + cap), end);
    StreamBatch sb;
    sb.begin = b; sb.end = e;
    st->batches.push_back(std::move(sb));
    st->max_batch_bytes = std::max(st->max_batch_bytes, e - b);
    b = e;
  }
  st->resident_budget = env_u64("PANTAS_RESIDENT_BYTES", (device_free_bytes(*g->dev) + parked_device_bytes()) / 10 * 6);   // (parked loader blocks are given back on demand)
  const bool on_device = effective_tokenizer(g) == PANTAS_TOKENIZER_DEVICE;
  TextPrefetcherPtr pf;
  // one batch: nothing to overlap, the tokeniser copies the text itself (and needs no second text buffer)
  if (on_device && st->batches.size() > 1) pf = make_text_prefetcher(*g->dev, size_t(st->max_batch_bytes));
  PhaseTimer pt("augment_stream rare pass");
  size_t started = 0;
  auto start_next = [&] {
    if (pf && started < st->batches.size()) {
      const StreamBatch& nb = st->batches[started++];
      prefetch_start(*pf, st->file.data + nb.begin, size_t(nb.end - nb.begin));
    }
  };
  start_next();
  float ms_total = 0;
  uint64_t line = first_line;
  size_t n_loaded = 0;
  for (size_t i = 0; i < st->batches.size(); ++i) {
    StreamBatch& sb = st->batches[i];
    sb.first_line = line;
    DeviceText txt;
    if (pf) { txt = prefetch_get(*pf); start_next(); }
    pantas_batch* raw = nullptr;
    pantas_batch_stats bs{};
    bs.struct_size = sizeof bs;
    load_gaf_common(g, st->file.data + sb.begin, size_t(sb.end - sb.begin), thr, line, &raw, &bs, pf ? &txt : nullptr, true);
    std::unique_ptr<pantas_batch> b(raw);
    add_stats(st->total, bs);
    line += bs.lines;
    ms_total += launch_rare(*g->dev, *b->dev);
    sb.device_bytes = batch_device_bytes(bs);
    if (st->resident_bytes + sb.device_bytes <= st->resident_budget) {
      st->resident_bytes += sb.device_bytes;
      sb.resident = std::move(b);
      ++st->kept_resident;
    }
    ++n_loaded;
    if (g->deferred) break;    // the reference dies at this line: what follows it in the file is never read
  }
  if (pf && started > n_loaded) {
    // a range was already on its way when the run stopped at a raising line: let its copy finish before the buffers go
    for (size_t k = n_loaded; k < started; ++k) (void)prefetch_get(*pf);
  }
  st->batches.resize(n_loaded);
  pt.lap("tokenise + rare-key kernels");
  put_out(stats, st->total, "pantas_batch_stats");
  if (rare_ms) *rare_ms = ms_total;
  g->stream = std::move(st);
}

static void stream_main_pass(pantas_graph* g, float* main_ms) {
  if (!g->stream) throw Error("ProtocolError", "pantas_augment_stream_main before pantas_augment_stream_rare");
  StreamState& st = *g->stream;
  if (st.main_done) throw Error("ProtocolError", "pantas_augment_stream_main twice in one run");
  PhaseTimer pt("augment_stream main pass");
  const bool on_device = effective_tokenizer(g) == PANTAS_TOKENIZER_DEVICE;
  std::vector<size_t> redo;
  for (size_t i = 0; i < st.batches.size(); ++i) if (!st.batches[i].resident) redo.push_back(i);
  TextPrefetcherPtr pf;
  if (on_device && redo.size() > 1) pf = make_text_prefetcher(*g->dev, size_t(st.max_batch_bytes));
  size_t started = 0;
  auto start_next = [&] {
    if (pf && started < redo.size()) {
      const StreamBatch& nb = st.batches[redo[started++]];
      prefetch_start(*pf, st.file.data + nb.begin, size_t(nb.end - nb.begin));
    }
  };
  float ms_total = 0;
  // resident batches first: their records are where they are
  for (StreamBatch& sb : st.batches)
    if (sb.resident) ms_total += launch_main(*g->dev, *sb.resident->dev);
  pt.lap("main kernels (resident batches)");
  start_next();
  for (size_t k = 0; k < redo.size(); ++k) {
    StreamBatch& sb = st.batches[redo[k]];
    DeviceText txt;
    if (pf) { txt = prefetch_get(*pf); start_next(); }
    pantas_batch* raw = nullptr;
    load_gaf_common(g, st.file.data + sb.begin, size_t(sb.end - sb.begin), st.thr, sb.first_line, &raw, nullptr, pf ? &txt : nullptr, false);
    std::unique_ptr<pantas_batch> b(raw);
    // the same records in the same order as in the rare-key pass: it stands in for the batch whose rare-key kernel ran
    adopt_rare_done(*g->dev, *b->dev);
    ms_total += launch_main(*g->dev, *b->dev);
    ++st.retokenised;
  }
  if (!redo.empty()) pt.lap("re-tokenise + main kernels (dropped batches)");
  st.main_done = true;
  if (main_ms) *main_ms = ms_total;
}

__attribute__((visibility("default"))) int pantas_augment_stream_rare(pantas_graph* g, const char* gaf_path, int thr, uint64_t begin, uint64_t end,
                                                                       uint64_t first_line, pantas_batch_stats* stats, float* rare_ms) {
  return guarded([&] {
    need_device(g);
    need(gaf_path != nullptr, "null argument");
    if (g->stream) throw Error("ProtocolError", "one streamed byte range per run: pantas_augment_reset first");
    stream_rare_pass(g, gaf_path, thr, begin, end, first_line, stats, rare_ms);
    if (g->deferred && is_lone_cr_error(*g->deferred) && begin == 0 && (end == 0 || end >= g->stream->file.size)) {
      // the whole file is this run's: read it again the way Python does (a byte range of a file shared between ranks keeps the
      // refusal: where its lines start and how they are numbered is the caller's arithmetic)
      if (reset_run(g, true) != PANTAS_OK) throw Error(g_kind.c_str(), g_err);
      stream_rare_pass(g, gaf_path, thr, 0, 0, first_line, stats, rare_ms, true);
    }
  });
}

__attribute__((visibility("default"))) int pantas_augment_stream_main(pantas_graph* g, float* main_ms) {
  return guarded([&] {
    need_device(g);
    stream_main_pass(g, main_ms);
  });
}

__attribute__((visibility("default"))) int pantas_augment_stream_info(const pantas_graph* g, uint64_t* batches, uint64_t* resident, uint64_t* retokenised) {
  return guarded([&] {
    need(g && g->stream, "no streamed run on this handle");
    if (batches) *batches = g->stream->batches.size();
    if (resident) *resident = g->stream->kept_resident;
    if (retokenised) *retokenised = g->stream->retokenised;
  });
}

__attribute__((visibility("default"))) int pantas_augment_run(pantas_graph* g, const char* gaf_path, int thr, int fd_out, pantas_batch_stats* stats,
                       pantas_launch_stats* launch) {
  int rc = PANTAS_OK;
  for (int attempt = 0; attempt < 7; ++attempt) {
    rc = reset_run(g, attempt > 0);
    if (rc) return rc;
    float rare = 0, mainms = 0;
    rc = pantas_augment_stream_rare(g, gaf_path, thr, 0, 0, 0, stats, &rare);
    if (rc) return rc;
    rc = pantas_augment_stream_main(g, &mainms);
    if (rc) return rc;
    if (launch) {
      pantas_launch_stats o{};
      o.rare_ms = rare; o.main_ms = mainms;
      const int r2 = guarded([&] { put_out(launch, o, "pantas_launch_stats"); });
      if (r2) return r2;
    }
    rc = pantas_augment_finish(g);
    if (rc != PANTAS_OK && g_kind == "Unsupported" && g_err.find(TILDE_FIRST_MARK) != std::string::npos && g->stream) {
      if (!g->tilde_tried) {
        // a line whose first cs op is '~' takes its length from the line in front of it (walk.h TildeCarry): work those lengths out on the
        // host, once, and run again with them
        g->tilde_tried = true;
        const int r2 = guarded([&] {
          g->tilde_tab = tilde_carries(*g->g, g->stream->file.data, g->stream->file.size, thr, 0);
          set_tilde_table(*g->dev, g->tilde_tab);
        });
        if (r2) return r2;
        continue;
      }
      // still stopped at such a line: the table says nothing in front of it took an op in hand
      const size_t at = g_err.find("GAF line ");
      const uint64_t line = at == std::string::npos ? 0 : strtoull(g_err.c_str() + at + 9, nullptr, 10);
      for (const TildeEntry& e : g->tilde_tab)
        if (uint64_t(e.line_id) + 1 == line && e.carry == TILDE_UNBOUND)
          return fail(Error("UnboundLocalError", "GAF line " + std::to_string(line) + ": '~' is the first cs op the walk reaches and no line before it left a length behind"));
    }
    if (rc == PANTAS_E_INTERNAL && g_kind == "TableOverflow") {
      // a table sized from a guess was too small: enlarge it and redo the run from the text (every batch is tokenised
      // again, so a deferred tokeniser error comes back with it)
      const bool hist = g_err.find("hist") != std::string::npos;
      if (hist) boost_hist_table(*g->dev); else boost_novel_table(*g->dev);
      // (only said where the test switch made it happen: the reference's stderr is part of what the command line is compared on)
      if (getenv("PANTAS_TABLE_SLOTS_FIRST")) fprintf(stderr, "pantas: the %s table was full, the run is redone with four times its room\n", hist ? "IL/OL key" : "novel-link");
      continue;
    }
    break;
  }
  if (g->stream) {   // the records are not needed once the counters are final
    for (StreamBatch& sb : g->stream->batches) sb.resident.reset();
    g->stream->file.close();
  }
  if (rc) return rc;
  if (fd_out >= 0) rc = pantas_augment_write_gfa(g, fd_out);
  return rc;
}


__attribute__((visibility("default"))) int pantas_call_prepare(pantas_graph* g, const char* gtf_path, int device) {
  return guarded([&] {
    need(g && g->g && gtf_path, "null argument");
    auto t = std::make_unique<CallTables>();
    build_call_tables(*g->g, gtf_path, *t);
    double t0 = now_s();
    g->call_dev = upload_call(*g->g, *t, device);
    g->call_upload_s = now_s() - t0;
    g->call_tables = std::move(t);
  });
}

__attribute__((visibility("default"))) int pantas_call_run(pantas_graph* g, const pantas_call_opts* opts_in, int fd_csv, int fd_err,
                                                            pantas_call_stats* stats_out) {
  return guarded([&] {
    need(g && g->g && opts_in, "null argument");
    if (!g->call_dev || !g->call_tables) throw Error("NoDevice", "call tables are not on a device: call pantas_call_prepare first");
    const pantas_call_opts opts_v = get_in(opts_in, "pantas_call_opts");   // fields the caller's declaration lacks read as 0
    const pantas_call_opts* opts = &opts_v;
    if (stats_out) (void)declared_size(stats_out, "pantas_call_stats");   // refuse before any work is done
    CallOptions o;
    o.rca = opts->rca;
    o.isw = opts->isw;
    o.minintron = opts->minintronsize;
    o.events = opts->events;
    o.novel = opts->novel != 0;
    o.no_annotated = opts->no_annotated != 0;
    o.debug = opts->debug != 0;
    if (opts->junction) { o.has_junction = true; o.junction = opts->junction; }
    if (opts->shard_world > 1) {
      need(opts->shard_rank < opts->shard_world, "shard_rank out of range");
      o.shard_rank = opts->shard_rank;
      o.shard_world = opts->shard_world;
    }
    CallOutput co;
    CallTimes times;
    run_call(*g->call_dev, *g->g, *g->call_tables, o, co, times);
    if (opts->release_device) {       // (see pantas_graph_release_device: scrubbed beside the rendering, not in front of the next process)
      g->stream.reset(); g->xchg.reset(); g->call_dev.reset(); g->dev.reset();
      release_parked_device_memory();
    }
    const RawVector<EventRec>& recs = co.recs;
    double t0 = now_s();
    // the reference prints rows as it finds them and dies on an input it cannot handle; here nothing may reach stdout
    // unless the whole call succeeds, and every failure of that kind is known before rendering starts (run_call
    // raised it) -- so the rows can go to the descriptor block by block instead of through one 10 GB string
    std::string err;
    if (fd_csv >= 0) write_fd(fd_csv, std::string(CALL_CSV_HEADER) + "\n");
    release_deferred();      // (common.h) the tag records of the table build are freed beside the rendering
    render_events_to_fd(*g->g, *g->call_tables, recs.data(), recs.size(), co.side.data(), co.tasks.data(), co.paths.data(), fd_csv, err);
    double t1 = now_s();
    if (fd_err >= 0) write_fd(fd_err, err);
    if (stats_out) {
      pantas_call_stats st_local{};
      pantas_call_stats* stats = &st_local;
      const CallTables& t = *g->call_tables;
      stats->junctions = t.junctions.size();
      stats->novel_links = t.novels.size();
      stats->evaluated = times.junctions;
      const unsigned T = recs.size() < 100000 ? 1u : hw_threads();
      std::vector<uint64_t> part(T, 0);
      parallel_chunks(recs.size(), T, [&](unsigned c, size_t lo, size_t hi) {
        uint64_t rows = 0;
        for (size_t i = lo; i < hi; ++i) {
          const EventRec& r = recs[i];
          if (r.type == T_IR_ANN || r.type == T_IR1_NOV || r.type == T_IR2_NOV) rows += co.tasks[r.j1].status == 1 ? 1 : 0;
          else rows += r.type < T_SKIP_ES ? 1 : 0;
        }
        part[c] = rows;
      });
      uint64_t rows = 0;
      for (uint64_t x : part) rows += x;
      stats->events = rows;
      stats->ex_entries = t.ex_tag.size();
      stats->jn_entries = t.jn_ftx.size();
      stats->parse_seconds = t.parse_s;
      stats->upload_seconds = g->call_upload_s;
      stats->render_seconds = t1 - t0;
      stats->annotated_ms = times.annotated_ms;
      stats->novel_ms = times.novel_ms;
      stats->around_ms = times.around_ms;
      stats->passes_ms = times.passes_ms;
      put_out(stats_out, st_local, "pantas_call_stats");
    }
  });
}


__attribute__((visibility("default"))) void pantas_synth_defaults(pantas_synth_params* out) {
  if (!out) return;
  SynthParams d;
  pantas_synth_params v{};
  pantas_synth_params* p = &v;
  p->seed = d.seed; p->n_genes = d.n_genes; p->n_chrom = d.n_chrom; p->n_lines = d.n_lines; p->base_id = d.base_id;
  p->bubble_rate = d.bubble_rate; p->max_exons = d.max_exons; p->intron_long_max = d.intron_long_max;
  p->intergenic_min = d.intergenic_min; p->intergenic_max = d.intergenic_max; p->read_len = d.read_len;
  p->with_path = d.with_path; p->zipf_a = d.zipf_a;
  p->p_lowmapq = d.p_lowmapq; p->p_unmapped = d.p_unmapped; p->p_sub = d.p_sub; p->p_indel = d.p_indel; p->p_highdv = d.p_highdv;
  p->p_novel_skip = d.p_novel_skip; p->p_novel_ce = d.p_novel_ce; p->p_intron = d.p_intron; p->p_novel_ss = d.p_novel_ss;
  p->p_clip = d.p_clip;
  p->line_offset = 0;
  p->gaf_only = d.gaf_only; p->hot_sites = d.hot_sites; p->p_del_hot = d.p_del_hot; p->p_two_indels = d.p_two_indels;
  (void)guarded([&] { put_out(out, v, "pantas_synth_params"); });   // a declaration without struct_size is left untouched
}

__attribute__((visibility("default"))) int pantas_synth_generate(const pantas_synth_params* p_in, pantas_synth** out,
                                                                  pantas_synth_stats* stats_out) {
  return guarded([&] {
    need(p_in && out, "null argument");
    const pantas_synth_params pv = get_in(p_in, "pantas_synth_params");
    const pantas_synth_params* p = &pv;
    SynthParams d;
    d.seed = p->seed; d.n_genes = p->n_genes; d.n_chrom = p->n_chrom ? p->n_chrom : 1; d.n_lines = p->n_lines; d.base_id = p->base_id;
    d.bubble_rate = p->bubble_rate; d.max_exons = p->max_exons; d.intron_long_max = p->intron_long_max;
    d.intergenic_min = p->intergenic_min; d.intergenic_max = p->intergenic_max; d.read_len = p->read_len;
    d.with_path = p->with_path; d.zipf_a = p->zipf_a;
    d.p_lowmapq = p->p_lowmapq; d.p_unmapped = p->p_unmapped; d.p_sub = p->p_sub; d.p_indel = p->p_indel; d.p_highdv = p->p_highdv;
    d.p_novel_skip = p->p_novel_skip; d.p_novel_ce = p->p_novel_ce; d.p_intron = p->p_intron; d.p_novel_ss = p->p_novel_ss;
    d.p_clip = p->p_clip;
    d.line_offset = p->line_offset;
    d.gaf_only = p->gaf_only; d.p_del_hot = p->p_del_hot; d.p_two_indels = p->p_two_indels;
    if (p->hot_sites) d.hot_sites = p->hot_sites;
    auto h = std::make_unique<pantas_synth>();
    double t0 = now_s();
    synth_generate(d, h->out);
    if (stats_out) {
      pantas_synth_stats st_local{};
      pantas_synth_stats* stats = &st_local;
      stats->nodes = h->out.nodes; stats->links = h->out.links; stats->junction_links = h->out.junction_links;
      stats->transcripts = h->out.transcripts; stats->gaf_lines = h->out.gaf_lines;
      stats->gfa_bytes = h->out.gfa.size(); stats->gtf_bytes = h->out.gtf.size(); stats->gaf_bytes = h->out.gaf.size();
      stats->seconds = now_s() - t0;
      put_out(stats_out, st_local, "pantas_synth_stats");
    }
    *out = h.release();
  });
}

__attribute__((visibility("default"))) int pantas_synth_text(const pantas_synth* s, int which, const char** data, uint64_t* size) {
  return guarded([&] {
    need(s && data && size && which >= 0 && which <= 2, "bad argument");
    *data = which == 0 ? s->out.gfa.data() : which == 1 ? s->out.gtf.data() : s->out.gaf.data();
    *size = which == 0 ? s->out.gfa.size() : which == 1 ? s->out.gtf.size() : s->out.gaf.size();
  });
}

__attribute__((visibility("default"))) int pantas_synth_write(const pantas_synth* s, const char* prefix, uint64_t max_gaf_lines) {
  return guarded([&] {
    need(s && prefix, "null argument");
    auto dump = [&](const char* ext, const char* p, size_t n) {
      std::string path = std::string(prefix) + ext;
      int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) throw Error("OSError", path + ": cannot create");
      try {
        while (n) {
          ssize_t w = ::write(fd, p, std::min<size_t>(n, size_t(1) << 30));
          if (w < 0) { if (errno == EINTR) continue; throw Error("OSError", path + ": write failed"); }
          p += w;
          n -= size_t(w);
        }
      } catch (...) { ::close(fd); throw; }
      ::close(fd);
    };
    dump(".gfa", s->out.gfa.data(), s->out.gfa.size());
    dump(".gtf", s->out.gtf.data(), s->out.gtf.size());
    size_t n = s->out.gaf.size();
    if (max_gaf_lines) {
      const char* p = s->out.gaf.data();
      const char* e = p + n;
      uint64_t k = 0;
      while (p < e && k < max_gaf_lines) {
        const void* nl = memchr(p, '\n', size_t(e - p));
        if (!nl) { p = e; break; }
        p = static_cast<const char*>(nl) + 1;
        ++k;
      }
      n = size_t(p - s->out.gaf.data());
    }
    dump(".gaf", s->out.gaf.data(), n);
  });
}

__attribute__((visibility("default"))) void pantas_synth_free(pantas_synth* s) { delete s; }

__attribute__((visibility("default"))) int pantas_graph_load_mem(const char* gfa_text, uint64_t size, pantas_graph** out) {
  return guarded([&] {
    need(gfa_text && out, "null argument");
    double t0 = now_s();
    auto h = std::make_unique<pantas_graph>();
    h->g = load_graph_mem(gfa_text, size_t(size));
    h->load_s = now_s() - t0;
    *out = h.release();
  });
}

__attribute__((visibility("default"))) int pantas_call_prepare_mem(pantas_graph* g, const char* gtf_text, uint64_t size, int device) {
  return guarded([&] {
    need(g && g->g && gtf_text, "null argument");
    auto t = std::make_unique<CallTables>();
    build_call_tables_mem(*g->g, gtf_text, size_t(size), *t);
    double t0 = now_s();
    g->call_dev = upload_call(*g->g, *t, device);
    g->call_upload_s = now_s() - t0;
    g->call_tables = std::move(t);
  });
}


__attribute__((visibility("default"))) int pantas_call_prepare_fused(pantas_graph* g, const char* gtf_path, const char* gtf_text,
                                                                      uint64_t size, int device) {
  return guarded([&] {
    need(g && g->g && (gtf_path || gtf_text), "null argument");
    need(g->have_result, "no finished augment run on this handle");
    auto t = std::make_unique<CallTables>();
    MappedFile f;
    if (gtf_text) f.attach(gtf_text, size_t(size)); else f.open(gtf_path);
    // a second replicate on this handle: the tables of the first hand their graph-and-GTF-only part over (and are gutted by it,
    // so they and their device copy go first)
    std::unique_ptr<CallTables> previous = std::move(g->call_tables);
    g->call_dev.reset();
    build_call_tables_mem(*g->g, f.data, f.size, *t, 0, &g->res, previous.get());
    previous.reset();
    double t0 = now_s();
    g->call_dev = upload_call(*g->g, *t, device);
    g->call_upload_s = now_s() - t0;
    g->call_tables = std::move(t);
  });
}

}  // extern "C"
