// Device-resident alignment records shared by augment.hip (consumer) and tokenize.hip (producer).
// HIP translation units only.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <vector>

#include "common.h"
#include "walk.h"

namespace pantas {

#define HIP_OK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      throw Error("HipError", std::string(#expr) + ": " + hipGetErrorString(e__));                     \
  } while (0)

// A non-blocking stream on `device`: one of those device start-up made while the process was small, else a new one.  (A stream made
// after a whole-genome graph was parsed -- 35 GB resident -- took 0.3 - 0.5 s on the GPU box, 9 ms in a fresh process: its queue is
// mapped against the process as it is then.)  augment.hip
hipStream_t take_stream(int device);

// identity of a resident batch for the rare-before-main protocol of a run (an address can be reused after a free)
inline uint64_t next_batch_id() {
  static std::atomic<uint64_t> n{1};
  return n.fetch_add(1);
}

struct DeviceBatch {
  uint64_t id = next_batch_id();
  LineHdr* hdr = nullptr;
  uint32_t *node_off = nullptr, *op_off = nullptr, *nodes = nullptr, *ops = nullptr, *del_idx = nullptr;
  uint4* rec = nullptr;          // derived per record, 48 bytes: {node_off, op_off, path nodes (8 bits, saturated) | cs ops << 8, id of
                                 // node 0} {ids of nodes 1 .. 4, clamped to the path} {the line's ShortOps (walk.h): total | stars << 16,
                                 // star positions, the indel} -- with hdr[] everything the short-walk kernel needs to start a record
                                 // AND its gather pipeline, in four 16-byte loads (instead of hdr, two offset pairs, nodes[0..2], then
                                 // ops[] and a scan of it: three dependent load levels and a loop less per record)
  // round 4, the short walk with one lane per path position (walk.h): the short-walk class laid out in tiles of 64 positions
  uint32_t* tile_first = nullptr;    // [n_tiles] first record | TILE_NOT_PLAIN
  uint32_t* pnodes = nullptr;        // [64 x n_tiles]
  uint16_t* pmeta = nullptr;         // [64 x n_tiles] walk.h pos_meta
  PosRec* prec = nullptr;            // [n_short] 32 bytes: line record + op summary
  uint32_t* chunk_tile = nullptr;    // [ceil(n_short / 2048) + 1]
  uint32_t n_tiles = 0;
  uint32_t n = 0, n_del = 0;
  uint32_t n_short = 0;          // records [0, n_short) are the short-walk class (walk.h), bucketed apart from the rest
  uint64_t n_nodes = 0, n_ops = 0, del_nodes = 0;
  uint64_t short_nodes = 0, short_ops = 0;   // path nodes / cs ops of the short-walk class
};

constexpr uint32_t REC_QUADS = 3;      // DeviceBatch::rec: three 16-byte words per record
constexpr uint32_t REC_OPS_MAX = SHORT_MAX_OPS - 1u;     // rec holds the op count in 24 bits (saturated; only short-walk records are read through it)

// rec[i] of a record whose k path nodes will stand at nodes[n0 ..] and whose c cs ops at ops[o0 ..] of the batch (see DeviceBatch::rec);
// `nodes` / `ops` point at the record's first node / first op, wherever they are read from
// (rec == nullptr: only the op summary is wanted -- a batch made for the position-parallel kernel keeps it in its line records and
// builds rec when a record-per-lane kernel is asked for)
__device__ __forceinline__ void make_rec_entry(uint4* rec, size_t i, uint32_t n0, uint32_t k, uint32_t o0, uint32_t c, const uint32_t* nodes,
                                               const uint32_t* ops, ShortOps* so_out = nullptr) {
  // the short walk's view of the cs (walk.h ShortOps), so that the dominant kernel neither reads ops[] nor scans it; only read for
  // records of the short-walk class (fewer than SHORT_MAX_OPS ops, total below 2^16)
  ShortOps so;
  if (c < SHORT_MAX_OPS) short_ops_scan(ops, c, so);
  if (so_out) *so_out = so;
  if (!rec) return;
  const uint32_t last = k ? k - 1u : 0u;
  auto id = [&](uint32_t j) { return k ? nodes[j < last ? j : last] : 0u; };
  rec[REC_QUADS * i] = make_uint4(n0, o0, (k < 255u ? k : 255u) | ((c < REC_OPS_MAX ? c : REC_OPS_MAX) << 8), id(0));
  rec[REC_QUADS * i + 1] = make_uint4(id(1), id(2), id(3), id(4));
  rec[REC_QUADS * i + 2] = make_uint4((so.T & 0xFFFFu) | (so.ns << 16), so.star01, so.star23, so.x);
}

// ---- tiles of the short-walk class (walk.h "one lane per path POSITION") ------------------------------------------------------
// place[r] of record r: tile within its chunk (bits 12 ..) | record within the tile (bits 6 .. 11) | first lane (bits 0 .. 5); the
// first record of a tile also holds the tile's lanes in use - 1 in bits 23 .. 28
constexpr uint32_t TILE_CHUNK = 2048;      // records per chunk of the aggregating kernels
__device__ __forceinline__ uint32_t place_tile(uint32_t pl) { return (pl >> 12) & 0x7FFu; }
__device__ __forceinline__ uint32_t place_ord(uint32_t pl) { return (pl >> 6) & 63u; }
__device__ __forceinline__ uint32_t place_lane(uint32_t pl) { return pl & 63u; }
__device__ __forceinline__ uint32_t place_used(uint32_t pl) { return ((pl >> 23) & 63u) + 1u; }
// what the first record of a tile adds to the tile: its word and the lanes behind the last record
__device__ __forceinline__ void tile_head(uint32_t tile, uint32_t rec_index, uint32_t used, uint32_t pad_node, uint32_t* tile_first, uint32_t* pnodes,
                                          uint16_t* pmeta) {
  atomicOr(&tile_first[tile], rec_index);                    // (the records of the tile OR TILE_NOT_PLAIN into the same word: zeroed beforehand)
  for (uint32_t l = used; l < 64u; ++l) { pnodes[size_t(tile) * 64u + l] = pad_node; pmeta[size_t(tile) * 64u + l] = uint16_t(POS_INACTIVE); }
}

// the GPU GFA loader parks its device memory instead of freeing it (tokenize.hip); no allocation may fail while blocks are parked
void release_parked_device_memory();
size_t parked_device_bytes();
// hipMalloc that gives parked blocks back and tries once more before it reports out-of-memory
hipError_t device_malloc_retry(void** p, size_t bytes);
template <class T> inline hipError_t dmalloc(T** p, size_t bytes) { return device_malloc_retry(reinterpret_cast<void**>(p), bytes); }

// Scratch memory of the GPU tokeniser, kept with the DeviceGraph between batches: hipMalloc / hipFree / hipHostMalloc
// cost milliseconds each, a batch needs about twenty temporaries.  take() hands out the smallest free block that
// fits (or allocates), give() returns it; everything is released with the graph.
struct ScratchPool {
  struct Block { void* p; size_t bytes; bool busy; };
  std::vector<Block> blocks;
  void* pinned[2] = {nullptr, nullptr};
  size_t pinned_bytes = 0;
  hipEvent_t pinned_done[2] = {nullptr, nullptr};
  void* take(size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    Block* best = nullptr;
    for (Block& b : blocks)
      if (!b.busy && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best && best->bytes <= 2 * bytes + (size_t(1) << 20)) { best->busy = true; return best->p; }
    void* p = nullptr;
    const size_t rounded = (bytes + 4095) / 4096 * 4096;
    if (hipMalloc(&p, rounded) != hipSuccess) {
      (void)hipGetLastError();
      trim();   // give cached blocks back -- and what the GPU GFA loader left parked (tokenize.hip) -- and try once more
      release_parked_device_memory();
      hipError_t e = hipMalloc(&p, rounded);
      if (e != hipSuccess) throw Error("HipError", std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    blocks.push_back(Block{p, rounded, true});
    return p;
  }
  void give(void* p) {
    for (Block& b : blocks) if (b.p == p) { b.busy = false; return; }
  }
  // ownership of a block leaves the pool (it becomes part of a DeviceBatch, freed with hipFree)
  void disown(void* p) {
    for (size_t i = 0; i < blocks.size(); ++i) if (blocks[i].p == p) { blocks.erase(blocks.begin() + long(i)); return; }
  }
  void trim() {
    for (size_t i = 0; i < blocks.size();) {
      if (!blocks[i].busy) { (void)hipFree(blocks[i].p); blocks.erase(blocks.begin() + long(i)); } else ++i;
    }
  }
  ~ScratchPool() {
    for (Block& b : blocks) (void)hipFree(b.p);
    for (int k = 0; k < 2; ++k) {
      if (pinned[k]) (void)hipHostFree(pinned[k]);
      if (pinned_done[k]) (void)hipEventDestroy(pinned_done[k]);
    }
  }
};

// Host array of gigabytes -> HBM: hipMemcpy from pageable memory stages through one runtime thread (~10 GB/s); this copies
// through two pinned 32 MB buffers that several host threads fill (the text upload of tokenize.hip reaches 50 GB/s that
// way).  Synchronous: returns when the bytes are on the device.  Small copies go straight through hipMemcpyAsync.
void h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t stream);
// the other way, into pageable host memory (several host threads copy one pinned buffer out while the next piece is on its way)
void d2h_staged(void* dst, const void* src, size_t bytes, hipStream_t stream);

// what tokenize.hip needs from a DeviceGraph without seeing its tables
struct DeviceGraphInfo {
  int device;
  hipStream_t stream;
  uint32_t V;
};
struct DeviceGraph;
DeviceGraphInfo graph_info(DeviceGraph& dg);
// what the default main kernel reads beside the batch's own arrays (augment.hip): the tiles of the position-parallel kernel, or db.rec
// for a record-per-lane default; both ways a batch comes to be call it last.  ensure_batch_rec: db.rec for a batch that has none (a
// record-per-lane kernel asked for by PANTAS_MAIN_KERNEL or taken because the batch is beyond the tiles' 32-bit offsets)
void build_batch_rec(DeviceGraph& dg, DeviceBatch& db);
void ensure_batch_rec(DeviceGraph& dg, DeviceBatch& db);
// the tiles of the short-walk class for the position-parallel kernel (augment.hip), from hdr / rec / nodes; after db.rec exists
void build_batch_tiles(DeviceGraph& dg, DeviceBatch& db);
// ... in two steps for a builder that writes the tile arrays itself (the GPU tokeniser's gather): place the records (db.chunk_tile and
// db.n_tiles are set, the tile arrays allocated, tile_first zeroed; returns the placement words, scratch of the pool: give it back)
struct ScratchPool;
uint32_t* place_batch_tiles(int device, hipStream_t stream, ScratchPool& pool, DeviceBatch& db, const uint32_t* node_off);
int main_kernel_variant();      // PANTAS_MAIN_KERNEL, default 120 (augment.hip)
struct Graph;
// dense name table (id - base -> node) in HBM, uploaded on first use and owned by the DeviceGraph
uint32_t* device_name_table(DeviceGraph& dg, const Graph& g);
// ... and its summary per block of NAME_BLOCK names (gaf_device.h DenseNames::blocks); valid after device_name_table
uint32_t* device_name_blocks(DeviceGraph& dg);
ScratchPool& scratch_pool(DeviceGraph& dg);

}  // namespace pantas
