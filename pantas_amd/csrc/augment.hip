// HIP kernels (gfx950) of the augment path and their host-side launchers.
//
// Kernels per batch of tokenised GAF records (reference scripts/alignments_augmentation_from_gaf.py:209-363):
//
//   augment_rare_kernel      only records whose cs holds a '-' op (a few %).  Records the IL/OL keys that come
//                            from deletions (key, count, first-seen order) in a device hash table and flags the
//                            node (two top bits of its node_len word, mirrored in g.info).
//   augment_main_agg_kernel  one thread per record, records bucketed by locus (gaf.cpp / tokenize.hip), in two
//                            instantiations side by side: the short-walk class (walk_short_info: ':' runs, a few '*',
//                            at most one '+' / '-'; one 8-byte gather per path node) and the general walk of walk.h
//                            for the rest.  NC, IL[0], OL[len] and RC increments are summed in workgroup-private LDS
//                            tables and reach HBM once per touched node per chunk of 2048 records (two 64-bit
//                            atomics on the node's 16-byte counter slot); links missing from the graph and the
//                            first-seen order of the common IL/OL key on flagged nodes are aggregated in LDS the
//                            same way before they go to the global tables.
//   augment_main_kernel      the general walk with direct global atomics (A/B reference, PANTAS_MAIN_KERNEL=0).
//
// The dense counters sit in ONE allocation [{NC, IL0, OLL, RC first slot} per node | RC per slot] so a multi-GPU run
// needs a single (block-sparse, exchange.hip) all-reduce.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <map>
#include <mutex>
#include <thread>

#include "augment.h"
#include "device_batch.h"
#include "walk.h"

namespace pantas {

constexpr uint64_t EMPTY_KEY = ~0ull;
constexpr uint32_t COMMON_POS = 0x7FFFFFFFu;

// One slot of a global hash table: key, first-seen order and count side by side.  The tables are sized for what a run might hold (a
// quarter of a batch's links: gigabytes) and hold a sliver of it, so every access is a miss in the caches and in the TLB -- one
// 32-byte sector per update here, three lines of three arrays in the planar form of rounds 1-3 (S4: 1.725 -> 1.69 ms).
struct __attribute__((aligned(32))) TableSlot {
  unsigned long long key;     // EMPTY_KEY: free
  unsigned long long first;   // min over the key's events of their order word
  uint32_t cnt;
  uint32_t pad[3];
};
static_assert(sizeof(TableSlot) == 32, "one sector");

struct TableView {
  TableSlot* slots;
  uint32_t mask;        // capacity - 1 (capacity is a power of two)
  uint32_t* n_keys;     // distinct keys inserted
  uint32_t* overflow;   // set when an insert found no free slot
};

// the two global hash tables and the status word (rare paths of the kernels)
struct SlowTables {
  TableView hist, novel;
  unsigned long long* status; // min over failing lines of (line_id << 8 | code)
  TildeCarry tilde;           // lengths that leak into lines whose first op is '~' (walk.h); empty unless the run found such lines
  // Round 5: one bit per block of TOUCHED_BLOCK_WORDS counters that an update has reached, or nullptr (a run on one GPU never
  // asks).  Read-sharded runs exchange only touched blocks (exchange.hip); with the bitmap kept here, by every site that adds to a
  // counter, the exchange no longer reads the 5 GB of counters to find them.
  uint32_t* touched;
};
constexpr uint32_t TOUCHED_BLOCK_WORDS = 8;    // the finest block of the exchange; coarser ones are ORs of neighbouring bits

struct GraphView {
  uint32_t* word;             // [V] segment length | FLAG_IL | FLAG_OL
  uint2* info;                // [V] {word, target of the node's first out-slot or NONE32}: all the short walk needs to know about a
                              // node in ONE 8-byte gather (instead of word, two row bounds and the dependent col[] load from three
                              // arrays); the flag bits of .x mirror word[]
  uint4* info3;               // [V] {word, targets of the node's first THREE out-slots, NONE32 beyond its degree} (walk.h NodeInfo3, round 3)
  const uint32_t* row_ptr;    // [V+1]
  const uint32_t* col;        // [A]
  uint32_t* cnt4;             // [4 V] per node {NC, IL[0], OL[len], RC of the node's first out-slot}: the counters one kept node
                              // bumps sit in ONE 16-byte slot -- at whole-genome scale every flushed counter is an HBM
                              // read-modify-write of its own cache line, and four planar arrays made that four lines per node
  uint32_t* rc;               // [A] RC per adjacency slot; a first slot's count is cnt4[4 n + 3] + rc[row_ptr[n]]
  const struct SlowTables* slow;   // what only rare paths touch, kept in memory: as kernel arguments these 24 words
                                   // stayed live in scalar registers across the walk loop and were spilled there
  uint32_t ablate;            // timing experiments only (PANTAS_ABLATE): 1 no flagged-key order, 2 no novel insert, 4 no link lookup, 8 no counter updates
};

struct BatchView {
  const LineHdr* hdr;
  const uint32_t* node_off;
  const uint32_t* op_off;
  const uint32_t* nodes;
  const uint32_t* ops;
  const uint32_t* del_idx;    // records holding a '-' op
  const uint4* rec;           // DeviceBatch::rec
  uint32_t n, n_del;
};

#define PANTAS_GLOBAL_AS __attribute__((address_space(1)))
// counter word `w` (index into [4 V | A]) has been added to: its block's bit in the run's bitmap, when one is kept.  The bit is
// looked at first -- an expressed gene's counters are reached by every chunk that holds its reads, and an atomic per visit on the
// same word would serialise -- and set by a fire-and-forget atomic.
typedef PANTAS_GLOBAL_AS uint32_t* TouchedPtr;
// the run's bitmap, or nullptr.  A flush loads it ONCE in front of its loops (a scalar load inside them would make every pass
// wait for the LDS reads in flight as well: lgkmcnt counts both); the rare paths inside the walk loops load it where they stand.
__device__ __forceinline__ TouchedPtr touched_of(const GraphView& g) { return (TouchedPtr)g.slow->touched; }
__device__ __forceinline__ void mark_touched(TouchedPtr t, uint64_t w) {
#ifdef PANTAS_NO_TOUCHED_MARKS      // (A/B builds only: what the marking costs the kernels of a run that does not ask for the bitmap)
  (void)t; (void)w;
  return;
#endif
  if (t == nullptr) return;
  const uint64_t blk = w / TOUCHED_BLOCK_WORDS;
  const uint32_t bit = 1u << (uint32_t(blk) & 31u);
  if (!(t[blk >> 5] & bit)) __hip_atomic_fetch_or(t + (blk >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t* nc_of(const GraphView& g, uint32_t n) { return g.cnt4 + 4 * size_t(n); }
__device__ __forceinline__ uint32_t* il0_of(const GraphView& g, uint32_t n) { return g.cnt4 + 4 * size_t(n) + 1; }
__device__ __forceinline__ uint32_t* oll_of(const GraphView& g, uint32_t n) { return g.cnt4 + 4 * size_t(n) + 2; }
__device__ __forceinline__ uint32_t* rc0_of(const GraphView& g, uint32_t n) { return g.cnt4 + 4 * size_t(n) + 3; }
// the four counters of node n as two 64-bit adds (counts stay below 2^31, so the low half never carries)
__device__ __forceinline__ void add_node4(const GraphView& g, uint32_t n, uint32_t a_nc, uint32_t a_il, uint32_t a_ol, uint32_t a_r0, TouchedPtr t) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(g.cnt4 + 4 * size_t(n));
  const unsigned long long lo = a_nc | (static_cast<unsigned long long>(a_il) << 32);
  const unsigned long long hi = a_ol | (static_cast<unsigned long long>(a_r0) << 32);
  if (lo) atomicAdd(p, lo);
  if (hi) atomicAdd(p + 1, hi);
  if (lo | hi) mark_touched(t, 4 * uint64_t(n));
}
// RC of adjacency slot s
__device__ __forceinline__ void add_rc(const GraphView& g, uint32_t s, uint32_t v, TouchedPtr t) {
  atomicAdd(&g.rc[s], v);
  mark_touched(t, uint64_t(g.rc - g.cnt4) + s);
}

// the graph as the short walks of walk.h see it: one 8-byte gather per node
struct DevInfo {
  const GraphView& g;
  __device__ __forceinline__ NodeInfo info(uint32_t n) const { return reinterpret_cast<const NodeInfo*>(g.info)[n]; }
  __device__ __forceinline__ NodeInfo3 info3(uint32_t n) const { return reinterpret_cast<const NodeInfo3*>(g.info3)[n]; }
};

__device__ __forceinline__ uint32_t mix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return uint32_t(k);
}

// The two global tables are reached through pointers loaded from memory (GraphView::slow), which the compiler can only
// treat as generic: it would emit flat_ instructions, and a flat access that MAY be outstanding makes every later
// s_waitcnt in the kernel a vmcnt(0) -- which undoes the software pipeline of the walk loops.  Casting to the global
// address space keeps them global_ instructions.
// (PANTAS_GLOBAL_AS is defined further up, in front of mark_touched)


// count += add, first = min(first, order) for `key`; inserts the key if absent.
__device__ __forceinline__ void table_update(const TableView& t, uint64_t key, uint32_t add, uint64_t order) {
  PANTAS_GLOBAL_AS TableSlot* slots = (PANTAS_GLOBAL_AS TableSlot*)t.slots;
  typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
  uint32_t h = mix64(key) & t.mask;
  for (uint32_t probe = 0; probe <= t.mask && probe < 8192u; ++probe) {
    // plain loads (an agent-scope atomic load is an sc1 load that goes past the caches): a stale EMPTY is settled by the
    // compare-and-swap, and a key never changes once it is set.  Key and first-seen order come in one 16-byte load.
    const ull2 kf = *reinterpret_cast<PANTAS_GLOBAL_AS const ull2*>(&slots[h]);
    unsigned long long cur = kf.x;
    if (cur == EMPTY_KEY) {
      unsigned long long expected = EMPTY_KEY;
      if (__hip_atomic_compare_exchange_strong(&slots[h].key, &expected, static_cast<unsigned long long>(key), __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_fetch_add((PANTAS_GLOBAL_AS uint32_t*)t.n_keys, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cur = key;
      } else {
        cur = expected;
      }
    }
    if (cur == key) {
      if (add) __hip_atomic_fetch_add(&slots[h].cnt, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // first-seen is a running minimum: once the early lines have landed nearly every later
      // candidate loses, and a (possibly stale, hence too large) read is enough to know it
      if (order < kf.y)
        __hip_atomic_fetch_min(&slots[h].first, static_cast<unsigned long long>(order), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    h = (h + 1) & t.mask;
  }
  __hip_atomic_exchange((PANTAS_GLOBAL_AS uint32_t*)t.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a line the walk refuses: min over failing lines of (line_id << 8 | code)
__device__ __forceinline__ void report_status(const GraphView& g, uint32_t line_id, int st) {
  __hip_atomic_fetch_min((PANTAS_GLOBAL_AS unsigned long long*)g.slow->status, (static_cast<unsigned long long>(line_id) << 8) | unsigned(st),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out-of-line copy for paths that almost never run (an LDS table is full): keeps their callers' loops small
__device__ __attribute__((noinline)) void table_update_cold(const TableView* t, uint64_t key, uint32_t add, uint64_t order) {
  table_update(*t, key, add, order);
}

__device__ __forceinline__ uint64_t hist_key(uint32_t node, uint32_t side, uint32_t pos_code) {
  return (uint64_t(node) << 32) | (uint64_t(side) << 31) | pos_code;
}

struct RareSink {
  const GraphView& g;
  __device__ uint32_t node_word(uint32_t n) const { return g.word[n]; }
  __device__ void nc(uint32_t) const {}
  __device__ void common(uint32_t, uint32_t, uint32_t, uint32_t, uint64_t) const {}
  __device__ void rare(uint32_t n, uint32_t side, int32_t pos, uint64_t order) const {
    table_update(g.slow->hist, hist_key(n, side, uint32_t(pos + 1) & 0x7FFFFFFFu), 1u, order);
    // set once: thousands of deletions land on the same node of an expressed gene, and atomics on one address serialise
    const uint32_t flag = side == SIDE_IL ? FLAG_IL : FLAG_OL;
    if (!(__builtin_nontemporal_load(&g.word[n]) & flag)) {
      atomicOr(&g.word[n], flag);
      atomicOr(&g.info[n].x, flag);
      atomicOr(&g.info3[n].x, flag);
    }
  }
  __device__ void edge(uint32_t, uint32_t, uint64_t) const {}
};

struct MainSink {
  const GraphView& g;
  __device__ uint32_t node_word(uint32_t n) const { return g.word[n]; }
  __device__ void nc(uint32_t n) const { atomicAdd(nc_of(g, n), 1u); mark_touched(touched_of(g), 4 * uint64_t(n)); }
  __device__ void common(uint32_t n, uint32_t word, uint32_t side, uint32_t times, uint64_t order) const {
    atomicAdd(side == SIDE_IL ? il0_of(g, n) : oll_of(g, n), times);
    mark_touched(touched_of(g), 4 * uint64_t(n));
    if (word & (side == SIDE_IL ? FLAG_IL : FLAG_OL)) table_update(g.slow->hist, hist_key(n, side, COMMON_POS), 0u, order);
  }
  __device__ void rare(uint32_t, uint32_t, int32_t, uint64_t) const {}
  __device__ void edge(uint32_t a, uint32_t b, uint64_t order) const {
    const uint32_t lo = g.row_ptr[a], hi = g.row_ptr[a + 1];
    for (uint32_t s = lo; s < hi; ++s) {
      if (g.col[s] == b) { add_rc(g, s, 1u, touched_of(g)); return; }
    }
    table_update(g.slow->novel, (uint64_t(a) << 32) | b, 1u, order);
  }
};

template <class Sink>
__device__ __forceinline__ void run_line(const GraphView& g, const BatchView& b, uint32_t i) {
  const LineHdr h = b.hdr[i];
  const uint32_t n0 = b.node_off[i], n1 = b.node_off[i + 1];
  const uint32_t o0 = b.op_off[i], o1 = b.op_off[i + 1];
  Sink sink{g};
  const int st = walk_line(h, b.nodes + n0, n1 - n0, b.ops + o0, o1 - o0, sink, &g.slow->tilde);
  if (st != WALK_OK) report_status(g, h.line_id, st);
}

__global__ __launch_bounds__(256) void augment_rare_kernel(GraphView g, BatchView b, uint32_t short_walk) {
  for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < b.n_del; t += gridDim.x * 256u) {
    const uint32_t i = b.del_idx[t];
    const LineHdr h = b.hdr[i];
    const uint32_t n0 = b.node_off[i], k = b.node_off[i + 1] - n0;
    if (short_walk && (h.flags & LINE_ONE_INDEL) && k != 0) {
      const uint32_t o0 = b.op_off[i];
      RareSink sink{g};
      const DevInfo ga{g};
      const int st = walk_short_rare(ga, h, b.nodes + n0, k, b.ops + o0, b.op_off[i + 1] - o0, sink);
      if (st != WALK_OK) report_status(g, h.line_id, st);
    } else {
      run_line<RareSink>(g, b, i);
    }
  }
}

// The rare-key pass with a workgroup-private stage.  On real data a true indel variant is ONE (node, side,
// position) key that every read covering it increments; with one global hash update per deletion those updates queue up on
// one address.  The records with a '-' are listed in batch order, i.e. by locus: a workgroup takes RCHUNK consecutive ones,
// sums equal keys in an LDS table (count, minimum first-seen order) and sends each distinct key to the global table once.
constexpr uint32_t RARE_THREADS = 256;
template <uint32_t RARE_LOG>
struct RareAggSink {
  const GraphView& g;
  unsigned long long* rkey; uint32_t* rcnt; unsigned long long* rmin;
  __device__ uint32_t node_word(uint32_t n) const { return g.word[n]; }
  __device__ void nc(uint32_t) const {}
  __device__ void common(uint32_t, uint32_t, uint32_t, uint32_t, uint64_t) const {}
  __device__ void edge(uint32_t, uint32_t, uint64_t) const {}
  __device__ void rare(uint32_t n, uint32_t side, int32_t pos, uint64_t order) const {
    constexpr uint32_t CAP = 1u << RARE_LOG;
    const unsigned long long key = hist_key(n, side, uint32_t(pos + 1) & 0x7FFFFFFFu);
    uint32_t h = mix64(key) & (CAP - 1);
    bool done = false;
#pragma unroll 1
    for (uint32_t p = 0; p < 8 && !done; ++p) {
      unsigned long long cur = __atomic_load_n(&rkey[h], __ATOMIC_RELAXED);
      if (cur == EMPTY_KEY) {
        cur = atomicCAS(&rkey[h], EMPTY_KEY, key);
        if (cur == EMPTY_KEY) cur = key;
      }
      if (cur == key) {
        atomicAdd(&rcnt[h], 1u);
        if (order < __atomic_load_n(&rmin[h], __ATOMIC_RELAXED)) atomicMin(&rmin[h], static_cast<unsigned long long>(order));
        done = true;
      }
      h = (h + 1) & (CAP - 1);
    }
    if (!done) table_update(g.slow->hist, key, 1u, order);
    // set once: thousands of deletions land on the same node of an expressed gene, and atomics on one address serialise
    const uint32_t flag = side == SIDE_IL ? FLAG_IL : FLAG_OL;
    if (!(__builtin_nontemporal_load(&g.word[n]) & flag)) {
      atomicOr(&g.word[n], flag);
      atomicOr(&g.info[n].x, flag);
      atomicOr(&g.info3[n].x, flag);
    }
  }
};

template <uint32_t RARE_CHUNK, uint32_t RARE_LOG>
__global__ __launch_bounds__(RARE_THREADS) void augment_rare_agg_kernel(GraphView g, BatchView b, uint32_t short_walk) {
  constexpr uint32_t CAP = 1u << RARE_LOG;
  __shared__ unsigned long long rkey[CAP];
  __shared__ unsigned long long rmin[CAP];
  __shared__ uint32_t rcnt[CAP];
  const uint32_t n_chunks = (b.n_del + RARE_CHUNK - 1) / RARE_CHUNK;
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    for (uint32_t i = threadIdx.x; i < CAP; i += RARE_THREADS) { rkey[i] = EMPTY_KEY; rmin[i] = ~0ull; rcnt[i] = 0u; }
    __syncthreads();
    const uint32_t lo = c * RARE_CHUNK, hi = min(b.n_del, lo + RARE_CHUNK);
    RareAggSink<RARE_LOG> sink{g, rkey, rcnt, rmin};
    for (uint32_t t = lo + threadIdx.x; t < hi; t += RARE_THREADS) {
      const uint32_t i = b.del_idx[t];
      const LineHdr h = b.hdr[i];
      const uint32_t n0 = b.node_off[i], k = b.node_off[i + 1] - n0;
      const uint32_t o0 = b.op_off[i], c_ops = b.op_off[i + 1] - o0;
      int st;
      if (short_walk && (h.flags & LINE_ONE_INDEL) && k != 0) {
        const DevInfo ga{g};
        st = walk_short_rare(ga, h, b.nodes + n0, k, b.ops + o0, c_ops, sink);
      } else {
        st = walk_line(h, b.nodes + n0, k, b.ops + o0, c_ops, sink, &g.slow->tilde);
      }
      if (st != WALK_OK) report_status(g, h.line_id, st);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CAP; i += RARE_THREADS)
      if (rkey[i] != EMPTY_KEY) table_update(g.slow->hist, rkey[i], rcnt[i], rmin[i]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void augment_main_kernel(GraphView g, BatchView b) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < b.n; i += gridDim.x * 256u)
    run_line<MainSink>(g, b, i);
}

constexpr uint32_t LDS_EMPTY = 0xFFFFFFFFu;

// ---- aggregating main kernel ------------------------------------------------------------------------
// Thread per record over fixed blocks of CHUNK bucketed records, workgroup-private aggregation in LDS:
//   node table   key = node            -> {NC, IL[0], OL[len]} increments (ONE probe per kept node)
//   link table   key = adjacency slot  -> RC increments
//   order table  key = node<<1 | side  -> min first-seen order of the common key on flagged nodes
// flushed as one global atomic per touched counter (one table update per flagged key) per chunk.
// Records whose cs holds only ':' and a few '*' ops (LINE_SIMPLE; the tokeniser puts them first in every
// block of 2048) take a short walk: such a node has either exactly one compacted ':' entry or, when every
// base it covers is a mismatch, none -- so no op list has to be built.  Both walks fetch the node's
// length word and its CSR row bounds together when the node is first touched; only the col[] scan of the
// link lookup depends on an earlier load.
// PACKED: the four counters of a node entry are 16-bit fields of ONE 64-bit LDS word, so that a kept node costs one
// ds_add_u64 instead of up to four ds_add_u32 (same-address LDS atomics serialise, and reads of one locus hit the
// same few nodes).  No field can carry into the next: a record adds at most 1 per field per visit (larger steps go to
// memory directly), only records with at most PACKED_MAX_PATH path nodes use the table, a chunk has 2048 records:
// 2048 * 31 < 65536.
constexpr uint32_t PACKED_MAX_PATH = SHORT_MAX_PATH;

// TRACK: the sink's counter updates mark the run's bitmap of touched blocks when one is kept (mark_touched).  false: no such code
// at all -- the instance the position-parallel kernel runs when nobody asked for the bitmap (a run on one GPU): even never-taken,
// the marking cost that kernel 1.7 % (profiles/r5_ab_s4_touched_marks.txt)
template <uint32_t LOG_N, uint32_t LOG_S, uint32_t LOG_F, bool PACKED = false, bool TRACK = true>
struct AggSink {
  static constexpr uint32_t CN = 1u << LOG_N, CS = 1u << LOG_S, CF = 1u << LOG_F, LOG_N_ = LOG_N;
  static constexpr bool TRACK_ = TRACK;
  const GraphView& g;
  __device__ __forceinline__ TouchedPtr touched() const { if constexpr (TRACK) return touched_of(g); else return nullptr; }
  uint32_t* nkey; uint32_t* ncnt;      // ncnt: [NC CN | IL0 CN | OLL CN | RC of the node's first out-slot CN]; PACKED: CN x u64 {NC, IL0, OLL, RC first slot}
  uint32_t* skey; uint32_t* scnt;
  uint32_t* fkey; unsigned long long* fmin;
  unsigned long long* vkey; uint32_t* vcnt; unsigned long long* vmin;   // links absent from the graph (CF entries)
  uint32_t* omin = nullptr;   // INFO kernels: [CN][2] min line id that bumped IL[0] / OL[len] of a flagged node, by node-table slot
  uint4* nvq = nullptr;       // NVQ kernels: links absent from the graph met by the short walk, queued for the end of the chunk
  uint32_t* nvq_n = nullptr;
  static constexpr uint32_t NVQ_CAP = 192;
  // the kernel over 16-byte node records (walk_short_w): hits on a node's second / third out-slot by node-table slot (low / high half),
  // and the workgroup's queue of links whose slot the walk could not tell (absent from the graph, a fourth or later slot)
  uint32_t* ncnt2 = nullptr;
  uint4* lq = nullptr;
  uint32_t* lq_n = nullptr;
  static constexpr uint32_t LQ_CAP = 512;
  uint32_t wab = 0;             // ablation bits of the ABL instance of augment_short_w_kernel (0 elsewhere: the tests fold away)

  // link (a, b) that is not in the graph: aggregate in LDS, else straight to the global table
  __device__ __forceinline__ void novel(uint32_t a, uint32_t b, uint64_t order) const {
    if (g.ablate & 2u) return;
    const unsigned long long key = (static_cast<unsigned long long>(a) << 32) | b;
    uint32_t h = mix64(key) & (CF - 1);
#pragma unroll 1
    for (uint32_t p = 0; p < 8; ++p) {
      unsigned long long cur = __atomic_load_n(&vkey[h], __ATOMIC_RELAXED);
      if (cur == EMPTY_KEY) {
        cur = atomicCAS(&vkey[h], EMPTY_KEY, key);
        if (cur == EMPTY_KEY) cur = key;
      }
      if (cur == key) {
        atomicAdd(&vcnt[h], 1u);
        if (order < __atomic_load_n(&vmin[h], __ATOMIC_RELAXED)) atomicMin(&vmin[h], static_cast<unsigned long long>(order));
        return;
      }
      h = (h + 1) & (CF - 1);
    }
    table_update(g.slow->novel, key, 1u, order);
  }
  // link (a -> b): the target of a's first slot was fetched when a was first touched; ha = a's node-table
  // slot if known.  A hit on the first slot (the usual n -> n+1 step) is counted in a's own entry.
  __device__ __forceinline__ void link_pf(uint32_t a, uint32_t ha, uint32_t lo, uint32_t hi, uint32_t col0, uint32_t b, uint64_t order) const {
    if (g.ablate & 4u) return;
    if (lo < hi) {
      if (col0 == b) {
        if (g.ablate & 8u) return;
        if (ha != LDS_EMPTY) atomicAdd(&ncnt[3u * CN + ha], 1u); else bump_slot(lo);
        return;
      }
      for (uint32_t s = lo + 1; s < hi; ++s) {
        if (g.col[s] == b) { bump_slot(s); return; }
      }
    }
    novel(a, b, order);
  }

  // HOME 1: the key's low bits as the home slot (the keys of a chunk are runs of consecutive ids: no two nodes of a run share a slot);
  // in the record-per-lane kernels it measured no faster than the multiplicative hash (2.427 vs 2.411 ms).
  // HOME 2: the table as buckets of four slots, a node's bucket from its low bits, linear probing from the bucket's first slot (the
  // position-parallel kernel reads the four keys of a bucket at once)
  template <uint32_t LOG, int HOME = 0>
  static __device__ __forceinline__ uint32_t probe(uint32_t* keys, uint32_t key) {
    constexpr uint32_t CAP = 1u << LOG;
    uint32_t h = HOME == 2 ? ((key << 2) & (CAP - 1u)) : HOME == 1 ? (key & (CAP - 1u)) : ((key * 2654435761u) >> (32 - LOG));
    // nearly every probe finds its key in the home slot (a chunk revisits the same few hundred nodes): one read, one compare
    if (__atomic_load_n(&keys[h], __ATOMIC_RELAXED) == key) return h;
#pragma unroll 1
    for (uint32_t p = 0; p < 10; ++p) {
      uint32_t cur = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
      if (cur == LDS_EMPTY) {
        cur = atomicCAS(&keys[h], LDS_EMPTY, key);
        if (cur == LDS_EMPTY) cur = key;
      }
      if (cur == key) return h;
      h = (h + 1) & (CAP - 1);
    }
    return LDS_EMPTY;
  }
  // PACKED: all four increments of a kept node (each 0 or 1; a_r0 = the link to the next node is the node's first
  // out-slot) in one LDS atomic
  __device__ __forceinline__ void bump_packed(uint32_t n, uint32_t row_lo, uint32_t a_nc, uint32_t a_il, uint32_t a_ol, uint32_t a_r0) const {
    if (g.ablate & 8u) return;
    const uint32_t h = probe<LOG_N>(nkey, n);
    if (h != LDS_EMPTY) {
      const unsigned long long inc = static_cast<unsigned long long>(a_nc) | (static_cast<unsigned long long>(a_il) << 16) |
                                     (static_cast<unsigned long long>(a_ol) << 32) | (static_cast<unsigned long long>(a_r0) << 48);
      atomicAdd(reinterpret_cast<unsigned long long*>(ncnt) + h, inc);
    } else {
      add_node4(g, n, a_nc, a_il, a_ol, a_r0, touched());
    }
  }
  // the same, plus the first-seen order of the common IL / OL key where the node is flagged: the line id alone decides
  // between lines, and a short-walk line has no other key on the node, so one ds_min on the node's own slot replaces a
  // probe of the order table per flagged side
  __device__ __forceinline__ void bump_packed_flagged(uint32_t n, uint32_t word, uint32_t line_id, uint32_t a_il, uint32_t a_ol, uint32_t a_r0,
                                                      uint64_t exact_order) const {
    // a_il / a_ol: 0 .. 3 steps (a node holding a '+' between ':' runs counts each compacted entry); walk.h bounds what a
    // line can add to one field so that 2048 lines stay below 2^16
    if (g.ablate & 8u) return;
    const bool f_il = a_il && (word & FLAG_IL), f_ol = a_ol && (word & FLAG_OL);
    const uint32_t h = probe<LOG_N>(nkey, n);
    if (h != LDS_EMPTY) {
      const unsigned long long inc = 1ull | (static_cast<unsigned long long>(a_il) << 16) |
                                     (static_cast<unsigned long long>(a_ol) << 32) | (static_cast<unsigned long long>(a_r0) << 48);
      if (!(g.ablate & 16u)) atomicAdd(reinterpret_cast<unsigned long long*>(ncnt) + h, inc);
      if (exact_order == 0) {
        // a running minimum: read first (lanes reading one address do not conflict, lanes doing an atomic on it serialise)
        if (f_il | f_ol) {
          // both minima of the slot in one 8-byte read (one region, one LDS round trip for a node flagged on both sides)
          const unsigned long long cur = __atomic_load_n(reinterpret_cast<const unsigned long long*>(omin + 2u * h), __ATOMIC_RELAXED);
          if (f_il && line_id < uint32_t(cur)) atomicMin(&omin[2u * h], line_id);
          if (f_ol && line_id < uint32_t(cur >> 32)) atomicMin(&omin[2u * h + 1u], line_id);
        }
      }
    } else {
      add_node4(g, n, 1u, a_il, a_ol, a_r0, touched());
    }
    // a line with a '-' op has rare keys of its own on the nodes it flags: its common key needs the exact within-line
    // order, through the order table of the general walk
    if (exact_order != 0 || h == LDS_EMPTY) {
      const uint64_t ord = exact_order != 0 ? exact_order : event_order(line_id, 0, 1);
      if (f_il) order_min(n, SIDE_IL, ord);
      if (f_ol) order_min(n, SIDE_OL, ord);
    }
  }
  // A link the graph does not have is met by about one line in thirty, so nearly every wave has a lane or two that need the
  // insertion into the novel-link table -- and a wave pays for that code (a 64-bit hash, a CAS loop, two atomics) whenever
  // ONE lane takes it.  With a queue the lane leaves {a, b, order} in LDS (one counter atomic, one 16-byte store) and the
  // workgroup does the insertions at the end of the chunk with all lanes busy.
  __device__ __forceinline__ void novel_later(uint32_t a, uint32_t b, uint64_t order) const {
    if (nvq) {
      const uint32_t i = atomicAdd(nvq_n, 1u);
      if (i < NVQ_CAP) { nvq[i] = make_uint4(a, b, uint32_t(order), uint32_t(order >> 32)); return; }
    }
    novel(a, b, order);
  }
  // link (a -> b) that is not a's first out-slot: the other slots, else a link the graph does not have
  __device__ __forceinline__ void link_rest(uint32_t a, uint32_t lo, uint32_t hi, uint32_t b, uint64_t order) const {
    if (g.ablate & 4u) return;
    for (uint32_t s = lo + 1; s < hi; ++s) {
      if (g.col[s] == b) { bump_slot(s); return; }
    }
    novel_later(a, b, order);
  }
  // returns the node's table slot (LDS_EMPTY when the increments went to memory)
  __device__ __forceinline__ uint32_t bump_node(uint32_t n, uint32_t a_nc, uint32_t a_il, uint32_t a_ol) const {
    if (g.ablate & 8u) return LDS_EMPTY;
    if constexpr (PACKED) {
      // general walk: steps above 1 bypass the 16-bit fields
      if (a_il > 1u) { atomicAdd(il0_of(g, n), a_il); a_il = 0; mark_touched(touched(), 4 * uint64_t(n)); }
      if (a_ol > 1u) { atomicAdd(oll_of(g, n), a_ol); a_ol = 0; mark_touched(touched(), 4 * uint64_t(n)); }
      if (a_nc | a_il | a_ol) bump_packed(n, 0u, a_nc, a_il, a_ol, 0u);
      return LDS_EMPTY;
    }
    const uint32_t h = probe<LOG_N>(nkey, n);
    if (h != LDS_EMPTY) {
      if (a_nc) atomicAdd(&ncnt[h], a_nc);
      if (a_il) atomicAdd(&ncnt[CN + h], a_il);
      if (a_ol) atomicAdd(&ncnt[2u * CN + h], a_ol);
    } else {
      add_node4(g, n, a_nc, a_il, a_ol, 0u, touched());
    }
    return h;
  }
  __device__ __forceinline__ void bump_slot(uint32_t s) const {
    if (g.ablate & 8u) return;
    const uint32_t h = probe<LOG_S>(skey, s);
    // two different instructions on purpose (scopes differ): written as atomicAdd on either pointer the compiler selects
    // the POINTER and emits one flat_atomic_add, and a flat access that may be outstanding turns the waits of the walk
    // loop into vmcnt(0)
    if (h != LDS_EMPTY) __hip_atomic_fetch_add(&scnt[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else { __hip_atomic_fetch_add((PANTAS_GLOBAL_AS uint32_t*)g.rc + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mark_touched(touched(), uint64_t(g.rc - g.cnt4) + s); }
  }
  __device__ __forceinline__ void order_min(uint32_t n, uint32_t side, uint64_t order) const {
    if (g.ablate & 1u) return;
    const uint32_t h = probe<LOG_F>(fkey, (n << 1) | side);
    if (h != LDS_EMPTY) {
      if (order < __atomic_load_n(&fmin[h], __ATOMIC_RELAXED)) atomicMin(&fmin[h], static_cast<unsigned long long>(order));
    } else {
      table_update(g.slow->hist, hist_key(n, side, COMMON_POS), 0u, order);
    }
  }
  // ---- walk_short_w ----
  // link (a -> b), slot unknown: the first slot of a's row that holds b, else a link the graph does not have.  Through the small slot
  // table: the reads of an expressed gene take the same fourth exit hundreds of times per chunk, and single global atomics on one
  // address serialise (measured: 6.6 ms instead of 2.0 with them going straight to memory)
  __device__ __forceinline__ void link_direct(uint32_t a, uint32_t b, uint64_t order) const {
    const uint32_t lo = g.row_ptr[a], hi = g.row_ptr[a + 1];
    for (uint32_t s = lo; s < hi; ++s) {
      if (g.col[s] == b) { bump_slot(s); return; }
    }
    novel(a, b, order);
  }
  __device__ __forceinline__ void link_later(uint32_t a, uint32_t b, uint64_t order) const {
    if (wab & 32u) return;
    const uint32_t i = __hip_atomic_fetch_add(lq_n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (i < LQ_CAP) lq[i] = make_uint4(a, b, uint32_t(order), uint32_t(order >> 32));
    else link_direct(a, b, order);
  }
  // one kept node: NC + 1, IL[0] / OL[len] + steps, one hit on out-slot code - 1 (code 1 .. 3), first-seen order of a flagged side
  template <int HOME = 0>
  __device__ __forceinline__ void bump_w(uint32_t n, uint32_t flags, uint32_t line_id, uint32_t a_il, uint32_t a_ol, uint32_t code, uint64_t exact_order) const {
    if (wab & 256u) { if ((n ^ flags ^ line_id ^ a_il ^ a_ol ^ code) == 0x12345677u) ncnt[0] = 1; return; }
    if (wab & 512u) { const uint32_t hh = probe<LOG_N>(nkey, n); if (hh != LDS_EMPTY) atomicAdd(reinterpret_cast<unsigned long long*>(ncnt) + hh, 1ull | (uint64_t(a_il) << 16) | (uint64_t(a_ol) << 32) | (uint64_t(code == 1u) << 48)); return; }
    // (most nodes carry no flag: one test in front of the two per side)
    bool f_il = false, f_ol = false;
    if (flags) { f_il = a_il && (flags & FLAG_IL); f_ol = a_ol && (flags & FLAG_OL); }
    const uint32_t h = probe<LOG_N, HOME>(nkey, n);
    if (h != LDS_EMPTY) {
      const unsigned long long inc = 1ull | (static_cast<unsigned long long>(a_il) << 16) | (static_cast<unsigned long long>(a_ol) << 32) |
                                     (static_cast<unsigned long long>(code == 1u ? 1u : 0u) << 48);
      atomicAdd(reinterpret_cast<unsigned long long*>(ncnt) + h, inc);
      if (code >= 2u) __hip_atomic_fetch_add(&ncnt2[h], code == 2u ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (exact_order == 0 && (f_il | f_ol)) {
        const unsigned long long cur = __atomic_load_n(reinterpret_cast<const unsigned long long*>(omin + 2u * h), __ATOMIC_RELAXED);
        if (f_il && line_id < uint32_t(cur)) atomicMin(&omin[2u * h], line_id);
        if (f_ol && line_id < uint32_t(cur >> 32)) atomicMin(&omin[2u * h + 1u], line_id);
      }
    } else {
      add_node4(g, n, 1u, a_il, a_ol, code == 1u ? 1u : 0u, touched());
      if (code >= 2u) { const uint32_t sl2 = g.row_ptr[n] + code - 1u; __hip_atomic_fetch_add((PANTAS_GLOBAL_AS uint32_t*)g.rc + sl2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mark_touched(touched(), uint64_t(g.rc - g.cnt4) + sl2); }
    }
    if (exact_order != 0 || h == LDS_EMPTY) {
      const uint64_t ord = exact_order != 0 ? exact_order : event_order(line_id, 0, 1);
      if (f_il) order_min(n, SIDE_IL, ord);
      if (f_ol) order_min(n, SIDE_OL, ord);
    }
  }
  // link (a -> b) given a's CSR row bounds
  __device__ __forceinline__ void link(uint32_t a, uint32_t lo, uint32_t hi, uint32_t b, uint64_t order) const {
    if (g.ablate & 4u) return;
    for (uint32_t s = lo; s < hi; ++s) {
      if (g.col[s] == b) { bump_slot(s); return; }
    }
    novel(a, b, order);
  }
  // ---- sink interface of walk_line / flush_kept (records that need the general walk)
  __device__ __forceinline__ uint32_t node_word(uint32_t n) const { return g.word[n]; }
  __device__ __forceinline__ void nc(uint32_t n) const { bump_node(n, 1u, 0u, 0u); }
  __device__ __forceinline__ void common(uint32_t n, uint32_t word, uint32_t side, uint32_t times, uint64_t order) const {
    bump_node(n, 0u, side == SIDE_IL ? times : 0u, side == SIDE_OL ? times : 0u);
    if (word & (side == SIDE_IL ? FLAG_IL : FLAG_OL)) order_min(n, side, order);
  }
  __device__ __forceinline__ void rare(uint32_t, uint32_t, int32_t, uint64_t) const {}
  __device__ __forceinline__ void edge(uint32_t a, uint32_t b, uint64_t order) const {
    link(a, g.row_ptr[a], g.row_ptr[a + 1], b, order);
  }
};

// The short walk itself lives in walk.h (walk_short: host + device source, so that the CPU harness runs the very same
// code against the goldens); here are its two device-side views: the graph as g.info gathers, the sink as the packed LDS
// entries of AggSink.
template <class Agg>
struct ShortSinkDev {
  const Agg& s;
  // deferred links (walk.h DEFER): up to four per line as {path index of a, of b, kept index}, 5 bits each, two per word,
  // in two LDS words of the thread (registers are what this kernel is short of: at 64 VGPRs two more spill)
  uint32_t* dq = nullptr;
  mutable uint32_t nq = 0;
  __device__ __forceinline__ void kept(uint32_t node, uint32_t word, uint32_t line_id, uint32_t a_il, uint32_t a_ol, uint32_t r0, uint64_t exact_order) const {
    s.bump_packed_flagged(node, word, line_id, a_il, a_ol, r0, exact_order);
  }
  __device__ __forceinline__ void link_rest(uint32_t a, uint32_t b, uint64_t order) const {
    s.link_rest(a, s.g.row_ptr[a], s.g.row_ptr[a + 1], b, order);
  }
  __device__ __forceinline__ void defer(uint32_t ia, uint32_t ib, uint32_t kidx, uint32_t a, uint32_t b, uint32_t line_id) const {
    if (nq < 4u) {
      const uint32_t e = ia | (ib << 5) | (kidx << 10);     // a short-walk line has at most 31 path nodes
      uint32_t* w = dq + (nq >> 1);
      *w = (nq & 1u) ? (*w | (e << 15)) : e;
      ++nq;
    } else {
      link_rest(a, b, event_order(line_id, kidx, 0));      // a fifth link off the first slots in one line: where it is met
    }
  }
  __device__ __forceinline__ void drain(const uint32_t* nodes, uint32_t line_id) const {
#pragma unroll 1
    for (uint32_t j = 0; j < nq; ++j) {
      const uint32_t e = (dq[j >> 1] >> ((j & 1u) * 15u)) & 0x7FFFu;
      const uint32_t a = nodes[e & 31u], b = nodes[(e >> 5) & 31u];
      link_rest(a, b, event_order(line_id, e >> 10, 0));
    }
  }
};
template <class Sink, int PF, bool DEFER = false, bool LEAN = false>
__device__ __forceinline__ int walk_short_info(const GraphView& g, const LineHdr& h, const uint32_t* nodes, uint32_t k,
                                               const uint32_t* ops, uint32_t nops, const Sink& sink, const uint32_t* first_ids, uint32_t* dq = nullptr,
                                               const ShortOps* pre = nullptr) {
  const DevInfo ga{g};
  const ShortSinkDev<Sink> ss{sink, dq};
  return walk_short<PF, DEFER, LEAN>(ga, h, nodes, k, ops, nops, ss, first_ids, pre);
}

template <uint32_t THREADS, uint32_t CHUNK, uint32_t LOG_N, uint32_t LOG_S, uint32_t LOG_F, uint32_t MINW = 1, bool PACKED = false, int ONLY = 0, bool ABL = false,
          bool INFO = false, int PF = 1, bool DEFER = false, bool NVQ = false, uint32_t XCD = 0, bool PRE = false, bool SNAKE = false>
__global__ __launch_bounds__(THREADS, MINW) void augment_main_agg_kernel(GraphView g_in, BatchView b, uint32_t begin, uint32_t end) {
  // the timing-only ablation switches (PANTAS_ABLATE) exist in the ABL instantiations only; elsewhere the tests fold away
  GraphView g = g_in;
  if constexpr (!ABL) g.ablate = 0u;
  using Sink = AggSink<LOG_N, LOG_S, LOG_F, PACKED>;
  constexpr uint32_t CN = Sink::CN, CS = Sink::CS, CF = Sink::CF;
  static_assert(!PACKED || (CHUNK * PACKED_MAX_PATH < 65536u && CHUNK * (SHORT_MAX_PATH_INDEL + 2u) < 65536u), "a 16-bit field could overflow");
  __shared__ unsigned long long fmin[CF];
  __shared__ unsigned long long vkey[CF];
  __shared__ unsigned long long vmin[CF];
  __shared__ uint32_t vcnt[CF];
  __shared__ uint32_t nkey[CN];
  __shared__ __attribute__((aligned(8))) uint32_t ncnt[(PACKED ? 2 : 4) * CN];
  __shared__ uint32_t skey[CS];
  __shared__ uint32_t scnt[CS];
  __shared__ uint32_t fkey[CF];
  __shared__ __attribute__((aligned(8))) uint32_t omin[INFO ? 2 * CN : 2];
  __shared__ uint32_t defer_q[DEFER ? 2 * THREADS : 2];
  __shared__ uint4 nvq[NVQ ? Sink::NVQ_CAP : 1];
  __shared__ uint32_t nvq_n;
  static_assert(!INFO || (PACKED && ONLY == 1), "the info walk exists for the short-walk class over packed entries");
  const uint32_t n_chunks = (end - begin + CHUNK - 1) / CHUNK;
  // XCD: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2): workgroup b takes
  // chunk (b % 8) * n/8 + b / 8, so that an XCD walks ONE contiguous eighth of the locus-ordered chunks -- the chunks in flight on it are
  // neighbours, and neighbours share graph nodes (an expressed gene's reads fill dozens of consecutive chunks)
  // XCD = 1: eighths; XCD = G > 1: runs of G consecutive chunks per XCD, the XCDs advancing through the loci together
  const uint32_t per_xcd = (n_chunks + 7u) / 8u;
  const uint32_t n_slots = XCD == 0 ? n_chunks : XCD == 1 ? 8u * per_xcd : (n_chunks + 8u * XCD - 1u) / (8u * XCD) * (8u * XCD);
  for (uint32_t c0 = blockIdx.x; c0 < n_slots; c0 += gridDim.x) {
    uint32_t c = c0;
    if constexpr (XCD == 1) c = (c0 & 7u) * per_xcd + (c0 >> 3);
    else if constexpr (XCD > 1) { const uint32_t j = c0 >> 3; c = (j / XCD) * (8u * XCD) + (c0 & 7u) * XCD + j % XCD; }
    if (XCD && c >= n_chunks) continue;
    for (uint32_t i = threadIdx.x; i < CN; i += THREADS) {
      if constexpr (INFO) { omin[i] = ~0u; omin[CN + i] = ~0u; }
      nkey[i] = LDS_EMPTY;
      ncnt[i] = 0u; ncnt[CN + i] = 0u;
      if constexpr (!PACKED) { ncnt[2 * CN + i] = 0u; ncnt[3 * CN + i] = 0u; }
    }
    for (uint32_t i = threadIdx.x; i < CS; i += THREADS) { skey[i] = LDS_EMPTY; scnt[i] = 0u; }
    for (uint32_t i = threadIdx.x; i < CF; i += THREADS) { fkey[i] = LDS_EMPTY; fmin[i] = ~0ull; vkey[i] = EMPTY_KEY; vmin[i] = ~0ull; vcnt[i] = 0u; }
    if (threadIdx.x == 0) nvq_n = 0u;
    __syncthreads();
    const Sink sink{g, nkey, ncnt, skey, scnt, fkey, fmin, vkey, vcnt, vmin, omin, NVQ ? nvq : nullptr, NVQ ? &nvq_n : nullptr};
    const uint32_t lo = begin + c * CHUNK;
    const uint32_t hi = min(end, lo + CHUNK);
    for (uint32_t r0 = lo, round = 0; r0 < hi; r0 += THREADS, ++round) {
      // a block is ordered by cost (plain lines by path length, then lines with a '*', then the one-indel lines): taken the same way round
      // in every round, the last wave of the workgroup would get the dearest 64 lines of every round -- odd rounds run the other way
      const uint32_t i = r0 + ((SNAKE && (round & 1u)) ? THREADS - 1u - threadIdx.x : threadIdx.x);
      if (i >= hi) continue;
      const LineHdr h = b.hdr[i];
      if constexpr (INFO) {
        // the record's offsets, sizes and first node id in one 16-byte load beside the header
        // (a line of the short-walk class has at most 31 path nodes and fewer than SHORT_MAX_OPS ops: both fit)
        const uint4 r = b.rec[REC_QUADS * size_t(i)], r2 = b.rec[REC_QUADS * size_t(i) + 1];
        const uint32_t first_ids[5] = {r.w, r2.x, r2.y, r2.z, r2.w};
        static_assert(PF <= 4, "rec holds the ids of the first five path nodes");
        int st;
        if constexpr (PRE) {
          const uint4 r3 = b.rec[REC_QUADS * size_t(i) + 2];
          ShortOps so;
          so.T = r3.x & 0xFFFFu; so.ns = r3.x >> 16; so.star01 = r3.y; so.star23 = r3.z; so.x = r3.w;
          st = walk_short_info<Sink, PF, DEFER>(g, h, b.nodes + r.x, r.z & 0xFFu, b.ops + r.y, r.z >> 8, sink, first_ids, defer_q + 2u * threadIdx.x, &so);
        } else
        st = walk_short_info<Sink, PF, DEFER>(g, h, b.nodes + r.x, r.z & 0xFFu, b.ops + r.y, r.z >> 8, sink, first_ids, defer_q + 2u * threadIdx.x);
        if (st != WALK_OK) report_status(g, h.line_id, st);
        continue;
      }
      const uint32_t n0 = b.node_off[i], n1 = b.node_off[i + 1];
      const uint32_t o0 = b.op_off[i], o1 = b.op_off[i + 1];
      int st;
      if (PACKED && n1 - n0 > PACKED_MAX_PATH) {
        // a path this long could overflow a 16-bit field of the LDS entries: straight to memory
        MainSink direct{g};
        st = walk_line(h, b.nodes + n0, n1 - n0, b.ops + o0, o1 - o0, direct, &g.slow->tilde);
      } else st = walk_line(h, b.nodes + n0, n1 - n0, b.ops + o0, o1 - o0, sink, &g.slow->tilde);
      if (st != WALK_OK) report_status(g, h.line_id, st);
    }
    __syncthreads();
    if constexpr (NVQ) {
      const uint32_t m = min(nvq_n, Sink::NVQ_CAP);
      for (uint32_t i = threadIdx.x; i < m; i += THREADS) {
        const uint4 e = nvq[i];
        sink.novel(e.x, e.y, (static_cast<uint64_t>(e.w) << 32) | e.z);
      }
      __syncthreads();
    }
    const TouchedPtr tb = touched_of(g);
    for (uint32_t i = threadIdx.x; i < CN; i += THREADS) {
      const uint32_t n = nkey[i];
      if (n == LDS_EMPTY) continue;
      uint32_t a, il, ol, r0;
      if constexpr (PACKED) {
        const unsigned long long v = reinterpret_cast<const unsigned long long*>(ncnt)[i];
        a = uint32_t(v) & 0xFFFFu; il = uint32_t(v >> 16) & 0xFFFFu; ol = uint32_t(v >> 32) & 0xFFFFu; r0 = uint32_t(v >> 48);
      } else {
        a = ncnt[i]; il = ncnt[CN + i]; ol = ncnt[2 * CN + i]; r0 = ncnt[3 * CN + i];
      }
      add_node4(g, n, a, il, ol, r0, tb);
      if constexpr (INFO) {
        if (omin[2 * i] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_IL, COMMON_POS), 0u, event_order(omin[2 * i], 0, 1));
        if (omin[2 * i + 1] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_OL, COMMON_POS), 0u, event_order(omin[2 * i + 1], 0, 1));
      }
    }
    for (uint32_t i = threadIdx.x; i < CS; i += THREADS) {
      const uint32_t s = skey[i];
      if (s != LDS_EMPTY && scnt[i]) add_rc(g, s, scnt[i], tb);
    }
    for (uint32_t i = threadIdx.x; i < CF; i += THREADS) {
      const uint32_t k = fkey[i];
      if (k != LDS_EMPTY && fmin[i] != ~0ull) table_update(g.slow->hist, hist_key(k >> 1, k & 1u, COMMON_POS), 0u, fmin[i]);
      if (vkey[i] != EMPTY_KEY) table_update(g.slow->novel, vkey[i], vcnt[i], vmin[i]);
    }
    __syncthreads();
  }
}

// ---- the short-walk class over 16-byte node records (round 3; walk.h walk_short_w) --------------------------------------
// Same chunking, tables and flush as augment_main_agg_kernel<..., short-walk class>; what differs is what a node visit does
// about its link: matched against the three targets that came with the node's record, a hit is one more field of the
// node's own increments -- no per-line list of deferred links, no lookup loop behind the node loop, no slot table.
template <class Agg>
struct ShortSinkW {
  const Agg& s;
  uint32_t line_id;
  __device__ __forceinline__ void kept(uint32_t node, uint32_t flags, uint32_t line_id_, uint32_t a_il, uint32_t a_ol, uint32_t code, uint64_t exact_order) const {
    s.bump_w(node, flags, line_id_, a_il, a_ol, code, exact_order);
  }
  // a link whose slot the walk could not tell (one link in fifty, one line in nine): into the workgroup's queue, looked up at the end
  // of the chunk.  Measured against this (S4, 2.07 ms): entries of path indices in four LDS slots of the thread's own, compacted at
  // the end of the chunk (no atomic inside the node loop, but two more dependent loads per lookup there: 2.15); the thread's first
  // in a slot of its own and only further ones through the counter (2.10); three per line in a register, queued behind the loop (2.56)
  __device__ __forceinline__ void later(uint32_t a, uint32_t b, uint32_t idx) const { s.link_later(a, b, event_order(line_id, idx, 0)); }
};

template <uint32_t THREADS, uint32_t CHUNK, uint32_t LOG_N, uint32_t LOG_F, uint32_t MINW, int PF, bool ABL = false>
__global__ __launch_bounds__(THREADS, MINW) void augment_short_w_kernel(GraphView g_in, BatchView b, uint32_t begin, uint32_t end) {
  GraphView g = g_in;
  // timing-only ablations (PANTAS_ABLATE, tools/w_ablate_pmc.sh; results wrong by construction) exist in the ABL instance only:
  // 32 nothing queued, 512 a kept node is one probe and one 64-bit atomic, 256 a kept node is nothing (32 + 256: the bare walk);
  // two A/B bits that leave the results right: 1024 every round of a chunk taken the same way round (2.13 against 2.09 ms at S4),
  // 2048 chunks in launch order instead of dealt to the XCDs in eighths (2.19)
  const uint32_t wab = ABL ? g.ablate : 0u;
  g.ablate = 0u;
  constexpr uint32_t LOG_S = 8;       // slot table of the queued links only
  using Sink = AggSink<LOG_N, LOG_S, LOG_F, true>;
  constexpr uint32_t CN = Sink::CN, CF = Sink::CF, CS = Sink::CS;
  __shared__ uint32_t skey[CS];
  __shared__ uint32_t scnt[CS];
  static_assert(CHUNK * PACKED_MAX_PATH < 65536u && CHUNK * (SHORT_MAX_PATH_INDEL + 2u) < 65536u, "a 16-bit field could overflow");
  static_assert(PF <= 4, "rec holds the ids of the first five path nodes");
  __shared__ unsigned long long fmin[CF];
  __shared__ unsigned long long vkey[CF];
  __shared__ unsigned long long vmin[CF];
  __shared__ uint32_t vcnt[CF];
  __shared__ uint32_t fkey[CF];
  __shared__ uint32_t nkey[CN];
  __shared__ __attribute__((aligned(8))) uint32_t ncnt[2 * CN];
  __shared__ uint32_t ncnt2[CN];
  __shared__ __attribute__((aligned(8))) uint32_t omin[2 * CN];
  __shared__ uint4 lq[Sink::LQ_CAP];
  __shared__ uint32_t lq_n;
  const uint32_t n_chunks = (end - begin + CHUNK - 1) / CHUNK;
  const uint32_t per_xcd = (n_chunks + 7u) / 8u;       // workgroup w takes chunk (w % 8) * n/8 + w / 8: an XCD walks one contiguous eighth
  for (uint32_t c0 = blockIdx.x; c0 < 8u * per_xcd; c0 += gridDim.x) {
    uint32_t c = (c0 & 7u) * per_xcd + (c0 >> 3);
    if (ABL && (wab & 2048u)) c = c0;                      // (A/B: chunks in launch order)
    if (c >= n_chunks) continue;
    for (uint32_t i = threadIdx.x; i < CN; i += THREADS) {
      omin[i] = ~0u; omin[CN + i] = ~0u;
      nkey[i] = LDS_EMPTY;
      ncnt[i] = 0u; ncnt[CN + i] = 0u; ncnt2[i] = 0u;
    }
    for (uint32_t i = threadIdx.x; i < CF; i += THREADS) { fkey[i] = LDS_EMPTY; fmin[i] = ~0ull; vkey[i] = EMPTY_KEY; vmin[i] = ~0ull; vcnt[i] = 0u; }
    for (uint32_t i = threadIdx.x; i < CS; i += THREADS) { skey[i] = LDS_EMPTY; scnt[i] = 0u; }
    if (threadIdx.x == 0) lq_n = 0u;
    __syncthreads();
    Sink sink{g, nkey, ncnt, skey, scnt, fkey, fmin, vkey, vcnt, vmin, omin, nullptr, nullptr};
    sink.ncnt2 = ncnt2; sink.lq = lq; sink.lq_n = &lq_n; sink.wab = wab;
    const DevInfo ga{g};
    const uint32_t lo = begin + c * CHUNK;
    const uint32_t hi = min(end, lo + CHUNK);
    for (uint32_t r0 = lo, round = 0; r0 < hi; r0 += THREADS, ++round) {
      const uint32_t i = r0 + (((round & 1u) && !(ABL && (wab & 1024u))) ? THREADS - 1u - threadIdx.x : threadIdx.x);     // odd rounds the other way round (see above; A/B bit 1024: not)
      if (i >= hi) continue;
      const LineHdr h = b.hdr[i];
      const uint4 r = b.rec[REC_QUADS * size_t(i)], r2 = b.rec[REC_QUADS * size_t(i) + 1], r3 = b.rec[REC_QUADS * size_t(i) + 2];
      const uint32_t first_ids[5] = {r.w, r2.x, r2.y, r2.z, r2.w};
      ShortOps so;
      so.T = r3.x & 0xFFFFu; so.ns = r3.x >> 16; so.star01 = r3.y; so.star23 = r3.z; so.x = r3.w;
      const ShortSinkW<Sink> ss{sink, h.line_id};
      const int st = walk_short_w<PF>(ga, h, b.nodes + r.x, r.z & 0xFFu, ss, first_ids, so);
      if (st != WALK_OK) report_status(g, h.line_id, st);
    }
    __syncthreads();
    {
      const uint32_t m = min(lq_n, Sink::LQ_CAP);
      for (uint32_t i = threadIdx.x; i < m; i += THREADS) {
        const uint4 e = lq[i];
        sink.link_direct(e.x, e.y, (static_cast<uint64_t>(e.w) << 32) | e.z);
      }
      __syncthreads();
    }
    const TouchedPtr tb = touched_of(g);
    for (uint32_t i = threadIdx.x; i < CN; i += THREADS) {
      const uint32_t n = nkey[i];
      if (n == LDS_EMPTY) continue;
      const unsigned long long v = reinterpret_cast<const unsigned long long*>(ncnt)[i];
      add_node4(g, n, uint32_t(v) & 0xFFFFu, uint32_t(v >> 16) & 0xFFFFu, uint32_t(v >> 32) & 0xFFFFu, uint32_t(v >> 48), tb);
      const uint32_t v2 = ncnt2[i];
      if (v2) {
        const uint32_t row = g.row_ptr[n];
        if (v2 & 0xFFFFu) add_rc(g, row + 1u, v2 & 0xFFFFu, tb);
        if (v2 >> 16) add_rc(g, row + 2u, v2 >> 16, tb);
      }
      if (omin[2 * i] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_IL, COMMON_POS), 0u, event_order(omin[2 * i], 0, 1));
      if (omin[2 * i + 1] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_OL, COMMON_POS), 0u, event_order(omin[2 * i + 1], 0, 1));
    }
    for (uint32_t i = threadIdx.x; i < CS; i += THREADS) {
      const uint32_t sl = skey[i];
      if (sl != LDS_EMPTY && scnt[i]) add_rc(g, sl, scnt[i], tb);
    }
    for (uint32_t i = threadIdx.x; i < CF; i += THREADS) {
      const uint32_t k = fkey[i];
      if (k != LDS_EMPTY && fmin[i] != ~0ull) table_update(g.slow->hist, hist_key(k >> 1, k & 1u, COMMON_POS), 0u, fmin[i]);
      if (vkey[i] != EMPTY_KEY) table_update(g.slow->novel, vkey[i], vcnt[i], vmin[i]);
    }
    __syncthreads();
  }
}

// ---- the short walk with one lane per path position (round 4; walk.h "one lane per path POSITION") --------------------------
// inclusive sum over the wave's 64 lanes (DPP: four row shifts, two row broadcasts)
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, false));      // row_shr:1
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, false));      // row_shr:2
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, false));      // row_shr:4
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, false));      // row_shr:8
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
  v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t lane_read(uint32_t v, uint32_t from_lane) {
  return uint32_t(__builtin_amdgcn_ds_bpermute(int(from_lane << 2), int(v)));
}

template <class Sink>
__device__ __forceinline__ void chunk_tables_clear(const Sink& k, uint32_t threads) {
  constexpr uint32_t CN = Sink::CN, CF = Sink::CF, CS = Sink::CS;
  for (uint32_t i = threadIdx.x; i < CN; i += threads) {
    k.omin[i] = ~0u; k.omin[CN + i] = ~0u;
    k.nkey[i] = LDS_EMPTY;
    k.ncnt[i] = 0u; k.ncnt[CN + i] = 0u; k.ncnt2[i] = 0u;
  }
  for (uint32_t i = threadIdx.x; i < CF; i += threads) { k.fkey[i] = LDS_EMPTY; k.fmin[i] = ~0ull; k.vkey[i] = EMPTY_KEY; k.vmin[i] = ~0ull; k.vcnt[i] = 0u; }
  for (uint32_t i = threadIdx.x; i < CS; i += threads) { k.skey[i] = LDS_EMPTY; k.scnt[i] = 0u; }
  if (threadIdx.x < 8u) k.lq_n[threadIdx.x] = 0u;      // (the position-parallel kernel: one count per wave)
}
// queued links looked up, then every table of the chunk added to memory (the two barriers inside; none behind)
template <class Sink>
__device__ __forceinline__ void chunk_tables_flush(const GraphView& g, const Sink& k, uint32_t threads) {
  constexpr uint32_t CN = Sink::CN, CF = Sink::CF, CS = Sink::CS;
  __syncthreads();
  {
    // the queue is one region per wave (LQ_CAP / waves entries each, filled without atomics): entry i belongs to wave i / region
    const uint32_t region = Sink::LQ_CAP / (threads / 64u);
    for (uint32_t i = threadIdx.x; i < Sink::LQ_CAP; i += threads) {
      if ((i % region) < k.lq_n[i / region]) {
        const uint4 e = k.lq[i];
        k.link_direct(e.x, e.y, (static_cast<uint64_t>(e.w) << 32) | e.z);
      }
    }
    __syncthreads();
  }
  const TouchedPtr tb = Sink::TRACK_ ? touched_of(g) : nullptr;
  for (uint32_t i = threadIdx.x; i < CN; i += threads) {
    const uint32_t n = k.nkey[i];
    if (n == LDS_EMPTY) continue;
    const unsigned long long v = reinterpret_cast<const unsigned long long*>(k.ncnt)[i];
    add_node4(g, n, uint32_t(v) & 0xFFFFu, uint32_t(v >> 16) & 0xFFFFu, uint32_t(v >> 32) & 0xFFFFu, uint32_t(v >> 48), tb);
    const uint32_t v2 = k.ncnt2[i];
    if (v2) {
      const uint32_t row = g.row_ptr[n];
      if (v2 & 0xFFFFu) add_rc(g, row + 1u, v2 & 0xFFFFu, tb);
      if (v2 >> 16) add_rc(g, row + 2u, v2 >> 16, tb);
    }
    if (k.omin[2 * i] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_IL, COMMON_POS), 0u, event_order(k.omin[2 * i], 0, 1));
    if (k.omin[2 * i + 1] != ~0u) table_update(g.slow->hist, hist_key(n, SIDE_OL, COMMON_POS), 0u, event_order(k.omin[2 * i + 1], 0, 1));
  }
  for (uint32_t i = threadIdx.x; i < CS; i += threads) {
    const uint32_t sl = k.skey[i];
    if (sl != LDS_EMPTY && k.scnt[i]) add_rc(g, sl, k.scnt[i], tb);
  }
  for (uint32_t i = threadIdx.x; i < CF; i += threads) {
    const uint32_t kk = k.fkey[i];
    if (kk != LDS_EMPTY && k.fmin[i] != ~0ull) table_update(g.slow->hist, hist_key(kk >> 1, kk & 1u, COMMON_POS), 0u, k.fmin[i]);
    if (k.vkey[i] != EMPTY_KEY) table_update(g.slow->novel, k.vkey[i], k.vcnt[i], k.vmin[i]);
  }
}

// OPT (timing-only instances, compiled with EXPERIMENTS=1; results wrong by construction): 16 = a kept position updates nothing,
// 32 = nothing is queued, 64 = nothing happens behind the mask of kept positions
template <uint32_t THREADS, uint32_t CHUNK, uint32_t LOG_N, uint32_t LOG_F, uint32_t MINW, uint32_t OPT>
__global__ __launch_bounds__(THREADS, MINW) void augment_short_p_kernel(GraphView g, BatchView b, const uint32_t* __restrict__ t_first,
                                                                        const uint32_t* __restrict__ t_pnodes, const uint16_t* __restrict__ t_pmeta,
                                                                        const PosRec* __restrict__ t_prec, const uint32_t* __restrict__ t_chunk_tile,
                                                                        uint32_t n_chunks) {
  // (the tile arrays are restrict pointers of their own: nothing the kernel writes aliases them, so the wave-uniform words are read
  // with scalar loads)
  constexpr uint32_t LOG_S = 8;
  using Sink = AggSink<LOG_N, LOG_S, LOG_F, true, (OPT & 2048u) != 0u>;       // OPT 2048: the instance that keeps the bitmap of touched blocks
  constexpr uint32_t CN = Sink::CN, CF = Sink::CF, CS = Sink::CS, NW = THREADS / 64u;
  static_assert(CHUNK * PACKED_MAX_PATH < 65536u && CHUNK * (SHORT_MAX_PATH_INDEL + 2u) < 65536u, "a 16-bit field could overflow");
  __shared__ uint32_t skey[CS];
  __shared__ uint32_t scnt[CS];
  __shared__ unsigned long long fmin[CF];
  __shared__ unsigned long long vkey[CF];
  __shared__ unsigned long long vmin[CF];
  __shared__ uint32_t vcnt[CF];
  __shared__ uint32_t fkey[CF];
  __shared__ __attribute__((aligned(16))) uint32_t nkey[CN];
  __shared__ __attribute__((aligned(8))) uint32_t ncnt[2 * CN];
  __shared__ uint32_t ncnt2[CN];
  __shared__ __attribute__((aligned(8))) uint32_t omin[2 * CN];
  __shared__ uint4 lq[Sink::LQ_CAP];
  __shared__ uint32_t lq_n[8];
  g.ablate = 0u;
  const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t per_xcd = (n_chunks + 7u) / 8u;       // workgroup w takes chunk (w % 8) * n/8 + w / 8: an XCD walks one contiguous eighth
  const NodeInfo3* info3 = reinterpret_cast<const NodeInfo3*>(g.info3);
  for (uint32_t c0 = blockIdx.x; c0 < 8u * per_xcd; c0 += gridDim.x) {
    const uint32_t c = (c0 & 7u) * per_xcd + (c0 >> 3);
    if (c >= n_chunks) continue;
    Sink sink{g, nkey, ncnt, skey, scnt, fkey, fmin, vkey, vcnt, vmin, omin, nullptr, nullptr};
    sink.ncnt2 = ncnt2; sink.lq = lq; sink.lq_n = lq_n;
    chunk_tables_clear(sink, THREADS);
    const uint32_t t_begin = t_chunk_tile[c], t_end = t_chunk_tile[c + 1];      // (t_end > t_begin: a chunk has records)
    const uint32_t t_last = t_end - 1u;
    // Pipeline over the wave's tiles t, t + NW, ...: while tile t is worked on, {first-record word, node ids, position words} of tile
    // t + 2 NW and {node records, line records, op summaries} of tile t + NW are in flight.  Every load is issued unconditionally
    // (tile index clamped to the chunk): a load that may or may not have been issued turns every later wait into a full one.
    uint32_t t = t_begin + wave;
    uint32_t queued = 0u;                                                 // links this wave has queued in its region (wave-uniform)
    uint32_t refused = 0xFFFFFFFFu;                                       // the lowest line of this lane's that the walk refuses
    constexpr uint32_t REGION = Sink::LQ_CAP / NW;
    static_assert(REGION >= 64u, "a tile can queue 64 links at once");
    auto drain_region = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");                // (the wave's own stores to its region are done)
      for (uint32_t j = lane; j < queued; j += 64u) {
        const uint4 e = lq[wave * REGION + j];
        const uint64_t order = (static_cast<uint64_t>(e.w) << 32) | (e.z & ~1u);
        // bit 0 of the entry's order word (its phase bits are 0): the node has at most two out-slots, both looked at where the link
        // was met -- the graph does not have the link, no row has to be read (most queued links: 1.9 % of all against 0.4 % that
        // take a fourth or later slot)
        if (e.z & 1u) sink.novel(e.x, e.y, order); else sink.link_direct(e.x, e.y, order);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    };
    auto line_of = [&](uint32_t first_word, uint32_t meta) {      // a lane behind the last record reads the tile's first line (and drops it)
      return (first_word & ~TILE_NOT_PLAIN) + (meta == POS_INACTIVE ? 0u : pos_ordinal(meta));
    };
    const uint32_t ta = min(t, t_last), tb = min(t + NW, t_last);
    uint32_t first_c = t_first[ta], first_n = t_first[tb];
    // (32-bit byte offsets on the arrays' scalar bases: the host has checked that tiles, line records and op summaries end below 4 GB)
    auto node_at = [&](uint32_t tt) { return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(t_pnodes) + ((tt << 8) + (lane << 2))); };
    auto meta_at = [&](uint32_t tt) { return uint32_t(*reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(t_pmeta) + ((tt << 7) + (lane << 1)))); };
    // (the two halves of the line's 32-byte record: one sector)
    auto line_at = [&](uint32_t rec) { return *reinterpret_cast<const PosLine*>(reinterpret_cast<const char*>(t_prec) + (rec << 5)); };
    auto ops_at = [&](uint32_t rec) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(t_prec) + ((rec << 5) + 16u)); };
    uint32_t node_c = node_at(ta), node_n = node_at(tb);
    uint32_t meta_c = meta_at(ta), meta_n = meta_at(tb);
    NodeInfo3 inf_c = info3[node_c];
    uint32_t rec_c = line_of(first_c, meta_c);
    PosLine pr_c = line_at(rec_c);
    uint4 so_c = make_uint4(0u, 0u, 0u, 0u);
    if (first_c & TILE_NOT_PLAIN) so_c = ops_at(rec_c);
    __syncthreads();
    [[maybe_unused]] unsigned long long st_last = 0, st_acc[6] = {0, 0, 0, 0, 0, 0};
    // (timing-only instance 130: shader-clock stamps between the parts of a tile's work, summed per wave; shares, not durations)
#define PANTAS_STAMP(K, DEP) do { if constexpr ((OPT & 1024u) != 0u) { asm volatile("" :: "v"(DEP)); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                                  st_acc[K] += now_ - st_last; st_last = now_; } } while (0)
    if constexpr ((OPT & 1024u) != 0u) st_last = __builtin_amdgcn_s_memtime();
    for (; t < t_end; t += NW) {
      const uint32_t first_word = first_c, node = node_c, meta = meta_c;
      const NodeInfo3 inf = inf_c;
      const PosLine pr = pr_c;
      const uint4 r3 = so_c;
      {
        const uint32_t t2 = min(t + 2u * NW, t_last);
        const uint32_t first_nn = t_first[t2];
        const uint32_t node_nn = node_at(t2);
        const uint32_t meta_nn = meta_at(t2);
        inf_c = info3[node_n];
        rec_c = line_of(first_n, meta_n);
        pr_c = line_at(rec_c);
        if (first_n & TILE_NOT_PLAIN) so_c = ops_at(rec_c);      // (wave-uniform)
        first_c = first_n; first_n = first_nn;
        node_c = node_n; node_n = node_nn;
        meta_c = meta_n; meta_n = meta_nn;
      }
      if constexpr ((OPT & 384u) != 0u) {     // (timing-only instances: 128 = 32 more vector instructions per tile, 256 = 32 more scalar ones)
        uint32_t x = node, y = meta;
        int sx = int(t);
        if constexpr ((OPT & 128u) != 0u) {
#pragma unroll
          for (int u = 0; u < 16; ++u) asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 3" : "+v"(x), "+v"(y));
        }
        if constexpr ((OPT & 256u) != 0u) {
#pragma unroll
          for (int u = 0; u < 32; ++u) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sx) : : "scc");
        }
        if ((x ^ y ^ uint32_t(sx)) == 0x12345677u) ncnt[0] = 1u;
      }
      // ---- the position's room and what stands in front of it in its record
      const uint32_t i = meta & 31u, rem = (meta >> 5) & 31u;
      const bool first = i == 0u, last = rem == 0u;
      const int32_t r = int32_t(inf.word & LEN_MASK) - (first ? pr.start : 0) - (last ? pr.tail - 1 : 0);
      const int32_t rc = r < int32_t(TILE_ROOM_CAP) ? r : int32_t(TILE_ROOM_CAP);
      const uint32_t room = (meta != POS_INACTIVE && rc > 0) ? uint32_t(rc) : 0u;
      PANTAS_STAMP(0, room);            // requests issued, this tile's node and line records arrived, room computed
      const uint32_t incl = wave_inclusive_sum(room);
      const uint32_t base = lane_read(incl, lane - i - 1u);               // the sum up to the lane in front of the record (not used at lane 0)
      const uint32_t in_front = incl - room - (i == lane ? 0u : base);
      PANTAS_STAMP(1, in_front);        // scan + lane read
      const uint32_t T = pr.tw & 0xFFFFu;
      bool keep = room != 0u && in_front < T;
      uint32_t plain = 1u;
      // a node with room behind the line's last op: the line is refused.  Remembered here and reported behind the chunk's tiles: a
      // report needs g.slow->status, and with the report in this loop the compiler loads that pointer in EVERY pass and waits for
      // it with vmcnt(0) -- the wait that also covers the requests just made for the next tiles (found by the stamps of instance
      // 130: 59 % of a wave's clocks stood there)
      if (room != 0u && in_front >= T) refused = min(refused, pr.line_id);
      if (first_word & TILE_NOT_PLAIN) {                                  // (wave-uniform) a '*' or an indel somewhere in the tile
        if (keep && (pr.tw & POS_NOT_PLAIN)) {
          ShortOps so;
          so.T = r3.x & 0xFFFFu; so.ns = r3.x >> 16; so.star01 = r3.y; so.star23 = r3.z; so.x = r3.w;
          const uint32_t left = T - in_front, le = room < left ? room : left, end = in_front + le;
          const bool indel_here = so.x_len() != 0u && so.x_pos() < end && so.x_pos() + so.x_len() > in_front;
          if (indel_here || le <= so.n_stars()) short_node_summary(so, in_front, le, keep, plain);
        }
      }
      // ---- first / last kept position of the record, the neighbour kept position
      const unsigned long long kept = __ballot(keep);
      const bool rev = (pr.tw & POS_REV) != 0u;
      bool is_first = first, is_last = last;
      uint32_t nb_lane = rev ? lane - 1u : lane + 1u, q = i;
      uint32_t tnode;
      if (kept == __ballot(meta != POS_INACTIVE)) {
        // (wave-uniform) every position of the tile is kept -- nearly every tile: the neighbours are the lanes next door
        const uint32_t prev = uint32_t(__builtin_amdgcn_update_dpp(0, int(node), 0x138, 0xf, 0xf, false));      // wave_shr:1: from lane - 1
        const uint32_t next = uint32_t(__builtin_amdgcn_update_dpp(0, int(node), 0x130, 0xf, 0xf, false));      // wave_shl:1: from lane + 1
        tnode = rev ? prev : next;
      } else {
        unsigned long long k2 = kept;
        asm volatile("; a tile with positions that are not kept" : "+s"(k2));      // (keeps the mask arithmetic out of the path above)
        const TileNeighbour nb = tile_neighbour(k2, tile_lane(meta, lane), lane, tile_lanes_le(lane), rev);
        is_first = nb.is_first; is_last = nb.is_last;
        nb_lane = nb.has ? nb.lane : (rev ? lane - i - 1u : lane + rem + 1u);      // (none: a lane outside the record)
        tnode = lane_read(node, nb.lane);
        q = tile_kept_index(k2, tile_lane(meta, lane), tile_lanes_le(lane));
      }
      if constexpr ((OPT & 64u) != 0u) {      // (timing-only instance)
        if (keep && (uint32_t(kept >> 7) ^ node ^ tnode ^ q ^ uint32_t(is_first) ^ nb_lane) == 0x12345677u) ncnt[0] = 1u;
        continue;
      }
      PANTAS_STAMP(2, tnode);           // kept / plain, ballot, neighbours
      // the link to the neighbour: forward to the next kept position, reverse to the previous one; its slot is looked for among
      // the node's first three out-targets -- forward only when the neighbour is the next path node, as in walk_short_w
      const bool has = keep && (rev ? !is_first : !is_last);
      const bool look = has && (rev || nb_lane == lane + 1u);
      const uint32_t code = !look ? 0u : inf.t0 == tnode ? 1u : inf.t1 == tnode ? 2u : inf.t2 == tnode ? 3u : 0u;
      const bool enter = plain != 0u && !is_first, leave = plain != 0u && !is_last;
      const uint32_t a_il = (rev ? leave : enter) ? plain : 0u, a_ol = (rev ? enter : leave) ? plain : 0u;
      // a link whose slot is not among the node's first three out-targets (or that the graph does not have): into the wave's own
      // region of the workgroup queue -- slots from a ballot, no atomic, no wait -- and looked up at the end of the chunk
      if constexpr ((OPT & 32u) == 0u) {
        const bool later = has && code == 0u;
        const unsigned long long lm = __ballot(later);
        if (lm != 0ull) {                                                   // (wave-uniform)
          const uint32_t more = uint32_t(__builtin_popcountll(lm));
          if (queued + more > REGION) { drain_region(); queued = 0u; }      // (wave-uniform) the region is full: its links looked up now, a lane each
          if (later) {
            const uint32_t slot = queued + __builtin_amdgcn_mbcnt_hi(uint32_t(lm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(lm), 0u));
            const uint64_t order = event_order(pr.line_id, (rev ? nb_lane : lane) - (lane - i), 0);
            lq[wave * REGION + slot] = make_uint4(node, tnode, uint32_t(order) | ((look && inf.t2 == NO_TARGET) ? 1u : 0u), uint32_t(order >> 32));
          }
          queued += more;
        }
      }
      PANTAS_STAMP(3, code);            // link matching, queue
      if (keep) {
        if constexpr ((OPT & 16u) != 0u) {    // (timing-only instance)
          if ((node ^ a_il ^ a_ol ^ code) == 0x12345677u) ncnt[0] = 1u;
          continue;
        }
        // the table update.  The node table is 256 buckets of four slots, a node's bucket from its low bits (the nodes of a chunk
        // are runs of consecutive ids): ONE 16-byte read finds the node -- or an empty slot to put it in -- for all but a node whose
        // bucket is full; then one 64-bit add (one more for a hit on the second / third out-slot).  A flagged node, a full bucket,
        // a lost race for an empty slot: through bump_w, which probes on from the bucket's first slot.
        const uint32_t flags = inf.word & (FLAG_IL | FLAG_OL);
        const uint32_t h0 = (node << 2) & (CN - 1u);
        const uint4 k4 = *reinterpret_cast<const uint4*>(&nkey[h0]);
        uint32_t slot = k4.x == node ? 0u : k4.y == node ? 1u : k4.z == node ? 2u : k4.w == node ? 3u : 4u;
        if (slot == 4u) {
          const uint32_t e = k4.x == LDS_EMPTY ? 0u : k4.y == LDS_EMPTY ? 1u : k4.z == LDS_EMPTY ? 2u : k4.w == LDS_EMPTY ? 3u : 4u;
          if (e < 4u) {
            const uint32_t old = atomicCAS(&nkey[h0 + e], LDS_EMPTY, node);
            if (old == LDS_EMPTY || old == node) slot = e;
          }
        }
        // (a flagged side of a node -- every node of an expressed gene is flagged before long -- asks for the first line that touched
        // its common key: a running minimum per slot, read before it is lowered; only a line with a '-' needs its exact order)
        const bool f_il = a_il != 0u && (flags & FLAG_IL) != 0u, f_ol = a_ol != 0u && (flags & FLAG_OL) != 0u;
        if (slot < 4u && !((f_il || f_ol) && (pr.tw & POS_HAS_DEL))) {
          const uint32_t h = h0 + slot;
          const unsigned long long inc = 1ull | (static_cast<unsigned long long>(a_il) << 16) | (static_cast<unsigned long long>(a_ol) << 32) |
                                         (static_cast<unsigned long long>(code == 1u ? 1u : 0u) << 48);
          atomicAdd(reinterpret_cast<unsigned long long*>(ncnt) + h, inc);
          if (code >= 2u) __hip_atomic_fetch_add(&ncnt2[h], code == 2u ? 1u : 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (f_il || f_ol) {
            const unsigned long long cur = __atomic_load_n(reinterpret_cast<const unsigned long long*>(__builtin_assume_aligned(omin, 8)) + h, __ATOMIC_RELAXED);
            if (f_il && pr.line_id < uint32_t(cur)) atomicMin(&omin[2u * h], pr.line_id);
            if (f_ol && pr.line_id < uint32_t(cur >> 32)) atomicMin(&omin[2u * h + 1u], pr.line_id);
          }
        } else {
          sink.template bump_w<2>(node, flags, pr.line_id, a_il, a_ol, code, (pr.tw & POS_HAS_DEL) ? event_order(pr.line_id, q, 1) : 0ull);
        }
      }
      PANTAS_STAMP(4, node);              // table update (issue: the adds themselves are not waited for)
    }
    if (refused != 0xFFFFFFFFu) report_status(g, refused, WALK_OPS_EXHAUSTED);
    drain_region();                          // what is left in the wave's region (lq_n stays 0: the flush below finds nothing queued)
    if constexpr ((OPT & 1024u) != 0u) {
      PANTAS_STAMP(5, lane);
      if (lane == 0u) for (int u = 0; u < 6; ++u) atomicAdd(reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(b.del_idx)) + u, st_acc[u]);
    }
#undef PANTAS_STAMP
    chunk_tables_flush(g, sink, THREADS);
    __syncthreads();
  }
}

// -- tiles of a batch's short-walk class --
// pass 1: greedy packing of every chunk's records in record order (walk.h tile_pack).  A packing is a chain from record to record, so
// a lane takes a CHUNK: a workgroup of 64 lanes stages the path lengths of 64 chunks in LDS (coalesced) and every lane walks its own.
template <uint32_t CHUNK>
__global__ __launch_bounds__(64) void tile_place_kernel(uint32_t n_short, const uint32_t* node_off, uint32_t* place, uint32_t* chunk_tiles) {
  constexpr uint32_t STRIDE = CHUNK + 4u;      // (lanes read different banks)
  __shared__ uint8_t ks[64u * STRIDE];
  static_assert(sizeof(ks) <= 160u * 1024u, "tile_place_kernel stages 64 chunks of path lengths: 128 KB of the 160 KB of LDS a gfx950 CU has "
                                            "(this library is built for gfx950 only; an architecture with 64 KB of LDS needs 16 chunks per workgroup)");
  const uint32_t c0 = blockIdx.x * 64u, n_chunks = (n_short + CHUNK - 1u) / CHUNK;
  for (uint32_t cc = 0; cc < 64u && c0 + cc < n_chunks; ++cc) {
    const uint32_t lo = (c0 + cc) * CHUNK, hi = min(n_short, lo + CHUNK);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 64u) ks[cc * STRIDE + (i - lo)] = uint8_t(min(node_off[i + 1] - node_off[i], 255u));
  }
  __syncthreads();
  const uint32_t c = c0 + threadIdx.x;
  if (c >= n_chunks) return;
  const uint32_t lo = c * CHUNK, hi = min(n_short, lo + CHUNK);
  const uint8_t* mine = ks + threadIdx.x * STRIDE;
  uint32_t fill = 0, tile = 0, ord = 0, head = lo, head_word = 0;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t k = max(1u, uint32_t(mine[i - lo]));      // (a record without path nodes -- none in the class -- would take a lane)
    uint32_t f = fill;
    if (tile_pack(f, k, i == lo)) {
      if (i != lo) { place[head] = head_word | ((fill - 1u) << 23); ++tile; }
      ord = 0; head = i; head_word = tile << 12;
    } else {
      place[i] = (tile << 12) | (ord << 6) | (f - k);
    }
    fill = f;
    ++ord;
  }
  place[head] = head_word | ((fill - 1u) << 23);
  chunk_tiles[c] = tile + 1u;
}
// pass 2 for a batch whose arrays exist (the host tokeniser's upload, a batch made under another default kernel), a thread per record:
// its positions, its line record, and -- the first record of a tile -- what tile_head adds.  (The GPU tokeniser's gather writes the same.)
template <uint32_t CHUNK>
__global__ void tile_fill_kernel(uint32_t n_short, const LineHdr* hdr, const uint32_t* node_off, const uint32_t* op_off, const uint32_t* nodes, const uint32_t* ops,
                                 const uint32_t* place, const uint32_t* chunk_tile, uint32_t* tile_first, uint32_t* pnodes, uint16_t* pmeta, PosRec* prec) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_short; i += gridDim.x * 256u) {
    const uint32_t pl = place[i], tile = chunk_tile[i / CHUNK] + place_tile(pl), ord = place_ord(pl), l0 = place_lane(pl);
    const uint32_t n0 = node_off[i], k = node_off[i + 1] - n0, o0 = op_off[i], c = op_off[i + 1] - o0;
    ShortOps so;
    if (c < SHORT_MAX_OPS) short_ops_scan(ops + o0, c, so);      // (as make_rec_entry)
    const PosRec pr = make_pos_rec(hdr[i], so);
    prec[i] = pr;
    if (pr.line.tw & POS_NOT_PLAIN) atomicOr(&tile_first[tile], TILE_NOT_PLAIN);
    for (uint32_t j = 0; j < k; ++j) { pnodes[size_t(tile) * 64u + l0 + j] = nodes[n0 + j]; pmeta[size_t(tile) * 64u + l0 + j] = uint16_t(pos_meta(j, k, ord)); }
    if (l0 == 0u) {
      if (k == 0u) { pnodes[size_t(tile) * 64u] = 0u; pmeta[size_t(tile) * 64u] = uint16_t(POS_INACTIVE); }
      tile_head(tile, i, place_used(pl), k ? nodes[n0] : 0u, tile_first, pnodes, pmeta);
    }
  }
}

__global__ void clear_flags_kernel(uint32_t* word, uint2* info, uint4* info3, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) { word[i] &= LEN_MASK; info[i].x &= LEN_MASK; info3[i].x &= LEN_MASK; }
}

__global__ void build_info_kernel(const uint32_t* word, const uint32_t* row_ptr, const uint32_t* col, uint2* info, uint4* info3, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint32_t lo = row_ptr[i], hi = row_ptr[i + 1];
    const uint32_t t0 = lo < hi ? col[lo] : NO_TARGET;
    info[i] = make_uint2(word[i], t0);
    info3[i] = make_uint4(word[i], t0, lo + 1u < hi ? col[lo + 1u] : NO_TARGET, lo + 2u < hi ? col[lo + 2u] : NO_TARGET);
  }
}

__global__ void compact_table_kernel(TableView t, uint64_t* keys, uint32_t* cnt, uint64_t* first, uint32_t* cursor) {
  const uint32_t cap = t.mask + 1u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < cap; i += gridDim.x * 256u) {
    const TableSlot sl = t.slots[i];
    if (sl.key != EMPTY_KEY) {
      const uint32_t o = atomicAdd(cursor, 1u);
      keys[o] = sl.key;
      cnt[o] = sl.cnt;
      first[o] = sl.first;
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct Table {
  TableView v{};
  uint32_t cap = 0;
  void free_all() {
    if (v.slots) (void)hipFree(v.slots);
    v.slots = nullptr; cap = 0;
  }
};

struct DeviceGraph {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t* evs = ev;           // the events the launch functions record: ev, or the slot of a launch that is not waited for
  // launches enqueued without a wait (launch_augment_async): their events, read by launch_collect
  struct InFlight { hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool split = false, short_all = false; };
  std::vector<InFlight> ring;
  std::vector<uint32_t> in_flight;      // ring slots, oldest first
  bool last_split = false, last_short_all = false;
  bool start_is_rare_end = false;      // set around launch_main_impl by the callers that have just enqueued the batch's rare-key kernel
  hipStream_t stream2 = nullptr;  // the general-walk kernel runs here, beside the short-walk kernel on `stream`
  float last_short_ms = 0.f;      // duration of the short-walk kernel of the last launch_main (0 when one kernel did everything)
  uint32_t V = 0;
  uint64_t A = 0;
  uint32_t* word = nullptr;
  uint2* info = nullptr;          // {word, first out-target} per node (GraphView::info)
  uint4* info3 = nullptr;         // {word, first three out-targets} per node (GraphView::info3)
  uint32_t* row_ptr = nullptr;
  uint32_t* col = nullptr;
  uint32_t* counters = nullptr;   // [{NC, IL0, OLL, RC first slot} x V | RC A]  (GraphView::cnt4, ::rc)
  std::unique_ptr<ScratchPool> scratch;   // tokenize.hip temporaries
  uint32_t* name_table = nullptr; // dense segment-name table for the GPU tokeniser (tokenize.hip), uploaded on first use
  uint32_t* name_blocks = nullptr; // per 1024 entries of it: the node of the block's first name when the block counts up one by one (gaf_device.h NAME_BLOCK)
  uint32_t* small = nullptr;      // [0..1] status (u64), [2] hist n_keys, [3] hist overflow, [4] novel n_keys, [5] novel overflow, [6] cursor
  Table hist, novel;
  // multi-batch protocol: every batch's rare-key kernel must run before the first main kernel of the run
  std::vector<uint64_t> main_done;   // ids of the batches whose main kernel has run at least once this run
  std::vector<uint64_t> rare_done;   // ids of the batches whose rare-key kernel has run (DeviceBatch::id: never reused, unlike addresses)
  bool main_started = false;
  uint64_t hist_bound = 0;        // bound on the distinct IL/OL keys of the batches registered so far this run (4 per path node of a record with a '-')
  uint64_t novel_room = 0;        // keys the novel-link table must be able to take (what it held when the last batch registered + that batch's guess)
  uint64_t novel_boost = 1;       // grown after a TableOverflow("novel") retry
  uint64_t hist_boost = 1;        // grown after a TableOverflow("hist") retry
  uint64_t imported_flags = 0;    // node flags received from other ranks this run
  uint64_t reserved_hist = 0;     // caller's bound on distinct IL/OL keys (sharded runs)
  // the run's bitmap of touched counter blocks (SlowTables::touched): kept when a caller asked for it (read-sharded runs),
  // valid when it has been kept since the counters were last zeroed
  uint32_t* touched = nullptr;
  uint64_t touched_words = 0;
  bool track_touched = false, touched_valid = false;
  SlowTables* slow_dev = nullptr;   // device copy of {hist, novel, status}
  SlowTables slow_host{};           // what slow_dev holds
  // brings the device copy up to date (table pointers change when a table grows); ordered on `stream`
  TildeEntry* tilde_dev = nullptr;  // walk.h TildeCarry of the run (set_tilde_table), kept until the next one
  uint32_t tilde_n = 0;
  void sync_slow() {
    SlowTables t{hist.v, novel.v, reinterpret_cast<unsigned long long*>(small), TildeCarry{tilde_dev, tilde_n}, (track_touched && touched_valid) ? touched : nullptr};
    if (memcmp(&t, &slow_host, sizeof t) != 0) {
      slow_host = t;
      HIP_OK(hipMemcpyAsync(slow_dev, &slow_host, sizeof t, hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));   // slow_host is the staging buffer
    }
  }
  GraphView view() const {
    GraphView g{};
    g.word = word; g.info = info; g.info3 = info3; g.row_ptr = row_ptr; g.col = col;
    g.cnt4 = counters; g.rc = counters + 4 * size_t(V);
    g.slow = slow_dev;
#ifdef PANTAS_EXPERIMENTS
    g.ablate = getenv("PANTAS_ABLATE") ? uint32_t(atoi(getenv("PANTAS_ABLATE"))) : 0u;      // timing experiments: results wrong by construction
#else
    // the shipped library has no timing-only instances: a run that asks for one is refused, not silently wrong
    if (getenv("PANTAS_ABLATE") && atoi(getenv("PANTAS_ABLATE")) != 0)
      throw Error("Unsupported", "PANTAS_ABLATE is a timing experiment of builds made with `make EXPERIMENTS=1`; this library does not have it");
    g.ablate = 0u;
#endif
    return g;
  }
};

DeviceGraphInfo graph_info(DeviceGraph& dg) { return DeviceGraphInfo{dg.device, dg.stream, dg.V}; }

ScratchPool& scratch_pool(DeviceGraph& dg) {
  if (!dg.scratch) dg.scratch = std::make_unique<ScratchPool>();
  return *dg.scratch;
}

uint32_t* device_name_table(DeviceGraph& dg, const Graph& g) {
  if (!dg.name_table) {
    HIP_OK(hipSetDevice(dg.device));
    HIP_OK(dmalloc(&dg.name_table, std::max<size_t>(1, g.names.table.size()) * 4));
    HIP_OK(hipMemcpyAsync(dg.name_table, g.names.table.data(), g.names.table.size() * 4, hipMemcpyHostToDevice, dg.stream));
    // Names of a graph are numbers that count up with the S lines, nodes count up with the S lines: over long stretches
    // table[x] = x + c.  One word per block of 1024 names says so (8 MB for a whole genome: it stays in the caches), and the
    // tokeniser's lookups only go to the table itself -- a gigabyte, every lookup a trip to HBM -- where a block does not.
    const size_t n = g.names.table.size(), nb = (n + NAME_BLOCK - 1) / NAME_BLOCK;
    std::vector<uint32_t> blocks(std::max<size_t>(1, nb), NAME_BLOCK_MIXED);
    parallel_chunks(nb, nb < 1024 ? 1u : hw_threads(), [&](unsigned, size_t b0, size_t b1) {
      for (size_t b = b0; b < b1; ++b) {
        const size_t lo = b * NAME_BLOCK, hi = std::min(n, lo + NAME_BLOCK);
        const uint32_t first = g.names.table[lo];
        bool runs = first != NONE32 && first < NAME_BLOCK_MIXED - NAME_BLOCK;
        for (size_t x = lo + 1; x < hi && runs; ++x) runs = g.names.table[x] == first + uint32_t(x - lo);
        if (runs) blocks[b] = first;
      }
    });
    HIP_OK(dmalloc(&dg.name_blocks, blocks.size() * 4));
    HIP_OK(hipMemcpyAsync(dg.name_blocks, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice, dg.stream));
    HIP_OK(hipStreamSynchronize(dg.stream));
  }
  return dg.name_table;
}
uint32_t* device_name_blocks(DeviceGraph& dg) { return dg.name_blocks; }

void DeviceGraphDeleter::operator()(DeviceGraph* d) const {
  if (!d) return;
  (void)hipSetDevice(d->device);
  for (void* p : {(void*)d->word, (void*)d->info, (void*)d->info3, (void*)d->row_ptr, (void*)d->col, (void*)d->counters, (void*)d->small, (void*)d->name_table, (void*)d->name_blocks, (void*)d->slow_dev, (void*)d->tilde_dev})
    if (p) (void)hipFree(p);
  d->scratch.reset();
  d->hist.free_all();
  d->novel.free_all();
  for (auto& e : d->ev) if (e) (void)hipEventDestroy(e);
  for (auto& f : d->ring) for (auto& e : f.ev) if (e) (void)hipEventDestroy(e);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  if (d->stream2) (void)hipStreamDestroy(d->stream2);
  delete d;
}

void DeviceBatchDeleter::operator()(DeviceBatch* b) const {
  if (!b) return;
  for (void* p : {(void*)b->hdr, (void*)b->node_off, (void*)b->op_off, (void*)b->nodes, (void*)b->ops, (void*)b->del_idx, (void*)b->rec,
                  (void*)b->tile_first, (void*)b->pnodes, (void*)b->pmeta, (void*)b->prec, (void*)b->chunk_tile})
    if (p) (void)hipFree(p);
  delete b;
}

// Two pinned staging buffers and their events PER DEVICE: an event belongs to the device that was current when it was made, and
// the C ABI lets one process drive several devices (one handle each) -- buffers made under device 0 and recorded on a stream of
// device 1 fail with hipErrorInvalidHandle.  One copy at a time per direction (the mutex); they live as long as the process.
namespace {
struct Staging {
  void* pinned[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
};
Staging& staging_of_current_device(std::map<int, Staging>& per_device, size_t bytes) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  Staging& s = per_device[dev];
  if (!s.pinned[0]) {
    for (int k = 0; k < 2; ++k) {
      HIP_OK(hipHostMalloc(&s.pinned[k], bytes, hipHostMallocDefault));
      HIP_OK(hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming));
    }
  }
  return s;
}
}  // namespace

void h2d_staged(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  constexpr size_t STAGE = size_t(32) << 20;
  if (bytes == 0) return;
  if (bytes < 2 * STAGE) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return;
  }
  static std::mutex mu;
  static std::map<int, Staging> per_device;
  std::lock_guard<std::mutex> lk(mu);
  Staging& sg = staging_of_current_device(per_device, STAGE);
  void* const* pinned = sg.pinned;
  hipEvent_t* done = sg.done;
  const unsigned threads = std::min<unsigned>(hw_threads(), 16);
  bool used[2] = {false, false};
  int k = 0;
  for (size_t off = 0; off < bytes; off += STAGE, k ^= 1) {
    const size_t len = std::min(STAGE, bytes - off);
    if (used[k]) HIP_OK(hipEventSynchronize(done[k]));
    char* stage = static_cast<char*>(pinned[k]);
    const char* from = static_cast<const char*>(src) + off;
    parallel_chunks(len, threads, [&](unsigned, size_t b, size_t e) { memcpy(stage + b, from + b, e - b); });
    HIP_OK(hipMemcpyAsync(static_cast<char*>(dst) + off, stage, len, hipMemcpyHostToDevice, stream));
    HIP_OK(hipEventRecord(done[k], stream));
    used[k] = true;
  }
  HIP_OK(hipStreamSynchronize(stream));
}

// the other way: device -> pageable host memory through two pinned buffers, the host side copied out by several threads while the
// next piece is on its way (hipMemcpy into pageable memory stages through ONE internal loop: 18 GB/s on the GPU box)
void d2h_staged(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  constexpr size_t STAGE = size_t(32) << 20;
  if (bytes == 0) return;
  if (bytes < 2 * STAGE) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return;
  }
  static std::mutex mu;
  static std::map<int, Staging> per_device;
  std::lock_guard<std::mutex> lk(mu);
  Staging& sg = staging_of_current_device(per_device, STAGE);
  void* const* pinned = sg.pinned;
  hipEvent_t* done = sg.done;
  const unsigned threads = std::min<unsigned>(hw_threads(), 16);
  const size_t n = (bytes + STAGE - 1) / STAGE;
  auto issue = [&](size_t i) {
    const size_t off = i * STAGE, len = std::min(STAGE, bytes - off);
    HIP_OK(hipMemcpyAsync(pinned[i & 1], static_cast<const char*>(src) + off, len, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipEventRecord(done[i & 1], stream));
  };
  issue(0);
  for (size_t i = 0; i < n; ++i) {
    HIP_OK(hipEventSynchronize(done[i & 1]));
    const size_t off = i * STAGE, len = std::min(STAGE, bytes - off);
    // piece i is in its buffer; piece i + 1 goes into the OTHER buffer while this one is copied out
    if (i + 1 < n) issue(i + 1);
    const char* stage = static_cast<const char*>(pinned[i & 1]);
    char* to = static_cast<char*>(dst) + off;
    parallel_chunks(len, threads, [&](unsigned, size_t b, size_t e) { memcpy(to + b, stage + b, e - b); });
  }
}

__global__ void warm_augment_kernel() {}
void warm_up_call_module(hipStream_t s);        // call.hip
void warm_up_tokenize_module(hipStream_t s);    // tokenize.hip

// What the first HIP calls of a process pay for -- the context, the queues behind the first stream, the first allocation, the
// code objects of this library (one per .hip file, loaded at the first launch of one of its kernels): about a second, spent here on
// a thread of its own while the graph is parsed.
static std::mutex g_spare_mu;
static std::vector<std::pair<int, hipStream_t>> g_spare_streams;      // made by device start-up, handed out by take_stream
constexpr int SPARE_STREAMS = 4;      // the graph's two, the call tables' one, the text prefetcher's one

hipStream_t take_stream(int device) {
  {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    for (size_t i = 0; i < g_spare_streams.size(); ++i)
      if (g_spare_streams[i].first == device) {
        hipStream_t s = g_spare_streams[i].second;
        g_spare_streams.erase(g_spare_streams.begin() + long(i));
        return s;
      }
  }
  hipStream_t s = nullptr;
  HIP_OK(hipSetDevice(device));
  HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  return s;
}

static void warm_up_body(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return;
  if (hipSetDevice(device) != hipSuccess) return;
  (void)hipFree(nullptr);          // forces the context
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return;
  void* p = nullptr;
  if (hipMalloc(&p, 1 << 20) == hipSuccess) (void)hipFree(p);
  hipLaunchKernelGGL(warm_augment_kernel, dim3(1), dim3(64), 0, s);
  warm_up_call_module(s);
  warm_up_tokenize_module(s);
  (void)hipGetLastError();
  (void)hipStreamSynchronize(s);
  // streams for later: made now, while the process is small
  std::vector<hipStream_t> made{s};
  for (int i = 1; i < SPARE_STREAMS; ++i) {
    hipStream_t x = nullptr;
    if (hipStreamCreateWithFlags(&x, hipStreamNonBlocking) != hipSuccess) break;
    made.push_back(x);
  }
  std::lock_guard<std::mutex> lk(g_spare_mu);
  for (hipStream_t x : made) g_spare_streams.emplace_back(device, x);
}
void warm_up_device(int device) { std::thread([device] { warm_up_body(device); }).detach(); }
void init_device(int device) { warm_up_body(device); }

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static uint32_t pow2_at_least(uint64_t x) {
  uint64_t c = 1024;
  while (c < x && c < (1ull << 31)) c <<= 1;
  return uint32_t(c);
}

static void table_alloc(Table& t, uint32_t cap, uint32_t* n_keys, uint32_t* overflow) {
  if (t.cap >= cap) { t.v.n_keys = n_keys; t.v.overflow = overflow; return; }
  t.free_all();
  HIP_OK(dmalloc(&t.v.slots, size_t(cap) * sizeof(TableSlot)));
  t.cap = cap;
  t.v.mask = cap - 1;
  t.v.n_keys = n_keys;
  t.v.overflow = overflow;
}

__global__ void table_clear_kernel(TableSlot* slots, uint32_t cap) {
  uint4* w = reinterpret_cast<uint4*>(slots);      // two 16-byte stores per slot: {key, first} = all ones, {cnt, pad} = 0
  for (size_t i = size_t(blockIdx.x) * 256u + threadIdx.x; i < 2u * size_t(cap); i += size_t(gridDim.x) * 256u)
    w[i] = (i & 1u) ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(~0u, ~0u, ~0u, ~0u);
}

static void table_clear(Table& t, hipStream_t s) {
  if (!t.cap) return;
  hipLaunchKernelGGL(table_clear_kernel, dim3(uint32_t(std::min<uint64_t>((2ull * t.cap + 255) / 256, 1u << 16))), dim3(256), 0, s, t.v.slots, t.cap);
  HIP_OK(hipGetLastError());
}

DeviceGraphPtr upload_graph(const Graph& g, int device) {
  if (device_count() <= device) throw Error("NoDevice", "no HIP device " + std::to_string(device) + " (the augment kernels have no CPU path)");
  HIP_OK(hipSetDevice(device));
  release_parked_device_memory_if_low();      // (what the GPU GFA loader left parked must not be why these allocations fail)
  PhaseTimer pt("upload_graph");
  DeviceGraphPtr d(new DeviceGraph());
  d->device = device;
  d->V = g.V;
  d->A = g.A;
  d->stream = take_stream(device);
  d->stream2 = take_stream(device);
  pt.lap("streams");
  for (auto& e : d->ev) HIP_OK(hipEventCreate(&e));
  pt.lap("events");
  HIP_OK(dmalloc(&d->word, std::max<size_t>(1, g.V) * 4));
  HIP_OK(dmalloc(&d->row_ptr, (size_t(g.V) + 1) * 4));
  HIP_OK(dmalloc(&d->col, std::max<size_t>(1, g.A) * 4));
  {
    // zero padding behind the counters: the exchange kernels (exchange.hip) read whole blocks
    const size_t w = 4 * size_t(g.V) + g.A;
    const size_t padded = std::max<size_t>(1, (w + COUNTER_PAD_WORDS - 1) / COUNTER_PAD_WORDS) * COUNTER_PAD_WORDS;
    HIP_OK(dmalloc(&d->counters, padded * 4));
    HIP_OK(hipMemsetAsync(d->counters + w, 0, (padded - w) * 4, d->stream));
  }
  HIP_OK(dmalloc(&d->small, 64));
  HIP_OK(dmalloc(&d->slow_dev, sizeof(SlowTables)));
  pt.lap("hipMalloc of the graph arrays and counters");
  h2d_staged(d->word, g.seg_len.data(), size_t(g.V) * 4, d->stream);
  h2d_staged(d->row_ptr, g.row_ptr.data(), (size_t(g.V) + 1) * 4, d->stream);
  h2d_staged(d->col, g.col.data(), size_t(g.A) * 4, d->stream);
  pt.lap("CSR to HBM");
  HIP_OK(dmalloc(&d->info, std::max<size_t>(1, g.V) * 8));
  HIP_OK(dmalloc(&d->info3, std::max<size_t>(1, g.V) * 16));
  if (g.V) {
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(g.V) + 255) / 256, 1u << 16));
    hipLaunchKernelGGL(build_info_kernel, dim3(blocks), dim3(256), 0, d->stream, d->word, d->row_ptr, d->col, d->info, d->info3, g.V);
    HIP_OK(hipGetLastError());
  }
  table_alloc(d->hist, 1024, d->small + 2, d->small + 3);
  table_alloc(d->novel, 1024, d->small + 4, d->small + 5);
  HIP_OK(hipStreamSynchronize(d->stream));
  pt.lap("node info records, tables");
  reset_counters(*d);
  pt.lap("counters zeroed");
  return d;
}

// lengths that leak into lines whose first op is '~' (walk.h TildeCarry), sorted by line id; stays until replaced
void set_tilde_table(DeviceGraph& d, const std::vector<TildeEntry>& tab) {
  HIP_OK(hipSetDevice(d.device));
  HIP_OK(hipStreamSynchronize(d.stream));
  if (d.tilde_dev) { (void)hipFree(d.tilde_dev); d.tilde_dev = nullptr; }
  d.tilde_n = uint32_t(tab.size());
  if (!tab.empty()) {
    HIP_OK(dmalloc(&d.tilde_dev, tab.size() * sizeof(TildeEntry)));
    HIP_OK(hipMemcpy(d.tilde_dev, tab.data(), tab.size() * sizeof(TildeEntry), hipMemcpyHostToDevice));
  }
}

void reset_counters(DeviceGraph& d) {
  HIP_OK(hipSetDevice(d.device));
  HIP_OK(hipMemsetAsync(d.counters, 0, (4 * size_t(d.V) + d.A) * 4, d.stream));
  d.touched_valid = false;
  if (d.track_touched) {
    const uint64_t w = 4 * uint64_t(d.V) + d.A;
    const uint64_t padded = std::max<uint64_t>(1, (w + COUNTER_PAD_WORDS - 1) / COUNTER_PAD_WORDS) * COUNTER_PAD_WORDS;
    const uint64_t need = padded / (32ull * TOUCHED_BLOCK_WORDS);
    if (need != d.touched_words) {
      if (d.touched) (void)hipFree(d.touched);
      d.touched = nullptr;
      HIP_OK(dmalloc(&d.touched, need * 4));
      d.touched_words = need;
    }
    HIP_OK(hipMemsetAsync(d.touched, 0, d.touched_words * 4, d.stream));
    d.touched_valid = true;       // from here on every site that adds to a counter sets its block's bit
  }
  HIP_OK(hipMemsetAsync(d.small, 0xFF, 8, d.stream));
  HIP_OK(hipMemsetAsync(d.small + 2, 0, 56, d.stream));
  table_clear(d.hist, d.stream);
  table_clear(d.novel, d.stream);
  if (d.V) {
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(d.V) + 255) / 256, 4096));
    hipLaunchKernelGGL(clear_flags_kernel, dim3(blocks), dim3(256), 0, d.stream, d.word, d.info, d.info3, d.V);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipStreamSynchronize(d.stream));
  d.in_flight.clear();          // (launches enqueued and never collected: their times go with the counters)
  d.imported_flags = 0;
  d.reserved_hist = 0;
  d.rare_done.clear();
  d.main_done.clear();
  d.main_started = false;
  d.hist_bound = 0;
  d.novel_room = 0;
}

template <class T>
static T* to_device(const std::vector<T>& v, hipStream_t s) {
  T* p = nullptr;
  HIP_OK(dmalloc(&p, std::max<size_t>(1, v.size()) * sizeof(T)));
  if (!v.empty()) HIP_OK(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return p;
}

// every live entry of `from` into `to` (to is empty and larger): counts add, first-seen minima carry over
__global__ void rehash_kernel(TableView from, TableView to) {
  const uint32_t cap = from.mask + 1u;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < cap; i += gridDim.x * 256u) {
    const TableSlot sl = from.slots[i];
    if (sl.key != EMPTY_KEY) table_update(to, sl.key, sl.cnt, sl.first);
  }
}

// Grows a table to `cap` slots keeping what it holds (a run over several batches learns its sizes batch by batch).
static void table_grow(DeviceGraph& d, Table& t, uint32_t cap, uint32_t* n_keys, uint32_t* overflow) {
  if (t.cap >= cap) return;
  Table big;
  table_alloc(big, cap, n_keys, overflow);
  table_clear(big, d.stream);
  if (t.cap) {
    // the rehash counts the keys again as it inserts them
    HIP_OK(hipMemsetAsync(n_keys, 0, 4, d.stream));
    TableView from = t.v;
    from.n_keys = n_keys; from.overflow = overflow;
    const uint32_t blocks = std::min<uint32_t>((t.cap + 255) / 256, 8192);
    hipLaunchKernelGGL(rehash_kernel, dim3(blocks), dim3(256), 0, d.stream, from, big.v);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipStreamSynchronize(d.stream));
  t.free_all();
  t = big;
}

__global__ void build_rec_kernel(uint32_t n, const uint32_t* node_off, const uint32_t* op_off, const uint32_t* nodes, const uint32_t* ops, uint4* rec) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint32_t n0 = node_off[i], o0 = op_off[i];
    make_rec_entry(rec, i, n0, node_off[i + 1] - n0, o0, op_off[i + 1] - o0, nodes + n0, ops + o0);
  }
}

int main_kernel_variant();
void ensure_batch_rec(DeviceGraph& d, DeviceBatch& b) {
  if (b.rec) return;
  HIP_OK(hipSetDevice(d.device));
  HIP_OK(dmalloc(&b.rec, std::max<size_t>(1, b.n) * REC_QUADS * sizeof(uint4)));
  if (b.n) {
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(b.n) + 255) / 256, 1u << 16));
    hipLaunchKernelGGL(build_rec_kernel, dim3(blocks), dim3(256), 0, d.stream, b.n, b.node_off, b.op_off, b.nodes, b.ops, b.rec);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipStreamSynchronize(d.stream));
}

// Since the line records of the tiles hold the op summaries (late round 4), the position-parallel kernel reads nothing of rec: a
// batch made while it is the default has tiles and no rec (48 bytes per record less to write and to keep); ensure_batch_rec makes
// it when a record-per-lane kernel is asked for.
void build_batch_rec(DeviceGraph& d, DeviceBatch& b) {
  if (main_kernel_variant() >= 120) { if (b.n_short && !b.tile_first) build_batch_tiles(d, b); }
  else ensure_batch_rec(d, b);
}

// PANTAS_MAIN_KERNEL as the default of this build: 120 = one lane per path position (round 4), 110 = one lane per record
// over 16-byte node records (round 3), 101 = round 2's, 0 = direct atomics
// (anything else is refused here, where every user of the switch reads it: a mistyped value used to count as "position-parallel"
// when the batch was built -- no rec -- and fall to the record-per-lane kernel at launch, which reads rec)
int main_kernel_variant() {
  const char* e = getenv("PANTAS_MAIN_KERNEL");
  if (!e || !*e) return 120;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  bool known = v == 0 || v == 101 || v == 110 || v == 120;
#ifdef PANTAS_EXPERIMENTS
  known = known || (v >= 124 && v <= 128) || v == 130;
#endif
  if (*end || !known) throw Error("ValueError", std::string("PANTAS_MAIN_KERNEL=") + e + ": not a kernel of this build (0, 101, 110, 120)");
  return int(v);
}

uint32_t* place_batch_tiles(int device, hipStream_t stream, ScratchPool& pool, DeviceBatch& b, const uint32_t* node_off) {
  struct { int device; hipStream_t stream; } d{device, stream};
  HIP_OK(hipSetDevice(d.device));
  for (void* p : {(void*)b.tile_first, (void*)b.pnodes, (void*)b.pmeta, (void*)b.prec, (void*)b.chunk_tile}) if (p) (void)hipFree(p);
  b.tile_first = nullptr; b.pnodes = nullptr; b.pmeta = nullptr; b.prec = nullptr; b.chunk_tile = nullptr; b.n_tiles = 0;
  if (!b.n_short) return nullptr;
  if (b.n_short >= (1u << 31)) throw Error("Unsupported", "more than 2^31 - 1 short-walk records in one batch");
  const uint32_t n_chunks = (b.n_short + TILE_CHUNK - 1u) / TILE_CHUNK;
  uint32_t* place = static_cast<uint32_t*>(pool.take(size_t(b.n_short) * 4));
  uint32_t* counts = static_cast<uint32_t*>(pool.take((size_t(n_chunks) + 1) * 4));
  HIP_OK(dmalloc(&b.chunk_tile, (size_t(n_chunks) + 1) * 4));
  HIP_OK(dmalloc(&b.prec, size_t(b.n_short) * sizeof(PosRec)));
  HIP_OK(hipMemsetAsync(counts + n_chunks, 0, 4, d.stream));
  hipLaunchKernelGGL((tile_place_kernel<TILE_CHUNK>), dim3((n_chunks + 63u) / 64u), dim3(64), 0, d.stream, b.n_short, node_off, place, counts);
  HIP_OK(hipGetLastError());
  size_t need = 0;
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, counts, b.chunk_tile, int(n_chunks + 1), d.stream));
  void* tmp = pool.take(need);
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp, need, counts, b.chunk_tile, int(n_chunks + 1), d.stream));
  uint32_t n_tiles = 0;
  HIP_OK(hipMemcpyAsync(&n_tiles, b.chunk_tile + n_chunks, 4, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipStreamSynchronize(d.stream));
  pool.give(counts); pool.give(tmp);
  b.n_tiles = n_tiles;
  HIP_OK(dmalloc(&b.tile_first, size_t(n_tiles) * 4));
  HIP_OK(dmalloc(&b.pnodes, size_t(n_tiles) * 64 * 4));
  HIP_OK(dmalloc(&b.pmeta, size_t(n_tiles) * 64 * 2));
  HIP_OK(hipMemsetAsync(b.tile_first, 0, size_t(n_tiles) * 4, d.stream));
  return place;
}

void build_batch_tiles(DeviceGraph& d, DeviceBatch& b) {
  uint32_t* place = place_batch_tiles(d.device, d.stream, scratch_pool(d), b, b.node_off);
  if (!place) return;
  const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(b.n_short) + 255) / 256, 1u << 16));
  hipLaunchKernelGGL((tile_fill_kernel<TILE_CHUNK>), dim3(blocks), dim3(256), 0, d.stream, b.n_short, b.hdr, b.node_off, b.op_off, b.nodes, b.ops, place,
                     b.chunk_tile, b.tile_first, b.pnodes, b.pmeta, b.prec);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(d.stream));
  scratch_pool(d).give(place);
}

DeviceBatchPtr upload_batch(DeviceGraph& d, const GafBatch& b) {
  HIP_OK(hipSetDevice(d.device));
  if (b.size() >= NONE32) throw Error("Unsupported", "more than 2^32-2 records in one batch");
  // host-side shape check before any kernel sees the batch (a fault can take the whole node down)
  for (size_t i = 0; i < b.nodes.size(); ++i)
    if (b.nodes[i] >= d.V) throw Error("InternalError", "batch names a node outside the graph");
  if (b.node_off.size() != b.size() + 1 || b.op_off.size() != b.size() + 1 || b.node_off.back() != b.nodes.size() ||
      b.op_off.back() != b.ops.size())
    throw Error("InternalError", "batch offsets do not match its arrays");
  DeviceBatchPtr p(new DeviceBatch());
  p->n = uint32_t(b.size());
  p->n_short = uint32_t(std::min<uint64_t>(b.n_short, b.size()));
  p->short_nodes = b.node_off[p->n_short];
  p->short_ops = b.op_off[p->n_short];
  p->n_nodes = b.nodes.size();
  p->n_ops = b.ops.size();
  p->del_nodes = b.del_nodes;
  std::vector<uint32_t> del;
  del.reserve(b.del_lines);
  for (size_t i = 0; i < b.size(); ++i) if (b.hdr[i].flags & LINE_HAS_DEL) del.push_back(uint32_t(i));
  p->n_del = uint32_t(del.size());
  p->hdr = to_device(b.hdr, d.stream);
  p->node_off = to_device(b.node_off, d.stream);
  p->op_off = to_device(b.op_off, d.stream);
  p->nodes = to_device(b.nodes, d.stream);
  p->ops = to_device(b.ops, d.stream);
  p->del_idx = to_device(del, d.stream);
  HIP_OK(hipStreamSynchronize(d.stream));
  build_batch_rec(d, *p);
  return p;
}

void download_batch(const DeviceGraph& d, const DeviceBatch& b, GafBatch& out) {
  HIP_OK(hipSetDevice(d.device));
  out = GafBatch{};
  out.hdr.resize(b.n);
  out.node_off.resize(size_t(b.n) + 1);
  out.op_off.resize(size_t(b.n) + 1);
  out.nodes.resize(b.n_nodes);
  out.ops.resize(b.n_ops);
  if (b.n) HIP_OK(hipMemcpy(out.hdr.data(), b.hdr, size_t(b.n) * sizeof(LineHdr), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out.node_off.data(), b.node_off, (size_t(b.n) + 1) * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out.op_off.data(), b.op_off, (size_t(b.n) + 1) * 4, hipMemcpyDeviceToHost));
  if (b.n_nodes) HIP_OK(hipMemcpy(out.nodes.data(), b.nodes, b.n_nodes * 4, hipMemcpyDeviceToHost));
  if (b.n_ops) HIP_OK(hipMemcpy(out.ops.data(), b.ops, b.n_ops * 4, hipMemcpyDeviceToHost));
  out.del_lines = b.n_del;
  out.del_nodes = b.del_nodes;
}

uint64_t batch_lines(const DeviceBatch& db) { return db.n; }
uint64_t batch_short_records(const DeviceBatch& db) { return db.n_short; }
void batch_sizes(const DeviceBatch& db, uint64_t* path_nodes, uint64_t* cs_ops) { *path_nodes = db.n_nodes; *cs_ops = db.n_ops; }

uint64_t batch_alg_bytes(const DeviceBatch& db) {
  // B_line = 16 + 16 k + 4 c + 21 (k - 1)
  const uint64_t n = db.n;
  return 16 * n + 16 * db.n_nodes + 4 * db.n_ops + 21 * (db.n_nodes - std::min<uint64_t>(n, db.n_nodes));
}

uint64_t batch_alg_bytes_short(const DeviceBatch& db) {
  const uint64_t n = db.n_short;
  return 16 * n + 16 * db.short_nodes + 4 * db.short_ops + 21 * (db.short_nodes - std::min<uint64_t>(n, db.short_nodes));
}

static BatchView batch_view(const DeviceBatch& b) {
  BatchView v{};
  v.hdr = b.hdr; v.node_off = b.node_off; v.op_off = b.op_off; v.nodes = b.nodes; v.ops = b.ops; v.del_idx = b.del_idx; v.rec = b.rec;
  v.n = b.n; v.n_del = b.n_del;
  return v;
}

// A batch enters the run (its rare-key kernel is about to be launched for the first time): account for the IL/OL keys it
// can add -- every rare event and every flagged common key is at most one key per (node, side) visit of a record with a
// '-' op: a true bound, summed over the batches of the run.
static void register_batch(DeviceGraph& d, const DeviceBatch& b) { d.hist_bound += 4 * b.del_nodes; }

// The first main kernel of a batch is about to run: room in the novel-link table for what it holds now plus a guess for
// this batch (a quarter of its links; an overflow is reported by finish and the caller redoes the run with novel_boost
// raised).  Sized batch by batch: a streamed run does not know its length when it starts.
static void register_batch_main(DeviceGraph& d, const DeviceBatch& b) {
  const uint64_t edges = b.n_nodes > b.n ? b.n_nodes - b.n : 0;
  const uint64_t guess = std::min<uint64_t>(edges, std::max<uint64_t>(1u << 16, edges / 4 * d.novel_boost));
  uint32_t held = 0;
  if (!d.main_done.empty()) {    // not the first batch of the run (whose table is empty: no round trip)
    HIP_OK(hipMemcpyAsync(&held, d.small + 4, 4, hipMemcpyDeviceToHost, d.stream));   // behind the kernels of the batches before
    HIP_OK(hipStreamSynchronize(d.stream));
  }
  d.novel_room = std::max<uint64_t>(d.novel_room, uint64_t(held) + guess);
}

// Tables large enough for everything registered so far; growing keeps their contents (rehash on the device).
static void ensure_tables(DeviceGraph& d) {
  const uint64_t hist_keys = std::max<uint64_t>(d.hist_bound, d.reserved_hist) * d.hist_boost;
  uint32_t hc = pow2_at_least(2 * hist_keys + 1024);
  uint32_t nc = pow2_at_least(2 * d.novel_room + 1024);
  // (tests: tables that start with PANTAS_TABLE_SLOTS_FIRST slots, so that a small workload overflows them and the run is redone
  // with four times the room -- pantas_augment_run's TableOverflow loop, tests/test_gpu_stream.py)
  static const uint64_t slots_first = getenv("PANTAS_TABLE_SLOTS_FIRST") ? strtoull(getenv("PANTAS_TABLE_SLOTS_FIRST"), nullptr, 10) : 0;
  if (slots_first) {
    hc = std::min<uint64_t>(hc, pow2_at_least(slots_first * d.hist_boost));
    nc = std::min<uint64_t>(nc, pow2_at_least(slots_first * d.novel_boost));
  }
  if (hc > d.hist.cap) table_grow(d, d.hist, hc, d.small + 2, d.small + 3);
  if (nc > d.novel.cap) table_grow(d, d.novel, nc, d.small + 4, d.small + 5);
}

void boost_novel_table(DeviceGraph& d) { d.novel_boost *= 4; }
void boost_hist_table(DeviceGraph& d) { d.hist_boost *= 4; }
uint64_t device_free_bytes(DeviceGraph& d) {
  HIP_OK(hipSetDevice(d.device));
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  return free_b;
}
void adopt_rare_done(DeviceGraph& d, DeviceBatch& b) {
  if (std::find(d.rare_done.begin(), d.rare_done.end(), b.id) == d.rare_done.end()) d.rare_done.push_back(b.id);
}
void reserve_hist(DeviceGraph& d, uint64_t keys) { d.reserved_hist = keys; }

static float launch_rare_impl(DeviceGraph& d, DeviceBatch& b, bool wait);
float launch_rare(DeviceGraph& d, DeviceBatch& b) { return launch_rare_impl(d, b, true); }

// wait = false: the kernel is only enqueued (launch_augment lets the main kernels follow it in the stream and reads the
// events once, instead of two host round trips per step)
static float launch_rare_impl(DeviceGraph& d, DeviceBatch& b, bool wait) {
  HIP_OK(hipSetDevice(d.device));
  const bool seen = std::find(d.rare_done.begin(), d.rare_done.end(), b.id) != d.rare_done.end();
  if (d.main_started && !seen)
    throw Error("ProtocolError", "rare-key kernel of a new batch after a main kernel of the same run: the node flags it sets "
                                 "come too late for the batches already walked (launch_rare for every batch first)");
  if (!seen) { d.rare_done.push_back(b.id); register_batch(d, b); }
  ensure_tables(d);
  d.sync_slow();
  const GraphView g = d.view();
  const BatchView bv = batch_view(b);
  HIP_OK(hipEventRecord(d.evs[0], d.stream));
  if (b.n_del) {
    // PANTAS_RARE_KERNEL (A/B): 0 = every line through the general walk, one global table update per key event; 1 = the
    // short walk for one-indel lines, same updates (round 1); 2 .. 5 = 1 with the workgroup-private LDS stage in front of the
    // global table, chunks of 2048 / 512 / 1024 / 256 records.  S4 (profiles/r2_ab_s4_rare*.txt), plain synthetic deletions /
    // 95 % of 4.4 M deletions on 10 sites: mode 1 0.146 / 16.2 ms, mode 2 0.227 / 0.339, mode 3 0.144 / 0.237, mode 5 (one
    // record per thread, 256 entries: default) 0.124 / 0.246 ms.
    const uint32_t mode = getenv("PANTAS_RARE_KERNEL") ? uint32_t(atoi(getenv("PANTAS_RARE_KERNEL"))) : 5u;
    if (mode >= 2u) {
#define PANTAS_RARE_AGG(CH, LG) hipLaunchKernelGGL((augment_rare_agg_kernel<CH, LG>), dim3(uint32_t(std::min<uint64_t>((uint64_t(b.n_del) + CH - 1) / CH, 1u << 20))), \
                                                   dim3(RARE_THREADS), 0, d.stream, g, bv, 1u)
      if (mode == 3u) PANTAS_RARE_AGG(512, 9);
      else if (mode == 4u) PANTAS_RARE_AGG(1024, 9);
      else if (mode == 5u) PANTAS_RARE_AGG(256, 8);
      else PANTAS_RARE_AGG(2048, 10);
#undef PANTAS_RARE_AGG
    } else {
      const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(b.n_del) + 255) / 256, 1u << 20));
      hipLaunchKernelGGL(augment_rare_kernel, dim3(blocks), dim3(256), 0, d.stream, g, bv, mode);
    }
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipEventRecord(d.evs[5], d.stream));
  if (!wait) return 0.f;
  HIP_OK(hipEventSynchronize(d.evs[5]));
  float a = 0;
  HIP_OK(hipEventElapsedTime(&a, d.evs[0], d.evs[5]));
  return a;
}

static float launch_main_impl(DeviceGraph& d, DeviceBatch& b, bool wait) {
  HIP_OK(hipSetDevice(d.device));
  if (std::find(d.rare_done.begin(), d.rare_done.end(), b.id) == d.rare_done.end())
    throw Error("ProtocolError", "main kernel of a batch whose rare-key kernel has not run");
  d.main_started = true;
  PhaseTimer pt("launch_main");
  if (std::find(d.main_done.begin(), d.main_done.end(), b.id) == d.main_done.end()) { register_batch_main(d, b); d.main_done.push_back(b.id); }
  pt.lap("register batch");
  ensure_tables(d);
  pt.lap("tables sized");
  d.sync_slow();
  const GraphView g = d.view();
  BatchView bv = batch_view(b);
  // (behind the rare-key kernel of the same call the start of the main kernels is that kernel's end event: every event record is a
  // barrier packet between two kernels)
  const hipEvent_t ev_start = d.start_is_rare_end ? d.evs[5] : d.evs[1];
  if (!d.start_is_rare_end) HIP_OK(hipEventRecord(ev_start, d.stream));
  bool short_all = false;
  d.last_short_ms = 0.f;
  if (b.n) {
    // PANTAS_MAIN_KERNEL (A/B, profiles/README.md): 0 = the general walk with direct global atomics for every record; the
    // others run the two record classes in kernels of their own -- the short-walk class [0, n_short) on the launch stream,
    // the rest with the general aggregating kernel beside it on a second stream: 101 = round 2's short walk, 110 = round 3's (default;
    // see the switch).  Every variant leaves the same counters (tools/ab_variants.py).
    const int variant = main_kernel_variant();   // read per launch: tools/ab_variants.py switches it
#define PANTAS_AGG_RANGE(STREAM, BEGIN, END, T, CH, LN, LS, LF, MW, ...) \
    do { if ((END) > (BEGIN)) hipLaunchKernelGGL((augment_main_agg_kernel<T, CH, LN, LS, LF, MW, ##__VA_ARGS__>), \
                                               dim3(((END) - (BEGIN) + CH - 1) / CH), dim3(T), 0, STREAM, g, bv, BEGIN, END); } while (0)
#define PANTAS_AGG_RANGE_ABL(STREAM, BEGIN, END, T, CH, LN, LS, LF, MW, PK, ON, INF, PFD, ...) \
    do { if ((END) > (BEGIN)) hipLaunchKernelGGL((augment_main_agg_kernel<T, CH, LN, LS, LF, MW, PK, ON, true, INF, PFD, ##__VA_ARGS__>), \
                                               dim3(((END) - (BEGIN) + CH - 1) / CH), dim3(T), 0, STREAM, g, bv, BEGIN, END); } while (0)
#define PANTAS_AGG(ID, ...) case ID: PANTAS_AGG_RANGE(d.stream, 0u, b.n, __VA_ARGS__); break;
    // one after the other on the launch stream
#define PANTAS_SPLIT_SEQ(ID, SHORT_ARGS, GENERAL_ARGS) case ID: \
      PANTAS_AGG_RANGE SHORT_ARGS; HIP_OK(hipEventRecord(d.evs[3], d.stream)); split = true; PANTAS_AGG_RANGE GENERAL_ARGS; break;
    // side by side: the general-walk kernel (few records, slow waves, often too few workgroups to fill the chip) on a
    // second stream fills what the short-walk kernel leaves; both add to the same counters with atomics
#define PANTAS_SPLIT(ID, SHORT_ARGS, GENERAL_ARGS) case ID: \
      HIP_OK(hipStreamWaitEvent(d.stream2, ev_start, 0)); \
      PANTAS_AGG_RANGE GENERAL_ARGS; HIP_OK(hipEventRecord(d.evs[4], d.stream2)); \
      PANTAS_AGG_RANGE SHORT_ARGS; HIP_OK(hipEventRecord(d.evs[3], d.stream)); split = true; \
      HIP_OK(hipStreamWaitEvent(d.stream, d.evs[4], 0)); break;
    bool split = false;
    const uint32_t n_ch = (b.n_short + 2047u) / 2048u;      // chunks of the short-walk class
    (void)n_ch;
    if (variant == 0) {
      const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(b.n) + 255) / 256, 1u << 20));
      hipLaunchKernelGGL(augment_main_kernel, dim3(blocks), dim3(256), 0, d.stream, g, bv);
    } else {
      int variant_run = variant;
      if (variant >= 120) {
        // the position-parallel kernel addresses its tiles and line records with 32-bit byte offsets: a batch beyond that (more than
        // 16.7 M tiles = a billion path positions, or 134 M short-walk records; the tokeniser's 8 GiB of text per batch stay below) is
        // walked by the record-per-lane kernel
        if (b.n_short && !b.tile_first) build_batch_tiles(d, b);       // (a batch made while another variant was the default)
        if (b.n_tiles >= (1u << 24) || b.n_short >= (1u << 27)) variant_run = 110;
      }
      // the record-per-lane kernels read rec, which a batch made for the position-parallel kernel does not have
      if ((variant_run < 120 || g.ablate) && !b.rec) { ensure_batch_rec(d, b); bv.rec = b.rec; }
#ifdef PANTAS_EXPERIMENTS
      if (g.ablate) fprintf(stderr, "pantas: PANTAS_ABLATE=%u -- timing experiment, THE RESULTS OF THIS RUN ARE INVALID\n", g.ablate);
      if (g.ablate && variant >= 110) {
        // timing experiments on the round-3 kernel: its ABL instance (the general class is not launched: no such records in the workloads)
        if (b.n_short) hipLaunchKernelGGL((augment_short_w_kernel<512, 2048, 10, 7, 8, 1, true>), dim3(8u * ((n_ch + 7u) / 8u)), dim3(512), 0, d.stream, g, bv, 0u, b.n_short);
        HIP_OK(hipEventRecord(d.evs[3], d.stream));
        split = true;
      } else if (g.ablate) {
        // timing experiments: the default pair of kernels with the ablation switches compiled in, one after the other
        PANTAS_AGG_RANGE_ABL(d.stream, 0u, b.n_short, 512, 2048, 10, 10, 7, 8, true, 1, true, 2, true, true);
        HIP_OK(hipEventRecord(d.evs[3], d.stream));
        split = true;
        PANTAS_AGG_RANGE_ABL(d.stream, b.n_short, b.n, 512, 2048, 10, 10, 7, 8, false, 2, false, 1);
      } else
#endif
      switch (variant_run) {
        // (variants 73 / 80 / 85 / 97 / 100 -- round 1's short walk and round 2's steps towards 101 -- were retired in round 3; their
        // measurements are in DESIGN.md section 3 and profiles/r2_ab_s4_*.txt)
        case 101:
          // round 2's short walk over 8-byte node records {length, first out-target}: links off the first out-slot deferred per line and
          // looked up behind the node loop, links absent from the graph queued per workgroup; chunks dealt to the XCDs in eighths; the
          // line's ShortOps with the record; odd rounds of a chunk the other way round.  The A/B partner of the default.
          HIP_OK(hipStreamWaitEvent(d.stream2, ev_start, 0));
          PANTAS_AGG_RANGE(d.stream2, b.n_short, b.n, 512, 2048, 10, 10, 7, 8, false, 2);
          HIP_OK(hipEventRecord(d.evs[4], d.stream2));
          if (b.n_short) hipLaunchKernelGGL((augment_main_agg_kernel<512, 2048, 10, 10, 7, 8, true, 1, false, true, 2, true, true, 1, true, true>),
                                            dim3(8u * ((n_ch + 7u) / 8u)), dim3(512), 0, d.stream, g, bv, 0u, b.n_short);
          HIP_OK(hipEventRecord(d.evs[3], d.stream)); split = true;
          HIP_OK(hipStreamWaitEvent(d.stream, d.evs[4], 0));
          break;
        case 120: case 124: case 125: case 126: case 127: case 128: case 130: {
          // round 4: the short-walk class with one lane per path POSITION (walk.h tile_*; augment_short_p_kernel) over the batch's tiles
          // of 64 positions; same tables, same chunks of 2048 records dealt to the XCDs in eighths
          // (the general-walk kernel beside it on the second stream -- when the batch has such records: two cross-stream waits cost a
          // small batch a tenth of its step)
          const bool general = b.n > b.n_short;
          if (general) {
            HIP_OK(hipStreamWaitEvent(d.stream2, ev_start, 0));
            PANTAS_AGG_RANGE(d.stream2, b.n_short, b.n, 512, 2048, 10, 10, 7, 8, false, 2);
            HIP_OK(hipEventRecord(d.evs[4], d.stream2));
          }
          if (b.n_short) {
#define PANTAS_P(OPT) hipLaunchKernelGGL((augment_short_p_kernel<512, 2048, 10, 7, 8, OPT>), dim3(8u * ((n_ch + 7u) / 8u)), dim3(512), 0, d.stream, g, bv, \
                                         b.tile_first, b.pnodes, b.pmeta, b.prec, b.chunk_tile, n_ch)
            switch (variant - 120) {
#ifdef PANTAS_EXPERIMENTS
              // timing-only instances (results wrong by construction): 124 a kept position updates nothing, 125 ... and nothing is queued,
              // 126 nothing behind the mask of kept positions
              case 4: PANTAS_P(16); break; case 5: PANTAS_P(48); break; case 6: PANTAS_P(64); break;
              case 7: PANTAS_P(128); break; case 8: PANTAS_P(256); break;      // 127 / 128: 32 more vector / scalar instructions per tile
              case 10: {                                                        // 130: shader-clock stamps between the parts of a tile's work
                static unsigned long long* stamps = nullptr;
                if (!stamps) HIP_OK(hipMalloc(&stamps, 64));
                HIP_OK(hipMemsetAsync(stamps, 0, 64, d.stream));
                BatchView bs = bv;
                bs.del_idx = reinterpret_cast<const uint32_t*>(stamps);
                hipLaunchKernelGGL((augment_short_p_kernel<512, 2048, 10, 7, 8, 1024>), dim3(8u * ((n_ch + 7u) / 8u)), dim3(512), 0, d.stream, g, bs, b.tile_first, b.pnodes,
                                   b.pmeta, b.prec, b.chunk_tile, n_ch);
                unsigned long long hs[6];
                HIP_OK(hipMemcpyAsync(hs, stamps, 48, hipMemcpyDeviceToHost, d.stream));
                HIP_OK(hipStreamSynchronize(d.stream));
                double tot = 0;
                for (double x : hs) tot += x;
                fprintf(stderr, "[stamps] loads+room %.1f %%, scan+lane read %.1f %%, kept+neighbours %.1f %%, links+queue %.1f %%, table update %.1f %%, drain %.1f %% of %.3g wave-clocks\n",
                        100 * hs[0] / tot, 100 * hs[1] / tot, 100 * hs[2] / tot, 100 * hs[3] / tot, 100 * hs[4] / tot, 100 * hs[5] / tot, tot);
                break;
              }
#endif
              default:
                if (d.track_touched && d.touched_valid) PANTAS_P(2048); else PANTAS_P(0);      // (2048: with the marking of touched counter blocks)
                break;
            }
#undef PANTAS_P
          }
          // (no general-class records: the short-walk kernel is the whole launch, its end the launch's end event)
          if (general) { HIP_OK(hipEventRecord(d.evs[3], d.stream)); split = true; HIP_OK(hipStreamWaitEvent(d.stream, d.evs[4], 0)); }
          else short_all = true;
          break;
        }
        default:
          // 110 (round 3): the short-walk class over 16-byte node records (walk.h walk_short_w, augment_short_w_kernel): a link is matched
          // against its node's first three out-targets where the node is visited and counted in the node's own table entry; same chunk
          // order, record layout and round order as 101.  S4: 2.07 against 2.17 ms (profiles/r3_ab_s4_wide.txt); two nodes ahead instead
          // of one measured the same (2.11).
          HIP_OK(hipStreamWaitEvent(d.stream2, ev_start, 0));
          PANTAS_AGG_RANGE(d.stream2, b.n_short, b.n, 512, 2048, 10, 10, 7, 8, false, 2);
          HIP_OK(hipEventRecord(d.evs[4], d.stream2));
          if (b.n_short) hipLaunchKernelGGL((augment_short_w_kernel<512, 2048, 10, 7, 8, 1>), dim3(8u * ((n_ch + 7u) / 8u)), dim3(512), 0, d.stream, g, bv, 0u, b.n_short);
          HIP_OK(hipEventRecord(d.evs[3], d.stream)); split = true;
          HIP_OK(hipStreamWaitEvent(d.stream, d.evs[4], 0));
          break;
      }
    }
#undef PANTAS_SPLIT_SEQ
#undef PANTAS_SPLIT
#undef PANTAS_AGG
#undef PANTAS_AGG_RANGE
#undef PANTAS_AGG_RANGE_ABL
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(d.evs[2], d.stream));
    d.last_split = split;
    d.last_short_all = short_all;
    if (!wait) return 0.f;
    HIP_OK(hipEventSynchronize(d.evs[2]));
    if (split) HIP_OK(hipEventElapsedTime(&d.last_short_ms, ev_start, d.evs[3]));
    else if (short_all) HIP_OK(hipEventElapsedTime(&d.last_short_ms, ev_start, d.evs[2]));
  } else {
    HIP_OK(hipEventRecord(d.evs[2], d.stream));
    d.last_split = false;
    d.last_short_all = false;
    if (!wait) return 0.f;
    HIP_OK(hipEventSynchronize(d.evs[2]));
  }
  float m = 0;
  HIP_OK(hipEventElapsedTime(&m, ev_start, d.evs[2]));
  pt.lap("kernels + wait");
  return m;
}

float launch_main(DeviceGraph& d, DeviceBatch& b) { return launch_main_impl(d, b, true); }

float last_short_kernel_ms(const DeviceGraph& d) { return d.last_short_ms; }

float launch_augment(DeviceGraph& d, DeviceBatch& b, float* ms_rare, float* ms_main) {
  (void)launch_rare_impl(d, b, false);
  d.start_is_rare_end = true;
  float m = 0;
  try { m = launch_main_impl(d, b, true); } catch (...) { d.start_is_rare_end = false; throw; }        // returns after everything enqueued so far has completed
  d.start_is_rare_end = false;
  float a = 0;
  HIP_OK(hipEventElapsedTime(&a, d.evs[0], d.evs[5]));
  if (ms_rare) *ms_rare = a;
  if (ms_main) *ms_main = m;
  return a + m;
}

// The same two kernels enqueued and NOT waited for: a caller that launches batch after batch (or one batch again and again: a
// benchmark) keeps the stream fed, where launch_augment pays a host round trip -- launch, wake-up, event reads, 50 microseconds --
// per call, a quarter of a small batch's step.  The launch's events go into a slot of a ring; launch_collect waits for everything
// enqueued and hands the times back, oldest first.
constexpr size_t LAUNCH_RING = 64;
void launch_augment_async(DeviceGraph& d, DeviceBatch& b) {
  HIP_OK(hipSetDevice(d.device));
  if (d.in_flight.size() >= LAUNCH_RING) throw Error("ProtocolError", "64 launches in flight: collect them (pantas_augment_launch_collect) before the next");
  if (d.ring.empty()) {
    d.ring.resize(LAUNCH_RING);
    for (auto& f : d.ring) for (auto& e : f.ev) HIP_OK(hipEventCreate(&e));
  }
  uint32_t slot = 0;
  while (std::find(d.in_flight.begin(), d.in_flight.end(), slot) != d.in_flight.end()) ++slot;
  d.evs = d.ring[slot].ev;
  try {
    (void)launch_rare_impl(d, b, false);
    d.start_is_rare_end = true;
    (void)launch_main_impl(d, b, false);
  } catch (...) {
    d.evs = d.ev;
    d.start_is_rare_end = false;
    throw;
  }
  d.evs = d.ev;
  d.start_is_rare_end = false;
  d.ring[slot].split = d.last_split;
  d.ring[slot].short_all = d.last_short_all;
  d.in_flight.push_back(slot);
}

size_t launch_collect(DeviceGraph& d, LaunchTimes* out, size_t capacity) {
  // whatever happens below, the ring is free again afterwards: a failed wait used to leave its slots taken for good
  // ("64 launches in flight" from every later launch)
  std::vector<uint32_t> slots;
  slots.swap(d.in_flight);
  HIP_OK(hipSetDevice(d.device));
  HIP_OK(hipStreamSynchronize(d.stream));      // (the second stream's kernels are joined into it by every launch)
  size_t n = 0;
  for (uint32_t slot : slots) {
    const auto& f = d.ring[slot];
    LaunchTimes t{};
    HIP_OK(hipEventElapsedTime(&t.rare_ms, f.ev[0], f.ev[5]));
    HIP_OK(hipEventElapsedTime(&t.main_ms, f.ev[5], f.ev[2]));      // (the main kernels start at the rare-key kernel's end event)
    if (f.split) HIP_OK(hipEventElapsedTime(&t.main_short_ms, f.ev[5], f.ev[3]));
    else if (f.short_all) t.main_short_ms = t.main_ms;
    if (n < capacity && out) out[n] = t;
    ++n;
  }
  return n;
}

// (node << 1 | side) of every node flag the rare kernel has set so far -- what other ranks need
// before THEIR main kernel so that the first-seen order of the common key is recorded everywhere
__global__ void collect_flags_kernel(const uint32_t* word, uint32_t n, uint32_t* out, uint32_t* cursor, uint32_t cap) {
  // one atomic per wave and side (every lane adding to the one cursor would serialise): ballot, the first lane claims room
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * 256u + (threadIdx.x & ~63u); base < n; base += gridDim.x * 256u) {
    const uint32_t i = base + lane;
    const uint32_t w = i < n ? word[i] : 0u;
    for (uint32_t side = 0; side < 2; ++side) {
      const bool has = (w & (side ? FLAG_OL : FLAG_IL)) != 0;
      const unsigned long long m = __ballot(has);
      if (m == 0ull) continue;
      uint32_t start = 0;
      if (lane == 0) start = atomicAdd(cursor, uint32_t(__popcll(m)));
      start = __builtin_amdgcn_readfirstlane(start);
      if (has) {
        const uint32_t o = start + __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u));
        if (o < cap) out[o] = (i << 1) | side;
      }
    }
  }
}

__global__ void apply_flags_kernel(uint32_t* word, uint2* info, uint4* info3, const uint32_t* codes, uint32_t n, uint32_t V) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint32_t c = codes[i];
    if ((c >> 1) < V) {
      atomicOr(&word[c >> 1], (c & 1u) ? FLAG_OL : FLAG_IL);
      atomicOr(&info[c >> 1].x, (c & 1u) ? FLAG_OL : FLAG_IL);
      atomicOr(&info3[c >> 1].x, (c & 1u) ? FLAG_OL : FLAG_IL);
    }
  }
}

void export_flags(DeviceGraph& d, std::vector<uint32_t>& codes) {
  HIP_OK(hipSetDevice(d.device));
  codes.clear();
  if (!d.V) return;
  uint32_t nk = 0;
  HIP_OK(hipMemcpy(&nk, d.small + 2, 4, hipMemcpyDeviceToHost));   // distinct rare keys bound the flag count
  const uint32_t cap = nk + 16;
  // (scratch pool of the graph: hipMalloc / hipFree cost milliseconds each, and this sits between the two kernels of a sharded run)
  uint32_t* out = static_cast<uint32_t*>(scratch_pool(d).take(size_t(cap) * 4));
  HIP_OK(hipMemsetAsync(d.small + 6, 0, 4, d.stream));
  const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(d.V) + 255) / 256, 4096));
  hipLaunchKernelGGL(collect_flags_kernel, dim3(blocks), dim3(256), 0, d.stream, d.word, d.V, out, d.small + 6, cap);
  uint32_t n = 0;
  HIP_OK(hipMemcpyAsync(&n, d.small + 6, 4, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipStreamSynchronize(d.stream));
  n = std::min(n, cap);
  codes.resize(n);
  if (n) HIP_OK(hipMemcpy(codes.data(), out, size_t(n) * 4, hipMemcpyDeviceToHost));
  scratch_pool(d).give(out);
  std::sort(codes.begin(), codes.end());
}

void import_flags(DeviceGraph& d, const uint32_t* codes, size_t n) {
  HIP_OK(hipSetDevice(d.device));
  if (!n) return;
  uint32_t* in = static_cast<uint32_t*>(scratch_pool(d).take(n * 4));
  HIP_OK(hipMemcpyAsync(in, codes, n * 4, hipMemcpyHostToDevice, d.stream));
  const uint32_t blocks = uint32_t(std::min<uint64_t>((n + 255) / 256, 4096));
  hipLaunchKernelGGL(apply_flags_kernel, dim3(blocks), dim3(256), 0, d.stream, d.word, d.info, d.info3, in, uint32_t(n), d.V);
  HIP_OK(hipStreamSynchronize(d.stream));
  scratch_pool(d).give(in);
  d.imported_flags += n;
}

void check_device_status(DeviceGraph& d, uint64_t first_record_line) {
  HIP_OK(hipSetDevice(d.device));
  uint32_t s[8];
  HIP_OK(hipMemcpy(s, d.small, 32, hipMemcpyDeviceToHost));
  const uint64_t st = uint64_t(s[0]) | (uint64_t(s[1]) << 32);
  if (st != ~0ull) {
    const uint64_t line = (st >> 8) + 1;
    const unsigned code = unsigned(st & 0xFF);
    if (code == WALK_OPS_EXHAUSTED)
      throw Error("IndexError", "GAF line " + std::to_string(line) + ": cs ops exhausted before the path end");
    if (line - 1 == first_record_line)
      throw Error("UnboundLocalError", "GAF line " + std::to_string(line) + ": '~' is the first cs op the walk reaches and no line before it left a length behind");
    throw Error("Unsupported", "GAF line " + std::to_string(line) + ": " + TILDE_FIRST_MARK + "; the reference lays it out with a length "
                               "left over from the GAF line before (alignments_augmentation_from_gaf.py:228-233): pantas_augment_run works those out "
                               "for a whole file, a byte range or a batch on its own cannot");
  }
  if (s[3]) throw Error("TableOverflow", "hist");
  if (s[5]) throw Error("TableOverflow", "novel");
}

static void download_table(DeviceGraph& d, Table& t, uint32_t n_keys, std::vector<uint64_t>& keys,
                           std::vector<uint32_t>& cnt, std::vector<uint64_t>& first) {
  keys.resize(n_keys);
  cnt.resize(n_keys);
  first.resize(n_keys);
  if (!n_keys) return;
  ScratchPool& pool = scratch_pool(d);
  uint64_t* dk = static_cast<uint64_t*>(pool.take(size_t(n_keys) * 8));
  uint64_t* df = static_cast<uint64_t*>(pool.take(size_t(n_keys) * 8));
  uint32_t* dc = static_cast<uint32_t*>(pool.take(size_t(n_keys) * 4));
  HIP_OK(hipMemsetAsync(d.small + 6, 0, 4, d.stream));
  const uint32_t blocks = std::min<uint32_t>((t.cap + 255) / 256, 8192);
  hipLaunchKernelGGL(compact_table_kernel, dim3(blocks), dim3(256), 0, d.stream, t.v, dk, dc, df, d.small + 6);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(keys.data(), dk, size_t(n_keys) * 8, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipMemcpyAsync(cnt.data(), dc, size_t(n_keys) * 4, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipMemcpyAsync(first.data(), df, size_t(n_keys) * 8, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipStreamSynchronize(d.stream));
  pool.give(dk); pool.give(df); pool.give(dc);
}

// node-interleaved counters -> the planar arrays the host side works with: nc / il0 / oll [V], and rc [A] with the
// first-slot counts kept in the node slots added in
__global__ void planar_nodes_kernel(const uint4* __restrict__ cnt4, const uint32_t* __restrict__ row_ptr, uint32_t V,
                                    uint32_t* __restrict__ nc, uint32_t* __restrict__ il0, uint32_t* __restrict__ oll, uint32_t* __restrict__ rc_out) {
  for (uint32_t n = blockIdx.x * 256u + threadIdx.x; n < V; n += gridDim.x * 256u) {
    const uint4 c = cnt4[n];
    nc[n] = c.x; il0[n] = c.y; oll[n] = c.z;
    if (c.w) rc_out[row_ptr[n]] += c.w;      // after the copy kernel on the same stream; one node per first slot
  }
}

// ---- the planar counters to the host, the blocks that hold a count only ------------------------------------------------------
// Reads cover exons: at whole-genome scale 3.7 % of the 64-byte blocks of the counters hold a non-zero word, and copying all
// 5 GB of them took 0.19 - 0.27 s of every `pantas augment`.  A bitmap of the non-zero blocks (one wave reads 1 KB and ballots),
// an exclusive scan of its popcounts, the blocks packed side by side: bitmap + packed blocks go to the host (10 + 190 MB) and
// are scattered into zeroed arrays by all threads.
constexpr uint32_t NZ_BLOCK_WORDS = 16;                 // 64 bytes
__global__ __launch_bounds__(256) void nz_bitmap_kernel(const uint4* __restrict__ p, uint64_t bitmap_words, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ cnt) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t(blockIdx.x) * 256u + threadIdx.x) >> 6, n_waves = (uint64_t(gridDim.x) * 256u) >> 6;
  for (uint64_t w = wave; w < bitmap_words; w += n_waves) {      // 32 blocks = 128 pieces of 16 bytes: two loads of 64 lanes
    uint32_t out = 0;
#pragma unroll
    for (uint32_t j = 0; j < 2u; ++j) {
      const uint4 v = p[w * 128u + j * 64u + lane];
      unsigned long long m = __ballot((v.x | v.y | v.z | v.w) != 0u);
      m |= m >> 1; m |= m >> 2;                                   // bit 4 i: block i of this load holds a count
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t i = 0; i < 16u; ++i) bits |= uint32_t((m >> (4u * i)) & 1ull) << i;
      out |= bits << (16u * j);
    }
    if (lane == 0) { bitmap[w] = out; cnt[w] = uint32_t(__popc(out)); }
  }
}
__global__ __launch_bounds__(256) void nz_pack_kernel(const uint4* __restrict__ p, uint64_t bitmap_words, const uint32_t* __restrict__ bitmap,
                                                      const uint32_t* __restrict__ off, uint4* __restrict__ packed) {
  // four lanes per block (a block is four 16-byte pieces), eight blocks in flight per 32 lanes... simply: one thread per piece of a set block
  for (uint64_t w = (uint64_t(blockIdx.x) * 256u + threadIdx.x) >> 2; w < bitmap_words; w += (uint64_t(gridDim.x) * 256u) >> 2) {
    uint32_t bits = bitmap[w];
    if (!bits) continue;
    const uint32_t q = threadIdx.x & 3u;
    uint64_t dst = uint64_t(off[w]) * 4u;
    while (bits) {
      const uint32_t b = uint32_t(__ffs(int(bits))) - 1u;
      bits &= bits - 1u;
      packed[dst + q] = p[(w * 32u + b) * 4u + q];
      dst += 4u;
    }
  }
}

// planar[0 .. words) on the device -> dst(word index) on the host, zero words not moved (the host side must be zero already)
template <class HostWord>
static void sparse_download(DeviceGraph& d, const uint32_t* planar, uint64_t words, HostWord host_word) {
  ScratchPool& pool = scratch_pool(d);
  const uint64_t blocks = (words + NZ_BLOCK_WORDS - 1) / NZ_BLOCK_WORDS, bitmap_words = (blocks + 31) / 32;
  // (planar is allocated padded to whole bitmap words: the kernels read 2 KB per bitmap word)
  uint32_t* bitmap = static_cast<uint32_t*>(pool.take(bitmap_words * 4));
  uint32_t* cnt = static_cast<uint32_t*>(pool.take((bitmap_words + 1) * 4));
  uint32_t* off = static_cast<uint32_t*>(pool.take((bitmap_words + 1) * 4));
  HIP_OK(hipMemsetAsync(cnt + bitmap_words, 0, 4, d.stream));
  const uint32_t grid = uint32_t(std::min<uint64_t>((bitmap_words + 3) / 4, 1u << 16));
  hipLaunchKernelGGL(nz_bitmap_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, d.stream, reinterpret_cast<const uint4*>(planar), bitmap_words, bitmap, cnt);
  HIP_OK(hipGetLastError());
  size_t need = 0;
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt, off, int(bitmap_words + 1), d.stream));
  void* tmp = pool.take(need);
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp, need, cnt, off, int(bitmap_words + 1), d.stream));
  uint32_t total = 0;
  HIP_OK(hipMemcpyAsync(&total, off + bitmap_words, 4, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipStreamSynchronize(d.stream));
  uint4* packed = static_cast<uint4*>(pool.take(std::max<uint64_t>(1, uint64_t(total)) * 64));
  if (total) {
    const uint32_t pgrid = uint32_t(std::min<uint64_t>((bitmap_words * 4 + 255) / 256, 1u << 16));
    hipLaunchKernelGGL(nz_pack_kernel, dim3(pgrid), dim3(256), 0, d.stream, reinterpret_cast<const uint4*>(planar), bitmap_words, bitmap, off, packed);
    HIP_OK(hipGetLastError());
  }
  RawVector<uint32_t> h_bitmap(bitmap_words), h_packed(size_t(total) * NZ_BLOCK_WORDS);
  d2h_staged(h_bitmap.data(), bitmap, bitmap_words * 4, d.stream);
  if (total) d2h_staged(h_packed.data(), packed, size_t(total) * 64, d.stream);
  HIP_OK(hipStreamSynchronize(d.stream));
  pool.give(bitmap); pool.give(cnt); pool.give(off); pool.give(tmp); pool.give(packed);
  // scatter: every thread takes a stretch of bitmap words; its first packed block = popcount of the words in front
  const unsigned T = bitmap_words < (1u << 16) ? 1u : hw_threads();
  std::vector<uint64_t> before(T + 1, 0);
  parallel_chunks(bitmap_words, T, [&](unsigned t, size_t b, size_t e) {
    uint64_t c = 0;
    for (size_t w = b; w < e; ++w) c += uint64_t(__builtin_popcount(h_bitmap[w]));
    before[t + 1] = c;
  });
  for (unsigned t = 0; t < T; ++t) before[t + 1] += before[t];
  parallel_chunks(bitmap_words, T, [&](unsigned t, size_t b, size_t e) {
    const uint32_t* src = h_packed.data() + before[t] * NZ_BLOCK_WORDS;
    for (size_t w = b; w < e; ++w) {
      uint32_t bits = h_bitmap[w];
      while (bits) {
        const uint32_t k = uint32_t(__builtin_ctz(bits));
        bits &= bits - 1u;
        const uint64_t g0 = (uint64_t(w) * 32u + k) * NZ_BLOCK_WORDS;
        for (uint32_t j = 0; j < NZ_BLOCK_WORDS; ++j)
          if (src[j] && g0 + j < words) *host_word(g0 + j) = src[j];
        src += NZ_BLOCK_WORDS;
      }
    }
  });
}

void download_result(DeviceGraph& d, const Graph& g, AugmentResult& r) {
  HIP_OK(hipSetDevice(d.device));
  PhaseTimer pt("download_result");
  const size_t V = d.V;
  const uint64_t words = 3 * uint64_t(V) + d.A;
  const bool sparse = (words >= (1u << 20) || getenv("PANTAS_SPARSE_DOWNLOAD")) && !getenv("PANTAS_DENSE_DOWNLOAD");    // (the second: tests, on small graphs)
  r.nc.resize(V); r.il0.resize(V); r.oll.resize(V); r.rc.resize(d.A);
  if (sparse) {           // zero pages, written (and so touched) by many threads
    par_assign(r.nc, V, 0u); par_assign(r.il0, V, 0u); par_assign(r.oll, V, 0u); par_assign(r.rc, size_t(d.A), 0u);
  } else if (V > (1u << 22)) {   // fresh pages of gigabytes: touched from many threads instead of by the copy engine's single staging loop
    prefault(r.nc.data(), V * 4); prefault(r.il0.data(), V * 4); prefault(r.oll.data(), V * 4); prefault(r.rc.data(), size_t(d.A) * 4);
  }
  pt.lap("host arrays");
  uint32_t* planar = nullptr;       // [nc V | il0 V | oll V | rc A], padded with zeros to whole bitmap words of the sparse download
  const uint64_t padded = (words + 32 * NZ_BLOCK_WORDS - 1) / (32 * NZ_BLOCK_WORDS) * (32 * NZ_BLOCK_WORDS);
  HIP_OK(dmalloc(&planar, std::max<uint64_t>(1, padded) * 4));
  if (padded > words) HIP_OK(hipMemsetAsync(planar + words, 0, (padded - words) * 4, d.stream));
  if (d.A) HIP_OK(hipMemcpyAsync(planar + 3 * V, d.counters + 4 * V, size_t(d.A) * 4, hipMemcpyDeviceToDevice, d.stream));
  if (V) {
    const uint32_t blocks = uint32_t(std::min<uint64_t>((uint64_t(V) + 255) / 256, 1u << 16));
    hipLaunchKernelGGL(planar_nodes_kernel, dim3(blocks), dim3(256), 0, d.stream, reinterpret_cast<const uint4*>(d.counters), d.row_ptr, d.V,
                       planar, planar + V, planar + 2 * V, planar + 3 * V);
    HIP_OK(hipGetLastError());
  }
  if (sparse) {
    uint32_t* nc = r.nc.data(); uint32_t* il0 = r.il0.data(); uint32_t* oll = r.oll.data(); uint32_t* rc = r.rc.data();
    sparse_download(d, planar, words, [=](uint64_t w) { return w < V ? nc + w : w < 2 * V ? il0 + (w - V) : w < 3 * V ? oll + (w - 2 * V) : rc + (w - 3 * V); });
  } else {
    if (V) {
      d2h_staged(r.nc.data(), planar, V * 4, d.stream);
      d2h_staged(r.il0.data(), planar + V, V * 4, d.stream);
      d2h_staged(r.oll.data(), planar + 2 * V, V * 4, d.stream);
    }
    if (d.A) d2h_staged(r.rc.data(), planar + 3 * V, size_t(d.A) * 4, d.stream);
  }
  uint32_t s[8];
  HIP_OK(hipMemcpyAsync(s, d.small, 32, hipMemcpyDeviceToHost, d.stream));
  HIP_OK(hipStreamSynchronize(d.stream));
  (void)hipFree(planar);
  pt.lap(sparse ? "planar counters to the host (non-zero blocks)" : "planar counters to the host");
  std::vector<uint64_t> keys, first;
  std::vector<uint32_t> cnt;
  download_table(d, d.hist, s[2], keys, cnt, first);
  r.hist.clear();
  r.hist.reserve(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) {
    const uint32_t node = uint32_t(keys[i] >> 32);
    const uint32_t side = uint32_t(keys[i] >> 31) & 1u;
    const uint32_t pc = uint32_t(keys[i]) & 0x7FFFFFFFu;
    const bool common = pc == COMMON_POS;
    r.hist.push_back(HistEntry{node, side, common ? 0 : int32_t(pc) - 1, cnt[i], first[i], common});
  }
  download_table(d, d.novel, s[4], keys, cnt, first);
  r.novel.clear();
  r.novel.reserve(keys.size());
  for (size_t i = 0; i < keys.size(); ++i)
    r.novel.push_back(NovelLink{uint32_t(keys[i] >> 32), uint32_t(keys[i]), cnt[i], first[i]});
  (void)g;
  pt.lap("sparse tables to the host");
  finalize_result(r);
  pt.lap("first-seen order of the sparse entries");
}

void set_track_touched(DeviceGraph& d, bool on) {
  if (on == d.track_touched) return;
  d.track_touched = on;
  d.touched_valid = false;          // (kept from the next zeroing of the counters on)
}
TouchedBitmap touched_bitmap_of(DeviceGraph& d) {
  TouchedBitmap t{};
  if (d.track_touched && d.touched_valid) { t.bits = d.touched; t.words = d.touched_words; t.block_words = TOUCHED_BLOCK_WORDS; }
  return t;
}

CounterPtrs counter_ptrs(DeviceGraph& d) {
  CounterPtrs p{};
  p.base = d.counters;
  p.words = 4 * uint64_t(d.V) + d.A;
  p.V = d.V;
  p.A = d.A;
  return p;
}

}  // namespace pantas
