// GAF text -> alignment records ON THE GPU: the step in front of the augment kernels.
// The text of one byte range goes to HBM once (pinned double-buffered copies); everything after that --
// line index, column split, mapq / unmapped / divergence filters, cs parsing, clipping, path split with the
// name -> node lookup, locus bucketing -- runs as kernels, and the resulting DeviceBatch is what
// augment.hip consumes.  Restates scripts/alignments_augmentation_from_gaf.py:142-197, :10-50 through
// gaf_device.h; gaf.cpp is the host form of the same lines and the two are checked against each other
// (tests/hostsim/hostsim_tokenize.cpp on the CPU, tests/test_gpu_tokenize.py on the GPU).
//
// Passes (all HBM-bound byte / integer work; T = text bytes, N = lines, n = kept records):
//   newline count per 16 KB tile -> exclusive scan -> line starts            reads T twice, writes 8 N
//   measure : thread per line, parse_gaf_line<CountOnly>                      reads T (through L2), writes 13 N
//   limit + totals, three exclusive scans (record index, node offset, op offset)
//   write   : thread per kept line, parse again, emit at the scanned offsets  reads T, writes the batch
//   bucket  : key = lowest end node, radix sort (rocPRIM), short-walk records first inside every 2048-block,
//             sizes -> scans -> gather                                        a few passes over the batch
// A line the reference raises on is found by the measure pass (atomicMin over (line, code)); the records of
// the lines before it are still produced, and the host re-reads that one line to word the error.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <atomic>
#include <thread>
#include <unistd.h>

#include "augment.h"
#include "device_batch.h"
#include "gaf_device.h"
#include "graph.h"

namespace pantas {

namespace {

constexpr uint32_t TILE = 16384;   // text bytes per workgroup of the line-index kernels
constexpr uint32_t TPB = 256;
constexpr uint32_t SORT_BLOCK = 2048;

struct DevBuf {
  ScratchPool* pool = nullptr;
  void* p = nullptr;
  DevBuf() = default;
  explicit DevBuf(ScratchPool& sp) : pool(&sp) {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void alloc(size_t bytes) {
    release();
    p = pool->take(bytes);
  }
  void release() { if (p) { pool->give(p); p = nullptr; } }
  // the block becomes part of the batch handed to the caller (freed with hipFree by its deleter)
  template <class T> T* hand_over() { T* q = static_cast<T*>(p); pool->disown(p); p = nullptr; return q; }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// 0x80 in every byte of w that equals '\n' (exact, no borrow artefacts)
__device__ __forceinline__ uint32_t newline_bits(uint32_t w) {
  const uint32_t x = w ^ 0x0A0A0A0Au;
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
  const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive prefix of v over the 256 threads of the block; *total = block sum
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wave_sum[TPB / 64];
  const uint32_t inc = wave_inclusive_scan(v);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (lane == 63) wave_sum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (uint32_t w = 0; w < TPB / 64; ++w) {
    const uint32_t s = wave_sum[w];
    if (w < wave) base += s;
    all += s;
  }
  *total = all;
  return base + inc - v;
}

// 0x80 in every byte of w that equals '\r'
__device__ __forceinline__ uint32_t cr_bits(uint32_t w) {
  const uint32_t x = w ^ 0x0D0D0D0Du;
  const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | x | 0x7F7F7F7Fu);
}

// text is padded with zero bytes to a multiple of TILE plus one tile, so the loads (and the one word a lane reads behind its 64 bytes)
// need no bound.  *lone_cr is set when the text holds a '\r' that is followed by a byte other than '\n' (the reference's universal
// newlines end a line there; to the kernels it is whitespace: gaf_device.h TE_LONE_CR) -- found here, where every byte is read anyway,
// since late round 4: the host looked for it in the pinned pieces on their way to HBM, and that scan, not PCIe, set the pace (40 GB/s
// against 54 without it).
__global__ void __launch_bounds__(TPB) count_newlines_kernel(const uint4* text, uint32_t* tile_cnt, uint32_t* lone_cr, uint64_t size) {
  const size_t byte0 = size_t(blockIdx.x) * TILE + threadIdx.x * 64u;
  const uint4* p = text + byte0 / 16u;
  uint32_t c = 0;
  uint32_t w[17];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint4 v = p[j];
    w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
  }
  w[16] = reinterpret_cast<const uint32_t*>(p + 4)[0];
  bool lone = false;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t nl = newline_bits(w[j]);
    c += __popc(nl);
    // byte i of the word is a '\r' and byte i + 1 (of this word or the first of the next) is no '\n'
    uint32_t m = cr_bits(w[j]) & ~((nl >> 8) | (newline_bits(w[j + 1]) << 24));
    while (m) {      // (rare: a text with '\r' in it at all)
      const uint32_t bit = uint32_t(__ffs(int(m))) - 1u;
      if (byte0 + uint32_t(j) * 4u + (bit >> 3) + 1u < size) lone = true;      // (a '\r' as the text's last byte ends nothing)
      m &= m - 1u;
    }
  }
  if (lone) atomicOr(lone_cr, 1u);
  uint32_t total;
  (void)block_exclusive_scan(c, &total);
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

// line i = [ls[i], ls[i+1] - 1)
__global__ void __launch_bounds__(TPB) line_starts_kernel(const uint4* text, const uint32_t* tile_off, uint64_t* ls) {
  const size_t byte0 = size_t(blockIdx.x) * TILE + threadIdx.x * 64u;
  const uint4* p = text + byte0 / 16u;
  uint32_t w[16];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint4 v = p[j];
    w[4 * j] = newline_bits(v.x); w[4 * j + 1] = newline_bits(v.y); w[4 * j + 2] = newline_bits(v.z); w[4 * j + 3] = newline_bits(v.w);
    c += __popc(w[4 * j]) + __popc(w[4 * j + 1]) + __popc(w[4 * j + 2]) + __popc(w[4 * j + 3]);
  }
  uint32_t total;
  uint32_t rank = tile_off[blockIdx.x] + block_exclusive_scan(c, &total);
  if (c == 0) return;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    uint32_t m = w[j];
    while (m) {
      const uint32_t bit = uint32_t(__ffs(int(m))) - 1u;
      ls[size_t(rank) + 1] = byte0 + uint32_t(j) * 4u + (bit >> 3) + 1u;
      ++rank;
      m &= m - 1u;
    }
  }
}

// One wave per workgroup, one line per lane.  The wave first copies the contiguous text of its 64 lines into LDS with
// 16-byte coalesced loads; the byte-at-a-time scans of parse_gaf_line then run at LDS latency instead of L2 latency
// (the parse is a chain of dependent byte loads).  A group whose text exceeds the staging buffer (very long lines)
// parses straight from HBM.
constexpr uint32_t LPB = 64;

// Lines are taken in GROUPS of 64; a workgroup (one wave) walks the groups g, g + gridDim.x, ... -- one workgroup per group
// was 781 000 single-wave workgroups with 12 KB of LDS each at whole-genome scale, and the kernels' run time was the time it
// takes to LAUNCH that many (27.45 ms whatever the code in them did; profiles/r3_tokenizer_steps.txt).
template <class Body>
__device__ __forceinline__ void staged_group(uint32_t group, const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, Body body) {
  extern __shared__ uint4 stage4[];
  const uint32_t first = group * LPB;
  const uint32_t last = min(first + LPB, n_lines);
  const uint64_t sb = ls[first], se = ls[last];   // se may be size + 1: the text is zero-padded
  const uint64_t a0 = sb & ~uint64_t(15);
  const uint64_t a1 = (se + 15) & ~uint64_t(15);
  const uint32_t i = first + threadIdx.x;
  if (a1 - a0 <= uint64_t(stage_bytes)) {
    const uint4* src = reinterpret_cast<const uint4*>(text + a0);
    const uint32_t n16 = uint32_t((a1 - a0) >> 4);
    // Eight 16-byte loads per lane in flight before the first one is stored: `stage4[k] = src[k]` in a plain loop is load, wait,
    // store, next -- ten trips to HBM one after the other per wave, which is where these kernels spent their time (their run time
    // did not move when the parse lost nine tenths of its instructions).  The loads are unconditional (index clamped): a load
    // that may not have been issued makes the compiler wait for all of them.
    constexpr uint32_t STAGE_UNROLL = 8;
    for (uint32_t k0 = threadIdx.x; k0 < n16; k0 += LPB * STAGE_UNROLL) {
      uint4 v[STAGE_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < STAGE_UNROLL; ++u) v[u] = src[min(k0 + u * LPB, n16 - 1u)];
#pragma unroll
      for (uint32_t u = 0; u < STAGE_UNROLL; ++u)
        if (k0 + u * LPB < n16) stage4[k0 + u * LPB] = v[u];
    }
    __syncthreads();
    // (the parser reads its line through aligned 32-bit words of the staged text: gaf_device.h TextWords)
    if (i < last) body(i, reinterpret_cast<const uint32_t*>(stage4), uint32_t(ls[i] - a0), uint32_t(ls[i + 1] - 1 - a0));
  } else if (i < last) {
    const uint64_t lb = ls[i], base = lb & ~uint64_t(3);
    body(i, reinterpret_cast<const uint32_t*>(text + base), uint32_t(lb - base), uint32_t(ls[i + 1] - 1 - base));
  }
}
// The same for a kernel that carries ONE copy of its line parser: the text is always staged.  A group whose 64 lines do not fit the
// staging buffer goes through it in rounds of as many leading lines as fit; a single line longer than the buffer is reported
// through `too_long` (the batch then takes the two-pass kernels, whose parser also reads from HBM) and skipped.
template <class Body>
__device__ __forceinline__ void staged_rounds(uint32_t group, const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, uint32_t* too_long,
                                              Body body) {
  extern __shared__ uint4 stage4[];
  const uint32_t first = group * LPB;
  const uint32_t last = min(first + LPB, n_lines);
  const uint32_t i = first + threadIdx.x;
  const uint64_t my_b = i < last ? ls[i] : 0, my_e = i < last ? ls[i + 1] : 0;      // my line and its '\n' (or the text's end + 1)
  uint32_t next = first;                   // first line not yet handled
  while (next < last) {
    const uint64_t a0 = __shfl(my_b, int(next - first), LPB) & ~uint64_t(15);
    // lines next .. that end inside the buffer (line ends grow: the lanes that say yes are a prefix of the rest)
    const bool fits = i >= next && i < last && ((my_e + 15) & ~uint64_t(15)) - a0 <= uint64_t(stage_bytes);
    const uint32_t take = uint32_t(__popcll(__ballot(fits)));
    if (take == 0) {                       // line `next` alone is longer than the buffer
      if (threadIdx.x == 0) atomicOr(too_long, 1u);
      ++next;
      continue;
    }
    const uint64_t a1 = (__shfl(my_e, int(next + take - 1u - first), LPB) + 15) & ~uint64_t(15);
    const uint4* src = reinterpret_cast<const uint4*>(text + a0);
    const uint32_t n16 = uint32_t((a1 - a0) >> 4);
    constexpr uint32_t STAGE_UNROLL = 8;
    for (uint32_t k0 = threadIdx.x; k0 < n16; k0 += LPB * STAGE_UNROLL) {
      uint4 v[STAGE_UNROLL];
#pragma unroll
      for (uint32_t u = 0; u < STAGE_UNROLL; ++u) v[u] = src[min(k0 + u * LPB, n16 - 1u)];
#pragma unroll
      for (uint32_t u = 0; u < STAGE_UNROLL; ++u)
        if (k0 + u * LPB < n16) stage4[k0 + u * LPB] = v[u];
    }
    __syncthreads();
    if (fits) body(i, reinterpret_cast<const uint32_t*>(stage4), uint32_t(my_b - a0), uint32_t(my_e - 1 - a0));
    __syncthreads();                       // the staged text is overwritten by the next round / group
    next += take;
  }
}

template <class Body>
__device__ __forceinline__ void staged_lines(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, Body body) {
  const uint32_t groups = (n_lines + LPB - 1u) / LPB;
  for (uint32_t g = blockIdx.x; g < groups; g += gridDim.x) {
    staged_group(g, text, ls, n_lines, stage_bytes, body);
    __syncthreads();        // the staged text is overwritten by the next group
  }
}
inline uint32_t line_group_grid(uint32_t n_lines) { return std::min<uint32_t>((n_lines + LPB - 1u) / LPB, 256u * 16u); }

__global__ void __launch_bounds__(LPB) measure_kernel(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, int thr, uint32_t check_cr,
                                                      DenseNames names, uint32_t* kept, uint32_t* cnt_nodes, uint32_t* cnt_ops, uint8_t* meta,
                                                      unsigned long long* err) {
  staged_lines(text, ls, n_lines, stage_bytes, [&](uint32_t i, const uint32_t* words, uint32_t b, uint32_t e) {
    CountOnly co;
    LineSummary r;
    TextWords T(words);
    parse_gaf_line_words(T, b, e, thr, names, co, r, check_cr != 0u);
    kept[i] = r.status == TOK_KEEP ? 1u : 0u;
    cnt_nodes[i] = r.n_nodes;
    cnt_ops[i] = r.n_ops;
    meta[i] = uint8_t(r.status | ((r.flags & LINE_HAS_DEL) ? 4u : 0u));
    if (r.status == TOK_ERROR) atomicMin(err, (static_cast<unsigned long long>(i) << 8) | r.err);
  });
}

// totals[0..6] = kept, nodes, ops, rejected, skipped, del lines, del nodes over the lines before the first raising one;
// the counts of that line and of the lines after it are zeroed so that the scans ignore them
__global__ void __launch_bounds__(TPB) limit_and_totals_kernel(uint32_t n_lines, const unsigned long long* err, uint32_t* kept, uint32_t* cnt_nodes,
                                                               uint32_t* cnt_ops, const uint8_t* meta, unsigned long long* totals) {
  const unsigned long long e = *err;
  const uint64_t lim = e == ~0ull ? uint64_t(n_lines) : uint64_t(e >> 8);
  unsigned long long t[7] = {0, 0, 0, 0, 0, 0, 0};
  for (uint64_t i = uint64_t(blockIdx.x) * TPB + threadIdx.x; i < n_lines; i += uint64_t(gridDim.x) * TPB) {
    if (i >= lim) { kept[i] = 0; cnt_nodes[i] = 0; cnt_ops[i] = 0; continue; }
    const uint32_t m = meta[i];
    t[0] += kept[i]; t[1] += cnt_nodes[i]; t[2] += cnt_ops[i];
    t[3] += (m & 3u) == TOK_REJECT; t[4] += (m & 3u) == TOK_SKIP;
    if (m & 4u) { t[5] += 1; t[6] += cnt_nodes[i]; }
  }
  __shared__ unsigned long long red[7];
  if (threadIdx.x < 7) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    unsigned long long v = t[k];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&red[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 7 && red[threadIdx.x]) atomicAdd(&totals[threadIdx.x], red[threadIdx.x]);
}

struct WriteOut {
  uint32_t* nodes;
  uint32_t* ops;
  __device__ __forceinline__ void node(uint32_t i, uint32_t idx) { nodes[i] = idx; }
  __device__ __forceinline__ void op(uint32_t i, uint32_t w) { ops[i] = w; }
  __device__ __forceinline__ void ops_done(uint32_t) {}
  __device__ __forceinline__ void nodes_done(uint32_t) {}
};

__global__ void __launch_bounds__(LPB) write_kernel(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, int thr, uint32_t check_cr,
                                                    DenseNames names, const uint32_t* kept, const uint32_t* rec_of, const uint32_t* node_scan,
                                                    const uint32_t* op_scan, uint32_t first_line, LineHdr* hdr, uint32_t* node_off,
                                                    uint32_t* op_off, uint32_t* nodes, uint32_t* ops) {
  staged_lines(text, ls, n_lines, stage_bytes, [&](uint32_t i, const uint32_t* words, uint32_t b, uint32_t e) {
    if (!kept[i]) return;
    const uint32_t rec = rec_of[i], no = node_scan[i], oo = op_scan[i];
    WriteOut wo{nodes + no, ops + oo};
    LineSummary r;
    TextWords T(words);
    parse_gaf_line_words(T, b, e, thr, names, wo, r, check_cr != 0u);
    hdr[rec] = LineHdr{r.start, r.tail, first_line + i, r.flags};
    node_off[rec] = no;
    op_off[rec] = oo;
  });
}


// ================================================================================================================
// ONE parse per line (round 3).  The two-pass form above reads and parses the text twice (count, scan, write) and then
// gathers every record from three file-order arrays through a permutation: 25 + 25 + 15 ms for 7.4 GB / 50 M lines.
// Here a line is parsed once and its record -- {LineHdr, k, c, nodes, ops}, PROV_HDR + k + c words -- is written straight
// into a word array CONGRUENT WITH THE TEXT: the record of the line that starts at byte b lives at word ceil(b / 4), and a
// line of L bytes has L / 4 words of room, which a GAF line never fills (a path node costs its name and a '>' in text, a cs op
// two or three bytes, the twelve mandatory columns pay for the header).  No offsets have to be known before the parse, so there
// is no count pass and no scan.  A line that does not fit (names of one or two characters) takes room from a small arena behind
// the array by one atomic and is parsed a second time into it; when the arena runs out, or a line makes the reference raise, the
// batch falls back to the two-pass form (which also words the error).
// The sort that buckets the records by locus carries, per line, a 64-bit value {record address, k, c, class bits}: everything
// the later passes need to size and order the batch sits in the sorted values, and the record itself is read exactly once, by
// the gather that writes the batch (and its `rec` entries).
constexpr uint32_t PROV_HDR = 6;                   // start, tail, line id, flags, k, c
constexpr uint64_t VAL_K_SHIFT = 32, VAL_C_SHIFT = 44, VAL_DEL = 1ull << 56, VAL_STARS = 1ull << 57, VAL_INDEL = 1ull << 58, VAL_BIG = 1ull << 59;
constexpr uint32_t VAL_SMALL_MAX = 4095;           // k and c ride in 12 bits each; beyond that VAL_BIG says "read them from the record"
constexpr uint32_t N_TOTALS = 10;                  // kept, nodes, ops, rejected, skipped, del lines, del nodes, short records, short nodes, short ops
constexpr uint32_t N_PARTIAL = N_TOTALS + 1;       // ... and, in a wave's row of partial sums, its first kept line (or NONE)

struct ProvOut {
  uint32_t* w;          // the record (global memory): header, nodes, ops
  uint32_t cap;         // words behind the header
  uint32_t k = 0, first = 0, last = 0;
  bool ovf = false;
  __device__ __forceinline__ void node(uint32_t i, uint32_t idx) {
    if (i == 0) first = idx;
    last = idx;
    if (i < cap) w[PROV_HDR + i] = idx; else ovf = true;
  }
  __device__ __forceinline__ void nodes_done(uint32_t n) { k = n; }
  __device__ __forceinline__ void op(uint32_t i, uint32_t x) {
    const uint32_t j = k + i;
    if (j < cap) w[PROV_HDR + j] = x; else ovf = true;
  }
};

__global__ void __launch_bounds__(LPB) __attribute__((amdgpu_waves_per_eu(4, 4))) parse_once_kernel(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, int thr, uint32_t check_cr,
                                                         DenseNames names, uint32_t first_line, uint32_t bits, uint32_t* prov, uint64_t prov_words,
                                                         uint32_t* arena_next, uint32_t arena_words, uint32_t* key, unsigned long long* val, uint32_t* partial,
                                                         unsigned long long* err, uint32_t* fallback, uint32_t stop, uint32_t align_min, uint32_t fast) {
  uint32_t t[N_TOTALS];
#pragma unroll
  for (uint32_t j = 0; j < N_TOTALS; ++j) t[j] = 0;
  uint32_t my_kept_line = 0xFFFFFFFFu;
  const uint32_t groups = (n_lines + LPB - 1u) / LPB;
  for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x)
  staged_rounds(grp, text, ls, n_lines, stage_bytes, fallback, [&](uint32_t i, const uint32_t* words, uint32_t b, uint32_t e) {
    const uint64_t lb = ls[i], le = ls[i + 1];
    const uint64_t slot = (lb + 3) >> 2, slot_end = le >> 2;
    const uint32_t room = slot_end > slot ? uint32_t(slot_end - slot) : 0u;
    if (stop == 9u) { key[i] = words[b >> 2]; return; }        // (ablation: staging only)
    // ONE call site of the parser (it is thousands of instructions long): a line whose record does not fit its own text is parsed
    // a second time by the same code, into room taken from the arena
    // (A/B switch, off by default: the record starts at the next 64-byte boundary inside the line's own words when that leaves it
    // align_min words, so that the gather, which reads the records in locus order, i.e. at random, touches one sector per record
    // instead of two; a record that does not fit there is parsed again into the whole of the line's words, one that does not fit
    // those into the arena.)
    const uint64_t slot_al = (slot + 15u) & ~uint64_t(15);
    const uint32_t room_al = slot_end > slot_al ? uint32_t(slot_end - slot_al) : 0u;
    const bool aligned = room_al >= align_min;
    uint64_t at = aligned ? slot_al : slot;
    uint32_t cur = aligned ? room_al : room;
    uint32_t cap = cur > PROV_HDR ? cur - PROV_HDR : 0u;
    LineSummary r;
    uint32_t first_node = 0, last_node = 0;
    bool ok = true;
#pragma unroll 1
    for (uint32_t pass = 0; pass < 2u; ++pass) {
      ProvOut po{prov + at, cap};
      TextWords T(words);
      // round 5: the form a mapper writes is taken by the lean parser (gaf_device.h parse_gaf_line_fast: a third of the vector
      // instructions); whatever it refuses -- other whitespace, signed integers, '=' / '~' ops, a line the reference raises on ...
      // -- is parsed by the general one, in this lane, as before (`fast` = 0: every line by the general parser, the A/B partner)
#ifdef PANTAS_EXPERIMENTS
      if (stop >= 10u) { (void)parse_gaf_line_fast(T, b, e, thr, names, po, r, false, stop - 10u); break; }      // (timing only: stages of the lean parser)
#endif
      if (!(fast && parse_gaf_line_fast(T, b, e, thr, names, po, r, check_cr != 0u))) {
        po = ProvOut{prov + at, cap};
        T = TextWords(words);
        parse_gaf_line_words(T, b, e, thr, names, po, r, check_cr != 0u && pass == 0u, stop);
      }
      first_node = po.first; last_node = po.last;
      if (pass == 1u || r.status != TOK_KEEP || !(po.ovf || cur < PROV_HDR)) break;
      const uint32_t need = PROV_HDR + r.n_ops + r.n_nodes;
      if (need <= room) { at = slot; cur = room; cap = room - PROV_HDR; continue; }      // (only an aligned first try gets here)
      const uint32_t pos = atomicAdd(arena_next, need);
      if (need > arena_words || pos > arena_words - need) { atomicOr(fallback, 1u); ok = false; break; }
      at = prov_words + pos;
      cur = need;
      cap = need - PROV_HDR;
    }
    if (stop) { key[i] = uint32_t(r.start) + r.status; return; }
    uint32_t k32 = 2u << bits;           // not a record: sorts behind both classes
    unsigned long long v = 0;
    if (r.status == TOK_ERROR) atomicMin(err, (static_cast<unsigned long long>(i) << 8) | r.err);
    else if (r.status == TOK_REJECT) t[3] += 1;
    else if (r.status == TOK_SKIP) t[4] += 1;
    else if (ok) {
      uint32_t* w = prov + at;
      w[0] = uint32_t(r.start); w[1] = uint32_t(r.tail); w[2] = first_line + i; w[3] = r.flags; w[4] = r.n_nodes; w[5] = r.n_ops;
      const bool is_short = short_walk_class(r.flags, r.n_nodes);
      k32 = ((is_short ? 0u : 1u) << bits) | (r.n_nodes ? min(first_node, last_node) : 0u);
      const bool big = r.n_nodes > VAL_SMALL_MAX || r.n_ops > VAL_SMALL_MAX;
      v = static_cast<unsigned long long>(at) | (static_cast<unsigned long long>(big ? 0u : r.n_nodes) << VAL_K_SHIFT) |
          (static_cast<unsigned long long>(big ? 0u : r.n_ops) << VAL_C_SHIFT) | ((r.flags & LINE_HAS_DEL) ? VAL_DEL : 0ull) |
          ((r.flags & LINE_STARS) ? VAL_STARS : 0ull) | ((r.flags & LINE_ONE_INDEL) ? VAL_INDEL : 0ull) | (big ? VAL_BIG : 0ull);
      t[0] += 1; t[1] += r.n_nodes; t[2] += r.n_ops;
      my_kept_line = min(my_kept_line, i);
      if (r.flags & LINE_HAS_DEL) { t[5] += 1; t[6] += r.n_nodes; }
      if (is_short) { t[7] += 1; t[8] += r.n_nodes; t[9] += r.n_ops; }
    }
    key[i] = k32;
    val[i] = v;
  });
  // what this wave's lines add to the batch totals: one row of partial sums per wave, summed by totals_kernel (a wave walks
  // n_lines / 64 / gridDim.x groups: 32-bit sums hold as long as a batch has fewer than 2^32 path nodes, which it must)
#pragma unroll
  for (uint32_t j = 0; j < N_TOTALS; ++j) {
    uint32_t x = t[j];
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if (threadIdx.x == 0) partial[size_t(blockIdx.x) * N_PARTIAL + j] = x;
  }
  for (int d = 32; d > 0; d >>= 1) my_kept_line = min(my_kept_line, uint32_t(__shfl_down(my_kept_line, d, 64)));
  if (threadIdx.x == 0) partial[size_t(blockIdx.x) * N_PARTIAL + N_TOTALS] = my_kept_line;
}

__global__ void __launch_bounds__(TPB) totals_kernel(const uint32_t* partial, uint32_t rows, unsigned long long* totals) {
  unsigned long long t[N_TOTALS];
#pragma unroll
  for (uint32_t j = 0; j < N_TOTALS; ++j) t[j] = 0;
  uint32_t first_kept = 0xFFFFFFFFu;
  for (uint64_t r = uint64_t(blockIdx.x) * TPB + threadIdx.x; r < rows; r += uint64_t(gridDim.x) * TPB) {
#pragma unroll
    for (uint32_t j = 0; j < N_TOTALS; ++j) t[j] += partial[r * N_PARTIAL + j];
    first_kept = min(first_kept, partial[r * N_PARTIAL + N_TOTALS]);
  }
  if (first_kept != 0xFFFFFFFFu) atomicMin(&totals[N_TOTALS], static_cast<unsigned long long>(first_kept));
  __shared__ unsigned long long red[N_TOTALS];
  if (threadIdx.x < N_TOTALS) red[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t j = 0; j < N_TOTALS; ++j) {
    unsigned long long x = t[j];
    for (int d = 32; d > 0; d >>= 1) x += __shfl_down(x, d, 64);
    if ((threadIdx.x & 63u) == 0 && x) atomicAdd(&red[j], x);
  }
  __syncthreads();
  if (threadIdx.x < N_TOTALS && red[threadIdx.x]) atomicAdd(&totals[threadIdx.x], red[threadIdx.x]);
}

__device__ __forceinline__ void val_sizes(unsigned long long v, const uint32_t* prov, uint32_t& k, uint32_t& c) {
  if (v & VAL_BIG) { const uint32_t* w = prov + (v & 0xFFFFFFFFull); k = w[4]; c = w[5]; }
  else { k = uint32_t(v >> VAL_K_SHIFT) & 0xFFFu; c = uint32_t(v >> VAL_C_SHIFT) & 0xFFFu; }
}

// inside every block of 2048 records of a class: by walk.h block_order_key, stable (the values carry what the key is made of)
__global__ void __launch_bounds__(TPB) order_blocks2_kernel(uint32_t n, const unsigned long long* in, const uint32_t* prov, unsigned long long* out) {
  __shared__ uint32_t sh[SORT_BLOCK];
  const uint32_t a = blockIdx.x * SORT_BLOCK;
  for (uint32_t j = threadIdx.x; j < SORT_BLOCK; j += TPB) {
    uint32_t x = 0xFFFFFFFFu;
    if (a + j < n) {
      const unsigned long long v = in[a + j];
      uint32_t k, c;
      val_sizes(v, prov, k, c);
      const uint32_t ok = ((v & VAL_INDEL) ? 0x20000u : 0u) | ((v & VAL_STARS) ? 0x10000u : 0u) | (k < 0xFFFFu ? k : 0xFFFFu);   // = block_order_key
      x = (ok << 11) | j;
    }
    sh[j] = x;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= SORT_BLOCK; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < SORT_BLOCK; i += TPB) {
        const uint32_t o = i ^ j;
        if (o > i) {
          const uint32_t x = sh[i], y = sh[o];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { sh[i] = y; sh[o] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t j = threadIdx.x; j < SORT_BLOCK; j += TPB)
    if (a + j < n) out[a + j] = in[a + (sh[j] & (SORT_BLOCK - 1u))];
}

__global__ void __launch_bounds__(TPB) sizes2_kernel(uint32_t n, const unsigned long long* val, const uint32_t* prov, uint32_t* nn, uint32_t* no, uint32_t* del) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long v = val[i];
  uint32_t k, c;
  val_sizes(v, prov, k, c);
  nn[i] = k; no[i] = c; del[i] = (v & VAL_DEL) ? 1u : 0u;
}

// the batch in its final order, every record read from where its line left it; its `rec` entry with it.  A workgroup takes 256
// consecutive records of the order: every lane fetches one record's header (six words of one sector, its LineHdr goes out as
// one 16-byte store), then the nodes and the ops of the 256 records are moved OUTPUT-driven -- lane t writes words t, t + 256, ...
// of the workgroup's contiguous stretch of nodes[] / ops[] and finds the record a word belongs to by a binary search in the 256
// offsets (LDS): coalesced stores, and every load of a lane is independent of its other loads, so a lane keeps several trips
// to memory in flight (one lane per record made fifteen dependent ones).
// Round 4: with `place` (device_batch.h; null = no tiles) the same pass writes the tiles of the short-walk class (records below
// n_short) for the position-parallel kernel: every node word goes out a second time, to its lane of its tile, with its position word;
// a thread per record writes the 16-byte line record, ORs TILE_NOT_PLAIN into its tile's word and, for the first record of a tile,
// does what tile_head does.  (tile_fill_kernel of augment.hip makes the same arrays from a finished batch.)
struct TileOut {
  const uint32_t* place;
  const uint32_t* chunk_tile;
  uint32_t* tile_first;
  uint32_t* pnodes;
  uint16_t* pmeta;
  PosRec* prec;
  uint32_t n_short;
};
__global__ void __launch_bounds__(TPB) gather2_kernel(uint32_t n, const unsigned long long* val, const uint32_t* prov, const uint32_t* node_off,
                                                      const uint32_t* op_off, const uint32_t* del_scan, LineHdr* hdr, uint32_t* nodes, uint32_t* ops,
                                                      uint32_t* del_idx, uint4* rec, TileOut to) {
  __shared__ uint32_t s_slot[TPB], s_k[TPB], s_noff[TPB + 1], s_ooff[TPB + 1];
  __shared__ uint32_t s_tpos[TPB];       // tile * 64 + first lane of the record (0xFFFFFFFF: no tile)
  __shared__ uint8_t s_ord[TPB];         // the record's ordinal in its tile
  // Round 5: the first GATHER_STAGE words of every record are read ONCE, by the record's own lane, into LDS (a record of the synthetic
  // workloads is 14 words: header 6, six path nodes, two ops), and the word-wise moves below take them from there.  The records stand at
  // random places of an array as large as the text: read three times -- header, then node and op words one by one, then again for the
  // op summary -- every record dragged its one or two 128-byte lines from memory again and again (24 GB of read requests for 2.8 GB of
  // records by the L2's counters); what a workgroup's 256 records touch is 32 KB, more than its share of an L2 that 128 workgroups use.
  constexpr uint32_t GATHER_STAGE = 16, GATHER_STRIDE = GATHER_STAGE + 1u;     // (the odd stride spreads the lanes' rows over the banks)
  __shared__ uint32_t s_rec[TPB * GATHER_STRIDE];
  const uint32_t r0 = blockIdx.x * TPB, t = threadIdx.x, i = r0 + t;
  const uint32_t cnt = min(TPB, n - r0);
  uint32_t slot = 0, k = 0, c = 0, flags = 0;
  if (t < cnt) {
    slot = uint32_t(val[i] & 0xFFFFFFFFull);
    const uint32_t* w = prov + slot;
    uint32_t head[GATHER_STAGE];
#pragma unroll
    for (uint32_t j = 0; j < GATHER_STAGE; ++j) head[j] = w[j];       // (behind a short record: words of the next line's room, or of the arena, or of the sixteen spare words behind it)
#pragma unroll
    for (uint32_t j = 0; j < GATHER_STAGE; ++j) s_rec[t * GATHER_STRIDE + j] = head[j];
    const uint32_t w0 = head[0], w1 = head[1], w2 = head[2];
    flags = head[3]; k = head[4]; c = head[5];
    hdr[i] = LineHdr{int32_t(w0), int32_t(w1), w2, flags};
    s_slot[t] = slot;
    s_k[t] = k;
    s_noff[t] = node_off[i];
    s_ooff[t] = op_off[i];
    uint32_t tpos = 0xFFFFFFFFu, ord = 0;
    if (to.place && i < to.n_short) {
      const uint32_t pl = to.place[i];
      tpos = (to.chunk_tile[i / TILE_CHUNK] + place_tile(pl)) * 64u + place_lane(pl);
      ord = place_ord(pl);
    }
    s_tpos[t] = tpos;
    s_ord[t] = uint8_t(ord);
  }
  if (t == 0) { s_noff[cnt] = node_off[r0 + cnt]; s_ooff[cnt] = op_off[r0 + cnt]; }
  __syncthreads();
  // record of output word p: the last one whose offset is <= p (records without nodes / ops share an offset with their successor)
  auto owner = [&](const uint32_t* off, uint32_t p) {
    uint32_t lo = 0, hi = cnt;      // off[lo] <= p < off[hi]
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  };
  for (uint32_t p = s_noff[0] + t; p < s_noff[cnt]; p += TPB) {
    const uint32_t r = owner(s_noff, p), j = p - s_noff[r];
    const uint32_t id = PROV_HDR + j < GATHER_STAGE ? s_rec[r * GATHER_STRIDE + PROV_HDR + j] : prov[s_slot[r] + PROV_HDR + j];
    nodes[p] = id;
    const uint32_t tpos = s_tpos[r];
    if (tpos != 0xFFFFFFFFu) { to.pnodes[size_t(tpos) + j] = id; to.pmeta[size_t(tpos) + j] = uint16_t(pos_meta(j, s_k[r], s_ord[r])); }
  }
  for (uint32_t p = s_ooff[0] + t; p < s_ooff[cnt]; p += TPB) {
    const uint32_t r = owner(s_ooff, p), q = PROV_HDR + s_k[r] + (p - s_ooff[r]);
    ops[p] = q < GATHER_STAGE ? s_rec[r * GATHER_STRIDE + q] : prov[s_slot[r] + q];
  }
  if (t < cnt) {
    if (flags & LINE_HAS_DEL) del_idx[del_scan[i]] = i;
    // (a record that fits the staged words is read from them; a longer one from the array)
    const bool staged = PROV_HDR + k + c <= GATHER_STAGE;
    const uint32_t* w = staged ? s_rec + t * GATHER_STRIDE : prov + slot;
    ShortOps so;
    make_rec_entry(rec, i, s_noff[t], k, s_ooff[t], c, w + PROV_HDR, w + PROV_HDR + k, &so);
    const uint32_t tpos = s_tpos[t];
    if (tpos != 0xFFFFFFFFu) {
      const PosRec pr = make_pos_rec(LineHdr{int32_t(w[0]), int32_t(w[1]), w[2], flags}, so);
      to.prec[i] = pr;
      const uint32_t tile = tpos >> 6;
      if (pr.line.tw & POS_NOT_PLAIN) atomicOr(&to.tile_first[tile], TILE_NOT_PLAIN);
      if ((tpos & 63u) == 0u) {
        if (k == 0u) { to.pnodes[tpos] = 0u; to.pmeta[tpos] = uint16_t(POS_INACTIVE); }
        tile_head(tile, i, place_used(to.place[i]), k ? w[PROV_HDR] : 0u, to.tile_first, to.pnodes, to.pmeta);
      }
    }
  }
}

// ---- locus bucketing (device form of gaf.cpp bucket_by_locus) ---------------------------------
__global__ void __launch_bounds__(TPB) bucket_keys_kernel(uint32_t n, const uint32_t* node_off, const uint32_t* nodes, uint32_t* key, uint32_t* idx) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = node_off[i], b = node_off[i + 1];
  key[i] = b > a ? min(nodes[a], nodes[b - 1]) : 0u;
  idx[i] = i;
}

// 1 for records of the short-walk class (walk.h), in locus order
__global__ void __launch_bounds__(TPB) class_flags_kernel(uint32_t n, const uint32_t* idx, const LineHdr* hdr, const uint32_t* node_off, uint32_t* cls) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = idx[i];
  cls[i] = short_walk_class(hdr[s].flags, node_off[s + 1] - node_off[s]) ? 1u : 0u;
}

// stable partition: short-walk class first
__global__ void __launch_bounds__(TPB) partition_kernel(uint32_t n, const uint32_t* idx, const uint32_t* cls, const uint32_t* before, uint32_t n_short, uint32_t* out) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = before[i];
  out[cls[i] ? b : n_short + (i - b)] = idx[i];
}

// inside every block of 2048 records of a class: by walk.h block_order_key, stable
__global__ void __launch_bounds__(TPB) order_blocks_kernel(uint32_t n, const uint32_t* idx_in, const LineHdr* hdr, const uint32_t* node_off, uint32_t* idx_out) {
  __shared__ uint32_t sh[SORT_BLOCK];
  const uint32_t a = blockIdx.x * SORT_BLOCK;
  for (uint32_t j = threadIdx.x; j < SORT_BLOCK; j += TPB) {
    uint32_t v = 0xFFFFFFFFu;
    if (a + j < n) {
      const uint32_t s = idx_in[a + j];
      v = (block_order_key(hdr[s].flags, node_off[s + 1] - node_off[s]) << 11) | j;
    }
    sh[j] = v;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= SORT_BLOCK; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < SORT_BLOCK; i += TPB) {
        const uint32_t o = i ^ j;
        if (o > i) {
          const uint32_t x = sh[i], y = sh[o];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { sh[i] = y; sh[o] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t j = threadIdx.x; j < SORT_BLOCK; j += TPB)
    if (a + j < n) idx_out[a + j] = idx_in[a + (sh[j] & (SORT_BLOCK - 1u))];
}

__global__ void __launch_bounds__(TPB) sorted_sizes_kernel(uint32_t n, const uint32_t* idx, const LineHdr* hdr, const uint32_t* node_off, const uint32_t* op_off,
                                                           uint32_t* nn, uint32_t* no, uint32_t* del) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = idx[i];
  nn[i] = node_off[s + 1] - node_off[s];
  no[i] = op_off[s + 1] - op_off[s];
  del[i] = (hdr[s].flags & LINE_HAS_DEL) ? 1u : 0u;
}

__global__ void __launch_bounds__(TPB) gather_kernel(uint32_t n, const uint32_t* idx, const LineHdr* hdr0, const uint32_t* node_off0, const uint32_t* op_off0,
                                                     const uint32_t* nodes0, const uint32_t* ops0, const uint32_t* node_off, const uint32_t* op_off,
                                                     const uint32_t* del_scan, LineHdr* hdr, uint32_t* nodes, uint32_t* ops, uint32_t* del_idx) {
  const uint32_t i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = idx[i];
  const LineHdr h = hdr0[s];
  hdr[i] = h;
  const uint32_t na = node_off0[s], nb = node_off0[s + 1], oa = op_off0[s], ob = op_off0[s + 1];
  uint32_t* dn = nodes + node_off[i];
  for (uint32_t k = na; k < nb; ++k) dn[k - na] = nodes0[k];
  uint32_t* dp = ops + op_off[i];
  for (uint32_t k = oa; k < ob; ++k) dp[k - oa] = ops0[k];
  if (h.flags & LINE_HAS_DEL) del_idx[del_scan[i]] = i;
}

struct Scanner {
  DevBuf tmp;
  size_t tmp_bytes = 0;
  explicit Scanner(ScratchPool& sp) : tmp(sp) {}
  // out[0..n] = exclusive prefix sums of in[0..n-1] (in must have n + 1 readable words, in[n] is ignored by value)
  void run(const uint32_t* in, uint32_t* out, size_t n_plus_1, hipStream_t s) {
    size_t need = 0;
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, int(n_plus_1), s));
    if (need > tmp_bytes) { tmp.alloc(need); tmp_bytes = need; }
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp.p, need, in, out, int(n_plus_1), s));
  }
};

double now() { return now_s(); }

// host text -> HBM through two pinned staging buffers (kept with the graph); several host threads fill one buffer while
// the other is in flight
// scan_cr: returns true when the text holds a '\r' that no '\n' follows (looked for while each piece is in cache): the reference's
// universal newlines end a line there, to the kernels it is whitespace -- the line that holds it is refused (gaf_device.h
// TE_LONE_CR), which costs a scan of every line, so the kernels only do it when told that there is something to find.  The callers
// in this file pass false since late round 4 (count_newlines_kernel finds it: the host scan held the copies to 40 GB/s)
// fd >= 0: the bytes come from the FILE (pread into the pinned buffer) instead of from its mapping -- a 19 GB GFA that was never
// touched through the mapping costs 4.7 M page faults to copy from (17 GB/s on the box's sixteen cores; pread: the page cache's speed)
bool upload_text(void* (&pinned)[2], hipEvent_t (&pinned_done)[2], size_t& pinned_bytes, const char* data, size_t size, uint8_t* dst, hipStream_t stream, int fd = -1,
                 bool scan_cr = true) {
  constexpr size_t STAGE = size_t(32) << 20;
  if (size == 0) return false;
  if (!pinned[0]) {
    for (int k = 0; k < 2; ++k) {
      HIP_OK(hipHostMalloc(&pinned[k], STAGE, hipHostMallocDefault));
      HIP_OK(hipEventCreateWithFlags(&pinned_done[k], hipEventDisableTiming));
    }
    pinned_bytes = STAGE;
  }
  size_t off = 0;
  int k = 0;
  bool used[2] = {false, false};
  const unsigned threads = std::min<unsigned>(hw_threads(), 16);
  std::atomic<size_t> first_lone_cr{SIZE_MAX};
  while (off < size) {
    const size_t len = std::min(STAGE, size - off);
    if (used[k]) HIP_OK(hipEventSynchronize(pinned_done[k]));
    char* dstp = static_cast<char*>(pinned[k]);
    const char* src = data + off;
    parallel_chunks(len, len < (size_t(4) << 20) ? 1u : threads, [&](unsigned, size_t b, size_t e) {
      if (fd >= 0) {
        for (size_t done = b; done < e;) {
          const ssize_t got = pread(fd, dstp + done, e - done, off_t(off + done));
          if (got <= 0) throw Error("OSError", "pread of the text failed");
          done += size_t(got);
        }
      } else {
        memcpy(dstp + b, src + b, e - b);
      }
      for (const char* q = dstp + b; scan_cr && q < dstp + e;) {      // (the GPU callers find it in count_newlines_kernel)
        q = static_cast<const char*>(memchr(q, '\r', size_t(dstp + e - q)));
        if (!q) break;
        const size_t at = off + size_t(q - dstp);     // position in the text
        if (at + 1 < size && data[at + 1] != '\n') { size_t none = SIZE_MAX; first_lone_cr.compare_exchange_strong(none, at); break; }
        ++q;
      }
    });
    HIP_OK(hipMemcpyAsync(dst + off, pinned[k], len, hipMemcpyHostToDevice, stream));
    HIP_OK(hipEventRecord(pinned_done[k], stream));
    used[k] = true;
    off += len;
    k ^= 1;
  }
  HIP_OK(hipStreamSynchronize(stream));
  return first_lone_cr.load() != SIZE_MAX;
}

// the kernels read whole tiles, and a line index entry may point one past the text: zero padding to a tile bound + one tile
inline size_t text_padded_bytes(size_t size) { return (size + TILE - 1) / TILE * TILE + TILE; }

inline uint32_t blocks_for(uint64_t n) { return uint32_t((n + TPB - 1) / TPB); }

}  // namespace

// ---- text of the NEXT batch on its way to HBM while the current one is tokenised and walked --------------------------
// A run over a GAF larger than one batch (pantas_augment_run) is bound by getting the text to the GPU: page-cache reads,
// the copy into pinned memory, PCIe.  The prefetcher owns two device text buffers, a copy stream and its own pinned
// staging buffers; start() hands a byte range to a worker thread, get() waits for the oldest started range.
struct TextPrefetcher {
  int device = 0;
  hipStream_t stream = nullptr;
  void* pinned[2] = {nullptr, nullptr};
  hipEvent_t pinned_done[2] = {nullptr, nullptr};
  size_t pinned_bytes = 0;
  uint8_t* buf[2] = {nullptr, nullptr};
  size_t cap = 0;                 // bytes of each device buffer
  struct Job { std::thread th; std::shared_ptr<Error> err; size_t size = 0; double seconds = 0; bool lone_cr = false; };
  Job job[2];
  unsigned head = 0, tail = 0;    // ranges started / handed out (slot = index & 1)
  double copy_seconds = 0;
};

void TextPrefetcherDeleter::operator()(TextPrefetcher* p) const {
  if (!p) return;
  for (auto& j : p->job) if (j.th.joinable()) j.th.join();
  (void)hipSetDevice(p->device);
  for (int k = 0; k < 2; ++k) {
    if (p->buf[k]) (void)hipFree(p->buf[k]);
    if (p->pinned[k]) (void)hipHostFree(p->pinned[k]);
    if (p->pinned_done[k]) (void)hipEventDestroy(p->pinned_done[k]);
  }
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

TextPrefetcherPtr make_text_prefetcher(DeviceGraph& dg, size_t max_bytes) {
  const DeviceGraphInfo gi = graph_info(dg);
  HIP_OK(hipSetDevice(gi.device));
  TextPrefetcherPtr p(new TextPrefetcher());
  p->device = gi.device;
  p->cap = text_padded_bytes(max_bytes);
  p->stream = take_stream(gi.device);
  for (int k = 0; k < 2; ++k) HIP_OK(dmalloc(&p->buf[k], p->cap));
  return p;
}

void prefetch_start(TextPrefetcher& p, const char* data, size_t size) {
  if (p.head - p.tail >= 2) throw Error("InternalError", "text prefetcher: both buffers are in use");
  if (text_padded_bytes(size) > p.cap) throw Error("InternalError", "text prefetcher: range larger than its buffers");
  const unsigned k = p.head & 1u;
  TextPrefetcher::Job& j = p.job[k];
  if (j.th.joinable()) j.th.join();
  j.err.reset();
  j.size = size;
  TextPrefetcher* pp = &p;
  j.th = std::thread([pp, k, data, size] {
    TextPrefetcher::Job& job = pp->job[k];
    const double t0 = now_s();
    try {
      HIP_OK(hipSetDevice(pp->device));
      HIP_OK(hipMemsetAsync(pp->buf[k] + size, 0, text_padded_bytes(size) - size, pp->stream));
      job.lone_cr = upload_text(pp->pinned, pp->pinned_done, pp->pinned_bytes, data, size, pp->buf[k], pp->stream, -1, false);      // (false: the tokeniser's line index looks)
      HIP_OK(hipStreamSynchronize(pp->stream));
    } catch (const Error& e) {
      job.err = std::make_shared<Error>(e);
    } catch (const std::exception& e) {
      job.err = std::make_shared<Error>("InternalError", e.what());
    }
    job.seconds = now_s() - t0;
  });
  ++p.head;
}

DeviceText prefetch_get(TextPrefetcher& p) {
  if (p.tail == p.head) throw Error("InternalError", "text prefetcher: nothing was started");
  const unsigned k = p.tail & 1u;
  TextPrefetcher::Job& j = p.job[k];
  if (j.th.joinable()) j.th.join();
  ++p.tail;
  p.copy_seconds += j.seconds;
  if (j.err) throw *j.err;
  return DeviceText{p.buf[k], j.size, j.lone_cr};
}

double prefetch_copy_seconds(const TextPrefetcher& p) { return p.copy_seconds; }


// The one-parse form of the batch build (kernels above).  nullptr: this batch needs the two-pass form -- a line the reference
// raises on, more lines without room than the arena takes, no records at all.
static DeviceBatchPtr tokenize_one_pass(const DeviceGraphInfo& gi, ScratchPool& pool, Scanner& scan, const uint8_t* d_text, size_t size, const uint64_t* d_ls,
                                        uint32_t n_lines, uint32_t stage_bytes, int thr, uint32_t check_cr, const DenseNames& names, uint64_t first_line_id,
                                        GafBatch& st, DeviceTokenizeTimes& tm, PhaseTimer& pt) {
  hipStream_t s = gi.stream;
  int bits = 1;
  while (bits < 32 && (uint64_t(1) << bits) < uint64_t(gi.V)) ++bits;
  const uint64_t prov_words = text_padded_bytes(size) / 4;
  uint64_t arena_words = std::max<uint64_t>(uint64_t(1) << 20, prov_words / 16);
  if (const char* a = getenv("PANTAS_TOKENIZER_ARENA_WORDS")) arena_words = std::max<uint64_t>(1, strtoull(a, nullptr, 10));    // (tests: a batch that runs out of it)
  if (bits > 30 || prov_words + arena_words >= 0xFFFFFFF0ull || n_lines == 0) return nullptr;
  double t0 = now();
  DevBuf prov(pool), key(pool), val(pool), key2(pool), val2(pool), partial(pool), small(pool), sort_tmp(pool);
  prov.alloc((prov_words + arena_words + 16) * 4);      // (+ 16: the gather reads sixteen words from every record's start, also the arena's last)
  key.alloc(size_t(n_lines) * 4); val.alloc(size_t(n_lines) * 8);
  // as many single-wave workgroups as the GPU holds at once (a grid that needs a second round of them waits for the first: 4096
  // workgroups where 3328 fit took 39 ms instead of 27), each walking groups of 64 lines
  uint32_t waves = line_group_grid(n_lines);
  {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, parse_once_kernel, int(LPB), stage_bytes) == hipSuccess && per_cu > 0 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gi.device) == hipSuccess && cus > 0)
      waves = std::min<uint32_t>((n_lines + LPB - 1u) / LPB, uint32_t(per_cu) * uint32_t(cus));
  }
  partial.alloc(size_t(waves) * N_PARTIAL * 4);
  // [0] err word, [1 .. N_TOTALS] totals, [N_TOTALS + 1] first kept line, then two words: arena cursor, fallback flag
  constexpr size_t SMALL_WORDS64 = N_TOTALS + 3;
  small.alloc(SMALL_WORDS64 * 8);
  HIP_OK(hipMemsetAsync(small.p, 0, SMALL_WORDS64 * 8, s));
  HIP_OK(hipMemsetAsync(small.p, 0xFF, 8, s));
  HIP_OK(hipMemsetAsync(small.as<unsigned long long>() + 1 + N_TOTALS, 0xFF, 8, s));
  uint32_t* d_arena_next = reinterpret_cast<uint32_t*>(small.as<unsigned long long>() + N_TOTALS + 2);
  // words a 64-byte-aligned start has to leave a line's record for the record to start there (PANTAS_TOK_ALIGN_MIN; 0 = never aligned,
  // the default: measured at S4 with 20 / 24 / 28 the gather gains 0.2 ms and the parse loses 0.1 - 0.9 to second passes --
  // the gather is not bound by the sectors a record straddles; profiles/r3_tok_align_ab.txt)
  uint32_t align_min = 0;
  if (const char* a = getenv("PANTAS_TOK_ALIGN_MIN")) align_min = uint32_t(atoi(a));
  if (align_min == 0) align_min = 0xFFFFFFFFu;
  // PANTAS_TOK_FAST=0: every line through the general parser (the A/B partner of the lean parser for plain lines; same arrays)
  const uint32_t tok_fast = (getenv("PANTAS_TOK_FAST") && atoi(getenv("PANTAS_TOK_FAST")) == 0) ? 0u : 1u;
  hipLaunchKernelGGL(parse_once_kernel, dim3(waves), dim3(LPB), stage_bytes, s, d_text, d_ls, n_lines, stage_bytes, thr, check_cr, names, uint32_t(first_line_id),
                     uint32_t(bits), prov.as<uint32_t>(), prov_words, d_arena_next, uint32_t(arena_words), key.as<uint32_t>(), val.as<unsigned long long>(),
                     partial.as<uint32_t>(), small.as<unsigned long long>(), d_arena_next + 1, 0u, align_min, tok_fast);
  HIP_OK(hipGetLastError());
#ifndef PANTAS_EXPERIMENTS
  if (getenv("PANTAS_TOK_STOP"))
    throw Error("Unsupported", "PANTAS_TOK_STOP is a timing experiment of builds made with `make EXPERIMENTS=1`; this library does not have it");
#else
  if (const char* ab = getenv("PANTAS_TOK_STOP")) {
    fprintf(stderr, "pantas: PANTAS_TOK_STOP=%s -- timing experiment (the relaunches write into the live buffers), THE RESULTS OF THIS RUN ARE INVALID\n", ab);
    // timing-only ablation of the parse (tools/tok_ablate.py): the kernel again, leaving every line after the named stage;
    // its outputs are garbage, the real launch above has already written the batch's
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    for (const char* p = ab; *p; ++p) {
      if (!((*p >= '0' && *p <= '9') || (*p >= 'a' && *p <= 'f'))) continue;
      const uint32_t stop_code = *p <= '9' ? uint32_t(*p - '0') : 10u + uint32_t(*p - 'a');      // a .. f: stages 0 .. 5 of the lean parser (a = all of it)
      DevBuf junk(pool);
      junk.alloc(size_t(n_lines) * 4);
      HIP_OK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(parse_once_kernel, dim3(waves), dim3(LPB), stage_bytes, s, d_text, d_ls, n_lines, stage_bytes, thr, check_cr, names, uint32_t(first_line_id),
                         uint32_t(bits), prov.as<uint32_t>(), prov_words, d_arena_next, uint32_t(arena_words), junk.as<uint32_t>(), val.as<unsigned long long>(),
                         partial.as<uint32_t>(), small.as<unsigned long long>() + N_TOTALS + 2 /* (never written in this mode) */, d_arena_next + 1, stop_code, align_min, *p == '0' ? tok_fast : 0u);
      HIP_OK(hipEventRecord(e1, s));
      HIP_OK(hipEventSynchronize(e1));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      fprintf(stderr, "[tok ablate] stop %c: %.3f ms for %u lines\n", *p, ms, n_lines);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
#endif
  hipLaunchKernelGGL(totals_kernel, dim3(std::min<uint32_t>(blocks_for(waves), 1024)), dim3(TPB), 0, s, partial.as<uint32_t>(), waves, small.as<unsigned long long>() + 1);
  HIP_OK(hipGetLastError());
  unsigned long long hs[SMALL_WORDS64];
  HIP_OK(hipMemcpyAsync(hs, small.p, sizeof hs, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  tm.measure_s = now() - t0;
  tm.write_s = 0;
  pt.lap("parse (one pass)");
  const uint32_t fallback = uint32_t(hs[N_TOTALS + 2] >> 32);
  const uint64_t n_rec = hs[1], n_nodes = hs[2], n_ops = hs[3];
  if (hs[0] != ~0ull || fallback || n_rec == 0) return nullptr;
  if (n_nodes >= NONE32 || n_ops >= NONE32 || n_rec >= NONE32) throw Error("Unsupported", "more than 2^32-2 path nodes or cs ops in one batch");
  st = GafBatch{};
  st.bytes = size;
  st.rejected = hs[4];
  st.skipped = hs[5];
  st.del_lines = hs[6];
  st.del_nodes = hs[7];
  st.lines_total = n_lines;
  st.first_record_line = first_line_id + hs[N_TOTALS + 1];
  partial.release();

  // ---- bucket by locus: one stable sort on {class, lowest end node}; lines that are no records sort behind both classes
  t0 = now();
  const uint32_t n = uint32_t(n_rec), n_short = uint32_t(hs[8]);
  key2.alloc(size_t(n_lines) * 4); val2.alloc(size_t(n_lines) * 8);
  pt.lap("  sort buffers allocated");
  size_t need = 0;
  HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key.as<uint32_t>(), key2.as<uint32_t>(), val.as<unsigned long long>(), val2.as<unsigned long long>(),
                                            int(n_lines), 0, bits + 2, s));
  sort_tmp.alloc(need);
  HIP_OK(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, need, key.as<uint32_t>(), key2.as<uint32_t>(), val.as<unsigned long long>(), val2.as<unsigned long long>(),
                                            int(n_lines), 0, bits + 2, s));
  if (n_short) {
    hipLaunchKernelGGL(order_blocks2_kernel, dim3((n_short + SORT_BLOCK - 1) / SORT_BLOCK), dim3(TPB), 0, s, n_short, val2.as<unsigned long long>(), prov.as<uint32_t>(),
                       val.as<unsigned long long>());
    HIP_OK(hipGetLastError());
  }
  if (n > n_short) {
    hipLaunchKernelGGL(order_blocks2_kernel, dim3((n - n_short + SORT_BLOCK - 1) / SORT_BLOCK), dim3(TPB), 0, s, n - n_short, val2.as<unsigned long long>() + n_short,
                       prov.as<uint32_t>(), val.as<unsigned long long>() + n_short);
    HIP_OK(hipGetLastError());
  }
  HIP_OK(hipStreamSynchronize(s));
  pt.lap("  sort + block order");
  DevBuf nn(pool), no(pool), dl(pool), dscan(pool), node_off1(pool), op_off1(pool);
  nn.alloc((size_t(n) + 1) * 4); no.alloc((size_t(n) + 1) * 4); dl.alloc((size_t(n) + 1) * 4); dscan.alloc((size_t(n) + 1) * 4);
  node_off1.alloc((size_t(n) + 1) * 4); op_off1.alloc((size_t(n) + 1) * 4);
  pt.lap("  size arrays allocated");
  HIP_OK(hipMemsetAsync(nn.as<uint32_t>() + n, 0, 4, s));
  HIP_OK(hipMemsetAsync(no.as<uint32_t>() + n, 0, 4, s));
  HIP_OK(hipMemsetAsync(dl.as<uint32_t>() + n, 0, 4, s));
  hipLaunchKernelGGL(sizes2_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, val.as<unsigned long long>(), prov.as<uint32_t>(), nn.as<uint32_t>(), no.as<uint32_t>(),
                     dl.as<uint32_t>());
  HIP_OK(hipGetLastError());
  scan.run(nn.as<uint32_t>(), node_off1.as<uint32_t>(), size_t(n) + 1, s);
  scan.run(no.as<uint32_t>(), op_off1.as<uint32_t>(), size_t(n) + 1, s);
  scan.run(dl.as<uint32_t>(), dscan.as<uint32_t>(), size_t(n) + 1, s);
  DeviceBatchPtr out(new DeviceBatch());
  out->n = n;
  out->n_nodes = n_nodes;
  out->n_ops = n_ops;
  out->del_nodes = st.del_nodes;
  out->n_del = uint32_t(st.del_lines);
  out->n_short = n_short;
  out->short_nodes = hs[9];
  out->short_ops = hs[10];
  DevBuf hdr1(pool), nodes1(pool), ops1(pool), del(pool);
  hdr1.alloc((n_rec + 1) * sizeof(LineHdr)); nodes1.alloc((n_nodes + 1) * 4); ops1.alloc((n_ops + 1) * 4);
  del.alloc((size_t(out->n_del) + 1) * 4);
  // (rec: only for a record-per-lane default kernel; the position-parallel one reads the op summaries out of its line records)
  if (main_kernel_variant() < 120) HIP_OK(dmalloc(&out->rec, size_t(n) * REC_QUADS * sizeof(uint4)));
  HIP_OK(hipStreamSynchronize(s));
  pt.lap("  sizes + scans; batch arrays allocated");
  // the tiles of the position-parallel kernel (the default main kernel) come out of the same gather: records placed first
  TileOut to{};
  uint32_t* place = nullptr;
  if (main_kernel_variant() >= 120 && n_short) {
    place = place_batch_tiles(gi.device, s, pool, *out, node_off1.as<uint32_t>());
    to = TileOut{place, out->chunk_tile, out->tile_first, out->pnodes, out->pmeta, out->prec, n_short};
    pt.lap("  records placed in tiles");
  }
  hipLaunchKernelGGL(gather2_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, val.as<unsigned long long>(), prov.as<uint32_t>(), node_off1.as<uint32_t>(),
                     op_off1.as<uint32_t>(), dscan.as<uint32_t>(), hdr1.as<LineHdr>(), nodes1.as<uint32_t>(), ops1.as<uint32_t>(), del.as<uint32_t>(), out->rec, to);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(s));
  if (place) pool.give(place);
  out->hdr = hdr1.hand_over<LineHdr>();
  out->node_off = node_off1.hand_over<uint32_t>();
  out->op_off = op_off1.hand_over<uint32_t>();
  out->nodes = nodes1.hand_over<uint32_t>();
  out->ops = ops1.hand_over<uint32_t>();
  out->del_idx = del.hand_over<uint32_t>();
  tm.bucket_s = now() - t0;
  pt.lap("bucket by locus (one pass)");
  return out;
}

DeviceBatchPtr tokenize_gaf_device(DeviceGraph& dg, const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id,
                                   GafBatch& st, DeviceTokenizeTimes* times, const DeviceText* resident) {
  if (!g.names.dense)
    throw Error("Unsupported", "the GPU tokeniser needs segment names that are canonical decimals (this graph's are not): use the host tokeniser");
  const DeviceGraphInfo gi = graph_info(dg);
  HIP_OK(hipSetDevice(gi.device));
  hipStream_t s = gi.stream;
  ScratchPool& pool = scratch_pool(dg);
  PhaseTimer pt("tokenize_gaf_device");
  DeviceTokenizeTimes tm{};
  const double t_begin = now();

  // ---- name table (kept with the graph between calls)
  uint32_t* d_table = device_name_table(dg, g);
  const DenseNames names{d_table, g.names.base, uint64_t(g.names.table.size()), getenv("PANTAS_NAME_BLOCKS") ? device_name_blocks(dg) : nullptr};    // (measured flat at S4: 17.99 vs 17.50 ms -- the lookups are not what the kernel waits for; off)

  // ---- text
  const size_t padded = text_padded_bytes(size);
  DevBuf text(pool);
  double t0 = now();
  const uint8_t* d_text;
  uint32_t check_cr = 0;
  if (resident) {
    // the text is in HBM already (TextPrefetcher copied it while the batch before this one was being handled)
    if (resident->size != size || !resident->dev) throw Error("InternalError", "resident text does not match the range");
    d_text = resident->dev;
    check_cr = 0u;      // (found by the line index below)
  } else {
    text.alloc(padded);
    HIP_OK(hipMemsetAsync(text.as<uint8_t>() + size, 0, padded - size, s));
    (void)upload_text(pool.pinned, pool.pinned_done, pool.pinned_bytes, data, size, text.as<uint8_t>(), s, -1, false);
    d_text = text.as<uint8_t>();
  }
  const uint4* d_text4 = reinterpret_cast<const uint4*>(d_text);
  tm.h2d_s = now() - t0;
  pt.lap("text -> HBM");

  // ---- line index
  const uint32_t tiles = uint32_t((size + TILE - 1) / TILE);
  if (uint64_t(tiles) * TILE / 2 >= 0xFFFFFFFFull) throw Error("Unsupported", "GAF range too large for one batch");
  Scanner scan(pool);
  DevBuf tile_cnt(pool), tile_off(pool);
  tile_cnt.alloc((size_t(tiles) + 2) * 4);      // (+ one word behind the counts: the lone-'\r' flag of the text)
  tile_off.alloc((size_t(tiles) + 1) * 4);
  HIP_OK(hipMemsetAsync(tile_cnt.p, 0, (size_t(tiles) + 2) * 4, s));
  t0 = now();
  if (tiles) {
    hipLaunchKernelGGL(count_newlines_kernel, dim3(tiles), dim3(TPB), 0, s, d_text4, tile_cnt.as<uint32_t>(), tile_cnt.as<uint32_t>() + tiles + 1, uint64_t(size));
    HIP_OK(hipGetLastError());
  }
  scan.run(tile_cnt.as<uint32_t>(), tile_off.as<uint32_t>(), size_t(tiles) + 1, s);
  uint32_t newlines = 0, lone_cr = 0;
  HIP_OK(hipMemcpyAsync(&newlines, tile_off.as<uint32_t>() + tiles, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(&lone_cr, tile_cnt.as<uint32_t>() + tiles + 1, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  check_cr = lone_cr ? 1u : 0u;
  const bool open_tail = size > 0 && data[size - 1] != '\n';
  const uint64_t n_lines64 = uint64_t(newlines) + (open_tail ? 1 : 0);
  if (first_line_id + n_lines64 > uint64_t(0xFFFFFFFFu)) throw Error("Unsupported", "more than 2^32-1 GAF lines");
  if (n_lines64 >= 0x7FFFFFFFull) throw Error("Unsupported", "more than 2^31-2 GAF lines in one batch");
  const uint32_t n_lines = uint32_t(n_lines64);
  DevBuf ls(pool);
  ls.alloc((size_t(n_lines) + 2) * 8);
  HIP_OK(hipMemsetAsync(ls.p, 0, 8, s));
  if (tiles) {
    hipLaunchKernelGGL(line_starts_kernel, dim3(tiles), dim3(TPB), 0, s, d_text4, tile_off.as<uint32_t>(), ls.as<uint64_t>());
    HIP_OK(hipGetLastError());
  }
  if (open_tail) {
    const uint64_t end1 = uint64_t(size) + 1;
    HIP_OK(hipMemcpyAsync(ls.as<uint64_t>() + n_lines, &end1, 8, hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
  }
  tile_cnt.release();
  if (times) { HIP_OK(hipStreamSynchronize(s)); tm.index_s = now() - t0; }
  pt.lap("line index");

  // ---- measure
  // LDS per wave: the text of 64 average lines plus a quarter, between 4 and 64 KB
  const uint32_t stage_bytes = uint32_t(std::min<uint64_t>(64 * 1024, std::max<uint64_t>(4096, ((n_lines ? size / n_lines : 0) * 80 + 1279) / 1024 * 1024)));
  // ---- one parse per line (PANTAS_TOKENIZER_PASSES=2: the two-pass form, which is also where a batch with a raising line ends up)
  {
    const char* passes = getenv("PANTAS_TOKENIZER_PASSES");
    if (!(passes && passes[0] == '2') && !getenv("PANTAS_NO_BUCKET")) {
      // (the one-parse kernel stages in rounds when a group does not fit: it takes a tighter buffer -- the text of 64 average lines
      // plus a twelfth, 512-byte steps -- for more waves per CU)
      const uint32_t stage_one = uint32_t(std::min<uint64_t>(64 * 1024, std::max<uint64_t>(4096, ((n_lines ? size / n_lines : 0) * 70 + 511) / 512 * 512)));
      DeviceBatchPtr one = tokenize_one_pass(gi, pool, scan, d_text, size, ls.as<uint64_t>(), n_lines, getenv("PANTAS_TOK_STAGE") ? uint32_t(atoi(getenv("PANTAS_TOK_STAGE"))) : stage_one,
                                             thr, check_cr, names, first_line_id, st, tm, pt);
      if (one) {
        tm.total_s = now() - t_begin;
        if (times) *times = tm;
        return one;
      }
    }
  }
  DevBuf kept(pool), cnt_nodes(pool), cnt_ops(pool), meta(pool), small(pool);
  const size_t np1 = size_t(n_lines) + 1;
  kept.alloc(np1 * 4); cnt_nodes.alloc(np1 * 4); cnt_ops.alloc(np1 * 4); meta.alloc(np1);
  small.alloc(64);   // [0] err word, [1..7] totals
  HIP_OK(hipMemsetAsync(kept.as<uint32_t>() + n_lines, 0, 4, s));
  HIP_OK(hipMemsetAsync(cnt_nodes.as<uint32_t>() + n_lines, 0, 4, s));
  HIP_OK(hipMemsetAsync(cnt_ops.as<uint32_t>() + n_lines, 0, 4, s));
  HIP_OK(hipMemsetAsync(small.p, 0xFF, 8, s));
  HIP_OK(hipMemsetAsync(small.as<uint8_t>() + 8, 0, 56, s));
  t0 = now();
  if (n_lines) {
    hipLaunchKernelGGL(measure_kernel, dim3(line_group_grid(n_lines)), dim3(LPB), stage_bytes, s, d_text, ls.as<uint64_t>(), n_lines, stage_bytes, thr, check_cr, names,
                       kept.as<uint32_t>(), cnt_nodes.as<uint32_t>(), cnt_ops.as<uint32_t>(), meta.as<uint8_t>(), small.as<unsigned long long>());
    HIP_OK(hipGetLastError());
    hipLaunchKernelGGL(limit_and_totals_kernel, dim3(std::min<uint32_t>(blocks_for(n_lines), 2048)), dim3(TPB), 0, s, n_lines,
                       small.as<unsigned long long>(), kept.as<uint32_t>(), cnt_nodes.as<uint32_t>(), cnt_ops.as<uint32_t>(), meta.as<uint8_t>(),
                       small.as<unsigned long long>() + 1);
    HIP_OK(hipGetLastError());
  }
  DevBuf rec_of(pool), node_scan(pool), op_scan(pool);
  rec_of.alloc(np1 * 4); node_scan.alloc(np1 * 4); op_scan.alloc(np1 * 4);
  scan.run(kept.as<uint32_t>(), rec_of.as<uint32_t>(), np1, s);
  scan.run(cnt_nodes.as<uint32_t>(), node_scan.as<uint32_t>(), np1, s);
  scan.run(cnt_ops.as<uint32_t>(), op_scan.as<uint32_t>(), np1, s);
  unsigned long long host_small[8];
  HIP_OK(hipMemcpyAsync(host_small, small.p, 64, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  tm.measure_s = now() - t0;
  pt.lap("measure + scans");
  cnt_nodes.release(); cnt_ops.release(); meta.release();

  const unsigned long long err = host_small[0];
  const uint64_t n_rec = host_small[1], n_nodes = host_small[2], n_ops = host_small[3];
  if (n_nodes >= NONE32 || n_ops >= NONE32 || n_rec >= NONE32) throw Error("Unsupported", "more than 2^32-2 path nodes or cs ops in one batch");
  st = GafBatch{};
  st.bytes = size;
  st.rejected = host_small[4];
  st.skipped = host_small[5];
  st.del_lines = host_small[6];
  st.del_nodes = host_small[7];
  st.lines_total = err == ~0ull ? uint64_t(n_lines) : uint64_t(err >> 8) + 1;
  if (err != ~0ull) {
    // word the error the way the host tokeniser does, from the one offending line
    const uint64_t line = err >> 8;
    uint64_t span[2];
    HIP_OK(hipMemcpy(span, ls.as<uint64_t>() + line, 16, hipMemcpyDeviceToHost));
    GafBatch one;
    const size_t lb = size_t(span[0]), le = std::min<size_t>(size_t(span[1]), size);
    tokenize_gaf(g, data + lb, le - lb, thr, first_line_id + line, 1, one);
    if (!one.deferred) throw Error("InternalError", "GPU tokeniser refused GAF line " + std::to_string(first_line_id + line + 1) + " (code " +
                                                      std::to_string(unsigned(err & 0xFF)) + ") but the host tokeniser accepts it");
    st.deferred = one.deferred;
  }

  // ---- write (file order)
  DevBuf hdr0(pool), node_off0(pool), op_off0(pool), nodes0(pool), ops0(pool);
  hdr0.alloc((n_rec + 1) * sizeof(LineHdr));
  node_off0.alloc((n_rec + 1) * 4); op_off0.alloc((n_rec + 1) * 4);
  nodes0.alloc((n_nodes + 1) * 4); ops0.alloc((n_ops + 1) * 4);
  t0 = now();
  if (n_lines) {
    hipLaunchKernelGGL(write_kernel, dim3(line_group_grid(n_lines)), dim3(LPB), stage_bytes, s, d_text, ls.as<uint64_t>(), n_lines, stage_bytes, thr, check_cr, names,
                       kept.as<uint32_t>(), rec_of.as<uint32_t>(), node_scan.as<uint32_t>(), op_scan.as<uint32_t>(), uint32_t(first_line_id),
                       hdr0.as<LineHdr>(), node_off0.as<uint32_t>(), op_off0.as<uint32_t>(), nodes0.as<uint32_t>(), ops0.as<uint32_t>());
    HIP_OK(hipGetLastError());
  }
  const uint32_t tot_nodes = uint32_t(n_nodes), tot_ops = uint32_t(n_ops);
  HIP_OK(hipMemcpyAsync(node_off0.as<uint32_t>() + n_rec, &tot_nodes, 4, hipMemcpyHostToDevice, s));
  HIP_OK(hipMemcpyAsync(op_off0.as<uint32_t>() + n_rec, &tot_ops, 4, hipMemcpyHostToDevice, s));
  uint32_t first_rec_line = 0;
  if (n_rec) HIP_OK(hipMemcpyAsync(&first_rec_line, &hdr0.as<LineHdr>()[0].line_id, 4, hipMemcpyDeviceToHost, s));   // file order here
  HIP_OK(hipStreamSynchronize(s));
  if (n_rec) st.first_record_line = first_rec_line;
  tm.write_s = now() - t0;
  pt.lap("write records");
  text.release(); ls.release(); kept.release(); rec_of.release(); node_scan.release(); op_scan.release();

  // ---- bucket by locus
  DeviceBatchPtr out(new DeviceBatch());
  out->n = uint32_t(n_rec);
  out->n_nodes = n_nodes;
  out->n_ops = n_ops;
  out->del_nodes = st.del_nodes;
  out->n_del = uint32_t(st.del_lines);
  t0 = now();
  const uint32_t n = uint32_t(n_rec);
  if (n == 0) {
    out->hdr = hdr0.hand_over<LineHdr>();
    out->node_off = node_off0.hand_over<uint32_t>();
    out->op_off = op_off0.hand_over<uint32_t>();
    out->nodes = nodes0.hand_over<uint32_t>();
    out->ops = ops0.hand_over<uint32_t>();
    DevBuf del(pool);
    del.alloc(16);
    out->del_idx = del.hand_over<uint32_t>();
  } else {
    uint32_t n_short = 0;
    DevBuf key(pool), idx(pool), key2(pool), idx2(pool), sort_tmp(pool);
    key.alloc(size_t(n) * 4); idx.alloc(size_t(n) * 4);
    hipLaunchKernelGGL(bucket_keys_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, node_off0.as<uint32_t>(), nodes0.as<uint32_t>(), key.as<uint32_t>(), idx.as<uint32_t>());
    HIP_OK(hipGetLastError());
    if (!getenv("PANTAS_NO_BUCKET")) {
      key2.alloc(size_t(n) * 4); idx2.alloc(size_t(n) * 4);
      int bits = 1;
      while (bits < 32 && (uint64_t(1) << bits) < uint64_t(gi.V)) ++bits;
      size_t need = 0;
      HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, key.as<uint32_t>(), key2.as<uint32_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), int(n), 0, bits, s));
      sort_tmp.alloc(need);
      HIP_OK(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, need, key.as<uint32_t>(), key2.as<uint32_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), int(n), 0, bits, s));
      // idx2 = records by locus -> key = the two classes apart (stable) -> idx = every 2048-block of a class by path length
      DevBuf cls(pool), before(pool);
      cls.alloc((size_t(n) + 1) * 4); before.alloc((size_t(n) + 1) * 4);
      HIP_OK(hipMemsetAsync(cls.as<uint32_t>() + n, 0, 4, s));
      hipLaunchKernelGGL(class_flags_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, idx2.as<uint32_t>(), hdr0.as<LineHdr>(), node_off0.as<uint32_t>(), cls.as<uint32_t>());
      HIP_OK(hipGetLastError());
      scan.run(cls.as<uint32_t>(), before.as<uint32_t>(), size_t(n) + 1, s);
      HIP_OK(hipMemcpyAsync(&n_short, before.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, s));
      HIP_OK(hipStreamSynchronize(s));
      hipLaunchKernelGGL(partition_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, idx2.as<uint32_t>(), cls.as<uint32_t>(), before.as<uint32_t>(), n_short, key.as<uint32_t>());
      HIP_OK(hipGetLastError());
      if (n_short) {
        hipLaunchKernelGGL(order_blocks_kernel, dim3((n_short + SORT_BLOCK - 1) / SORT_BLOCK), dim3(TPB), 0, s, n_short, key.as<uint32_t>(), hdr0.as<LineHdr>(),
                           node_off0.as<uint32_t>(), idx.as<uint32_t>());
        HIP_OK(hipGetLastError());
      }
      if (n > n_short) {
        hipLaunchKernelGGL(order_blocks_kernel, dim3((n - n_short + SORT_BLOCK - 1) / SORT_BLOCK), dim3(TPB), 0, s, n - n_short, key.as<uint32_t>() + n_short,
                           hdr0.as<LineHdr>(), node_off0.as<uint32_t>(), idx.as<uint32_t>() + n_short);
        HIP_OK(hipGetLastError());
      }
    }
    DevBuf nn(pool), no(pool), dl(pool), dscan(pool), node_off1(pool), op_off1(pool);
    nn.alloc((size_t(n) + 1) * 4); no.alloc((size_t(n) + 1) * 4); dl.alloc((size_t(n) + 1) * 4); dscan.alloc((size_t(n) + 1) * 4);
    node_off1.alloc((size_t(n) + 1) * 4); op_off1.alloc((size_t(n) + 1) * 4);
    HIP_OK(hipMemsetAsync(nn.as<uint32_t>() + n, 0, 4, s));
    HIP_OK(hipMemsetAsync(no.as<uint32_t>() + n, 0, 4, s));
    HIP_OK(hipMemsetAsync(dl.as<uint32_t>() + n, 0, 4, s));
    hipLaunchKernelGGL(sorted_sizes_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, idx.as<uint32_t>(), hdr0.as<LineHdr>(), node_off0.as<uint32_t>(),
                       op_off0.as<uint32_t>(), nn.as<uint32_t>(), no.as<uint32_t>(), dl.as<uint32_t>());
    HIP_OK(hipGetLastError());
    scan.run(nn.as<uint32_t>(), node_off1.as<uint32_t>(), size_t(n) + 1, s);
    scan.run(no.as<uint32_t>(), op_off1.as<uint32_t>(), size_t(n) + 1, s);
    scan.run(dl.as<uint32_t>(), dscan.as<uint32_t>(), size_t(n) + 1, s);
    DevBuf hdr1(pool), nodes1(pool), ops1(pool), del(pool);
    hdr1.alloc((n_rec + 1) * sizeof(LineHdr)); nodes1.alloc((n_nodes + 1) * 4); ops1.alloc((n_ops + 1) * 4);
    del.alloc((size_t(out->n_del) + 1) * 4);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks_for(n)), dim3(TPB), 0, s, n, idx.as<uint32_t>(), hdr0.as<LineHdr>(), node_off0.as<uint32_t>(),
                       op_off0.as<uint32_t>(), nodes0.as<uint32_t>(), ops0.as<uint32_t>(), node_off1.as<uint32_t>(), op_off1.as<uint32_t>(),
                       dscan.as<uint32_t>(), hdr1.as<LineHdr>(), nodes1.as<uint32_t>(), ops1.as<uint32_t>(), del.as<uint32_t>());
    HIP_OK(hipGetLastError());
    out->n_short = n_short;
    uint32_t ext[2] = {0, 0};
    HIP_OK(hipMemcpyAsync(&ext[0], node_off1.as<uint32_t>() + n_short, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(&ext[1], op_off1.as<uint32_t>() + n_short, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    out->short_nodes = ext[0];
    out->short_ops = ext[1];
    out->hdr = hdr1.hand_over<LineHdr>();
    out->node_off = node_off1.hand_over<uint32_t>();
    out->op_off = op_off1.hand_over<uint32_t>();
    out->nodes = nodes1.hand_over<uint32_t>();
    out->ops = ops1.hand_over<uint32_t>();
    out->del_idx = del.hand_over<uint32_t>();
  }
  build_batch_rec(dg, *out);
  tm.bucket_s = now() - t0;
  pt.lap("bucket by locus");
  tm.total_s = now() - t_begin;
  if (times) *times = tm;
  return out;
}


// ================================================================================================================
// GFA text -> the graph's arrays ON THE GPU (round 3; the host loader is graph.cpp, the reference's parsers
// scripts/alignments_augmentation_from_gaf.py:121-126, scripts/call.py:145-181, scripts/gfautils.py:92-128).
// Both command lines spent 4.4 - 4.9 s of a whole-genome run in the host parse of 19 - 23 GB of GFA text on the box's sixteen
// granted cores; the same text goes over PCIe in 0.4 s and is classified at memory speed.  The device form takes the graphs
// a graph builder writes -- every S / L line starts at the first byte of its line, names are canonical decimals in a dense
// range, one S line per name, both ends of every link known -- and gives ANYTHING else back to the host loader (nullptr), which
// knows every rule of the three reference parsers.  Same arrays, byte for byte: tests compare the two loaders through the
// binary graph cache they both can write.
//   line index (as for the GAF)  ->  one lane per line: kind, name / link ends as numbers, sequence length  ->  S lines numbered
//   by a scan (= node index: order of first appearance), name table by scatter  ->  links: ends through the table, stable radix
//   sorts by source (rows in L-line file order) and by target (in-lists in file order)  ->  arrays to the host.
// P lines are counted where they stand (a P line of a chromosome is hundreds of megabytes: no lane walks that): the host looks at
// the few of them afterwards.
namespace {

enum : uint8_t { GK_OTHER = 0, GK_S = 1, GK_L = 2, GK_P = 3 };
struct KindIs {
  uint8_t want;
  __host__ __device__ uint32_t operator()(uint8_t k) const { return k == want ? 1u : 0u; }
};
struct GfaLine { unsigned long long a, b; };     // S: {name as a number, seq len | name delta << 32 | name len << 48}; L: {source, target}

struct GfaFlags {        // [0] odd (a line the device form does not take), [1] L lines with fewer than 6 tokens: first offset
  unsigned long long odd, first_short_l;
};

// a token of at most 18 plain digits without a leading zero (Span parse_canonical_u64)
__device__ __forceinline__ bool gfa_number(TextWords& T, uint32_t b, uint32_t e, unsigned long long& v) {
  const uint32_t n = e - b;
  if (n == 0 || n > 18) return false;
  if (n > 1 && T.at(b) == '0') return false;
  unsigned long long x = 0;
  for (uint32_t c = b; c < e; ++c) {
    const uint32_t d = T.at(c) - '0';
    if (d > 9u) return false;
    x = x * 10ull + d;
  }
  v = x;
  return true;
}

__global__ void __launch_bounds__(LPB) gfa_lines_kernel(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, uint8_t* kind, GfaLine* rec,
                                                        unsigned long long* flags, uint32_t* too_long_scratch) {
  const uint32_t groups = (n_lines + LPB - 1u) / LPB;
  for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    // lines that do not fit the staging buffer: a P line is named by its first byte, anything else is odd
    {
      const uint32_t i = grp * LPB + threadIdx.x;
      if (i < n_lines) {
        const uint64_t lb = ls[i], le = ls[i + 1];
        if (le - lb > uint64_t(stage_bytes) - 32u) {
          const uint8_t c = lb < le - 1 ? text[lb] : 0;
          kind[i] = c == 'P' ? GK_P : GK_OTHER;
          if (c == 'S' || c == 'L' || c == ' ' || (c >= 9 && c <= 13)) atomicMin(&flags[0], static_cast<unsigned long long>(lb));
        }
      }
    }
    staged_rounds(grp, text, ls, n_lines, stage_bytes, too_long_scratch, [&](uint32_t i, const uint32_t* words, uint32_t b, uint32_t e) {
      TextWords T(words);
      uint8_t k = GK_OTHER;
      GfaLine r{0, 0};
      const uint64_t off = ls[i];
      auto odd = [&] { atomicMin(&flags[0], static_cast<unsigned long long>(off)); };
      // (a '\r' in front of the '\n' is whitespace at the line's end; one anywhere else was looked for when the text was copied)
      uint32_t end = e;
      while (end > b && tk_ws(T.at(end - 1u))) --end;
      const uint32_t c0 = b < end ? T.at(b) : 0u;
      if (b < end && tk_ws(c0)) odd();                                   // whitespace in front of a line: the host knows what each parser makes of it
      else if (c0 == 'S' || c0 == 'L') {
        uint32_t tb[6], te[6], n = 0, p = b;
        while (n < 6u) {
          while (p < end && tk_ws(T.at(p))) ++p;
          if (p >= end) break;
          tb[n] = p;
          while (p < end && !tk_ws(T.at(p))) ++p;
          te[n] = p;
          ++n;
        }
        if (c0 == 'S') {
          unsigned long long id;
          if (n < 3u || !gfa_number(T, tb[1], te[1], id) || te[2] - tb[2] >= (1u << 30) || tb[1] - b > 65535u) odd();
          else { k = GK_S; r.a = id; r.b = static_cast<unsigned long long>(te[2] - tb[2]) | (static_cast<unsigned long long>(tb[1] - b) << 32) | (static_cast<unsigned long long>(te[1] - tb[1]) << 48); }
        } else if (end - b > 1u) {                                        // (a bare "L" is no link: the loader skips it)
          unsigned long long u, v;
          if (n < 4u || !gfa_number(T, tb[1], te[1], u) || !gfa_number(T, tb[3], te[3], v)) odd();
          else {
            k = GK_L; r.a = u; r.b = v;
            if (n < 6u) atomicMin(&flags[1], static_cast<unsigned long long>(off));
          }
        }
      } else if (c0 == 'P') k = GK_P;
      kind[i] = k;
      rec[i] = r;
    });
  }
}

// ---- the tags of the S / L lines as `pantas call` reads them (call_host.cpp build_call_tables phase A; scripts/call.py:25-32,
// :145-181), for the command line that goes on to call: a second staged pass over the text, once the S / L ordinals are known.
// A tag is name:type:value with exactly two ':'; NC / LN / RC values are ints, IL / OL values lists of int.int; the value of EX
// (S lines) and JN (L lines) is kept as a span of the text, the presence of ID (L lines) as a flag; a later tag of a name
// overrides an earlier one.  Only the plain form is taken -- digits without sign or '_' for the ints (IL / OL: `tag_plain_hist`),
// a value below 2^31 -- anything else sets `odd` and the host scans the lines itself (and words the error).
constexpr uint32_t TAG_HAS_NC = 1u, TAG_HAS_EX = 2u;      // nflags: bit 0 = call_device.h N_HAS_NC
__device__ __forceinline__ bool tag_plain_int(TextWords& T, uint32_t b, uint32_t e, uint32_t& v) {
  const uint32_t n = e - b;
  if (n == 0 || n > 10) return false;
  unsigned long long x = 0;
  for (uint32_t c = b; c < e; ++c) {
    const uint32_t d = T.at(c) - '0';
    if (d > 9u) return false;
    x = x * 10ull + d;
  }
  if (x > 0x7FFFFFFFull) return false;
  v = uint32_t(x);
  return true;
}
__device__ __forceinline__ bool tag_plain_hist(TextWords& T, uint32_t b, uint32_t e) {
  // entries between ',', pieces between '.': 1 .. 18 digits each, at least two pieces per entry and a second piece that is
  // not zero -- what `pantas augment` writes.  build_attrs clusters the entries and divides by the sum of a cluster's counts
  // (scripts/call.py:15-21, :37-66): with every count positive nothing there can raise; an entry without a count, a sign or
  // a zero count goes to the host's scan, which follows the clustering and words the IndexError / ZeroDivisionError.
  uint32_t digits = 0, piece = 0;
  bool nonzero = false, count_ok = false;
  for (uint32_t c = b; c <= e; ++c) {
    const uint32_t ch = c < e ? T.at(c) : uint32_t(',');
    if (ch == ',' || ch == '.') {
      if (digits == 0) return false;
      if (piece == 1u) count_ok = nonzero;
      if (ch == ',') { if (piece == 0u || !count_ok) return false; piece = 0; }
      else ++piece;
      digits = 0; nonzero = false;
    }
    else if (ch - '0' <= 9u) { if (++digits > 18u) return false; nonzero |= ch != '0'; }
    else return false;
  }
  return true;
}
__global__ void __launch_bounds__(LPB) gfa_tags_kernel(const uint8_t* text, const uint64_t* ls, uint32_t n_lines, uint32_t stage_bytes, const uint8_t* kind,
                                                       const uint32_t* s_ord, const uint32_t* l_ord, int32_t* nc, uint32_t* nflags, uint64_t* ex_off, uint32_t* ex_len,
                                                       LinkTagRec* la_file, unsigned long long* odd, uint32_t* too_long_scratch) {
  const uint32_t groups = (n_lines + LPB - 1u) / LPB;
  for (uint32_t grp = blockIdx.x; grp < groups; grp += gridDim.x)
    staged_rounds(grp, text, ls, n_lines, stage_bytes, too_long_scratch, [&](uint32_t i, const uint32_t* words, uint32_t b, uint32_t e) {
      const uint8_t k = kind[i];
      if (k != GK_S && k != GK_L) return;
      TextWords T(words);
      const uint64_t off = ls[i];
      uint32_t end = e;
      while (end > b && tk_ws(T.at(end - 1u))) --end;
      uint32_t p = b;
      const uint32_t skip = k == GK_S ? 3u : 6u;
      for (uint32_t n = 0; n < skip; ++n) {
        while (p < end && tk_ws(T.at(p))) ++p;
        while (p < end && !tk_ws(T.at(p))) ++p;
      }
      uint32_t val = 0, fl = 0, sp_b = 0, sp_n = 0;       // NC / RC value; flags; span of EX / JN
      bool has_val = false, has_span = false, has_id = false, bad = false;
      for (;;) {
        while (p < end && tk_ws(T.at(p))) ++p;
        if (p >= end) break;
        const uint32_t tb = p;
        uint32_t c1 = 0, c2 = 0, colons = 0;
        while (p < end && !tk_ws(T.at(p))) {
          if (T.at(p) == ':') { if (colons == 0) c1 = p; else if (colons == 1) c2 = p; ++colons; }
          ++p;
        }
        if (colons != 2u) { bad = true; break; }            // (the reference: ValueError, call.py:28)
        if (c1 - tb != 2u) continue;
        const uint32_t n0 = T.at(tb), n1 = T.at(tb + 1u), vb = c2 + 1u, ve = p;
        const uint32_t name = n0 | (n1 << 8);
        const uint32_t own = k == GK_S ? ('N' | ('C' << 8)) : ('R' | ('C' << 8));
        if (name == ('N' | ('C' << 8)) || name == ('R' | ('C' << 8)) || name == ('L' | ('N' << 8))) {
          uint32_t v;
          if (!tag_plain_int(T, vb, ve, v)) { bad = true; break; }
          if (name == own) { val = v; has_val = true; }
        } else if (name == ('I' | ('L' << 8)) || name == ('O' | ('L' << 8))) {
          if (!tag_plain_hist(T, vb, ve)) { bad = true; break; }
        } else if (name == (k == GK_S ? ('E' | ('X' << 8)) : ('J' | ('N' << 8)))) { sp_b = vb; sp_n = ve - vb; has_span = true; }
        else if (k == GK_L && name == ('I' | ('D' << 8))) has_id = true;
      }
      if (bad) { atomicMin(odd, static_cast<unsigned long long>(off)); return; }
      (void)fl;
      if (k == GK_S) {
        const uint32_t node = s_ord[i];
        nc[node] = int32_t(val);
        nflags[node] = (has_val ? TAG_HAS_NC : 0u) | (has_span ? TAG_HAS_EX : 0u);
        ex_off[node] = off + (sp_b - b);
        ex_len[node] = sp_n;
      } else {
        LinkTagRec r;
        r.jn_off = has_span ? off + (sp_b - b) : 0ull;
        r.jn_len = has_span ? sp_n : 0u;
        r.has_jn = has_span; r.has_id = has_id; r.has_rc = has_val;
        r.rc = int32_t(val);
        la_file[l_ord[i]] = r;
      }
    });
}
// nodes with an EX tag, ascending: node, offset, length (positions from a scan of the flags)
__global__ void __launch_bounds__(TPB) gfa_ex_list_kernel(uint32_t V, const uint32_t* nflags_in, const uint32_t* pos, const uint64_t* ex_off, const uint32_t* ex_len,
                                                          uint32_t* o_node, uint64_t* o_off, uint32_t* o_len, uint32_t* nflags_out) {
  const uint32_t n = blockIdx.x * TPB + threadIdx.x;
  if (n >= V) return;
  const uint32_t f = nflags_in[n];
  nflags_out[n] = f & TAG_HAS_NC;
  if (f & TAG_HAS_EX) { const uint32_t q = pos[n]; o_node[q] = n; o_off[q] = ex_off[n]; o_len[q] = ex_len[n]; }
}
struct HasEx {
  __host__ __device__ uint32_t operator()(uint32_t f) const { return (f & TAG_HAS_EX) ? 1u : 0u; }
};
__global__ void __launch_bounds__(TPB) gfa_la_slots_kernel(uint32_t A, const uint32_t* oidx, const LinkTagRec* la_file, LinkTagRec* la_slot) {
  const uint32_t q = blockIdx.x * TPB + threadIdx.x;
  if (q < A) la_slot[q] = la_file[oidx[q]];
}

// S lines -> node arrays; id range; the name table is filled by a second kernel once base and size are known
__global__ void __launch_bounds__(TPB) gfa_nodes_kernel(uint32_t n_lines, const uint8_t* kind, const GfaLine* rec, const uint64_t* ls, const uint32_t* s_ord,
                                                        uint32_t* seg_len, uint64_t* seg_line_off, uint64_t* seg_name_off, uint16_t* seg_name_len,
                                                        unsigned long long* id_min, unsigned long long* id_max, uint32_t* len_max) {
  unsigned long long lo = ~0ull, hi = 0;
  uint32_t lm = 0;
  for (uint64_t i = uint64_t(blockIdx.x) * TPB + threadIdx.x; i < n_lines; i += uint64_t(gridDim.x) * TPB) {
    if (kind[i] != GK_S) continue;
    const GfaLine r = rec[i];
    const uint32_t node = s_ord[i];
    const uint32_t len = uint32_t(r.b & 0xFFFFFFFFull);
    seg_len[node] = len;
    seg_line_off[node] = ls[i];
    seg_name_off[node] = ls[i] + ((r.b >> 32) & 0xFFFFull);
    seg_name_len[node] = uint16_t(r.b >> 48);
    lo = min(lo, r.a); hi = max(hi, r.a); lm = max(lm, len);
  }
  for (int d = 32; d > 0; d >>= 1) {
    lo = min(lo, static_cast<unsigned long long>(__shfl_down(lo, d, 64)));
    hi = max(hi, static_cast<unsigned long long>(__shfl_down(hi, d, 64)));
    lm = max(lm, uint32_t(__shfl_down(lm, d, 64)));
  }
  if ((threadIdx.x & 63u) == 0) { atomicMin(id_min, lo); atomicMax(id_max, hi); atomicMax(len_max, lm); }
}
// where the P lines start (a handful): the host counts their tokens
__global__ void __launch_bounds__(TPB) gfa_plines_kernel(uint32_t n_lines, const uint8_t* kind, const uint64_t* ls, uint32_t cap, uint64_t* list, uint32_t* n) {
  for (uint64_t i = uint64_t(blockIdx.x) * TPB + threadIdx.x; i < n_lines; i += uint64_t(gridDim.x) * TPB) {
    if (kind[i] != GK_P) continue;
    const uint32_t k = atomicAdd(n, 1u);
    if (k < cap) { list[2 * k] = ls[i]; list[2 * k + 1] = ls[i + 1] - 1; }
  }
}
__global__ void __launch_bounds__(TPB) gfa_table_kernel(uint32_t n_lines, const uint8_t* kind, const GfaLine* rec, const uint32_t* s_ord, unsigned long long base,
                                                        uint32_t* table, unsigned long long* flags, const uint64_t* ls) {
  for (uint64_t i = uint64_t(blockIdx.x) * TPB + threadIdx.x; i < n_lines; i += uint64_t(gridDim.x) * TPB) {
    if (kind[i] != GK_S) continue;
    if (atomicCAS(&table[rec[i].a - base], NONE32, s_ord[i]) != NONE32) atomicMin(&flags[0], static_cast<unsigned long long>(ls[i]));     // a second S line of the name
  }
}
// L lines in file order: ends through the table
__global__ void __launch_bounds__(TPB) gfa_links_kernel(uint32_t n_lines, const uint8_t* kind, const GfaLine* rec, const uint64_t* ls, const uint32_t* l_ord,
                                                        unsigned long long base, unsigned long long size, const uint32_t* table, const uint64_t* seg_line_off,
                                                        uint32_t* fsrc, uint32_t* fdst, uint64_t* foff, uint32_t* fidx, uint32_t* out_cnt, uint32_t* in_cnt,
                                                        unsigned long long* flags, uint32_t* before) {
  for (uint64_t i = uint64_t(blockIdx.x) * TPB + threadIdx.x; i < n_lines; i += uint64_t(gridDim.x) * TPB) {
    if (kind[i] != GK_L) continue;
    const GfaLine r = rec[i];
    const uint32_t f = l_ord[i];
    uint32_t u = NONE32, v = NONE32;
    if (r.a >= base && r.a - base < size) u = table[r.a - base];
    if (r.b >= base && r.b - base < size) v = table[r.b - base];
    if (u == NONE32 || v == NONE32) { atomicMin(&flags[0], static_cast<unsigned long long>(ls[i])); u = v = 0; }      // an end without S line: the host words it
    else if (seg_line_off[u] > ls[i] || seg_line_off[v] > ls[i]) *before = 1u;
    fsrc[f] = u; fdst[f] = v; foff[f] = ls[i]; fidx[f] = f;
    atomicAdd(&out_cnt[u], 1u);
    atomicAdd(&in_cnt[v], 1u);
  }
}
__global__ void __launch_bounds__(TPB) gfa_out_kernel(uint32_t A, const uint32_t* oidx, const uint32_t* fdst, const uint64_t* foff, uint32_t* col, uint64_t* slot_off,
                                                      uint32_t* slot_of) {
  const uint32_t q = blockIdx.x * TPB + threadIdx.x;
  if (q >= A) return;
  const uint32_t f = oidx[q];
  col[q] = fdst[f];
  slot_off[q] = foff[f];
  slot_of[f] = q;
}
__global__ void __launch_bounds__(TPB) gfa_in_kernel(uint32_t A, const uint32_t* iidx, const uint32_t* fsrc, const uint32_t* slot_of, uint32_t* icol, uint32_t* islot) {
  const uint32_t q = blockIdx.x * TPB + threadIdx.x;
  if (q >= A) return;
  const uint32_t f = iidx[q];
  icol[q] = fsrc[f];
  islot[q] = slot_of[f];
}

// Device memory a load is done with is PARKED, not freed: on this driver a hipMalloc that is handed recently freed (and written)
// memory waits for it to be scrubbed -- 16 GB: 1.9 s, 36 GB in eight blocks: 2.7 - 3.0 s, where fresh memory takes a millisecond
// (tools/hipmalloc_probe.hip, profiles/r3_hipmalloc_probe.txt) -- and the loader's 60 - 85 GB were freed right in front of the
// hipMalloc of the graph arrays (1.6 s on some runs) and, half way through, of its own CSR buffers (1.1 - 1.3 s).  With 288 GB of
// HBM the blocks simply stay allocated until the handle they were made for is freed (release_parked_device_memory, capi.cpp)
// or the process ends; beyond 40 % of the device's memory they are freed at once.
std::mutex g_park_mu;
std::vector<void*> g_parked;
size_t g_parked_bytes = 0;
void park_or_free(void* p, size_t bytes) {
  if (!p) return;
  size_t free_b = 0, total_b = 0;
  const bool keep = !getenv("PANTAS_FREE_DEVICE_AT_ONCE") && hipMemGetInfo(&free_b, &total_b) == hipSuccess;
  std::lock_guard<std::mutex> lk(g_park_mu);
  if (keep && g_parked_bytes + bytes <= total_b / 5 * 2) { g_parked.push_back(p); g_parked_bytes += bytes; }
  else (void)hipFree(p);
}

struct DevMem {                      // plain device allocations of one load (there is no DeviceGraph and no pool yet)
  std::vector<std::pair<void*, size_t>> all;
  ~DevMem() { for (auto& b : all) park_or_free(b.first, b.second); }
  template <class X> X* get(size_t n) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(X);
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();                 // (the failure is handled here: later checks of the last error must not see it)
      release_parked_device_memory();          // out of memory with blocks parked: give them back and try once more
      HIP_OK(hipMalloc(&p, bytes));
    }
    all.emplace_back(p, bytes);
    return static_cast<X*>(p);
  }
  void drop(void*) {}                // (parked with the rest when the load ends: see above)
};

template <class X>
void scan_u32(DevMem& m, const uint32_t* in, uint32_t* out, size_t n_plus_1, hipStream_t s, X& tmp, size_t& tmp_bytes) {
  size_t need = 0;
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, in, out, int(n_plus_1), s));
  if (need > tmp_bytes) { tmp = m.get<uint8_t>(need); tmp_bytes = need; }
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(tmp, need, in, out, int(n_plus_1), s));
}

}  // namespace

size_t parked_device_bytes() {
  std::lock_guard<std::mutex> lk(g_park_mu);
  return g_parked_bytes;
}

hipError_t device_malloc_retry(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess || parked_device_bytes() == 0) return e;
  (void)hipGetLastError();
  release_parked_device_memory();
  return hipMalloc(p, bytes);
}

void release_parked_device_memory() {
  std::vector<void*> todo;
  {
    std::lock_guard<std::mutex> lk(g_park_mu);
    todo.swap(g_parked);
    g_parked_bytes = 0;
  }
  for (void* p : todo) (void)hipFree(p);
}

// called where the large allocations of a handle are made (graph arrays, call tables): parked blocks must never be the reason
// one of them fails
void release_parked_device_memory_if_low() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < total_b / 4) release_parked_device_memory();
}

// nullptr: not a graph for the device form (or no device) -- the caller takes the host loader
std::unique_ptr<Graph> load_graph_device(const char* path, int device, bool call_tags) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= device) return nullptr;
  PhaseTimer pt("load_graph_device");
  auto g = std::make_unique<Graph>();
  g->path = path;
  g->text.open(path);
  const char* data = g->text.data;
  const size_t size = g->text.size;
  if (size < (size_t(1) << 20) || size >= (size_t(1) << 44)) return nullptr;          // small graphs: the host is done before the copy is
  HIP_OK(hipSetDevice(device));
  hipStream_t s = take_stream(device);
  DevMem m;
  // ---- text
  const size_t padded = text_padded_bytes(size);
  uint8_t* d_text = m.get<uint8_t>(padded);
  HIP_OK(hipMemsetAsync(d_text + size, 0, padded - size, s));
  void* pinned[2] = {nullptr, nullptr};
  hipEvent_t pinned_done[2] = {nullptr, nullptr};
  size_t pinned_bytes = 0;
  // (pread of the file into the pinned buffers instead of a copy out of the mapping was measured: 1.6 - 1.9 s against 1.1 s for 19 GB)
  (void)upload_text(pinned, pinned_done, pinned_bytes, data, size, d_text, s, getenv("PANTAS_GFA_PREAD") ? g->text.fd : -1, false);
  for (int k = 0; k < 2; ++k) { if (pinned[k]) (void)hipHostFree(pinned[k]); if (pinned_done[k]) (void)hipEventDestroy(pinned_done[k]); }
  pt.lap("text -> HBM");
  // ---- line index
  const uint32_t tiles = uint32_t((size + TILE - 1) / TILE);
  uint8_t* scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  uint32_t* tile_cnt = m.get<uint32_t>(size_t(tiles) + 2);      // (+ the lone-'\r' flag behind the counts)
  uint32_t* tile_off = m.get<uint32_t>(size_t(tiles) + 1);
  HIP_OK(hipMemsetAsync(tile_cnt, 0, (size_t(tiles) + 2) * 4, s));
  hipLaunchKernelGGL(count_newlines_kernel, dim3(tiles), dim3(TPB), 0, s, reinterpret_cast<const uint4*>(d_text), tile_cnt, tile_cnt + tiles + 1, uint64_t(size));
  HIP_OK(hipGetLastError());
  scan_u32(m, tile_cnt, tile_off, size_t(tiles) + 1, s, scan_tmp, scan_tmp_bytes);
  uint32_t newlines = 0, lone_cr = 0;
  HIP_OK(hipMemcpyAsync(&newlines, tile_off + tiles, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(&lone_cr, tile_cnt + tiles + 1, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  if (lone_cr) return nullptr;                 // universal newlines: the host loader reads a copy with them applied
  const bool open_tail = data[size - 1] != '\n';
  const uint64_t n_lines64 = uint64_t(newlines) + (open_tail ? 1 : 0);
  if (n_lines64 >= 0x7FFFFFFFull) return nullptr;
  const uint32_t n_lines = uint32_t(n_lines64);
  uint64_t* ls = m.get<uint64_t>(size_t(n_lines) + 2);
  HIP_OK(hipMemsetAsync(ls, 0, 8, s));
  hipLaunchKernelGGL(line_starts_kernel, dim3(tiles), dim3(TPB), 0, s, reinterpret_cast<const uint4*>(d_text), tile_off, ls);
  HIP_OK(hipGetLastError());
  if (open_tail) {
    const uint64_t end1 = uint64_t(size) + 1;
    HIP_OK(hipMemcpyAsync(ls + n_lines, &end1, 8, hipMemcpyHostToDevice, s));
    HIP_OK(hipStreamSynchronize(s));
  }
  pt.lap("line index");
  // ---- every line: kind, numbers
  uint8_t* kind = m.get<uint8_t>(size_t(n_lines) + 1);
  GfaLine* rec = m.get<GfaLine>(n_lines);
  unsigned long long* flags = m.get<unsigned long long>(8);      // [0] odd line, [1] first short L, [2] id min, [3] id max, [4].lo len max, [5].lo link before segment
  HIP_OK(hipMemsetAsync(flags, 0xFF, 24, s));
  HIP_OK(hipMemsetAsync(flags + 3, 0, 40, s));
  uint32_t* too_long = reinterpret_cast<uint32_t*>(flags + 6);
  const uint32_t stage_bytes = uint32_t(std::min<uint64_t>(48 * 1024, std::max<uint64_t>(4096, ((size / std::max<uint32_t>(n_lines, 1)) * 96 + 2047) / 1024 * 1024)));
  {
    int per_cu = 0, cus = 0;
    uint32_t grid = 4096;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gfa_lines_kernel, int(LPB), stage_bytes) == hipSuccess && per_cu > 0 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      grid = uint32_t(per_cu) * uint32_t(cus);
    grid = std::min<uint32_t>(grid, (n_lines + LPB - 1) / LPB);
    hipLaunchKernelGGL(gfa_lines_kernel, dim3(std::max(grid, 1u)), dim3(LPB), stage_bytes, s, d_text, ls, n_lines, stage_bytes, kind, rec, flags, too_long);
    HIP_OK(hipGetLastError());
  }
  // S / L ordinals (file order) by scans of the kinds
  uint32_t* s_ord = m.get<uint32_t>(size_t(n_lines) + 1);
  uint32_t* l_ord = m.get<uint32_t>(size_t(n_lines) + 1);
  {
    // (two small transform kernels would do; hipcub's transform iterator keeps it in the scan)
    auto to_flag = [=](uint8_t want) { return hipcub::TransformInputIterator<uint32_t, KindIs, const uint8_t*>(kind, KindIs{want}); };
    size_t need = 0;
    auto it_s = to_flag(GK_S);
    auto it_l = to_flag(GK_L);
    HIP_OK(hipMemsetAsync(kind + n_lines, 0, 1, s));
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, it_s, s_ord, int(n_lines) + 1, s));
    if (need > scan_tmp_bytes) { scan_tmp = m.get<uint8_t>(need); scan_tmp_bytes = need; }
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, need, it_s, s_ord, int(n_lines) + 1, s));
    HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, need, it_l, l_ord, int(n_lines) + 1, s));
  }
  // the P lines' places
  constexpr uint32_t P_CAP = 1u << 16;
  uint64_t* d_plist = m.get<uint64_t>(size_t(2) * P_CAP);
  uint32_t* d_pn = m.get<uint32_t>(1);
  HIP_OK(hipMemsetAsync(d_pn, 0, 4, s));
  hipLaunchKernelGGL(gfa_plines_kernel, dim3(uint32_t(std::min<uint64_t>((uint64_t(n_lines) + TPB - 1) / TPB, 1u << 16))), dim3(TPB), 0, s, n_lines, kind, ls, P_CAP, d_plist, d_pn);
  HIP_OK(hipGetLastError());
  uint32_t n_p = 0;
  HIP_OK(hipMemcpyAsync(&n_p, d_pn, 4, hipMemcpyDeviceToHost, s));
  uint32_t counts[2] = {0, 0};
  unsigned long long hflags[8];
  HIP_OK(hipMemcpyAsync(&counts[0], s_ord + n_lines, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(&counts[1], l_ord + n_lines, 4, hipMemcpyDeviceToHost, s));
  HIP_OK(hipMemcpyAsync(hflags, flags, 64, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  pt.lap("lines classified");
  if (hflags[0] != ~0ull) return nullptr;        // an odd line (among them: an S / L line longer than the staging buffer)
  const uint32_t V = counts[0], A = counts[1];
  if (V == 0 || V >= NONE32 - 1 || A >= NONE32 - 1) return nullptr;
  if (n_p > P_CAP) return nullptr;                // (a graph of more than 65 536 paths: the host loader)
  std::vector<uint64_t> plist(size_t(2) * n_p);
  if (n_p) HIP_OK(hipMemcpy(plist.data(), d_plist, size_t(16) * n_p, hipMemcpyDeviceToHost));
  // ---- nodes
  uint32_t* seg_len = m.get<uint32_t>(V);
  uint64_t* seg_line_off = m.get<uint64_t>(V);
  uint64_t* seg_name_off = m.get<uint64_t>(V);
  uint16_t* seg_name_len = m.get<uint16_t>(V);
  const uint32_t lgrid = uint32_t(std::min<uint64_t>((uint64_t(n_lines) + TPB - 1) / TPB, 1u << 16));
  hipLaunchKernelGGL(gfa_nodes_kernel, dim3(lgrid), dim3(TPB), 0, s, n_lines, kind, rec, ls, s_ord, seg_len, seg_line_off, seg_name_off, seg_name_len, flags + 2, flags + 3,
                     reinterpret_cast<uint32_t*>(flags + 4));
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(hflags, flags, 64, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  const uint64_t lo = hflags[2], hi = hflags[3];
  if (hi - lo >= std::max<uint64_t>(uint64_t(V) * 16, 1u << 20)) return nullptr;      // (the host keeps such names in a hash map)
  const uint64_t tsize = hi - lo + 1;
  uint32_t* table = m.get<uint32_t>(tsize);
  HIP_OK(hipMemsetAsync(table, 0xFF, tsize * 4, s));
  hipLaunchKernelGGL(gfa_table_kernel, dim3(lgrid), dim3(TPB), 0, s, n_lines, kind, rec, s_ord, lo, table, flags, ls);
  HIP_OK(hipGetLastError());
  // ---- links
  uint32_t* fsrc = m.get<uint32_t>(A);
  uint32_t* fdst = m.get<uint32_t>(A);
  uint64_t* foff = m.get<uint64_t>(A);
  uint32_t* fidx = m.get<uint32_t>(A);
  uint32_t* out_cnt = m.get<uint32_t>(size_t(V) + 1);
  uint32_t* in_cnt = m.get<uint32_t>(size_t(V) + 1);
  uint32_t* row_ptr = m.get<uint32_t>(size_t(V) + 1);
  uint32_t* irow_ptr = m.get<uint32_t>(size_t(V) + 1);
  HIP_OK(hipMemsetAsync(out_cnt, 0, (size_t(V) + 1) * 4, s));
  HIP_OK(hipMemsetAsync(in_cnt, 0, (size_t(V) + 1) * 4, s));
  hipLaunchKernelGGL(gfa_links_kernel, dim3(lgrid), dim3(TPB), 0, s, n_lines, kind, rec, ls, l_ord, lo, tsize, table, seg_line_off, fsrc, fdst, foff, fidx, out_cnt, in_cnt,
                     flags, reinterpret_cast<uint32_t*>(flags + 5));
  HIP_OK(hipGetLastError());
  scan_u32(m, out_cnt, row_ptr, size_t(V) + 1, s, scan_tmp, scan_tmp_bytes);
  scan_u32(m, in_cnt, irow_ptr, size_t(V) + 1, s, scan_tmp, scan_tmp_bytes);
  HIP_OK(hipMemcpyAsync(hflags, flags, 64, hipMemcpyDeviceToHost, s));
  HIP_OK(hipStreamSynchronize(s));
  if (hflags[0] != ~0ull) return nullptr;             // a second S line of a name, a link end without S line
  pt.lap("nodes, name table, link ends");
  // ---- the tags `pantas call` reads, while the text is here (asked for by the command line that goes on to call)
  int32_t* tg_nc = nullptr;
  uint32_t* tg_nflags = nullptr;
  uint32_t* tg_ex_node = nullptr; uint64_t* tg_ex_off = nullptr; uint32_t* tg_ex_len = nullptr;
  LinkTagRec* la_file = nullptr;
  uint32_t n_ex = 0;
  bool have_tags = false;
  if (call_tags) {
    tg_nc = m.get<int32_t>(V);
    uint32_t* raw = m.get<uint32_t>(size_t(V) + 1);
    uint64_t* ex_off = m.get<uint64_t>(V);
    uint32_t* ex_len = m.get<uint32_t>(V);
    la_file = m.get<LinkTagRec>(A);
    HIP_OK(hipMemsetAsync(flags + 7, 0xFF, 8, s));
    {
      int per_cu = 0, cus = 0;
      uint32_t grid = 4096;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gfa_tags_kernel, int(LPB), stage_bytes) == hipSuccess && per_cu > 0 &&
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
        grid = uint32_t(per_cu) * uint32_t(cus);
      grid = std::min<uint32_t>(grid, (n_lines + LPB - 1) / LPB);
      hipLaunchKernelGGL(gfa_tags_kernel, dim3(std::max(grid, 1u)), dim3(LPB), stage_bytes, s, d_text, ls, n_lines, stage_bytes, kind, s_ord, l_ord, tg_nc, raw, ex_off, ex_len,
                         la_file, flags + 7, too_long);
      HIP_OK(hipGetLastError());
    }
    uint32_t* pos = m.get<uint32_t>(size_t(V) + 1);
    HIP_OK(hipMemsetAsync(raw + V, 0, 4, s));
    {
      hipcub::TransformInputIterator<uint32_t, HasEx, const uint32_t*> it(raw, HasEx{});
      size_t need = 0;
      HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, it, pos, int(V) + 1, s));
      if (need > scan_tmp_bytes) { scan_tmp = m.get<uint8_t>(need); scan_tmp_bytes = need; }
      HIP_OK(hipcub::DeviceScan::ExclusiveSum(scan_tmp, need, it, pos, int(V) + 1, s));
    }
    unsigned long long tag_odd = 0;
    HIP_OK(hipMemcpyAsync(&n_ex, pos + V, 4, hipMemcpyDeviceToHost, s));
    HIP_OK(hipMemcpyAsync(&tag_odd, flags + 7, 8, hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    have_tags = tag_odd == ~0ull;          // else: a tag of another form somewhere -- the host scans the lines (and words what it finds)
    if (have_tags) {
      tg_nflags = m.get<uint32_t>(V);
      tg_ex_node = m.get<uint32_t>(n_ex); tg_ex_off = m.get<uint64_t>(n_ex); tg_ex_len = m.get<uint32_t>(n_ex);
      hipLaunchKernelGGL(gfa_ex_list_kernel, dim3(blocks_for(V)), dim3(TPB), 0, s, V, raw, pos, ex_off, ex_len, tg_ex_node, tg_ex_off, tg_ex_len, tg_nflags);
      HIP_OK(hipGetLastError());
      HIP_OK(hipStreamSynchronize(s));
    } else { m.drop(tg_nc); m.drop(la_file); tg_nc = nullptr; la_file = nullptr; }
    m.drop(raw); m.drop(ex_off); m.drop(ex_len); m.drop(pos);
    pt.lap("tags of the S / L lines (for call)");
  }
  m.drop(kind); m.drop(rec); m.drop(s_ord); m.drop(l_ord); m.drop(d_text); m.drop(ls);
  int bits = 1;
  while (bits < 32 && (uint64_t(1) << bits) < uint64_t(V)) ++bits;
  uint32_t* col = m.get<uint32_t>(A);
  uint64_t* slot_off = m.get<uint64_t>(A);
  uint32_t* icol = m.get<uint32_t>(A);
  uint32_t* islot = m.get<uint32_t>(A);
  LinkTagRec* la_slot = nullptr;
  if (A) {
    uint32_t* key2 = m.get<uint32_t>(A);
    uint32_t* oidx = m.get<uint32_t>(A);
    uint32_t* iidx = m.get<uint32_t>(A);
    uint32_t* slot_of = m.get<uint32_t>(A);
    size_t need = 0;
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, fsrc, key2, fidx, oidx, int(A), 0, bits, s));
    uint8_t* sort_tmp = m.get<uint8_t>(need);
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(sort_tmp, need, fsrc, key2, fidx, oidx, int(A), 0, bits, s));          // rows: the links of a source in file order
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(sort_tmp, need, fdst, key2, fidx, iidx, int(A), 0, bits, s));          // in-lists: file order too (call.py:175)
    hipLaunchKernelGGL(gfa_out_kernel, dim3(blocks_for(A)), dim3(TPB), 0, s, A, oidx, fdst, foff, col, slot_off, slot_of);
    hipLaunchKernelGGL(gfa_in_kernel, dim3(blocks_for(A)), dim3(TPB), 0, s, A, iidx, fsrc, slot_of, icol, islot);
    HIP_OK(hipGetLastError());
    if (have_tags) {
      la_slot = m.get<LinkTagRec>(A);
      hipLaunchKernelGGL(gfa_la_slots_kernel, dim3(blocks_for(A)), dim3(TPB), 0, s, uint32_t(A), oidx, la_file, la_slot);
      HIP_OK(hipGetLastError());
    }
  }
  HIP_OK(hipStreamSynchronize(s));
  pt.lap("CSR (two stable sorts)");
  // ---- to the host
  g->V = V;
  g->A = A;
  g->n_lines = n_lines;
  g->n_s_lines = V;
  g->n_l_lines = A;
  g->names.dense = true;
  g->names.base = lo;
  g->max_seg_len = uint32_t(hflags[4]);
  g->link_before_segment = uint32_t(hflags[5]) != 0;
  g->first_short_l_off = hflags[1] == ~0ull ? UINT64_MAX : hflags[1];
  auto down = [&](auto& vec, const void* src, size_t n) {
    vec.resize(n);
    if (n) d2h_staged(vec.data(), src, n * sizeof(vec[0]), s);
  };
  down(g->names.table, table, tsize);
  down(g->seg_len, seg_len, V);
  down(g->seg_line_off, seg_line_off, V);
  down(g->seg_name_off, seg_name_off, V);
  down(g->seg_name_len, seg_name_len, V);
  down(g->row_ptr, row_ptr, size_t(V) + 1);
  down(g->col, col, A);
  down(g->slot_off, slot_off, A);
  down(g->irow_ptr, irow_ptr, size_t(V) + 1);
  down(g->icol, icol, A);
  down(g->islot, islot, A);
  if (have_tags) {
    auto ct = std::make_unique<CallTagScan>();
    down(ct->nc, tg_nc, V);
    down(ct->nflags, tg_nflags, V);
    down(ct->ex_node, tg_ex_node, n_ex);
    down(ct->ex_off, tg_ex_off, n_ex);
    down(ct->ex_len, tg_ex_len, n_ex);
    if (A) down(ct->la, la_slot, A);
    g->call_tags = std::move(ct);
  }
  HIP_OK(hipStreamSynchronize(s));
  pt.lap("arrays -> host");
  // ---- the P lines (a handful, each as long as a chromosome): exactly four tokens, or call.py:156 raises
  {
    g->n_p_lines = n_p;
    std::vector<uint64_t> bad(n_p, UINT64_MAX);
    parallel_chunks(n_p, unsigned(std::max<size_t>(1, std::min<size_t>(n_p, hw_threads()))), [&](unsigned, size_t b, size_t e) {
      for (size_t i = b; i < e; ++i) {
        Tokens tk(strip(Span{data + plist[2 * i], data + std::min<uint64_t>(plist[2 * i + 1], size)}));
        Span t;
        int n = 0;
        while (n < 5 && tk.next(t)) ++n;
        if (n != 4) bad[i] = plist[2 * i];
      }
    });
    uint64_t first = UINT64_MAX;
    for (uint64_t b : bad) first = std::min(first, b);
    g->first_bad_p_off = first;
  }
  pt.lap("P lines (host)");
  return g;
}

// loads this file's code object ahead of its first real kernel (warm_up_device, on a thread beside the graph parse)
__global__ void warm_tokenize_kernel() {}
void warm_up_tokenize_module(hipStream_t s) { hipLaunchKernelGGL(warm_tokenize_kernel, dim3(1), dim3(64), 0, s); }

}  // namespace pantas
