// Block-sparse exchange of the dense coverage counters between the GPUs of one node.
//
// A read-sharded replicate ends with ONE sum over [{NC, IL0, OLL, RC first slot} per node | RC per slot] (reference: the single-process dicts of
// scripts/alignments_augmentation_from_gaf.py:263-363 become per-GPU arrays that must be added up).  On a spliced
// pangenome the reads cover exons only: at human whole-genome scale about 1 % of the 10^9 counters is ever non-zero,
// and xGMI is point-to-point -- a ring all-reduce of the whole 4 GB array is bound by one link (SURVEY 8-e,
// MI355X_MICROARCH.md).  So the ranks exchange only the blocks somebody touched:
//
//   touched_bitmap   one bit per block of B counters (B = 8 .. 64 words) that holds a non-zero word    reads W
//   (caller)         all-gather of the bitmaps (W / B bits per rank)
//   pack             union of the bitmaps, popcount + exclusive scan, touched blocks -> staging            reads W/B bits + touched
//   (caller)         all-reduce(sum) of the staging buffer
//   unpack           staging -> counters
//
// Every rank derives the same union, hence the same staging layout; no index lists travel.
// The collectives themselves are the caller's (RCCL through torch.distributed, pantas_amd/distributed.py).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "augment.h"
#include "device_batch.h"

namespace pantas {

struct CounterExchange {
  uint32_t group = 0;            // lanes (16-byte loads) per block: block = 4 * group words
  uint64_t words = 0;            // counters incl. padding (multiple of COUNTER_PAD_WORDS)
  uint64_t bitmap_words = 0;
  uint32_t* bitmap = nullptr;    // this rank's touched blocks
  uint32_t* uni = nullptr;       // union over ranks
  uint32_t* cnt = nullptr;       // popcount per union word, then its exclusive scan ([bitmap_words + 1])
  uint32_t* off = nullptr;
  void* scan_tmp = nullptr;
  size_t scan_tmp_bytes = 0;
  uint4* staging = nullptr;
  uint64_t staging_blocks = 0;
  bool from_kernels = false;     // the last bitmap was the one the kernels kept (folded), not a pass over the counters
  ScratchPool* pool = nullptr;
  void drop() {
    for (void* p : {(void*)bitmap, (void*)uni, (void*)cnt, (void*)off, scan_tmp, (void*)staging})
      if (p && pool) pool->give(p);
    bitmap = uni = cnt = off = nullptr; scan_tmp = nullptr; scan_tmp_bytes = 0; staging = nullptr; staging_blocks = 0;
  }
};

void CounterExchangeDeleter::operator()(CounterExchange* x) const {
  if (!x) return;
  x->drop();
  delete x;
}

namespace {

// One bitmap word per wave and iteration: 32 blocks of GROUP 16-byte pieces = GROUP / 2 loads of 64 lanes, each giving
// 64 / GROUP block bits through a ballot folded over the GROUP lanes of a block.
template <uint32_t GROUP>
__global__ __launch_bounds__(256) void touched_bitmap_kernel(const uint4* __restrict__ c, uint64_t bitmap_words, uint32_t* __restrict__ bitmap) {
  static_assert(GROUP == 2 || GROUP == 4 || GROUP == 8 || GROUP == 16, "block of 8, 16, 32 or 64 counters");
  constexpr uint32_t LOADS = GROUP / 2, BITS = 64 / GROUP;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t(blockIdx.x) * 256u + threadIdx.x) >> 6;
  const uint64_t n_waves = (uint64_t(gridDim.x) * 256u) >> 6;
  for (uint64_t w = wave; w < bitmap_words; w += n_waves) {
    const uint4* base = c + w * (32u * GROUP);
    uint32_t out = 0;
#pragma unroll
    for (uint32_t j = 0; j < LOADS; ++j) {
      const uint4 v = base[j * 64u + lane];
      unsigned long long m = __ballot((v.x | v.y | v.z | v.w) != 0u);
#pragma unroll
      for (uint32_t s = 1; s < GROUP; s <<= 1) m |= m >> s;
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t i = 0; i < BITS; ++i) bits |= uint32_t((m >> (GROUP * i)) & 1ull) << i;
      out |= bits << (j * BITS);
    }
    if (lane == 0) bitmap[w] = out;
  }
}

__global__ __launch_bounds__(256) void union_kernel(const uint32_t* __restrict__ maps, uint32_t n_maps, uint64_t bitmap_words,
                                                    uint32_t* __restrict__ uni, uint32_t* __restrict__ cnt) {
  for (uint64_t w = uint64_t(blockIdx.x) * 256u + threadIdx.x; w <= bitmap_words; w += uint64_t(gridDim.x) * 256u) {
    uint32_t u = 0;
    if (w < bitmap_words)
      for (uint32_t r = 0; r < n_maps; ++r) u |= maps[r * bitmap_words + w];
    if (w < bitmap_words) uni[w] = u;
    cnt[w] = __popc(u);          // cnt[bitmap_words] = 0: its scan is the total
  }
}

// thread per 16 bytes of counters; TO_STAGING: counters -> staging, else staging -> counters
template <bool TO_STAGING>
__global__ __launch_bounds__(256) void move_blocks_kernel(uint4* __restrict__ c, uint4* __restrict__ staging, const uint32_t* __restrict__ uni,
                                                          const uint32_t* __restrict__ off, uint64_t n16, uint32_t log_group) {
  for (uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x; t < n16; t += uint64_t(gridDim.x) * 256u) {
    const uint64_t blk = t >> log_group;
    const uint32_t u = uni[blk >> 5], bit = uint32_t(blk) & 31u;
    if (!((u >> bit) & 1u)) continue;
    const uint64_t dst = ((uint64_t(off[blk >> 5]) + __popc(u & ((1u << bit) - 1u))) << log_group) + (t & ((1u << log_group) - 1u));
    if (TO_STAGING) staging[dst] = c[t]; else c[t] = staging[dst];
  }
}

// the kernels' own bitmap (one bit per `fine` counters) as the exchange's (one bit per fine * FOLD counters): a bit is the OR of FOLD
template <uint32_t FOLD>
__global__ __launch_bounds__(256) void fold_bitmap_kernel(const uint32_t* __restrict__ fine, uint64_t fine_words, uint64_t out_words, uint32_t* __restrict__ out) {
  for (uint64_t w = uint64_t(blockIdx.x) * 256u + threadIdx.x; w < out_words; w += uint64_t(gridDim.x) * 256u) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 0; k < FOLD; ++k) {
      const uint64_t i = w * FOLD + k;
      uint32_t x = i < fine_words ? fine[i] : 0u;
      // OR of every group of FOLD neighbouring bits, gathered into 32 / FOLD bits
#pragma unroll
      for (uint32_t s = 1; s < FOLD; s <<= 1) x |= x >> s;
      uint32_t packed = 0;
#pragma unroll
      for (uint32_t j = 0; j < 32u / FOLD; ++j) packed |= ((x >> (j * FOLD)) & 1u) << j;
      r |= packed << (k * (32u / FOLD));
    }
    out[w] = r;
  }
}
// bits of `want` that `have` lacks (PANTAS_EXCHANGE_VERIFY)
__global__ __launch_bounds__(256) void missing_bits_kernel(const uint32_t* __restrict__ want, const uint32_t* __restrict__ have, uint64_t words, unsigned long long* __restrict__ missing) {
  unsigned long long n = 0;
  for (uint64_t w = uint64_t(blockIdx.x) * 256u + threadIdx.x; w < words; w += uint64_t(gridDim.x) * 256u) n += __popc(want[w] & ~have[w]);
  if (n) atomicAdd(missing, n);
}

uint32_t grid_for(uint64_t items, uint32_t cap = 1u << 20) { return uint32_t(std::min<uint64_t>((items + 255) / 256, cap)); }

}  // namespace

void exchange_bitmap(DeviceGraph& dg, CounterExchangePtr& xp, uint32_t block_words, void** dev_bitmap, uint64_t* bitmap_words) {
  if (block_words != 8 && block_words != 16 && block_words != 32 && block_words != 64)
    throw Error("ValueError", "counter exchange block must be 8, 16, 32 or 64 words");
  const DeviceGraphInfo gi = graph_info(dg);
  HIP_OK(hipSetDevice(gi.device));
  const CounterPtrs cp = counter_ptrs(dg);
  if (!xp) xp.reset(new CounterExchange());
  CounterExchange& x = *xp;
  x.drop();
  x.pool = &scratch_pool(dg);
  x.group = block_words / 4;
  const uint64_t w = cp.words;
  x.words = (w + COUNTER_PAD_WORDS - 1) / COUNTER_PAD_WORDS * COUNTER_PAD_WORDS;
  if (x.words == 0) x.words = COUNTER_PAD_WORDS;
  x.bitmap_words = x.words / (32ull * block_words);
  x.bitmap = static_cast<uint32_t*>(x.pool->take(x.bitmap_words * 4));
  const uint64_t waves_needed = x.bitmap_words;
  const uint32_t blocks = uint32_t(std::min<uint64_t>((waves_needed + 3) / 4, 1u << 16));
  const uint4* c = static_cast<const uint4*>(cp.base);
  auto by_reading_the_counters = [&](uint32_t* out) {
    switch (x.group) {
      case 2: hipLaunchKernelGGL(touched_bitmap_kernel<2>, dim3(blocks), dim3(256), 0, gi.stream, c, x.bitmap_words, out); break;
      case 4: hipLaunchKernelGGL(touched_bitmap_kernel<4>, dim3(blocks), dim3(256), 0, gi.stream, c, x.bitmap_words, out); break;
      case 8: hipLaunchKernelGGL(touched_bitmap_kernel<8>, dim3(blocks), dim3(256), 0, gi.stream, c, x.bitmap_words, out); break;
      default: hipLaunchKernelGGL(touched_bitmap_kernel<16>, dim3(blocks), dim3(256), 0, gi.stream, c, x.bitmap_words, out); break;
    }
    HIP_OK(hipGetLastError());
  };
  // Round 5: when the kernels kept the bitmap themselves (augment.h set_track_touched) it is only folded to this exchange's block
  // size -- 20 MB read instead of the 5 GB of counters of a whole-genome graph (0.94 ms per replicate).  A kept bitmap may hold a
  // block whose counters are all zero (a flushed zero): harmless, the block is summed like any other.  PANTAS_EXCHANGE_VERIFY=1 also
  // reads the counters and refuses to go on when a non-zero block is missing from the kept bitmap -- a counter update site that
  // does not mark would otherwise be a silently wrong sum.
  const TouchedBitmap kept = touched_bitmap_of(dg);
  x.from_kernels = kept.bits != nullptr;
  if (kept.bits) {
    const uint32_t fold = block_words / kept.block_words;
    const uint32_t fb = grid_for(x.bitmap_words, 4096);
    switch (fold) {
      case 1: hipLaunchKernelGGL(fold_bitmap_kernel<1>, dim3(fb), dim3(256), 0, gi.stream, kept.bits, kept.words, x.bitmap_words, x.bitmap); break;
      case 2: hipLaunchKernelGGL(fold_bitmap_kernel<2>, dim3(fb), dim3(256), 0, gi.stream, kept.bits, kept.words, x.bitmap_words, x.bitmap); break;
      case 4: hipLaunchKernelGGL(fold_bitmap_kernel<4>, dim3(fb), dim3(256), 0, gi.stream, kept.bits, kept.words, x.bitmap_words, x.bitmap); break;
      default: hipLaunchKernelGGL(fold_bitmap_kernel<8>, dim3(fb), dim3(256), 0, gi.stream, kept.bits, kept.words, x.bitmap_words, x.bitmap); break;
    }
    HIP_OK(hipGetLastError());
    if (const char* v = getenv("PANTAS_EXCHANGE_VERIFY"); v && atoi(v) != 0) {
      uint32_t* check = static_cast<uint32_t*>(x.pool->take(x.bitmap_words * 4));
      unsigned long long* missing = static_cast<unsigned long long*>(x.pool->take(8));
      HIP_OK(hipMemsetAsync(missing, 0, 8, gi.stream));
      by_reading_the_counters(check);
      hipLaunchKernelGGL(missing_bits_kernel, dim3(grid_for(x.bitmap_words, 4096)), dim3(256), 0, gi.stream, check, x.bitmap, x.bitmap_words, missing);
      HIP_OK(hipGetLastError());
      unsigned long long m = 0;
      HIP_OK(hipMemcpyAsync(&m, missing, 8, hipMemcpyDeviceToHost, gi.stream));
      HIP_OK(hipStreamSynchronize(gi.stream));
      x.pool->give(check); x.pool->give(missing);
      if (m) throw Error("InternalError", std::to_string(m) + " counter blocks hold counts but are missing from the bitmap the kernels kept (a counter update site that does not mark)");
    }
  } else {
    by_reading_the_counters(x.bitmap);
  }
  HIP_OK(hipStreamSynchronize(gi.stream));     // the caller's collective runs on another stream
  *dev_bitmap = x.bitmap;
  *bitmap_words = x.bitmap_words;
}

void exchange_pack(DeviceGraph& dg, CounterExchange& x, const void* dev_bitmaps, uint32_t n_bitmaps, void** dev_staging, uint64_t* staging_words) {
  if (!x.bitmap) throw Error("ProtocolError", "counter exchange: pack before bitmap");
  if (!dev_bitmaps || n_bitmaps == 0) throw Error("ValueError", "counter exchange: no bitmaps");
  const DeviceGraphInfo gi = graph_info(dg);
  HIP_OK(hipSetDevice(gi.device));
  HIP_OK(hipDeviceSynchronize());              // the gathered bitmaps were written by the caller's collective
  ScratchPool& pool = *x.pool;
  if (!x.uni) {
    x.uni = static_cast<uint32_t*>(pool.take(x.bitmap_words * 4));
    x.cnt = static_cast<uint32_t*>(pool.take((x.bitmap_words + 1) * 4));
    x.off = static_cast<uint32_t*>(pool.take((x.bitmap_words + 1) * 4));
  }
  hipLaunchKernelGGL(union_kernel, dim3(grid_for(x.bitmap_words + 1)), dim3(256), 0, gi.stream,
                     static_cast<const uint32_t*>(dev_bitmaps), n_bitmaps, x.bitmap_words, x.uni, x.cnt);
  HIP_OK(hipGetLastError());
  if (x.bitmap_words + 1 > 0x7FFFFFFFull) throw Error("Unsupported", "counter exchange: bitmap too large");
  size_t need = 0;
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, need, x.cnt, x.off, int(x.bitmap_words + 1), gi.stream));
  if (need > x.scan_tmp_bytes) {
    if (x.scan_tmp) pool.give(x.scan_tmp);
    x.scan_tmp = pool.take(need);
    x.scan_tmp_bytes = need;
  }
  HIP_OK(hipcub::DeviceScan::ExclusiveSum(x.scan_tmp, need, x.cnt, x.off, int(x.bitmap_words + 1), gi.stream));
  uint32_t total = 0;
  HIP_OK(hipMemcpyAsync(&total, x.off + x.bitmap_words, 4, hipMemcpyDeviceToHost, gi.stream));
  HIP_OK(hipStreamSynchronize(gi.stream));
  if (x.staging) { pool.give(x.staging); x.staging = nullptr; }
  x.staging_blocks = total;
  x.staging = static_cast<uint4*>(pool.take(std::max<uint64_t>(1, uint64_t(total)) * x.group * 16));
  if (total) {
    const uint32_t lg = x.group == 2 ? 1u : x.group == 4 ? 2u : x.group == 8 ? 3u : 4u;
    const CounterPtrs cp = counter_ptrs(dg);
    hipLaunchKernelGGL(move_blocks_kernel<true>, dim3(grid_for(x.words / 4)), dim3(256), 0, gi.stream,
                       static_cast<uint4*>(cp.base), x.staging, x.uni, x.off, x.words / 4, lg);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(gi.stream));
  }
  *dev_staging = x.staging;
  *staging_words = uint64_t(total) * x.group * 4;
}

void exchange_unpack(DeviceGraph& dg, CounterExchange& x) {
  if (!x.staging) throw Error("ProtocolError", "counter exchange: unpack before pack");
  const DeviceGraphInfo gi = graph_info(dg);
  HIP_OK(hipSetDevice(gi.device));
  HIP_OK(hipDeviceSynchronize());              // the staging buffer was summed by the caller's collective
  if (x.staging_blocks) {
    const uint32_t lg = x.group == 2 ? 1u : x.group == 4 ? 2u : x.group == 8 ? 3u : 4u;
    const CounterPtrs cp = counter_ptrs(dg);
    hipLaunchKernelGGL(move_blocks_kernel<false>, dim3(grid_for(x.words / 4)), dim3(256), 0, gi.stream,
                       static_cast<uint4*>(cp.base), x.staging, x.uni, x.off, x.words / 4, lg);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(gi.stream));
  }
  x.drop();
}

void exchange_drop(CounterExchange& x) { x.drop(); }
void* exchange_stream(DeviceGraph& dg) { return graph_info(dg).stream; }
void exchange_stream_wait(DeviceGraph& dg, void* stream) {
  HIP_OK(hipSetDevice(graph_info(dg).device));
  HIP_OK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
}
void* exchange_scratch_take(DeviceGraph& dg, size_t bytes) {
  HIP_OK(hipSetDevice(graph_info(dg).device));
  return scratch_pool(dg).take(bytes);
}
void exchange_scratch_give(DeviceGraph& dg, void* p) { scratch_pool(dg).give(p); }

}  // namespace pantas
