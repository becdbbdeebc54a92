// GFA -> CSR node/link layout (host side).  Replaces the reference's dict-of-objects
// graph model (scripts/gfautils.py:92-128) and the inline S/L parsers of the two hot-path
// scripts (scripts/alignments_augmentation_from_gaf.py:121-126,377-424; scripts/call.py:145-181).
//
// Layout (all indices uint32):
//   node index      = order of first appearance of the segment name in an S line
//   seg_len[V]      = len(sequence) of the LAST S line carrying that name (dict overwrite)
//   out CSR         = row_ptr[V+1], col[A]; one slot per L line whose two ends have S lines,
//                     rows in L-line file order (adjacency order the caller relies on,
//                     call.py:174-175); duplicate (u,v) lines keep separate slots
//   slot_off[A]     = byte offset (stripped line start) of the L line behind a slot
//   in CSR          = irow_ptr[V+1], icol[A] (source node), islot[A] (out-CSR slot)
// The GFA text stays mmapped: the augment writer re-emits it line by line.
#pragma once
#include "common.h"
#include <memory>
#include <unordered_map>

namespace pantas {

struct NameIndex {
  // dense table for canonical decimal names, else a string map
  bool dense = false;
  uint64_t base = 0;
  RawVector<uint32_t> table;                             // id - base -> node, NONE32 = absent
  std::unordered_map<std::string_view, uint32_t> map;    // keys point into the mmap
  uint32_t find(Span name) const {
    if (dense) {
      uint64_t id;
      if (!parse_canonical_u64(name, id) || id < base || id - base >= table.size()) return NONE32;
      return table[id - base];
    }
    auto it = map.find(name.sv());
    return it == map.end() ? NONE32 : it->second;
  }
};

// Tags of the S / L lines as `pantas call` reads them (scripts/call.py:145-181), taken on the GPU while the text of the graph is
// in HBM for the loader (tokenize.hip gfa_tags_kernel; round 3).  Only made for a text every tag of which is of the plain form the
// writers produce -- anything else leaves it absent and call_host.cpp scans the lines itself (and words the error).
struct LinkTagRec { uint64_t jn_off; uint32_t jn_len; bool has_jn, has_id, has_rc; int32_t rc; };      // = call_host.h LinkAttr
struct CallTagScan {
  RawVector<int32_t> nc;             // [V] NC value (0 without the tag)
  RawVector<uint32_t> nflags;        // [V] call_device.h N_HAS_NC
  RawVector<uint32_t> ex_node;       // nodes whose S line carries an EX tag, ascending ...
  RawVector<uint64_t> ex_off;        // ... offset and length of its value in the text
  RawVector<uint32_t> ex_len;
  RawVector<LinkTagRec> la;          // [A] by graph slot
};

struct Graph {
  MappedFile text;
  RawVector<char> text_owned;      // the text with universal newlines applied (common.h lone_cr): `text` then points here, not at the file

  std::string path;
  uint32_t V = 0;          // distinct segment names
  uint64_t A = 0;          // adjacency slots (L lines with both ends known)
  uint64_t n_lines = 0, n_s_lines = 0, n_l_lines = 0, n_p_lines = 0;
  NameIndex names;
  // (RawVector: gigabytes each at whole-genome scale -- huge pages, no serial zero-fill; every element is written by the loader)
  RawVector<uint32_t> seg_len;       // [V]
  RawVector<uint64_t> seg_line_off;  // [V] stripped-line offset of the defining (last) S line
  RawVector<uint64_t> seg_name_off;  // [V] offset of the name token (first S line)
  RawVector<uint16_t> seg_name_len;  // [V]
  RawVector<uint32_t> row_ptr, col;  // out CSR
  RawVector<uint64_t> slot_off;      // [A]
  RawVector<uint32_t> irow_ptr, icol, islot;  // in CSR
  // facts the two consumers validate differently
  bool dup_names = false;              // some name has more than one S line
  bool dup_ws = false;                 // ... and one of those lines stands behind leading whitespace (augment and call read such a line differently)
  RawVector<uint64_t> seg_first_off;   // [V], only with dup_names: offset of the FIRST S line of the name (seg_line_off: the last)
  bool link_before_segment = false;    // an L line precedes the S line of one of its ends
  uint64_t links_unknown_end = 0;      // L lines naming a segment without S line
  uint64_t first_bad_p_off = UINT64_MAX;   // first P line without exactly 4 tokens
  uint64_t first_short_l_off = UINT64_MAX; // first non-bare L line with fewer than 6 tokens
  uint64_t ws_only_seg_off = UINT64_MAX;   // first S line behind leading whitespace whose name no plain S line defines
  uint32_t max_seg_len = 0;
  mutable std::unique_ptr<CallTagScan> call_tags;   // load_graph_device(.., call_tags = true); build_call_tables takes it

  Span name_of(uint32_t node) const {
    const char* b = text.data + seg_name_off[node];
    return Span{b, b + seg_name_len[node]};
  }
};

// what the augment script cannot take although call.py can (both the C ABI and the CPU harness of the tests ask this)
inline void require_augmentable(const Graph& g) {
  if (g.ws_only_seg_off != UINT64_MAX)     // alignments_augmentation_from_gaf.py:123 misses the line, :387 dies on it
    throw Error("KeyError", "GFA byte " + std::to_string(g.ws_only_seg_off) + ": S line behind leading whitespace names a segment no plain S line defines");
}

// Parses the GFA at `path`.  Throws Error on inputs the reference's first GFA pass dies on.
std::unique_ptr<Graph> load_graph(const char* path, unsigned threads = 0);
// The same arrays built on the GPU (tokenize.hip): for the graphs a graph builder writes; nullptr for anything else -- no device, a
// small file, names that are no decimals in a dense range, duplicate or whitespace-led S lines, lone carriage returns -- which the
// host loader then takes.
std::unique_ptr<Graph> load_graph_device(const char* path, int device, bool call_tags = false);
// the device memory of finished loads is parked, not freed (tokenize.hip): given back here
void release_parked_device_memory();
size_t parked_device_bytes();
void release_parked_device_memory_if_low();      // ... when less than a quarter of the device's memory is free
// Same over text already in memory (the buffer must outlive the graph).
std::unique_ptr<Graph> load_graph_mem(const char* data, size_t size, unsigned threads = 0);

// Binary cache of the parsed arrays (graph_cache.cpp).  save: false when the graph cannot be cached (names that are not
// decimals, text not from a file).  load: nullptr when there is no usable cache for exactly this GFA file.
bool save_graph_cache(const Graph& g, const char* cache_path);
std::unique_ptr<Graph> load_graph_cached(const char* gfa_path, const char* cache_path, unsigned threads = 0);

}  // namespace pantas
