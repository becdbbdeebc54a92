// GAF text -> binary alignment records (host side, multi-threaded).
// Covers the text half of the reference's per-line work: column split, mapq / unmapped /
// divergence filters, cs parsing, the two-op clipping rule and the path split
// (scripts/alignments_augmentation_from_gaf.py:142-197, :10-50).  The numeric half (:209-363)
// runs on the GPU over these records.
//
// Record layout (structure of arrays, what the kernels read with coalesced loads):
//   hdr[n]        LineHdr {start, tail, line_id, flags}
//   node_off[n+1] offsets into nodes[]   (node indices of the path, consecutive repeats removed)
//   op_off[n+1]   offsets into ops[]     ((length << 3) | op code)
#pragma once
#include "common.h"
#include "graph.h"
#include "walk.h"
#include <memory>

namespace pantas {

struct GafBatch {
  uint64_t lines_total = 0;   // text lines seen
  uint64_t rejected = 0;      // mapq below threshold (reference's "Rejected alignments")
  uint64_t skipped = 0;       // unmapped ('*' path) or dv > 0.1
  std::vector<LineHdr> hdr;
  std::vector<uint32_t> node_off, op_off, nodes, ops;
  uint64_t del_lines = 0;     // records with a '-' op
  uint64_t del_nodes = 0;     // sum of path lengths over those records
  uint64_t bytes = 0;         // text bytes tokenised
  uint64_t first_record_line = UINT64_MAX;   // line id of the first record in FILE order (none: UINT64_MAX); see walk.h WALK_TILDE_FIRST
  uint64_t n_short = 0;       // after bucket_by_locus: records [0, n_short) are the short-walk class (walk.h)
  // A line the reference raises on while splitting / filtering it.  The batch then holds the records of the
  // lines BEFORE it: they still have to be walked, because an earlier line that only fails in the walk is what
  // the reference would report (it handles the file line by line).
  std::shared_ptr<Error> deferred;
  size_t size() const { return hdr.size(); }
};

// Reorders the records of a batch: short-walk class first, then the rest; inside each class by the lowest end node of
// the path (stable), so that a workgroup's consecutive records touch the same few hundred counters and graph rows;
// inside every block of 2048 of a class by path length, so that the lanes of a wave loop alike.  The
// result of a run does not depend on record order (line_id carries the file order).
void bucket_by_locus(GafBatch& b, unsigned threads = 0);

// Tokenises the byte range [data, data+size) (whole lines).  first_line_id = index of its
// first line in the file.  A line the reference raises on ends the batch and is reported in out.deferred.
void tokenize_gaf(const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id,
                  unsigned threads, GafBatch& out);

// The lengths that leak into the lines whose cs starts with a '~' op (walk.h TildeCarry; the reference's `curr_cigar_len` is a variable
// of main() that outlives its line): for every such line of [data, data + size), whose first line has id first_line_id, the length the
// last line in front of it that took an op in hand left behind -- found by walking backwards from the line, one line at a time, with the
// host tokeniser and walk_line.  Sorted by line id; TILDE_UNBOUND where no such line exists (the reference: UnboundLocalError).  A
// slow, sequential look that only a file with such lines pays for.
std::vector<TildeEntry> tilde_carries(const Graph& g, const char* data, size_t size, int thr, uint64_t first_line_id);

}  // namespace pantas
