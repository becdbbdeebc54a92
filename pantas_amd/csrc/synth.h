// Scalable synthetic workload generator (SURVEY.md 8-d recipe; C++ sibling of pantas_amd/synth.py
// for the BASELINE.json scales).  Produces the three TEXT inputs of the hot path in memory --
// annotated GFA, GTF, GAF -- so that benchmarks and scale tests go through the product's real
// parsers.  Deterministic from (seed, sizes); genes and GAF lines are generated in parallel
// from per-gene / per-line random streams.
#pragma once
#include "common.h"
#include <memory>

namespace pantas {

struct SynthParams {
  uint64_t seed = 1;
  uint32_t n_genes = 100;
  uint64_t n_lines = 10000;
  uint64_t line_offset = 0;         // generate lines [line_offset, line_offset + n_lines) of the read stream
  uint64_t base_id = 1000000;       // first node id (fixed-width ids: choose 10^k with k > log10(V))
  double bubble_rate = 0.06;        // SNP bubbles per eligible slot
  uint32_t max_exons = 9;
  uint32_t intron_long_max = 40;    // long introns draw U[10, intron_long_max] slots (30 % of introns)
  uint32_t intergenic_min = 4, intergenic_max = 12;
  uint32_t read_len = 100;
  double zipf_a = 1.2;              // expression skew over genes
  double p_lowmapq = 0.05, p_unmapped = 0.01, p_sub = 0.30, p_indel = 0.05, p_highdv = 0.01;
  double p_novel_skip = 0.04, p_novel_ce = 0.02, p_intron = 0.06, p_novel_ss = 0.04, p_clip = 0.03;
  uint32_t n_chrom = 1;             // genes are dealt to chromosomes in contiguous blocks
  int with_path = 1;                // P line per chromosome
  int gaf_only = 0;                 // 1: no GFA / GTF text (further blocks of the read stream of a graph generated before)
  // Deletion skew of real data: a true indel variant is ONE (node, position) key that every read covering it increments,
  // while the plain recipe drops deletions at random positions.  With p_del_hot > 0 that share of the deletion reads is
  // aimed at `hot_sites` fixed sites on the first transcripts of the most expressed genes (and the rest of the deletion
  // reads stay random).
  double p_del_hot = 0.0;
  uint32_t hot_sites = 1000;
  // share of the mapped reads with TWO indels in their cs (":a+xx:b-yy:c"): what the short-walk class of the augment kernels
  // does not take (walk.h line_class_flags) -- the recipe's own reads carry one indel at most
  double p_two_indels = 0.0;
};

// text buffer without the zero-fill of std::string::resize (the GFA is ~12 GB at the human scale)
struct TextBuf {
  std::unique_ptr<char[]> p;
  size_t n = 0;
  const char* data() const { return p.get(); }
  size_t size() const { return n; }
};

struct SynthOutput {
  TextBuf gfa, gaf;
  std::string gtf;
  uint64_t nodes = 0, links = 0, junction_links = 0, transcripts = 0, gaf_lines = 0;
};

void synth_generate(const SynthParams& p, SynthOutput& out, unsigned threads = 0);

}  // namespace pantas
