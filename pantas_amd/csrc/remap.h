// `pantas remap` in native code (remap.cpp): scripts/remap.py for quantification CSVs of whole-genome size.
#pragma once
#include <cstdint>
#include <string>

namespace pantas {

// Reference-genome coordinates for the junctions of a `pantas quant` CSV, as CSV on fd_out: the bytes scripts/remap.py prints.
// min_intron: -i (novel rows with a region shorter than this are dropped).  Throws Error with the class the reference raises
// (ValueError, IndexError, AttributeError, AssertionError); nothing is written then.
void remap_run(const std::string& csv_path, const std::string& gtf_path, int64_t min_intron, int fd_out, unsigned threads = 0);

}  // namespace pantas
