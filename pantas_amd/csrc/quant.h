// `pantas quant` in native code (quant.cpp): scripts/quantify.py for event files of whole-genome size.
#pragma once
#include <string>
#include <vector>

namespace pantas {

// PSI / dPSI of the events of the replicates `c1` against those of `c2` (paths of `pantas call` CSVs) as CSV on fd_out: the bytes
// scripts/quantify.py prints.  w: -w (minimum coverage of both junction groups), both: --both.  Throws Error with the class the
// reference raises (TypeError: a row without 14 columns; ValueError: a coverage / node id that is no number; AttributeError: an
// unknown event type).
// repr(x) of a Python float (shortest digits that round-trip, Python's choice of notation)
void py_float_repr(std::string& s, double x);

void quant_run(const std::vector<std::string>& c1, const std::vector<std::string>& c2, int w, bool both, int fd_out, unsigned threads = 0);

}  // namespace pantas
