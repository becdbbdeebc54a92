// One GAF text line -> one alignment record, written so that a GPU thread can do it: no allocation,
// no strings, a single forward scan per column group.  tokenize.hip runs it twice per line (measure, then
// write at the scanned offsets); tests/hostsim compiles the same header for the CPU and checks it, line by
// line, against the host tokeniser (gaf.cpp), which is an independent restatement of the same reference
// lines (scripts/alignments_augmentation_from_gaf.py:142-197, :10-50).
//
// Differences in mechanism, not in result:
//   * the "cs:Z:" removal (:158) is a byte stream that steps over every "cs:Z:" instead of a copy;
//   * float(dv) > 0.1 (:180) is decided on the decimal digits: the double nearest to the text exceeds the
//     double 0.1 exactly when the text exceeds the midpoint between 0.1 and the next double (a tie rounds
//     to the even mantissa, which is 0.1's);
//   * the first two cs ops are held back until a third shows up, because the clipping rule (:40-50, :164-167)
//     only applies to lists of exactly two.
#pragma once
#include "walk.h"

namespace pantas {

enum : uint32_t { TOK_KEEP = 0, TOK_REJECT = 1, TOK_SKIP = 2, TOK_ERROR = 3 };

// why a line makes the reference raise (the host turns the code into the message)
enum : uint32_t {
  TE_NONE = 0,
  TE_COLUMNS = 1,       // IndexError: fewer than 12 columns (:143)
  TE_MAPQ = 2,          // ValueError: int(tokens[11]) (:143)
  TE_POSITIONS = 3,     // ValueError: int() of path length / start / end (:151-153)
  TE_OP_LEN = 4,        // Unsupported: cs op longer than 2^29-1
  TE_NO_DV = 5,         // ValueError: float("*") (:180)
  TE_NO_NODE = 6,       // AssertionError (:197)
  TE_UNKNOWN_NODE = 7,  // KeyError: path node without S line (:212)
  TE_OFFSETS = 8,       // Unsupported: start / end offsets beyond 32 bits
  TE_LONE_CR = 9,       // Unsupported: a '\r' inside the line (Python's universal newlines would end the line there)
};

// id - base -> node for graphs whose segment names are canonical decimals (graph.h NameIndex, dense form)
struct DenseNames {
  const uint32_t* table;
  uint64_t base;
  uint64_t size;
};

constexpr uint32_t TK_NONE32 = 0xFFFFFFFFu;

PANTAS_HD bool tk_ws(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
PANTAS_HD bool tk_opchar(uint32_t c) { return c == '=' || c == ':' || c == '*' || c == '+' || c == '-' || c == '~'; }
PANTAS_HD uint32_t tk_opcode(uint32_t c) {
  return c == ':' ? OP_COLON : c == '*' ? OP_STAR : c == '+' ? OP_PLUS : c == '-' ? OP_MINUS : c == '=' ? OP_EQUAL : OP_TILDE;
}
PANTAS_HD bool tk_digit(uint32_t c) { return c >= '0' && c <= '9'; }

// Python int(str), ASCII subset: optional sign, digits with single '_' separators; beyond 2^60 saturates at +-2^61.
PANTAS_HD bool tk_py_int(const uint8_t* c, const uint8_t* e, int64_t& v) {
  if (c >= e) return false;
  bool neg = false;
  if (*c == '+' || *c == '-') { neg = (*c == '-'); ++c; }
  if (c >= e || !tk_digit(*c)) return false;
  uint64_t x = 0;
  bool prev_us = false, big = false;
  for (; c < e; ++c) {
    const uint32_t ch = *c;
    if (ch == '_') { if (prev_us) return false; prev_us = true; continue; }
    if (!tk_digit(ch)) return false;
    prev_us = false;
    if (x > (uint64_t(1) << 60)) big = true;
    if (!big) x = x * 10 + uint64_t(ch - '0');
  }
  if (prev_us) return false;
  if (big) x = uint64_t(1) << 61;
  v = neg ? -int64_t(x) : int64_t(x);
  return true;
}

PANTAS_HD uint32_t tk_lookup(const DenseNames& nt, const uint8_t* b, const uint8_t* e) {
  const size_t n = size_t(e - b);
  if (n == 0 || n > 18) return TK_NONE32;
  if (n > 1 && b[0] == '0') return TK_NONE32;
  uint64_t x = 0;
  for (const uint8_t* c = b; c < e; ++c) {
    if (!tk_digit(*c)) return TK_NONE32;
    x = x * 10 + uint64_t(*c - '0');
  }
  if (x < nt.base || x - nt.base >= nt.size) return TK_NONE32;
  return nt.table[x - nt.base];
}

// digit i (0-based, after "0.") of the midpoint between the double 0.1 and the next double above it:
// 0.100000000000000012490009027033011079765856266021728515625 (57 digits, exact)
PANTAS_HD uint32_t tk_dv_mid_digit(uint32_t i) {
  if (i >= 57) return 0;
  const uint64_t w = i < 16 ? 0x1000000000000000ull : i < 32 ? 0x1249000902703301ull : i < 48 ? 0x1079765856266021ull : 0x7285156250000000ull;
  return uint32_t(w >> (4 * (15 - (i & 15)))) & 15u;
}

// s points at the first digit after "dv:f:"; true iff float(<digits>[.<digits>]) > 0.1
PANTAS_HD bool tk_dv_above_tenth(const uint8_t* s, const uint8_t* e) {
  bool int_nonzero = false;
  const uint8_t* t = s;
  while (t < e && tk_digit(*t)) { int_nonzero |= *t != '0'; ++t; }
  if (int_nonzero) return true;
  if (!(t + 1 < e && *t == '.' && tk_digit(t[1]))) return false;   // 0 > 0.1 is false
  ++t;
  uint32_t i = 0;
  int cmp = 0;   // sign of (text - midpoint) over the digits seen so far
  while (t < e && tk_digit(*t)) {
    if (cmp == 0) {
      const uint32_t d = uint32_t(*t - '0'), m = tk_dv_mid_digit(i);
      cmp = d > m ? 1 : d < m ? -1 : 0;
    }
    ++t;
    ++i;
  }
  if (cmp == 0 && i < 57) cmp = -1;   // the midpoint has further non-zero digits
  return cmp > 0;
}

struct LineSummary {
  uint32_t status, err;
  uint32_t n_nodes, n_ops, flags;
  int32_t start, tail;
};

// Out concept:  void op(uint32_t i, uint32_t word);  void ops_done(uint32_t n_ops);  void node(uint32_t i, uint32_t idx);
// -- every op of the line (in order), then their number once, then every path node (in order)
struct CountOnly {
  PANTAS_HD void node(uint32_t, uint32_t) {}
  PANTAS_HD void op(uint32_t, uint32_t) {}
  PANTAS_HD void ops_done(uint32_t) {}
};

template <class Out>
struct OpEmitter {
  Out& out;
  uint32_t seen = 0, a = 0, b = 0, written = 0, stars = 0, indels = 0, others = 0;
  bool has_del = false, slow = false;
  uint64_t total = 0;
  PANTAS_HD explicit OpEmitter(Out& o) : out(o) {}
  PANTAS_HD void settle(uint32_t w) {
    const uint32_t code = w & 7u, len = w >> 3;
    has_del |= code == OP_MINUS;
    if (len == 0 || code == OP_TILDE) slow = true;
    if (code == OP_STAR) ++stars;
    else if (code == OP_PLUS || code == OP_MINUS) ++indels;
    else if (code != OP_COLON) ++others;
    total += len;
    out.op(written++, w);
  }
  PANTAS_HD void push(uint32_t w) {
    if (seen == 0) a = w;
    else if (seen == 1) b = w;
    else {
      if (seen == 2) { settle(a); settle(b); }
      settle(w);
    }
    ++seen;
  }
  // the two-op clipping rule; may move the start position
  PANTAS_HD void finish(int64_t& start) {
    if (seen == 1) settle(a);
    else if (seen == 2) {
      if ((a & 7u) == OP_PLUS && (b & 7u) == OP_COLON) { start += int64_t(a >> 3); settle(b); }
      else if ((a & 7u) == OP_COLON && (b & 7u) == OP_PLUS) settle(a);
      else { settle(a); settle(b); }
    }
  }
};

// bytes of a cs field with every "cs:Z:" stepped over (Python's left-to-right str.replace)
struct CsBytes {
  const uint8_t* s;
  const uint8_t* e;
  PANTAS_HD int next() {
    while (e - s >= 5 && s[0] == 'c' && s[1] == 's' && s[2] == ':' && s[3] == 'Z' && s[4] == ':') s += 5;
    if (s >= e) return -1;
    return int(*s++);
  }
};

// [b, e) = one line without its '\n'.
// check_cr: the text is known to hold a '\r' that no '\n' follows (the copy to the device looked): find the line.
template <class Out>
PANTAS_HD void parse_gaf_line(const uint8_t* b, const uint8_t* e, int thr, const DenseNames& names, Out& out, LineSummary& r, bool check_cr = true) {
  r.status = TOK_ERROR; r.err = TE_NONE; r.n_nodes = 0; r.n_ops = 0; r.flags = 0; r.start = 0; r.tail = 0;
  if (check_cr) {
    // [b, e) has no '\n': any '\r' before its last byte is a lone one (as its last byte it is the CR of a CRLF)
    for (const uint8_t* q = b; q + 1 < e; ++q)
      if (*q == '\r') { r.err = TE_LONE_CR; return; }
  }
  // ---- columns 0..11 (str.split(): runs of whitespace, leading/trailing ignored)
  const uint8_t *p5b = b, *p5e = b, *p6b = b, *p6e = b, *p7b = b, *p7e = b, *p8b = b, *p8e = b, *p11b = b, *p11e = b;
  const uint8_t* p = b;
  uint32_t ntok = 0;
  while (ntok < 12) {
    while (p < e && tk_ws(*p)) ++p;
    if (p >= e) break;
    const uint8_t* s = p;
    while (p < e && !tk_ws(*p)) ++p;
    if (ntok == 5) { p5b = s; p5e = p; }
    else if (ntok == 6) { p6b = s; p6e = p; }
    else if (ntok == 7) { p7b = s; p7e = p; }
    else if (ntok == 8) { p8b = s; p8e = p; }
    else if (ntok == 11) { p11b = s; p11e = p; }
    ++ntok;
  }
  if (ntok < 12) { r.err = TE_COLUMNS; return; }
  const uint8_t* const opt = p;   // optional fields start here
  int64_t mapq;
  if (!tk_py_int(p11b, p11e, mapq)) { r.err = TE_MAPQ; return; }
  if (mapq < int64_t(thr)) { r.status = TOK_REJECT; return; }
  if (p5e - p5b == 1 && *p5b == '*') { r.status = TOK_SKIP; return; }
  int64_t path_len, start, endpos;
  if (!tk_py_int(p6b, p6e, path_len) || !tk_py_int(p7b, p7e, start) || !tk_py_int(p8b, p8e, endpos)) { r.err = TE_POSITIONS; return; }

  // ---- cs: first "cs:" of any optional field, to the end of that field
  OpEmitter<Out> em(out);
  {
    bool found = false;
    const uint8_t* q = opt;
    while (q < e && !found) {
      while (q < e && tk_ws(*q)) ++q;
      const uint8_t* fb = q;
      while (q < e && !tk_ws(*q)) ++q;
      const uint8_t* fe = q;
      for (const uint8_t* c = fb; c + 3 <= fe; ++c) {
        if (c[0] == 'c' && c[1] == 's' && c[2] == ':') {
          found = true;
          CsBytes st{c, fe};
          int ch = st.next();
          while (ch >= 0 && !tk_opchar(uint32_t(ch))) ch = st.next();
          while (ch >= 0) {
            const uint32_t op = uint32_t(ch);
            uint64_t run = 0, val = 0;
            bool digits = true, over = false;
            ch = st.next();
            while (ch >= 0 && !tk_opchar(uint32_t(ch))) {
              ++run;
              if (!tk_digit(uint32_t(ch))) digits = false;
              else if (digits && !over) {
                val = val * 10 + uint64_t(ch - '0');
                if (val > OP_MAX_LEN) over = true;
              }
              ch = st.next();
            }
            uint64_t v;
            if (op == '*') v = 1;
            else if (run > 0 && digits) {
              if (over) { r.err = TE_OP_LEN; return; }
              v = val;
            } else {
              v = run;
              if (v > OP_MAX_LEN) { r.err = TE_OP_LEN; return; }
            }
            em.push(uint32_t(v << 3) | tk_opcode(op));
          }
          break;
        }
      }
    }
    if (!found) em.push((1u << 3) | OP_STAR);
  }
  em.finish(start);
  out.ops_done(em.written);

  // ---- dv: first "dv:f:<digit>" of any optional field
  {
    bool found = false, above = false;
    const uint8_t* q = opt;
    while (q < e && !found) {
      while (q < e && tk_ws(*q)) ++q;
      const uint8_t* fb = q;
      while (q < e && !tk_ws(*q)) ++q;
      const uint8_t* fe = q;
      for (const uint8_t* c = fb; c + 6 <= fe; ++c) {
        if (c[0] == 'd' && c[1] == 'v' && c[2] == ':' && c[3] == 'f' && c[4] == ':' && tk_digit(c[5])) {
          found = true;
          above = tk_dv_above_tenth(c + 5, fe);
          break;
        }
      }
    }
    if (!found) { r.err = TE_NO_DV; return; }
    if (above) { r.status = TOK_SKIP; return; }
  }

  // ---- path
  const bool rev = *p5b != '>';
  const uint8_t sep = rev ? '<' : '>';
  const uint8_t* q = p5b;
  while (q < p5e && *q != sep) ++q;
  if (q >= p5e) { r.err = TE_NO_NODE; return; }
  uint32_t n_nodes = 0, last = TK_NONE32;
  while (q < p5e) {
    const uint8_t* s = q + 1;
    const uint8_t* t = s;
    while (t < p5e && *t != sep) ++t;
    const uint32_t idx = tk_lookup(names, s, t);
    if (idx == TK_NONE32) { r.err = TE_UNKNOWN_NODE; return; }
    if (n_nodes == 0 || idx != last) { out.node(n_nodes++, idx); last = idx; }
    q = t;
  }
  const int64_t tail = path_len - endpos;
  if (start < -2147483648ll || start > 2147483647ll || tail < -2147483648ll || tail > 2147483647ll) { r.err = TE_OFFSETS; return; }
  uint32_t flags = rev ? LINE_REV : 0u;
  if (em.slow) flags |= LINE_SLOW;
  flags |= line_class_flags(em.written, em.stars, em.indels, em.others, em.slow, em.total, start, tail);
  if (em.has_del) flags |= LINE_HAS_DEL;
  r.status = TOK_KEEP;
  r.n_nodes = n_nodes;
  r.n_ops = em.written;
  r.flags = flags;
  r.start = int32_t(start);
  r.tail = int32_t(tail);
}

}  // namespace pantas
