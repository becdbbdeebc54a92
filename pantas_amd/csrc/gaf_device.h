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
  const uint32_t* blocks = nullptr;   // optional: per NAME_BLOCK entries of `table`, table[first] when table[first + j] = table[first] + j for the whole block
};
// table[rel] (rel < size)
PANTAS_HD uint32_t tk_node_of(const DenseNames& nt, uint32_t rel) {
  if (nt.blocks) {
    const uint32_t f = nt.blocks[rel / NAME_BLOCK];
    if (f != NAME_BLOCK_MIXED) return f + rel % NAME_BLOCK;
  }
  return nt.table[rel];
}

constexpr uint32_t TK_NONE32 = 0xFFFFFFFFu;

PANTAS_HD bool tk_ws(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
PANTAS_HD bool tk_opchar(uint32_t c) { return c == '=' || c == ':' || c == '*' || c == '+' || c == '-' || c == '~'; }
PANTAS_HD uint32_t tk_opcode(uint32_t c) {
  return c == ':' ? OP_COLON : c == '*' ? OP_STAR : c == '+' ? OP_PLUS : c == '-' ? OP_MINUS : c == '=' ? OP_EQUAL : OP_TILDE;
}
PANTAS_HD bool tk_digit(uint32_t c) { return c >= '0' && c <= '9'; }

// The bytes of a line, read through ALIGNED 32-bit words: one load serves four byte reads.  On the GPU the text of a wave's
// lines sits in LDS and every byte the parser looks at is the end of a dependent chain (load, compare, branch) -- with one
// LDS round trip per byte the kernel waited for LDS 570 SIMD cycles per byte and lane (round 2: 25 ms per parse of 7.4 GB).
// Positions are byte offsets from `w`; the words that hold the first and the last byte of a line are read whole, so up to
// three bytes in front of and behind the line must be readable (padded text, 16-byte staging; the host callers pad).
struct TextWords {
  const uint32_t* w;
  uint32_t idx = 0xFFFFFFFFu, cur = 0;
  PANTAS_HD explicit TextWords(const uint32_t* words) : w(words) {}
  PANTAS_HD uint32_t at(uint32_t pos) {
    const uint32_t i = pos >> 2;
    if (i != idx) { cur = w[i]; idx = i; }
    return (cur >> ((pos & 3u) << 3)) & 0xFFu;
  }
};

// Python int(str), ASCII subset: optional sign, digits with single '_' separators; beyond 2^60 saturates at +-2^61.
PANTAS_HD bool tk_py_int(TextWords& T, uint32_t c, uint32_t e, int64_t& v) {
  if (c >= e) return false;
  bool neg = false;
  uint32_t ch = T.at(c);
  if (ch == '+' || ch == '-') { neg = (ch == '-'); ++c; }
  if (c >= e || !tk_digit(T.at(c))) return false;
  uint64_t x = 0;
  bool prev_us = false, big = false;
  for (; c < e; ++c) {
    ch = T.at(c);
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

PANTAS_HD uint32_t tk_lookup(const DenseNames& nt, TextWords& T, uint32_t b, uint32_t e) {
  const uint32_t n = e - b;
  if (n == 0 || n > 18) return TK_NONE32;
  if (n > 1 && T.at(b) == '0') return TK_NONE32;
  uint64_t x = 0;
  if (n <= 9) {      // (fits 32 bits: no 64-bit multiply per digit)
    uint32_t y = 0;
    for (uint32_t c = b; c < e; ++c) {
      const uint32_t ch = T.at(c);
      if (!tk_digit(ch)) return TK_NONE32;
      y = y * 10u + (ch - '0');
    }
    x = y;
  } else {
    for (uint32_t c = b; c < e; ++c) {
      const uint32_t ch = T.at(c);
      if (!tk_digit(ch)) return TK_NONE32;
      x = x * 10 + uint64_t(ch - '0');
    }
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
PANTAS_HD bool tk_dv_above_tenth(TextWords& T, uint32_t s, uint32_t e) {
  bool int_nonzero = false;
  uint32_t t = s;
  while (t < e && tk_digit(T.at(t))) { int_nonzero |= T.at(t) != '0'; ++t; }
  if (int_nonzero) return true;
  if (!(t + 1 < e && T.at(t) == '.' && tk_digit(T.at(t + 1)))) return false;   // 0 > 0.1 is false
  ++t;
  uint32_t i = 0;
  int cmp = 0;   // sign of (text - midpoint) over the digits seen so far
  while (t < e && tk_digit(T.at(t))) {
    if (cmp == 0) {
      const uint32_t d = T.at(t) - '0', m = tk_dv_mid_digit(i);
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

// Out concept:  void node(uint32_t i, uint32_t idx);  void nodes_done(uint32_t n_nodes);  void op(uint32_t i, uint32_t word);
// -- every path node (in order), then their number once, then every op of the line (in order).  (parse_gaf_line_bytewise emits
// the ops first and calls ops_done between: an Out that cares about the order is for one of the two forms.)
struct CountOnly {
  PANTAS_HD void node(uint32_t, uint32_t) {}
  PANTAS_HD void op(uint32_t, uint32_t) {}
  PANTAS_HD void ops_done(uint32_t) {}
  PANTAS_HD void nodes_done(uint32_t) {}
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
  TextWords& T;
  uint32_t s, e;
  PANTAS_HD int next() {
    while (e - s >= 5 && T.at(s) == 'c' && T.at(s + 1) == 's' && T.at(s + 2) == ':' && T.at(s + 3) == 'Z' && T.at(s + 4) == ':') s += 5;
    if (s >= e) return -1;
    return int(T.at(s++));
  }
};

// [b, e) = one line without its '\n', as byte positions in T.
// check_cr: the text is known to hold a '\r' that no '\n' follows (the copy to the device looked): find the line.
// The plain form: one byte at a time, every rule spelled out.  The kernels run parse_gaf_line_words below, which finds columns
// and tags a word at a time; tests/hostsim holds the two to the same result on every line it sees.
template <class Out>
PANTAS_HD void parse_gaf_line_bytewise(TextWords& T, const uint32_t b, const uint32_t e, int thr, const DenseNames& names, Out& out, LineSummary& r, bool check_cr = true) {
  r.status = TOK_ERROR; r.err = TE_NONE; r.n_nodes = 0; r.n_ops = 0; r.flags = 0; r.start = 0; r.tail = 0;
  if (check_cr) {
    // [b, e) has no '\n': any '\r' before its last byte is a lone one (as its last byte it is the CR of a CRLF)
    for (uint32_t q = b; q + 1 < e; ++q)
      if (T.at(q) == '\r') { r.err = TE_LONE_CR; return; }
  }
  // ---- columns 0..11 (str.split(): runs of whitespace, leading/trailing ignored)
  uint32_t p5b = b, p5e = b, p6b = b, p6e = b, p7b = b, p7e = b, p8b = b, p8e = b, p11b = b, p11e = b;
  uint32_t p = b;
  uint32_t ntok = 0;
  while (ntok < 12) {
    while (p < e && tk_ws(T.at(p))) ++p;
    if (p >= e) break;
    const uint32_t s = p;
    while (p < e && !tk_ws(T.at(p))) ++p;
    if (ntok == 5) { p5b = s; p5e = p; }
    else if (ntok == 6) { p6b = s; p6e = p; }
    else if (ntok == 7) { p7b = s; p7e = p; }
    else if (ntok == 8) { p8b = s; p8e = p; }
    else if (ntok == 11) { p11b = s; p11e = p; }
    ++ntok;
  }
  if (ntok < 12) { r.err = TE_COLUMNS; return; }
  const uint32_t opt = p;   // optional fields start here
  int64_t mapq;
  if (!tk_py_int(T, p11b, p11e, mapq)) { r.err = TE_MAPQ; return; }
  if (mapq < int64_t(thr)) { r.status = TOK_REJECT; return; }
  if (p5e - p5b == 1 && T.at(p5b) == '*') { r.status = TOK_SKIP; return; }
  int64_t path_len, start, endpos;
  if (!tk_py_int(T, p6b, p6e, path_len) || !tk_py_int(T, p7b, p7e, start) || !tk_py_int(T, p8b, p8e, endpos)) { r.err = TE_POSITIONS; return; }

  // ---- cs: first "cs:" of any optional field, to the end of that field
  OpEmitter<Out> em(out);
  {
    bool found = false;
    uint32_t q = opt;
    while (q < e && !found) {
      while (q < e && tk_ws(T.at(q))) ++q;
      const uint32_t fb = q;
      while (q < e && !tk_ws(T.at(q))) ++q;
      const uint32_t fe = q;
      for (uint32_t c = fb; c + 3 <= fe; ++c) {
        if (T.at(c) == 'c' && T.at(c + 1) == 's' && T.at(c + 2) == ':') {
          found = true;
          CsBytes st{T, c, fe};
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
    uint32_t q = opt;
    while (q < e && !found) {
      while (q < e && tk_ws(T.at(q))) ++q;
      const uint32_t fb = q;
      while (q < e && !tk_ws(T.at(q))) ++q;
      const uint32_t fe = q;
      for (uint32_t c = fb; c + 6 <= fe; ++c) {
        if (T.at(c) == 'd' && T.at(c + 1) == 'v' && T.at(c + 2) == ':' && T.at(c + 3) == 'f' && T.at(c + 4) == ':' && tk_digit(T.at(c + 5))) {
          found = true;
          above = tk_dv_above_tenth(T, c + 5, fe);
          break;
        }
      }
    }
    if (!found) { r.err = TE_NO_DV; return; }
    if (above) { r.status = TOK_SKIP; return; }
  }

  // ---- path
  const bool rev = T.at(p5b) != '>';
  const uint32_t sep = rev ? '<' : '>';
  uint32_t q = p5b;
  while (q < p5e && T.at(q) != sep) ++q;
  if (q >= p5e) { r.err = TE_NO_NODE; return; }
  uint32_t n_nodes = 0, last = TK_NONE32;
  while (q < p5e) {
    const uint32_t s = q + 1;
    uint32_t t = s;
    while (t < p5e && T.at(t) != sep) ++t;
    const uint32_t idx = tk_lookup(names, T, s, t);
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

// ---- a word at a time ---------------------------------------------------------------------------------------------------
// 0x80 in every byte of w that is Python whitespace (tk_ws: 9..13, 28..32).  x has bit 7 set in every byte, so the four
// subtractions never borrow across bytes: bit 7 of a byte of (x - n) says "the low seven bits are >= n".
PANTAS_HD uint32_t sw_ws(uint32_t w) {
  const uint32_t x = (w & 0x7F7F7F7Fu) | 0x80808080u;
  const uint32_t ge9 = x - 0x09090909u, ge14 = x - 0x0E0E0E0Eu, ge28 = x - 0x1C1C1C1Cu, ge33 = x - 0x21212121u;
  return ~w & 0x80808080u & ge9 & ~ge33 & (~ge14 | ge28);
}
// 0x80 in every byte of w equal to the byte replicated in c4 (exact: no carries leave a byte)
PANTAS_HD uint32_t sw_eq(uint32_t w, uint32_t c4) {
  const uint32_t z = w ^ c4;
  return ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z | 0x7F7F7F7Fu);
}
// 0x80 in every byte of word i (bytes 4 i .. 4 i + 3) that lies outside [b, e)
PANTAS_HD uint32_t sw_outside(uint32_t i, uint32_t b, uint32_t e) {
  const uint32_t p = i << 2;
  const uint32_t lo = b > p ? (b - p < 4u ? b - p : 4u) : 0u;            // bytes in front of b
  const uint32_t hi = e < p + 4u ? (p + 4u - e < 4u ? p + 4u - e : 4u) : 0u;   // bytes from e on
  const uint32_t mlo = lo ? (lo >= 4u ? 0x80808080u : 0x80808080u >> (32u - 8u * lo)) : 0u;
  const uint32_t mhi = hi ? (hi >= 4u ? 0x80808080u : 0x80808080u << (32u - 8u * hi)) : 0u;
  return mlo | mhi;
}
PANTAS_HD uint32_t sw_ctz(uint32_t x) { return uint32_t(__builtin_ctz(x)); }

// four bytes of the text from byte position p on (the words that hold them are read whole)
PANTAS_HD uint32_t sw_bytes4(const TextWords& T, uint32_t p) {
  const uint32_t i = p >> 2, o = (p & 3u) << 3;
  const uint64_t two = uint64_t(T.w[i]) | (uint64_t(T.w[i + 1u]) << 32);
  return uint32_t(two >> o);
}
// up to four ASCII digits in the LOW n bytes of x (first digit in byte 0) as a number; `ok` is cleared when a byte is no digit
PANTAS_HD uint32_t sw_digits4(uint32_t x, uint32_t n, bool& ok) {
  // the n bytes to the top of a word, '0' below them (n == 0: "0000")
  const uint32_t y = uint32_t(((uint64_t(x) << 32) | 0x30303030ull) >> (8u * n));
  ok &= ((((y & 0xF0F0F0F0u) ^ 0x30303030u) | (((y & 0x0F0F0F0Fu) + 0x06060606u) & 0xF0F0F0F0u)) == 0u);
  uint32_t d = y & 0x0F0F0F0Fu;
  d = d * 10u + (d >> 8);                                 // bytes 0 and 2: the two pairs of digits
  return (d & 0xFFu) * 100u + ((d >> 16) & 0xFFu);
}
// a token of one to nine plain digits as a number (what the mappers write); false: not that -- the general int() has to look
PANTAS_HD bool sw_uint9(const TextWords& T, uint32_t tb, uint32_t te, int64_t& v) {
  const uint32_t n = te - tb;
  if (n == 0u || n > 9u) return false;
  bool ok = true;
  const uint32_t n0 = n < 4u ? n : 4u, n1 = n > 4u ? (n - 4u < 4u ? n - 4u : 4u) : 0u, n2 = n > 8u ? 1u : 0u;
  uint32_t x = sw_digits4(sw_bytes4(T, tb), n0, ok);
  if (n1) x = x * (n1 == 1u ? 10u : n1 == 2u ? 100u : n1 == 3u ? 1000u : 10000u) + sw_digits4(sw_bytes4(T, tb + 4u), n1, ok);
  if (n2) x = x * 10u + sw_digits4(sw_bytes4(T, tb + 8u), 1u, ok);
  v = int64_t(x);
  return ok;
}
PANTAS_HD bool tk_int_fast(TextWords& T, uint32_t tb, uint32_t te, int64_t& v) { return sw_uint9(T, tb, te, v) || tk_py_int(T, tb, te, v); }

// the cs ops of a line as the word-at-a-time parser emits them.  The first two are held back until a third shows up (the rule
// for lists of exactly two, cigar_clipping :40-50, may drop one: nothing is written that the final list does not hold); what the
// list adds up to is counted as the ops come and set right when the rule strikes
template <class Out>
struct OpList {
  Out& out;
  uint32_t n = 0, a = 0, b = 0, stars = 0, indels = 0, others = 0;
  bool has_del = false, slow = false;
  uint64_t total = 0;
  PANTAS_HD explicit OpList(Out& o) : out(o) {}
  PANTAS_HD void push(uint32_t w) {
    const uint32_t code = w & 7u, len = w >> 3;
    has_del |= code == OP_MINUS;
    slow |= len == 0u || code == OP_TILDE;
    stars += code == OP_STAR ? 1u : 0u;
    indels += (code == OP_PLUS || code == OP_MINUS) ? 1u : 0u;
    others += (code == OP_EQUAL || code == OP_TILDE) ? 1u : 0u;
    total += len;
    if (n == 0u) a = w;
    else if (n == 1u) b = w;
    else {
      if (n == 2u) { out.op(0, a); out.op(1, b); }
      out.op(n, w);
    }
    ++n;
  }
  PANTAS_HD void finish(int64_t& start) {
    if (n == 1u) out.op(0, a);
    else if (n == 2u) {
      uint32_t keep = 0xFFFFFFFFu;
      if ((a & 7u) == OP_PLUS && (b & 7u) == OP_COLON) { start += int64_t(a >> 3); keep = b; }
      else if ((a & 7u) == OP_COLON && (b & 7u) == OP_PLUS) keep = a;
      if (keep != 0xFFFFFFFFu) {
        out.op(0, keep);
        n = 1;
        has_del = false; slow = (keep >> 3) == 0u; stars = 0; indels = 0; others = 0; total = keep >> 3;
      } else {
        out.op(0, a); out.op(1, b);
      }
    }
  }
};

// The same result as parse_gaf_line_bytewise at about a tenth of its instructions (the kernels are bound by instruction issue
// under divergence, not by bytes: profiles/r3_pmc_tokenizer_*).  The line is read ONCE, front to back:
//   * the five columns in front of the path are stepped over a word at a time (whitespace masks, token starts counted by popcount);
//   * the path is parsed where it stands, which also finds where it ends; the name -> node lookups of up to eight nodes are issued
//     together (a lookup is a trip to HBM, and a line of six nodes was six trips long);
//   * the three integers, two columns that are not read, and the mapping quality follow, then the search for "cs:" and
//     "dv:f:<digit>" in the optional fields (byte-equality masks for 'c' and 'd' on words).
// What the reference decides in ITS order (mapping quality before the path, the path's own errors last) is decided in that order
// at the end: the path's verdict is only held back.
// Out: node(i, idx) for every path node, nodes_done(k) once, then op(i, word) for every op.
// `stop`: timing-only ablation (tools/tok_ablate.py): leave after stage 1 columns + path, 2 integers, 3 tag search, 4 cs, 5 dv; 0 = all
template <class Out>
PANTAS_HD void parse_gaf_line_words(TextWords& T, const uint32_t b, const uint32_t e, int thr, const DenseNames& names, Out& out, LineSummary& r, bool check_cr = true,
                                    uint32_t stop = 0) {
  r.status = TOK_ERROR; r.err = TE_NONE; r.n_nodes = 0; r.n_ops = 0; r.flags = 0; r.start = 0; r.tail = 0;
  if (e <= b) { r.err = TE_COLUMNS; return; }
  const uint32_t wl = (e - 1u) >> 2;     // last word of the line
  if (check_cr) {
    // [b, e) has no '\n': any '\r' before its last byte is a lone one (as its last byte it is the CR of a CRLF)
    for (uint32_t i = b >> 2; i <= wl; ++i)
      if (sw_eq(T.w[i], 0x0D0D0D0Du) & ~sw_outside(i, b, e - 1u)) { r.err = TE_LONE_CR; return; }
  }
  // ---- where column 5 starts: the sixth token start of the line
  uint32_t p5b = TK_NONE32;
  {
    uint32_t tok = 0;
    bool prev_ws = true;
    for (uint32_t i = b >> 2; i <= wl; ++i) {
      const uint32_t m = sw_ws(T.w[i]) | sw_outside(i, b, e);
      const uint32_t starts = ~m & 0x80808080u & ((m << 8) | (prev_ws ? 0x80u : 0u));      // a token begins: the byte in front is whitespace
      prev_ws = (m >> 31) != 0u;
      const uint32_t n = uint32_t(__builtin_popcount(starts));
      if (tok + n > 5u) {
        uint32_t s = starts;
        for (uint32_t k = 5u - tok; k > 0u; --k) s &= s - 1u;       // (a word holds at most two token starts)
        p5b = (i << 2) + (sw_ctz(s) >> 3);
        break;
      }
      tok += n;
    }
  }
  if (p5b == TK_NONE32) { r.err = TE_COLUMNS; return; }
  if (stop == 6u) { r.start = int32_t(p5b); return; }
  // ---- column 5, the path: split at '>' (or '<'), the piece in front of the first separator dropped, consecutive repeats dropped.
  // The name -> node lookups are gathers from a table of graph size (a gigabyte for a whole genome): the names of up to
  // PATH_BATCH nodes become table indices first, their lookups are issued together, and only then looked at.
  const uint32_t c5 = T.at(p5b);
  const bool rev = c5 != '>';
  const uint32_t sep = rev ? '<' : '>';
  uint32_t path_err = TE_NONE, n_nodes = 0;
  uint32_t q = p5b;
  {
    uint32_t ch = c5;
    while (ch != sep && !tk_ws(ch)) { ++q; if (q >= e) break; ch = T.at(q); }
    if (q >= e || ch != sep) path_err = TE_NO_NODE;      // (q stands at the end of the token)
  }
  if (path_err == TE_NONE) {
    // Names are read a word at a time: the first separator / whitespace byte of the word (byte-equality and whitespace masks)
    // says how many of its bytes are digits of the name in hand; those (up to four) become a number with two multiplications
    // (pairs of digits, then pairs of pairs) and join the name's value.  Node indices wait in a shift register of PATH_BATCH.
    constexpr uint32_t PATH_BATCH = 8;
    const uint32_t sep4 = sep * 0x01010101u;
    const uint32_t we = e >> 2, emask = 0x80808080u << (8u * (e & 3u));      // the word that holds byte e, its bytes from e on
    uint32_t R[PATH_BATCH];
PANTAS_UNROLL
    for (uint32_t j = 0; j < PATH_BATCH; ++j) R[j] = TK_NONE32;
    uint32_t cnt = 0, last = TK_NONE32;
    auto flush = [&]() {          // the lookups of the batch together, then in path order (R[cnt - 1] is the oldest)
      uint32_t idx[PATH_BATCH];
PANTAS_UNROLL
      for (uint32_t j = 0; j < PATH_BATCH; ++j) idx[j] = R[j] != TK_NONE32 ? tk_node_of(names, R[j]) : TK_NONE32;
PANTAS_UNROLL
      for (uint32_t jj = 0; jj < PATH_BATCH; ++jj) {
        const uint32_t j = PATH_BATCH - 1u - jj;
        if (j < cnt && path_err == TE_NONE) {
          if (idx[j] == TK_NONE32) path_err = TE_UNKNOWN_NODE;
          else if (n_nodes == 0 || idx[j] != last) { out.node(n_nodes++, idx[j]); last = idx[j]; }
        }
      }
PANTAS_UNROLL
      for (uint32_t j = 0; j < PATH_BATCH; ++j) R[j] = TK_NONE32;
      cnt = 0;
    };
    uint32_t t = q + 1u;            // q stands on a separator: the first byte of a name (or the byte behind the path)
    bool more = true;
    while (more) {
      // up to PATH_BATCH names.  The step below has NO branch but the loop's own: the lanes of a wave stand at different places
      // of different paths, and every `if` of a divergent loop is paid by all of them on every pass (the name in hand ending
      // is a select, not a branch).
      uint64_t val = 0;
      uint32_t len = 0, first = 0;
      bool ok = true;
      for (;;) {
        const uint32_t i = t >> 2, o = t & 3u;
        const bool inside = t < e;
        const uint32_t w = T.w[inside ? i : wl];                // (clamped: an unconditional load)
        // bytes of this word from the line's end on count as delimiters (t >= b always: nothing to mask in front)
        const uint32_t beyond = i < we ? 0u : i == we ? emask : 0x80808080u;
        uint32_t dm = inside ? (sw_eq(w, sep4) | sw_ws(w) | beyond) : 0x80808080u;   // t == e: the name ends here
        dm &= 0x80808080u << (8u * o);                          // delimiters at or behind byte o of the word
        const bool ends = dm != 0u;
        const uint32_t p = ends ? (sw_ctz(dm) >> 3) : 4u;       // the first of them (4: none in this word)
        const uint32_t n = p - o;                               // digits of the name in this word (0 .. 4)
        const uint32_t x = w >> (8u * o);
        first = len == 0u ? (x & 0xFFu) : first;
        const uint32_t chunk = sw_digits4(x, n, ok);            // (n == 0: "0000", no digit to object to)
        const uint32_t p10 = n == 4u ? 10000u : uint32_t(0x03E80064000A0001ull >> (16u * n)) & 0xFFFFu;   // 1, 10, 100, 1000
        val = val * p10 + chunk;                                // (a name of more than 18 digits overflows: it is refused by its length)
        len += n;
        t += n;
        // the name in hand ends at t (when `ends`): into the shift register, everything else by select
        const uint64_t off = val - names.base;                  // (val < base wraps to something huge)
        const uint32_t rel = (ok && len - 1u <= 17u && !(len > 1u && first == '0') && off < names.size) ? uint32_t(off) : TK_NONE32;
PANTAS_UNROLL
        for (uint32_t j = PATH_BATCH - 1u; j > 0u; --j) R[j] = ends ? R[j - 1u] : R[j];
        R[0] = ends ? rel : R[0];
        cnt += ends ? 1u : 0u;
        const bool goes_on = ends && t < e && ((w >> (8u * (p & 3u))) & 0xFFu) == sep && p < 4u;
        more = !ends || goes_on;
        if (ends && (!goes_on || cnt == PATH_BATCH)) { t += goes_on ? 1u : 0u; break; }
        t += ends ? 1u : 0u;
        val = ends ? 0u : val; len = ends ? 0u : len; ok = ends ? true : ok;
      }
      flush();
      if (path_err != TE_NONE) more = false;
    }
    q = t;
    if (path_err != TE_NONE) while (q < e && !tk_ws(T.at(q))) ++q;      // the rest of the token is only stepped over
  }
  const uint32_t p5e = q;
  out.nodes_done(n_nodes);
  if (stop == 7u) { r.start = int32_t(p5e + n_nodes); return; }
  // ---- columns 6 .. 11: a bitmap of the whitespace of the 64 bytes behind the path, one bit per byte; the six tokens are its
  // first six runs of zeros (the mandatory columns behind the path are a few short integers: they fit; when they do not, or the
  // line ends early, the bytes are walked one by one)
  uint32_t p6b = 0, p6e = 0, p7b = 0, p7e = 0, p8b = 0, p8e = 0, p11b = 0, p11e = 0;
  bool short_line = false, walked = true;
  {
    const uint32_t base = p5e & ~3u;                       // the window starts at the word that holds the end of the path
    uint64_t ws = 0;
PANTAS_UNROLL
    for (uint32_t j = 0; j < 16u; ++j) {
      const uint32_t i = (base >> 2) + j;
      const uint32_t m = i <= wl ? (sw_ws(T.w[i]) | sw_outside(i, p5e, e)) : 0x80808080u;     // in front of the path's end and behind the line: whitespace
      ws |= uint64_t(((m >> 7) * 0x00204081u >> 21) & 0xFu) << (4u * j);                      // bits 7, 15, 23, 31 -> four bits
    }
    uint64_t starts = ~ws & (ws << 1), ends = ws & (~ws << 1);     // (bit 0 is in front of the path's end: whitespace)
    if (uint32_t(__builtin_popcountll(ends)) >= 6u) {
      walked = false;
      auto take = [&](uint64_t& bits) { const uint32_t at = uint32_t(__builtin_ctzll(bits)); bits &= bits - 1ull; return base + at; };
      p6b = take(starts); p6e = take(ends);
      p7b = take(starts); p7e = take(ends);
      p8b = take(starts); p8e = take(ends);
      (void)take(starts); (void)take(ends);
      (void)take(starts); (void)take(ends);
      p11b = take(starts); p11e = take(ends);
    }
  }
  if (walked) {
    auto token = [&](uint32_t& tb, uint32_t& te) {
      while (q < e && tk_ws(T.at(q))) ++q;
      tb = q;
      if (q >= e) { short_line = true; te = q; return; }
      while (q < e && !tk_ws(T.at(q))) ++q;
      te = q;
    };
    token(p6b, p6e); token(p7b, p7e); token(p8b, p8e);
    { uint32_t x, y; token(x, y); token(x, y); }
    token(p11b, p11e);
  }
  if (short_line) { r.err = TE_COLUMNS; return; }
  const uint32_t opt = p11e;   // optional fields start here
  if (stop == 1u) { r.start = int32_t(p5b + p5e + p6b + p6e + p7b + p7e + p8b + p8e + p11b + p11e + opt + n_nodes); return; }
  // ---- the reference's order from here: mapping quality, '*' path, the three integers
  int64_t mapq;
  if (!tk_int_fast(T, p11b, p11e, mapq)) { r.err = TE_MAPQ; return; }
  if (mapq < int64_t(thr)) { r.status = TOK_REJECT; return; }
  if (p5e - p5b == 1 && c5 == '*') { r.status = TOK_SKIP; return; }
  int64_t path_len, start, endpos;
  if (!tk_int_fast(T, p6b, p6e, path_len) || !tk_int_fast(T, p7b, p7e, start) || !tk_int_fast(T, p8b, p8e, endpos)) { r.err = TE_POSITIONS; return; }
  if (stop == 2u) { r.start = int32_t(path_len + start + endpos); return; }

  // ---- the optional fields: where the first "cs:" and the first "dv:f:<digit>" stand ('s', ':', 'v', 'f' and digits are no
  // whitespace: a match never spans two fields)
  uint32_t cs_at = TK_NONE32, dv_at = TK_NONE32;
  if (opt < e) {
    for (uint32_t i = opt >> 2; i <= wl && (cs_at == TK_NONE32 || dv_at == TK_NONE32); ++i) {
      const uint32_t w = T.w[i], in = ~sw_outside(i, opt, e);
      uint32_t cm = cs_at == TK_NONE32 ? (sw_eq(w, 0x63636363u) & in) : 0u;
      uint32_t dm = dv_at == TK_NONE32 ? (sw_eq(w, 0x64646464u) & in) : 0u;
      while (cm) {
        const uint32_t pos = (i << 2) + (sw_ctz(cm) >> 3);
        cm &= cm - 1u;
        if (pos + 3u <= e && T.at(pos + 1u) == 's' && T.at(pos + 2u) == ':') { cs_at = pos; break; }
      }
      while (dm) {
        const uint32_t pos = (i << 2) + (sw_ctz(dm) >> 3);
        dm &= dm - 1u;
        if (pos + 6u <= e && T.at(pos + 1u) == 'v' && T.at(pos + 2u) == ':' && T.at(pos + 3u) == 'f' && T.at(pos + 4u) == ':' && tk_digit(T.at(pos + 5u))) { dv_at = pos; break; }
      }
    }
  }
  if (stop == 3u) { r.start = int32_t(cs_at + dv_at); return; }
  // ---- cs: from its "cs:" to the end of that field, every "cs:Z:" stepped over; the bytes in front of the first op character
  // are dropped, then every op character takes the run behind it (:10-37)
  OpList<Out> em(out);
  if (cs_at != TK_NONE32) {
    uint32_t fe = cs_at + 3u;
    for (uint32_t i = fe >> 2; i <= wl; ++i) {             // the field's end: the first whitespace behind "cs:"
      const uint32_t m = (sw_ws(T.w[i]) | sw_outside(i, fe, e)) & (i == (fe >> 2) ? 0x80808080u << (8u * (fe & 3u)) : 0x80808080u);
      if (m) { fe = (i << 2) + (sw_ctz(m) >> 3); break; }
      if (i == wl) fe = e;
    }
    uint32_t err = TE_NONE;
    // What a mapper writes: "cs:Z:" once, at the start.  Then the op characters are found a word at a time (a bitmap of them, one
    // bit per byte) and every op is its character plus the distance to the next one: a run of one to nine digits is a number
    // (two multiplications per four digits), any other run counts its bytes.
    const uint32_t s0 = cs_at + 5u;
    bool plain = fe >= s0 && fe - s0 <= 60u && sw_bytes4(T, cs_at + 1u) == 0x3A5A3A73u /* "s:Z:" */;
    uint64_t opbits = 0;
    const uint32_t base = s0 & ~3u;
    if (plain) {
      for (uint32_t i = base >> 2, j = 0; i <= wl && (i << 2) < fe; ++i, ++j) {
        const uint32_t w = T.w[i], in = ~sw_outside(i, s0, fe);
        // '*' and '+' differ in their lowest bit only: one comparison for both
        const uint32_t m = (sw_eq(w & 0xFEFEFEFEu, 0x2A2A2A2Au) | sw_eq(w, 0x2D2D2D2Du) | sw_eq(w, 0x3A3A3A3Au) | sw_eq(w, 0x3D3D3D3Du) | sw_eq(w, 0x7E7E7E7Eu)) & in;
        if (sw_eq(w, 0x73737373u) & in) plain = false;          // an 's': there may be another "cs:Z:" to step over
        opbits |= uint64_t(((m >> 7) * 0x00204081u >> 21) & 0xFu) << (4u * j);
      }
    }
    if (plain) {
      while (opbits && err == TE_NONE) {
        const uint32_t pb = uint32_t(__builtin_ctzll(opbits));
        opbits &= opbits - 1ull;
        const uint32_t at = base + pb, nxt = opbits ? base + uint32_t(__builtin_ctzll(opbits)) : fe;
        const uint32_t op = T.at(at), run = nxt - at - 1u;
        uint32_t v = run;
        int64_t num;
        if (op == '*') v = 1;
        else if (run <= 9u) { if (sw_uint9(T, at + 1u, nxt, num)) v = uint32_t(num); }     // (at most 999 999 999: a valid length or TE_OP_LEN below)
        else {                                                  // ten bytes or more: one by one (leading zeros keep a long number small)
          bool digits = true, over = false;
          uint32_t val = 0;
          for (uint32_t c = at + 1u; c < nxt && digits; ++c) {
            const uint32_t d = T.at(c) - '0';
            digits = d <= 9u;
            if (digits && !over) { val = val * 10u + d; over = val > OP_MAX_LEN; }
          }
          if (digits) { v = val; if (over) err = TE_OP_LEN; }
        }
        if (v > OP_MAX_LEN) err = TE_OP_LEN;
        if (err == TE_NONE) em.push((v << 3) | tk_opcode(op));
      }
    } else {
      uint32_t cur = 0, run = 0, val = 0;
      bool digits = true, over = false;
      auto close = [&]() {                                   // the op in hand with the run behind it
        uint32_t v;
        if (cur == '*') v = 1;
        else if (run > 0 && digits) { if (over) err = TE_OP_LEN; v = val; }
        else { v = run; if (v > OP_MAX_LEN) err = TE_OP_LEN; }
        if (err == TE_NONE) em.push((v << 3) | tk_opcode(cur));
      };
      for (uint32_t sp = cs_at; sp < fe && err == TE_NONE;) {
        const uint32_t ch = T.at(sp);
        if (ch == 'c' && fe - sp >= 5u && T.at(sp + 1u) == 's' && T.at(sp + 2u) == ':' && T.at(sp + 3u) == 'Z' && T.at(sp + 4u) == ':') { sp += 5u; continue; }
        ++sp;
        if (tk_opchar(ch)) {
          if (cur) close();
          cur = ch; run = 0; val = 0; digits = true; over = false;
        } else if (cur) {
          ++run;
          const uint32_t d = ch - '0';
          if (d > 9u) digits = false;
          else if (digits && !over) { val = val * 10u + d; over = val > OP_MAX_LEN; }
        }
      }
      if (cur && err == TE_NONE) close();
    }
    if (err != TE_NONE) { r.err = err; return; }
  } else {
    em.push((1u << 3) | OP_STAR);
  }
  em.finish(start);
  if (stop == 4u) { r.start = int32_t(start + em.n); return; }

  // ---- dv
  if (dv_at == TK_NONE32) { r.err = TE_NO_DV; return; }
  if (tk_dv_above_tenth(T, dv_at + 5u, e)) { r.status = TOK_SKIP; return; }      // (digits and '.' end at the field's end: whitespace)
  if (stop == 5u) { r.start = int32_t(start); return; }

  // ---- the path's own verdict
  if (path_err != TE_NONE) { r.err = path_err; return; }
  const int64_t tail = path_len - endpos;
  if (start < -2147483648ll || start > 2147483647ll || tail < -2147483648ll || tail > 2147483647ll) { r.err = TE_OFFSETS; return; }
  uint32_t flags = rev ? LINE_REV : 0u;
  if (em.slow) flags |= LINE_SLOW;
  flags |= line_class_flags(em.n, em.stars, em.indels, em.others, em.slow, em.total, start, tail);
  if (em.has_del) flags |= LINE_HAS_DEL;
  r.status = TOK_KEEP;
  r.n_nodes = n_nodes;
  r.n_ops = em.n;
  r.flags = flags;
  r.start = int32_t(start);
  r.tail = int32_t(tail);
}

// ---- the PLAIN line, fast (round 5) ----------------------------------------------------------------------------------------
// parse_gaf_line_words above takes every line the reference takes -- any Python whitespace between columns, signed and
// underscored integers, "cs:Z:" more than once, '=' and '~' ops, names of any width -- and pays for it on every line: 107 vector
// instructions per line and lane at whole-genome scale, a kernel at half the chip's vector-issue peak (profiles/r5_tok_*).
// A mapper writes ONE form: columns separated by single tabs and no other control character or space anywhere, a path of
// '>' (or '<') and decimal names of at most twelve digits without leading zeros, plain unsigned integers of at most nine digits,
// a cs of "cs:Z:" once with ':' '*' '+' '-' ops of at most 60 bytes, "dv:f:<digits>".  parse_gaf_line_fast takes that form and
// nothing else: it returns false -- having written nothing that matters: the caller hands the line to the general parser -- the
// moment a line is anything else (including every line the reference would raise on).  What it returns true on, it decides and
// emits exactly as parse_gaf_line_words does; tests/hostsim/hostsim_tokenize.cpp holds the two to the same summaries, nodes and
// ops on every line the fast form accepts (goldens, crash inputs, fuzzed files).
//
// Structure: no byte-wise scans.  Tabs, control characters, digits and op characters are found a word at a time (SWAR masks);
// a name's end is the first non-digit of the four words that hold it, its value three groups of up to four digits read from those
// same words; the node lookups of up to eight names are issued together.

// 0x80 in every byte of w below 0x21 (control characters and the space); bytes from 0x80 up are not
PANTAS_HD uint32_t sw_ctrl(uint32_t w) { return ~(((w & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | w) & 0x80808080u; }
// 0x80 in every byte of w that is no ASCII digit
PANTAS_HD uint32_t sw_nondigit(uint32_t w) {
  const uint32_t d = w ^ 0x30303030u;
  return (((d & 0x7F7F7F7Fu) + 0x76767676u) | d) & 0x80808080u;
}
// 0x80 in every byte of w outside '`' .. 'z' (0x60 .. 0x7A)
PANTAS_HD uint32_t sw_nonlower(uint32_t w) {
  const uint32_t l = w ^ 0x60606060u;
  return (((l & 0x7F7F7F7Fu) + 0x65656565u) | l) & 0x80808080u;
}
// bits 7, 15, 23, 31 of m as bits 0 .. 3
PANTAS_HD uint32_t sw_nibble(uint32_t m) { return ((m >> 7) * 0x00204081u >> 21) & 0xFu; }
// the n (0 .. 4) ASCII digits in the low bytes of x (first digit in byte 0) as a number; the bytes ARE digits
PANTAS_HD uint32_t sw_digits4_known(uint32_t x, uint32_t n) {
  const uint32_t y = uint32_t(((uint64_t(x) << 32) | 0x30303030ull) >> (8u * n));
  uint32_t d = y & 0x0F0F0F0Fu;
  d = d * 10u + (d >> 8);
  return (d & 0xFFu) * 100u + ((d >> 16) & 0xFFu);
}
PANTAS_HD uint32_t sw_pow10(uint32_t n) { return n == 4u ? 10000u : uint32_t(0x03E80064000A0001ull >> (16u * n)) & 0xFFFFu; }   // n = 0 .. 4

template <class Out>
PANTAS_HD bool parse_gaf_line_fast(TextWords& T, const uint32_t b, const uint32_t e, int thr, const DenseNames& names, Out& out, LineSummary& r, bool check_cr,
                                   uint32_t stop = 0) {       // stop: timing-only ablation (tools/tok_ablate.py letters a .. f): leave after that stage
  r.status = TOK_ERROR; r.err = TE_NONE; r.n_nodes = 0; r.n_ops = 0; r.flags = 0; r.start = 0; r.tail = 0;
  // check_cr: the text holds a '\r' that no '\n' follows somewhere (the line index looked): such a line is refused whatever else
  // it holds, before the mapping quality is looked at -- every line of such a text goes to the parser that looks for it
  if (e <= b || check_cr) return false;
  const uint32_t we = e >> 2;                                   // the word that holds byte e (the '\n', or padding behind the text)
  const uint32_t ebits = 0x80808080u << (8u * (e & 3u));        // its bytes from e on
  auto word_at = [&](uint32_t i) { return T.w[i < we ? i : we]; };                       // (never read behind the line's own end)
  auto beyond = [&](uint32_t i) { return i < we ? 0u : i == we ? ebits : 0x80808080u; };  // bytes of word i at or behind e

  // ---- columns 0 .. 4: five tabs, and no other control character or space in front of them
  uint32_t p5b = TK_NONE32;
  {
    uint32_t seen = 0;
    uint32_t front = ~(0xFFFFFFFFu << (8u * (b & 3u)));        // bytes of the first word in front of b
    uint32_t carry = 0x80u << (8u * (b & 3u));                 // "a tab stands in front of this byte": the line's start counts as one
    for (uint32_t i = b >> 2; i <= we; ++i) {
      const uint32_t w = T.w[i];
      const uint32_t keep = 0x80808080u & ~front & ~beyond(i);
      front = 0u;
      const uint32_t tab = sw_eq(w, 0x09090909u) & keep, ctl = sw_ctrl(w) & keep;
      if (tab != ctl) return false;
      if (tab & ((tab << 8) | carry)) return false;            // an empty column: str.split() would close the gap
      carry = tab >> 31 ? 0x80u : 0u;
      const uint32_t n = uint32_t(__builtin_popcount(tab));
      if (seen + n >= 5u) {
        uint32_t s = tab;
        for (uint32_t k = 4u - seen; k > 0u; --k) s &= s - 1u;
        p5b = (i << 2) + (sw_ctz(s) >> 3) + 1u;
        break;
      }
      seen += n;
    }
  }
  if (p5b == TK_NONE32 || p5b >= e) return false;
  if (stop == 1u) { r.start = int32_t(p5b); return true; }

  // ---- column 5, the path
  const uint32_t c5 = T.at(p5b);
  uint32_t p5e, n_nodes = 0;
  bool star_path = false;
  if (c5 == '*') {
    p5e = p5b + 1u;
    star_path = true;
  } else {
    if (c5 != '>' && c5 != '<') return false;
    uint32_t t = p5b + 1u;
    // name -> node lookups are gathers from a table of graph size (a gigabyte for a whole genome: every one a trip to memory of
    // microseconds under load): the table indices of up to FAST_BATCH names wait in registers and are asked for TOGETHER -- looked
    // up one by one, even a name ahead, a line of six nodes was six trips long and the kernel waited 85 % of its time
    constexpr uint32_t FAST_BATCH = 8;
    uint32_t R[FAST_BATCH];
PANTAS_UNROLL
    for (uint32_t j = 0; j < FAST_BATCH; ++j) R[j] = 0u;
    uint32_t cnt = 0, last = TK_NONE32;
    bool unknown = false;
    auto flush = [&]() {
      uint32_t idx[FAST_BATCH];
PANTAS_UNROLL
      for (uint32_t j = 0; j < FAST_BATCH; ++j) idx[j] = tk_node_of(names, R[j]);      // (slots beyond cnt hold an index that was valid: loads without a condition)
PANTAS_UNROLL
      for (uint32_t j = 0; j < FAST_BATCH; ++j) {
        if (j < cnt) {
          if (idx[j] == TK_NONE32) unknown = true;
          else if (!unknown && (n_nodes == 0u || idx[j] != last)) { out.node(n_nodes++, idx[j]); last = idx[j]; }
        }
      }
      cnt = 0;
    };
    for (;;) {
      // the four words that hold a name of up to twelve digits from byte t on
      const uint32_t i = t >> 2, o8 = (t & 3u) << 3;
      const uint32_t w0 = word_at(i), w1 = word_at(i + 1u), w2 = word_at(i + 2u), w3 = word_at(i + 3u);
      // its end: the first byte at or behind t that is no digit (a byte at or behind e never is)
      uint32_t nd = sw_nibble(sw_nondigit(w0) | beyond(i)) | (sw_nibble(sw_nondigit(w1) | beyond(i + 1u)) << 4) |
                    (sw_nibble(sw_nondigit(w2) | beyond(i + 2u)) << 8) | (sw_nibble(sw_nondigit(w3) | beyond(i + 3u)) << 12);
      nd &= 0xFFFFu << (t & 3u);
      if (nd == 0u) return false;                              // thirteen digits or more
      const uint32_t q = (i << 2) + sw_ctz(nd), n = q - t;
      if (n == 0u || n > 12u || q >= e) return false;          // (an empty name; a line that ends inside its path)
      const uint32_t g0 = uint32_t((uint64_t(w0) | (uint64_t(w1) << 32)) >> o8);
      const uint32_t g1 = uint32_t((uint64_t(w1) | (uint64_t(w2) << 32)) >> o8);
      const uint32_t g2 = uint32_t((uint64_t(w2) | (uint64_t(w3) << 32)) >> o8);
      if (n > 1u && (g0 & 0xFFu) == '0') return false;         // a leading zero names no segment
      const uint32_t k0 = n < 4u ? n : 4u, k1 = n > 4u ? (n - 4u < 4u ? n - 4u : 4u) : 0u, k2 = n > 8u ? n - 8u : 0u;
      const uint64_t val = (uint64_t(sw_digits4_known(g0, k0)) * sw_pow10(k1) + sw_digits4_known(g1, k1)) * sw_pow10(k2) + sw_digits4_known(g2, k2);
      const uint64_t off = val - names.base;
      if (off >= names.size) return false;
PANTAS_UNROLL
      for (uint32_t j = 0; j < FAST_BATCH; ++j) R[j] = j == cnt ? uint32_t(off) : R[j];
      ++cnt;
      const uint32_t delim = T.at(q);
      if (delim == c5) {
        t = q + 1u;
        if (cnt == FAST_BATCH) { flush(); if (unknown) return false; }
        continue;
      }
      if (delim != '\t') return false;                         // the other orientation character, a letter, ...
      p5e = q;
      break;
    }
    flush();
    if (unknown) return false;
  }
  if (p5e >= e || T.at(p5e) != '\t') return false;
  out.nodes_done(n_nodes);
  if (stop == 2u) { r.start = int32_t(p5e + n_nodes); return true; }

  // ---- columns 6 .. 11: the seven tabs from the one behind the path on (the last may be the line's end), nothing else below 0x21
  uint32_t tp[7];
  {
    uint32_t found = 0;
    uint32_t front = ~(0xFFFFFFFFu << (8u * (p5e & 3u)));
    for (uint32_t i = p5e >> 2, j = 0; j < 14u && found < 7u; ++i, ++j) {
      const uint32_t w = word_at(i);
      const uint32_t keep = 0x80808080u & ~front & ~beyond(i);
      front = 0u;
      uint32_t tab = sw_eq(w, 0x09090909u) & keep;
      if (tab != (sw_ctrl(w) & keep)) return false;
      if (i >= we) tab |= i == we ? (0x80u << (8u * (e & 3u))) : 0u;       // the line's end closes column 11 like a tab
      while (tab && found < 7u) {
        const uint32_t at = (i << 2) + (sw_ctz(tab) >> 3);
        tab &= tab - 1u;
PANTAS_UNROLL
        for (uint32_t k = 0; k < 7u; ++k) tp[k] = k == found ? at : tp[k];
        ++found;
      }
      if (i >= we) break;
    }
    if (found < 7u) return false;                              // fewer than twelve columns, or columns wider than the window
    bool gap = false;                                          // an empty column: str.split() would close the gap and shift the rest
PANTAS_UNROLL
    for (uint32_t k = 0; k < 6u; ++k) gap |= tp[k + 1u] == tp[k] + 1u;
    if (gap) return false;
  }
  const uint32_t opt = tp[6];                                   // end of column 11: a tab, or e
  if (stop == 3u) { r.start = int32_t(opt + tp[0] + tp[1] + tp[2] + tp[3] + tp[5]); return true; }
  int64_t mapq, path_len, start, endpos;
  if (!sw_uint9(T, tp[5] + 1u, tp[6], mapq)) return false;
  if (mapq < int64_t(thr)) { r.status = TOK_REJECT; return true; }
  if (star_path) { r.status = TOK_SKIP; return true; }
  if (!sw_uint9(T, tp[0] + 1u, tp[1], path_len) || !sw_uint9(T, tp[1] + 1u, tp[2], start) || !sw_uint9(T, tp[2] + 1u, tp[3], endpos)) return false;

  // ---- the optional fields: tabs and nothing else below 0x21; the first "cs:" and the first "dv:f:<digit>"
  uint32_t cs_at = TK_NONE32, dv_at = TK_NONE32;
  if (opt < e) {
    uint32_t front = ~(0xFFFFFFFFu << (8u * (opt & 3u)));
    for (uint32_t i = opt >> 2; i <= we; ++i) {
      const uint32_t w = T.w[i];
      const uint32_t keep = 0x80808080u & ~front & ~beyond(i);
      front = 0u;
      if ((sw_eq(w, 0x09090909u) & keep) != (sw_ctrl(w) & keep)) return false;
      uint32_t cm = cs_at == TK_NONE32 ? (sw_eq(w, 0x63636363u) & keep) : 0u;
      uint32_t dm = dv_at == TK_NONE32 ? (sw_eq(w, 0x64646464u) & keep) : 0u;
      while (cm) {
        const uint32_t pos = (i << 2) + (sw_ctz(cm) >> 3);
        cm &= cm - 1u;
        if (pos + 3u <= e && T.at(pos + 1u) == 's' && T.at(pos + 2u) == ':') { cs_at = pos; break; }
      }
      while (dm) {
        const uint32_t pos = (i << 2) + (sw_ctz(dm) >> 3);
        dm &= dm - 1u;
        if (pos + 6u <= e && T.at(pos + 1u) == 'v' && T.at(pos + 2u) == ':' && T.at(pos + 3u) == 'f' && T.at(pos + 4u) == ':' && tk_digit(T.at(pos + 5u))) { dv_at = pos; break; }
      }
    }
  }
  if (dv_at == TK_NONE32) return false;                         // (the reference raises: the general parser words it)
  if (stop == 4u) { r.start = int32_t(cs_at + dv_at + path_len + start + endpos); return true; }

  // ---- cs: "cs:Z:" once, then ':' '*' '+' '-' ops over at most 60 bytes
  OpList<Out> em(out);
  if (cs_at != TK_NONE32) {
    if (cs_at + 5u > e || sw_bytes4(T, cs_at + 1u) != 0x3A5A3A73u /* "s:Z:" */) return false;
    const uint32_t s0 = cs_at + 5u, base = s0 & ~3u;
    // op candidates (neither digit nor '`' .. 'z'), digits, and the field's end (the first tab, or e), one bit per byte
    uint64_t opbits = 0, digbits = 0;
    uint32_t fe = TK_NONE32;
    for (uint32_t i = base >> 2, j = 0; j < 16u; ++i, ++j) {
      const uint32_t w = word_at(i);
      const uint32_t keep = 0x80808080u & ~(i == (base >> 2) ? ~(0xFFFFFFFFu << (8u * (s0 & 3u))) : 0u);
      const uint32_t endm = (sw_eq(w, 0x09090909u) | beyond(i)) & keep;
      const uint32_t nd = sw_nondigit(w);
      if (sw_eq(w, 0x73737373u) & keep & ~(endm ? (0xFFFFFFFFu << sw_ctz(endm)) : 0u)) return false;      // an 's': there may be another "cs:Z:" to step over
      opbits |= uint64_t(sw_nibble(nd & sw_nonlower(w) & keep)) << (4u * j);
      digbits |= uint64_t(sw_nibble(~nd & keep)) << (4u * j);
      if (endm) { fe = (i << 2) + (sw_ctz(endm) >> 3); break; }
    }
    if (fe == TK_NONE32) return false;                          // a cs of more than 60 bytes
    const uint32_t len = fe - base;                             // bits at and behind len are not the field's
    const uint64_t inside = len >= 64u ? ~0ull : ((1ull << len) - 1ull);
    opbits &= inside;
    digbits &= inside;
    // nothing behind "cs:Z:", or bytes in front of the first op character (the reference drops them): the general parser's business
    if (opbits == 0ull || uint32_t(__builtin_ctzll(opbits)) != s0 - base) return false;
    while (opbits) {
      const uint32_t pb = uint32_t(__builtin_ctzll(opbits));
      opbits &= opbits - 1ull;
      const uint32_t nxt = opbits ? uint32_t(__builtin_ctzll(opbits)) : len;
      const uint32_t at = base + pb, run = nxt - pb - 1u;
      const uint32_t op = T.at(at);
      if (op != ':' && op != '*' && op != '+' && op != '-') return false;      // '=' and '~' (and any other byte that is no run content)
      uint32_t v = run;
      if (op == '*') v = 1u;
      else if (run > 0u) {
        const uint64_t runmask = ((1ull << run) - 1ull) << (pb + 1u);          // (run <= 59)
        if ((digbits & runmask) == runmask) {                                   // all digits: the number
          if (run > 9u) return false;
          int64_t num;
          if (!sw_uint9(T, at + 1u, base + nxt, num)) return false;
          v = uint32_t(num);
        }
      }
      if (v > OP_MAX_LEN) return false;
      em.push((v << 3) | tk_opcode(op));
    }
  } else {
    em.push((1u << 3) | OP_STAR);
  }
  em.finish(start);
  if (stop == 5u) { r.start = int32_t(start + em.n); return true; }

  // ---- dv, then what the general parser decides last
  if (tk_dv_above_tenth(T, dv_at + 5u, e)) { r.status = TOK_SKIP; return true; }
  const int64_t tail = path_len - endpos;
  if (start < -2147483648ll || start > 2147483647ll || tail < -2147483648ll || tail > 2147483647ll) return false;
  uint32_t flags = c5 != '>' ? LINE_REV : 0u;
  if (em.slow) flags |= LINE_SLOW;
  flags |= line_class_flags(em.n, em.stars, em.indels, em.others, em.slow, em.total, start, tail);
  if (em.has_del) flags |= LINE_HAS_DEL;
  r.status = TOK_KEEP;
  r.n_nodes = n_nodes;
  r.n_ops = em.n;
  r.flags = flags;
  r.start = int32_t(start);
  r.tail = int32_t(tail);
  return true;
}

// the same for a line given by pointers: [b, e) plus up to three bytes on either side must be readable
template <class Out>
PANTAS_HD void parse_gaf_line(const uint8_t* b, const uint8_t* e, int thr, const DenseNames& names, Out& out, LineSummary& r, bool check_cr = true) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(b) & ~uintptr_t(3);
  TextWords T(reinterpret_cast<const uint32_t*>(a));
  const uint32_t off = uint32_t(reinterpret_cast<uintptr_t>(b) - a);
  parse_gaf_line_words(T, off, off + uint32_t(e - b), thr, names, out, r, check_cr);
}

}  // namespace pantas
