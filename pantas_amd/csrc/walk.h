// Per-GAF-line walk: lays the line's cs ops over its path nodes and reports what the
// line contributes to node coverage (NC), link coverage (RC) and the in/out-link position
// histograms (IL/OL).  This is the body of the augment kernels (augment.hip instantiates it
// with a sink that does device atomics); it is a header so that tests/hostsim can compile
// the very same code for the CPU under ASan/UBSan.  It is NOT a CPU path of the product.
//
// Behaviour restated from scripts/alignments_augmentation_from_gaf.py:
//   op distribution over nodes            :209-255
//   lone '+'/'-' node dropped, compaction :63-107
//   NC                                    :263-269
//   IL/OL, forward and reverse strand     :276-353
//   RC on consecutive kept nodes          :357-363
//
// Instead of materialising the per-node op lists, the walk keeps a constant-size summary of
// the compacted list of the node in hand (first entry, last entry, how many entries are
// neither '-' nor '*') -- everything :276-353 reads -- and holds each kept node back until
// the next kept node (or the end of the line) is known, because the IL/OL rules ask "is this
// the last kept node?".
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define PANTAS_HD __host__ __device__ __forceinline__
#define PANTAS_UNROLL _Pragma("unroll")
#else
#define PANTAS_HD inline
#define PANTAS_UNROLL
#endif

namespace pantas {

// cs op codes of the tokenised record: word = (length << 3) | code
enum : uint32_t { OP_COLON = 0, OP_STAR = 1, OP_PLUS = 2, OP_MINUS = 3, OP_EQUAL = 4, OP_TILDE = 5 };
constexpr uint32_t OP_MAX_LEN = (1u << 29) - 1;

// node_len word: low 30 bits = segment length; the two top bits mark nodes whose IL / OL
// histogram holds a key other than the common one (set by the first augment kernel)
constexpr uint32_t LEN_MASK = 0x3FFFFFFFu;
constexpr uint32_t FLAG_IL = 1u << 30;
constexpr uint32_t FLAG_OL = 1u << 31;

constexpr uint32_t LINE_REV = 1u;       // path written with '<'
constexpr uint32_t LINE_HAS_DEL = 2u;   // cs string holds a '-' op
constexpr uint32_t LINE_SLOW = 4u;      // cs holds a zero-length or '~' op: walked by the scalar code only
constexpr uint32_t LINE_SIMPLE = 8u;    // cs holds only ':' and at most 4 '*' ops (no zero lengths): eligible for the short walk
constexpr uint32_t LINE_STARS = 32u;      // a short-walk line with at least one '*' op (set beside LINE_SIMPLE / LINE_ONE_INDEL)
constexpr uint32_t LINE_ONE_INDEL = 16u;  // cs holds ':' ops, at most 4 '*' ops and exactly ONE '+' or '-' op (no zero lengths): short walk too
constexpr uint32_t SIMPLE_MAX_STARS = 4;
// Records of a batch come in two classes, each bucketed by locus: first the ones the short-walk kernel takes
// (LINE_SIMPLE and at most SHORT_MAX_PATH path nodes: its LDS entries hold 16-bit counts, 2048 records x 31 visits
// < 65536), then the rest (general walk).
constexpr uint32_t SHORT_MAX_PATH = 31;
constexpr uint32_t SHORT_MAX_OPS = 1u << 24;    // the short-walk kernel reads a record's op count from 24 bits (':' runs may repeat)
// The ops of a short-walk line add up to fewer than 2^16 bases (a line of at most 31 path nodes is a short read): positions
// in op space then fit 16 bits, and the walk keeps the four '*' positions and the indel in three registers instead of
// seven -- registers are what bounds the kernel (64 VGPRs = 8 waves per SIMD, and 7 waves cost 17 % of its speed).
constexpr uint64_t SHORT_MAX_TOTAL = 1ull << 16;
// A one-indel line can add 2 more IL / OL steps than it has path nodes (the node holding ':' '+' ':' counts three
// entries), so it may have two path nodes fewer: 2048 x (29 + 2) < 65536 as well.
constexpr uint32_t SHORT_MAX_PATH_INDEL = SHORT_MAX_PATH - 2;
PANTAS_HD bool short_walk_class(uint32_t flags, uint32_t path_nodes) {
  return ((flags & LINE_SIMPLE) != 0 && path_nodes <= SHORT_MAX_PATH) || ((flags & LINE_ONE_INDEL) != 0 && path_nodes <= SHORT_MAX_PATH_INDEL);
}
// order of the records inside a block of 2048 of a class (stable): by path length, so that the lanes of a wave loop
// alike, with the one-indel lines behind the others -- a wave pays for the indel code whenever ONE of its lanes needs it
// ... and, in front of those, the lines with a '*' behind the plain ones: a wave whose lines are all ':' runs skips the per-node
// look at the star positions
PANTAS_HD uint32_t block_order_key(uint32_t flags, uint32_t path_nodes) {
  return ((flags & LINE_ONE_INDEL) ? 0x20000u : 0u) | ((flags & LINE_STARS) ? 0x10000u : 0u) | (path_nodes < 0xFFFFu ? path_nodes : 0xFFFFu);
}
// what the tokenisers set from a line's op statistics (both tokenisers call this: one definition of the classes)
// start / tail: the line's offsets on its first / last path node.  The short walk computes a node's room in 32 bits
// (length < 2^30 minus the offsets): lines with offsets beyond +-2^29 -- none a mapper writes -- take the general walk.
PANTAS_HD uint32_t line_class_flags(uint32_t n_ops, uint32_t stars, uint32_t indels, uint32_t others, bool slow, uint64_t total_len,
                                    int64_t start, int64_t tail) {
  constexpr int64_t OFF_MAX = int64_t(1) << 29;
  if (start < -OFF_MAX || start > OFF_MAX || tail < -OFF_MAX || tail > OFF_MAX) return 0u;
  if (n_ops == 0 || n_ops >= SHORT_MAX_OPS || others != 0 || slow || stars > SIMPLE_MAX_STARS || total_len >= SHORT_MAX_TOTAL) return 0u;
  return (indels == 0 ? LINE_SIMPLE : indels == 1 ? LINE_ONE_INDEL : 0u) | (indels <= 1 && stars ? LINE_STARS : 0u);
}

// the GPU tokeniser's summary of the dense name table (gaf_device.h DenseNames::blocks, built in augment.hip device_name_table)
constexpr uint32_t NAME_BLOCK = 1024;                 // names per entry
constexpr uint32_t NAME_BLOCK_MIXED = 0xFFFFFFF0u;     // the block does not count up one by one: ask the table

struct LineHdr {
  int32_t start;     // start position on the first node (after the 2-op clipping rule)
  int32_t tail;      // path_len - end position
  uint32_t line_id;  // 0-based index of the line in the GAF file (first-seen ordering)
  uint32_t flags;
};

// WALK_TILDE_FIRST: a '~' op is the first op the walk of a line reaches.  The reference never assigns a length for '~'
// (:228-233) and reuses whatever the variable held: the length of the op in hand one iteration earlier.  Inside a line
// that is reproduced (walk_line below); for the first op of a line the value leaks in from the PREVIOUS GAF line (or is
// unbound when no line before it took an op in hand: UnboundLocalError), which a line-parallel walk does not have.  Such
// lines get the value from a table the host computes for them (TildeCarry: a sequential look at the lines in front of each,
// done only for a file that has such lines -- no mapper writes a cs that starts inside an intron); without an entry the
// walk stops with this code.
enum : int { WALK_OK = 0, WALK_OPS_EXHAUSTED = 1, WALK_TILDE_FIRST = 2 };

constexpr const char* TILDE_FIRST_MARK = "'~' is the first cs op the walk of this line reaches";      // in the text of the refusal
// {line id, length that leaks into the line} sorted by line id; TILDE_UNBOUND: no line in front took an op in hand
constexpr uint32_t TILDE_UNBOUND = 0xFFFFFFFFu;
struct TildeEntry { uint32_t line_id, carry; };
struct TildeCarry {
  const TildeEntry* tab = nullptr;
  uint32_t n = 0;
};
PANTAS_HD bool tilde_lookup(const TildeCarry* t, uint32_t line_id, uint32_t& carry) {
  if (!t || t->n == 0) return false;
  uint32_t lo = 0, hi = t->n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t->tab[mid].line_id < line_id) lo = mid + 1; else hi = mid;
  }
  if (lo >= t->n || t->tab[lo].line_id != line_id || t->tab[lo].carry == TILDE_UNBOUND) return false;
  carry = t->tab[lo].carry;
  return true;
}

enum : uint32_t { SIDE_IL = 0, SIDE_OL = 1 };

// first-seen order of an event: (line, index of the node among the line's kept nodes, phase)
// phase 0 = first compacted entry, 1 = any entry that is neither '-' nor '*', 2 = last entry
PANTAS_HD uint64_t event_order(uint32_t line_id, uint32_t kept_idx, uint32_t phase) {
  return (uint64_t(line_id) << 32) | (uint64_t(kept_idx) << 2) | phase;
}

// ---- the short walk's view of a cs string of its class (LINE_SIMPLE / LINE_ONE_INDEL) -------------------------------
// Op space = the concatenation of all ops; a path node with room takes the interval [P, P + le) of it.  What :63-107 and
// :276-353 make of the pieces that fall into the interval depends only on (a) whether pieces other than the indel exist
// (a node holding the lone '+' / '-' piece is dropped), (b) how many compacted entries are neither '-' nor '*': a ':' run
// before the indel, the indel itself when it is a '+', a ':' run after it.  '*' pieces never make an entry of their own:
// a leading one is swallowed by the next op, a later one merges into the entry before it.
struct ShortOps {
  uint32_t T = 0;                       // total op length (< 2^16)
  uint32_t ns = 0;                      // '*' ops, each one base long; bit 8: the indel is a '+'
  uint32_t star01 = 0, star23 = 0;      // their positions, 16 bits each (only the first ns count)
  uint32_t x = 0;                       // the indel's interval: position | length << 16 (length 0: none)
  PANTAS_HD uint32_t n_stars() const { return ns & 0xFFu; }
  PANTAS_HD bool x_plus() const { return (ns & 0x100u) != 0u; }
  PANTAS_HD uint32_t x_pos() const { return x & 0xFFFFu; }
  PANTAS_HD uint32_t x_len() const { return x >> 16; }
};

PANTAS_HD void short_ops_scan(const uint32_t* ops, uint32_t n_ops, ShortOps& so) {
  uint32_t n = 0;
  for (uint32_t j = 0; j < n_ops; ++j) {
    const uint32_t w = ops[j], code = w & 7u;
    if (code == OP_STAR) {
      if (n == 0) so.star01 = so.T;
      else if (n == 1) so.star01 |= so.T << 16;
      else if (n == 2) so.star23 = so.T;
      else if (n == 3) so.star23 |= so.T << 16;
      if (n < SIMPLE_MAX_STARS) ++n;
    }
    else if (code == OP_PLUS || code == OP_MINUS) { so.x = so.T | ((w >> 3) << 16); if (code == OP_PLUS) so.ns |= 0x100u; }
    so.T += w >> 3;
  }
  so.ns |= n;
}

// true when [lo, hi) (non-empty) holds a base that is not a '*'
PANTAS_HD bool short_has_colon(const ShortOps& so, uint32_t lo, uint32_t hi) {
  const uint32_t len = hi - lo, ns = so.n_stars();
  if (len > ns) return true;
  static_assert(SIMPLE_MAX_STARS == 4, "unrolled below");
  // len <= 4 here; star j counts when it exists (j < ns) and lies inside
  const uint32_t stars_in = ((ns > 0u && (so.star01 & 0xFFFFu) - lo < len) ? 1u : 0u) + ((ns > 1u && (so.star01 >> 16) - lo < len) ? 1u : 0u) +
                            ((ns > 2u && (so.star23 & 0xFFFFu) - lo < len) ? 1u : 0u) + ((ns > 3u && (so.star23 >> 16) - lo < len) ? 1u : 0u);
  return stars_in < len;
}

// the node that takes [P, P + le), le > 0: is it kept, and how many of its compacted entries are neither '-' nor '*'
PANTAS_HD void short_node_summary(const ShortOps& so, uint32_t P, uint32_t le, bool& kept, uint32_t& plain) {
  const uint32_t end = P + le, x_pos = so.x_pos(), x_len = so.x_len(), x_end = x_pos + x_len;
  if (x_len == 0 || x_pos >= end || x_end <= P) {      // no piece of the indel here
    kept = true;
    plain = short_has_colon(so, P, end) ? 1u : 0u;
    return;
  }
  const bool before = x_pos > P, after = x_end < end;
  kept = before || after;                                     // else the lone '+' / '-' piece (:101-102)
  plain = (so.x_plus() ? 1u : 0u) + ((before && short_has_colon(so, P, x_pos)) ? 1u : 0u) + ((after && short_has_colon(so, x_end, end)) ? 1u : 0u);
}

struct KeptNode {
  uint32_t node;
  uint32_t word;       // node_len word as loaded (length + flags)
  uint32_t entries;    // compacted list length
  uint32_t first_op, first_len, last_op, last_len;
  uint32_t plain;      // entries whose op is neither '-' nor '*'
};

// Compaction of one node's op list, one piece at a time (:63-94), and the lone '+'/'-' rule (:101-102).
struct PieceAcc {
  KeptNode kn;
  uint32_t pieces;
  uint32_t piece0_op;
  PANTAS_HD void begin(uint32_t node, uint32_t word) {
    kn = KeptNode{node, word, 0, 0, 0, 0, 0, 0};
    pieces = 0;
    piece0_op = 0;
  }
  PANTAS_HD void add(uint32_t op, uint32_t take) {
    if (pieces == 0) {
      piece0_op = op;
      if (op != OP_STAR) {
        kn.entries = 1;
        kn.first_op = kn.last_op = op;
        kn.first_len = kn.last_len = take;
        kn.plain = (op != OP_MINUS) ? 1u : 0u;
      }
    } else if (kn.entries == 0) {
      kn.entries = 1;
      kn.first_op = kn.last_op = op;
      kn.first_len = kn.last_len = take + 1;
      kn.plain = (op != OP_MINUS && op != OP_STAR) ? 1u : 0u;
    } else if (op == kn.last_op || op == OP_STAR) {
      kn.last_len += take;
      if (kn.entries == 1) kn.first_len = kn.last_len;
    } else {
      ++kn.entries;
      kn.last_op = op;
      kn.last_len = take;
      if (op != OP_MINUS) ++kn.plain;   // op cannot be '*' here
    }
    ++pieces;
  }
  // touched by the read and not a lone insertion / deletion
  PANTAS_HD bool kept() const { return pieces != 0 && !(pieces == 1 && (piece0_op == OP_PLUS || piece0_op == OP_MINUS)); }
};

// Sink concept:
//   uint32_t node_word(uint32_t node)                 gather of the node_len word
//   void nc(uint32_t node)
//   void common(uint32_t node, uint32_t word, uint32_t side, uint32_t times, uint64_t order)
//        IL key 0 / OL key len(node), `times` increments
//   void rare(uint32_t node, uint32_t side, int32_t pos, uint64_t order)   one increment
//   void edge(uint32_t from, uint32_t to, uint64_t order)
template <class Sink>
PANTAS_HD void flush_kept(const KeptNode& kn, uint32_t idx, bool is_last, const LineHdr& h, Sink& sink) {
  const bool rev = (h.flags & LINE_REV) != 0;
  const bool is_first = idx == 0;
  const int32_t len = int32_t(kn.word & LEN_MASK);
  sink.nc(kn.node);
  if (kn.entries == 0) return;
  // which histogram faces the previous kept node / the next kept node of the read
  const uint32_t enter = rev ? SIDE_OL : SIDE_IL;
  const uint32_t leave = rev ? SIDE_IL : SIDE_OL;
  if (!is_first && kn.first_op == OP_MINUS) {
    int32_t pos = rev ? len - 1 - int32_t(kn.first_len) : int32_t(kn.first_len);
    sink.rare(kn.node, enter, pos, event_order(h.line_id, idx, 0));
  }
  if (kn.plain) {
    // forward: IL[0] unless first, OL[len] unless last; reverse: IL[0] unless last, OL[len] unless first
    if (!is_first) sink.common(kn.node, kn.word, enter, kn.plain, event_order(h.line_id, idx, 1));
    if (!is_last) sink.common(kn.node, kn.word, leave, kn.plain, event_order(h.line_id, idx, 1));
  }
  if (!is_last && kn.last_op == OP_MINUS) {
    int32_t pos = rev ? int32_t(kn.last_len) : len - int32_t(kn.last_len) - 1;
    sink.rare(kn.node, leave, pos, event_order(h.line_id, idx, 2));
  }
}

// tc: lengths that leak into lines whose first op is '~' (may be null).  out_stale: where the length the NEXT line would inherit
// goes when this line took an op in hand (host use: the computation of such a table).
template <class Sink>
PANTAS_HD int walk_line(const LineHdr& h, const uint32_t* nodes, uint32_t n_nodes, const uint32_t* ops,
                        uint32_t n_ops, Sink& sink, const TildeCarry* tc = nullptr, uint32_t* out_stale = nullptr) {
  uint32_t oi = 0;          // index of the op in hand (or of the next one when !have)
  uint32_t cur_op = 0, cur_rem = 0;
  bool have = false;
  uint32_t stale = 0;       // the reference's curr_cigar_len: length of the last op in hand that was not a '~' (:228-233)
  bool have_stale = false;
  bool pending = false;
  KeptNode pend{};
  uint32_t kept = 0;        // kept nodes flushed or pending so far
  const bool rev = (h.flags & LINE_REV) != 0;

  for (uint32_t i = 0; i < n_nodes; ++i) {
    const uint32_t node = nodes[i];
    const uint32_t word = sink.node_word(node);
    int64_t room = int64_t(word & LEN_MASK);
    if (i == 0) room -= h.start;
    if (i == n_nodes - 1) room = room - h.tail + 1;
    PieceAcc acc;
    acc.begin(node, word);
    while (room > 0) {
      if (!have) {
        if (oi >= n_ops) return WALK_OPS_EXHAUSTED;   // reference: IndexError at :227
        const uint32_t w = ops[oi];
        cur_op = w & 7u;
        cur_rem = w >> 3;
        have = true;
      }
      if (cur_op == OP_TILDE) {
        // no branch of :228-233 assigns a length for '~': the op is laid out with the length of the op that was in hand
        // one iteration earlier (its own length never matters), and what is written back as "left" is stale - room (:239)
        if (!have_stale) {
          if (!tilde_lookup(tc, h.line_id, stale)) return WALK_TILDE_FIRST;
          have_stale = true;
        }
        cur_rem = stale;
      } else {
        stale = cur_rem;
        have_stale = true;
      }
      uint32_t take;
      const uint32_t op = cur_op;
      if (room <= int64_t(cur_rem)) {
        take = uint32_t(room);
        cur_rem -= take;
        room = 0;
        if (cur_rem == 0) { have = false; ++oi; }
      } else {
        take = cur_rem;
        room -= int64_t(cur_rem);
        have = false;
        ++oi;
      }
      acc.add(op, take);
      if (!have && oi >= n_ops) break;                // ops used up: this node is closed (:252-255)
    }
    if (!acc.kept()) continue;                        // no room, or a lone '+'/'-' (:101-102)
    const KeptNode& kn = acc.kn;
    if (pending) {
      flush_kept(pend, kept - 1, false, h, sink);
      if (rev) sink.edge(node, pend.node, event_order(h.line_id, kept - 1, 0));
      else sink.edge(pend.node, node, event_order(h.line_id, kept - 1, 0));
    }
    pend = kn;
    pending = true;
    ++kept;
  }
  if (pending) flush_kept(pend, kept - 1, true, h, sink);
  if (out_stale && have_stale) *out_stale = stale;      // what `curr_cigar_len` holds behind this line
  return WALK_OK;
}


// ---- the short walk (LINE_SIMPLE / LINE_ONE_INDEL lines) ---------------------------------------------------------------
// Equivalent to walk_line on lines of its classes, without materialising pieces: short_node_summary says what a node keeps.
//   GraphAcc:  NodeInfo info(node)        {length word with the flag bits, target of the node's first out-slot or NONE}
//   Sink:      void kept(node, word, line_id, il_steps, ol_steps, first_slot_hits, exact_order)
//                    one kept node: NC + 1, IL[0] / OL[len] + steps (0 .. 3), RC of the node's FIRST out-slot + hits;
//                    first-seen order of the common key on a flagged side = exact_order, or, when that is 0, any order
//                    of this line (event_order(line_id, *, 1)): a line without '-' has no other key on the node
//              void link_rest(a, b, order)  link a -> b that is NOT a's first out-slot (a later slot, or absent from the graph)
// PF >= 1 = how many nodes ahead the info gathers run (the GPU waits on the chain nodes[] -> info[]): while node i is
// processed, the infos of nodes i+1 .. i+PF and the id of node i+PF+1 are in flight.
struct alignas(8) NodeInfo {
  uint32_t word, first;
};

// DEFER: links that are not a first out-slot are not looked up where they are met -- about one lane in eight of a wave
// meets one in any given pass of the node loop, and a wave pays for the whole lookup (row bounds, column scan, table
// update) whenever ONE of its lanes needs it.  They are handed to the sink as path indices (sink.defer(ia, ib, kept_idx):
// the link nodes[ia] -> nodes[ib] met behind kept node kept_idx) and looked up by sink.drain() after the node loop, when
// most lanes of the wave have one or two of them to do at the same time.
// LEAN: the line is LINE_SIMPLE (':' runs and at most four '*', no '+' / '-'): every node with room is kept, has one
// compacted entry unless all its bases are '*', and no event of the line needs its exact within-line order -- the indel
// and exact-order code is compiled out (the kernel takes this instance for a wave none of whose lines has an indel).
// pre: the line's ShortOps when the caller already holds it (the GPU batch keeps it beside the record: `ops` is then not read)
template <int PF, bool DEFER = false, bool LEAN = false, class GraphAcc, class Sink>
PANTAS_HD int walk_short(const GraphAcc& ga, const LineHdr& h, const uint32_t* nodes, uint32_t k, const uint32_t* ops, uint32_t nops,
                         const Sink& sink, const uint32_t* first_ids,   // first_ids[j] = nodes[min(j, k - 1)], j = 0 .. PF: the caller may
                                                                        // already hold them (the GPU batch keeps them beside the offsets)
                         const ShortOps* pre = nullptr) {
  if (k == 0) return WALK_OK;
  ShortOps so;
  if (pre) so = *pre; else short_ops_scan(ops, nops, so);
  const uint32_t T = so.T;
  const bool rev = (h.flags & LINE_REV) != 0;
  const bool exact = !LEAN && (h.flags & LINE_HAS_DEL) != 0;
  const bool all_colon = (so.ns | so.x) == 0u;      // no '*', no indel (the tokenisers order such lines first inside a block)
  uint32_t P = 0, kept = 0;
  bool pending = false;
  // the pending kept node: id, first out-target, and in one word {plain entries (2 bits), first-slot hit owed (bit 2),
  // path index (bits 8..12), the node's IL / OL flags (bits 30, 31 as in its length word)}
  uint32_t p_node = 0, p_c0 = 0, p_meta = 0;
  // forward: IL[0] unless first, OL[len] unless last; reverse: IL[0] unless last, OL[len] unless first (:298-351), once per
  // compacted entry that is neither '-' nor '*' (p_plain of them)
  auto flush = [&](bool is_last, uint32_t r0) {
    const uint32_t p_plain = p_meta & 3u;
    const bool enter = p_plain != 0u && kept != 1u, leave = p_plain != 0u && !is_last;      // the pending node is kept node number kept - 1
    sink.kept(p_node, p_meta & (FLAG_IL | FLAG_OL), h.line_id, (rev ? leave : enter) ? p_plain : 0u, (rev ? enter : leave) ? p_plain : 0u, r0,
              exact ? event_order(h.line_id, kept - 1u, 1) : 0ull);
  };
  // Pipeline registers: at the top of pass i, nid[j] = id of node i+j (j = 0 .. PF) and inf[j] = info of node i+j
  // (j = 0 .. PF-1).  The pass requests id i+PF+1 and info i+PF, then works on node i.  Every load is unconditional
  // (indices clamped to the path): a load that may or may not have been issued forces the compiler to wait for ALL
  // outstanding loads (s_waitcnt vmcnt(0)) wherever an older one is needed, which undoes the pipeline.  Arrays are only
  // indexed by constants (unrolled), so they are registers.
  const uint32_t last = k - 1u;
  uint32_t nid[PF + 1];
  NodeInfo inf[PF];
PANTAS_UNROLL
  for (int j = 0; j <= PF; ++j) nid[j] = first_ids[j];
PANTAS_UNROLL
  for (int j = 0; j < PF; ++j) inf[j] = ga.info(nid[j]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (uint32_t i = 0; i < k; ++i) {
    const uint32_t node = nid[0], word = inf[0].word, col0 = inf[0].first;
    const uint32_t ahead = i + uint32_t(PF) + 1u;
    const uint32_t new_id = nodes[ahead < last ? ahead : last];
    const NodeInfo new_inf = ga.info(nid[PF]);
    // room the node offers to the ops: its length, less the start offset on the first node, less tail - 1 on the last
    // (:213-216).  32 bits hold it: lengths are below 2^30 and the class keeps both offsets within +-2^29.
    const int32_t r = int32_t(word & LEN_MASK) - (i == 0 ? h.start : 0) - (i == last ? h.tail - 1 : 0);
    const bool has_room = r > 0;
    const uint32_t room = uint32_t(r);
    if (has_room) {
      if (P >= T) return WALK_OPS_EXHAUSTED;
      const uint32_t left = T - P;
      const uint32_t le = room < left ? room : left;
      bool keep;
      uint32_t plain;
      if constexpr (LEAN) { keep = true; plain = short_has_colon(so, P, P + le) ? 1u : 0u; }
      else if (all_colon) { keep = true; plain = 1u; }      // one ':' run after another: every node with room is kept with one plain entry
      else short_node_summary(so, P, le, keep, plain);
      P += le;                      // == T when the ops end inside this node
      if (keep) {
        uint32_t r0_here = 0;
        if (pending) {
          const uint32_t a = rev ? node : p_node, bnode = rev ? p_node : node;
          const bool first_slot = (rev ? col0 : p_c0) == bnode;          // NONE for a node without out-links
          flush(false, rev ? ((p_meta >> 2) & 1u) : (first_slot ? 1u : 0u));
          if (rev && first_slot) r0_here = 1u;
          if (!first_slot) {
            const uint32_t p_i = (p_meta >> 8) & 31u;
            if constexpr (DEFER) sink.defer(rev ? i : p_i, rev ? p_i : i, kept - 1u, a, bnode, h.line_id);
            else sink.link_rest(a, bnode, event_order(h.line_id, kept - 1u, 0));
          }
        }
        p_node = node; p_c0 = col0;
        p_meta = plain | (r0_here << 2) | ((i & 31u) << 8) | (word & (FLAG_IL | FLAG_OL));
        pending = true;
        ++kept;
      }                             // else: the node holds the lone '+' / '-' piece and is dropped (:101-102)
    }
    // advance the pipeline
PANTAS_UNROLL
    for (int j = 0; j < PF; ++j) nid[j] = nid[j + 1];
    nid[PF] = new_id;
PANTAS_UNROLL
    for (int j = 0; j + 1 < PF; ++j) inf[j] = inf[j + 1];
    inf[PF - 1] = new_inf;
  }
  if (pending) flush(true, rev ? ((p_meta >> 2) & 1u) : 0u);
  if constexpr (DEFER) sink.drain(nodes, h.line_id);
  return WALK_OK;
}

// ---- the short walk over 16-byte node records (round 3) ----------------------------------------------------------------
// One link in five of a spliced read is NOT its node's first out-slot (a bubble's second allele, a junction's second
// exit), so three lines in four have one: with the 8-byte record every such link is a lookup of its own (row bounds, a scan
// of col[], a probe of the slot table) that a whole wave pays for.  Here a node's record holds the targets of its first
// THREE out-slots (NONE beyond the degree), and the link to the neighbour kept node is matched against them where the node
// is visited: a hit on slot 0 / 1 / 2 goes into the node's own table entry with its other increments (slot_code 1 / 2 / 3).
// What is left -- links absent from the graph, later slots of a node with more than three -- is handed to the sink
// (link_later: the kernel queues it per workgroup and looks it up at the end of the chunk).
//   forward lines: the link leaves THIS node for the NEXT kept node, which is nearly always the next path node: matched
//     against that id now (it is in the pipeline registers), taken when the next kept node turns out to be that one;
//   reverse lines: the link leaves this node for the PREVIOUS kept node (pending): matched exactly.
// No state but the pending node's id and one meta word crosses a pass of the loop.
//   GraphAcc:  NodeInfo3 info3(node)
//   Sink:      void kept(node, flags, line_id, il_steps, ol_steps, slot_code, exact_order)   slot_code: 0 none, 1 .. 3 = out-slot 0 .. 2 hit once
//              void later(a, b, idx)             the link a -> b, slot unknown (any slot, or absent from the graph); idx = path index of
//                                                the pending node: the link's first-seen order is event_order(line_id, idx, 0) -- idx
//                                                grows from one such link of a line to the next, which is all the order of a link
//                                                absent from the graph is compared on inside a line
struct alignas(16) NodeInfo3 {
  uint32_t word, t0, t1, t2;
};
constexpr uint32_t NO_TARGET = 0xFFFFFFFFu;

template <int PF, class GraphAcc, class Sink>
PANTAS_HD int walk_short_w(const GraphAcc& ga, const LineHdr& h, const uint32_t* nodes, uint32_t k, const Sink& sink, const uint32_t* first_ids,
                           const ShortOps& so) {
  if (k == 0) return WALK_OK;
  const uint32_t T = so.T;
  const bool rev = (h.flags & LINE_REV) != 0;
  const bool exact = (h.flags & LINE_HAS_DEL) != 0;
  const bool all_colon = (so.ns | so.x) == 0u;
  uint32_t P = 0, kept = 0;
  bool pending = false;
  // the pending kept node: id, and {plain entries (bits 0, 1), slot code of its link (bits 2, 3), path index (bits 8 .. 12), IL / OL flags (30, 31)}
  uint32_t p_node = 0, p_meta = 0;
  auto flush = [&](bool is_last, uint32_t code) {
    const uint32_t p_plain = p_meta & 3u;
    const bool enter = p_plain != 0u && kept != 1u, leave = p_plain != 0u && !is_last;
    sink.kept(p_node, p_meta & (FLAG_IL | FLAG_OL), h.line_id, (rev ? leave : enter) ? p_plain : 0u, (rev ? enter : leave) ? p_plain : 0u, code,
              exact ? event_order(h.line_id, kept - 1u, 1) : 0ull);
  };
  const uint32_t last = k - 1u;
  uint32_t nid[PF + 1];
  NodeInfo3 inf[PF];
PANTAS_UNROLL
  for (int j = 0; j <= PF; ++j) nid[j] = first_ids[j];
PANTAS_UNROLL
  for (int j = 0; j < PF; ++j) inf[j] = ga.info3(nid[j]);
  // unrolled PF + 1 times so that the pipeline registers rotate by renaming: a move of a value still in flight is a full wait
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll PF + 1
#endif
  for (uint32_t i = 0; i < k; ++i) {
    const uint32_t node = nid[0];
    const NodeInfo3 cur = inf[0];
    const uint32_t ahead = i + uint32_t(PF) + 1u;
    const uint32_t new_id = nodes[ahead < last ? ahead : last];
    const NodeInfo3 new_inf = ga.info3(nid[PF]);
    const int32_t r = int32_t(cur.word & LEN_MASK) - (i == 0 ? h.start : 0) - (i == last ? h.tail - 1 : 0);
    if (r > 0) {
      if (P >= T) return WALK_OPS_EXHAUSTED;
      const uint32_t room = uint32_t(r), left = T - P;
      const uint32_t le = room < left ? room : left;
      bool keep;
      uint32_t plain;
      if (all_colon) { keep = true; plain = 1u; }
      else short_node_summary(so, P, le, keep, plain);
      P += le;
      if (keep) {
        const uint32_t target = rev ? p_node : nid[1];      // nid[1] = this node again on the last path node: a forward line's last code is never read
        uint32_t code = cur.t0 == target ? 1u : cur.t1 == target ? 2u : cur.t2 == target ? 3u : 0u;
        if (rev && !pending) code = 0u;
        if (pending) {
          uint32_t pc = (p_meta >> 2) & 3u;
          bool later;
          if (!rev) {
            if (((p_meta >> 8) & 31u) + 1u != i) pc = 0u;      // a dropped node or one without room in between: the code was for another target
            later = pc == 0u;
          } else later = code == 0u;
          if (later) sink.later(rev ? node : p_node, rev ? p_node : node, (p_meta >> 8) & 31u);
          flush(false, pc);
        }
        p_node = node;
        p_meta = plain | (code << 2) | ((i & 31u) << 8) | (cur.word & (FLAG_IL | FLAG_OL));
        pending = true;
        ++kept;
      }
    }
PANTAS_UNROLL
    for (int j = 0; j < PF; ++j) nid[j] = nid[j + 1];
    nid[PF] = new_id;
PANTAS_UNROLL
    for (int j = 0; j + 1 < PF; ++j) inf[j] = inf[j + 1];
    inf[PF - 1] = new_inf;
  }
  if (pending) flush(true, rev ? ((p_meta >> 2) & 3u) : 0u);
  return WALK_OK;
}

// ---- the short walk, one lane per path POSITION (round 4) ---------------------------------------------------------------
// walk_short_w gives a record to a lane and loops over its path: 63 % of the lanes work in the walk, 31 % in the table code
// behind it (lines of different lengths, kept / not kept, link found / not found), and the loop control is scalar work per pass.
// Here the path nodes of the short-walk class are laid out in TILES of 64 positions (a record never straddles a tile; a tile
// is padded behind its last record), a wave takes a tile, and every lane owns ONE path position: what the loop carried from
// node to node -- the op-space position P, "is this the first / last kept node", the neighbour kept node -- becomes one wave
// scan of the rooms and bit arithmetic on two 64-bit masks (where records start, which positions are kept).
//   room_i   = len_i - (i == 0 ? start : 0) - (i == last ? tail - 1 : 0), counted when > 0                       (:213-216)
//   P_i      = min(T, sum of the rooms in front of i); the line is refused when a node with room finds P_i >= T
//   le_i     = min(room_i, T - P_i); kept / plain entries from (P_i, le_i) exactly as in walk_short_w
//   link     = forward: to the NEXT kept position, counted in the node's own entry when that is the next path node and one of
//              the node's first three out-targets, else handed to sink.later; reverse: to the PREVIOUS kept position, matched
//              exactly -- the same calls with the same arguments as walk_short_w makes, in another order (all sums and minima).
// The helpers below are what a lane computes for itself; the kernel (augment.hip augment_short_p_kernel) and the CPU harness
// (tests/hostsim/short_walk_host.h, which loops over the 64 lanes) put the same ones around their cross-lane steps.
constexpr uint32_t TILE_LANES = 64;
// Layout of a tile: for every lane its node (pnodes), and pmeta = where the lane stands in its record and which record that is,
// counted from the tile's first: i | (k - 1 - i) << 5 | ordinal << 10 (k <= 31 path nodes, at most 64 records).  A lane behind the
// tile's last record has POS_INACTIVE (and a valid node, so that its gather is harmless).  Per tile one word: the batch index of its
// first record, TILE_NOT_PLAIN set when a record of the tile has a '*' or an indel.
constexpr uint32_t POS_INACTIVE = 0xFFFFu;
constexpr uint32_t TILE_NOT_PLAIN = 1u << 31;
PANTAS_HD uint32_t pos_meta(uint32_t i, uint32_t k, uint32_t ordinal) { return i | ((k - 1u - i) << 5) | (ordinal << 10); }
PANTAS_HD uint32_t pos_ordinal(uint32_t meta) { return meta >> 10; }
// what a position needs of its record: ONE 32-byte sector instead of hdr + rec (64 bytes of two arrays) -- the line's start / tail /
// id / flags and, in the second half, the short walk's view of its cs (ShortOps as DeviceBatch::rec packs it; only read for a tile
// that holds a record with a '*' or an indel)
struct alignas(16) PosLine {
  int32_t start, tail;
  uint32_t line_id;
  uint32_t tw;                   // T (bits 0 .. 15) | POS_REV | POS_HAS_DEL | POS_NOT_PLAIN
};
struct alignas(32) PosRec {
  PosLine line;
  uint32_t so_tn, so_star01, so_star23, so_x;      // T | ns << 16, star01, star23, x
};
static_assert(sizeof(PosRec) == 32, "one sector");
constexpr uint32_t POS_REV = 1u << 16, POS_HAS_DEL = 1u << 17, POS_NOT_PLAIN = 1u << 18;
constexpr uint32_t TILE_ROOM_CAP = 1u << 16;      // rooms only matter up to T < 2^16: capped so that 64 of them add up in 32 bits

PANTAS_HD PosRec make_pos_rec(const LineHdr& h, const ShortOps& so) {
  return PosRec{PosLine{h.start, h.tail, h.line_id,
                        (so.T & 0xFFFFu) | ((h.flags & LINE_REV) ? POS_REV : 0u) | ((h.flags & LINE_HAS_DEL) ? POS_HAS_DEL : 0u) |
                            ((so.ns | so.x) ? POS_NOT_PLAIN : 0u)},
                (so.T & 0xFFFFu) | (so.ns << 16), so.star01, so.star23, so.x};
}
// greedy packing of a chunk's records into tiles, in record order: `fill` = lanes taken in the open tile; true = the record of
// k path nodes (1 <= k <= 64) opens a new tile.  Afterwards fill = lanes taken including this record, which starts at fill - k.
PANTAS_HD bool tile_pack(uint32_t& fill, uint32_t k, bool first_of_chunk) {
  const bool open = first_of_chunk || fill + k > TILE_LANES;
  if (open) fill = 0;
  fill += k;
  return open;
}

PANTAS_HD uint32_t tile_clz64(uint64_t x) {      // x != 0
#if defined(__HIPCC__) || defined(__GNUC__)
  return uint32_t(__builtin_clzll(x));
#else
  uint32_t n = 0; while (!(x >> 63)) { x <<= 1; ++n; } return n;
#endif
}
PANTAS_HD uint32_t tile_ctz64(uint64_t x) {      // x != 0
#if defined(__HIPCC__) || defined(__GNUC__)
  return uint32_t(__builtin_ctzll(x));
#else
  uint32_t n = 0; while (!(x & 1)) { x >>= 1; ++n; } return n;
#endif
}
PANTAS_HD uint32_t tile_popc64(uint64_t x) {
#if defined(__HIPCC__) || defined(__GNUC__)
  return uint32_t(__builtin_popcountll(x));
#else
  uint32_t n = 0; while (x) { x &= x - 1; ++n; } return n;
#endif
}
PANTAS_HD uint64_t tile_lanes_le(uint32_t lane) { return lane >= 63u ? ~0ull : ((2ull << lane) - 1ull); }      // lanes 0 .. lane
PANTAS_HD uint64_t tile_lanes_lt(uint32_t lane) { return (1ull << lane) - 1ull; }                               // lanes 0 .. lane - 1

struct TileLane {
  uint32_t seg_start, seg_end;   // first / last lane of the lane's record
};
PANTAS_HD TileLane tile_lane(uint32_t meta, uint32_t lane) { return TileLane{lane - (meta & 31u), lane + ((meta >> 5) & 31u)}; }
// room of the position (0: none), capped
PANTAS_HD uint32_t tile_room(uint32_t word, bool first, bool last, const PosLine& pr) {
  const int32_t r = int32_t(word & LEN_MASK) - (first ? pr.start : 0) - (last ? pr.tail - 1 : 0);
  const int32_t c = r < int32_t(TILE_ROOM_CAP) ? r : int32_t(TILE_ROOM_CAP);
  return uint32_t(c > 0 ? c : 0);
}
// from the sum of the rooms in front: kept?  plain entries?  refused?   (`so` is only read for a record with POS_NOT_PLAIN)
PANTAS_HD void tile_keep(const PosLine& pr, const ShortOps& so, uint32_t room, uint32_t in_front, bool& keep, uint32_t& plain, bool& exhausted) {
  const uint32_t T = pr.tw & 0xFFFFu;
  exhausted = room != 0u && in_front >= T;
  keep = room != 0u && in_front < T;
  plain = 1u;
  if (keep && (pr.tw & POS_NOT_PLAIN)) {
    // a line with '*' or an indel: still one plain entry on every node longer than the line has '*' that holds no piece of the indel
    const uint32_t left = T - in_front, le = room < left ? room : left, end = in_front + le;
    const bool indel_here = so.x_len() != 0u && so.x_pos() < end && so.x_pos() + so.x_len() > in_front;
    if (indel_here || le <= so.n_stars()) short_node_summary(so, in_front, le, keep, plain);
  }
}
struct TileNeighbour {
  bool is_first, is_last;        // no kept position of the record in front of / behind this one
  bool has;                      // the link's other end exists (forward: a next kept node; reverse: a previous one)
  uint32_t lane;                 // ... and stands at this lane (any lane when there is none)
};
// kept = the tile's kept positions; le = lanes 0 .. lane
PANTAS_HD TileNeighbour tile_neighbour(uint64_t kept, const TileLane& tl, uint32_t lane, uint64_t le, bool rev) {
  const uint64_t before = kept & (le >> 1), after = kept & ~le;      // (le >> 1 = lanes 0 .. lane - 1)
  (void)lane;
  const uint32_t pl = 63u - tile_clz64(before | 1ull), nl = after ? tile_ctz64(after) : 64u;
  const bool has_prev = before != 0 && pl >= tl.seg_start, has_next = nl <= tl.seg_end;
  TileNeighbour n;
  n.is_first = !has_prev;
  n.is_last = !has_next;
  n.has = rev ? has_prev : has_next;
  n.lane = (rev ? pl : nl) & 63u;
  return n;
}
// index of a kept position among the kept positions of its record (only a line with a '-' asks)
PANTAS_HD uint32_t tile_kept_index(uint64_t kept, const TileLane& tl, uint64_t le) {
  return tile_popc64(kept & (le >> 1) & ~tile_lanes_lt(tl.seg_start));
}
// the calls of one kept position (tnode = the node at nb.lane, q = tile_kept_index when the line has a '-')
template <class Sink>
PANTAS_HD void tile_emit(const Sink& sink, uint32_t node, const NodeInfo3& info, uint32_t plain, const PosLine& pr, const TileLane& tl,
                         const TileNeighbour& nb, uint32_t lane, uint32_t tnode, uint32_t q) {
  const bool rev = (pr.tw & POS_REV) != 0;
  // forward, a dropped node or one without room in between: the link's slot is not looked for here, as in walk_short_w
  const bool look = nb.has && (rev || nb.lane == lane + 1u);
  const uint32_t code = !look ? 0u : info.t0 == tnode ? 1u : info.t1 == tnode ? 2u : info.t2 == tnode ? 3u : 0u;
  const bool enter = plain != 0u && !nb.is_first, leave = plain != 0u && !nb.is_last;
  if (nb.has && code == 0u) sink.later(node, tnode, (rev ? nb.lane : lane) - tl.seg_start);
  sink.kept(node, info.word & (FLAG_IL | FLAG_OL), pr.line_id, (rev ? leave : enter) ? plain : 0u, (rev ? enter : leave) ? plain : 0u, code,
            (pr.tw & POS_HAS_DEL) ? event_order(pr.line_id, q, 1) : 0ull);
}

// The rare-key pass over a one-indel line: only the nodes that hold a piece of the '-' can own a deletion-derived key, so
// every other kept node is just counted (its index orders the events) and the piece-by-piece compaction (PieceAcc, fed
// exactly as walk_line feeds it) runs for those one or two nodes alone.  Sink: the walk_line sink concept (rare() matters).
template <class GraphAcc, class Sink>
PANTAS_HD int walk_short_rare(const GraphAcc& ga, const LineHdr& h, const uint32_t* nodes, uint32_t k, const uint32_t* ops, uint32_t nops,
                              Sink& sink) {
  if (k == 0) return WALK_OK;
  ShortOps so;
  short_ops_scan(ops, nops, so);
  const uint32_t T = so.T, x_end = so.x_pos() + so.x_len();
  uint32_t P = 0, kept = 0;
  bool pending = false;
  KeptNode pend{};
  uint32_t node_c = nodes[0];
  uint32_t word_c = ga.info(node_c).word, node_1 = 0, word_1 = 0;
  if (k > 1) { node_1 = nodes[1]; word_1 = ga.info(node_1).word; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (uint32_t i = 0; i < k; ++i) {
    const uint32_t node = node_c, word = word_c;
    node_c = node_1; word_c = word_1;
    if (i + 2 < k) { node_1 = nodes[i + 2]; word_1 = ga.info(node_1).word; }     // (unconditional loads measured slower here)
    const uint32_t len = word & LEN_MASK;
    int64_t r = int64_t(len);
    if (i == 0) r -= h.start;
    if (i == k - 1) r = r - h.tail + 1;
    if (r <= 0) continue;
    if (P >= T) return WALK_OPS_EXHAUSTED;
    const uint32_t room = r > int64_t(0xFFFFFFFFll) ? 0xFFFFFFFFu : uint32_t(r);
    const uint32_t left = T - P;
    const uint32_t le = room < left ? room : left, end = P + le;
    KeptNode kn{node, word, 1u, OP_COLON, le, OP_COLON, le, 1u};      // a node without a piece of the indel: no rare key
    bool keep = true;
    if (so.x_len() != 0 && so.x_pos() < end && x_end > P) {
      PieceAcc acc;
      acc.begin(node, word);
      uint32_t q = 0;
      for (uint32_t j = 0; j < nops && q < end; ++j) {
        const uint32_t w = ops[j], ol = w >> 3;
        const uint32_t a = q > P ? q : P, e = q + ol < end ? q + ol : end;
        if (e > a) acc.add(w & 7u, e - a);
        q += ol;
      }
      keep = acc.kept();
      kn = acc.kn;
    }
    P = end;
    if (!keep) continue;
    if (pending) flush_kept(pend, kept - 1u, false, h, sink);
    pend = kn;
    pending = true;
    ++kept;
  }
  if (pending) flush_kept(pend, kept - 1u, true, h, sink);
  return WALK_OK;
}

}  // namespace pantas
