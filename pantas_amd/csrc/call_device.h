// Per-junction event enumeration of the call path -- the body of the call kernels
// (call.hip instantiates these templates once with a counting emitter and once with a
// writing emitter); a header so that tests/hostsim can compile the same source for the CPU.
// It is NOT a CPU path of the product.
//
// Restated from scripts/call.py:
//   annotated_junction   check_nonnovel  :239-524   (ES :316-375, SS :377-463, IR :465-524)
//   novel_link           check_novel     :531-893   (ES :550-614, SS-1 :616-726, SS-2 :728-835, IR-1 :837-893)
//   novel_around         check_novel     :895-1102  (CE :954-984, IR-2 :985-1102)
//
// Everything is integer set algebra over interned tags (call_host.cpp builds the tables):
// EX tags of a node are a sorted array of tag ids with the transcript and gene of each; the
// JN tags of a link are kept in file order (for printing) and as sorted transcript / gene /
// tag-id sets.  Node names are compared through their lexicographic rank because the reference
// compares node ids as strings (:477, :494, :848, :961, :1038).
//
// Where the reference leaves a choice to Python's set iteration order, a fixed rule is used
// (rules R2-R4 of DESIGN.md): the exon tag / node that appears first in the GFA.
#pragma once
#include <cstddef>
#include <cstdint>

#ifndef PANTAS_HD
#if defined(__HIPCC__)
#define PANTAS_HD __host__ __device__ __forceinline__
#else
#define PANTAS_HD inline
#endif
#endif

namespace pantas {

constexpr uint32_t CNONE = 0xFFFFFFFFu;

// link flags (per canonical slot)
enum : uint32_t { L_JUNC = 1, L_NOVEL = 2, L_HAS_RC = 4, L_GENE_KEYERR = 8, L_FIRST = 16 };
// node flags
enum : uint32_t { N_HAS_NC = 1 };
// event selection
enum : uint32_t { EV_ES = 1, EV_SS = 2, EV_IR = 4 };

// record types (EventRec.type)
enum : uint32_t {
  T_ES_ANN = 0, T_SS_ANN_A = 1, T_SS_ANN_B = 2, T_IR_ANN = 3, T_ES_NOV = 4, T_SS1_NOV_A = 5, T_SS1_NOV_B = 6,
  T_SS2_NOV = 7, T_IR1_NOV = 8, T_CE_NOV = 9, T_IR2_NOV = 10, T_SKIP_ES = 11, T_SKIP_SS = 12, T_SKIP_IR = 13,
  // --debug (scripts/call.py:8-12, :1176-1177 turn eprint into print): one record per "Checking ..." line, in the place
  // of the reference's eprint call; j0 = the junction
  T_DBG_ANN_J = 14, T_DBG_ANN_ES = 15, T_DBG_ANN_SS = 16, T_DBG_ANN_IR = 17, T_DBG_NOV_J = 18, T_DBG_NOV_ES = 19,
  T_DBG_NOV_SS1 = 20, T_DBG_NOV_SS2 = 21, T_DBG_NOV_IR1 = 22, T_DBG_NOV_CE = 23, T_DBG_NOV_IR2 = 24
};

// errors: where the reference raises (or this engine refuses)
enum : uint32_t {
  CE_OK = 0, CE_KEY_RC = 1, CE_KEY_TX = 2, CE_ASSERT_EXONS = 3, CE_ASSERT_JN = 4, CE_ASSERT_PATH = 5, CE_HANG = 6,
  CE_KEY_NC = 7, CE_VALUE_INT = 8, CE_VALUE_MAX = 9, CE_FRONTIER = 10
};

struct EventRec {
  uint32_t type, gene;
  uint32_t j0, j1, j2;       // canonical link slots (meaning depends on type)
  uint32_t side_off, side_len;  // payload words in the side buffer
  uint32_t mean;             // ceil(mean NC) for IR records
};

struct CallView {
  uint32_t V;
  const uint32_t* row_ptr;  const uint32_t* col;                       // out CSR (duplicate link lines keep slots)
  const uint32_t* irow_ptr; const uint32_t* icol; const uint32_t* islot;  // in CSR: source node, out slot
  const uint32_t* first_slot;   // [A] first slot of the same (u, v) pair
  const uint32_t* src;          // [A]
  const uint32_t* lflags;       // [A] valid on canonical slots
  const uint8_t* hid;           // [A] or nullptr: bit 0 / 1 -- the slot is no member of its source's "O" / its target's "I" list
                                // (a later S line of that name restarted the lists, call.py:153-154); gfaL still holds the link
  const int32_t* rc;            // [A]
  const int32_t* nc;            // [V]
  const uint32_t* nflags;       // [V]
  const uint32_t* ln;           // [V] len(sequence)
  const uint32_t* rank;         // [V] lexicographic rank of the segment name
  const uint32_t* nxt1;         // [V] the one out-neighbour of a segment, CNONE when it has none or several
  // Round 5, the intron walks by RUNS (nullptr: none, the walk steps segment by segment): a run is a stretch of segments
  // n, n + 1, ... whose one out-neighbour is the next index -- where "first out-neighbour that ..., else the last" has no choice.
  const uint32_t* run_end_rev;  // [V] at V - 1 - n: the first segment m >= n with nxt1[m] != m + 1 (n .. m is walked in one step)
  const uint64_t* pl;           // [V + 1] sums of ln[0 .. n)
  const uint64_t* pnc;          // [V + 1] sums of nc[0 .. n); the three are only offered when EVERY segment has an NC tag
  // Round 5: one bit per link slot, set when source and target of the slot share an exon tag (nullptr: not offered).  The rows of an
  // expressed gene's segments hold thousands of links the reads added, nearly all of them into OTHER exons; loops that ask every
  // neighbour of a row "do you carry one of my exon tags?" read this bit -- sixteen slots per cache line, in row order -- before
  // they chase the neighbour's tag list through three dependent loads (the stage kernels are one round of waves: their time is
  // their slowest lane, and that lane was walking such a row).
  const uint32_t* share_ex;
  // ... and what the loops over an IN-row ask of every slot, in in-row order (nullptr: not offered): a byte per in-slot q with
  // I_CANON (islot[q] is the first slot of its pair), I_JUNC / I_NOVEL (the pair's first slot carries L_JUNC / L_NOVEL) and I_SHARE
  // (share_ex of islot[q]).  The in-row of a segment lists slots scattered all over the out-CSR: asking first_slot[islot[q]] and
  // lflags[...] per link of a row of thousands was two cache misses per link, and the one junction at such a segment was the
  // whole run time of its kernel (36 M clocks for one item, 2 M for every other: profiles/r5_call_stamps_s4.txt).
  const uint8_t* iflags;
  // Per junction ordinal j = jrow[slot] (nullptr: not offered), made once per table set by a WAVE per junction (call.hip
  // junction_rows_kernel) -- what annotated_junction and novel_around used to find by walking the whole out-row of the junction's
  // head and in-row of its tail, one lane per junction: a lane's pass over a row costs a third of a microsecond per link whatever it
  // does there, the junctions of an expressed gene stand at rows of a thousand links the reads added, and a stage kernel is one
  // round of waves -- its time was the time of the wave that held those junctions (profiles/r5_call_stamps_s4.txt).
  const uint32_t* j_ht_n;                 // [2 nJ] heads / tails of gene g among the junction links at u / v (annotated_junction), | J_KEYERR
  const uint32_t* j_ht;                   // [2 J_HT_MAX nJ] the first J_HT_MAX heads, then tails, in row order
  const unsigned long long* j_carried;    // [2 nJ] bit i: an out-neighbour of u carries u's exon tag i / an in-neighbour of v carries v's tag i
  // Per node, the slots of its out-row (oj: slot indices) and of its in-row (ij: in-slot positions) whose link is a junction, in
  // row order, duplicates included (nullptr: not offered; call.hip junction_slot_lists_kernel).  What novel_link and the search
  // of rule R3 ask a row for is its few junction links, the rows the reads made are thousands of novel links long, and every
  // novel link of such a row asked again: 78 % of the novel-link pass at S5 was the search reading its anchor's row for its targets.
  const uint32_t* oj_ptr; const uint32_t* oj;       // [V + 1], [oj_ptr[V]]
  const uint32_t* ij_ptr; const uint32_t* ij;       // [V + 1], [ij_ptr[V]]
  unsigned long long* stamps;   // timing experiments only (EXPERIMENTS builds)
  uint32_t frontier_lds;        // words of the workgroup's LDS the R3 search may use between its registers and the arena (0: none; frontier_search)
  uint32_t rank_monotone;       // the names' string order is the order of the S lines (rank[n] < rank[n + 1] throughout): what lets the
                                // greedy walk, whose step asks "name <= v's name", take a run as well
  const uint32_t* ex_ptr; const uint32_t* ex_tag; const uint32_t* ex_tx; const uint32_t* ex_gene;  // per node, sorted by tag id
  // JN data hangs off the junction links only (5e5 of 2.6e8 slots at whole-genome scale): jrow[s] = ordinal of slot s
  // among the junction slots or CNONE, the four offset arrays are per junction ordinal (+ 1)
  const uint32_t* jrow;
  const uint32_t* jn_ptr; const uint32_t* jn_ftx;     // per junction, FILE order: filter-transcript of each JN entry
  const uint32_t* jt_ptr; const uint32_t* jt;         // sorted unique JN tag ids
  const uint32_t* jx_ptr; const uint32_t* jx;         // sorted unique transcripts
  const uint32_t* jg_ptr; const uint32_t* jg;         // sorted unique genes
  const uint32_t* tx_gene;      // [n_tx] gene of a transcript, CNONE when the GTF lacks it
  const uint32_t* probe;        // [nJ*6] IR-2 probes per junction: (u,u+1..u+3) then (v-3..v-1,v); canonical slot or CNONE
  const uint32_t* probe_ok;     // [nJ] bit 0: int(u) parses, bit 1: int(v) parses
  int32_t rca, isw, minintron;
  uint32_t events;
  uint32_t debug;              // --debug: emit the T_DBG_* records
  // overflow space of the intronic search: slices of arena_cap words claimed with a counter
  uint32_t* arena;
  uint32_t* arena_next;
  uint32_t arena_slots, arena_cap;
};

// Overflow room of the intronic search: `slots` slices of `cap` words; word 0 of a slice says whether a lane holds it (round 5:
// a slice goes back when its search is over -- level order makes many more searches outgrow their registers than the 4 096 slices
// a pass used to hand out once and for all).  Returns the slice, or `slots` when a whole round of them was busy.
PANTAS_HD uint32_t claim_slice(uint32_t* counter, uint32_t* arena, uint32_t slots, uint32_t cap) {
  for (uint32_t tries = 0; tries < slots; ++tries) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t s = atomicAdd(counter, 1u) % slots;
    if (atomicCAS(arena + size_t(s) * cap, 0u, 1u) == 0u) return s;
#else
    const uint32_t s = (*counter)++ % slots;
    if (arena[size_t(s) * cap] == 0u) { arena[size_t(s) * cap] = 1u; return s; }
#endif
  }
  return slots;
}
PANTAS_HD void release_slice(uint32_t* arena, uint32_t slice, uint32_t cap) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicExch(arena + size_t(slice) * cap, 0u);
#else
  arena[size_t(slice) * cap] = 0u;
#endif
}

// ---- small set helpers ------------------------------------------------------------------------
PANTAS_HD bool sorted_has(const uint32_t* p, uint32_t n, uint32_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (p[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && p[lo] == x;
}

struct Ex { const uint32_t* tag; const uint32_t* tx; const uint32_t* gene; uint32_t n; };

PANTAS_HD Ex ex_of(const CallView& c, uint32_t node) {
  uint32_t b = c.ex_ptr[node];
  return Ex{c.ex_tag + b, c.ex_tx + b, c.ex_gene + b, c.ex_ptr[node + 1] - b};
}
PANTAS_HD bool ex_has(const Ex& s, uint32_t tag) { return sorted_has(s.tag, s.n, tag); }
enum : uint32_t { I_CANON = 1, I_JUNC = 2, I_NOVEL = 4, I_SHARE = 8 };
constexpr uint32_t J_HT_MAX = 16;
constexpr uint32_t J_KEYERR = 0x80000000u;     // a junction link of the row has a transcript the GTF lacks (the row scan raised there)
constexpr uint32_t J_NO_MASK = 0x40000000u;    // more than 64 exon tags at this end: no carried mask
// in-slot q cannot pass a test that needs ALL bits of `need` (true: it may -- or the bytes are not offered)
PANTAS_HD bool in_slot_may(const CallView& c, uint32_t q, uint32_t need) { return c.iflags == nullptr || (c.iflags[q] & need) == need; }
// false: the two ends of link slot t have no exon tag in common (true: they may -- or the bits are not offered)
PANTAS_HD bool may_share_exon(const CallView& c, uint32_t t) { return c.share_ex == nullptr || ((c.share_ex[t >> 5] >> (t & 31u)) & 1u) != 0u; }
// the bit of slot t, by merging the two sorted tag lists (what makes CallView::share_ex)
PANTAS_HD bool ends_share_exon(const uint32_t* ex_ptr, const uint32_t* ex_tag, uint32_t u, uint32_t v) {
  uint32_t a = ex_ptr[u], ae = ex_ptr[u + 1], b = ex_ptr[v], be = ex_ptr[v + 1];
  while (a < ae && b < be) {
    const uint32_t x = ex_tag[a], y = ex_tag[b];
    if (x == y) return true;
    if (x < y) ++a; else ++b;
  }
  return false;
}
// index of tag in s, or CNONE
PANTAS_HD uint32_t ex_find(const Ex& s, uint32_t tag) {
  uint32_t lo = 0, hi = s.n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (s.tag[mid] < tag) lo = mid + 1; else hi = mid;
  }
  return (lo < s.n && s.tag[lo] == tag) ? lo : CNONE;
}
// ex_has for a walk's step: the list's tags four at a time, all asked for at once -- the binary search is three or four trips to
// memory, one behind the other, and a walk's time is its number of trips.
PANTAS_HD bool ex_has_flat(const CallView& c, uint32_t node, uint32_t tag) {
  const uint32_t b = c.ex_ptr[node], n = c.ex_ptr[node + 1] - b;
  bool hit = false;
  for (uint32_t i = 0; i < n; i += 4u) {
    const uint32_t last = n - 1u;
    const uint32_t t0 = c.ex_tag[b + i], t1 = c.ex_tag[b + (i + 1u < n ? i + 1u : last)], t2 = c.ex_tag[b + (i + 2u < n ? i + 2u : last)],
                   t3 = c.ex_tag[b + (i + 3u < n ? i + 3u : last)];
    hit |= t0 == tag || t1 == tag || t2 == tag || t3 == tag;
  }
  return hit;
}
// tag is an exon of `s` that belongs to gene g
PANTAS_HD bool ex_has_gene(const Ex& s, uint32_t tag, uint32_t g) {
  uint32_t i = ex_find(s, tag);
  return i != CNONE && s.gene[i] == g;
}

struct Set { const uint32_t* p; uint32_t n; };
// the range of slot s in one of the per-junction arrays (empty for a slot that is no junction)
PANTAS_HD Set jrange(const CallView& c, const uint32_t* ptr, const uint32_t* data, uint32_t s) {
  const uint32_t j = c.jrow[s];
  if (j == CNONE) return Set{data, 0};
  const uint32_t lo = ptr[j];
  return Set{data + lo, ptr[j + 1] - lo};
}
PANTAS_HD Set jx_of(const CallView& c, uint32_t s) { return jrange(c, c.jx_ptr, c.jx, s); }
PANTAS_HD Set jg_of(const CallView& c, uint32_t s) { return jrange(c, c.jg_ptr, c.jg, s); }
PANTAS_HD Set jt_of(const CallView& c, uint32_t s) { return jrange(c, c.jt_ptr, c.jt, s); }
PANTAS_HD bool set_has(const Set& s, uint32_t x) { return sorted_has(s.p, s.n, x); }

// first slot of row u that leads to v, or CNONE
PANTAS_HD uint32_t find_slot(const CallView& c, uint32_t u, uint32_t v) {
  for (uint32_t s = c.row_ptr[u]; s < c.row_ptr[u + 1]; ++s)
    if (c.col[s] == v) return s;
  return CNONE;
}
PANTAS_HD bool is_junction(const CallView& c, uint32_t u, uint32_t v) {
  uint32_t s = find_slot(c, u, v);
  return s != CNONE && (c.lflags[s] & L_JUNC);
}

// slot s counts as an out-neighbour of its source / an in-neighbour of its target
PANTAS_HD bool in_out_list(const CallView& c, uint32_t s) { return c.hid == nullptr || !(c.hid[s] & 1u); }
PANTAS_HD bool in_in_list(const CallView& c, uint32_t s) { return c.hid == nullptr || !(c.hid[s] & 2u); }

PANTAS_HD uint32_t ceil_mean(uint64_t total, uint32_t n) { return uint32_t((total + n - 1) / n); }

// Error latch of one junction's evaluation
struct Fault {
  uint32_t code = CE_OK;
  PANTAS_HD void raise(uint32_t c) { if (code == CE_OK) code = c; }
  PANTAS_HD bool bad() const { return code != CE_OK; }
};

// ---- walks ----------------------------------------------------------------------------------------
// [:488-499, :843-852] follow the first out-neighbour whose name sorts <= goal's and that
// passes ok(); otherwise the LAST out-neighbour.  pass 0 counts, pass 1 pushes into the emitter.
template <class Emit, class Ok>
PANTAS_HD bool walk_exon(const CallView& c, uint32_t start, uint32_t goal, Ok ok, Emit* emit, uint32_t& n_path,
                         uint64_t& nc_sum, uint64_t& ln_sum, bool& nc_ok, Fault& f) {
  uint32_t cur = start;
  n_path = 0;
  nc_sum = 0;
  ln_sum = 0;
  nc_ok = true;
  uint32_t guard = c.V + 2;
  const uint32_t goal_rank = c.rank[goal];
  for (;;) {
    // A walk through an intron is a chain of trips to memory -- its length is its time -- so everything that hangs on `cur` alone
    // is asked for at once: where the run that begins here ends (round 5), the segment's own numbers, its one out-neighbour
    // and the bounds of its row.
    const uint32_t m = c.run_end_rev ? c.run_end_rev[c.V - 1u - cur] : cur;
    const uint32_t fl = c.nflags[cur], lnv = c.ln[cur];
    const int32_t ncv = c.nc[cur];
    uint32_t one = c.nxt1[cur];
    uint32_t b = c.row_ptr[cur], e = c.row_ptr[cur + 1];
    const uint32_t last = (goal >= cur && goal <= m) ? goal : m;
    if (last > cur) {
      // the run cur .. last in ONE step: consecutive indices, each the only out-neighbour of the one in front ("first
      // out-neighbour that ..., else the last" has no choice there).  The goal inside the run ends the walk.  A second trip
      // brings the prefix sums at both ends and what the last segment needs for the step behind the run.
      const uint32_t k = last - cur;
      if (emit) emit->push_range(cur, last);
      n_path += k + 1u;
      nc_sum += c.pnc[last + 1u] - c.pnc[cur];             // (run arrays are only offered when every segment has an NC tag)
      ln_sum += c.pl[last + 1u] - c.pl[cur];
      if (guard <= k) { f.raise(CE_HANG); return false; }
      guard -= k;
      cur = last;
      if (cur == goal) return true;
      one = c.nxt1[cur];
      b = c.row_ptr[cur]; e = c.row_ptr[cur + 1];
    } else {
      if (emit) emit->push(cur);
      ++n_path;
      if (fl & N_HAS_NC) nc_sum += uint64_t(int64_t(ncv)); else nc_ok = false;   // read only when a row is printed
      ln_sum += lnv;
      if (cur == goal) return true;
    }
    // one out-neighbour: it is taken whether it passes the test or not ("first that ..., else the last")
    if (one != CNONE) {
      cur = one;
      if (--guard == 0) { f.raise(CE_HANG); return false; }
      continue;
    }
    uint32_t nxt = CNONE;
    for (uint32_t s = b; s < e; ++s) {
      if (!in_out_list(c, s)) continue;
      const uint32_t nn = c.col[s];
      nxt = nn;                                     // (the loop variable stays on the last out-neighbour)
      if (c.rank[nn] <= goal_rank && ok(nn)) break;
    }
    if (nxt == CNONE) { f.raise(CE_ASSERT_PATH); return false; }
    cur = nxt;
    if (--guard == 0) { f.raise(CE_HANG); return false; }
    // The segment chosen at a fork is as a rule one allele of a bubble: no run begins there and one way leads on.  Its turn of the
    // loop is taken HERE -- the same bookkeeping -- so that a bubble is one turn for every lane of the wave (lanes that take the
    // fork's turn and the allele's turn out of step with each other pay for both every time).
    if (c.run_end_rev && c.run_end_rev[c.V - 1u - cur] == cur && cur != goal) {
      const uint32_t one2 = c.nxt1[cur];
      if (one2 != CNONE) {
        if (emit) emit->push(cur);
        ++n_path;
        if (c.nflags[cur] & N_HAS_NC) nc_sum += uint64_t(int64_t(c.nc[cur])); else nc_ok = false;
        ln_sum += c.ln[cur];
        cur = one2;
        if (--guard == 0) { f.raise(CE_HANG); return false; }
      }
    }
  }
}

// [:1035-1050] greedy walk of novel IR (2)
template <class Emit>
PANTAS_HD bool walk_greedy(const CallView& c, uint32_t u, uint32_t v, Emit* emit, uint32_t& n_path, uint64_t& nc_sum,
                           uint64_t& ln_inner, bool& nc_ok, Fault& f) {
  uint32_t cur = u;
  n_path = 0;
  nc_sum = 0;
  ln_inner = 0;
  nc_ok = true;
  uint32_t guard = c.V + 2;
  for (;;) {
    if (emit) emit->push(cur);
    ++n_path;
    if (c.nflags[cur] & N_HAS_NC) nc_sum += uint64_t(int64_t(c.nc[cur])); else nc_ok = false;
    if (cur == v) return true;
    if (cur != u) ln_inner += c.ln[cur];
    if (c.run_end_rev && c.rank_monotone && n_path > 1u && cur != u && v > cur) {
      // Round 5, the run that begins at cur in one step (see walk_exon).  A segment x of the run has ONE out-neighbour, x + 1: the
      // greedy rule goes to v when that is v, and else takes it -- it is no later than v by name (string order = S-line order here,
      // and x + 1 < v), it has an out-neighbour itself (it is inside the run) and an NC tag (all segments have).  The step INTO the
      // run's last segment m is left to the rule below: m may be a dead end.
      const uint32_t m = c.run_end_rev[c.V - 1u - cur];
      const uint32_t last = v <= m ? v : (m > cur ? m - 1u : cur);
      if (last > cur) {
        if (emit) emit->push_range(cur + 1u, last);
        n_path += last - cur;
        nc_sum += c.pnc[last + 1u] - c.pnc[cur + 1u];
        ln_inner += c.pl[last == v ? v : last + 1u] - c.pl[cur + 1u];       // (v itself is no inner segment)
        if (guard <= last - cur) { f.raise(CE_HANG); return false; }
        guard -= last - cur;
        cur = last;
        if (cur == v) return true;
      }
    }
    const uint32_t b = c.row_ptr[cur], e = c.row_ptr[cur + 1];
    uint32_t occ = 0;
    for (uint32_t s = b; s < e; ++s) occ += (c.col[s] == v) ? 1u : 0u;
    if (n_path == 1) {
      if (occ == 0) { f.raise(CE_VALUE_MAX); return false; }   // list.remove(x): x not in list
      --occ;
    }
    if (occ > 0) { cur = v; continue; }
    uint32_t best = CNONE;
    int32_t best_nc = 0;
    for (uint32_t s = b; s < e; ++s) {
      const uint32_t x = c.col[s];
      if (x == v || c.rank[x] > c.rank[v]) continue;
      if (c.row_ptr[x + 1] == c.row_ptr[x]) continue;
      if (!(c.nflags[x] & N_HAS_NC)) { f.raise(CE_KEY_NC); return false; }
      if (best == CNONE || c.nc[x] > best_nc) { best = x; best_nc = c.nc[x]; }
    }
    if (best == CNONE) { f.raise(CE_VALUE_MAX); return false; }
    cur = best;
    if (--guard == 0) { f.raise(CE_HANG); return false; }
  }
}

// [:734-750, :787-803] level-bounded frontier search; pop() takes the members of a level, in GFA order, before anything the
// level adds, the hit is the target that comes first in the GFA (rule R3).  dir 0: follow out-links from `start`, targets = {n : (anchor, n) junction};
// dir 1: follow in-links, targets = {p : (p, anchor) junction}.
constexpr uint32_t FRONTIER_LOCAL = 64;
constexpr uint32_t FRONTIER_TARGETS = 8;
// Between the lane's 64 words and a slice of the arena in HBM: 32 KB of the workgroup's LDS, for ONE search of the workgroup at a
// time (frontier_lds_held; a second one that outgrows its registers while the first holds it goes to the arena).  A search that
// runs into a segment with thousands of links is a lane alone with a sort and a copy per level -- every step a dependent access,
// half a microsecond each in HBM (S5: one such search was 20 of the novel-link pass's 22 ms), a few tens of nanoseconds in LDS.
constexpr uint32_t FRONTIER_LDS = 8192;
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ uint32_t frontier_lds[FRONTIER_LDS];
__shared__ uint32_t frontier_lds_held;
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void frontier_lds_init() {      // every kernel that may search calls this first
#if defined(__HIP_DEVICE_COMPILE__)
  if (threadIdx.x == 0) frontier_lds_held = 0u;
  __syncthreads();
#endif
}
#endif
#if defined(PANTAS_EXPERIMENTS) && defined(__HIP_DEVICE_COMPILE__)
#define PANTAS_FRONTIER_T0() const unsigned long long ft0_ = ((c.events & 0x4000u) && buf != local) ? __builtin_amdgcn_s_memtime() : 0ull
#define PANTAS_FRONTIER_T(K) do { if (ft0_) { fph_[K] += __builtin_amdgcn_s_memtime() - ft0_; atomicMax(c.stamps + (K), fph_[K]); } } while (0)
#define PANTAS_FRONTIER_SIZE(N) do { if (c.events & 0x4000u) atomicMax(c.stamps + 3, (unsigned long long)(N)); } while (0)
#else
#define PANTAS_FRONTIER_SIZE(N) do {} while (0)
#define PANTAS_FRONTIER_T0() do {} while (0)
#define PANTAS_FRONTIER_T(K) do {} while (0)
#endif
// Where the words of a search stand -- the lane's own 64, the workgroup's LDS, a slice of the arena in HBM -- is SAID to the
// compiler in the two loops that are all dependent accesses (the sort of a level, its move to the front): through a generic
// pointer every access is a flat one that waits for both memory counters before the next is issued.
struct AnyWords {
  uint32_t* p;
  PANTAS_HD uint32_t get(uint32_t i) const { return p[i]; }
  PANTAS_HD void set(uint32_t i, uint32_t v) const { p[i] = v; }
};
#if defined(__HIP_DEVICE_COMPILE__)
struct LdsWords {
  uint32_t off;
  __device__ __forceinline__ uint32_t get(uint32_t i) const { return frontier_lds[off + i]; }
  __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { frontier_lds[off + i] = v; }
};
struct HbmWords {
  __attribute__((address_space(1))) uint32_t* p;
  __device__ __forceinline__ uint32_t get(uint32_t i) const { return p[i]; }
  __device__ __forceinline__ void set(uint32_t i, uint32_t v) const { p[i] = v; }
};
#endif
// a[0 .. m) sorted (heapsort: in place) and made unique; returns the new count
template <class W>
PANTAS_HD uint32_t sort_unique_words(const W a, uint32_t m) {
  if (m < 2u) return m;
  auto sift = [&](uint32_t root, uint32_t size) {
    const uint32_t v = a.get(root);
    for (;;) {
      uint32_t ch = 2u * root + 1u;
      if (ch >= size) break;
      uint32_t cv = a.get(ch);
      if (ch + 1u < size) { const uint32_t c1 = a.get(ch + 1u); if (c1 > cv) { cv = c1; ++ch; } }
      if (cv <= v) break;
      a.set(root, cv);
      root = ch;
    }
    a.set(root, v);
  };
  for (uint32_t i = m / 2u; i-- > 0u;) sift(i, m);
  for (uint32_t size = m; size-- > 1u;) { const uint32_t t = a.get(0u); a.set(0u, a.get(size)); a.set(size, t); sift(0u, size); }
  uint32_t k = 1;
  uint32_t prev = a.get(0u);
  for (uint32_t i = 1; i < m; ++i) { const uint32_t v = a.get(i); if (v != prev) { a.set(k++, v); prev = v; } }
  return k;
}
// dst[0 .. n) = src[0 .. n), eight words asked for at a time (dst may lie below src in the same buffer)
template <class D, class S>
PANTAS_HD void move_words(const D dst, const S src, uint32_t n) {
  uint32_t i = 0;
  for (; i + 8u <= n; i += 8u) {
    const uint32_t a = src.get(i), b = src.get(i + 1u), c2 = src.get(i + 2u), d = src.get(i + 3u), e = src.get(i + 4u), g = src.get(i + 5u), h = src.get(i + 6u),
                   k = src.get(i + 7u);
    dst.set(i, a); dst.set(i + 1u, b); dst.set(i + 2u, c2); dst.set(i + 3u, d); dst.set(i + 4u, e); dst.set(i + 5u, g); dst.set(i + 6u, h); dst.set(i + 7u, k);
  }
  for (; i < n; ++i) dst.set(i, src.get(i));
}

PANTAS_HD uint32_t frontier_search(const CallView& c, uint32_t start, uint32_t anchor, uint32_t dir, Fault& f) {
  // The targets -- the other ends of the anchor's junction links -- are a handful: taken from the anchor's own row (dir 0) or
  // in-list (dir 1) once, so that a node is tested when it ENTERS the set and against that handful, instead of every member of
  // the set being looked up in a row at every pop (a frontier that runs into a node with thousands of novel links made one lane
  // of the novel-link pass run for 1.3 s at S5).  L_JUNC sits on the first slot of a pair, which is the slot is_junction finds.
  uint32_t tg[FRONTIER_TARGETS];
  uint32_t n_tg = 0;
  bool tg_all = true;           // false: more targets than tg holds -- every test asks is_junction
  {
    const bool lst = c.oj_ptr != nullptr;            // the junction slots of the row alone
    const uint32_t* ptr = lst ? (dir == 0 ? c.oj_ptr : c.ij_ptr) : (dir == 0 ? c.row_ptr : c.irow_ptr);
    const uint32_t* at = dir == 0 ? c.oj : c.ij;
    const uint32_t b = ptr[anchor], e = ptr[anchor + 1];
    for (uint32_t a = b; a < e && tg_all; ++a) {
      const uint32_t s = lst ? at[a] : a;
      const uint32_t slot = dir == 0 ? s : c.islot[s];
      const uint32_t fs = c.first_slot[slot];
      if (fs != slot || !(c.lflags[fs] & L_JUNC)) continue;      // (a later duplicate of a pair: its first slot is listed too)
      if (n_tg == FRONTIER_TARGETS) { tg_all = false; break; }
      tg[n_tg++] = dir == 0 ? c.col[s] : c.icol[s];
    }
  }
  auto is_target = [&](uint32_t x) {
    if (!tg_all) return dir == 0 ? is_junction(c, anchor, x) : is_junction(c, x, anchor);
    bool hit = false;
    for (uint32_t i = 0; i < FRONTIER_TARGETS; ++i) hit |= i < n_tg && tg[i] == x;
    return hit;
  };
  // Rule R3 (round 5: level order).  The reference's `visit` is kept as two stretches of one buffer: lev[0 .. nl), the members the
  // set held when the level began -- its `pvisitl` of them -- sorted by node index and popped in that order, and behind them
  // nxt[0 .. nn), what the pops of this level add, APPENDED as it comes (a node may stand there more than once).  A node that is
  // added while it still waits in lev is in the set already and changes nothing; one that was popped and is added again belongs to
  // the next level, like everything else in nxt.  When the level's last member has been popped, nxt is sorted, made unique and
  // becomes lev.  (Until the first form of this round the set was ONE sorted array: every insertion shifted half of it, and
  // level order -- unlike the depth-first pop it replaced, which left a wide frontier quickly -- expands whole levels: a search that
  // meets a segment with thousands of links added by reads cost n^2 moves; 200 M reads: the novel-link pass 0.75 s instead of 0.2.)
  uint32_t local[FRONTIER_LOCAL];
  uint32_t* buf = local;      // moves to the workgroup's LDS, then to an arena slice, as it outgrows its room
  uint32_t cap = FRONTIER_LOCAL;
  struct Held {               // the LDS and the slice go back wherever the search ends
    uint32_t* arena; uint32_t slice, cap; bool lds;
    PANTAS_HD ~Held() {
#if defined(__HIP_DEVICE_COMPILE__)
      if (lds) atomicExch(&frontier_lds_held, 0u);
#endif
      if (slice != CNONE) release_slice(arena, slice, cap);
    }
  } held{c.arena, CNONE, c.arena_cap, false};
  uint32_t nl = 1, nn = 0, pos = 0;
  // what a level adds often comes in ascending order (the members are popped in that order, and away from a segment the reads hung
  // links on a neighbour is the segment next door): while it does -- a copy of the last addition is simply dropped -- the additions
  // are a sorted, unique level as they stand, and the sort at the level's end (two thirds of a wide search's clocks) is skipped
  bool ascending = true;
  uint32_t last_added = 0;
  buf[0] = start;
  // the smallest target that has entered the set; members only leave by the pop below, which is not reached while there is one
  uint32_t tmin = is_target(start) ? start : CNONE;
  int32_t level = 0;
#if defined(PANTAS_EXPERIMENTS) && defined(__HIP_DEVICE_COMPILE__)
  unsigned long long fph_[3] = {0, 0, 0};      // this search's clocks per phase (the slowest search's show in the stamps)
#endif
  // nxt (buf[from .. from + m)) in place: sorted, then unique; returns the new count
  auto sort_unique = [&](uint32_t from, uint32_t m) -> uint32_t {
#if defined(__HIP_DEVICE_COMPILE__)
    if (held.lds) return sort_unique_words(LdsWords{from}, m);
    if (held.slice != CNONE) return sort_unique_words(HbmWords{(__attribute__((address_space(1))) uint32_t*)(buf + from)}, m);
#endif
    return sort_unique_words(AnyWords{buf + from}, m);
  };
  // the level's additions move to the front
  auto to_front = [&](uint32_t from, uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (held.lds) { move_words(LdsWords{0u}, LdsWords{from}, m); return; }
    if (held.slice != CNONE) {
      move_words(HbmWords{(__attribute__((address_space(1))) uint32_t*)buf}, HbmWords{(__attribute__((address_space(1))) uint32_t*)(buf + from)}, m);
      return;
    }
#endif
    move_words(AnyWords{buf}, AnyWords{buf + from}, m);
  };
  for (;;) {
    if (tmin != CNONE) return level < c.isw ? tmin : CNONE;
    if (level >= c.isw) return CNONE;
    if (pos == nl) {
      // the level is through: what it added is the next one
      PANTAS_FRONTIER_T0();
      PANTAS_FRONTIER_SIZE(nl + nn);
      const uint32_t m = ascending ? nn : sort_unique(nl, nn);
      PANTAS_FRONTIER_T(0);
      if (m == 0u) return CNONE;                                // (the set is empty)
      to_front(nl, m);
      PANTAS_FRONTIER_T(1);
      nl = m; nn = 0; pos = 0;
      ascending = true;
    }
    PANTAS_FRONTIER_T0();
    const uint32_t x = buf[pos++];
    const uint32_t b = dir == 0 ? c.row_ptr[x] : c.irow_ptr[x];
    const uint32_t e = dir == 0 ? c.row_ptr[x + 1] : c.irow_ptr[x + 1];
    // what still waits in lev lies in [wait_lo, wait_hi]: a neighbour outside that range -- against the links' direction nearly
    // every one: the waiting members come after x, its in-neighbours before it -- is not looked up (the lookup is a binary search
    // of dependent loads in the arena: twelve trips per neighbour on a level of thousands)
    const uint32_t wait_lo = pos < nl ? buf[pos] : CNONE, wait_hi = pos < nl ? buf[nl - 1u] : 0u;
    for (uint32_t s = b; s < e; ++s) {
      const uint32_t y = dir == 0 ? c.col[s] : c.icol[s];
      // still waiting in lev (behind pos): in the set already
      if (y >= wait_lo && y <= wait_hi) {
        uint32_t lo = pos, hi = nl;
        while (lo < hi) {
          const uint32_t mid = (lo + hi) >> 1;
          if (buf[mid] < y) lo = mid + 1; else hi = mid;
        }
        if (lo < nl && buf[lo] == y) continue;
      }
      if (nl + nn == cap) {
        if (!ascending) { nn = sort_unique(nl, nn); ascending = true; last_added = nn ? buf[nl + nn - 1u] : 0u; }      // copies of one node give their room back first
        // ... but a buffer that stays three quarters full would be sorted again a few nodes later, and again: the next room up while
        // there is one (the top one is only given up when it is full)
        const bool top = held.slice != CNONE || c.arena == nullptr;
        if (nl + nn == cap || (!top && nl + nn > cap - cap / 4u)) {
          uint32_t* big = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
          if (buf == local && c.frontier_lds > 2u * FRONTIER_LOCAL && atomicCAS(&frontier_lds_held, 0u, 1u) == 0u) {
            held.lds = true;
            big = frontier_lds;
            cap = c.frontier_lds < FRONTIER_LDS ? c.frontier_lds : FRONTIER_LDS;
          }
#endif
          if (!big && !top) {
            const uint32_t slice = claim_slice(c.arena_next, c.arena, c.arena_slots, c.arena_cap);
            if (slice < c.arena_slots) {
              held.slice = slice;
              big = c.arena + size_t(slice) * c.arena_cap + 1u;      // (word 0 is the slice's flag)
              cap = c.arena_cap - 1u;
            }
          }
          if (big) {
            move_words(AnyWords{big}, AnyWords{buf}, nl + nn);
#if defined(__HIP_DEVICE_COMPILE__)
            if (held.lds && big != frontier_lds) { held.lds = false; atomicExch(&frontier_lds_held, 0u); }      // (outgrown as well)
#endif
            buf = big;
          } else if (nl + nn == cap) {                               // nowhere to go and no word left
            f.raise(CE_FRONTIER);
            return CNONE;
          }
        }
      }
      if (ascending && nn != 0u && y <= last_added) {
        if (y == last_added) continue;                          // (the node just added: in the set already)
        ascending = false;                                      // (its place among the last eight additions found and made instead: 15.0 against 15.3 ms at S5 -- dropped)
      }
      if (nn == 0u || y > last_added) last_added = y;
      buf[nl + nn++] = y;
      if (y < tmin && is_target(y)) tmin = y;
    }
    PANTAS_FRONTIER_T(2);
    if (pos == nl) ++level;            // the reference counts the level when its last member has been popped [:746-749]
  }
}

// ---- helpers shared by the three passes ---------------------------------------------------------------
// every exon of `node` must have a transcript the GTF knows (the reference evaluates
// transcript2gene[...] for each of them, :258-275); returns whether some exon is of gene g
PANTAS_HD bool same_gene_exons(const CallView& c, uint32_t node, uint32_t g, Fault& f) {
  const Ex s = ex_of(c, node);
  bool any = false;
  for (uint32_t i = 0; i < s.n; ++i) {
    if (s.gene[i] == CNONE) { f.raise(CE_KEY_TX); return false; }
    any |= s.gene[i] == g;
  }
  return any;
}

// JN entries of link s (file order) whose filter-transcript is in (jx(a) & jx(b)) minus jx(minus)
template <class Emit>
PANTAS_HD void push_filtered(const CallView& c, uint32_t s, uint32_t a, uint32_t b, uint32_t minus, Emit& emit) {
  const Set xa = jx_of(c, a), xb = jx_of(c, b);
  const Set xm = minus == CNONE ? Set{nullptr, 0} : jx_of(c, minus);
  const Set fl = jrange(c, c.jn_ptr, c.jn_ftx, s);
  const uint32_t lo = uint32_t(fl.p - c.jn_ftx), hi = lo + fl.n;
  uint32_t cnt = 0;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t t = c.jn_ftx[i];
    if (set_has(xa, t) && set_has(xb, t) && !set_has(xm, t)) ++cnt;
  }
  emit.push(cnt);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t t = c.jn_ftx[i];
    if (set_has(xa, t) && set_has(xb, t) && !set_has(xm, t)) emit.push(i - lo);
  }
}

// gene of a set of transcripts given by a predicate over candidates; returns CNONE-1 for "several"
constexpr uint32_t GENE_MULTI = 0xFFFFFFFEu;
PANTAS_HD void gene_accumulate(const CallView& c, uint32_t tx, uint32_t& gene, Fault& f) {
  const uint32_t g = c.tx_gene[tx];
  if (g == CNONE) { f.raise(CE_KEY_TX); return; }
  if (gene == CNONE) gene = g;
  else if (gene != g) gene = GENE_MULTI;
}

// timing only (EXPERIMENTS builds, PANTAS_CALL_STAMPS): shader clocks of the sections of a HEAVY item (a row of more than 500 links) summed
// into words 43 .. 47 behind the status word
#if defined(PANTAS_EXPERIMENTS) && defined(__HIP_DEVICE_COMPILE__)
#define PANTAS_SECTION_BEGIN(HEAVY) const bool sec_on_ = (c.events & 0x8000u) && (HEAVY); unsigned long long sec_t_ = sec_on_ ? __builtin_amdgcn_s_memtime() : 0ull
#define PANTAS_SECTION(K) do { if (sec_on_) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(c.stamps + (K), n_ - sec_t_); sec_t_ = n_; } } while (0)
#else
#define PANTAS_SECTION_BEGIN(HEAVY) do {} while (0)
#define PANTAS_SECTION(K) do {} while (0)
#endif
#define PANTAS_DBG(TYPE, SLOT) do { if (c.debug) { emit.begin(TYPE, CNONE, SLOT, CNONE, CNONE); emit.end(0); } } while (0)

// The exon tags of a junction's end, held by the lane (up to 16 of them; every end of the whole-genome workloads has at most 12):
// which of them a neighbour carries is then a compare per pair in registers.  Looked up tag by tag in the end's list -- a binary
// search of dependent loads per tag of the neighbour -- the neighbours of a row of a few thousand links that do share an exon with
// the end cost 12 us each, and their lane held its wave for the length of the kernel.
struct EndTags {
  static constexpr uint32_t N = 16;
  uint32_t t[N];
  bool held;
  Ex ex;
  PANTAS_HD explicit EndTags(const Ex& e) : held(e.n <= N), ex(e) {
    for (uint32_t k = 0; k < N; ++k) t[k] = (held && k < e.n) ? e.tag[k] : CNONE;
  }
  // bit k: the end's tag k is among the tags of `en`
  PANTAS_HD unsigned long long carried_by(const Ex& en) const {
    unsigned long long m = 0;
    if (held) {
      for (uint32_t i = 0; i < en.n; ++i) {
        const uint32_t tg = en.tag[i];
        uint32_t hit = 0;
        for (uint32_t k = 0; k < N; ++k) if (t[k] == tg) hit |= 1u << k;
        m |= hit;
      }
    } else {
      for (uint32_t i = 0; i < en.n; ++i) { const uint32_t k = ex_find(ex, en.tag[i]); if (k != CNONE) m |= 1ull << k; }
    }
    return m;
  }
};

// A row scanned ROW_BATCH neighbours at a time: which neighbours the filters leave (node_at: CNONE = dropped), where their tag lists
// stand and the lists themselves are asked for TOGETHER -- one trip to memory per batch and step instead of two per neighbour -- and
// each(position, tags of the end the neighbour carries) is then called in row order (false = stop, scan_row returns false).  A lane's
// pass over a row of thousands of links of which hundreds share an exon with the end was a chain of dependent loads per neighbour.
constexpr uint32_t ROW_BATCH = 8;
template <class NodeAt, class Each>
PANTAS_HD bool scan_row(const CallView& c, uint32_t b, uint32_t e, const EndTags& end, NodeAt node_at, Each each) {
  for (uint32_t p0 = b; p0 < e; p0 += ROW_BATCH) {
    uint32_t nd[ROW_BATCH], lo[ROW_BATCH], len[ROW_BATCH];
    unsigned long long mine[ROW_BATCH];
    uint32_t longest = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < ROW_BATCH; ++k) nd[k] = p0 + k < e ? node_at(p0 + k) : CNONE;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t k = 0; k < ROW_BATCH; ++k) {
      const uint32_t n = nd[k] == CNONE ? 0u : nd[k];
      lo[k] = c.ex_ptr[n];
      len[k] = nd[k] == CNONE ? 0u : c.ex_ptr[n + 1u] - lo[k];
      mine[k] = 0;
      if (len[k] > longest) longest = len[k];
    }
    if (end.held) {
      for (uint32_t i = 0; i < longest; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (uint32_t k = 0; k < ROW_BATCH; ++k) {
          if (i >= len[k]) continue;
          const uint32_t tg = c.ex_tag[lo[k] + i];
          uint32_t hit = 0;
          for (uint32_t j = 0; j < EndTags::N; ++j) if (end.t[j] == tg) hit |= 1u << j;
          mine[k] |= hit;
        }
      }
    } else {
      for (uint32_t k = 0; k < ROW_BATCH; ++k)
        if (len[k]) mine[k] = end.carried_by(Ex{c.ex_tag + lo[k], c.ex_tx + lo[k], c.ex_gene + lo[k], len[k]});
    }
    for (uint32_t k = 0; k < ROW_BATCH; ++k)
      if (nd[k] != CNONE && !each(p0 + k, mine[k])) return false;
  }
  return true;
}

// ---- pass 1: annotated junctions (:239-524) ----------------------------------------------------------------
template <class Emit>
PANTAS_HD void annotated_junction(const CallView& c, uint32_t s, Emit& emit, Fault& f) {
  const uint32_t u = c.src[s], v = c.col[s];
  if (!(c.lflags[s] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
  if (c.rc[s] < c.rca) return;
  if (c.lflags[s] & L_GENE_KEYERR) { f.raise(CE_KEY_TX); return; }
  const Set genes = jg_of(c, s);
  if (genes.n != 1) return;
  const uint32_t g = genes.p[0];
  PANTAS_DBG(T_DBG_ANN_J, s);                       // :254
  const bool has_u = same_gene_exons(c, u, g, f);
  if (f.bad()) return;
  const bool has_v = same_gene_exons(c, v, g, f);
  if (f.bad()) return;
  if (!has_u || !has_v) { f.raise(CE_ASSERT_EXONS); return; }
  const Ex exu = ex_of(c, u), exv = ex_of(c, v);
  const Set txs = jx_of(c, s);
  // heads: other junctions leaving u; tails: other junctions entering v; both of gene g
  const uint32_t ob = c.row_ptr[u], oe = c.row_ptr[u + 1];
  const uint32_t ib = c.irow_ptr[v], ie = c.irow_ptr[v + 1];
  // The rows of u and v also hold every link the reads added (thousands on expressed genes); the
  // junctions among them are few, so they are listed once and the pair loops below run over the lists.
  constexpr uint32_t HT_MAX = J_HT_MAX;
  uint32_t heads[HT_MAX], tails[HT_MAX];
  uint32_t n_heads = 0, n_tails = 0;
  PANTAS_SECTION_BEGIN(oe - ob > 500u || ie - ib > 500u);
  const bool listed = c.j_ht_n != nullptr;
  if (listed) {
    const uint32_t j = c.jrow[s];
    const uint32_t nh = c.j_ht_n[2u * j], nt = c.j_ht_n[2u * j + 1u];
    if ((nh | nt) & J_KEYERR) { f.raise(CE_KEY_TX); return; }
    n_heads = nh & ~(J_KEYERR | J_NO_MASK);
    n_tails = nt & ~(J_KEYERR | J_NO_MASK);
    for (uint32_t i = 0; i < HT_MAX; ++i) {
      heads[i] = i < n_heads ? c.j_ht[2u * HT_MAX * j + i] : CNONE;
      tails[i] = i < n_tails ? c.j_ht[2u * HT_MAX * j + HT_MAX + i] : CNONE;
    }
  }
  for (uint32_t t = ob; t < oe && !listed; ++t) {
    if (c.first_slot[t] != t || t == s || !(c.lflags[t] & L_JUNC)) continue;
    if (c.lflags[t] & L_GENE_KEYERR) { f.raise(CE_KEY_TX); return; }
    if (set_has(jg_of(c, t), g)) { if (n_heads < HT_MAX) heads[n_heads] = t; ++n_heads; }
  }
  for (uint32_t q = ib; q < ie && !listed; ++q) {
    if (!in_slot_may(c, q, I_CANON | I_JUNC)) continue;
    const uint32_t t = c.islot[q];
    if (c.first_slot[t] != t || t == s || !(c.lflags[t] & L_JUNC)) continue;
    if (c.lflags[t] & L_GENE_KEYERR) { f.raise(CE_KEY_TX); return; }
    if (set_has(jg_of(c, t), g)) { if (n_tails < HT_MAX) tails[n_tails] = t; ++n_tails; }
  }
  auto is_head = [&](uint32_t t) {
    return c.first_slot[t] == t && t != s && (c.lflags[t] & L_JUNC) && set_has(jg_of(c, t), g);
  };
  // i-th head / tail; beyond the local list (rare) it is found by scanning the row again
  auto head_at = [&](uint32_t i) -> uint32_t {
    if (i < HT_MAX) return heads[i];
    uint32_t k = 0;
    for (uint32_t t = ob; t < oe; ++t) if (is_head(t)) { if (k == i) return t; ++k; }
    return CNONE;
  };
  auto tail_at = [&](uint32_t i) -> uint32_t {
    if (i < HT_MAX) return tails[i];
    uint32_t k = 0;
    for (uint32_t q = ib; q < ie; ++q) { if (!in_slot_may(c, q, I_CANON | I_JUNC)) continue; const uint32_t t = c.islot[q]; if (is_head(t)) { if (k == i) return t; ++k; } }
    return CNONE;
  };

  PANTAS_SECTION(0);
  if (c.events & EV_ES) {
    PANTAS_DBG(T_DBG_ANN_ES, s);                    // :317
    for (uint32_t hi_ = 0; hi_ < n_heads; ++hi_) {
      const uint32_t t1 = head_at(hi_);
      const Set x1 = jx_of(c, t1);
      for (uint32_t ti_ = 0; ti_ < n_tails; ++ti_) {
        const uint32_t t2 = tail_at(ti_);
        const Set x2 = jx_of(c, t2);
        bool inc = false;
        for (uint32_t i = 0; i < x1.n && !inc; ++i) inc = set_has(x2, x1.p[i]) && !set_has(txs, x1.p[i]);
        if (!inc) continue;
        // an exon on the far ends that is on neither end of the junction (for gene g)
        bool fresh = false;
        const Ex a = ex_of(c, c.col[t1]), b = ex_of(c, c.src[t2]);
        for (uint32_t i = 0; i < a.n && !fresh; ++i) fresh = !ex_has_gene(exu, a.tag[i], g) && !ex_has_gene(exv, a.tag[i], g);
        for (uint32_t i = 0; i < b.n && !fresh; ++i) fresh = !ex_has_gene(exu, b.tag[i], g) && !ex_has_gene(exv, b.tag[i], g);
        if (!fresh) continue;
        emit.begin(T_ES_ANN, g, s, t1, t2);
        push_filtered(c, t1, t1, t2, s, emit);
        push_filtered(c, t2, t1, t2, s, emit);
        emit.end(0);
      }
    }
  }

  PANTAS_SECTION(1);
  if (c.events & EV_SS) {
    PANTAS_DBG(T_DBG_ANN_SS, s);                    // :378
    // Round 5.  Both SS loops ask, for every neighbour n of a row and every tail (head) of the junction, whether an exon tag of
    // the junction's end is carried by n AND by the tail's far end -- a tag list walked per (neighbour, tail) pair, at the rows of
    // an expressed gene hundreds of neighbours times a dozen tails.  With at most 64 exon tags at the end the answers are bit
    // masks over its tag list: one per tail (made once), one per neighbour, a pair is two ANDs.
    // (PANTAS_CALL_NO_SHARE_BITS leaves the pair-by-pair loops below, the A/B the parity tests compare.)
    const bool by_masks = c.share_ex && exu.n <= 64u && exv.n <= 64u && n_tails <= HT_MAX && n_heads <= HT_MAX;
    unsigned long long far_has[HT_MAX], far_new[HT_MAX];       // per tail / head: tags of gene g its far end carries; those of a transcript the junction lacks
    if (n_tails) {
      unsigned long long of_gene = 0;
      const EndTags end_tags(exu);
      if (by_masks) {
        for (uint32_t k = 0; k < exu.n; ++k) if (exu.gene[k] == g) of_gene |= 1ull << k;
        for (uint32_t ti_ = 0; ti_ < n_tails; ++ti_) {
          far_has[ti_] = 0; far_new[ti_] = 0;
          const uint32_t from = c.src[tails[ti_]];
          if (from == u) continue;
          const Ex ef = ex_of(c, from);
          for (uint32_t i = 0; i < ef.n; ++i) {
            const uint32_t k = ex_find(exu, ef.tag[i]);
            if (k == CNONE || exu.gene[k] != g) continue;
            far_has[ti_] |= 1ull << k;
            if (!set_has(txs, ef.tx[i])) far_new[ti_] |= 1ull << k;
          }
        }
      }
      if (by_masks) {
        // a tail that shares a transcript with the junction [the assert of :405] stops the run where its first row would be written
        uint32_t clash = 0;
        for (uint32_t ti_ = 0; ti_ < n_tails; ++ti_) {
          const Set ta = jt_of(c, s), tb = jt_of(c, tails[ti_]);
          for (uint32_t i = 0; i < ta.n; ++i) if (set_has(tb, ta.p[i])) clash |= 1u << ti_;
        }
        const bool through = scan_row(c, ob, oe, end_tags,
          [&](uint32_t t) -> uint32_t {               // every out-link line of u, duplicates included
            if (!in_out_list(c, t) || !may_share_exon(c, t)) return CNONE;
            const uint32_t n = c.col[t];
            return n == v ? CNONE : n;
          },
          [&](uint32_t, unsigned long long mine) -> bool {      // mine: tags of u that the neighbour carries
            if (!(mine & of_gene)) return true;
            for (uint32_t ti_ = 0; ti_ < n_tails; ++ti_) {
              if (!(far_has[ti_] & mine) || !(far_new[ti_] & mine)) continue;
              if ((clash >> ti_) & 1u) { f.raise(CE_ASSERT_JN); return false; }
              emit.begin(T_SS_ANN_A, g, s, tails[ti_], CNONE);
              emit.end(0);
            }
            return true;
          });
        if (!through) return;
      }
      for (uint32_t t = ob; t < oe && !by_masks; ++t) {          // every out-link line of u, duplicates included
        if (!in_out_list(c, t)) continue;
        if (!may_share_exon(c, t)) continue;        // (the neighbour carries none of u's exons: `cont` below would stay false)
        const uint32_t n = c.col[t];
        if (n == v) continue;
        const Ex en = ex_of(c, n);
        bool cont = false;
        for (uint32_t i = 0; i < en.n && !cont; ++i) cont = ex_has_gene(exu, en.tag[i], g);
        if (!cont) continue;
        for (uint32_t ti_ = 0; ti_ < n_tails; ++ti_) {
          const uint32_t t2 = tail_at(ti_);
          const uint32_t from = c.src[t2];
          if (from == u) continue;
          const Ex ef = ex_of(c, from);
          bool shared = false, changes = false;
          for (uint32_t i = 0; i < ef.n; ++i) {
            const uint32_t tg = ef.tag[i];
            if (ex_has(en, tg) && ex_has_gene(exu, tg, g)) {
              shared = true;
              if (!set_has(txs, ef.tx[i])) changes = true;
            }
          }
          if (!shared || !changes) continue;
          const Set ta = jt_of(c, s), tb = jt_of(c, t2);
          for (uint32_t i = 0; i < ta.n; ++i)
            if (set_has(tb, ta.p[i])) { f.raise(CE_ASSERT_JN); return; }
          emit.begin(T_SS_ANN_A, g, s, t2, CNONE);
          emit.end(0);
        }
      }
    }
    if (n_heads) {
      unsigned long long of_gene = 0;
      const EndTags end_tags(exv);
      if (by_masks) {
        for (uint32_t k = 0; k < exv.n; ++k) if (exv.gene[k] == g) of_gene |= 1ull << k;
        for (uint32_t hi_ = 0; hi_ < n_heads; ++hi_) {
          far_has[hi_] = 0; far_new[hi_] = 0;
          const uint32_t to = c.col[heads[hi_]];
          if (to == v) continue;
          const Ex et = ex_of(c, to);
          for (uint32_t i = 0; i < et.n; ++i) {
            const uint32_t k = ex_find(exv, et.tag[i]);
            if (k == CNONE || exv.gene[k] != g) continue;
            far_has[hi_] |= 1ull << k;
            if (!set_has(txs, et.tx[i])) far_new[hi_] |= 1ull << k;
          }
        }
      }
      if (by_masks) {
        scan_row(c, ib, ie, end_tags,
          [&](uint32_t q0) -> uint32_t {              // every in-link line of v
            if (!in_slot_may(c, q0, I_SHARE)) return CNONE;
            const uint32_t sl = c.islot[q0];
            if (!in_in_list(c, sl) || !may_share_exon(c, sl)) return CNONE;
            const uint32_t n = c.icol[q0];
            return n == u ? CNONE : n;
          },
          [&](uint32_t, unsigned long long mine) -> bool {      // mine: tags of v that the neighbour carries
            if (!(mine & of_gene)) return true;
            for (uint32_t hi_ = 0; hi_ < n_heads; ++hi_) {
              if (!(far_has[hi_] & mine) || !(far_new[hi_] & mine)) continue;
              emit.begin(T_SS_ANN_B, g, s, heads[hi_], CNONE);
              emit.end(0);
            }
            return true;
          });
      }
      for (uint32_t q0 = ib; q0 < ie && !by_masks; ++q0) {        // every in-link line of v
        if (!in_slot_may(c, q0, I_SHARE)) continue;        // (the neighbour carries none of v's exons: `cont` below would stay false)
        if (!in_in_list(c, c.islot[q0])) continue;
        if (!may_share_exon(c, c.islot[q0])) continue;
        const uint32_t n = c.icol[q0];
        if (n == u) continue;
        const Ex en = ex_of(c, n);
        bool cont = false;
        for (uint32_t i = 0; i < en.n && !cont; ++i) cont = ex_has_gene(exv, en.tag[i], g);
        if (!cont) continue;
        for (uint32_t hi_ = 0; hi_ < n_heads; ++hi_) {
          const uint32_t t1 = head_at(hi_);
          const uint32_t to = c.col[t1];
          if (to == v) continue;
          const Ex et = ex_of(c, to);
          bool shared = false, changes = false;
          for (uint32_t i = 0; i < et.n; ++i) {
            const uint32_t tg = et.tag[i];
            if (ex_has(en, tg) && ex_has_gene(exv, tg, g)) {
              shared = true;
              if (!set_has(txs, et.tx[i])) changes = true;
            }
          }
          if (!shared || !changes) continue;
          emit.begin(T_SS_ANN_B, g, s, t1, CNONE);
          emit.end(0);
        }
      }
    }
  }

  PANTAS_SECTION(2);
  if (c.events & EV_IR) {
    PANTAS_DBG(T_DBG_ANN_IR, s);                    // :466
    // exons (of gene g) that lie on both ends of the junction: only they can retain the intron.  Testing
    // this first, and per out-neighbour, keeps the out x in neighbour product of :473 off hub nodes.
    bool both_any = false;
    for (uint32_t i = 0; i < exu.n && !both_any; ++i) both_any = exu.gene[i] == g && ex_has_gene(exv, exu.tag[i], g);
    bool done = !both_any;
    for (uint32_t t = ob; t < oe && !done; ++t) {
      if (!in_out_list(c, t)) continue;
      if (!may_share_exon(c, t)) continue;          // (n1_any below needs a tag of n1 that is also u's)
      const uint32_t n1 = c.col[t];
      if (n1 == v) continue;
      const Ex e1 = ex_of(c, n1);
      bool n1_any = false;
      for (uint32_t i = 0; i < e1.n && !n1_any; ++i) n1_any = ex_has_gene(exu, e1.tag[i], g) && ex_has_gene(exv, e1.tag[i], g);
      if (!n1_any) continue;
      for (uint32_t q = ib; q < ie && !done; ++q) {
        if (!in_slot_may(c, q, I_SHARE)) continue;
        if (!in_in_list(c, c.islot[q])) continue;
        if (!may_share_exon(c, c.islot[q])) continue;      // (`pick` below is a tag of n2 that is also v's)
        const uint32_t n2 = c.icol[q];
        if (n2 == u) continue;
        if (c.rank[n2] < c.rank[n1]) continue;
        const Ex e2 = ex_of(c, n2);
        uint32_t pick = CNONE;   // smallest tag id in EX(n1) & EX(n2) & ex_u & ex_v (rule R2)
        for (uint32_t i = 0; i < e1.n; ++i) {
          const uint32_t tg = e1.tag[i];
          if (ex_has(e2, tg) && ex_has_gene(exu, tg, g) && ex_has_gene(exv, tg, g)) { pick = tg; break; }
        }
        if (pick == CNONE) continue;
        // the intron walk itself is deferred: tasks are compacted and walked by their own kernels
        emit.begin(T_IR_ANN, g, s, CNONE, CNONE);
        emit.task();
        emit.push(n1);
        emit.push(n2);
        emit.push(pick);
        emit.end(0);
        done = true;
      }
    }
  }
  PANTAS_SECTION(3);
}

// ---- pass 2: links seen only in reads (:531-893) ------------------------------------------------------------
template <class Emit>
PANTAS_HD void novel_link(const CallView& c, uint32_t s, Emit& emit, Fault& f) {
  const uint32_t u = c.src[s], v = c.col[s];
  if (!(c.lflags[s] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
  if (c.rc[s] < c.rca) return;
  const Ex ex0 = ex_of(c, u), ex1 = ex_of(c, v);
  if (ex0.n == 0 && ex1.n == 0) return;
  PANTAS_DBG(T_DBG_NOV_J, s);                       // :541
  // (every loop below keeps the junction links of a row and drops the rest: with the per-node lists it runs over those alone)
  const bool lst = c.oj_ptr != nullptr;
  const uint32_t ojb = lst ? c.oj_ptr[u] : c.row_ptr[u], oje = lst ? c.oj_ptr[u + 1] : c.row_ptr[u + 1];
  const uint32_t ijb = lst ? c.ij_ptr[v] : c.irow_ptr[v], ije = lst ? c.ij_ptr[v + 1] : c.irow_ptr[v + 1];
  // cap = transcripts with an exon on both ends
  auto in_cap = [&](uint32_t tx) {
    bool a = false, b = false;
    for (uint32_t i = 0; i < ex0.n && !a; ++i) a = ex0.tx[i] == tx;
    for (uint32_t i = 0; i < ex1.n && !b; ++i) b = ex1.tx[i] == tx;
    return a && b;
  };
  bool cap_any = false;
  for (uint32_t i = 0; i < ex0.n && !cap_any; ++i)
    for (uint32_t k = 0; k < ex1.n && !cap_any; ++k) cap_any = ex0.tx[i] == ex1.tx[k];

  PANTAS_SECTION_BEGIN(true);
  if (c.events & EV_ES) {
    PANTAS_DBG(T_DBG_NOV_ES, s);                    // :551
    if (ex0.n && ex1.n) {
      uint32_t both = 0;
      for (uint32_t i = 0; i < ex0.n; ++i) both += ex_has(ex1, ex0.tag[i]) ? 1u : 0u;
      if (both != ex0.n) {
        for (uint32_t a = ojb; a < oje; ++a) {
          const uint32_t t = lst ? c.oj[a] : a;
          const uint32_t j1 = c.first_slot[t];
          if (!(c.lflags[j1] & L_JUNC)) continue;
          for (uint32_t a2 = ijb; a2 < ije; ++a2) {
            const uint32_t q = lst ? c.ij[a2] : a2;
            if (!in_slot_may(c, q, I_JUNC)) continue;
            const uint32_t j2 = c.first_slot[c.islot[q]];
            if (!(c.lflags[j2] & L_JUNC)) continue;
            const Set x1 = jx_of(c, j1), x2 = jx_of(c, j2);
            uint32_t gene = CNONE;
            bool any = false;
            for (uint32_t i = 0; i < x1.n; ++i)
              if (set_has(x2, x1.p[i])) { any = true; gene_accumulate(c, x1.p[i], gene, f); }
            if (!any) continue;
            if (f.bad()) return;
            if (gene == GENE_MULTI) { emit.begin(T_SKIP_ES, CNONE, s, CNONE, CNONE); emit.end(0); continue; }
            emit.begin(T_ES_NOV, gene, s, j1, j2);
            push_filtered(c, j1, j1, j2, CNONE, emit);
            push_filtered(c, j2, j1, j2, CNONE, emit);
            emit.end(0);
          }
        }
      }
    }
  }

  PANTAS_SECTION(0);
  if (c.events & EV_SS) {
    if (cap_any) PANTAS_DBG(T_DBG_NOV_SS1, s);      // :618
    if (cap_any) {
      // side 0: annotated junctions (u, n) into an exon that also covers v; side 1: (p, v) out of an exon covering u
      for (uint32_t side = 0; side < 2; ++side) {
        const Ex spl = side == 0 ? ex1 : ex0;
        if (spl.n == 0) continue;
        const uint32_t b = side == 0 ? ojb : ijb, e = side == 0 ? oje : ije;
        for (uint32_t a = b; a < e; ++a) {
          const uint32_t t = lst ? (side == 0 ? c.oj[a] : c.ij[a]) : a;
          if (side == 1 && !in_slot_may(c, t, I_JUNC)) continue;
          const uint32_t n = side == 0 ? c.col[t] : c.icol[t];
          const uint32_t jj = c.first_slot[side == 0 ? t : c.islot[t]];
          if (!(c.lflags[jj] & L_JUNC)) continue;       // (asked last until round 5: the rows are long, the junctions in them few)
          const Ex en = ex_of(c, n);
          bool shared = false;
          uint32_t gene = CNONE;
          bool keep = false;
          for (uint32_t i = 0; i < en.n; ++i) {
            if (!ex_has(spl, en.tag[i])) continue;
            shared = true;
            if (in_cap(en.tx[i])) keep = true;
          }
          if (!shared || !keep || !(c.lflags[jj] & L_JUNC)) continue;
          for (uint32_t i = 0; i < en.n; ++i)
            if (ex_has(spl, en.tag[i]) && in_cap(en.tx[i])) gene_accumulate(c, en.tx[i], gene, f);
          if (f.bad()) return;
          if (gene == GENE_MULTI) { emit.begin(T_SKIP_SS, CNONE, s, CNONE, CNONE); emit.end(0); continue; }
          emit.begin(side == 0 ? T_SS1_NOV_A : T_SS1_NOV_B, gene, s, jj, CNONE);
          emit.end(0);
        }
      }
    }
    PANTAS_SECTION(1);
    PANTAS_DBG(T_DBG_NOV_SS2, s);                   // :728
    uint32_t j1 = CNONE;
    if (ex0.n && !ex1.n) {
      const uint32_t hit = frontier_search(c, v, u, 0, f);
      if (f.bad()) return;
      if (hit != CNONE) j1 = find_slot(c, u, hit);
    } else if (!ex0.n && ex1.n) {
      const uint32_t hit = frontier_search(c, u, v, 1, f);
      if (f.bad()) return;
      if (hit != CNONE) j1 = find_slot(c, hit, v);
    }
    if (j1 != CNONE) {
      if (c.lflags[j1] & L_GENE_KEYERR) { f.raise(CE_KEY_TX); return; }
      const Set gs = jg_of(c, j1);
      if (gs.n > 1) { emit.begin(T_SKIP_SS, CNONE, s, CNONE, CNONE); emit.end(0); }
      else { emit.begin(T_SS2_NOV, gs.p[0], j1, s, CNONE); emit.end(0); }
    }
  }

  PANTAS_SECTION(2);
  if ((c.events & EV_IR) && cap_any) {
    PANTAS_DBG(T_DBG_NOV_IR1, s);                   // :839
    bool inside_any = false;
    for (uint32_t i = 0; i < ex0.n && !inside_any; ++i) inside_any = ex_has(ex1, ex0.tag[i]);
    if (inside_any) {
      // gene of the retained exons now, the walk later (run_walk_task); a transcript the GTF lacks only
      // matters if the intron turns out long enough (:853-858), so it travels as a status
      uint32_t gene = CNONE, n_in = 0;
      Fault gf;
      for (uint32_t i = 0; i < ex0.n; ++i)
        if (ex_has(ex1, ex0.tag[i])) { ++n_in; gene_accumulate(c, ex0.tx[i], gene, gf); }
      emit.begin(T_IR1_NOV, gf.bad() ? CNONE : gene, s, CNONE, CNONE);
      emit.task();
      emit.push(n_in);
      for (uint32_t i = 0; i < ex0.n; ++i)
        if (ex_has(ex1, ex0.tag[i])) emit.push(ex0.tag[i]);
      emit.end(0);
    }
  }
}

// ---- pass 3: annotated junctions against the novel links around them (:895-1102) --------------------------
template <class Emit>
PANTAS_HD void novel_around(const CallView& c, uint32_t s, uint32_t jidx, Emit& emit, Fault& f) {
  if (!(c.events & (EV_IR | EV_ES))) return;
  const uint32_t u = c.src[s], v = c.col[s];
  if (!(c.lflags[s] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
  if (c.rc[s] < c.rca) return;
  if (c.lflags[s] & L_GENE_KEYERR) { f.raise(CE_KEY_TX); return; }
  const Set genes = jg_of(c, s);
  if (genes.n != 1) return;
  const uint32_t g = genes.p[0];
  PANTAS_DBG(T_DBG_ANN_J, s);                       // :911
  const bool has_u = same_gene_exons(c, u, g, f);
  if (f.bad()) return;
  const bool has_v = same_gene_exons(c, v, g, f);
  if (f.bad()) return;
  if (!has_u || !has_v) { f.raise(CE_ASSERT_EXONS); return; }
  const Ex exu = ex_of(c, u), exv = ex_of(c, v);
  const uint32_t ob = c.row_ptr[u], oe = c.row_ptr[u + 1];
  const uint32_t ib = c.irow_ptr[v], ie = c.irow_ptr[v + 1];
  PANTAS_SECTION_BEGIN(oe - ob > 500u || ie - ib > 500u);

  // cassette exons (:954-984) are enumerated by cassette_item, one (junction, novel head link) per thread
  if (c.debug && (c.events & EV_ES)) {
    // :957 prints when a novel link leaves u and a novel link enters v (Js1, Js2: sets of (from, to) pairs, so a link
    // listed twice counts once -- its first slot)
    bool out_nov = false, in_nov = false;
    for (uint32_t t = ob; t < oe && !out_nov; ++t) out_nov = c.first_slot[t] == t && t != s && (c.lflags[t] & L_NOVEL);
    for (uint32_t q = ib; q < ie && !in_nov; ++q) { const uint32_t t = c.islot[q]; in_nov = c.first_slot[t] == t && t != s && (c.lflags[t] & L_NOVEL); }
    if (out_nov && in_nov) PANTAS_DBG(T_DBG_NOV_CE, s);
  }

  if (c.events & EV_IR) {
    if (!(c.probe_ok[jidx] & 1u)) { f.raise(CE_VALUE_INT); return; }   // int(u) fails
    const uint32_t* pr = c.probe + uint64_t(jidx) * 6;
    bool left = false, right = false;
    for (uint32_t d = 0; d < 3; ++d) {
      const uint32_t t = pr[d];
      if (t == CNONE || (c.lflags[t] & L_JUNC)) continue;
      if (!(c.lflags[t] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
      left |= c.rc[t] >= c.rca;
    }
    if (!left) return;
    if (!(c.probe_ok[jidx] & 2u)) { f.raise(CE_VALUE_INT); return; }   // int(v) fails
    for (uint32_t d = 3; d < 6; ++d) {
      const uint32_t t = pr[d];
      if (t == CNONE || (c.lflags[t] & L_JUNC)) continue;
      if (!(c.lflags[t] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
      right |= c.rc[t] >= c.rca;
    }
    if (!right) return;
    PANTAS_DBG(T_DBG_NOV_IR2, s);                   // :1002
    // exons of gene g that END at u (no out-neighbour carries them) and START at v
    auto ends_at_u = [&](uint32_t i) {
      if (exu.gene[i] != g) return false;
      for (uint32_t t = ob; t < oe; ++t) if (may_share_exon(c, t) && ex_has(ex_of(c, c.col[t]), exu.tag[i])) return false;
      return true;
    };
    auto starts_at_v = [&](uint32_t k) {
      if (exv.gene[k] != g) return false;
      for (uint32_t q = ib; q < ie; ++q) if (in_slot_may(c, q, I_SHARE) && may_share_exon(c, c.islot[q]) && ex_has(ex_of(c, c.icol[q]), exv.tag[k])) return false;
      return true;
    };
    // Each test walks a whole row of u / v (thousands of links the reads added on an expressed gene): it is made ONCE per exon
    // and kept as a bit -- the pair loops below asked it up to (exons of u) x (exons of v) times, twice over
    PANTAS_SECTION(0);
    uint64_t end_ok = 0, start_ok = 0;
    const bool masks = exu.n <= 64u && exv.n <= 64u;
    if (masks && c.j_carried != nullptr) {
      // (exon i of u ends at u: it is of gene g and no out-neighbour carries it)
      const uint32_t j = c.jrow[s];
      const unsigned long long co = c.j_carried[2u * j], ci = c.j_carried[2u * j + 1u];
      for (uint32_t i = 0; i < exu.n; ++i) if (exu.gene[i] == g && !((co >> i) & 1ull)) end_ok |= 1ull << i;
      for (uint32_t k = 0; k < exv.n; ++k) if (exv.gene[k] == g && !((ci >> k) & 1ull)) start_ok |= 1ull << k;
    } else if (masks) {
      for (uint32_t i = 0; i < exu.n; ++i) if (ends_at_u(i)) end_ok |= 1ull << i;
      if (end_ok)
        for (uint32_t k = 0; k < exv.n; ++k) if (starts_at_v(k)) start_ok |= 1ull << k;
    }
    auto is_end = [&](uint32_t i) { return masks ? ((end_ok >> i) & 1ull) != 0ull : ends_at_u(i); };
    auto is_start = [&](uint32_t k) { return masks ? ((start_ok >> k) & 1ull) != 0ull : starts_at_v(k); };
    PANTAS_SECTION(1);
    uint32_t n_pairs = 0;
    uint32_t gene2 = CNONE;
    Fault gf;
    for (uint32_t i = 0; i < exu.n; ++i) {
      if (!is_end(i)) continue;
      for (uint32_t k = 0; k < exv.n; ++k) {
        if (exu.tx[i] != exv.tx[k] || !is_start(k)) continue;
        ++n_pairs;
        gene_accumulate(c, exu.tx[i], gene2, gf);
      }
    }
    if (n_pairs == 0) return;
    emit.begin(T_IR2_NOV, gf.bad() ? CNONE : gene2, s, CNONE, CNONE);
    emit.task();
    emit.push(n_pairs);
    for (uint32_t i = 0; i < exu.n; ++i) {
      if (!is_end(i)) continue;
      for (uint32_t k = 0; k < exv.n; ++k) {
        if (exu.tx[i] != exv.tx[k] || !is_start(k)) continue;
        emit.push(exu.tag[i]);
        emit.push(exv.tag[k]);
      }
    }
    emit.end(0);
    PANTAS_SECTION(2);
  }
}

// ---- pass 4: cassette exons (:954-984), one work item per (annotated junction, novel link leaving its head) ----
// The rows of one junction are the product (novel links from its head) x (novel links into its tail); on
// expressed genes that is thousands of rows, so the product is spread over threads by head link.
template <class Emit>
PANTAS_HD void cassette_item(const CallView& c, uint32_t s, uint32_t t1, Emit& emit, Fault& f) {
  if (!(c.events & EV_ES)) return;
  const uint32_t u = c.src[s], v = c.col[s];
  // same gate as novel_around (which also reports the errors)
  if (!(c.lflags[s] & L_HAS_RC) || c.rc[s] < c.rca || (c.lflags[s] & L_GENE_KEYERR)) return;
  const Set genes = jg_of(c, s);
  if (genes.n != 1) return;
  const uint32_t g = genes.p[0];
  Fault gate;
  const bool has_u = same_gene_exons(c, u, g, gate);
  const bool has_v = same_gene_exons(c, v, g, gate);
  if (gate.bad() || !has_u || !has_v) return;
  const uint32_t ib = c.irow_ptr[v], ie = c.irow_ptr[v + 1];
  const uint32_t es_rank = c.rank[c.col[t1]];
  for (uint32_t q = ib; q < ie; ++q) {
    if (!in_slot_may(c, q, I_CANON | I_NOVEL)) continue;
    const uint32_t t2 = c.islot[q];
    if (c.first_slot[t2] != t2 || t2 == s || !(c.lflags[t2] & L_NOVEL)) continue;
    if (es_rank <= c.rank[c.src[t2]]) {
      if (!(c.lflags[t1] & L_HAS_RC) || !(c.lflags[t2] & L_HAS_RC)) { f.raise(CE_KEY_RC); return; }
      emit.begin(T_CE_NOV, g, s, t1, t2);
      emit.end(0);
    }
  }
}

// ---- deferred intron walks ---------------------------------------------------------------------------------
// Outcome of one walk task (an IR record whose row depends on a walk through the intron).
struct TaskOut {
  uint32_t np;        // nodes on the path
  uint32_t mean;      // ceil(mean NC) over the path
  uint32_t status;    // 0: no row, 1: row, 2: "Skipping novel IR due to multiple genes"
  uint32_t path_off;  // filled by the scan
};

// A path is a list of words: a segment index, or PATH_RANGE | first followed by last for a run of consecutive indices
// (walk_exon; segment indices stay below 2^31).  TaskOut.np counts SEGMENTS; the words of a path are never more.
constexpr uint32_t PATH_RANGE = 0x80000000u;
struct PathEmit {
  uint32_t* out;
  uint32_t i;
  PANTAS_HD void push(uint32_t w) { out[i++] = w; }
  PANTAS_HD void push_range(uint32_t a, uint32_t b) { if (a == b) out[i++] = a; else { out[i++] = PATH_RANGE | a; out[i++] = b; } }
};
// A walk that writes its path while it decides: room for `cap` segments (an upper bound known before the walk, walk_bound); a path
// that outgrows it is counted to its end and written by a second walk.
struct BoundEmit {
  uint32_t* out;
  uint32_t cap;
  uint32_t n = 0;            // words the path takes
  PANTAS_HD void push(uint32_t w) { if (n < cap) out[n] = w; ++n; }
  PANTAS_HD void push_range(uint32_t a, uint32_t b) { if (a == b) push(a); else { push(PATH_RANGE | a); push(b); } }
};
// Upper bound of a walk's path: it runs from `start` to `goal` through segments whose names sort at or below the goal's, and in a
// graph whose names follow its topological order (every graph vg writes) each step moves up: at most rank(goal) - rank(start) + 1
// segments.  0: no bound offered (the names run the other way, or the span is huge) -- such a walk is measured first, then written.
constexpr uint32_t WALK_BOUND_MAX = 1u << 20;
PANTAS_HD uint32_t walk_bound(const CallView& c, const EventRec& r, const uint32_t* side) {
  uint32_t start, goal;
  if (r.type == T_IR_ANN) { start = side[r.side_off]; goal = side[r.side_off + 1]; }
  else { start = c.src[r.j0]; goal = c.col[r.j0]; }
  const uint32_t a = c.rank[start], b = c.rank[goal];
  if (b < a || b - a >= WALK_BOUND_MAX) return 0;
  return b - a + 1u;
}

// Walks the intron of an IR record.  path == nullptr: decide (status, np, mean); else: also hand every segment of the path to it.
template <class Emit>
PANTAS_HD void run_walk_task(const CallView& c, const EventRec& r, const uint32_t* side, Emit* path, TaskOut& out, Fault& f) {
  out.np = 0;
  out.mean = 0;
  out.status = 0;
  uint32_t np = 0;
  uint64_t ncs = 0, ln = 0;
  bool nc_ok = true;
  const uint32_t u = c.src[r.j0], v = c.col[r.j0];
  if (r.type == T_IR_ANN) {
    const uint32_t n1 = side[r.side_off], n2 = side[r.side_off + 1], pick = side[r.side_off + 2];
    auto ok = [&](uint32_t nn) { return ex_has_flat(c, nn, pick); };
    if (!walk_exon<Emit>(c, n1, n2, ok, path, np, ncs, ln, nc_ok, f)) return;
  } else if (r.type == T_IR1_NOV) {
    const Ex ex0 = ex_of(c, u), ex1 = ex_of(c, v);
    // the tags on BOTH ends, found once and held by the lane (a step then compares the neighbour's tags with them in registers
    // instead of looking each of them up in both lists); more than EndTags holds: the lists are asked as before
    uint32_t both[EndTags::N];
    uint32_t n_both = 0;
    bool held = true;
    for (uint32_t i = 0; i < ex0.n && held; ++i)
      if (ex_has(ex1, ex0.tag[i])) { if (n_both == EndTags::N) held = false; else both[n_both++] = ex0.tag[i]; }
    const EndTags on_both(Ex{both, both, both, held ? n_both : 0u});
    auto ok = [&](uint32_t nn) {
      const Ex en = ex_of(c, nn);
      if (held) return on_both.carried_by(en) != 0ull;
      for (uint32_t i = 0; i < en.n; ++i)
        if (ex_has(ex0, en.tag[i]) && ex_has(ex1, en.tag[i])) return true;
      return false;
    };
    if (!walk_exon<Emit>(c, u, v, ok, path, np, ncs, ln, nc_ok, f)) return;
    if (int64_t(ln) < int64_t(c.minintron)) return;          // sum of LN over the whole path (:853)
  } else {
    if (!walk_greedy<Emit>(c, u, v, path, np, ncs, ln, nc_ok, f)) return;
    if (int64_t(ln) < int64_t(c.minintron)) return;          // inner nodes only (:1052)
  }
  if (r.type != T_IR_ANN) {
    if (r.gene == CNONE) { f.raise(CE_KEY_TX); return; }
    if (r.gene == GENE_MULTI) { out.status = 2; return; }
  }
  if (!nc_ok) { f.raise(CE_KEY_NC); return; }
  out.np = np;
  out.mean = ceil_mean(ncs, np);
  out.status = 1;
}

}  // namespace pantas
