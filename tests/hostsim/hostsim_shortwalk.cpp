// CPU check of the short walk's per-node summary (walk.h short_node_summary) against the piece-by-piece
// compaction of the general walk (walk.h PieceAcc, the restatement of alignments_augmentation_from_gaf.py:63-107)
// on random cs strings of the short-walk classes laid over random node lengths.  Test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "walk.h"

using namespace pantas;

int main(int argc, char** argv) {
  const uint64_t cases = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000;
  std::mt19937_64 rng(12345);
  auto rnd = [&](uint32_t lo, uint32_t hi) { return lo + uint32_t(rng() % (hi - lo + 1)); };
  uint64_t nodes_checked = 0, dropped = 0, plain_hist[4] = {0, 0, 0, 0};
  for (uint64_t c = 0; c < cases; ++c) {
    // ops: ':' runs, up to 4 '*', zero or one indel, in random order, no two ':' adjacent is not required
    std::vector<uint32_t> ops;
    const uint32_t n_pieces = rnd(1, 7);
    uint32_t stars = 0;
    const bool want_indel = rng() % 3 != 0;
    const uint32_t indel_at = rnd(0, n_pieces - 1);
    for (uint32_t i = 0; i < n_pieces; ++i) {
      if (want_indel && i == indel_at) { ops.push_back((rnd(1, 6) << 3) | (rng() % 2 ? OP_PLUS : OP_MINUS)); continue; }
      if (stars < SIMPLE_MAX_STARS && rng() % 3 == 0) { ops.push_back((1u << 3) | OP_STAR); ++stars; }
      else ops.push_back((rnd(1, 12) << 3) | OP_COLON);
    }
    ShortOps so;
    short_ops_scan(ops.data(), uint32_t(ops.size()), so);
    // node rooms: random small lengths until the ops are used up
    uint32_t P = 0, oi = 0, cur_rem = 0, cur_op = 0;
    bool have = false;
    while (P < so.T) {
      const uint32_t room0 = rnd(1, 9);
      const uint32_t le = std::min(room0, so.T - P);
      // reference: distribute pieces like walk_line
      PieceAcc acc;
      acc.begin(0, 0);
      uint32_t room = room0;
      while (room > 0) {
        if (!have) {
          if (oi >= ops.size()) break;
          cur_op = ops[oi] & 7u; cur_rem = ops[oi] >> 3; have = true;
        }
        uint32_t take;
        const uint32_t op = cur_op;
        if (room <= cur_rem) { take = room; cur_rem -= take; room = 0; if (cur_rem == 0) { have = false; ++oi; } }
        else { take = cur_rem; room -= cur_rem; have = false; ++oi; }
        acc.add(op, take);
        if (!have && oi >= ops.size()) break;
      }
      bool kept; uint32_t plain;
      short_node_summary(so, P, le, kept, plain);
      const bool ref_kept = acc.kept();
      const uint32_t ref_plain = acc.kn.plain;
      if (kept != ref_kept || (kept && plain != ref_plain)) {
        fprintf(stderr, "MISMATCH case %llu P=%u le=%u: kept %d/%d plain %u/%u ops:", (unsigned long long)c, P, le, kept, ref_kept, plain, ref_plain);
        for (uint32_t w : ops) fprintf(stderr, " %c%u", ":*+-=~"[w & 7u], w >> 3);
        fprintf(stderr, "\n");
        return 1;
      }
      ++nodes_checked;
      if (!kept) ++dropped; else ++plain_hist[plain > 3 ? 3 : plain];
      P += le;
    }
  }
  printf("ok %llu nodes, %llu dropped, plain 0/1/2/3: %llu %llu %llu %llu\n", (unsigned long long)nodes_checked, (unsigned long long)dropped,
         (unsigned long long)plain_hist[0], (unsigned long long)plain_hist[1], (unsigned long long)plain_hist[2], (unsigned long long)plain_hist[3]);
  return 0;
}
