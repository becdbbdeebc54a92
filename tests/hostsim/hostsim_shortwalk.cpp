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
  // ---- round 4: the tile geometry of the position-parallel walk (walk.h tile_*) against plain loops over random tiles
  uint64_t tiles_checked = 0;
  for (uint64_t c = 0; c < cases / 20; ++c) {
    // records of random lengths packed greedily (tile_pack), as the batch builder does for a chunk
    std::vector<uint32_t> ks;
    for (uint32_t i = 0, n = rnd(1, 60); i < n; ++i) ks.push_back(rng() % 8 == 0 ? rnd(20, 31) : rnd(1, 9));
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> tiles;      // per tile: (first lane, k) of its records
    uint32_t fill = 0;
    for (size_t i = 0; i < ks.size(); ++i) {
      if (tile_pack(fill, ks[i], i == 0)) tiles.emplace_back();
      if (fill > TILE_LANES || fill < ks[i]) { fprintf(stderr, "tile_pack: fill %u after a record of %u\n", fill, ks[i]); return 1; }
      tiles.back().emplace_back(fill - ks[i], ks[i]);
    }
    for (size_t t = 0; t + 1 < tiles.size(); ++t) {      // greedy: the next tile's first record did not fit
      const auto& last = tiles[t].back();
      if (last.first + last.second + tiles[t + 1][0].second <= TILE_LANES) { fprintf(stderr, "tile_pack: a tile was closed early\n"); return 1; }
    }
    for (const auto& tile : tiles) {
      uint64_t heads = 0;
      uint32_t used = 0;
      int seg_of[64];
      for (int& x : seg_of) x = -1;
      for (size_t r = 0; r < tile.size(); ++r) {
        heads |= 1ull << tile[r].first;
        for (uint32_t j = 0; j < tile[r].second; ++j) seg_of[tile[r].first + j] = int(r);
        used = tile[r].first + tile[r].second;
      }
      const uint64_t kept = rng() & rng() & (used == 64 ? ~0ull : ((1ull << used) - 1ull)) | (rng() % 4 == 0 ? 0ull : (rng() & ((used == 64) ? ~0ull : ((1ull << used) - 1ull))));
      for (uint32_t l = 0; l < 64; ++l) {
        const uint64_t le = tile_lanes_le(l);
        if (l >= used) continue;
        const auto& rec = tile[size_t(seg_of[l])];
        const uint32_t meta = pos_meta(l - rec.first, rec.second, uint32_t(seg_of[l]));
        if (meta >= POS_INACTIVE || pos_ordinal(meta) != uint32_t(seg_of[l])) { fprintf(stderr, "pos_meta lane %u: %x\n", l, meta); return 1; }
        const TileLane tl = tile_lane(meta, l);
        if (tl.seg_start != rec.first || tl.seg_end != rec.first + rec.second - 1u) { fprintf(stderr, "tile_lane lane %u: [%u, %u]\n", l, tl.seg_start, tl.seg_end); return 1; }
        if (!((kept >> l) & 1u)) continue;
        for (int rev = 0; rev < 2; ++rev) {
          const TileNeighbour nb = tile_neighbour(kept, tl, l, le, rev != 0);
          int prev = -1, next = -1;
          uint32_t q = 0;
          for (uint32_t j = tl.seg_start; j <= tl.seg_end; ++j) {
            if (!((kept >> j) & 1u)) continue;
            if (j < l) { prev = int(j); ++q; }
            if (j > l && next < 0) next = int(j);
          }
          const int other = rev ? prev : next;
          if (nb.is_first != (prev < 0) || nb.is_last != (next < 0) || nb.has != (other >= 0) || (other >= 0 && int(nb.lane) != other) || nb.lane > 63u ||
              tile_kept_index(kept, tl, le) != q) {
            fprintf(stderr, "tile_neighbour lane %u rev %d: first %d last %d has %d lane %u q %u; want prev %d next %d q %u\n", l, rev, nb.is_first, nb.is_last,
                    nb.has, nb.lane, tile_kept_index(kept, tl, le), prev, next, q);
            return 1;
          }
        }
      }
      ++tiles_checked;
    }
  }
  printf("ok %llu tiles\n", (unsigned long long)tiles_checked);
  printf("ok %llu nodes, %llu dropped, plain 0/1/2/3: %llu %llu %llu %llu\n", (unsigned long long)nodes_checked, (unsigned long long)dropped,
         (unsigned long long)plain_hist[0], (unsigned long long)plain_hist[1], (unsigned long long)plain_hist[2], (unsigned long long)plain_hist[3]);
  return 0;
}
