// TEST HARNESS ONLY -- not part of the product and never loaded by it.
// The stage kernel's emitter (pantas_amd/csrc/call_host.h StageEmit: an item's own four records, then a chunk of 64, then one of
// 4 096, then "evaluate again") against the write emitter on made-up items of every size around its steps, under ASan/UBSan.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "call_host.h"

using namespace pantas;

// an item of `nrec` records; record k carries (k * 7 + salt) % (max_words + 1) payload words and is a walk task when k % 5 == 2
template <class Emit>
static void item(Emit& e, uint32_t nrec, uint32_t max_words, uint32_t salt) {
  for (uint32_t k = 0; k < nrec; ++k) {
    const bool task = k % 5u == 2u;
    e.begin(task ? T_IR_ANN : T_ES_ANN, k ^ salt, k + 1u, task ? CNONE : k + 2u, k + 3u);
    if (task) e.task();
    const uint32_t nw = max_words ? (k * 7u + salt) % (max_words + 1u) : 0u;
    for (uint32_t w = 0; w < nw; ++w) e.push(k * 1000u + w + salt);
    e.end(k % 3u);
  }
}

int main() {
  const uint32_t MID_N = 2, BIG_N = 1;
  std::vector<EventRec> mid_r(size_t(MID_N) * STAGE_MID_RECS), big_r(size_t(BIG_N) * STAGE_BIG_RECS);
  std::vector<uint32_t> mid_w(size_t(MID_N) * STAGE_MID_WORDS), big_w(size_t(BIG_N) * STAGE_BIG_WORDS);
  size_t where_seen[4] = {0, 0, 0, 0};
  const uint32_t sizes[] = {0, 1, 3, 4, 5, 17, 63, 64, 65, 66, 1000, 4095, 4096, 4097, 6000};
  const uint32_t words[] = {0, 1, 3, 30, 200};        // (few words per record ... records that outgrow the WORD room first)
  for (uint32_t with_pool = 0; with_pool < 2; ++with_pool)
    for (uint32_t nrec : sizes)
      for (uint32_t mw : words)
        for (uint32_t salt = 0; salt < 3; ++salt) {
          uint32_t taken[2] = {salt == 2 ? MID_N : 0u, 0u};      // (salt 2: the small chunks are gone)
          const StagePool pool = with_pool ? StagePool{mid_r.data(), mid_w.data(), MID_N, big_r.data(), big_w.data(), BIG_N, taken} : StagePool{};
          CountEmit ce;
          item(ce, nrec, mw, salt);
          std::vector<EventRec> wr(ce.recs + 1);
          std::vector<uint32_t> ww(ce.words + 1), tl(ce.tasks + 1);
          WriteEmit we{wr.data(), ww.data(), 0, 0, tl.data(), 0};
          item(we, nrec, mw, salt);
          EventRec own_r[STAGE_RECS];
          uint32_t own_w[STAGE_WORDS];
          StageEmit se{own_r, own_w, pool};
          item(se, nrec, mw, salt);
          if (se.recs != ce.recs || se.words != ce.words || se.tasks != ce.tasks) { fprintf(stderr, "counts differ at %u records, %u words\n", nrec, mw); return 1; }
          ++where_seen[se.fits ? se.where : uint8_t(FITS_NO)];
          // what must fit where
          const bool fits_own = ce.recs <= STAGE_RECS && ce.words <= STAGE_WORDS;
          if (fits_own != (se.fits && se.where == FITS_OWN)) { fprintf(stderr, "own area: %u records %u words, fits %d where %u\n", nrec, ce.words, int(se.fits), se.where); return 1; }
          if (!with_pool && !fits_own && se.fits) { fprintf(stderr, "staged without a pool\n"); return 1; }
          if (ce.recs > STAGE_BIG_RECS && se.fits) { fprintf(stderr, "more than the large chunk holds, and staged\n"); return 1; }
          if (!se.fits) continue;
          for (uint32_t k = 0; k < se.recs; ++k) {
            const EventRec &a = se.rec[k], &b = wr[k];
            if (a.type != b.type || a.gene != b.gene || a.j0 != b.j0 || a.j1 != b.j1 || a.j2 != b.j2 || a.side_off != b.side_off || a.side_len != b.side_len || a.mean != b.mean) {
              fprintf(stderr, "record %u of %u differs (where %u)\n", k, nrec, se.where);
              return 1;
            }
          }
          for (uint32_t k = 0; k < se.words; ++k)
            if (se.side[k] != ww[k]) { fprintf(stderr, "word %u of an item of %u records differs (where %u)\n", k, nrec, se.where); return 1; }
        }
  printf("evaluated again %zu, own area %zu, chunk of 64 %zu, chunk of 4096 %zu\n", where_seen[FITS_NO], where_seen[FITS_OWN], where_seen[FITS_MID], where_seen[FITS_BIG]);
  return where_seen[FITS_NO] && where_seen[FITS_OWN] && where_seen[FITS_MID] && where_seen[FITS_BIG] ? 0 : 1;
}
