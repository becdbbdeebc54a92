// TEST HARNESS ONLY -- not part of the product and never loaded by it.
// Runs the call kernels' per-junction source (pantas_amd/csrc/call_device.h) on the CPU over
// the product's own host tables, with the same count -> scan -> write protocol as call.hip,
// under ASan/UBSan.  usage: hostsim_call [call.py options] GFA GTF > events.csv
#include <cstdio>
#include "call_host.h"

using namespace pantas;

struct Out {
  std::vector<EventRec> recs;
  std::vector<uint32_t> side;
  std::vector<TaskOut> tasks;
  std::vector<uint32_t> paths;
};

template <class Fn>
static void run_pass(const CallView& c, const std::vector<uint32_t>& slots, Out& o, Fn fn) {
  std::vector<uint32_t> nrec(slots.size() + 1, 0), nword(slots.size() + 1, 0), ntask(slots.size() + 1, 0);
  for (size_t i = 0; i < slots.size(); ++i) {
    CountEmit e;
    Fault f;
    fn(c, slots[i], uint32_t(i), e, f);
    if (f.bad()) throw Error(call_error_kind(f.code), call_error_text(f.code));
    nrec[i + 1] = nrec[i] + e.recs;
    nword[i + 1] = nword[i] + e.words;
    ntask[i + 1] = ntask[i] + e.tasks;
  }
  const size_t r0 = o.recs.size(), w0 = o.side.size(), t0 = o.tasks.size();
  const bool stage_check = getenv("HOSTSIM_CALL_STAGE_CHECK") != nullptr;
  const uint32_t MID_N = 48, BIG_N = 2;
  std::vector<EventRec> mid_r(stage_check ? size_t(MID_N) * STAGE_MID_RECS : 0), big_r(stage_check ? size_t(BIG_N) * STAGE_BIG_RECS : 0);
  std::vector<uint32_t> mid_w(stage_check ? size_t(MID_N) * STAGE_MID_WORDS : 0), big_w(stage_check ? size_t(BIG_N) * STAGE_BIG_WORDS : 0);
  uint32_t taken[2] = {0, 0};
  const StagePool pool{mid_r.data(), mid_w.data(), MID_N, big_r.data(), big_w.data(), BIG_N, taken};
  size_t staged_items[4] = {0, 0, 0, 0};
  o.recs.resize(r0 + nrec.back());
  o.side.resize(w0 + nword.back());
  std::vector<uint32_t> task_list(ntask.back());
  for (size_t i = 0; i < slots.size(); ++i) {
    // same protocol as call.hip: offsets are pass-local while the kernels run
    WriteEmit e{o.recs.data() + r0, o.side.data() + w0, nrec[i], nword[i], task_list.data(), ntask[i]};
    Fault f;
    fn(c, slots[i], uint32_t(i), e, f);
    if (e.rec_i != nrec[i + 1] || e.word_i != nword[i + 1] || e.task_i != ntask[i + 1]) throw Error("InternalError", "count/write mismatch");
    if (stage_check) {
      // HOSTSIM_CALL_STAGE_CHECK=1: the item once more through the stage kernel's emitter (call_host.h StageEmit: its own four
      // records, then chunks of a pool -- a small one here, so that some items find none): what it staged is what was written
      EventRec own_r[STAGE_RECS];
      uint32_t own_w[STAGE_WORDS];
      StageEmit se{own_r, own_w, pool};
      Fault f2;
      fn(c, slots[i], uint32_t(i), se, f2);
      if (se.recs != nrec[i + 1] - nrec[i] || se.words != nword[i + 1] - nword[i] || se.tasks != ntask[i + 1] - ntask[i]) throw Error("InternalError", "stage: counts differ");
      if (se.fits) {
        ++staged_items[se.where];
        for (uint32_t k = 0; k < se.recs; ++k) {
          const EventRec& a = se.rec[k];
          const EventRec& b = o.recs[r0 + nrec[i] + k];
          const bool task = rec_is_task(a.type);
          if (a.type != b.type || a.gene != b.gene || a.j0 != b.j0 || a.j2 != b.j2 || a.side_len != b.side_len || a.mean != b.mean ||
              a.side_off + nword[i] != b.side_off || (task ? a.j1 + ntask[i] != b.j1 : a.j1 != b.j1))
            throw Error("InternalError", "stage: a staged record differs from the written one");
        }
        for (uint32_t k = 0; k < se.words; ++k)
          if (se.side[k] != o.side[w0 + nword[i] + k]) throw Error("InternalError", "stage: a staged payload word differs");
      } else ++staged_items[FITS_NO];
    }
  }
  if (stage_check) fprintf(stderr, "[hostsim stage check] %zu items: %zu evaluated again, %zu in their own area, %zu in a chunk of 64 records, %zu in a chunk of 4096\n", slots.size(),
                           staged_items[FITS_NO], staged_items[FITS_OWN], staged_items[FITS_MID], staged_items[FITS_BIG]);
  // deferred walks: decide + measure, then write at the scanned offsets
  o.tasks.resize(t0 + task_list.size());
  std::vector<uint32_t> poff(task_list.size() + 1, 0);
  for (size_t k = 0; k < task_list.size(); ++k) {
    Fault f;
    TaskOut to;
    run_walk_task<PathEmit>(c, o.recs[r0 + task_list[k]], o.side.data() + w0, nullptr, to, f);
    if (f.bad()) throw Error(call_error_kind(f.code), call_error_text(f.code));
    poff[k + 1] = poff[k] + (to.status == 1 ? to.np : 0);
    o.tasks[t0 + k] = to;
  }
  const size_t p0 = o.paths.size();
  o.paths.resize(p0 + poff.back());
  for (size_t k = 0; k < task_list.size(); ++k) {
    TaskOut& to = o.tasks[t0 + k];
    to.path_off = uint32_t(p0 + poff[k]);
    if (to.status != 1) continue;
    Fault f;
    TaskOut again;
    PathEmit pe{o.paths.data() + p0, poff[k]};
    run_walk_task(c, o.recs[r0 + task_list[k]], o.side.data() + w0, &pe, again, f);
    if (pe.i > poff[k + 1] || again.np != to.np) throw Error("InternalError", "walk count/write mismatch");     // (np words of room: a run takes two)
  }
  for (size_t i = r0; i < o.recs.size(); ++i) {
    EventRec& r = o.recs[i];
    r.side_off += uint32_t(w0);
    if (r.type == T_IR_ANN || r.type == T_IR1_NOV || r.type == T_IR2_NOV) r.j1 += uint32_t(t0);
  }
}

int main(int argc, char** argv) {
  CallOptions o;
  bool no_header = false;
  std::vector<const char*> pos;
  try {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a == "--debug") o.debug = true;
      else if (a == "--rca") o.rca = atoi(argv[++i]);
      else if (a == "--isw") o.isw = atoi(argv[++i]);
      else if (a == "--minintronsize") o.minintron = atoi(argv[++i]);
      else if (a == "--novel") o.novel = true;
      else if (a == "--no-annotated") o.no_annotated = true;
      else if (a == "--junction") { o.has_junction = true; o.junction = argv[++i]; }
      else if (a == "--shard") { o.shard_rank = uint32_t(atoi(argv[++i])); o.shard_world = uint32_t(atoi(argv[++i])); }
      else if (a == "--no-header") no_header = true;
      else if (a == "--events") {
        o.events = 0;
        while (i + 1 < argc && argv[i + 1][0] != '-' && (std::string(argv[i + 1]) == "ES" || std::string(argv[i + 1]) == "SS" || std::string(argv[i + 1]) == "IR")) {
          std::string e = argv[++i];
          o.events |= e == "ES" ? EV_ES : e == "SS" ? EV_SS : EV_IR;
        }
      } else pos.push_back(argv[i]);
    }
    if (pos.size() != 2) { fprintf(stderr, "usage: hostsim_call [options] GFA GTF\n"); return 2; }
    auto g = load_graph(pos[0], 3);
    CallTables t;
    build_call_tables(*g, pos[1], t, 3);
    CallView c = host_view(*g, t, o);
    // the intron walks' runs (CallView::run_end_rev / pl / pnc): the product makes them on the device (call.hip build_walk_runs);
    // the same three arrays by plain loops, so that the walks of this harness jump run by run like the kernels' (HOSTSIM_CALL_NO_RUNS=1:
    // segment by segment -- tests hold the two to the same rows)
    std::vector<uint32_t> run_end_rev;
    std::vector<uint64_t> pl, pnc;
    if (!getenv("HOSTSIM_CALL_NO_RUNS") && c.V) {
      bool all_nc = true;
      for (uint32_t n = 0; n < c.V; ++n) all_nc = all_nc && (c.nflags[n] & N_HAS_NC);
      if (all_nc) {
        run_end_rev.assign(c.V, 0);
        pl.assign(size_t(c.V) + 1, 0);
        pnc.assign(size_t(c.V) + 1, 0);
        uint32_t brk = c.V - 1;
        for (uint32_t n = c.V; n-- > 0;) {
          if (c.nxt1[n] != n + 1) brk = n;
          run_end_rev[c.V - 1 - n] = brk;
        }
        for (uint32_t n = 0; n < c.V; ++n) { pl[n + 1] = pl[n] + c.ln[n]; pnc[n + 1] = pnc[n] + uint64_t(uint32_t(c.nc[n])); }
        c.run_end_rev = run_end_rev.data(); c.pl = pl.data(); c.pnc = pnc.data();
        c.rank_monotone = 1;
        for (uint32_t n = 0; n + 1 < c.V; ++n) if (c.rank[n + 1] <= c.rank[n]) c.rank_monotone = 0;
      }
    }
    // ... and the shared-exon bit per link slot (CallView::share_ex; HOSTSIM_CALL_NO_SHARE_BITS=1: every neighbour's tags are read)
    std::vector<uint32_t> share((t.A + 31) / 32 + 1, 0u);
    if (!getenv("HOSTSIM_CALL_NO_SHARE_BITS")) {
      for (size_t sl = 0; sl < t.A; ++sl)
        if (ends_share_exon(c.ex_ptr, c.ex_tag, c.src[sl], c.col[sl])) share[sl >> 5] |= 1u << (sl & 31);
      c.share_ex = share.data();
    }
    std::vector<uint8_t> iflags(t.A + 1, 0);
    if (!getenv("HOSTSIM_CALL_NO_SHARE_BITS")) {
      for (size_t q = 0; q < t.A; ++q) {
        const uint32_t sl = c.islot[q], fs = c.first_slot[sl], fl = c.lflags[fs];
        iflags[q] = uint8_t((fs == sl ? I_CANON : 0u) | ((fl & L_JUNC) ? I_JUNC : 0u) | ((fl & L_NOVEL) ? I_NOVEL : 0u) | (((share[sl >> 5] >> (sl & 31)) & 1u) ? I_SHARE : 0u));
      }
      c.iflags = iflags.data();
    }
    // ... the junction slots of every out-row and in-row (CallView::oj / ij), by the filter they replace
    std::vector<uint32_t> oj_ptr(size_t(c.V) + 1, 0u), ij_ptr(size_t(c.V) + 1, 0u), oj, ij;
    if (!getenv("HOSTSIM_CALL_NO_SHARE_BITS")) {
      for (uint32_t n = 0; n < c.V; ++n) {
        for (uint32_t s2 = c.row_ptr[n]; s2 < c.row_ptr[n + 1]; ++s2)
          if (c.lflags[c.first_slot[s2]] & L_JUNC) oj.push_back(s2);
        oj_ptr[n + 1] = uint32_t(oj.size());
        for (uint32_t q = c.irow_ptr[n]; q < c.irow_ptr[n + 1]; ++q)
          if (c.lflags[c.first_slot[c.islot[q]]] & L_JUNC) ij.push_back(q);
        ij_ptr[n + 1] = uint32_t(ij.size());
      }
      oj.push_back(0); ij.push_back(0);       // (data() of an empty vector may be null: the view tests the pointers)
      c.oj_ptr = oj_ptr.data(); c.oj = oj.data(); c.ij_ptr = ij_ptr.data(); c.ij = ij.data();
    }
    // ... and the per-junction lists and masks (CallView::j_ht_n / j_ht / j_carried), by the loops they replace
    const size_t nJ = t.junctions.size();
    std::vector<uint32_t> j_ht_n(2 * nJ + 1, 0u), j_ht(2 * J_HT_MAX * nJ + 1, CNONE);
    std::vector<unsigned long long> j_carried(2 * nJ + 1, 0ull);
    if (!getenv("HOSTSIM_CALL_NO_SHARE_BITS")) {
      for (size_t w = 0; w < nJ; ++w) {
        const uint32_t sl = t.junctions[w], j = c.jrow[sl], u = c.src[sl], v = c.col[sl];
        const Set genes = jg_of(c, sl);
        const bool one = genes.n == 1;
        const uint32_t g = one ? genes.p[0] : CNONE;
        for (uint32_t side = 0; side < 2; ++side) {
          const uint32_t b = side == 0 ? c.row_ptr[u] : c.irow_ptr[v], e = side == 0 ? c.row_ptr[u + 1] : c.irow_ptr[v + 1];
          uint32_t n = 0;
          bool keyerr = false;
          for (uint32_t q = b; q < e; ++q) {
            const uint32_t tt = side == 0 ? q : c.islot[q];
            if (c.first_slot[tt] != tt || tt == sl || !(c.lflags[tt] & L_JUNC)) continue;
            keyerr = keyerr || (c.lflags[tt] & L_GENE_KEYERR);
            if (one && set_has(jg_of(c, tt), g)) { if (n < J_HT_MAX) j_ht[2 * J_HT_MAX * j + side * J_HT_MAX + n] = tt; ++n; }
          }
          j_ht_n[2 * j + side] = n | (keyerr ? J_KEYERR : 0u);
          const Ex mine = ex_of(c, side == 0 ? u : v);
          unsigned long long m = 0;
          if (mine.n && mine.n <= 64)
            for (uint32_t q = b; q < e; ++q) {
              const Ex en = ex_of(c, side == 0 ? c.col[q] : c.icol[q]);
              for (uint32_t i = 0; i < mine.n; ++i) if (ex_has(en, mine.tag[i])) m |= 1ull << i;
            }
          j_carried[2 * j + side] = m;
        }
      }
      c.j_ht_n = j_ht_n.data(); c.j_ht = j_ht.data(); c.j_carried = j_carried.data();
    }
    std::vector<uint32_t> arena(size_t(64) * 4096);
    uint32_t arena_next = 0;
    c.arena = arena.data(); c.arena_next = &arena_next; c.arena_slots = 64; c.arena_cap = 4096;
    std::vector<uint32_t> ann, nov, around, around_j;
    pass_lists(*g, t, o, ann, nov, around, around_j);
    Out out;
    if (!no_header) printf("%s\n", CALL_CSV_HEADER);
    run_pass(c, ann, out, [](const CallView& c, uint32_t s, uint32_t, auto& e, Fault& f) { annotated_junction(c, s, e, f); });
    run_pass(c, nov, out, [](const CallView& c, uint32_t s, uint32_t, auto& e, Fault& f) { novel_link(c, s, e, f); });
    run_pass(c, around, out, [&](const CallView& c, uint32_t s, uint32_t i, auto& e, Fault& f) { novel_around(c, s, around_j[i], e, f); });
    if (o.novel && (o.events & EV_ES)) {
      std::vector<uint32_t> cj, ch;
      cassette_items(t, around, cj, ch);
      run_pass(c, cj, out, [&](const CallView& c, uint32_t s, uint32_t i, auto& e, Fault& f) { cassette_item(c, s, ch[i], e, f); });
    }
    std::string csv, err;
    render_events(*g, t, out.recs.data(), out.recs.size(), out.side.data(), out.tasks.data(), out.paths.data(), csv, err);
    fwrite(csv.data(), 1, csv.size(), stdout);
    fwrite(err.data(), 1, err.size(), stderr);
    return 0;
  } catch (const Error& e) {
    fprintf(stderr, "%s\n", e.what());
    return 1;
  }
}
