#!/usr/bin/env python3
"""Benchmark of the augment(+call) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scale s1|s2|s3|s4|auto]

A "step" is one pass of the two augment kernels over one batch of tokenised GAF lines that is
already resident in HBM (graph CSR and records are uploaded before the timed region).  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds a replica of the
graph and its own batch of the same size (weak scaling: read shards), and the timed region ends
with the ONE sum over the coverage arrays that a sharded replicate needs: the ranks all-gather a
bitmap of the counter blocks they touched and all-reduce only the union's blocks (pantas_amd/csrc/exchange.hip).
After the timed region rank 0 runs the call kernels on the accumulated coverage (fused path, no
GFA text round trip) and reports events/s beside it.  Rank 0 prints one JSON line.

Workloads are synthetic (pantas_amd/csrc/synth.cpp, SURVEY.md 8-d recipe) at the BASELINE.json
scales: s2 = Drosophila whole genome, s3 = human chr1, s4 = human whole genome (one replicate,
50 M reads); they go through the product's own GFA / GAF / GTF parsers.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

SCALES = {
    # name: (description, synth parameters)
    "s1": ("chr4-like plumbing case", dict(n_genes=114, n_lines=500, base_id=10**6)),
    "s2": ("Drosophila whole-genome scale (exps/1-dm-sim), 1M read pairs",
           dict(n_genes=18000, n_lines=2_000_000, base_id=10**7, intron_long_max=320, n_chrom=6, zipf_a=1.0)),
    "s3": ("human chr1 + 1000G scale, 10M GAF lines",
           dict(n_genes=5500, n_lines=10_000_000, base_id=10**8, intron_long_max=4280, n_chrom=1, zipf_a=1.0,
                read_len=150, with_path=0)),
    "s4": ("human whole-genome scale (exps/3-homo-real layout), one replicate of 50M GAF lines",
           dict(n_genes=62000, n_lines=50_000_000, base_id=10**9, intron_long_max=4780, n_chrom=24, zipf_a=1.0,
                read_len=150, with_path=0)),
}
CPU_SAMPLE = dict(n_genes=1500, n_lines=300_000, base_id=10**7, intron_long_max=320, n_chrom=2, zipf_a=1.0)


def host_ram_gb() -> float:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemTotal:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(sample_lines: int):
    """The oracle (a Python port that keeps the reference's dict-based data structures) on a
    bounded sample: a smaller instance of the same synthetic recipe, written to disk."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import augment_oracle
    import call_oracle
    from pantas_amd.engine import Synth
    tmp = tempfile.mkdtemp(prefix="pantas_cpu_")
    pre = os.path.join(tmp, "sample")
    kw = dict(CPU_SAMPLE)
    kw["n_lines"] = sample_lines
    Synth(seed=11, **kw).write(pre)
    a = augment_oracle.Augmenter(20)
    a.load_gfa(pre + ".gfa")
    t0 = time.perf_counter()
    a.feed_file(pre + ".gaf")
    dt = time.perf_counter() - t0
    with open(pre + ".aug.gfa", "w") as f:
        a.emit(pre + ".gfa", f)
    t1 = time.perf_counter()
    rows = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, rca=3, novel=True)
    dc = time.perf_counter() - t1
    return {"value": sample_lines / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{sample_lines} GAF lines on a {kw['n_genes']}-gene instance of the same synthetic recipe, "
                      f"{dt:.1f} s in oracle/augment_oracle.py (Python port of the reference's per-line loop)",
            "call_events_per_s": len(rows) / dc,
            "call_sample": f"{len(rows)} events in {dc:.1f} s, oracle/call_oracle.py incl. GFA/GTF parsing "
                           "(indexed junction lookup: faster than the reference's O(J^2) scan)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", default=os.environ.get("PANTAS_BENCH_SCALE", "auto"))
    ap.add_argument("--lines", type=int, default=0, help="override the scale's GAF lines per GPU")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-call", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with gloo carrying the device tensors: exercises the sharded path on a "
                         "one-GPU box; the number it prints is not a scaling measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and "PANTAS_THREADS" not in os.environ:
        os.environ["PANTAS_THREADS"] = str(max(4, (os.cpu_count() or 8) // world))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from pantas_amd.engine import Graph, Synth

    scale = args.scale
    if scale == "auto":
        scale = "s4" if host_ram_gb() >= 400 else "s2"
    desc, kw = SCALES[scale]
    kw = dict(kw)
    if args.lines:
        kw["n_lines"] = args.lines
    t_gen = time.perf_counter()
    # every rank: the same graph (same seed), its own block of the read stream
    syn = Synth(seed=args.seed, line_offset=rank * kw["n_lines"], **kw)
    t_gen = time.perf_counter() - t_gen
    g = Graph(text=syn.gfa).to_device(local_rank)
    gs = g.stats
    g.augment_reset()
    batch = g.load_gaf_mem(syn.gaf, first_line=rank * kw["n_lines"])
    bs = batch.stats

    class _Exchange:   # what pantas_amd.distributed.allreduce_counters drives (zero-copy views of the library's buffers)
        def counters_tensor(self): return g.counters_tensor(local_rank)
        def touched_bitmap(self, bw): return g.touched_bitmap(bw, local_rank)
        def counters_pack(self, bm, n): return g.counters_pack(bm, n, local_rank)
        def counters_unpack(self): g.counters_unpack()
    exchange = None

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    warm_ms = []
    for _ in range(args.warmup):
        warm_ms.append(round(g.launch(batch).main_ms, 3))
    if dist is not None:
        # warm-up of the two collectives of the counter exchange on buffers of their sizes (the library sets up channels and
        # protocols on first use); the counters themselves are not touched before the timed region
        dev = torch.device("cuda", local_rank)
        words = g.counters()[1]
        bm = torch.zeros(words // 512 + 1, dtype=torch.int32, device=dev)
        allbm = torch.empty(world * bm.numel(), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allbm, bm)
        st = torch.zeros(max(1 << 20, words // 32), dtype=torch.int32, device=dev)
        dist.all_reduce(st)
        del bm, allbm, st
    sync()
    main_ms, rare_ms, short_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ls = g.launch(batch)      # returns after both kernels completed (HIP events on the launch stream)
        main_ms.append(ls.main_ms)
        rare_ms.append(ls.rare_ms)
        short_ms.append(ls.main_short_ms)
    allreduce_ms = 0.0
    if dist is not None:
        # the ONE sum over the coverage arrays that ends a read-sharded replicate: bitmap of touched blocks ->
        # all-reduce of the union's blocks (dense all-reduce when most blocks are touched)
        from pantas_amd.distributed import allreduce_counters
        t1 = time.perf_counter()
        exchange = allreduce_counters(_Exchange(), dist)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) * 1e3
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    t_fin = time.perf_counter()
    g.finish()   # raises if a kernel flagged a line; downloads the counters
    t_fin = time.perf_counter() - t_fin

    call = None
    if rank == 0 and not args.no_call:
        t1 = time.perf_counter()
        g.call_prepare_fused(text=syn.gtf, device=local_rank)
        t_prep = time.perf_counter() - t1
        fd = os.open(os.devnull, os.O_WRONLY)
        # coverage has accumulated over every launch of the same batch: scale the read-count threshold with it,
        # so that the events are those of `--rca 3` on one pass over the batch
        rca = 3 * (args.steps + args.warmup)
        runs = []
        for _ in range(3):
            t1 = time.perf_counter()
            cs = g.call(fd, -1, rca=rca, novel=True)
            runs.append((time.perf_counter() - t1, cs))
        os.close(fd)
        wall, cs = min(runs, key=lambda x: x[0])
        k_ms = cs.annotated_ms + cs.novel_ms + cs.around_ms
        call = {"events": cs.events, "junctions": cs.junctions, "novel_links": cs.novel_links,
                "evaluated": cs.evaluated, "flags": f"--rca {rca} --novel (= --rca 3 per pass over the batch)",
                "events_per_s": cs.events / wall, "wall_s": wall, "kernel_ms": k_ms,
                "kernel_ms_by_pass": {"annotated": cs.annotated_ms, "novel_links": cs.novel_ms, "novel_around": cs.around_ms},
                "junction_evals_per_s_kernel": cs.evaluated / (k_ms * 1e-3) if k_ms else None,
                "prepare_s": t_prep, "tag_parse_s": cs.parse_seconds, "render_s": cs.render_seconds}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = bs.lines * world / (elapsed / args.steps)
        avg_main = sum(main_ms) / len(main_ms)
        avg_short = sum(short_ms) / len(short_ms)
        if avg_short > 0:
            # two main kernels: the short-walk kernel takes ~93 % of the records and of the time -- the dominant one
            dom_name, dom_ms, dom_bytes = "augment_main_agg_kernel<512,2048,10,10,7,8,packed,short-walk class,info gathers 2 ahead>", avg_short, bs.short_alg_bytes
        else:
            dom_name, dom_ms, dom_bytes = "augment_main_agg_kernel<512,2048,10,10,7,8>", avg_main, bs.alg_bytes
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if scale in tj and not args.lines:
                traffic = tj[scale]["traffic_bytes"]     # measured in a separate --pmc run of this workload
        except (OSError, ValueError):
            pass
        out = collections.OrderedDict()
        out.update({
            "metric": "augment GAF reads/s (+ call events/s) on synthetic spliced pangenome",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{scale}: {desc}; {gs.segments} segments, {gs.link_slots} links, "
                                   f"{kw['n_genes']} genes; {bs.lines} GAF lines per GPU per step "
                                   f"({bs.records} walked, {bs.path_nodes / max(1, bs.records):.2f} nodes/line, "
                                   f"{bs.cs_ops / max(1, bs.records):.2f} cs ops/line), gene expression Zipf a={kw.get('zipf_a', 1.2)}",
                       "parallelism": f"read shards x{world}, graph replicated, one block-sparse all-reduce of the coverage counters per replicate"},
            "kernel_ms": {"augment_main": avg_main, "augment_main_short_walk": avg_short, "augment_main_general": avg_main - avg_short if avg_short else None,
                          "augment_rare": sum(rare_ms) / len(rare_ms),
                          "augment_main_min": min(main_ms), "augment_main_max": max(main_ms),
                          "augment_main_all": [round(x, 3) for x in main_ms],
                          # the very first launch after the upload runs several times longer (cold TLBs / caches over 20 GB of
                          # graph, batch and counters); rocprofv3's per-kernel average includes it, the timed steps do not
                          "augment_main_warmup": warm_ms},
            "allreduce_ms": allreduce_ms, "allreduce": exchange,
            "records_per_s_kernel_only": bs.records / (avg_main * 1e-3),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom_name, "alg_bytes_per_launch": dom_bytes, "records_per_launch": bs.short_records if avg_short > 0 else bs.records,
                         "whole_main_step": {"ms": avg_main, "alg_bytes": bs.alg_bytes, "achieved_GBps": bs.alg_bytes / (avg_main * 1e-3) / 1e9},
                         "traffic_note": "HBM bytes per launch from profiles/r1_traffic.json (separate rocprofv3 --pmc passes "
                                         "of this workload), null when no profile of this scale is committed"},
            "call": call,
            "host": {"generate_s": t_gen, "gfa_load_s": gs.load_seconds, "gaf_tokenize_s": bs.tokenize_seconds,
                     "gaf_upload_s": bs.upload_seconds, "finish_download_s": t_fin,
                     "gaf_text_bytes": bs.text_bytes, "gfa_text_bytes": gs.text_bytes,
                     "tokenize_reads_per_s": bs.lines / bs.tokenize_seconds if bs.tokenize_seconds else None,
                     "cores": os.cpu_count(), "ram_gb": host_ram_gb(),
                     "peak_rss_gb": __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6},
            # the step in front of the path: GAF text -> records; never part of `value`
            "tokenizer": {"where": {1: "host threads (gaf.cpp)", 2: "GPU kernels (tokenize.hip)"}.get(bs.tokenizer),
                          "text_to_hbm_s": bs.upload_seconds if bs.tokenizer == 2 else None,
                          "line_index_s": bs.tok_index_seconds, "measure_and_scans_s": bs.tok_measure_seconds,
                          "write_s": bs.tok_write_seconds, "bucket_s": bs.tok_bucket_seconds,
                          "text_GBps_kernels": (bs.text_bytes / 1e9 / (bs.tok_index_seconds + bs.tok_measure_seconds +
                                                                        bs.tok_write_seconds + bs.tok_bucket_seconds)
                                                if bs.tokenizer == 2 else None),
                          "pcie_inclusive_reads_per_s": bs.lines / (bs.tokenize_seconds + bs.upload_seconds + ms_per_step * 1e-3)},
        })
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
