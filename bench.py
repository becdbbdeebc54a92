#!/usr/bin/env python3
"""Benchmark of the augment(+call) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the two augment kernels over one batch of tokenised GAF lines
that is already resident in HBM (graph CSR and records uploaded before the timed region).
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds a replica
of the graph and its own batch of the same size (weak scaling: read shards), and the timed
region ends with the ONE all-reduce over the coverage arrays that a sharded replicate needs
(SURVEY.md 8-e).  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def make_workload(args, rank, tmp):
    """Synthetic inputs (pantas_amd.synth recipe).  Returns paths + description."""
    from pantas_amd import synth
    pre = os.path.join(tmp, f"w{rank}")
    m = synth.make_model(seed=args.seed, n_genes=args.genes)
    synth.write_gfa(m, pre + ".gfa")
    synth.write_gtf(m, pre + ".gtf")
    synth.write_gaf(m, pre + ".gaf", seed=args.seed + 1000 * rank, n_lines=args.lines)
    return pre


def cpu_baseline(pre, sample_lines):
    """The oracle (a Python port with the reference's data structures) on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import augment_oracle
    a = augment_oracle.Augmenter(20)
    a.load_gfa(pre + ".gfa")
    n = 0
    t0 = time.perf_counter()
    with open(pre + ".gaf") as f:
        for line in f:
            a.feed(line, n + 1)
            n += 1
            if n >= sample_lines:
                break
    dt = time.perf_counter() - t0
    return n / dt, n, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--genes", type=int, default=int(os.environ.get("PANTAS_BENCH_GENES", 400)))
    ap.add_argument("--lines", type=int, default=int(os.environ.get("PANTAS_BENCH_LINES", 200000)))
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cpu-sample", type=int, default=200000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from pantas_amd.engine import Graph

    tmp = tempfile.mkdtemp(prefix="pantas_bench_")
    pre = make_workload(args, rank, tmp)
    g = Graph(pre + ".gfa").to_device(local_rank)
    gs = g.stats
    g.augment_reset()
    batch = g.load_gaf(pre + ".gaf")
    bs = batch.stats

    counters_t = None
    if world > 1:
        ptr, words = g.counters()

        class _Dev:  # zero-copy view of the library's counter allocation for RCCL
            __cuda_array_interface__ = {"shape": (words,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        counters_t = torch.as_tensor(_Dev(), device=torch.device("cuda", local_rank))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        g.launch(batch)
    sync()
    main_ms, rare_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ls = g.launch(batch)      # returns after both kernels completed (HIP events on the launch stream)
        main_ms.append(ls.main_ms)
        rare_ms.append(ls.rare_ms)
    t_kernels = time.perf_counter() - t0
    allreduce_ms = 0.0
    if counters_t is not None:
        t1 = time.perf_counter()
        dist.all_reduce(counters_t)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) * 1e3
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    g.finish()   # raises if a kernel flagged a line

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        lines_per_step = bs.lines * world
        value = lines_per_step / (elapsed / args.steps)
        avg_main = sum(main_ms) / len(main_ms)
        achieved = bs.alg_bytes / (avg_main * 1e-3) / 1e9
        out = {
            "metric": "augment GAF reads/s (+ call events/s) on synthetic spliced pangenome",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"synthetic gene-model graph: {gs.segments} segments, {gs.link_slots} links, "
                                   f"{args.genes} genes; {bs.lines} GAF lines per GPU per step "
                                   f"({bs.records} walked, {bs.path_nodes / max(1, bs.records):.2f} nodes/line, "
                                   f"{bs.cs_ops / max(1, bs.records):.2f} cs ops/line)",
                       "parallelism": f"read shards x{world}, graph replicated, one all-reduce per run"},
            "kernel_ms": {"augment_main": avg_main, "augment_rare": sum(rare_ms) / len(rare_ms)},
            "allreduce_ms": allreduce_ms,
            "records_per_s_kernel_only": bs.records / (avg_main * 1e-3),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "augment_main_kernel", "alg_bytes_per_launch": bs.alg_bytes},
            "host": {"tokenize_s": bs.tokenize_seconds, "upload_s": bs.upload_seconds,
                     "gfa_load_s": gs.load_seconds, "gaf_text_bytes": bs.text_bytes},
        }
        if not args.no_cpu:
            rate, n, dt = cpu_baseline(pre, args.cpu_sample)
            out["cpu_baseline"] = {"value": rate, "unit": "reads/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n} GAF lines of the same workload, {dt:.1f} s, "
                                             "oracle/augment_oracle.py (Python port of the reference loop)"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
