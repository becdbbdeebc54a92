#!/usr/bin/env python3
"""Diagnostic (GPU box): how many distinct path nodes does a 2048-record chunk of the bucketed batch touch?  The main
kernel's workgroup-private node table has 1024 slots; a chunk that touches more falls back to memory atomics.

    python tools/chunk_stats.py --scale s4 [--lines N]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="s4")
    ap.add_argument("--lines", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=2048)
    ap.add_argument("--sample", type=int, default=4000, help="chunks looked at (spread evenly)")
    a = ap.parse_args()
    from bench import SCALES
    from pantas_amd.engine import Graph, Synth
    desc, kw, *_ = SCALES[a.scale]
    kw = dict(kw)
    if a.lines:
        kw["n_lines"] = a.lines
    syn = Synth(seed=7, **kw)
    g = Graph(text=syn.gfa).to_device(0)
    g.augment_reset()
    b = g.load_gaf_mem(syn.gaf)
    hdr, node_off, op_off, nodes, ops = g.download_batch(b)
    n = b.stats.short_records
    n_chunks = (n + a.chunk - 1) // a.chunk
    pick = np.unique(np.linspace(0, n_chunks - 1, min(a.sample, n_chunks)).astype(np.int64))
    distinct, visits, links = [], [], []
    for c in pick:
        lo, hi = c * a.chunk, min(n, (c + 1) * a.chunk)
        sl = nodes[node_off[lo]:node_off[hi]]
        distinct.append(np.unique(sl).size)
        visits.append(sl.size)
    distinct = np.array(distinct)
    visits = np.array(visits)
    res = {"scale": a.scale, "records_short_class": int(n), "chunks": int(n_chunks), "sampled": int(pick.size),
           "distinct_nodes_per_chunk": {q: float(np.quantile(distinct, q)) for q in (0.05, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99)},
           "visits_per_chunk_mean": float(visits.mean()),
           "share_of_chunks_over": {str(t): float((distinct > t).mean()) for t in (256, 512, 700, 1024, 2048, 4096)},
           "share_of_visits_in_chunks_over": {str(t): float(visits[distinct > t].sum() / visits.sum()) for t in (256, 512, 700, 1024, 2048, 4096)},
           "reuse_visits_per_distinct_node": {q: float(np.quantile(visits / distinct, q)) for q in (0.05, 0.25, 0.5, 0.75, 0.95)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
