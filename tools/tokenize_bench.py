#!/usr/bin/env python3
"""GAF text -> records: the GPU tokeniser (tokenize.hip) beside the multi-threaded host tokeniser (gaf.cpp) on one
synthetic GAF held in memory.  Prints one JSON object; phases of the device tokeniser come from the batch stats.

    python tools/tokenize_bench.py [--genes 5500] [--lines 10000000] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=5500)
    ap.add_argument("--lines", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from pantas_amd.engine import Graph, Synth
    syn = Synth(seed=3, n_genes=a.genes, n_lines=a.lines, base_id=10**7)
    g = Graph(text=syn.gfa).to_device(0)
    gaf = syn.gaf
    res = {"lines": a.lines, "text_bytes": gaf[1], "cpus": os.cpu_count(), "runs": {}}
    for where in ("host", "device"):
        g.set_tokenizer(where)
        runs = []
        for _ in range(a.reps):
            g.augment_reset()
            t0 = time.perf_counter()
            b = g.load_gaf_mem(gaf)
            dt = time.perf_counter() - t0
            st = b.stats
            runs.append({"wall_s": dt, "tokenize_s": st.tokenize_seconds, "upload_s": st.upload_seconds, "index_s": st.tok_index_seconds,
                         "measure_s": st.tok_measure_seconds, "write_s": st.tok_write_seconds, "bucket_s": st.tok_bucket_seconds,
                         "records": st.records})
            ls = g.launch(b)
            runs[-1]["main_kernel_ms"] = ls.main_ms
            b.free()
        best = min(runs, key=lambda r: r["wall_s"])
        best["lines_per_s"] = a.lines / best["wall_s"]
        best["text_GBps"] = gaf[1] / best["wall_s"] / 1e9
        if where == "device":
            k = best["index_s"] + best["measure_s"] + best["write_s"] + best["bucket_s"]
            best["kernels_only_s"] = k
            best["kernels_text_GBps"] = gaf[1] / k / 1e9
        res["runs"][where] = best
    res["device_over_host"] = res["runs"]["host"]["wall_s"] / res["runs"]["device"]["wall_s"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
