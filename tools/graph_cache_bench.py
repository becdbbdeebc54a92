#!/usr/bin/env python3
"""GFA text parse vs the binary graph cache on a synthetic graph written to disk.

    python tools/graph_cache_bench.py [--scale s3|s4] [--dir /tmp]
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
    ap.add_argument("--scale", default="s3")
    ap.add_argument("--dir", default="/tmp")
    a = ap.parse_args()
    import bench
    from pantas_amd.engine import Graph, Synth
    desc, kw = bench.SCALES[a.scale]
    kw = dict(kw)
    kw["n_lines"] = 1000
    pre = os.path.join(a.dir, f"pantas_gc_{a.scale}")
    syn = Synth(seed=7, **kw)
    syn.write(pre)
    del syn
    gfa, cache = pre + ".gfa", pre + ".gfa.pgc"
    if os.path.exists(cache):
        os.remove(cache)
    res = {"scale": a.scale, "gfa_bytes": os.path.getsize(gfa), "cpus": os.cpu_count()}
    t0 = time.perf_counter(); g = Graph(gfa); res["parse_s"] = time.perf_counter() - t0
    st = g.stats
    res["segments"], res["links"] = st.segments, st.link_slots
    g.close()
    t0 = time.perf_counter(); g = Graph(gfa, cache=cache); res["parse_and_write_cache_s"] = time.perf_counter() - t0
    assert not g.used_cache
    g.close()
    res["cache_bytes"] = os.path.getsize(cache)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter(); g = Graph(gfa, cache=cache); runs.append(time.perf_counter() - t0)
        assert g.used_cache and g.stats.segments == res["segments"]
        g.close()
    res["load_cached_s"] = min(runs)
    res["speedup"] = res["parse_s"] / res["load_cached_s"]
    for f in (gfa, cache, pre + ".gtf", pre + ".gaf"):
        if os.path.exists(f):
            os.remove(f)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
