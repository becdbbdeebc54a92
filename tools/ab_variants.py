#!/usr/bin/env python3
"""A/B of main-kernel variants (PANTAS_MAIN_KERNEL) on ONE resident workload, one process.

    python tools/ab_variants.py --scale s4 --variants 60,70,71 --steps 8

Per variant: reset, 2 warm-up launches, `steps` timed launches (HIP events inside the library), then the counters
are hashed -- every variant must leave the same counters and sparse results (same launches, same batch)."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="s3")
    ap.add_argument("--variants", default="60")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--lines", type=int, default=0)
    ap.add_argument("--env", default="", help="extra NAME=VALUE pairs applied to every variant, comma separated")
    ap.add_argument("--del-hot", type=float, default=0.0, help="synthetic deletion skew (bench.py --del-hot)")
    ap.add_argument("--hot-sites", type=int, default=1000)
    ap.add_argument("--no-check", action="store_true", help="timing-only instances among the variants (results wrong by construction): digests are printed, not compared")
    ap.add_argument("--p-indel", type=float, default=-1.0, help="indel rate of the synthetic reads (default: the scale's own, 0.05)")
    args = ap.parse_args()
    from bench import SCALES
    from pantas_amd.engine import Graph, Synth
    import numpy as np
    desc, kw, *_ = SCALES[args.scale]
    kw = dict(kw)
    if args.lines:
        kw["n_lines"] = args.lines
    if args.del_hot:
        kw["p_del_hot"] = args.del_hot
        kw["hot_sites"] = args.hot_sites
    if args.p_indel >= 0:
        kw["p_indel"] = args.p_indel
    syn = Synth(seed=7, **kw)
    g = Graph(text=syn.gfa).to_device(0)
    g.augment_reset()
    batch = g.load_gaf_mem(syn.gaf)
    bs = batch.stats
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    ref = None
    for v in args.variants.split(","):
        # "main" or "main:rare" (PANTAS_MAIN_KERNEL : PANTAS_RARE_KERNEL)
        v, _, rv = v.partition(":")
        rv, _, abl = rv.partition(":")      # "main:rare:ablate" -- PANTAS_ABLATE for this variant (timing only: results wrong by construction)
        if abl:
            os.environ["PANTAS_ABLATE"] = abl
        else:
            os.environ.pop("PANTAS_ABLATE", None)
        os.environ["PANTAS_MAIN_KERNEL"] = v
        if rv:
            os.environ["PANTAS_RARE_KERNEL"] = rv
        else:
            os.environ.pop("PANTAS_RARE_KERNEL", None)
        g.augment_reset()
        for _ in range(2):
            g.launch(batch)
        ms, short, rare = [], [], []
        for _ in range(args.steps):
            ls = g.launch(batch)
            ms.append(ls.main_ms)
            short.append(ls.main_short_ms)
            rare.append(ls.rare_ms)
        try:
            g.finish()
        except Exception as e:      # a timing-only instance may leave a line flagged: its times are what is asked for
            if not args.no_check:
                raise
            print(json.dumps({"variant": int(v), "main_ms": round(sum(ms) / len(ms), 4), "short_ms": round(sum(short) / len(short), 4), "finish": str(e)[:80]}), flush=True)
            continue
        h = hashlib.sha256()
        for a in g.read_counters():
            h.update(np.ascontiguousarray(a).tobytes())
        sp = g.sparse_export().copy()
        sp[:, 2] >>= 32       # first-seen: the line id decides between lines; kernels may differ in the low (within-line) bits
        h.update(np.ascontiguousarray(sp[np.lexsort(sp.T[::-1])]).tobytes())
        dig = h.hexdigest()[:16]
        if abl:
            print(json.dumps({"variant": int(v), "ablate": int(abl), "main_ms": round(sum(ms) / len(ms), 4), "short_ms": round(sum(short) / len(short), 4)}), flush=True)
            continue
        ref = ref or dig
        avg_s = sum(short) / len(short)
        print(json.dumps({"variant": int(v), "rare_variant": rv or "default", "del_hot": args.del_hot, "scale": args.scale, "main_ms": round(sum(ms) / len(ms), 4), "main_min": round(min(ms), 4),
                          "short_ms": round(avg_s, 4), "rare_ms": round(sum(rare) / len(rare), 4), "short_frac_of_8TBps": round(bs.short_alg_bytes / (avg_s * 1e-3) / 8e12, 4) if avg_s else None,
                          "whole_frac": round(bs.alg_bytes / (sum(ms) / len(ms) * 1e-3) / 8e12, 4),
                          "digest": dig, "same_as_first": dig == ref}), flush=True)
        assert args.no_check or dig == ref, "variant changed the result"


if __name__ == "__main__":
    main()
