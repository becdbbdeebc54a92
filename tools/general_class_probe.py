#!/usr/bin/env python3
"""What the general-walk class of a synthetic batch looks like (path nodes, cs ops, flags), and the general kernel alone."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import SCALES
from pantas_amd.engine import Graph, Synth
scale = sys.argv[1] if len(sys.argv) > 1 else "s3"
desc, kw = SCALES[scale]
syn = Synth(seed=7, **kw)
g = Graph(text=syn.gfa).to_device(0)
g.augment_reset()
b = g.load_gaf_mem(syn.gaf)
bs = b.stats
hdr, node_off, op_off, _nodes, _ops = g.download_batch(b)
n_short = bs.short_records
k = np.diff(node_off.astype(np.int64)); c = np.diff(op_off.astype(np.int64))
flags = hdr[:, 3]
for name, sl in (("short", slice(0, n_short)), ("general", slice(n_short, None))):
    kk, cc, ff = k[sl], c[sl], flags[sl]
    if len(kk) == 0: continue
    print(json.dumps({"class": name, "records": int(len(kk)), "k_mean": float(kk.mean()), "k_p50": int(np.percentile(kk, 50)), "k_p99": int(np.percentile(kk, 99)),
                      "k_max": int(kk.max()), "ops_mean": float(cc.mean()), "ops_max": int(cc.max()),
                      "has_del": float(((ff & 2) != 0).mean()), "slow": float(((ff & 4) != 0).mean()), "simple": float(((ff & 8) != 0).mean()),
                      "k_gt_31": float((kk > 31).mean())}))
for v in ("65", "60", "73"):
    os.environ["PANTAS_MAIN_KERNEL"] = v
    g.augment_reset()
    for _ in range(2): g.launch(b)
    ms = []; sh = []
    for _ in range(6):
        ls = g.launch(b); ms.append(ls.main_ms); sh.append(ls.main_short_ms)
    print(json.dumps({"variant": v, "main_ms": sum(ms)/6, "short_ms": sum(sh)/6, "rest_ms": (sum(ms)-sum(sh))/6}))
g.finish()
