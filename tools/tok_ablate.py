#!/usr/bin/env python3
"""Where the one-parse tokeniser kernel spends its time: the kernel run again and again on one resident batch, every line
leaving after a named stage (PANTAS_TOK_STOP digits: 9 staging only, 1 columns, 2 integers, 3 tag search, 4 cs, 5 dv, 0 all).

    python tools/tok_ablate.py [--scale s3] [--stops 912345 0]

Needs a library built with `make -C pantas_amd/csrc EXPERIMENTS=1` (the shipped build refuses PANTAS_TOK_STOP).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="s3")
    ap.add_argument("--stops", default="9123450")
    a = ap.parse_args()
    from pantas_amd.engine import Graph, Synth
    desc, kw, *_ = bench.SCALES[a.scale]
    syn = Synth(seed=7, **kw)
    g = Graph(text=syn.gfa).to_device(0)
    g.set_tokenizer("device")
    g.augment_reset()
    g.load_gaf_mem(syn.gaf).free()          # warm: pool blocks, code objects
    os.environ["PANTAS_TOK_STOP"] = a.stops
    g.augment_reset()
    b = g.load_gaf_mem(syn.gaf)
    print(f"{desc}: {b.stats.lines} lines, {b.stats.text_bytes / 1e9:.2f} GB of text", file=sys.stderr)


if __name__ == "__main__":
    main()
