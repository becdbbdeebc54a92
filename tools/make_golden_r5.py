#!/usr/bin/env python3
"""Round-5 goldens, written by the REAL reference scripts in the build container (they never travel).

    python tools/make_golden_r5.py [--ref /root/reference] [--seeds 32] [--jobs 8] [--only tags|mid]

1. tests/golden/tags/cases.json -- IL / OL tag values at the edge of `build_attrs` (scripts/call.py:25-69: the position
   clustering nobody reads can still raise IndexError / ZeroDivisionError / ValueError).  Every case is the reference-augmented
   micro graph (tests/golden/micro/expected.aug.gfa) with ONE tag of ONE line replaced or added; the file is rebuilt from
   {line prefix, tag name, value} by the tests (data, not code).  Recorded: return code, exception class, and that a run that
   succeeds prints the unmodified graph's rows.

2. tests/golden/mid/ -- a mid-size workload (700 genes, 120 k GAF lines, bubbles and novel links turned up) from the frozen
   generator pantas_amd/synth.py: inputs, the reference's augmented GFA, and for two `call` flag sets the rows of >= 32
   PYTHONHASHSEEDs as {normalised row: number of seeds that printed it (with multiplicity)} -- the reference iterates Python sets
   of strings, 1 % of its rows depend on the seed.  xz-compressed.
"""
from __future__ import annotations

import argparse
import collections
import concurrent.futures as cf
import json
import lzma
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import call_oracle  # noqa: E402
import make_golden as mg  # noqa: E402
from pantas_amd import synth  # noqa: E402

# (line prefix in the augmented micro graph, tag name, value, what the engine does: "same" | "refuses")
TAG_CASES = [
    # one entry: only max(_v, key=x[1]) looks at it [:66]
    ("S\t1003\t", "IL", "20", "same"),
    ("S\t1003\t", "OL", "20", "same"),
    ("S\t1003\t", "IL", "0.0", "same"),
    ("S\t1003\t", "IL", "0.1.2", "same"),
    ("S\t1003\t", "IL", "-3.1", "same"),
    ("S\t1003\t", "IL", "+3.1", "same"),
    ("S\t1003\t", "IL", "1_0.1", "same"),
    ("S\t1003\t", "IL", "7.-2", "same"),
    # one cluster (positions closer than 3) [:43-45]
    ("S\t1003\t", "IL", "0.0,2.0", "same"),
    ("S\t1003\t", "IL", "0.1,2.-1", "same"),
    ("S\t1003\t", "IL", "3.0,3.0", "same"),
    ("S\t1003\t", "IL", "5,5.1", "same"),
    ("S\t1003\t", "IL", "5.1,5", "same"),
    ("S\t1003\t", "IL", "0.2,2.3", "same"),
    ("S\t1003\t", "IL", "0.2,1.0,2.3", "same"),
    # two clusters [:47-61]
    ("S\t1003\t", "IL", "0.0,10.0", "same"),
    ("S\t1003\t", "IL", "0.1,10.0", "same"),
    ("S\t1003\t", "IL", "0.0,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,10", "same"),
    ("S\t1003\t", "IL", "0,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,5.2,10", "same"),
    ("S\t1003\t", "IL", "0.1,0,10.2", "same"),
    ("S\t1003\t", "IL", "0,0,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,0.1.5,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,5.-1,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,4.-1,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,10.1,10.5", "same"),
    ("S\t1003\t", "IL", "0.1,10.1,10.-1", "same"),
    ("S\t1003\t", "IL", "0.1,0.1,9,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,10.1,0.1,10.1", "same"),
    ("S\t1003\t", "IL", "0.1,10.1,0.-1,10.-1", "same"),
    ("S\t1003\t", "IL", "0.1,10.1,0.1.7", "same"),
    ("S\t1003\t", "IL", "0.0,10.1,0.0.7", "same"),
    ("S\t1003\t", "IL", "10.1,0.1,3", "same"),
    ("S\t1003\t", "OL", "30.2,0", "same"),
    # not ints
    ("S\t1003\t", "IL", "a.1", "same"),
    ("S\t1003\t", "IL", "", "same"),
    ("S\t1003\t", "IL", "1..2", "same"),
    ("S\t1003\t", "IL", "1.2,", "same"),
    ("S\t1003\t", "IL", "1.2,,3.4", "same"),
    ("S\t1003\t", "IL", "1.2.", "same"),
    ("S\t1003\t", "IL", "0x1.2", "same"),
    ("S\t1003\t", "IL", "1.5e3", "same"),
    # L lines go through the same build_attrs [:166-172]
    ("L\t1002\t+\t1003\t", "IL", "20", "same"),
    ("L\t1002\t+\t1004\t", "OL", "0.0,2.0", "same"),
    ("L\t1002\t+\t1004\t", "OL", "0.1,2.1", "same"),
    # beyond the engine's 2^60: Python ints are unbounded
    ("S\t1003\t", "IL", "99999999999999999999.1", "refuses"),
]
TAG_FLAGS = ["--rca", "0", "--novel"]


def with_tag(lines, prefix, name, value):
    """The line that starts with `prefix`: its `name` tag replaced by name:Z:value, or that tag appended."""
    out = []
    hit = 0
    for l in lines:
        if l.startswith(prefix):
            hit += 1
            tok = l.split("\t")
            tok = [t for t in tok if not t.startswith(name + ":")] + [f"{name}:Z:{value}"]
            l = "\t".join(tok)
        out.append(l)
    assert hit == 1, (prefix, hit)
    return out


def make_tags(ref, out):
    base = open(os.path.join(ROOT, "tests", "golden", "micro", "expected.aug.gfa")).read().splitlines()
    gtf = os.path.join(ROOT, "tests", "golden", "micro", "in.gtf")
    os.makedirs(out, exist_ok=True)
    r = mg.run_ref(ref, "call.py", [*TAG_FLAGS, os.path.join(ROOT, "tests", "golden", "micro", "expected.aug.gfa"), gtf], seed=0)
    assert r.returncode == 0
    base_rows = sorted(call_oracle.normalise_row(x) for x in r.stdout.splitlines()[1:])
    meta = {"base": "micro/expected.aug.gfa", "gtf": "micro/in.gtf", "flags": TAG_FLAGS, "rows": base_rows, "cases": []}
    with tempfile.TemporaryDirectory() as td:
        for prefix, name, value, engine in TAG_CASES:
            p = os.path.join(td, "g.gfa")
            mg.write_lines(p, with_tag(base, prefix, name, value))
            r = mg.run_ref(ref, "call.py", [*TAG_FLAGS, p, gtf], seed=0)
            exc = r.stderr.strip().splitlines()[-1].split(":")[0].strip() if r.returncode else None
            if r.returncode == 0:
                rows = sorted(call_oracle.normalise_row(x) for x in r.stdout.splitlines()[1:])
                assert rows == base_rows, (name, value)
            meta["cases"].append({"line": prefix, "tag": name, "value": value, "rc": r.returncode, "exception": exc, "engine": engine})
            print(f"  {prefix.strip()!r:24} {name}:Z:{value:28} -> {exc or 'ok'}")
    json.dump(meta, open(os.path.join(out, "cases.json"), "w"), indent=1)


MID = dict(seed=11, n_genes=700, n_lines=120000, bubble_rate=0.05, p_novel_skip=0.04, p_novel_ce=0.02, p_intron=0.05, p_novel_ss=0.04)
MID_FLAGS = [["--rca", "3", "--novel"], ["--rca", "0", "--novel", "--minintronsize", "20", "--isw", "3"]]


def _one_call(job):
    ref, flags, aug, gtf, seed = job
    r = mg.run_ref(ref, "call.py", [*flags, aug, gtf], seed=seed)
    assert r.returncode == 0, r.stderr[-500:]
    rows = r.stdout.splitlines()
    assert rows[0] == call_oracle.HEADER
    return collections.Counter(call_oracle.normalise_row(x) for x in rows[1:])


def xz(src, dst):
    with open(src, "rb") as f, lzma.open(dst, "wb", preset=9) as g:
        shutil.copyfileobj(f, g)


def make_mid(ref, out, nseeds, jobs):
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        synth.write_all(f"{td}/in", **MID)
        r = mg.run_ref(ref, "alignments_augmentation_from_gaf.py", [f"{td}/in.gaf", f"{td}/in.gfa"])
        assert r.returncode == 0, r.stderr
        open(f"{td}/aug.gfa", "w").write(r.stdout)
        rej = int([l for l in r.stderr.splitlines() if l.startswith("Rejected alignments:")][0].split(":")[1])
        for n in ("in.gfa", "in.gaf", "in.gtf"):
            xz(f"{td}/{n}", f"{out}/{n}.xz")
        xz(f"{td}/aug.gfa", f"{out}/expected.aug.gfa.xz")
        meta = {"generator": "pantas_amd.synth.write_all", "params": MID, "rejected": rej, "seeds": nseeds, "calls": []}
        for k, flags in enumerate(MID_FLAGS):
            with cf.ProcessPoolExecutor(jobs) as ex:
                runs = list(ex.map(_one_call, [(ref, flags, f"{td}/aug.gfa", f"{td}/in.gtf", s) for s in range(nseeds)]))
            # per row: in how many seeds it was printed (a row printed twice by one seed counts twice there: the multiset of
            # every seed must be reproducible from {row: [count under seed 0, ...]} -- stored compactly as the stable
            # multiset plus, per unstable row, the number of seeds that have it beyond the stable count)
            stable = runs[0]
            for x in runs[1:]:
                stable = stable & x
            freq = collections.Counter()
            for x in runs:
                for row in (x - stable).keys():
                    freq[row] += 1
            name = f"call_{k}.json.xz"
            with lzma.open(f"{out}/{name}", "wt", preset=9) as g:
                json.dump({"flags": flags, "seeds": nseeds, "stable": sorted(stable.elements()),
                           "unstable": {row: n for row, n in sorted(freq.items())}}, g, indent=0)
            sizes = sorted(sum(x.values()) for x in runs)
            meta["calls"].append({"flags": flags, "file": name, "rows_min": sizes[0], "rows_max": sizes[-1],
                                  "stable": sum(stable.values()), "unstable": len(freq),
                                  "unstable_at_90pct": sum(1 for n in freq.values() if n >= 0.9 * nseeds)})
            print(f"  mid {' '.join(flags)}: {sizes[0]}..{sizes[-1]} rows, {sum(stable.values())} in every seed, "
                  f"{len(freq)} seed-dependent ({meta['calls'][-1]['unstable_at_90pct']} of them under >= 90 % of the seeds)")
        json.dump(meta, open(f"{out}/meta.json", "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--seeds", type=int, default=32)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", choices=["tags", "mid"])
    a = ap.parse_args()
    gold = os.path.join(ROOT, "tests", "golden")
    if a.only in (None, "tags"):
        make_tags(a.ref, os.path.join(gold, "tags"))
    if a.only in (None, "mid"):
        make_mid(a.ref, os.path.join(gold, "mid"), a.seeds, a.jobs)


if __name__ == "__main__":
    main()
