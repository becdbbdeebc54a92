#!/usr/bin/env python3
"""Build container only: the mutated tags / GTFs of tests/test_oracle_differential.py through the REAL reference call.py and
through oracle/call_oracle.py -- same exception class, or the oracle's rows within the reference's rows under a few hash seeds
(multiset; rows are compared after the one normalisation of SURVEY 8-P #2).  Prints one summary line (kept in profiles/).

    python tools/mutation_check_vs_reference.py [--ref /root/reference] [--trials 60]
"""
import argparse
import collections
import json
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import call_oracle  # noqa: E402
import test_oracle_differential as t  # noqa: E402
from conftest import GOLDEN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--trials", type=int, default=60)
    a = ap.parse_args()
    td = tempfile.mkdtemp()
    gfa, gtf = td + "/g.gfa", td + "/g.gtf"
    n = agree = 0
    kinds = collections.Counter()
    bad = []
    for case in ("micro", "quirks", "synth_a", "synth_b"):
        base = open(os.path.join(GOLDEN, case, "expected.aug.gfa")).read().splitlines()
        gtf0 = open(os.path.join(GOLDEN, case, "in.gtf")).read().splitlines()
        for trial in range(a.trials):
            rnd = random.Random(f"{case}/{trial}")
            what = trial % 3
            open(gfa, "w").write("\n".join(t.mutate_gfa(base, rnd) if what != 1 else base) + "\n")
            open(gtf, "w").write("\n".join(t.mutate_gtf(gtf0, rnd) if what != 0 else gtf0) + "\n")
            flags = t.FLAG_SETS[trial % len(t.FLAG_SETS)]
            mine = t.python_verdict(flags, gfa, gtf)
            refs = []
            for seed in (0, 1, 2):
                r = subprocess.run([sys.executable, os.path.join(a.ref, "scripts", "call.py"), *flags, gfa, gtf], capture_output=True, text=True,
                                   env={**os.environ, "PYTHONHASHSEED": str(seed)})
                refs.append(r)
                if r.returncode:
                    break
            n += 1
            r = refs[-1]
            if r.returncode:
                exc = r.stderr.strip().splitlines()[-1].split(":")[0]
                ok = mine == ("raise", exc)
                kinds[exc] += 1
            else:
                kinds["rows"] += 1
                ok = mine[0] == "rows"
                if ok:
                    got = collections.Counter(map(call_oracle.normalise_row, mine[1]))
                    seen = [collections.Counter(map(call_oracle.normalise_row, x.stdout.splitlines()[1:])) for x in refs]
                    union = collections.Counter()
                    stable = seen[0]
                    for s in seen:
                        union |= s
                        stable &= s
                    ok = not (stable - got) and not (got - union)
            agree += ok
            if not ok:
                bad.append((case, trial, mine[0], mine[1] if mine[0] == "raise" else len(mine[1]), r.stderr.strip().splitlines()[-1][:80] if r.returncode else "rows"))
    print(json.dumps({"inputs": n, "agree": agree, "reference_outcomes": dict(kinds), "disagreements": bad[:20]}))


if __name__ == "__main__":
    main()
