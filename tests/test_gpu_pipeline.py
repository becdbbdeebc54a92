"""The whole user-visible chain on the GPU box: `pantas augment` (graph cache on) -> `pantas call` -> `pantas quant` -> `pantas remap`
for two conditions x three replicates, against tests/golden/quant: the event CSVs there came from the oracle on the same six
synthetic read sets (tools/make_golden_quant.py regenerates them from seeds), the quant / remap outputs from the REAL reference
scripts run on those CSVs.  The engine's CSVs must equal the stored ones row for row (same rows; the order is the engine's and
the oracle's common rule R1), hence quant and remap on them reproduce the reference's bytes."""
import collections
import gzip
import json
import os
import subprocess
import sys

import pytest

import call_oracle
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
Q = os.path.join(GOLDEN, "quant")


def expected(name):
    return gzip.open(os.path.join(Q, name + ".gz"), "rt").read()


def test_augment_call_quant_remap_chain(tmp_path):
    from pantas_amd import synth
    meta = json.load(open(os.path.join(Q, "expected.json")))
    base = dict(seed=5, n_genes=7, max_exons=7, bubble_rate=0.05, p_novel_skip=0.08, p_novel_ce=0.05, p_intron=0.12, p_novel_ss=0.06)
    env = {**os.environ, "PANTAS_GRAPH_CACHE": str(tmp_path / "cache")}     # the graph is the same for all six: parsed once, cached
    csvs = {}
    cache_used = 0
    for cond in (1, 2):
        for rep in (1, 2, 3):
            pre = str(tmp_path / f"c{cond}_r{rep}")
            synth.write_all(pre, n_lines=2500 + 400 * rep, read_seed=100 * cond + rep, zipf_a=0.6 if cond == 1 else 1.6, **base)
            if (cond, rep) != (1, 1):
                os.replace(pre + ".gaf", str(tmp_path / f"reads_c{cond}_r{rep}.gaf"))
                gfa, gaf = str(tmp_path / "c1_r1.gfa"), str(tmp_path / f"reads_c{cond}_r{rep}.gaf")     # ONE graph file: the cache applies
            else:
                gfa, gaf = pre + ".gfa", pre + ".gaf"
            with open(pre + ".aug.gfa", "wb") as f:
                r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", gaf, gfa], stdout=f, stderr=subprocess.PIPE, cwd=ROOT,
                                   env={**env, "PANTAS_TIMING": "1"})
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            cache_used += b"load_graph_cached: arrays from cache" in r.stderr
            flags = ["--rca", "2" if rep != 3 else "-1", "--novel", "--minintronsize", "5"]
            with open(pre + ".csv", "wb") as f:
                r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, pre + ".aug.gfa", str(tmp_path / "c1_r1.gtf")],
                                   stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            mine = open(pre + ".csv").read().splitlines()
            want = open(os.path.join(Q, f"c{cond}_r{rep}.csv")).read().splitlines()
            assert mine[0] == want[0] == call_oracle.HEADER
            norm = call_oracle.normalise_row
            assert collections.Counter(map(norm, mine[1:])) == collections.Counter(map(norm, want[1:])), (cond, rep)
            csvs[(cond, rep)] = (pre + ".csv", mine == want)
    assert cache_used == 5, "the binary graph cache must serve every replicate after the first"
    c1 = [csvs[(1, r)][0] for r in (1, 2, 3)]
    c2 = [csvs[(2, r)][0] for r in (1, 2, 3)]
    same_order = all(v[1] for v in csvs.values())
    for case in meta["quant"]:
        n = case["replicates"]
        r = subprocess.run([sys.executable, "-m", "pantas_amd.quantify", *case["flags"], "--c1", *c1[:n], "--c2", *c2[:n]],
                           capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr
        if same_order:
            assert r.stdout == expected(case["expected"]), case
        else:       # quant prints events in the order it first meets them: same rows, the engine's order
            assert sorted(r.stdout.splitlines()) == sorted(expected(case["expected"]).splitlines()), case
        if case["expected"] in ("quant_0_3.csv", "quant_2_1.csv"):
            src = str(tmp_path / case["expected"])
            open(src, "w").write(r.stdout)
            for rm in meta["remap"]:
                if rm["input"] != case["expected"]:
                    continue
                r2 = subprocess.run([sys.executable, "-m", "pantas_amd.remap", *rm["flags"], src, os.path.join(Q, "anno.gtf")],
                                    capture_output=True, text=True, cwd=ROOT)
                assert r2.returncode == 0, r2.stderr
                if same_order:
                    assert r2.stdout == expected(rm["expected"])
                else:
                    assert sorted(r2.stdout.splitlines()) == sorted(expected(rm["expected"]).splitlines())
