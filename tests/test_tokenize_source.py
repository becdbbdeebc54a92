"""The device line parser (pantas_amd/csrc/gaf_device.h, what tokenize.hip runs per GPU thread) compiled for the CPU
under ASan/UBSan and compared, array by array, with the host tokeniser on the same text
(tests/hostsim/hostsim_tokenize.cpp).  The GPU kernels themselves are checked under -m gpu (test_gpu_tokenize.py)."""
import json
import os
import random
import subprocess

import pytest

import fuzz_gaf
from conftest import CASES, GOLDEN, ROOT
from test_hostsim import BUILD, CSRC, CXX


@pytest.fixture(scope="module")
def tokenize_sim():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "hostsim_tokenize")
    subprocess.check_call(CXX + ["-o", exe, os.path.join(ROOT, "tests/hostsim/hostsim_tokenize.cpp"),
                                 *[os.path.join(CSRC, f) for f in ("graph.cpp", "common.cpp", "gaf.cpp")], "-lpthread"])
    return exe


def run(exe, *args):
    r = subprocess.run([exe, *args], capture_output=True, text=True)
    assert r.returncode == 0, (args, r.stdout, r.stderr[-2000:])
    return r.stdout


def test_dv_threshold_on_decimal_digits(tokenize_sim):
    """float(text) > 0.1 decided on the digits agrees with strtod on 200 k strings around the rounding boundary"""
    assert run(tokenize_sim, "--dv").startswith("OK dv")


@pytest.mark.parametrize("case", CASES)
def test_goldens(tokenize_sim, case):
    d = os.path.join(GOLDEN, case)
    out = run(tokenize_sim, f"{d}/in.gaf", f"{d}/in.gfa")
    assert out.startswith("OK"), out
    for thr in ("0", "61"):
        assert run(tokenize_sim, f"{d}/in.gaf", f"{d}/in.gfa", thr).startswith("OK")


def test_crash_inputs_same_class_same_line(tokenize_sim):
    cd = os.path.join(GOLDEN, "crashes")
    seen = set()
    for c in json.load(open(os.path.join(GOLDEN, "crashes.json"))):
        if c["stage"] != "augment":
            continue
        out = run(tokenize_sim, f"{cd}/{c['name']}.gaf", f"{cd}/{c['name']}.gfa")
        seen.add(out.strip().split("error=")[-1].split()[0])
    assert {"IndexError", "ValueError", "KeyError", "AssertionError"} <= seen, seen


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzzed_text(tokenize_sim, tmp_path, seed):
    rng = random.Random(seed)
    gfa, gaf = str(tmp_path / "g.gfa"), str(tmp_path / "r.gaf")
    fuzz_gaf.write_graph(gfa)
    kinds = {}
    for _ in range(250):
        nl = rng.choice([1, 2, 5, 30, 200])
        with open(gaf, "w", encoding="latin-1", newline="") as f:
            for i in range(nl):
                l = fuzz_gaf.random_line(rng, i)
                if rng.random() < (0.6 if nl < 10 else 0.02):
                    l = fuzz_gaf.mutate_text(rng, l)
                f.write(l + ("\n" if (i < nl - 1 or rng.random() < 0.9) else ""))
        k = run(tokenize_sim, gaf, gfa).strip().split("error=")[-1].split()[0]
        kinds[k] = kinds.get(k, 0) + 1
    assert kinds.get("none", 0) > 50 and len(kinds) >= 4, kinds


def test_multithreaded_host_tokeniser_same_records(tokenize_sim, tmp_path):
    """host tokeniser with 4 threads (chunk merge) against the per-line device parser"""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "s")
    Synth(seed=41, n_genes=50, n_lines=60000, p_indel=0.2, p_novel_skip=0.05).write(pre)
    out = run(tokenize_sim, pre + ".gaf", pre + ".gfa", "20", "4")
    assert out.startswith("OK lines=60000")
    # the fast form for plain lines (parse_gaf_line_fast) takes what a mapper writes: every line of the synthetic recipe
    assert int(out.split("fast=")[1]) >= 59990, out


def test_fast_form_on_mutated_plain_lines(tokenize_sim, tmp_path):
    """parse_gaf_line_fast must either refuse a line or agree with the general parser: plain synthetic lines with one byte
    changed, inserted or dropped at a random place (tabs doubled, digits turned into letters, names cut, tags renamed ...)"""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "s")
    Synth(seed=43, n_genes=30, n_lines=3000, p_indel=0.3, p_sub=0.4, p_novel_skip=0.05, p_two_indels=0.1).write(pre)
    lines = open(pre + ".gaf").read().splitlines()
    rng = random.Random(5)
    alphabet = "\t\t\t  >><<**::++--==~~0123456789acgtZsdvf.c_"
    taken = 0
    for trial in range(12):         # (the round's campaign ran 40 x 120 lines here and 3 000 fuzzed files beside: no difference)
        out_lines = []
        for l in rng.sample(lines, 100):
            for _ in range(rng.choice([1, 1, 2])):
                at = rng.randrange(len(l) + 1)
                kind = rng.random()
                if kind < 0.4 and at < len(l):
                    l = l[:at] + rng.choice(alphabet) + l[at + 1:]
                elif kind < 0.7:
                    l = l[:at] + rng.choice(alphabet) + l[at:]
                elif at < len(l):
                    l = l[:at] + l[at + 1:]
            out_lines.append(l)
        # one mutated line per file in front of plain ones: a line the reference dies on ends the comparison of a file
        for k, l in enumerate(out_lines):
            with open(pre + ".m.gaf", "w") as f:
                f.write(l + "\n" + "\n".join(lines[k * 3:k * 3 + 3]) + "\n")
            out = run(tokenize_sim, pre + ".m.gaf", pre + ".gfa")
            taken += int(out.split("fast=")[1])
    assert taken > 2500         # (most mutations leave a line plain, or touch a free-text column)

