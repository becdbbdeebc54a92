"""Differential runs of `pantas augment` against the C++ oracle (oracle/augment_oracle.cpp, pinned to the reference's goldens) on
generated workloads whose parameters are drawn to reach the kernels' rare paths: many links the graph does not have per chunk (the
novel-link queue overflows into the inline insertion), indel-heavy reads (one-indel walk, general walk, rare-key table), long reads
(paths beyond the short-walk class), a handful of genes under a million reads (every lane of a wave on the same node, the 16-bit
LDS fields near their bound), wide graphs with flat expression (node tables overflow into memory atomics), tiny batches of a
streamed run, '~' ops planted in front of one cs in forty (their length leaks in from the line before).  The augmented GFAs must be
the same file; since round 5 `pantas call` with drawn flags follows on every augmented GFA, against oracle/call_oracle.cpp (call_leg).

PANTAS_FUZZ_RUNS=<n> draws n parameter sets instead of the four of the default run (a campaign of 40 was run at the end of round 2:
profiles/r2_fuzz_cpp_oracle.txt)."""
import os
import random
import shutil
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def draw(rng):
    kind = rng.choice(["novel", "indel", "long", "hub", "wide", "stream", "mix", "tilde"])
    kw = dict(seed=rng.randrange(1, 10**6), n_genes=rng.choice([40, 300, 2000]), n_lines=rng.choice([200_000, 600_000]),
              base_id=10**7, zipf_a=rng.choice([0.6, 1.0, 1.4]), p_sub=rng.choice([0.1, 0.3, 0.6]),
              p_indel=rng.choice([0.0, 0.05, 0.3]), p_novel_skip=rng.choice([0.0, 0.04, 0.3]), p_novel_ce=rng.choice([0.0, 0.02, 0.2]),
              p_clip=rng.choice([0.0, 0.03, 0.3]), read_len=rng.choice([50, 100, 150]), bubble_rate=rng.choice([0.0, 0.06, 0.3]),
              intron_long_max=rng.choice([40, 300]), n_chrom=rng.choice([1, 3]))
    env = {}
    if kind == "novel":
        kw.update(p_novel_skip=0.5, p_novel_ce=0.3, n_genes=60, zipf_a=1.4)
    elif kind == "indel":
        kw.update(p_indel=0.6, p_del_hot=rng.choice([0.0, 0.9]), hot_sites=20)
    elif kind == "long":
        kw.update(read_len=rng.choice([600, 1500]), n_lines=150_000)
    elif kind == "hub":
        kw.update(n_genes=8, zipf_a=1.6, n_lines=1_000_000)
    elif kind == "wide":
        kw.update(n_genes=6000, zipf_a=0.2, n_lines=600_000)
    elif kind == "stream":
        env["PANTAS_BATCH_BYTES"] = str(rng.choice([1 << 20, 7 << 20]))
        env["PANTAS_RESIDENT_BYTES"] = str(rng.choice([0, 1 << 40]))
    elif kind == "tilde":
        kw.update(n_lines=rng.choice([20_000, 120_000]), n_genes=rng.choice([40, 300]))
    if rng.random() < 0.3:
        env["PANTAS_TOKENIZER"] = "host"
    return kind, kw, env


def plant_tilde_first_ops(path, rng):
    """every ~40th line gets a '~' op in front of its cs (its length then leaks in from the line before: the reference's
    `curr_cigar_len` outlives a line) and a long ':' run behind it, so that the ops never run out before the path does"""
    out = []
    for i, line in enumerate(open(path, "rb").read().split(b"\n")):
        if i > 3 and rng.random() < 0.025 and b"\tcs:Z:" in line:
            head, tail = line.split(b"\tcs:Z:", 1)
            cs, _, rest = tail.partition(b"\t")
            line = head + b"\tcs:Z:~ac" + str(rng.randrange(1, 90)).encode() + b"gt" + cs + b":5000" + (b"\t" + rest if rest else b"")
        out.append(line)
    open(path, "wb").write(b"\n".join(out))


def one(kind, kw, env, tmp):
    from pantas_amd.engine import Synth
    pre = os.path.join(tmp, "w")
    Synth(**kw).write(pre)
    if kind == "tilde":
        plant_tilde_first_ops(pre + ".gaf", random.Random(kw["seed"]))
    with open(pre + ".gpu.gfa", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT,
                           env={**os.environ, **env})
    with open(pre + ".or.gfa", "wb") as f:
        # (a '~' first op takes its length from the line before: the oracle reads such a file on one thread)
        o = subprocess.run([os.path.join(ROOT, "oracle", "bin", "augment_oracle_cpp"), "--threads", "1" if kind == "tilde" else "8", pre + ".gaf", pre + ".gfa"],
                           stdout=f, stderr=subprocess.PIPE)
    what = f"{kind} {kw} {env}"
    assert o.returncode == 0, what + o.stderr.decode()[-500:]
    assert r.returncode == 0, what + r.stderr.decode()[-1500:]
    same = open(pre + ".gpu.gfa", "rb").read() == open(pre + ".or.gfa", "rb").read()
    assert same, what
    rows = call_leg(pre, random.Random(kw["seed"] + 1), what) if os.environ.get("PANTAS_FUZZ_CALL", "1") != "0" else None
    return os.path.getsize(pre + ".gpu.gfa"), rows


def call_leg(pre, rng, what):
    """round 5: `pantas call` on the augmented GFA of the run, with drawn flags, against oracle/call_oracle.cpp -- the same rows as a
    multiset (the oracle's 128-bit digest of the normalised rows), or the same exception where the reference raises.  The rows the
    reads made thousands of links long (kind "hub"), the searches of rule R3 under small and large --isw, the staging pool, the
    lists and masks made once per table set: every table and switch of DESIGN 3's call row meets graphs of other shapes here."""
    import json
    flags = ["--rca", str(rng.choice([0, 1, 3, 8])), "--isw", str(rng.choice([1, 2, 3, 5, 8])), "--minintronsize", str(rng.choice([1, 20, 100]))]
    if rng.random() < 0.85:
        flags.append("--novel")
    if rng.random() < 0.15:
        flags.append("--no-annotated")
    if rng.random() < 0.3:          # (in front: the list takes every word up to the next option, as the reference's parser does)
        flags = ["--events", *rng.sample(["ES", "SS", "IR"], rng.choice([1, 2]))] + flags
    exe = os.path.join(ROOT, "oracle", "bin", "call_oracle_cpp")
    what = f"{what} call {flags}"
    with open(pre + ".gpu.csv", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, pre + ".or.gfa", pre + ".gtf"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
    o = subprocess.run([exe, "--threads", "8", "--digest", "--no-out", *flags, pre + ".or.gfa", pre + ".gtf"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if o.returncode != 0:           # the reference raises here: the same exception class, no rows compared
        want = o.stderr.decode().strip().splitlines()[-1].split(":", 1)[0]
        assert r.returncode != 0 and r.stderr.decode().strip().splitlines()[-1].startswith(want), what + r.stderr.decode()[-800:] + o.stderr.decode()[-800:]
        return f"both raise {want}"
    if r.returncode != 0 and r.stderr.decode().strip().splitlines()[-1].startswith("Unsupported"):
        return "engine refuses: " + r.stderr.decode().strip().splitlines()[-1][:160]      # (inputs whose rows depend on the reference's hash order)
    assert r.returncode == 0, what + r.stderr.decode()[-1500:]
    want = json.loads(o.stderr.decode().strip().splitlines()[-1])
    got = json.loads(subprocess.run([exe, "--threads", "8", "--digest-csv", pre + ".gpu.csv"], capture_output=True, check=True).stdout)
    keys = ("rows", "s1", "s2", "x")
    assert {k: got[k] for k in keys} == {k: want[k] for k in keys}, what + f" {got} {want}"
    return f"{want['rows']} rows {' '.join(flags)}"


def test_generated_workloads_against_the_cpp_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    runs = int(os.environ.get("PANTAS_FUZZ_RUNS", "4"))
    rng = random.Random(int(os.environ.get("PANTAS_FUZZ_SEED", "20261004")))
    log = os.environ.get("PANTAS_FUZZ_LOG")
    for i in range(runs):
        kind, kw, env = draw(rng)
        tmp = tempfile.mkdtemp(prefix="pantas_fuzz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            size, rows = one(kind, kw, env, tmp)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if log:
            with open(log, "a") as f:
                f.write(f"run {i}: {kind} same file ({size} bytes); call: {rows}; {kw} {env}\n")
