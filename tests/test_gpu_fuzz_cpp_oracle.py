"""Differential runs of `pantas augment` against the C++ oracle (oracle/augment_oracle.cpp, pinned to the reference's goldens) on
generated workloads whose parameters are drawn to reach the kernels' rare paths: many links the graph does not have per chunk (the
novel-link queue overflows into the inline insertion), indel-heavy reads (one-indel walk, general walk, rare-key table), long reads
(paths beyond the short-walk class), a handful of genes under a million reads (every lane of a wave on the same node, the 16-bit
LDS fields near their bound), wide graphs with flat expression (node tables overflow into memory atomics), tiny batches of a
streamed run, '~' ops planted in front of one cs in forty (their length leaks in from the line before).  The augmented GFAs must be
the same file.

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
    return os.path.getsize(pre + ".gpu.gfa")


def test_generated_workloads_against_the_cpp_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    runs = int(os.environ.get("PANTAS_FUZZ_RUNS", "4"))
    rng = random.Random(int(os.environ.get("PANTAS_FUZZ_SEED", "20261004")))
    log = os.environ.get("PANTAS_FUZZ_LOG")
    for i in range(runs):
        kind, kw, env = draw(rng)
        tmp = tempfile.mkdtemp(prefix="pantas_fuzz_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            size = one(kind, kw, env, tmp)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        if log:
            with open(log, "a") as f:
                f.write(f"run {i}: {kind} same file ({size} bytes) {kw} {env}\n")
