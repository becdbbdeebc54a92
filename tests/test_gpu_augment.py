"""Parity of the HIP augment path (through the C ABI) with the reference goldens and the oracle."""
import io
import json
import os
import subprocess
import sys
import tempfile

import pytest

import augment_oracle
from conftest import CASES, GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def run_engine(gfa, gaf, thr=20):
    from pantas_amd.engine import Graph
    g = Graph(gfa).to_device(0)
    with tempfile.TemporaryFile() as f:
        bs, ls = g.augment(gaf, thr, f.fileno())
        f.seek(0)
        return f.read().decode(), bs, g


@pytest.mark.parametrize("case", CASES)
def test_golden_bytes(case):
    d = os.path.join(GOLDEN, case)
    got, bs, _ = run_engine(f"{d}/in.gfa", f"{d}/in.gaf")
    assert got == open(f"{d}/expected.aug.gfa").read()
    assert bs.rejected == json.load(open(f"{d}/expected.json"))["rejected"]


@pytest.mark.parametrize("case", CASES)
def test_golden_bytes_through_the_sparse_counter_download(case, monkeypatch):
    """the counters come back as non-zero 64-byte blocks + a bitmap (what a whole-genome run does); forced here on the goldens'
    small graphs, where block borders fall inside nc / il0 / oll / rc and the last bitmap word is ragged"""
    monkeypatch.setenv("PANTAS_SPARSE_DOWNLOAD", "1")
    d = os.path.join(GOLDEN, case)
    got, bs, _ = run_engine(f"{d}/in.gfa", f"{d}/in.gaf")
    assert got == open(f"{d}/expected.aug.gfa").read()
    monkeypatch.delenv("PANTAS_SPARSE_DOWNLOAD")
    monkeypatch.setenv("PANTAS_DENSE_DOWNLOAD", "1")
    again, _, _ = run_engine(f"{d}/in.gfa", f"{d}/in.gaf")
    assert again == got


def test_cli_matches_reference_streams():
    d = os.path.join(GOLDEN, "quirks")
    r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", f"{d}/in.gaf", f"{d}/in.gfa"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(f"{d}/expected.aug.gfa").read()
    want = json.load(open(f"{d}/expected.json"))["stderr"]
    assert r.stderr.splitlines() == want


def test_crash_inputs_fail_with_empty_stdout():
    from pantas_amd.engine import PantasError
    crashes = [c for c in json.load(open(os.path.join(GOLDEN, "crashes.json"))) if c["stage"] == "augment"]
    cd = os.path.join(GOLDEN, "crashes")
    for c in crashes:
        with pytest.raises(PantasError) as ei:
            run_engine(f"{cd}/{c['name']}.gfa", f"{cd}/{c['name']}.gaf")
        assert ei.value.kind == c["exception"], c["name"]
        r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", f"{cd}/{c['name']}.gaf", f"{cd}/{c['name']}.gfa"],
                           capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 1 and r.stdout == ""


@pytest.mark.parametrize("seed,genes,lines,kw", [
    (11, 60, 20000, {}),
    (12, 30, 30000, dict(p_indel=0.3, p_sub=0.5, bubble_rate=0.15, p_novel_skip=0.1, p_novel_ce=0.05)),
    (13, 5, 20000, dict(zipf_a=3.0)),          # heavy skew: a few hot nodes take most increments
])
def test_oracle_parity_larger(tmp_path, seed, genes, lines, kw):
    from pantas_amd import synth
    pre = str(tmp_path / "s")
    synth.write_all(pre, seed=seed, n_genes=genes, n_lines=lines, **kw)
    got, bs, _ = run_engine(pre + ".gfa", pre + ".gaf")
    want = io.StringIO()
    a = augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    assert bs.rejected == a.rejected
    assert got == want.getvalue()


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_around_the_short_walk_bounds(tmp_path, seed):
    """tests/long_path_case.py: paths of 27 .. 40 nodes (the short-walk class ends at 31, at 29 with an indel), one
    '+' / '-' at every position class with mismatches around it, both strands -- engine against the oracle, byte for byte."""
    import long_path_case
    pre = str(tmp_path / "lp")
    long_path_case.write_case(pre, seed, 20000)
    got, bs, _ = run_engine(pre + ".gfa", pre + ".gaf")
    want = io.StringIO()
    a = augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    assert bs.rejected == a.rejected
    assert got == want.getvalue()


def test_threshold_argument(tmp_path):
    d = os.path.join(GOLDEN, "synth_b")
    for thr in (0, 5, 61):
        got, bs, _ = run_engine(f"{d}/in.gfa", f"{d}/in.gaf", thr)
        want = io.StringIO()
        a = augment_oracle.augment(f"{d}/in.gaf", f"{d}/in.gfa", want, thr=thr)
        assert got == want.getvalue() and bs.rejected == a.rejected


def test_relaunch_is_linear():
    """counters after two launches over the same resident batch = 2 x one launch (size-independent property)."""
    from pantas_amd.engine import Graph
    d = os.path.join(GOLDEN, "synth_a")
    g = Graph(f"{d}/in.gfa").to_device(0)
    g.augment_reset()
    b = g.load_gaf(f"{d}/in.gaf")
    g.launch(b)
    g.finish()
    one = [x.copy() for x in g.read_counters()]
    g.launch(b)
    g.finish()
    two = g.read_counters()
    for a, c in zip(one, two):
        assert (2 * a.astype("int64") == c.astype("int64")).all()


def test_launches_that_are_not_waited_for(tmp_path):
    """pantas_augment_launch_async / _collect (ABI 7): launches enqueued back to back leave what the same number of waited-for launches
    leaves -- counters, sparse tables, the written GFA --, the collect returns one triple of kernel times per launch, oldest first, and
    the 65th launch in flight is refused."""
    from pantas_amd import synth
    from pantas_amd.engine import Graph, PantasError
    pre = str(tmp_path / "s")
    synth.write_all(pre, seed=77, n_genes=60, n_lines=40000, p_indel=0.2, p_sub=0.4, p_novel_skip=0.08, p_novel_ce=0.04)

    def run(pipelined, launches):
        g = Graph(pre + ".gfa").to_device(0)
        g.augment_reset()
        b = g.load_gaf(pre + ".gaf")
        if pipelined:
            for _ in range(launches):
                g.launch_async(b)
            times = g.launch_collect()
            assert len(times) == launches and all(m > 0 and r >= 0 and 0 <= s <= m for r, m, s in times)
            assert g.launch_collect() == []
        else:
            for _ in range(launches):
                g.launch(b)
        g.finish()
        with tempfile.TemporaryFile() as f:
            g.write_gfa(f.fileno())
            f.seek(0)
            return f.read()

    assert run(True, 5) == run(False, 5)
    # one launch through either door is the oracle's file
    want = io.StringIO()
    augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    assert run(True, 1).decode() == want.getvalue()
    g = Graph(pre + ".gfa").to_device(0)
    g.augment_reset()
    b = g.load_gaf(pre + ".gaf")
    for _ in range(64):
        g.launch_async(b)
    with pytest.raises(PantasError):
        g.launch_async(b)
    assert len(g.launch_collect()) == 64
    g.launch_async(b)                      # room again
    assert len(g.launch_collect()) == 1
    g.finish()


def test_sharded_driver_nccl_single_rank(tmp_path):
    """pantas_amd.distributed with the real library and the RCCL backend (one rank: the only GPU here).
    Exercises the zero-copy tensor view of the counters, the collectives and the rank-0 writer."""
    d = os.path.join(GOLDEN, "synth_b")
    out = tmp_path / "out.gfa"
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    with open(out, "wb") as f:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pantas_amd.distributed",
                            f"{d}/in.gaf", f"{d}/in.gfa"], stdout=f, stderr=subprocess.PIPE, text=True, cwd=ROOT,
                           timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out).read() == open(f"{d}/expected.aug.gfa").read()
    assert f"Rejected alignments: {json.load(open(f'{d}/expected.json'))['rejected']}" in r.stderr


def test_sharded_driver_two_ranks_one_parses_the_graph(tmp_path):
    """two ranks of the read-sharded driver (both on the box's one GPU, gloo between them: PANTAS_DIST_REHEARSE_ON_ONE_GPU): rank 0
    parses the GFA and leaves the binary cache, rank 1 loads it (round 3), each tokenises its byte range of the GAF, the flags and
    the counters are exchanged, rank 0 writes -- the reference's bytes"""
    d = os.path.join(GOLDEN, "synth_c")
    out = tmp_path / "out.gfa"
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {**os.environ, "PANTAS_DIST_REHEARSE_ON_ONE_GPU": "1", "PANTAS_TIMING": "1"}
    with open(out, "wb") as f:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pantas_amd.distributed",
                            f"{d}/in.gaf", f"{d}/in.gfa"], stdout=f, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(out).read() == open(f"{d}/expected.aug.gfa").read()
    assert f"Rejected alignments: {json.load(open(f'{d}/expected.json'))['rejected']}" in r.stderr
    # one rank read the text, the other the cache it left (graph_cache.cpp prints which under PANTAS_TIMING)
    assert r.stderr.count("graph cache: loaded") >= 1, r.stderr[-3000:]


def test_release_device_keeps_the_finished_run(tmp_path):
    """pantas_graph_release_device (ABI 5): the device memory goes back after the last kernel, the finished run can still be written,
    and the graph can go back to the device for another run"""
    import tempfile
    from pantas_amd.engine import Graph
    d = os.path.join(GOLDEN, "synth_b")
    g = Graph(f"{d}/in.gfa").to_device(0)
    g.augment(f"{d}/in.gaf", 20, -1)
    g.release_device()
    with tempfile.TemporaryFile() as f:
        g.write_gfa(f.fileno())
        f.seek(0)
        assert f.read().decode() == open(f"{d}/expected.aug.gfa").read()
    g.to_device(0)
    with tempfile.TemporaryFile() as f:
        g.augment(f"{d}/in.gaf", 20, f.fileno())
        f.seek(0)
        assert f.read().decode() == open(f"{d}/expected.aug.gfa").read()
