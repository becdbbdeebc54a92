"""The device-side sources (pantas_amd/csrc/walk.h, call_device.h) and the product's host code
(loader, tokeniser, writer, tag interning, renderer) compiled for the CPU with ASan/UBSan by
tests/hostsim/*.cpp and checked against the reference goldens and the oracle.

This harness is test infrastructure: it lets the CPU-only suite exercise the exact kernel
source.  The product never links or loads it (its kernels run on the GPU or fail)."""
import collections
import io
import json
import os
import subprocess

import pytest

import augment_oracle
import call_oracle
from conftest import CASES, GOLDEN, ROOT
from test_oracle_golden import _flags_to_opts, check_rows

from conftest import BUILD, CSRC, CXX  # noqa: E402,F401  (the `hostsim` fixture lives in conftest.py: every test file builds it once per session)


def run(cmd):
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("case", CASES)
def test_walk_source_vs_reference_bytes(hostsim, case):
    d = os.path.join(GOLDEN, case)
    r = run([hostsim["augment"], f"{d}/in.gaf", f"{d}/in.gfa"])
    assert r.returncode == 0, r.stderr
    assert r.stdout == open(f"{d}/expected.aug.gfa").read()
    assert r.stderr.splitlines() == json.load(open(f"{d}/expected.json"))["stderr"]
    # stdout on a regular file (the `> out.gfa` of the command line): the writer's pieces go out concurrently with pwrite,
    # behind whatever the file already holds
    import tempfile
    with tempfile.TemporaryFile() as f:
        f.write(b"# already here\n")
        f.flush()
        r2 = subprocess.run([hostsim["augment"], f"{d}/in.gaf", f"{d}/in.gfa"], stdout=f, stderr=subprocess.PIPE)
        assert r2.returncode == 0
        f.seek(0)
        assert f.read().decode() == "# already here\n" + r.stdout
    # the shell's `> out.gfa` is a write-only descriptor: pwrite at the piece offsets (default), the opt-in shared mapping
    # (through a read-write reopening of the same file), write() in order
    for env in ({}, {"PANTAS_OUTPUT_MMAP": "1"}, {"PANTAS_WRITER_SEQUENTIAL": "1"}):
        with tempfile.NamedTemporaryFile(delete=False) as tf:
            path = tf.name
        try:
            with open(path, "wb") as f:
                r3 = subprocess.run([hostsim["augment"], f"{d}/in.gaf", f"{d}/in.gfa"], stdout=f, stderr=subprocess.PIPE, env={**os.environ, **env})
            assert r3.returncode == 0, r3.stderr
            assert open(path).read() == r.stdout, env
        finally:
            os.unlink(path)


@pytest.mark.parametrize("case", CASES)
def test_call_source_vs_reference_rows(hostsim, case):
    d = os.path.join(GOLDEN, case)
    for gc in json.load(open(f"{d}/expected.json"))["calls"]:
        r = run([hostsim["call"], *gc["flags"], f"{d}/expected.aug.gfa", f"{d}/in.gtf"])
        assert r.returncode == 0, r.stderr
        rows = r.stdout.splitlines()
        assert rows[0] == call_oracle.HEADER
        check_rows(rows[1:], gc)
        # the same with every host step of the table building on its parallel path (sharded tag interning, chunked GTF, ...),
        # which inputs of this size would otherwise never take: identical rows in identical order
        r2 = subprocess.run([hostsim["call"], *gc["flags"], f"{d}/expected.aug.gfa", f"{d}/in.gtf"], capture_output=True, text=True,
                            env={**os.environ, "PANTAS_TEST_SMALL_PARALLEL": "1"})
        assert r2.returncode == 0, r2.stderr
        assert r2.stdout == r.stdout


def test_crash_inputs(hostsim):
    cd = os.path.join(GOLDEN, "crashes")
    for c in json.load(open(os.path.join(GOLDEN, "crashes.json"))):
        if c["stage"] == "augment":
            r = run([hostsim["augment"], f"{cd}/{c['name']}.gaf", f"{cd}/{c['name']}.gfa"])
        else:
            r = run([hostsim["call"], *c["flags"], f"{cd}/{c['call_input']}", f"{cd}/{c['name']}.gtf"])
        assert r.returncode == 1, c["name"]
        assert r.stderr.strip().splitlines()[-1].startswith(c["exception"]), (c["name"], r.stderr)


@pytest.mark.parametrize("seed,genes,lines,kw,flags", [
    (21, 40, 6000, {}, ["--rca", "2", "--novel", "--minintronsize", "30"]),
    (22, 25, 8000, dict(bubble_rate=0.15, p_novel_skip=0.1, p_novel_ce=0.06, p_intron=0.15, p_novel_ss=0.1),
     ["--rca", "1", "--novel", "--isw", "3", "--minintronsize", "10"]),
    (23, 30, 5000, dict(bubble_rate=0.0), ["--events", "ES", "SS", "--novel", "--rca", "0"]),
])
def test_sources_vs_oracle_exact(hostsim, tmp_path, seed, genes, lines, kw, flags):
    """Same fixed choice rules on both sides (R1-R4): rows must agree exactly as a multiset."""
    from pantas_amd import synth
    pre = str(tmp_path / "s")
    synth.write_all(pre, seed=seed, n_genes=genes, n_lines=lines, **kw)
    want = io.StringIO()
    augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    r = run([hostsim["augment"], pre + ".gaf", pre + ".gfa"])
    assert r.returncode == 0 and r.stdout == want.getvalue()
    open(pre + ".aug.gfa", "w").write(r.stdout)
    rows = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, **_flags_to_opts(flags))
    c = run([hostsim["call"], *flags, pre + ".aug.gfa", pre + ".gtf"])
    assert c.returncode == 0, c.stderr
    mine = collections.Counter(call_oracle.normalise_row(x) for x in c.stdout.splitlines()[1:])
    ref = collections.Counter(call_oracle.normalise_row(x) for x in rows)
    assert len(rows) > 50
    assert mine == ref


def test_wide_short_walk_source(hostsim, tmp_path):
    """walk.h walk_short_w (16-byte node records: links matched against a node's first three out-slots where it is visited; round 3's
    dominant kernel) for EVERY short-walk line: the goldens' bytes, and the oracle on workloads with many bubbles (second / third
    out-slots), links the graph does not have, indels, dropped nodes in between."""
    env = {**os.environ, "HOSTSIM_WIDE_WALK": "1"}
    for case in CASES:
        d = os.path.join(GOLDEN, case)
        r = subprocess.run([hostsim["augment"], f"{d}/in.gaf", f"{d}/in.gfa"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout == open(f"{d}/expected.aug.gfa").read(), case
    from pantas_amd import synth
    for seed, kw in ((31, dict(bubble_rate=0.3, p_novel_skip=0.15, p_novel_ce=0.1)), (32, dict(p_indel=0.4, p_sub=0.5, bubble_rate=0.2)),
                     (33, dict(bubble_rate=0.5, p_novel_ss=0.2, p_intron=0.2))):
        pre = str(tmp_path / f"w{seed}")
        synth.write_all(pre, seed=seed, n_genes=30, n_lines=8000, **kw)
        want = io.StringIO()
        augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
        r = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and r.stdout == want.getvalue(), seed


def test_position_parallel_short_walk_source(hostsim, tmp_path):
    """walk.h tile_* (round 4's dominant kernel: one lane per path position, tiles of 64 lanes, the op-space position of a node from
    a scan of the rooms, first / last / neighbour kept node from 64-bit masks) for EVERY short-walk line, the line standing alone or
    among copies of itself in the tile: the goldens' bytes, the edge goldens the CPU harness takes, and the oracle on workloads with
    many bubbles, links the graph does not have, indels (dropped nodes in between, exact within-line orders), long reads"""
    env = {**os.environ, "HOSTSIM_TILE_WALK": "1"}
    for case in CASES:
        d = os.path.join(GOLDEN, case)
        r = subprocess.run([hostsim["augment"], f"{d}/in.gaf", f"{d}/in.gfa"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert r.stdout == open(f"{d}/expected.aug.gfa").read(), case
    from pantas_amd import synth
    for seed, kw in ((41, dict(bubble_rate=0.3, p_novel_skip=0.15, p_novel_ce=0.1)), (42, dict(p_indel=0.4, p_sub=0.5, bubble_rate=0.2)),
                     (43, dict(bubble_rate=0.5, p_novel_ss=0.2, p_intron=0.2)), (44, dict(read_len=400, p_indel=0.2, p_sub=0.3))):
        pre = str(tmp_path / f"w{seed}")
        synth.write_all(pre, seed=seed, n_genes=30, n_lines=8000, **kw)
        want = io.StringIO()
        augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
        r = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and r.stdout == want.getvalue(), seed


@pytest.mark.parametrize("fan,isw", [(10, 5), (40, 5), (300, 5), (40, 2)])
def test_intronic_search_beyond_local_frontier(hostsim, tmp_path, fan, isw):
    from hub_case import write_hub_case
    pre = str(tmp_path / "hub")
    write_hub_case(pre, fan)
    flags = ["--novel", "--rca", "0", "--isw", str(isw)]
    want = call_oracle.call(pre + ".gfa", pre + ".gtf", None, **_flags_to_opts(flags))
    r = run([hostsim["call"], *flags, pre + ".gfa", pre + ".gtf"])
    assert r.returncode == 0, r.stderr
    assert sorted(r.stdout.splitlines()[1:]) == sorted(want)
    if isw >= 3:
        assert any(x.startswith("A3,novel") for x in want)      # the search reaches the acceptor through the fan


def test_parallel_loader_equals_sequential(hostsim, tmp_path):
    """Above ~65 k segments the loader builds the name index and both CSRs in parallel; the sequential path
    (also the fallback for duplicate names / unsorted links) must give the same augment and call results."""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "w")
    Synth(seed=31, n_genes=700, n_lines=40000, base_id=10**7, intron_long_max=200).write(pre)
    outs = []
    # (the augmented GFA carries the added links behind everything else: its CSR is the sorted head merged with that tail,
    # or, PANTAS_CSR_NO_MERGE, a radix sort of all links)
    for env in ({}, {"PANTAS_LOADER_SEQUENTIAL": "1"}, {"PANTAS_CSR_NO_MERGE": "1"}):
        e = {**os.environ, **env, "HOSTSIM_THREADS": "4"}
        a = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=e)
        assert a.returncode == 0, a.stderr
        open(pre + ".aug.gfa", "w").write(a.stdout)
        c = subprocess.run([hostsim["call"], "--rca", "1", "--novel", pre + ".aug.gfa", pre + ".gtf"], capture_output=True, text=True, env=e)
        assert c.returncode == 0, c.stderr
        outs.append((a.stdout, c.stdout))
    assert outs[0][0].count("\n") > 150000
    assert outs[0] == outs[1] == outs[2]


def test_call_source_parameter_sweep(hostsim, tmp_path):
    """Many small synthetic graphs with varied structure / read mixes / flags: call rows of the device source
    equal the oracle's exactly (same fixed choice rules)."""
    import random
    from pantas_amd.engine import Synth
    rng = random.Random(5)
    total = 0
    for i in range(14):
        pre = str(tmp_path / f"p{i}")
        kw = dict(seed=100 + i, n_genes=rng.choice([3, 8, 20]), n_lines=rng.choice([300, 1500, 4000]),
                  bubble_rate=rng.choice([0.0, 0.05, 0.2]), max_exons=rng.choice([4, 9]),
                  intron_long_max=rng.choice([12, 40, 120]), p_novel_skip=rng.choice([0.0, 0.05, 0.2]),
                  p_novel_ce=rng.choice([0.0, 0.03, 0.1]), p_intron=rng.choice([0.0, 0.1, 0.3]),
                  p_novel_ss=rng.choice([0.0, 0.05, 0.2]), p_indel=rng.choice([0.0, 0.1, 0.4]), zipf_a=rng.choice([0.5, 1.2, 2.5]))
        Synth(**kw).write(pre)
        a = run([hostsim["augment"], pre + ".gaf", pre + ".gfa"])
        assert a.returncode == 0, a.stderr
        open(pre + ".aug.gfa", "w").write(a.stdout)
        flags = rng.choice([["--novel", "--rca", "0"], ["--novel", "--rca", "2", "--minintronsize", "20"],
                            ["--events", "IR", "--novel", "--minintronsize", "1", "--rca", "1"],
                            ["--events", "SS", "ES", "--novel", "--isw", str(rng.choice([0, 1, 3, 9]))],
                            ["--rca", "1"], ["--no-annotated", "--novel"]])
        want = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, **_flags_to_opts(flags))
        c = run([hostsim["call"], *flags, pre + ".aug.gfa", pre + ".gtf"])
        assert c.returncode == 0, (c.stderr, kw, flags)
        mine = collections.Counter(call_oracle.normalise_row(x) for x in c.stdout.splitlines()[1:])
        ref = collections.Counter(call_oracle.normalise_row(x) for x in want)
        assert mine == ref, (kw, flags)
        total += len(want)
    assert total > 500


def test_graph_cache_round_trip(hostsim, tmp_path):
    """binary graph cache: first run parses and writes it, the second reads it -- same bytes out; a changed GFA is
    parsed again; the C ABI entry reports which happened"""
    import ctypes as C
    import shutil
    import time
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "c")
    Synth(seed=51, n_genes=30, n_lines=4000, p_indel=0.2, p_novel_skip=0.05).write(pre)
    cache = pre + ".pgc"
    env = dict(os.environ, HOSTSIM_GRAPH_CACHE=cache)
    plain = run([hostsim["augment"], pre + ".gaf", pre + ".gfa"])
    first = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
    second = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
    assert "graph cache: parsed" in first.stderr and "graph cache: used" in second.stderr
    assert plain.returncode == 0 and plain.stdout == first.stdout == second.stdout
    # the same bytes under a new mtime, then different bytes: both must be parsed again
    time.sleep(0.01)
    os.utime(pre + ".gfa")
    third = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
    assert "graph cache: parsed" in third.stderr and third.stdout == plain.stdout
    with open(pre + ".gfa", "a") as f:
        f.write("S\t999999\tACGT\n")
    fourth = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
    assert "graph cache: parsed" in fourth.stderr and fourth.stdout.count("\n") == plain.stdout.count("\n") + 1
    # through the C ABI (host only)
    from pantas_amd._lib import lib, check, GraphStats
    for want in (1, 1):
        h, used = C.c_void_p(), C.c_int32()
        check(lib.pantas_graph_load_cached(os.fsencode(pre + ".gfa"), os.fsencode(cache), 1, C.byref(h), C.byref(used)))
        st = GraphStats()
        check(lib.pantas_graph_get_stats(h, C.byref(st)))
        assert used.value == want and st.segments > 100
        lib.pantas_graph_free(h)
    # a damaged cache (same length, an index pushed out of range somewhere in its arrays) is parsed again, never trusted
    good = open(cache, "rb").read()
    for where in (0.3, 0.5, 0.7, 0.9):
        bad = bytearray(good)
        at = int(len(bad) * where) // 8 * 8
        bad[at:at + 8] = b"\xff\xff\xff\x7f\xff\xff\xff\x7f"
        open(cache, "wb").write(bad)
        h, used = C.c_void_p(), C.c_int32()
        check(lib.pantas_graph_load_cached(os.fsencode(pre + ".gfa"), os.fsencode(cache), 0, C.byref(h), C.byref(used)))
        st = GraphStats()
        check(lib.pantas_graph_get_stats(h, C.byref(st)))
        assert st.segments > 100
        lib.pantas_graph_free(h)
        # (damage that lands in padding or in a sequence-length word is harmless and may be accepted: what must never
        # happen is an index outside its array, which the loader checks -- a used cache therefore still gives the right graph)
        if used.value:
            r = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env=env)
            assert r.returncode == 0
    open(cache, "wb").write(good)
    # a truncated cache is ignored
    with open(cache, "r+b") as f:
        f.truncate(os.path.getsize(cache) // 2)
    h, used = C.c_void_p(), C.c_int32()
    check(lib.pantas_graph_load_cached(os.fsencode(pre + ".gfa"), os.fsencode(cache), 0, C.byref(h), C.byref(used)))
    assert used.value == 0
    lib.pantas_graph_free(h)


def test_short_walk_node_summary_vs_piece_compaction():
    """walk.h short_node_summary (what the short-walk kernel makes of a path node: kept?, how many compacted entries are
    neither '-' nor '*') against PieceAcc, the piece-by-piece restatement of :63-107, on random cs strings of the
    short-walk classes (':' runs, up to four '*', at most one '+' / '-') laid over random node lengths.  Round 4: and the tile geometry
    of the position-parallel walk (tile_pack, tile_rec_ordinal, tile_lane, tile_neighbour, tile_kept_index) against plain loops over
    randomly packed tiles and random kept masks."""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "hostsim_shortwalk")
    subprocess.check_call(CXX + ["-o", exe, os.path.join(ROOT, "tests/hostsim/hostsim_shortwalk.cpp")])
    r = run([exe, "400000"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok ") and " tiles" in r.stdout.splitlines()[0]


def test_stage_emitter_against_the_write_emitter():
    """call_host.h StageEmit (what the call stage kernels emit through, round 5: an item's own four records, then a chunk of 64 of a
    pool, then one of 4 096, then "evaluate again") against WriteEmit on made-up items of every size around those steps, with a
    pool, without one and with one that has run out: the same records and payload words wherever an item is staged."""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "hostsim_stage")
    subprocess.check_call(CXX + ["-o", exe, os.path.join(ROOT, "tests/hostsim/hostsim_stage.cpp")])
    r = run([exe])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert r.stdout.startswith("evaluated again ")


@pytest.mark.parametrize("seed", [5, 6])
def test_walk_source_around_the_short_walk_bounds(hostsim, tmp_path, seed):
    """tests/long_path_case.py: paths of 27 .. 40 nodes (the short-walk class ends at 31, at 29 with an indel), one
    '+' / '-' at every position class with mismatches around it, both strands -- harness (which dispatches lines to the
    short walks like the kernels) against the oracle, byte for byte; and with every line through the general walk."""
    import long_path_case
    pre = str(tmp_path / "lp")
    long_path_case.write_case(pre, seed, 4000)
    want = io.StringIO()
    augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    for env in ({}, {"HOSTSIM_BUCKET": "1"}, {"HOSTSIM_GENERAL_WALK": "1"}):
        r = subprocess.run([hostsim["augment"], pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, env={**os.environ, **env})
        assert r.returncode == 0, r.stderr
        assert r.stdout == want.getvalue(), env
