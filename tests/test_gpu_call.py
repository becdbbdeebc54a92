"""Parity of the HIP call path (through the C ABI) with the reference goldens and the oracle."""
import collections
import io
import json
import os
import subprocess
import sys
import tempfile

import pytest

import augment_oracle
import call_oracle
from conftest import CASES, GOLDEN, ROOT
from test_oracle_golden import _flags_to_opts, check_rows

pytestmark = pytest.mark.gpu


def engine_call(gfa, gtf, flags):
    from pantas_amd.engine import Graph
    o = _flags_to_opts(flags)
    g = Graph(gfa)
    g.call_prepare(gtf, 0)
    with tempfile.TemporaryFile() as f:
        st = g.call(f.fileno(), -1, rca=o["rca"], novel=o["novel"], no_annotated=o["no_annotated"],
                    events=o["events"], isw=o["isw"], minintronsize=o["minintronsize"], junction=o["junction"])
        f.seek(0)
        rows = f.read().decode().splitlines()
    assert rows[0] == call_oracle.HEADER
    assert st.events == len(rows) - 1
    return rows[1:]


@pytest.mark.parametrize("case", CASES)
def test_golden_rows(case):
    d = os.path.join(GOLDEN, case)
    for gc in json.load(open(f"{d}/expected.json"))["calls"]:
        check_rows(engine_call(f"{d}/expected.aug.gfa", f"{d}/in.gtf", gc["flags"]), gc)


def test_cli_streams():
    d = os.path.join(GOLDEN, "synth_a")
    flags = ["--rca", "3", "--novel", "--minintronsize", "25"]
    r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, f"{d}/expected.aug.gfa", f"{d}/in.gtf"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    rows = r.stdout.splitlines()
    assert rows[0] == call_oracle.HEADER
    gc = [c for c in json.load(open(f"{d}/expected.json"))["calls"] if c["flags"] == flags][0]
    check_rows(rows[1:], gc)
    err = r.stderr.splitlines()
    assert err[0].endswith("Parsing GFA..") and err[1].endswith("Parsing GTF..")


def test_crash_inputs():
    from pantas_amd.engine import PantasError
    cd = os.path.join(GOLDEN, "crashes")
    for c in json.load(open(os.path.join(GOLDEN, "crashes.json"))):
        if c["stage"] != "call":
            continue
        with pytest.raises(PantasError) as ei:
            engine_call(f"{cd}/{c['call_input']}", f"{cd}/{c['name']}.gtf", c["flags"])
        assert ei.value.kind == c["exception"], c["name"]


def test_two_failing_passes_report_the_first(tmp_path):
    """The passes run side by side (call.hip run_call); the reference runs them in turn and dies in the first that fails.  A transcript the
    GTF lacks (annotated pass) AND a novel link without RC (novel-link pass): the error is the annotated pass's, whichever lane is
    through first -- and the novel-link pass's without the annotated pass.  The oracle says which is which."""
    import re
    from pantas_amd.engine import PantasError
    d = os.path.join(GOLDEN, "synth_a")
    gfa = open(f"{d}/expected.aug.gfa").read().splitlines()
    k = next(i for i, l in enumerate(gfa) if l.startswith("L") and "ID:Z:N" in l)
    gfa[k] = re.sub(r"\tRC:i:\d+", "", gfa[k])
    gtf = [l for l in open(f"{d}/in.gtf").read().splitlines() if 'transcript_id "G10T1"' not in l]
    open(tmp_path / "a.gfa", "w").write("\n".join(gfa) + "\n")
    open(tmp_path / "a.gtf", "w").write("\n".join(gtf) + "\n")
    for flags, word in ((["--novel", "--rca", "0"], "transcript"), (["--novel", "--rca", "0", "--no-annotated"], "RC")):
        with pytest.raises(augment_oracle.ReferenceCrash) as want:
            call_oracle.call(str(tmp_path / "a.gfa"), str(tmp_path / "a.gtf"), None, **_flags_to_opts(flags))
        assert str(want.value).startswith("KeyError") and word in str(want.value)
        for _ in range(4):
            with pytest.raises(PantasError) as got:
                engine_call(str(tmp_path / "a.gfa"), str(tmp_path / "a.gtf"), flags)
            assert got.value.kind == "KeyError" and word in str(got.value), str(got.value)


@pytest.mark.parametrize("seed,genes,lines,kw,flags", [
    (31, 120, 30000, {}, ["--rca", "2", "--novel", "--minintronsize", "30"]),
    (32, 60, 30000, dict(bubble_rate=0.15, p_novel_skip=0.1, p_novel_ce=0.06, p_intron=0.15, p_novel_ss=0.1),
     ["--rca", "1", "--novel", "--isw", "3", "--minintronsize", "10"]),
    (33, 80, 20000, dict(bubble_rate=0.0), ["--events", "IR", "SS", "--novel", "--rca", "0", "--minintronsize", "1"]),
])
def test_augment_then_call_vs_oracle(tmp_path, seed, genes, lines, kw, flags):
    """Whole hot path on the GPU (augment, then call on its output) against the oracle's rows."""
    from pantas_amd import synth
    from pantas_amd.engine import Graph
    pre = str(tmp_path / "s")
    synth.write_all(pre, seed=seed, n_genes=genes, n_lines=lines, **kw)
    g = Graph(pre + ".gfa").to_device(0)
    with open(pre + ".aug.gfa", "wb") as f:
        g.augment(pre + ".gaf", 20, f.fileno())
    want_gfa = io.StringIO()
    augment_oracle.augment(pre + ".gaf", pre + ".gfa", want_gfa)
    assert open(pre + ".aug.gfa").read() == want_gfa.getvalue()
    rows = engine_call(pre + ".aug.gfa", pre + ".gtf", flags)
    want = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, **_flags_to_opts(flags))
    assert len(want) > 100
    assert collections.Counter(map(call_oracle.normalise_row, rows)) == collections.Counter(map(call_oracle.normalise_row, want))


def test_call_is_deterministic_and_ordered():
    """Two runs give byte-identical CSV (fixed emission order, no atomics in the output path)."""
    d = os.path.join(GOLDEN, "synth_b")
    a = engine_call(f"{d}/expected.aug.gfa", f"{d}/in.gtf", ["--rca", "2", "--novel", "--minintronsize", "40"])
    b = engine_call(f"{d}/expected.aug.gfa", f"{d}/in.gtf", ["--rca", "2", "--novel", "--minintronsize", "40"])
    assert a == b


def test_fused_equals_text_round_trip(tmp_path):
    """augment -> call on the same handle (no GFA text) == call on the written augmented GFA, byte for byte."""
    from pantas_amd.engine import Graph, Synth
    syn = Synth(seed=5, n_genes=400, n_lines=60000)
    pre = str(tmp_path / "w")
    syn.write(pre)
    g = Graph(pre + ".gfa").to_device(0)
    with open(pre + ".aug.gfa", "wb") as f:
        g.augment(pre + ".gaf", 20, f.fileno())
    flags = dict(rca=2, novel=True, minintronsize=30)
    g.call_prepare_fused(pre + ".gtf")
    with tempfile.TemporaryFile() as f:
        g.call(f.fileno(), -1, **flags)
        f.seek(0)
        fused = f.read()
    g2 = Graph(pre + ".aug.gfa")
    g2.call_prepare(pre + ".gtf", 0)
    with tempfile.TemporaryFile() as f:
        st = g2.call(f.fileno(), -1, **flags)
        f.seek(0)
        text = f.read()
    assert st.events > 500
    assert fused == text
    want = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, **flags)
    norm = call_oracle.normalise_row
    assert collections.Counter(map(norm, text.decode().splitlines()[1:])) == collections.Counter(map(norm, want))


@pytest.mark.parametrize("fan", [10, 40, 300, 5000])
def test_intronic_search_beyond_local_frontier(tmp_path, fan, monkeypatch):
    from hub_case import write_hub_case
    pre = str(tmp_path / "hub")
    write_hub_case(pre, fan)
    flags = ["--novel", "--rca", "0"]
    want = sorted(call_oracle.call(pre + ".gfa", pre + ".gtf", None, **_flags_to_opts(flags)))
    assert sorted(engine_call(pre + ".gfa", pre + ".gtf", flags)) == want
    # the search's three rooms (call_device.h frontier_search): the lane's 64 words, the workgroup's LDS, a slice of the arena --
    # without the LDS, and with so little of it that the search outgrows it on its way
    for words in ("0", "200"):
        monkeypatch.setenv("PANTAS_CALL_LDS_FRONTIER", words)
        assert sorted(engine_call(pre + ".gfa", pre + ".gtf", flags)) == want


def test_sharded_call_driver_nccl_single_rank(tmp_path):
    """`python -m pantas_amd.distributed call ...` with the real library over the RCCL backend (one rank here)."""
    import socket
    d = os.path.join(GOLDEN, "synth_a")
    flags = ["--rca", "3", "--novel", "--minintronsize", "25"]
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    out = tmp_path / "events.csv"
    with open(out, "wb") as f:
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pantas_amd.distributed",
                            "call", *flags, f"{d}/expected.aug.gfa", f"{d}/in.gtf"], stdout=f, stderr=subprocess.PIPE,
                           text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = open(out).read().splitlines()
    assert rows[0] == call_oracle.HEADER
    gc = [c for c in json.load(open(f"{d}/expected.json"))["calls"] if c["flags"] == flags][0]
    check_rows(rows[1:], gc)


def test_release_device_after_the_kernels():
    """pantas_call_opts.release_device (ABI 5): the same rows, and the handle has to be prepared again before another run"""
    from pantas_amd.engine import Graph, PantasError
    d = os.path.join(GOLDEN, "synth_c")
    g = Graph(f"{d}/expected.aug.gfa")
    g.call_prepare(f"{d}/in.gtf", 0)
    with tempfile.TemporaryFile() as f:
        g.call(f.fileno(), -1, rca=2, novel=True)
        f.seek(0)
        want = f.read()
    with tempfile.TemporaryFile() as f:
        g.call(f.fileno(), -1, rca=2, novel=True, release_device=True)
        f.seek(0)
        assert f.read() == want and want.count(b"\n") > 10
    with pytest.raises(PantasError):
        g.call(-1, -1, rca=2, novel=True)
    g.call_prepare(f"{d}/in.gtf", 0)
    with tempfile.TemporaryFile() as f:
        g.call(f.fileno(), -1, rca=2, novel=True)
        f.seek(0)
        assert f.read() == want
