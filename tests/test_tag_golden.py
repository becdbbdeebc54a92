"""IL / OL tag values at the edge of `build_attrs` (scripts/call.py:25-69): the position clustering of :33-66 is never read,
but it raises -- IndexError for an entry without a count, ZeroDivisionError for a cluster whose counts sum to 0, ValueError
for a piece that is no int.  tests/golden/tags/cases.json holds what the REAL reference did with every value
(tools/make_golden_r5.py): the reference-augmented micro graph with one tag of one line replaced.  Both oracles, the
product's host parser (CPU harness) and the engine on the GPU (whose loader reads tags on the device and hands anything but
the plain form to the host parser) are held to it."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

import augment_oracle
import call_oracle
from conftest import GOLDEN, ROOT
from test_oracle_golden import _flags_to_opts

META = json.load(open(os.path.join(GOLDEN, "tags", "cases.json")))
CASES = META["cases"]
IDS = [f"{c['line'].split()[0]}-{c['tag']}:{c['value'] or 'empty'}" for c in CASES]
BASE = open(os.path.join(GOLDEN, META["base"])).read().splitlines()
GTF = os.path.join(GOLDEN, META["gtf"])


def tagged(c, path):
    """the base graph with c's tag on c's line (the tag of that name replaced, else appended)"""
    out, hit = [], 0
    for l in BASE:
        if l.startswith(c["line"]):
            hit += 1
            l = "\t".join([t for t in l.split("\t") if not t.startswith(c["tag"] + ":")] + [f"{c['tag']}:Z:{c['value']}"])
        out.append(l)
    assert hit == 1
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return path


def rows_ok(rows):
    assert sorted(call_oracle.normalise_row(r) for r in rows) == META["rows"]


def test_the_table_covers_every_outcome():
    seen = {c["exception"] for c in CASES}
    assert seen == {None, "IndexError", "ZeroDivisionError", "ValueError"}
    assert sum(1 for c in CASES if c["rc"] == 0) >= 10 and len(CASES) >= 40


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_python_oracle(c, tmp_path):
    p = tagged(c, str(tmp_path / "g.gfa"))
    if c["rc"]:
        with pytest.raises(augment_oracle.ReferenceCrash) as ei:
            call_oracle.call(p, GTF, None, **_flags_to_opts(META["flags"]))
        assert ei.value.kind == c["exception"]
    else:
        rows_ok(call_oracle.call(p, GTF, None, **_flags_to_opts(META["flags"])))


@pytest.fixture(scope="module")
def cpp_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return os.path.join(ROOT, "oracle", "bin", "call_oracle_cpp")


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_cpp_oracle(cpp_oracle, c, tmp_path):
    p = tagged(c, str(tmp_path / "g.gfa"))
    for threads in (1, 3):
        r = subprocess.run([cpp_oracle, "--threads", str(threads), *META["flags"], p, GTF], capture_output=True, text=True)
        if c["engine"] == "refuses":       # (integers beyond 64 bits: the C++ form says so instead of computing with them)
            assert r.returncode == 1 and r.stdout == "" and r.stderr.startswith("Unsupported:")
        elif c["rc"]:
            assert r.returncode == 1 and r.stderr.strip().splitlines()[-1].startswith(c["exception"] + ":"), r.stderr
        else:
            assert r.returncode == 0, r.stderr
            rows_ok(r.stdout.splitlines()[1:])


def check_product(c, rc, stdout, kind):
    if c["engine"] == "refuses":
        assert rc and stdout == "" and kind == "Unsupported", "a refused input fails loudly and prints nothing"
        return
    if c["rc"]:
        assert rc and kind == c["exception"] and stdout == "", (kind, c["exception"])
    else:
        assert rc == 0, kind
        rows_ok(stdout.splitlines()[1:])


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_product_host_parser(hostsim, c, tmp_path):
    """pantas_amd/csrc/call_host.cpp compiled for the CPU (tests/hostsim)"""
    from conftest import BUILD
    exe = os.path.join(BUILD, "hostsim_call")
    p = tagged(c, str(tmp_path / "g.gfa"))
    r = subprocess.run([exe, *META["flags"], p, GTF], capture_output=True, text=True)
    check_product(c, r.returncode, r.stdout, r.stderr.strip().splitlines()[-1].split(":")[0] if r.returncode else "")


@pytest.mark.gpu
def test_engine(tmp_path):
    """through the C ABI on the GPU: the device tag scan (tokenize.hip gfa_tags_kernel) takes the plain values and sends the
    others to the host parser"""
    from pantas_amd.engine import Graph, PantasError
    for c in CASES:
        p = tagged(c, str(tmp_path / "g.gfa"))
        rc, out, kind = 0, "", ""
        try:
            g = Graph(p)
            g.call_prepare(GTF, 0)
            with tempfile.TemporaryFile() as f:
                g.call(f.fileno(), -1, **_flags_to_opts(META["flags"]))
                f.seek(0)
                out = f.read().decode()
        except PantasError as e:
            rc, kind = 1, e.kind
        check_product(c, rc, out, kind)


@pytest.mark.gpu
def test_command_line(tmp_path):
    """`pantas call` itself (device loader first) on one value of every outcome"""
    picked = {}
    for c in CASES:
        picked.setdefault((c["exception"], c["engine"]), c)
    for c in picked.values():
        p = tagged(c, str(tmp_path / "g.gfa"))
        r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *META["flags"], p, GTF], capture_output=True, text=True, cwd=ROOT)
        kind = r.stderr.strip().splitlines()[-1].split(":")[0] if r.returncode else ""
        check_product(c, r.returncode, r.stdout, kind)
