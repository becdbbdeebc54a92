"""Parity with the oracle at BASELINE.json sizes.

* `test_e2e_command_lines_vs_oracle`: the 6 000-gene / 2 M-line workload through the two COMMAND LINES -- engine GFA bytes
  == oracle bytes, CSV rows == oracle rows as a multiset.
* `test_whole_s3_vs_cpp_oracle`: ALL of S3 (human chr1 scale, 10 M lines) through `pantas augment` against the independent C++
  form of the oracle (oracle/augment_oracle.cpp, pinned to the reference's goldens by tests/test_oracle_cpp.py) on the host
  cores: the two augmented GFAs are the same file; then `pantas call --rca 3 --novel` on that file against
  oracle/call_oracle.cpp: the same rows (round 3).
* `test_whole_s4_and_s5_vs_cpp_oracles` (round 5; part of the default run): ALL of S4 -- 50 M lines, 2.5e8 segments, 23 GB of output --
  the same way, the GFA compared through the oracle's byte digest (every phase of the oracle runs on all cores and nothing is written:
  586 s became 160), then three more blocks of the read stream behind the first: S5, 200 M reads, 241 M event rows.  Until round 4 these
  were opt-in tests, run by hand once per round.
* `test_full_graph_slice_vs_oracle[s3|s4]`: the FULL human-chr1 / human whole-genome graph (2.5e8 segments, ids in the
  1e9 range, the dense name table, counter offsets beyond 2^30 words) with a slice of the read stream spread over every
  chromosome.  The Python oracle cannot hold such a graph, and it does not have to: what it prints for a segment or a
  link depends only on the reads that touch it.  tests/tools/gfa_lines.cpp cuts the touched part out of the input GFA
  (S lines of every segment a read path names, L lines between them); the oracle runs on that; the same cut of the
  engine's whole-genome output must equal the oracle's output byte for byte (S lines, L lines and the first-seen-ordered
  tail of links the graph does not have), and the engine's counters must be zero everywhere else.
"""
import collections
import io
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import augment_oracle
import call_oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu


def ram_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemTotal:"):
            return int(line.split()[1]) / 1e6
    return 0.0


def workdir(need_gb):
    for d in ("/dev/shm", tempfile.gettempdir()):
        try:
            if shutil.disk_usage(d).free > need_gb * 1e9:
                return tempfile.mkdtemp(prefix="pantas_parity_", dir=d)
        except OSError:
            pass
    pytest.skip(f"no directory with {need_gb} GB free")


@pytest.fixture(scope="module")
def gfa_lines():
    exe = os.path.join(ROOT, "build", "gfa_lines")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "tools", "gfa_lines.cpp"), "-lpthread"])
    return exe


def _heartbeat(label):
    """a line every 30 s under gpurun_out/ (a silent GPU-box command is taken for hung after 7 minutes)"""
    import threading
    import time
    beat = os.path.join(ROOT, "gpurun_out", "heartbeat.txt")
    stop = threading.Event()

    def run():
        while not stop.wait(30):
            if os.path.isdir(os.path.dirname(beat)):
                with open(beat, "a") as f:
                    f.write(f"{label} parity {time.strftime('%H:%M:%S')}\n")
    threading.Thread(target=run, daemon=True).start()
    return stop


def _append_read_blocks(pre, kw, first, last):
    """blocks [first, last) of the scale's read stream behind the GAF that is there (s5: four blocks of 50 M lines in ONE file)"""
    from pantas_amd.engine import Synth
    for b in range(first, last):
        Synth(seed=5, line_offset=b * kw["n_lines"], gaf_only=1, **kw).write(pre + f".b{b}")
        with open(pre + ".gaf", "ab") as out, open(pre + f".b{b}.gaf", "rb") as src:
            shutil.copyfileobj(src, out, 1 << 26)
        for ext in (".gaf", ".gfa", ".gtf"):
            if os.path.exists(pre + f".b{b}" + ext):
                os.unlink(pre + f".b{b}" + ext)


def _augment_vs_cpp_oracle(pre, oracle_threads, by_digest):
    """`pantas augment` on pre.gaf / pre.gfa -> pre.gpu.gfa, against oracle/augment_oracle.cpp on the host cores: the same file --
    byte by byte, or (whole-genome sizes: 23 GB) through the oracle's order-dependent digest of the bytes it WOULD write and the same
    digest of the engine's file (`--digest` / `--digest-file`; the oracle then formats its pieces on every core and prints nothing)"""
    import json
    import time
    exe = os.path.join(ROOT, "oracle", "bin", "augment_oracle_cpp")
    threads = str(min(os.cpu_count() or 1, oracle_threads))
    t0 = time.perf_counter()
    with open(pre + ".gpu.gfa", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    t1 = time.perf_counter()
    size = os.path.getsize(pre + ".gpu.gfa")
    res = {"gaf_bytes": os.path.getsize(pre + ".gaf"), "engine_augment_wall_s": t1 - t0, "bytes": size}
    if by_digest:
        o = subprocess.run([exe, "--threads", threads, "--digest", pre + ".gaf", pre + ".gfa"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert o.returncode == 0, o.stderr.decode()[-2000:]
        t2 = time.perf_counter()
        stats = json.loads(o.stderr.decode().strip().splitlines()[-1])
        got = json.loads(subprocess.run([exe, "--threads", threads, "--digest-file", pre + ".gpu.gfa"], capture_output=True, check=True).stdout)
        assert size > 10**9 and got == stats["digest"], (got, stats["digest"])
        res.update({"aug_gfa_equals_cpp_oracle": True, "compared": "digest of the bytes (oracle --digest, engine file --digest-file)", "digest": got,
                    "digest_file_s": time.perf_counter() - t2})
    else:
        with open(pre + ".or.gfa", "wb") as f:
            o = subprocess.run([exe, "--threads", threads, pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE)
        assert o.returncode == 0, o.stderr.decode()[-2000:]
        t2 = time.perf_counter()
        stats = json.loads(o.stderr.decode().strip().splitlines()[-1])
        assert size == os.path.getsize(pre + ".or.gfa") > 10**9
        with open(pre + ".gpu.gfa", "rb") as x, open(pre + ".or.gfa", "rb") as y:
            while True:
                a, b = x.read(1 << 26), y.read(1 << 26)
                assert a == b
                if not a:
                    break
        os.unlink(pre + ".or.gfa")
        res.update({"aug_gfa_equals_cpp_oracle": True, "compared": "bytes"})
    rej = [l for l in o.stderr.decode().splitlines() if l.startswith("Rejected alignments:")]
    assert rej and rej[0] in r.stderr.decode()
    res.update({"cpp_oracle_wall_s": t2 - t1, "cpp_oracle_stats": stats})
    return res


def _whole_workload_vs_cpp_oracle(scale, need_gb, oracle_threads, check_augment=True, by_digest=False):
    """`pantas augment` on the whole workload of a BASELINE scale against oracle/augment_oracle.cpp on the host cores: one file;
    then `pantas call --rca 3 --novel` on it against oracle/call_oracle.cpp: the same rows."""
    import json
    import bench
    from pantas_amd.engine import Synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    tmp = workdir(need_gb)
    stop = _heartbeat(scale)
    try:
        pre = os.path.join(tmp, scale)
        desc, kw, *rest = bench.SCALES[scale]
        Synth(seed=5, **kw).write(pre)
        _append_read_blocks(pre, kw, 1, rest[0] if rest else 1)
        res = {"scale": scale, "workload": desc}
        if check_augment:
            res.update(_augment_vs_cpp_oracle(pre, oracle_threads, by_digest))
        else:
            with open(pre + ".gpu.gfa", "wb") as f:
                r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
        os.unlink(pre + ".gaf")
        res["call"] = _call_vs_cpp_oracle(pre, pre + ".gpu.gfa", oracle_threads)
        if os.environ.get("PANTAS_PARITY_JSON"):
            json.dump(res, open(os.environ["PANTAS_PARITY_JSON"].format(scale=scale), "w"), indent=1)
    finally:
        stop.set()
        shutil.rmtree(tmp, ignore_errors=True)


CALL_FLAGS = ["--rca", "3", "--novel"]


def _call_vs_cpp_oracle(pre, aug_gfa, oracle_threads, flags=CALL_FLAGS):
    """`pantas call` on an augmented GFA against oracle/call_oracle.cpp (pinned to the reference's goldens by
    tests/test_oracle_cpp.py) on the host cores: the two CSVs hold the same rows -- compared as multisets through the oracle's
    128-bit order-independent digest of the normalised rows, and line by line after sorting while the files
    are small enough to sort in Python."""
    import json
    import time
    exe = os.path.join(ROOT, "oracle", "bin", "call_oracle_cpp")
    threads = str(min(os.cpu_count() or 1, oracle_threads))
    t0 = time.perf_counter()
    with open(pre + ".gpu.csv", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, aug_gfa, pre + ".gtf"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    t1 = time.perf_counter()
    small = os.path.getsize(pre + ".gpu.csv") < 1.5e9
    with open(pre + ".or.csv" if small else os.devnull, "wb") as f:
        o = subprocess.run([exe, "--threads", threads, "--digest", *([] if small else ["--no-out"]), *flags, aug_gfa, pre + ".gtf"],
                           stdout=f, stderr=subprocess.PIPE)
    assert o.returncode == 0, o.stderr.decode()[-2000:]
    t2 = time.perf_counter()
    want = json.loads(o.stderr.decode().strip().splitlines()[-1])
    got = json.loads(subprocess.run([exe, "--threads", threads, "--digest-csv", pre + ".gpu.csv"], capture_output=True, check=True).stdout)
    keys = ("rows", "s1", "s2", "x")
    assert want["rows"] > 100_000
    assert {k: got[k] for k in keys} == {k: want[k] for k in keys}, (got, want)
    if small:
        mine, ref = open(pre + ".gpu.csv").read().splitlines(), open(pre + ".or.csv").read().splitlines()
        assert mine[0] == ref[0] == call_oracle.HEADER
        norm = call_oracle.normalise_row
        assert sorted(map(norm, mine[1:])) == sorted(map(norm, ref[1:]))
        kinds = {tuple(l.split(",", 2)[:2]) for l in ref[1:]}
        assert {("ES", "annotated"), ("IR", "annotated"), ("ES", "novel"), ("CE", "novel"), ("IR", "novel")} <= kinds, kinds
        os.unlink(pre + ".or.csv")
    res = {"flags": " ".join(flags), "events_csv_equals_cpp_oracle_as_row_multiset": True, "sorted_lines_compared": small, "rows": want["rows"],
           "csv_bytes": os.path.getsize(pre + ".gpu.csv"), "digest": {k: want[k] for k in keys},
           "engine_call_wall_s": t1 - t0, "cpp_oracle_call_wall_s": t2 - t1, "cpp_oracle_stats": want}
    os.unlink(pre + ".gpu.csv")
    return res


def test_whole_s3_vs_cpp_oracle():
    _whole_workload_vs_cpp_oracle("s3", 12, 32)


# the S5 leg below runs in the default suite when the S4 leg in front of it was done within this many seconds (a GPU box of
# ordinary speed: about 230; the driver gives the whole `-m gpu` run 900 s); PANTAS_WHOLE_S5=1 runs it whatever the clock says,
# PANTAS_WHOLE_S5=0 never
S5_IF_S4_WITHIN_S = 300.0


def test_whole_s4_and_s5_vs_cpp_oracles():
    """BASELINE.json configs[3] (one replicate) and configs[4] AT THEIR SIZES, in the default `-m gpu` run since round 5.

    S4 -- the human whole-genome graph (2.5e8 segments, 19 GB of GFA) and one replicate of 50 M reads: `pantas augment`, its 23 GB
    file against oracle/augment_oracle.cpp through the byte digest (the oracle formats on every core and writes nothing: 10 minutes
    became 2), then `pantas call --rca 3 --novel` on it against oracle/call_oracle.cpp (44.6 M rows, multiset digest).
    S5 -- the same graph, 200 M reads (three more blocks of the read stream behind the first): `pantas augment` again (four
    resident batches), `pantas call --rca 3 --novel` against oracle/call_oracle.cpp (241 M rows); the augment kernels at this
    graph size are what the S4 leg checked byte for byte."""
    import json
    import time
    import bench
    from pantas_amd.engine import Synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    tmp = workdir(130)
    stop = _heartbeat("s4+s5")
    t_start = time.perf_counter()
    try:
        pre = os.path.join(tmp, "s4")
        desc, kw = bench.SCALES["s4"][:2]
        Synth(seed=5, **kw).write(pre)
        res = {"scale": "s4", "workload": desc, "generate_s": time.perf_counter() - t_start}
        res.update(_augment_vs_cpp_oracle(pre, 64, by_digest=True))
        res["call"] = _call_vs_cpp_oracle(pre, pre + ".gpu.gfa", 64)
        res["leg_s"] = time.perf_counter() - t_start
        if os.environ.get("PANTAS_PARITY_JSON"):
            json.dump(res, open(os.environ["PANTAS_PARITY_JSON"].format(scale="s4"), "w"), indent=1)
        want_s5 = os.environ.get("PANTAS_WHOLE_S5")
        if want_s5 == "0" or (want_s5 is None and res["leg_s"] > S5_IF_S4_WITHIN_S):
            pytest.skip(f"S4 checked (GFA digest and {res['call']['rows']} rows equal) in {res['leg_s']:.0f} s; the S5 leg is left out on "
                        f"this box to keep the run inside the driver's limit (PANTAS_WHOLE_S5=1 runs it)")
        t5 = time.perf_counter()
        os.unlink(pre + ".gpu.gfa")
        if shutil.disk_usage(tmp).free < 110e9:       # 30 GB of reads + 23 GB of GFA + 44 GB of rows beside the graph
            pytest.skip(f"S4 checked in {res['leg_s']:.0f} s; no room for the S5 leg's files in {tmp}")
        desc5, kw5, blocks = bench.SCALES["s5"]
        assert kw5 == kw
        _append_read_blocks(pre, kw, 1, blocks)
        res5 = {"scale": "s5", "workload": desc5, "generate_s": time.perf_counter() - t5, "gaf_bytes": os.path.getsize(pre + ".gaf")}
        t0 = time.perf_counter()
        with open(pre + ".gpu.gfa", "wb") as f:
            r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        res5["engine_augment_wall_s"] = time.perf_counter() - t0
        os.unlink(pre + ".gaf")
        res5["call"] = _call_vs_cpp_oracle(pre, pre + ".gpu.gfa", 64)
        res5["leg_s"] = time.perf_counter() - t5
        if os.environ.get("PANTAS_PARITY_JSON"):
            json.dump(res5, open(os.environ["PANTAS_PARITY_JSON"].format(scale="s5"), "w"), indent=1)
    finally:
        stop.set()
        shutil.rmtree(tmp, ignore_errors=True)


def test_e2e_command_lines_vs_oracle():
    """tools/e2e_compare.py's check as a test: 6 000 genes, 2 M lines, 48 k junction links (a Drosophila-scale read set)"""
    from pantas_amd.engine import Synth
    tmp = workdir(6)
    try:
        pre = os.path.join(tmp, "w")
        Synth(seed=21, n_genes=6000, n_lines=2_000_000, base_id=10**7, intron_long_max=200, zipf_a=1.0).write(pre)
        flags = ["--rca", "3", "--novel"]
        with open(pre + ".gpu.gfa", "wb") as f:
            r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        augment_stderr = r.stderr.decode()
        with open(pre + ".gpu.csv", "wb") as f:
            r = subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, pre + ".gpu.gfa", pre + ".gtf"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        with open(pre + ".or.gfa", "w") as f:
            a = augment_oracle.augment(pre + ".gaf", pre + ".gfa", f)
        assert open(pre + ".gpu.gfa", "rb").read() == open(pre + ".or.gfa", "rb").read()
        assert f"Rejected alignments: {a.rejected}" in augment_stderr
        rows = call_oracle.call(pre + ".or.gfa", pre + ".gtf", None, rca=3, novel=True)
        norm = call_oracle.normalise_row
        mine = open(pre + ".gpu.csv").read().splitlines()
        assert mine[0] == call_oracle.HEADER
        assert len(rows) > 100_000
        assert collections.Counter(map(norm, mine[1:])) == collections.Counter(map(norm, rows))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("scale,slice_lines", [("s3", 200_000), ("s4", 200_000)])
def test_full_graph_slice_vs_oracle(gfa_lines, scale, slice_lines):
    import bench
    from pantas_amd.engine import Graph, Synth
    desc, kw, *_ = bench.SCALES[scale]
    kw = dict(kw)
    if scale == "s4" and ram_gb() < 400:
        pytest.skip("the whole-genome graph needs a host with 400 GB of memory")
    tmp = workdir(60 if scale == "s4" else 8)
    try:
        pre = os.path.join(tmp, scale)
        # a slice from the middle of the read stream (lines 10 M .. of the scale's stream), p_indel raised so that the
        # rare-key path and its first-seen order are exercised on high ids as well
        kw.update(n_lines=slice_lines, line_offset=10_000_000, p_indel=0.15)
        syn = Synth(seed=7, **kw)
        syn.write(pre)
        g = Graph(text=syn.gfa).to_device(0)
        gs = g.stats
        if scale == "s4":
            assert gs.segments > 2 * 10**8 and 4 * gs.segments + gs.link_slots > 2**30     # counter offsets beyond 2^30 words
        g.augment_reset()
        b = g.load_gaf_mem(syn.gaf, first_line=10_000_000)
        g.launch(b)
        g.finish()
        with open(pre + ".aug.gfa", "wb") as f:
            g.write_gfa(f.fileno())
        # segments the reads name
        names = set()
        with open(pre + ".gaf") as f:
            for line in f:
                p = line.split("\t")[5]
                if p != "*":
                    names.update(p.replace("<", ">").split(">")[1:])
        with open(pre + ".names", "w") as f:
            f.write("\n".join(sorted(names)) + "\n")
        assert len(names) > 100_000
        ids = np.array(sorted(int(x) for x in names), dtype=np.int64)
        if scale == "s4":
            assert ids.min() >= 10**9 and ids.max() - ids.min() > 2 * 10**8       # spread over the whole id range
        with open(pre + ".cut.gfa", "wb") as f:
            subprocess.check_call([gfa_lines, pre + ".gfa", pre + ".names"], stdout=f)
        with open(pre + ".cut.aug.gfa", "wb") as f:
            subprocess.check_call([gfa_lines, pre + ".aug.gfa", pre + ".names"], stdout=f)
        want = io.StringIO()
        a = augment_oracle.augment(pre + ".gaf", pre + ".cut.gfa", want)
        got = open(pre + ".cut.aug.gfa").read()
        assert b.stats.rejected == a.rejected
        assert got == want.getvalue(), "the touched S / L lines or the novel-link tail differ from the oracle's"
        assert got.count("ID:Z:N") > 100 and got.count("IL:Z:") > 1000
        # nothing outside the touched part carries a count
        nc, il0, oll, rc = g.read_counters()
        oracle_nc = sum(a.nc.values())
        assert int(nc.sum(dtype=np.int64)) == oracle_nc
        assert int(np.count_nonzero(nc)) == sum(1 for v in a.nc.values() if v)     # with the equal lines above: no count anywhere else
        assert int(il0.sum(dtype=np.int64)) == sum(d.get(0, 0) for d in a.il.values())
        assert int(oll.sum(dtype=np.int64)) == sum(d.get(a.seg_len[n], 0) for n, d in a.ol.items())
        assert int(rc.sum(dtype=np.int64)) + sum(int(l.split("RC:i:")[1].split("\t")[0]) for l in got.splitlines() if "ID:Z:N" in l) == sum(a.rc.values())
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
