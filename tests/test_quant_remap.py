"""`pantas quant` / `pantas remap` mirrors against outputs of the real reference scripts (tools/make_golden_quant.py
ran scripts/quantify.py and scripts/remap.py in the build container; tests/golden/quant holds inputs and outputs)."""
import gzip
import io
import json
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

Q = os.path.join(GOLDEN, "quant")
META = json.load(open(os.path.join(Q, "expected.json")))
C1 = [os.path.join(Q, f"c1_r{i}.csv") for i in (1, 2, 3)]
C2 = [os.path.join(Q, f"c2_r{i}.csv") for i in (1, 2, 3)]


def expected(name):
    return gzip.open(os.path.join(Q, name + ".gz"), "rt").read()


@pytest.mark.parametrize("how", ["native", "python"])
@pytest.mark.parametrize("case", META["quant"], ids=lambda c: c["expected"])
def test_quantify_bytes(case, how):
    """the native join (csrc/quant.cpp, what the command line runs) and the interpreted mirror: the reference's bytes"""
    from pantas_amd import quantify
    n = case["replicates"]
    out = io.StringIO()
    a = quantify.build_parser().parse_args([*case["flags"], "--c1", *C1[:n], "--c2", *C2[:n]])
    (quantify.quantify if how == "native" else quantify.quantify_python)(a.c1, a.c2, a.w, a.both, out)
    assert out.getvalue() == expected(case["expected"])
    assert case["rows"] > 50


def test_native_quant_equals_the_mirror_on_generated_replicates(tmp_path):
    """three + three replicates of `call --rca 3 --novel` events on one graph (the C++ oracles make them: ~10^5 rows each, every
    event kind, events shared by some replicates and missing from others): the native join prints what the interpreted mirror
    prints, with and without --both and -w"""
    import bench  # noqa: F401  (path set-up)
    from pantas_amd import quantify
    from pantas_amd.engine import Synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    kw = dict(n_genes=1500, n_lines=250_000, base_id=10**7, intron_long_max=200, zipf_a=1.0)
    pre = str(tmp_path / "g")
    Synth(seed=9, **kw).write(pre)
    csvs = []
    for r in range(6):
        gaf = pre + f".r{r}"
        Synth(seed=9, line_offset=(r + 1) * kw["n_lines"], gaf_only=1, **kw).write(gaf)
        with open(gaf + ".aug.gfa", "wb") as f:
            subprocess.run([os.path.join(ROOT, "oracle", "bin", "augment_oracle_cpp"), "--threads", "8", gaf + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.DEVNULL, check=True)
        with open(gaf + ".csv", "wb") as f:
            subprocess.run([os.path.join(ROOT, "oracle", "bin", "call_oracle_cpp"), "--threads", "8", "--rca", "3", "--novel", gaf + ".aug.gfa", pre + ".gtf"],
                           stdout=f, stderr=subprocess.DEVNULL, check=True)
        os.unlink(gaf + ".aug.gfa")
        csvs.append(gaf + ".csv")
    rows = sum(1 for _ in open(csvs[0]))
    assert rows > 20_000
    for w, both in ((0, False), (0, True), (5, False), (-1, False)):
        a, b = io.StringIO(), io.StringIO()
        quantify.quantify(csvs[:3], csvs[3:], w, both, a)
        quantify.quantify_python(csvs[:3], csvs[3:], w, both, b)
        assert a.getvalue() == b.getvalue(), (w, both)
        assert a.getvalue().count("\n") > 1000
    kinds = {l.split(",", 1)[0] for l in a.getvalue().splitlines()[1:]}
    assert kinds == {"ES", "CE", "IR", "A3", "A5"}


def test_quantify_cli_and_unequal_replicates():
    r = subprocess.run([sys.executable, "-m", "pantas_amd.quantify", "--both", "--c1", *C1, "--c2", *C2], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and r.stdout == expected("quant_1_3.csv")
    r = subprocess.run([sys.executable, "-m", "pantas_amd.quantify", "--c1", *C1, "--c2", *C2[:2]], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == META["unequal"]["returncode"] and r.stdout == META["unequal"]["stdout"]


@pytest.mark.parametrize("how", ["native", "python"])
@pytest.mark.parametrize("case", META["remap"], ids=lambda c: c["expected"])
def test_remap_bytes(case, how, tmp_path):
    """the native form (csrc/remap.cpp, what the command line runs) and the interpreted mirror: the reference's bytes"""
    from pantas_amd import remap
    src = str(tmp_path / case["input"])
    open(src, "w").write(expected(case["input"]))
    out = io.StringIO()
    a = remap.build_parser().parse_args([*case["flags"], src, os.path.join(Q, "anno.gtf")])
    (remap.remap if how == "native" else remap.remap_python)(a.CSV, a.GTF, a.min_intron_size, out)
    assert out.getvalue() == expected(case["expected"])


def test_native_remap_equals_the_mirror_on_perturbed_rows(tmp_path):
    """rows of a golden quantification with their fields disturbed one at a time (names without a reference copy, exon indices out of
    range or negative, a third index, imprecise haplotype copies on novel rows, swapped junction lengths, dpsi spellings python's
    float() takes, CRLF line ends, a row too short / too long, GTF lines without transcript_id): the native form prints what the mirror
    prints or raises what it raises, on one thread's worth of rows and on a file large enough to be cut into chunks"""
    import random
    from pantas_amd import remap
    rows = expected("quant_0_3.csv").splitlines()
    header, data = rows[0], rows[1:]
    gtf = os.path.join(Q, "anno.gtf")
    rng = random.Random(5)

    def run(fn, path, gtf_path=gtf, i=100):
        out = io.StringIO()
        try:
            fn(path, gtf_path, i, out)
            return ("ok", out.getvalue())
        except Exception as e:      # noqa: BLE001  (the class is what is compared)
            return (type(e).__name__, None)

    def mutate(fields):
        k = rng.randrange(14)
        f = list(fields)
        j = rng.choice([5, 6, 7])
        if k == 0: f[j] = "?"
        elif k == 1: f[j] = f[j].replace("_R", "_H")
        elif k == 2: f[j] = f[j] + ".3"
        elif k == 3: f[j] = f[j].split(".")[0] + ".0"
        elif k == 4: f[j] = f[j].split(".")[0] + ".-1"
        elif k == 5: f[j] = f[j].split(".")[0] + ".999"
        elif k == 6: f[j] = f[j].split(".")[0] + ".x"
        elif k == 7: f[15] = rng.choice(["1e-3", " 0.5 ", "-0.0", "nan", "-inf", "1_0.5", "+.5", "5.", "abc", "1__0", ""])
        elif k == 8: f[1] = rng.choice(["annotated", "novel"]); f[0] = rng.choice(["ES", "A3", "A5", "IR", "CE", ""])
        elif k == 9: f[5], f[6] = f[6], f[5]
        elif k == 10: f[j] = "x_R1_"
        elif k == 11: f[j] = ""
        elif k == 12: f[2] = "chr:1"
        else: f[j] = "noUnderscoreR.1.2"
        return f

    # single rows: every disturbance alone, result or exception class equal
    for n in range(400):
        f = mutate(rng.choice(data).split(","))
        p = str(tmp_path / "one.csv")
        open(p, "w").write(header + "\n" + ",".join(f) + "\n")
        assert run(remap.remap, p) == run(remap.remap_python, p), f
    for bad in (["a"] * 15, ["a"] * 17, ["a"] * 7, [""]):
        p = str(tmp_path / "one.csv")
        open(p, "w").write(header + "\n" + data[0] + "\n" + ",".join(bad) + "\n" + data[1] + "\n")
        assert run(remap.remap, p) == run(remap.remap_python, p), bad
    # a large file (chunks on many threads): the disturbances that do not raise, then one that does in the middle
    big = [header]
    while sum(len(x) for x in big) < 3_000_000:
        f = mutate(rng.choice(data).split(","))
        p = str(tmp_path / "probe.csv")
        open(p, "w").write(header + "\n" + ",".join(f) + "\n")
        if run(remap.remap_python, p)[0] == "ok":
            big.append(",".join(f))
    p = str(tmp_path / "big.csv")
    open(p, "w").write("\n".join(big) + "\n")
    a, b = run(remap.remap, p), run(remap.remap_python, p)
    assert a == b and a[0] == "ok" and a[1].count("\n") > 1000
    open(p, "w", newline="").write("\r\n".join(big) + "\r\n")
    assert run(remap.remap, p) == b
    big[len(big) // 2] = big[len(big) // 2].replace(",", ",,", 1)
    open(p, "w").write("\n".join(big) + "\n")
    assert run(remap.remap, p) == run(remap.remap_python, p) == ("ValueError", None)
    # GTF lines the script dies on
    lines = open(gtf).read().splitlines()
    ex = next(i for i, l in enumerate(lines) if "\texon\t" in l)
    for name, fn in (("no_tid", lambda l: l.replace("transcript_id", "transcript_name")), ("short", lambda l: "\t".join(l.split("\t")[:2])),
                     ("empty", lambda l: ""), ("bad_int", lambda l: l.replace("\t" + l.split("\t")[3] + "\t", "\tx\t", 1))):
        g2 = str(tmp_path / f"{name}.gtf")
        open(g2, "w").write("\n".join(lines[:ex] + [fn(lines[ex])] + lines[ex + 1:]) + "\n")
        src = str(tmp_path / "q.csv")
        open(src, "w").write(expected("quant_0_3.csv"))
        assert run(remap.remap, src, g2)[0] == run(remap.remap_python, src, g2)[0], name


def test_native_remap_prints_the_rows_in_front_of_a_failing_row(tmp_path, monkeypatch):
    """remap.py prints row by row: when a row makes it raise, the header and the rows in front are out already.  The native form
    (streamed in waves of pieces since round 4) leaves the same bytes -- on a file of several pieces, with the bad row in a late one"""
    from pantas_amd import remap
    rows = expected("quant_0_3.csv").splitlines()
    header, data = rows[0], rows[1:]
    big = [header]
    while sum(len(x) + 1 for x in big) < 5_000_000:
        big.extend(data)
    bad_at = len(big) * 3 // 4
    big[bad_at] = ",".join(["a"] * 15)     # a row of 15 fields: ValueError
    p = str(tmp_path / "big.csv")
    open(p, "w").write("\n".join(big) + "\n")
    gtf = os.path.join(Q, "anno.gtf")
    outs = []
    for fn in (remap.remap, remap.remap_python):
        out = io.StringIO()
        with pytest.raises(ValueError):
            fn(p, gtf, 100, out)
        outs.append(out.getvalue())
    assert outs[0] == outs[1] and outs[0].count("\n") > 1000
    # ... and through a descriptor (what the command line does)
    with open(tmp_path / "o.csv", "w+") as f:
        with pytest.raises(ValueError):
            remap.remap(p, gtf, 100, f)
        f.seek(0)
        assert f.read() == outs[1]


def test_inputs_the_reference_raises_on():
    from pantas_amd import quantify, remap
    classes = {"TypeError": TypeError, "ValueError": ValueError, "AttributeError": AttributeError, "IndexError": IndexError,
               "KeyError": KeyError, "AssertionError": AssertionError}
    for c in META["crashes"]:
        path = os.path.join(Q, c["input"])
        with pytest.raises(classes[c["exception"]]):
            if c["script"] == "quantify":
                quantify.quantify([path], [C2[0]], 0, False, io.StringIO())
            else:
                remap.remap(path, os.path.join(Q, "anno.gtf"), 100, io.StringIO())
        if c["script"] != "quantify":       # ... and the interpreted mirror likewise
            with pytest.raises(classes[c["exception"]]):
                remap.remap_python(path, os.path.join(Q, "anno.gtf"), 100, io.StringIO())
        if c["script"] == "quantify":       # ... and the interpreted mirror likewise
            with pytest.raises(classes[c["exception"]]):
                quantify.quantify_python([path], [C2[0]], 0, False, io.StringIO())


def test_dispatcher_quant_and_remap(tmp_path):
    """bin/pantas quant / remap: the reference dispatcher's argument handling (pantas:176-232)"""
    exe = os.path.join(ROOT, "bin", "pantas")
    r = subprocess.run([exe, "quant", *C1, *C2], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == expected("quant_1_3.csv")        # default: only events in both conditions
    r = subprocess.run([exe, "quant", "-a", *C1, *C2], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == expected("quant_0_3.csv")
    r = subprocess.run([exe, "quant", *C1, *C2[:2]], capture_output=True, text=True)
    assert r.returncode == 1 and "Wrong number of replicates" in r.stderr
    src = str(tmp_path / "q.csv")
    open(src, "w").write(expected("quant_0_3.csv"))
    r = subprocess.run([exe, "remap", "-i", "1", src, os.path.join(Q, "anno.gtf")], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == expected("remap_quant_0_3_1.csv")


def test_merge_is_a_join_not_a_scan(tmp_path):
    """40 k distinct events per replicate: the reference's pairwise eq_event scan would need ~10^9 comparisons"""
    import time
    from pantas_amd import quantify
    paths = []
    for rep in range(2):
        p = str(tmp_path / f"big{rep}.csv")
        with open(p, "w") as f:
            f.write("event_type,annotation_type,chrom,gene,strand,junction1_name,junction2_name,junction3_name,junction1_nodes,"
                    "junction1_coverage,junction2_nodes,junction2_coverage,junction3_nodes,junction3_coverage\n")
            for i in range(40000):
                a = 1000 + 10 * i
                f.write(f"ES,annotated,chr1,g{i % 300},+,T{i}_R1.1.3,T{i}_R1.1.2,T{i}_R1.2.3,{a}>{a + 5},{3 + i % 7 + rep},{a}>{a + 2},{5 + rep},{a + 3}>{a + 5},6\n")
        paths.append(p)
    t0 = time.perf_counter()
    out = io.StringIO()
    quantify.quantify(paths, list(reversed(paths)), 0, True, out)
    assert time.perf_counter() - t0 < 20
    assert len(out.getvalue().splitlines()) == 40001
