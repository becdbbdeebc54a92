"""The GPU tokeniser (tokenize.hip) against the host tokeniser (gaf.cpp): same records, same order, same counters,
same refusals -- and the same GFA bytes as the reference goldens / the oracle behind either of them."""
import io
import json
import os
import random
import tempfile

import numpy as np
import pytest

import augment_oracle
import fuzz_gaf
from conftest import CASES, GOLDEN

pytestmark = pytest.mark.gpu


def both(g, gaf, thr=20):
    """(stats, arrays) through the host and the device tokeniser"""
    out = []
    for where in ("host", "device"):
        g.set_tokenizer(where)
        g.augment_reset()
        b = g.load_gaf(gaf, thr)
        out.append((b.stats, g.download_batch(b), b))
    return out


def assert_same(host, dev):
    hs, ha, _ = host
    ds, da, _ = dev
    assert hs.tokenizer == 1 and ds.tokenizer == 2
    for f in ("lines", "rejected", "skipped", "records", "path_nodes", "cs_ops", "del_records", "del_path_nodes", "text_bytes", "alg_bytes"):
        assert getattr(hs, f) == getattr(ds, f), f
    for x, y, name in zip(ha, da, ("hdr", "node_off", "op_off", "nodes", "ops")):
        assert np.array_equal(x, y), name


@pytest.mark.parametrize("case", [c for c in CASES if c != "mixedwidth"])
def test_golden_batches_equal(case):
    from pantas_amd.engine import Graph
    d = os.path.join(GOLDEN, case)
    g = Graph(f"{d}/in.gfa").to_device(0)
    h, dv = both(g, f"{d}/in.gaf")
    assert_same(h, dv)
    # and the whole run behind the device tokeniser is the reference's bytes
    g.set_tokenizer("device")
    with tempfile.TemporaryFile() as f:
        bs, _ = g.augment(f"{d}/in.gaf", 20, f.fileno())
        f.seek(0)
        assert f.read().decode() == open(f"{d}/expected.aug.gfa").read()
    assert bs.tokenizer == 2 and bs.rejected == json.load(open(f"{d}/expected.json"))["rejected"]


def test_names_that_are_not_decimals_are_refused_or_routed():
    from pantas_amd.engine import Graph, PantasError
    d = os.path.join(GOLDEN, "mixedwidth")
    g = Graph(f"{d}/in.gfa").to_device(0)
    g.set_tokenizer("auto")
    b = g.load_gaf(f"{d}/in.gaf")
    if b.stats.tokenizer == 2:
        pytest.skip("this golden graph has decimal names")
    g.set_tokenizer("device")
    with pytest.raises(PantasError) as ei:
        g.load_gaf(f"{d}/in.gaf")
    assert ei.value.kind == "Unsupported"


def test_synthetic_batches_equal_and_bucketed(tmp_path):
    from pantas_amd.engine import Graph, Synth
    pre = str(tmp_path / "s")
    Synth(seed=31, n_genes=300, n_lines=200000, p_indel=0.1, p_novel_skip=0.05).write(pre)
    g = Graph(pre + ".gfa").to_device(0)
    h, dv = both(g, pre + ".gaf")
    assert_same(h, dv)
    assert dv[0].records > 150000
    for thr in (0, 61):
        h, dv = both(g, pre + ".gaf", thr)
        assert_same(h, dv)


@pytest.mark.parametrize("seed", [5, 6])
def test_fuzzed_text(tmp_path, seed):
    """Damaged text: both tokenisers keep the same records and refuse the same line with the same class; the run
    behind the device tokeniser ends like the oracle's."""
    from pantas_amd.engine import Graph, PantasError
    rng = random.Random(seed)
    gfa = str(tmp_path / "g.gfa")
    fuzz_gaf.write_graph(gfa)
    g = Graph(gfa).to_device(0)
    ok = bad = 0
    for it in range(150):
        gaf = str(tmp_path / "r.gaf")
        nl = rng.choice([1, 2, 5, 40, 300])
        with open(gaf, "w", encoding="latin-1", newline="") as f:
            for i in range(nl):
                l = fuzz_gaf.random_line(rng, i)
                if rng.random() < (0.5 if nl < 10 else 0.01):
                    l = fuzz_gaf.mutate_text(rng, l)
                f.write(l + ("\n" if (i < nl - 1 or rng.random() < 0.9) else ""))
        h, dv = both(g, gaf)
        assert_same(h, dv)
        results = []
        for where in ("host", "device"):
            g.set_tokenizer(where)
            try:
                with tempfile.TemporaryFile() as f:
                    g.augment(gaf, 20, f.fileno())
                    f.seek(0)
                    results.append(("ok", f.read()))
            except PantasError as e:
                results.append(("err", e.kind, str(e)))
        assert results[0] == results[1], open(gaf, encoding="latin-1").read()
        ok += results[0][0] == "ok"
        bad += results[0][0] == "err"
    assert ok > 20 and bad > 20


def test_oracle_parity_behind_device_tokenizer(tmp_path):
    from pantas_amd.engine import Graph, Synth
    pre = str(tmp_path / "s")
    Synth(seed=32, n_genes=40, n_lines=30000, p_indel=0.3, bubble_rate=0.1, p_novel_skip=0.1, p_novel_ce=0.05).write(pre)
    g = Graph(pre + ".gfa").to_device(0)
    g.set_tokenizer("device")
    with tempfile.TemporaryFile() as f:
        bs, _ = g.augment(pre + ".gaf", 20, f.fileno())
        f.seek(0)
        got = f.read().decode()
    want = io.StringIO()
    a = augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    assert bs.tokenizer == 2 and bs.rejected == a.rejected and got == want.getvalue()


def test_empty_and_unterminated_input(tmp_path):
    from pantas_amd.engine import Graph
    d = os.path.join(GOLDEN, "micro")
    g = Graph(f"{d}/in.gfa").to_device(0)
    empty = str(tmp_path / "e.gaf")
    open(empty, "w").close()
    text = open(f"{d}/in.gaf").read()
    cut = str(tmp_path / "c.gaf")
    open(cut, "w").write(text.rstrip("\n"))
    for path in (empty, cut):
        h, dv = both(g, path)
        assert_same(h, dv)
