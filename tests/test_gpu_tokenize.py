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


def test_names_that_are_not_decimals_are_refused_or_routed(tmp_path):
    """segment names that are no canonical decimals (here: the micro golden with an 's' in front of every name): `auto` routes
    the batch to the host tokeniser, `device` refuses it -- and the run behind `auto` is the reference's bytes for the renamed graph"""
    import re
    from pantas_amd.engine import Graph, PantasError
    d = os.path.join(GOLDEN, "micro")
    gfa, gaf = str(tmp_path / "n.gfa"), str(tmp_path / "n.gaf")
    rename_gfa = lambda t: re.sub(r"(?m)^(S\t)(\d+)", r"\1s\2", re.sub(r"(?m)^(L\t)(\d+)(\t[+-]\t)(\d+)", r"\1s\2\3s\4", t))
    text = open(f"{d}/in.gfa").read()
    open(gfa, "w").write(rename_gfa(text))
    open(gaf, "w").write(re.sub(r"([<>])(\d+)", r"\1s\2", open(f"{d}/in.gaf").read()))
    g = Graph(gfa).to_device(0)
    g.set_tokenizer("auto")
    b = g.load_gaf(gaf)
    assert b.stats.tokenizer == 1 and b.stats.records > 0          # routed to the host tokeniser
    g.set_tokenizer("device")
    with pytest.raises(PantasError) as ei:
        g.load_gaf(gaf)
    assert ei.value.kind == "Unsupported"
    g.set_tokenizer("auto")
    with tempfile.TemporaryFile() as f:
        g.augment(gaf, 20, f.fileno())
        f.seek(0)
        got = f.read().decode()
    assert got == rename_gfa(open(f"{d}/expected.aug.gfa").read())


def _with_env(**kw):
    import contextlib

    @contextlib.contextmanager
    def cm():
        old = {k: os.environ.get(k) for k in kw}
        os.environ.update({k: str(v) for k, v in kw.items()})
        try:
            yield
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return cm()


@pytest.mark.parametrize("names", ["short", "seven_digits"])
def test_one_parse_per_line_equals_two_pass_and_host(tmp_path, names):
    """the one-parse build (records written into a word array congruent with the text, round 3) against the two-pass build and the
    host tokeniser: identical arrays.  "short": names of one to six characters -- lines whose records do not fit their own
    text and take arena room; an arena of eight words: the batch falls back to the two-pass build."""
    from pantas_amd.engine import Graph, Synth
    pre = str(tmp_path / "s")
    Synth(seed=41, n_genes=120, n_lines=60000, p_indel=0.15, p_novel_skip=0.05, base_id=10**6, read_len=150 if names == "short" else 100).write(pre)
    if names == "short":       # segment 1000000 + i becomes i + 1: names of one to six characters, '>7>8>9' paths
        import re
        small = lambda m: m.group(1) + str(int(m.group(2)) - 10**6 + 1)
        gfa = re.sub(r"(?m)^(L\t)(\d+)(\t[+-]\t)(\d+)", lambda m: m.group(1) + str(int(m.group(2)) - 10**6 + 1) + m.group(3) + str(int(m.group(4)) - 10**6 + 1),
                     re.sub(r"(?m)^(S\t)(\d+)", small, open(pre + ".gfa").read()))
        open(pre + ".gfa", "w").write(gfa)
        open(pre + ".gaf", "w").write(re.sub(r"([<>])(\d+)", small, open(pre + ".gaf").read()))
    g = Graph(pre + ".gfa").to_device(0)
    h, one = both(g, pre + ".gaf")
    assert_same(h, one)
    with _with_env(PANTAS_TOKENIZER_PASSES=2):
        _, two = both(g, pre + ".gaf")
    assert_same(h, two)
    with _with_env(PANTAS_TOKENIZER_ARENA_WORDS=8):
        _, starved = both(g, pre + ".gaf")
    assert_same(h, starved)
    for x, y in zip(one[1], two[1]):
        assert np.array_equal(x, y)
    # and the walk that reads the `rec` entries the one-parse gather wrote prints the oracle's bytes
    g.set_tokenizer("device")
    with tempfile.TemporaryFile() as f:
        bs, _ = g.augment(pre + ".gaf", 20, f.fileno())
        f.seek(0)
        got = f.read().decode()
    want = io.StringIO()
    a = augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
    assert bs.rejected == a.rejected and got == want.getvalue()


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


def test_a_lone_carriage_return_is_found_wherever_it_stands(tmp_path):
    """Python's universal newlines end a line at a '\\r' that no '\\n' follows; the kernels only look for one when the line index says
    the text holds one (count_newlines_kernel, since late round 4: the host used to look while it staged the text).  The '\\r' is put at
    byte offsets around the kernel's borders -- words, a lane's 64 bytes, the 16 KB tile, the text's last byte -- by lengthening the
    first line's read name; both tokenisers must end alike (same bytes or the same refusal), and a '\\r\\n' or a '\\r' as the last
    byte must change nothing."""
    from pantas_amd.engine import Graph, PantasError
    d = os.path.join(GOLDEN, "synth_a")
    raw = open(f"{d}/in.gaf", "rb").read()
    assert len(raw) > 40000
    lines = raw.split(b"\n")
    g = Graph(f"{d}/in.gfa").to_device(0)

    def ends(text):
        gaf = str(tmp_path / "r.gaf")
        open(gaf, "wb").write(text)
        out = []
        for where in ("host", "device"):
            g.set_tokenizer(where)
            try:
                with tempfile.TemporaryFile() as f:
                    g.augment(gaf, 20, f.fileno())
                    f.seek(0)
                    out.append(("ok", f.read()))
            except PantasError as e:
                out.append(("err", e.kind, str(e)))
        assert out[0] == out[1], (out[0][:2], out[1][:2])
        return out[0]

    plain = ends(raw)
    assert plain[0] == "ok"
    seen_change = 0
    def with_cr_at(target):
        """the text with one '\\r' inside a read name, at byte offset `target`"""
        name0 = lines[0].split(b"\t")[0]
        if target < 100:                                  # inside the first line's own name, grown to hold it
            grown = name0 + b"x" * max(0, target + 3 - len(name0))
            grown = grown[:target] + b"\r" + grown[target + 1:]
            return b"\n".join([b"\t".join([grown] + lines[0].split(b"\t")[1:])] + lines[1:])
        off = 0
        for i, ln in enumerate(lines):
            n = target - 1 - off                          # the first line's name grown by n puts byte 1 of line i's name at the target
            if i > 0 and 0 <= n < 400 and len(ln.split(b"\t")[0]) >= 4:
                mod = list(lines)
                mod[0] = b"\t".join([name0 + b"x" * n] + lines[0].split(b"\t")[1:])
                mod[i] = ln[:1] + b"\r" + ln[2:]
                return b"\n".join(mod)
            off += len(ln) + 1
        raise AssertionError(f"no line for offset {target}")

    for border in (16, 64, 4096, 16384, 32768):
        for delta in (-2, -1, 0, 1):
            text = with_cr_at(border + delta)
            assert text[border + delta] == 13 and text.count(b"\r") == 1
            seen_change += ends(text) != plain
    assert seen_change >= 15                              # (a read name cut in two is two broken lines: the run ends differently)
    assert ends(raw.replace(b"\n", b"\r\n")) == plain
    assert ends(raw.rstrip(b"\n") + b"\r") == ends(raw.rstrip(b"\n"))


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


def test_lean_parser_for_plain_lines_equals_general_parser(tmp_path):
    """round 5: plain lines go through parse_gaf_line_fast inside parse_once_kernel, everything else through the general parser
    in the same lane (PANTAS_TOK_FAST=0: all of them through the general one).  Same arrays either way and as the host tokeniser's --
    on the synthetic recipe (all plain, two-indel reads among them) and on a file where every fifth line is mutated (other
    whitespace, signs, '=' ops, doubled tabs, cut columns: lines the lean parser must hand over)."""
    from pantas_amd.engine import Graph, Synth
    pre = str(tmp_path / "s")
    Synth(seed=47, n_genes=200, n_lines=120000, p_indel=0.2, p_sub=0.4, p_novel_skip=0.05, p_two_indels=0.1).write(pre)
    g = Graph(pre + ".gfa").to_device(0)
    h, fast = both(g, pre + ".gaf")
    assert_same(h, fast)
    with _with_env(PANTAS_TOK_FAST=0):
        _, general = both(g, pre + ".gaf")
    assert_same(h, general)
    # mutations that keep a line acceptable to the reference (no raise): the two parsers and the host must agree line by line
    rng = random.Random(9)
    lines = open(pre + ".gaf").read().splitlines()[:40000]

    def mutate(l):
        t = l.split("\t")
        if len(t) < 14:              # (an unmapped read: twelve columns)
            return l
        k = rng.randrange(8)
        if k == 0:
            return l.replace("\t", " ", 1)                        # a space where a tab stood
        if k == 1:
            return l.replace("\t", "\t\t", 1)                     # an empty column: split() closes the gap
        if k == 2:
            t[11] = "+" + t[11]
        elif k == 3:
            t[7] = "0" + t[7] if t[5] != "*" else t[7]           # a leading zero in an integer
        elif k == 4:
            t[12] = t[12] + " "                                   # a space behind a tag (still the same tokens)
        elif k == 5:
            t = [x.replace("cs:Z::", "cs:Z:=ACGT:") if x.startswith("cs:Z::") else x for x in t]
        elif k == 6:
            t.append("zz:Z:cs:x dv:f:1")
        else:
            t = t[:12] + t[13:]                                   # a tag dropped
        return "\t".join(t)
    with open(pre + ".m.gaf", "w") as f:
        for i, l in enumerate(lines):
            f.write((mutate(l) if i % 5 == 0 else l) + "\n")
    h, fast = both(g, pre + ".m.gaf")
    assert_same(h, fast)
    with _with_env(PANTAS_TOK_FAST=0):
        _, general = both(g, pre + ".m.gaf")
    for x, y in zip(fast[1], general[1]):
        assert np.array_equal(x, y)
