"""The GFA parsed on the GPU (tokenize.hip load_graph_device, round 3) against the host loader (graph.cpp): the same arrays.

Both loaders can write the binary graph cache (graph_cache.cpp) -- every array of the graph, the name table and the facts the two
command lines validate -- so "the same graph" is "the same cache file".  Graphs the device form does not take (small files, names
that are no dense decimals, duplicate S lines, whitespace-led lines, lone carriage returns) go to the host loader by themselves."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys
sys.path.insert(0, ROOT)
from pantas_amd._lib import lib
from pantas_amd.engine import Graph
lib.pantas_device_init(0)
g = Graph(GFA, cache=CACHE)
assert not g.used_cache
s = g.stats
print("OK", s.segments, s.link_slots, s.lines)
"""


def cache_of(gfa, how, tmp_path, timing=False):
    cache = str(tmp_path / f"{how}.pgc")
    env = {**os.environ, "PANTAS_GFA_LOADER": how}
    if timing:
        env["PANTAS_TIMING"] = "1"
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\nGFA = {gfa!r}\nCACHE = {cache!r}\n" + CHILD], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    r.stats_line = r.stdout.strip().splitlines()[-1]
    return (open(cache, "rb").read() if os.path.exists(cache) else None), r


@pytest.mark.parametrize("kw", [dict(n_genes=3000, n_lines=1000, base_id=10**7, intron_long_max=400, n_chrom=3),
                                dict(n_genes=800, n_lines=1000, base_id=10**9, intron_long_max=3000, n_chrom=2, bubble_rate=0.02)], ids=["3000 genes", "ids of ten digits"])
def test_device_loader_builds_the_host_loaders_arrays(tmp_path, kw):
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "g")
    Synth(seed=3, **kw).write(pre)
    assert os.path.getsize(pre + ".gfa") > (1 << 20)
    host, _ = cache_of(pre + ".gfa", "host", tmp_path)
    dev, r = cache_of(pre + ".gfa", "device", tmp_path, timing=True)
    assert "load_graph_device" in r.stderr, r.stderr[-2000:]
    assert host == dev


def test_augmented_gfa_links_in_any_order(tmp_path):
    """an augmented GFA carries the links the reads added behind its P lines (sources in any order), RC / NC tags on every line"""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "g")
    Synth(seed=4, n_genes=2500, n_lines=300_000, base_id=10**7, p_novel_skip=0.05).write(pre)
    with open(pre + ".aug.gfa", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"ID:Z:N" in open(pre + ".aug.gfa", "rb").read()[-4000:]
    host, _ = cache_of(pre + ".aug.gfa", "host", tmp_path)
    dev, _ = cache_of(pre + ".aug.gfa", "device", tmp_path)
    assert host == dev


def test_graphs_the_device_form_hands_back(tmp_path):
    """whitespace-led S line, duplicate S line, a name that is no number, CRLF line ends: `auto` loads them like the host loader"""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "g")
    Synth(seed=5, n_genes=2500, n_lines=100, base_id=10**7).write(pre)
    text = open(pre + ".gfa").read()
    lines = text.splitlines()
    first_s = next(i for i, l in enumerate(lines) if l.startswith("S"))
    variants = {
        "ws": "\n".join(lines[:first_s] + [" " + lines[first_s]] + lines[first_s + 1:]) + "\n",
        "dup": "\n".join(lines[:first_s + 1] + [lines[first_s]] + lines[first_s + 1:]) + "\n",
        "name": text.replace(lines[first_s].split("\t")[1], "x" + lines[first_s].split("\t")[1]),
        "crlf": text.replace("\n", "\r\n"),
    }
    for name, t in variants.items():
        p = str(tmp_path / f"{name}.gfa")
        open(p, "w", newline="").write(t)
        host, rh = cache_of(p, "host", tmp_path)
        auto, ra = cache_of(p, "auto", tmp_path)
        assert rh.stats_line == ra.stats_line, name
        assert host == auto, name           # (None == None where the graph is not one the cache holds: duplicate names, names that are no decimals)


def _call(aug_gfa, gtf, env_extra, flags=("--rca", "2", "--novel")):
    env = {**os.environ, "PANTAS_TIMING": "1", **env_extra}
    return subprocess.run([sys.executable, "-m", "pantas_amd.call", *flags, aug_gfa, gtf], capture_output=True, cwd=ROOT, env=env)


def test_tags_for_call_taken_by_the_device_loader(tmp_path):
    """`pantas call` on a graph the device loader takes: the tags of the S / L lines (NC, RC, EX, JN, ID) come from the loader's visit of
    the text in HBM (gfa_tags_kernel, pantas_graph_load_for_call) -- the CSV must be the one the host scan of the lines gives, with both
    flag sets; a tag of another form anywhere sends the scan back to the host, which raises what the reference raises"""
    from pantas_amd.engine import Synth
    pre = str(tmp_path / "g")
    Synth(seed=6, n_genes=2500, n_lines=400_000, base_id=10**7, p_novel_skip=0.05, p_novel_ce=0.03, bubble_rate=0.05).write(pre)
    with open(pre + ".aug.gfa", "wb") as f:
        r = subprocess.run([sys.executable, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], stdout=f, stderr=subprocess.PIPE, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for flags in (("--rca", "2", "--novel"), ("--rca", "1")):
        dev = _call(pre + ".aug.gfa", pre + ".gtf", {}, flags)
        assert dev.returncode == 0, dev.stderr.decode()[-3000:]
        assert b"tags of the S / L lines (for call)" in dev.stderr, dev.stderr.decode()[-3000:]
        assert b"tag scan" in dev.stderr
        host = _call(pre + ".aug.gfa", pre + ".gtf", {"PANTAS_CALL_TAGS_HOST": "1"}, flags)
        assert host.returncode == 0, host.stderr.decode()[-3000:]
        assert dev.stdout == host.stdout and dev.stdout.count(b"\n") > 1000
        plain = _call(pre + ".aug.gfa", pre + ".gtf", {"PANTAS_GFA_LOADER": "host"}, flags)
        assert plain.returncode == 0 and plain.stdout == dev.stdout
    # tags of other forms: ints python takes and the plain scan does not ('+', '_', leading zeros are fine for both), a tag without its
    # second ':', an int that is none, a hist value with an empty piece -- equal output or equal failure, with and without the device scan
    text = open(pre + ".aug.gfa").read().split("\n")
    s_at = [i for i, l in enumerate(text) if l.startswith("S") and "EX:Z:" in l][50]
    l_at = [i for i, l in enumerate(text) if l.startswith("L") and "JN:Z:" in l][20]
    import re
    variants = {
        "plus_int": (s_at, lambda l: re.sub(r"NC:i:(\d+)", r"NC:i:+\1", l)),
        "underscore_int": (l_at, lambda l: re.sub(r"RC:i:(\d+)", r"RC:i:0_\1", l) if re.search(r"RC:i:\d", l) else l + "\tRC:i:1_0"),
        "two_fields": (s_at, lambda l: l + "\tXX:Z"),
        "bad_int": (l_at, lambda l: l + "\tRC:i:12x"),
        "hist_empty_piece": (s_at, lambda l: l + "\tIL:Z:3..4"),
        "later_nc_wins": (s_at, lambda l: l + "\tNC:i:77"),
        "four_fields": (l_at, lambda l: l + "\tJN:Z:a:b"),
    }
    for name, (at, fn) in variants.items():
        t = list(text)
        t[at] = fn(t[at])
        p = str(tmp_path / f"{name}.gfa")
        open(p, "w").write("\n".join(t))
        dev = _call(p, pre + ".gtf", {})
        host = _call(p, pre + ".gtf", {"PANTAS_GFA_LOADER": "host"})
        assert dev.returncode == host.returncode, (name, dev.stderr.decode()[-1500:], host.stderr.decode()[-1500:])
        assert dev.stdout == host.stdout, name
        if dev.returncode != 0:
            last = lambda r: [l for l in r.stderr.decode().splitlines() if l and not l.startswith("[timing]")][-1]
            assert last(dev) == last(host), name
