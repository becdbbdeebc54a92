"""Fuzzed GAF input: the device walk (CPU harness here, the real kernels under -m gpu) against the oracle.
The oracle itself was run against the real reference on 1500 files of this generator in the build container
(882 accepted, 618 raising: outputs byte-identical, exception classes equal)."""
import io
import os
import random
import subprocess

import pytest

import augment_oracle
import fuzz_gaf


def oracle_result(gaf, gfa):
    out = io.StringIO()
    try:
        augment_oracle.augment(gaf, gfa, out)
        return out.getvalue(), None
    except augment_oracle.ReferenceCrash as e:
        return None, e.kind


def make_case(tmp_path, rng, n_lines):
    gfa, gaf = str(tmp_path / "fz.gfa"), str(tmp_path / "fz.gaf")
    fuzz_gaf.write_graph(gfa)
    with open(gaf, "w") as f:
        for i in range(n_lines):
            f.write(fuzz_gaf.random_line(rng, i) + "\n")
    return gfa, gaf


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_walk_source_fuzz(hostsim, tmp_path, seed):  # noqa: F811
    rng = random.Random(seed)
    ok = crash = 0
    for _ in range(120):
        gfa, gaf = make_case(tmp_path, rng, rng.choice([1, 1, 2, 5, 20]))
        want, kind = oracle_result(gaf, gfa)
        for env in ({}, {"HOSTSIM_BUCKET": "1"}):
            r = subprocess.run([hostsim["augment"], gaf, gfa], capture_output=True, text=True, env={**os.environ, **env})
            if kind is None:
                assert r.returncode == 0 and r.stdout == want, open(gaf).read()
            else:
                assert r.returncode == 1 and r.stdout == "", open(gaf).read()
                assert r.stderr.strip().splitlines()[-1].startswith(kind), (r.stderr, kind)
        ok += kind is None
        crash += kind is not None
    assert ok > 30 and crash > 30


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_kernels_fuzz(tmp_path, seed):
    """Through the C ABI on the GPU: accepted files byte-identical, raising files raise with the same class.
    Lines the reference accepts are also concatenated into one large file so that the short walk, the general
    walk and the aggregation tables see them together."""
    import tempfile
    from pantas_amd.engine import Graph, PantasError
    rng = random.Random(seed)
    good = []
    for _ in range(150):
        gfa, gaf = make_case(tmp_path, rng, rng.choice([1, 2, 5, 20]))
        want, kind = oracle_result(gaf, gfa)
        g = Graph(gfa).to_device(0)
        if kind is None:
            with tempfile.TemporaryFile() as f:
                g.augment(gaf, 20, f.fileno())
                f.seek(0)
                assert f.read().decode() == want, open(gaf).read()
            good.extend(open(gaf).read().splitlines())
        else:
            with pytest.raises(PantasError) as ei:
                g.augment(gaf, 20, -1)
            assert ei.value.kind == kind, open(gaf).read()
    assert len(good) > 100
    big = str(tmp_path / "big.gaf")
    with open(big, "w") as f:
        for rep in range(40):
            for line in good:
                f.write(line + "\n")
    want, kind = oracle_result(big, gfa)
    assert kind is None
    with tempfile.TemporaryFile() as f:
        Graph(gfa).to_device(0).augment(big, 20, f.fileno())
        f.seek(0)
        assert f.read().decode() == want
