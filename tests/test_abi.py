"""The C-ABI library loads without a GPU and exports every symbol include/pantas_gpu.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pantas_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pantas_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_header_symbols():
    from pantas_amd import _lib
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/pantas_gpu.h but not exported"
    assert set(_lib.SIGNATURES) == set(names)


def test_no_silent_cpu_path():
    """Without a device the engine refuses instead of computing on the host."""
    from pantas_amd import _lib
    from pantas_amd.engine import Graph, PantasError
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    g = Graph(os.path.join(ROOT, "tests/golden/micro/in.gfa"))
    assert g.stats.segments == 6
    with pytest.raises(PantasError) as ei:
        g.to_device(0)
    assert ei.value.kind == "NoDevice"
    g2 = Graph(os.path.join(ROOT, "tests/golden/micro/expected.aug.gfa"))
    with pytest.raises(PantasError) as ei:
        g2.call_prepare(os.path.join(ROOT, "tests/golden/micro/in.gtf"), 0)
    assert ei.value.kind == "NoDevice"


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pantas_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read()
                assert "augment_oracle" not in src and "call_oracle" not in src, f
