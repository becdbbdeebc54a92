"""The ONE JSON line of bench.py must stay short enough for the driver's tail of stdout (round 3's 22 KB line was cut and recorded as
unparsed) and still carry the contract's keys."""
import importlib.util
import io
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("recorded", ["r3_bench_default_s4_final.json", "r4_bench_default_s4_details.json"])
def test_bench_line_is_short_and_complete(capsys, monkeypatch, tmp_path, recorded):
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.loads(open(os.path.join(ROOT, "profiles", recorded)).read())   # a recorded full result
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full, types.SimpleNamespace(cpu_sample=800000))
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1
    assert len(out[0]) < 1900, len(out[0])
    line = json.loads(out[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "model" not in line["config"] and "workload" in line["config"]
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    details = json.load(open(os.path.join(str(tmp_path), line["details"])))
    assert details["value"] == full["value"] and "e2e" in details
