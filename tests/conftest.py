import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

ROOT_GOLDEN = None
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["micro", "quirks", "mixedwidth", "synth_a", "synth_b", "synth_c"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---- the CPU harness (tests/hostsim): the product's host code and kernel sources compiled for the CPU under ASan / UBSan.
# Test infrastructure; built once per session, by whichever test file asks first.
BUILD = os.path.join(ROOT, "build", "hostsim")
CSRC = os.path.join(ROOT, "pantas_amd", "csrc")
CXX = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
       "-fno-omit-frame-pointer", "-I", CSRC]


@pytest.fixture(scope="session")
def hostsim():
    os.makedirs(BUILD, exist_ok=True)
    common = [os.path.join(CSRC, f) for f in ("graph.cpp", "graph_cache.cpp", "common.cpp")]
    aug = os.path.join(BUILD, "hostsim_augment")
    call = os.path.join(BUILD, "hostsim_call")
    subprocess.check_call(CXX + ["-o", aug, os.path.join(ROOT, "tests/hostsim/hostsim_augment.cpp"), *common,
                                 os.path.join(CSRC, "gaf.cpp"), os.path.join(CSRC, "augment_host.cpp"), "-lpthread"])
    subprocess.check_call(CXX + ["-o", call, os.path.join(ROOT, "tests/hostsim/hostsim_call.cpp"), *common,
                                 os.path.join(CSRC, "call_host.cpp"), "-lpthread"])
    return {"augment": aug, "call": call}
