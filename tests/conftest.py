import os
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
