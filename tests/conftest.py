import os
import sys

import pytest

# torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the whole process, so load
# torch's before libturbometrics_hip.so pulls in /opt/rocm's (bench.py does the same).  Tests only use torch
# as a device allocator for the "frames already in HBM" case.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_tables():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")) as f:
        return json.load(f)
