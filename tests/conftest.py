import os
import sys

import pytest
import torch  # noqa: F401  - before libmgf_hip.so: torch bundles its own libamdhip64 (same soname as /opt/rocm's);
#                  whichever is loaded first serves both, and torch only finds the GPU through its own copy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)
