import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a ROCm device: on a box without one they are skipped (with the reason) instead of erroring.  With a
    device present nothing is skipped -- a missing libvirnet_hip.so must fail loudly there, not hide."""
    try:
        import torch
        ok = torch.cuda.is_available()
        why = "no ROCm device visible (run with -m gpu on the MI355X box)"
    except Exception as e:                                    # pragma: no cover
        ok, why = False, f"torch import failed: {e}"
    if ok:
        return
    skip = pytest.mark.skip(reason=why)
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_golden(tag):
    return dict(np.load(os.path.join(GOLDEN, tag + ".npz")))
