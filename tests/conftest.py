import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need the MI355X: without one they are skipped (not failed with hipErrorNoDevice), so a plain
    `pytest tests` works on a CPU-only machine."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def weights():
    w = load_npz("weights_vn.npz")
    return {k: v for k, v in w.items() if not k.startswith("__")}


@pytest.fixture(scope="session")
def g1():
    return load_npz("g1_realistic.npz")


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))
