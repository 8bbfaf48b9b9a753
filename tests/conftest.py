import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vdn-nerf_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return {k.replace("__", "/"): v for k, v in d.items()}


def relmax(a, b):
    """max |a-b| / max |b|  (the 'rel' of SURVEY.md 4 / BASELINE north-star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def relelem(a, b, rtol=1e-4, floor=1e-6):
    """Element-wise form of "within 1e-4 rel": worst |a-b| / (rtol |b| + floor max|b|) over the tensor; <= 1 passes.
    Unlike relmax, an entry much smaller than the tensor's largest one is still held to rtol of ITS size (down to the
    floor, 1e-6 of the largest entry - fp32 round-off of the sums that produce it)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / (rtol * np.abs(b) + floor * np.abs(b).max() + 1e-300)).max())


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
