import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def unit_rows_f32(seed, n, d):
    """SURVEY.md §8d synthetic-table recipe (same as tools/capture_goldens.py)."""
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32).astype(np.float64)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)
