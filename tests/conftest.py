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

# the oracle's C checker is built / loaded here, before any test touches a GPU (its `make` is a fork + exec, which a
# GPU-initialised process must not do - oracle/c_oracle.py refuses it)
from oracle import c_oracle as _c_oracle  # noqa: E402
_c_oracle.lib()


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


def g12_scenario(g, tag):
    """Inputs of a G12 scenario (tools/capture_goldens.py g12) regenerated from its recipe: per-type feature blocks in the
    SORTED-type-name numbering, type ids, edges, the order in which the reference visited the types, n_dims."""
    types = [str(t) for t in g[tag + "_types"]]
    T = len(types)
    rec = [int(x) for x in g[tag + "_recipe"]]
    N, seed, n_edges, n_dims = rec[:4]
    dims, with_data = rec[4:4 + T], [bool(b) for b in rec[4 + T:4 + 2 * T]]
    type_of = {"small": lambda i: i % 3, "multi": lambda i: 0 if (i * 7) % 10 < 3 else 1}[tag]
    rng = np.random.default_rng(seed)
    tid = np.asarray([type_of(i) for i in range(N)], dtype=np.int32)
    feats = [rng.standard_normal((int((tid == ti).sum()), dims[ti])) for ti in range(T)]
    src = rng.integers(0, N, n_edges)
    dst = rng.integers(0, N, n_edges)
    srt = sorted(types)
    remap = np.asarray([srt.index(t) for t in types], dtype=np.int32)
    feats_sorted, data_sorted = [None] * T, [None] * T
    for ti in range(T):
        feats_sorted[remap[ti]] = feats[ti]
        data_sorted[remap[ti]] = with_data[ti]
    order = [srt.index(str(t)) for t in g[tag + "_type_order"]]
    return dict(types=srt, type_id=remap[tid], feats=feats_sorted, with_data=data_sorted, src=src, dst=dst, order=order,
                n_dims=n_dims, N=N)


def g15_scenario(g):
    """Inputs of G15 (tools/capture_goldens.py g15) regenerated from its recipe: 6 000 nodes in 3 types, per-type feature blocks, edges
    (24 000 random + the first 200 repeated), and the float64 table E the reference's blend stage saw (captured PCA-unit block + one-hot of
    the SORTED type names padded to 4 columns)."""
    N, seed, n_edges, n_dup, n_dims, k = (int(x) for x in g["recipe"][:6])
    dims = [int(x) for x in g["recipe"][6:9]]
    types = [str(t) for t in g["types"]]
    tid = np.asarray([0 if i % 6 == 0 else (2 if i % 6 == 1 else 1) for i in range(N)], dtype=np.int32)
    rng = np.random.default_rng(seed)
    feats = [rng.standard_normal((int((tid == ti).sum()), dims[ti])) for ti in range(3)]
    src = rng.integers(0, N, n_edges)
    dst = rng.integers(0, N, n_edges)
    src = np.concatenate([src, src[:n_dup]])
    dst = np.concatenate([dst, dst[:n_dup]])
    srt = sorted(types)
    cat = np.asarray([srt.index(types[t]) for t in tid])
    ohe = np.zeros((N, 4))
    ohe[np.arange(N), cat] = 1.0
    E = np.concatenate([g["pca_unit"], ohe], axis=1)
    return dict(N=N, n_dims=n_dims, k=k, dims=dims, types=types, tid=tid, feats=feats, src=src, dst=dst, E=E,
                users=np.flatnonzero(tid == 0), items=np.flatnonzero(tid == 1))
