"""Persisted index format: everything the retrieval path needs to come back up without re-fitting.

The reference keeps its index only in memory (a dict of scikit-learn KD-trees rebuilt by every `fit`).  This build's
index is three flat arrays, so the on-disk format is a single uncompressed `.npz`:

    format        "hwer-index/1" | "hwer-index/2" (round 6: /2 = /1 + `vectors64`)
    vectors       (N, d) float32   the `vectors` table, rows in registry order (what the screen kernels read)
    vectors64     (N, d) float64   ONLY for GENUINE float64 `vectors` (what ContentRecommendation.fit returns: hwer/content_recommender.py:96): the rows
                                   every value is computed from (distances, scores, predict) - `vectors` is then their float32 rounding
    arith         int              0: the reference's float64 score arithmetic, 1: float32 (GcnNCF flavour)
    scoring       str              "base" | "gcn"
    type_names    (T,) str         node types with a retrieval table
    type_rows_<i> (N_t,) int64     ascending global row ids of type i
    node_types    (N,) str         } registry: node i = Node(node_types[i], node_ids[i]) -> row i   (optional)
    node_ids      (N,) str         }

`save_index` / `load_index` are host-side only (numpy); `restore_model` rebuilds a fitted `VectorIndex` on the GPU.
float32 storage is exact for float32 tables and for float64 tables whose values are float32-representable; a table of genuine float64 rows is
stored in full beside its rounding (format /2; rounds 3-5 stored the rounding only and a restored ContentRecommendation answered from it).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

FORMAT = "hwer-index/1"
FORMAT64 = "hwer-index/2"


def save_index(path: str, vectors: np.ndarray, type_rows: Dict[str, np.ndarray], arith: int, scoring: str = "base",
               nodes: Optional[Sequence] = None) -> None:
    """vectors: (N, d) float32, or float64 - stored as its float32 rounding and, when the values are not float32-representable, in full as well
    (`vectors64`, format /2).  nodes: optional registry in row order (objects with .node_type / .node_external_id, e.g. hwer.Node)."""
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    if v.ndim != 2:
        raise ValueError("vectors must be (N, d)")
    v64 = None
    if np.asarray(vectors).dtype == np.float64 and not np.array_equal(v.astype(np.float64), vectors, equal_nan=True):
        v64 = np.ascontiguousarray(vectors, dtype=np.float64)
    names = sorted(type_rows)
    payload = {"format": np.array(FORMAT if v64 is None else FORMAT64), "vectors": v, "arith": np.array(int(arith)), "scoring": np.array(str(scoring)),
               "type_names": np.array(names, dtype=str)}
    if v64 is not None:
        payload["vectors64"] = v64
    for i, t in enumerate(names):
        rows = np.ascontiguousarray(type_rows[t], dtype=np.int64)
        if rows.size and (rows.min() < 0 or rows.max() >= len(v) or (np.diff(rows) <= 0).any()):
            raise ValueError("rows of type %r must be ascending row ids of `vectors`" % t)
        payload["type_rows_%d" % i] = rows
    if nodes is not None:
        if len(nodes) != len(v):
            raise ValueError("registry and vectors differ in length")
        payload["node_types"] = np.array([str(n.node_type) for n in nodes], dtype=str)
        payload["node_ids"] = np.array([str(n.node_external_id) for n in nodes], dtype=str)
    with open(path, "wb") as fh:
        np.savez(fh, **payload)


def load_index(path: str) -> dict:
    """-> {"vectors", "vectors64" | None, "type_rows", "arith", "scoring", "node_types" | None, "node_ids" | None} (numpy, host)."""
    with np.load(path, allow_pickle=False) as z:
        if str(z["format"]) not in (FORMAT, FORMAT64):
            raise ValueError("not a %s file: %r" % (FORMAT, str(z["format"])))
        names = [str(t) for t in z["type_names"]]
        out = {"vectors": z["vectors"], "vectors64": z["vectors64"] if "vectors64" in z.files else None, "arith": int(z["arith"]), "scoring": str(z["scoring"]),
               "type_rows": {t: z["type_rows_%d" % i] for i, t in enumerate(names)},
               "node_types": z["node_types"] if "node_types" in z.files else None,
               "node_ids": z["node_ids"] if "node_ids" in z.files else None}
    return out


def save_model(path: str, model) -> None:
    """Persist a fitted RecommendationBase (its device table is downloaded once)."""
    assert model.fit_done
    ix = model.knn.index
    table = model.knn.idx_to_node()
    v = ix.V64[:, :ix.d].cpu().numpy() if ix.V64 is not None else ix.V[:, :ix.d].cpu().numpy()      # (genuine float64 rows are persisted as they are)
    save_index(path, v, {t: tt.rows for t, tt in ix.tables.items()}, ix.arith, model.scoring, table)


def restore_model(path: str, device=None):
    """Fitted model (VectorIndex / GcnScoredIndex flavour) from a file written by save_model: same neighbours, same
    scores - the stored float32 rows are the rows the kernels searched before."""
    from . import GcnScoredIndex, Node, VectorIndex
    z = load_index(path)
    if z["node_types"] is None:
        raise ValueError("the file has no node registry: use load_index + DeviceIndex for array-level access")
    nodes = [Node(t, i) for t, i in zip(z["node_types"].tolist(), z["node_ids"].tolist())]
    cls = GcnScoredIndex if z["scoring"] == "gcn" else VectorIndex
    m = cls(node_types=set(z["type_rows"]), n_dims=int(z["vectors"].shape[1]))
    v = z["vectors64"] if z["vectors64"] is not None else (z["vectors"] if z["arith"] == 1 else z["vectors"].astype(np.float64))
    if cls is GcnScoredIndex:   # rows are already unit length: skip prepare_for_knn
        VectorIndex.fit(m, nodes, v)
    else:
        m.fit(nodes, v)
    return m
