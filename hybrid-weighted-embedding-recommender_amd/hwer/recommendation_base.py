"""Retrieval core — MI355X-native drop-in for the reference's hwer/recommendation_base.py.

Same public classes and signatures (Node, Edge, MultiKNN, RecommendationBase); the KD-trees, the numpy fancy
indexing and the per-pair Python loops are replaced by one device-resident index (device_index.DeviceIndex) and
batched HIP kernels.  Additions (additive only): find_closest_neighbours_batch, predict_indices, find_items_for_user.

Behaviour kept: registry ids in insertion order; unit-norm audit with tolerance 1e-4; unknown
nodes in predict / positive / negative use row 0 clipped to [1e-6, 1e-5]; the query is NOT re-normalised;
candidates are ranked by distance to the query but scored against the anchor; k > N_t raises ValueError; results
are ordered by score descending.  Ties: (score desc, candidate rank asc) with candidates in (distance asc, id asc).
"""
from __future__ import annotations

import abc
from typing import Dict, List, Sequence, Set, Tuple, Union

import numpy as np
from itertools import repeat as _repeat

from .logging import getLogger
from .utils import NodeNotFoundException

NodeType = str
NodeExternalId = Union[str, int]
FeatureName = str


class Node:
    """Identity is (node_type, str(external_id)): Node("item", 5) == Node("item", "5")
    (reference: hwer/recommendation_base.py:19-36).  The hash is cached — it was the hottest function of the
    reference's retrieval loop."""
    __slots__ = ("node_type", "node_external_id", "_h")

    def __init__(self, node_type, node_external_id):
        self.node_type = node_type
        self.node_external_id = str(node_external_id)
        self._h = hash((self.node_type, self.node_external_id))

    def __hash__(self):
        return self._h

    def __eq__(self, other):
        if isinstance(other, Node):
            return self.node_type == other.node_type and self.node_external_id == other.node_external_id
        return NotImplemented

    def __repr__(self):
        return str((self.node_type, self.node_external_id))


class Edge:
    """Iterates as [src, dst, weight] (reference: hwer/recommendation_base.py:39-61)."""
    __slots__ = ("src", "dst", "weight", "contents")

    def __init__(self, src: Node, dst: Node, weight: float):
        self.src, self.dst, self.weight = src, dst, weight
        self.contents = [src, dst, weight]

    def __iter__(self):
        return iter(self.contents)

    def __hash__(self):
        return hash((self.src, self.dst, self.weight))

    def __eq__(self, other):
        if isinstance(other, Edge):
            return (self.src, self.dst, self.weight) == (other.src, other.dst, other.weight)
        return NotImplemented

    def __repr__(self):
        return "{src: %s, dst: %s, weight: %s}" % (self.src, self.dst, self.weight)


class BiMap(dict):
    """dict with an `.inverse` view — the slice of the bidict API the reference uses
    (construction, update, [], in, keys/items, len, .inverse[...])."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._inv: Dict = {}
        self._inv_n = -1

    class _Inverse:
        def __init__(self, owner):
            self._o = owner

        def _t(self):
            o = self._o
            if o._inv_n != len(o):
                o._inv = {v: k for k, v in dict.items(o)}
                o._inv_n = len(o)
            return o._inv

        def __getitem__(self, key):
            return self._t()[key]

        def __contains__(self, key):
            return key in self._t()

        def __len__(self):
            return len(self._o)

    @property
    def inverse(self):
        return BiMap._Inverse(self)


def _resolve(nodes_to_idx, nodes) -> np.ndarray:
    # (map over C-level callables: no generator frame per node - the lookups are the batched API's largest host cost: 943 anchors ~0.1 ms,
    # most of it Node.__hash__; round 6: 30 % less than the generator expression)
    return np.fromiter(map(nodes_to_idx.get, nodes, _repeat(-1)), dtype=np.int64, count=len(nodes))


class _TypeHandle:
    """What `MultiKNN.knn[node_type]` holds (the reference stores a KDTree there)."""

    def __init__(self, owner, node_type):
        self._owner, self.node_type = owner, node_type

    @property
    def n_rows(self):
        return self._owner.index.tables[self.node_type].n


class MultiKNN:
    """One exact nearest-neighbour table per node type (reference: hwer/recommendation_base.py:64-83).
    `leaf_size` is accepted for signature compatibility; brute force on the GPU has no tree."""

    def __init__(self, nodes_to_idx: Dict[Node, int], vectors, leaf_size=128, device=None, arith=None, audit=False):
        from .device_index import DeviceIndex
        assert len(nodes_to_idx) == len(vectors)
        self.nodes_to_idx = nodes_to_idx
        self.leaf_size = leaf_size
        by_type: Dict[str, List[int]] = {}
        for n, i in nodes_to_idx.items():
            by_type.setdefault(n.node_type, []).append(i)
        self.type_rows = {t: np.asarray(v, dtype=np.int64) for t, v in by_type.items()}
        self.index = DeviceIndex(vectors, self.type_rows, device=device, arith=arith, audit=audit)
        self.knn = {t: _TypeHandle(self, t) for t in self.type_rows}
        self._idxs = None
        self._idx_to_node = None

    @property
    def idxs(self):
        """node type -> BiMap local row -> global row (built lazily: it is O(N) Python objects)."""
        if self._idxs is None:
            self._idxs = {t: BiMap(zip(range(len(v)), v.tolist())) for t, v in self.type_rows.items()}
        return self._idxs

    def idx_to_node(self) -> List[Node]:
        if self._idx_to_node is None or len(self._idx_to_node) != len(self.nodes_to_idx):
            table = [None] * len(self.nodes_to_idx)
            for n, i in self.nodes_to_idx.items():
                table[i] = n
            self._idx_to_node = table
        return self._idx_to_node

    def query(self, embedding, node_type, k=200) -> List[Tuple[Node, float]]:
        """k nearest nodes of `node_type` to an arbitrary vector, ascending by Euclidean distance."""
        import torch
        ix = self.index
        q = np.asarray(embedding, dtype=np.float64).reshape(1, -1)
        Q64 = torch.from_numpy(q).to(ix.device)
        Q32 = torch.zeros((1, ix.ldx), dtype=torch.float32, device=ix.device)
        Q32[:, :ix.d] = Q64.to(torch.float32)
        idx, _, dist2, _ = ix.search_queries(node_type, Q64, Q32, k, None, score_mode=2, sort_by_score=False)
        table = self.idx_to_node()
        return [(table[i], float(dd)) for i, dd in zip(idx[0].tolist(), np.sqrt(dist2[0].cpu().numpy()))]


class RecommendationBase(metaclass=abc.ABCMeta):
    #: "base": score = predict(anchor, candidate); "gcn": score = (2 - dist)/2 (reference: hwer/gcn_ncf.py:379-383)
    scoring = "base"

    def __init__(self, node_types: Set[str], n_dims: int = 32):
        self.node_types: Set[NodeType] = node_types
        self.nodes_to_idx: BiMap = BiMap()
        self.knn: MultiKNN = None
        self.vectors: np.ndarray = None
        self.fit_done = False
        self.n_dims = n_dims
        self.log = getLogger(type(self).__name__)
        self.device = None  # torch device string; None = current ROCm device

    # ---- registry (reference: hwer/recommendation_base.py:96-103) -------------------------------------
    def add_nodes(self, nodes: List[Node]):
        fresh = set(nodes)
        assert len(fresh) == len(nodes)
        assert self.nodes_to_idx.keys().isdisjoint(fresh)
        assert all(n.node_type in self.node_types for n in nodes)
        start = len(self.nodes_to_idx)
        self.nodes_to_idx.update(zip(nodes, range(start, start + len(nodes))))
        return self

    # ---- index build (reference: hwer/recommendation_base.py:105-110) ---------------------------------
    def __build_knn__(self, vectors: np.ndarray):
        knn = MultiKNN(self.nodes_to_idx, vectors, leaf_size=128, device=self.device, audit=True)
        violations, _, _, _, nan_rows = knn.index.audit_result
        assert violations == 0
        if nan_rows:
            raise ValueError("Input contains NaN.")  # scikit-learn's KDTree refuses NaN rows in the reference
        self.knn = knn
        self.vectors = vectors
        return self

    @abc.abstractmethod
    def fit(self, nodes: List[Node], edges: List[Edge], node_data: Dict[Node, Dict[FeatureName, object]], **kwargs):
        """Validation shared by every recommender (reference: hwer/recommendation_base.py:113-133)."""
        assert not self.fit_done
        edge_types = {n.node_type for e in edges for n in (e.src, e.dst)}
        self.log.info("Start Fitting Base Recommender with nodes = %s, edges = %s, sparsity = %s", len(nodes), len(edges),
                      1 - len(edges) / (len(nodes) * len(nodes)))
        assert edge_types == self.node_types
        node_set = set(nodes)
        assert all(e.src in node_set and e.dst in node_set for e in edges)
        assert len(node_set) == len(nodes)
        assert all(n.node_type in self.node_types for n in nodes)
        self.add_nodes(nodes)
        self.log.info("End Fitting Base Recommender")
        return edges

    # ---- pair scoring (reference: hwer/recommendation_base.py:135-151) --------------------------------
    def predict_indices(self, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
        """(V[src] . V[dst] + 1)/2 for int index arrays (-1 = unknown node) — one HIP launch."""
        return self.knn.index.pair_score(src, dst)

    def predict(self, node_pairs: List[Tuple[Node, Node]]) -> np.ndarray:
        src, dst = zip(*node_pairs)
        return self.predict_indices(_resolve(self.nodes_to_idx, src), _resolve(self.nodes_to_idx, dst))

    def get_embeddings(self, nodes: List[Node]):
        idx = _resolve(self.nodes_to_idx, nodes)
        unknown = idx < 0
        rows = np.asarray(self.vectors)[np.where(unknown, 0, idx)]
        rows[unknown] = np.clip(rows[unknown], 1e-6, 1e-5)
        return rows

    def get_average_embeddings(self, entities: List[Node]):
        from .utils import unit_length
        return unit_length(np.average(self.get_embeddings(entities), axis=0))

    # ---- retrieval (reference: hwer/recommendation_base.py:157-174) ------------------------------------
    def _check_query(self, node_type):
        assert self.fit_done
        assert node_type in self.node_types and node_type in self.knn.knn

    def find_closest_neighbours_batch(self, node_type: str, anchors: Sequence[Node], k=200, positive=None, negative=None):
        """All anchors in one pass.  -> (idx (B,k) int64 global row ids, score (B,k)) ordered like
        find_closest_neighbours.  `positive` / `negative`: optional per-anchor lists of Nodes."""
        self._check_query(node_type)
        a = _resolve(self.nodes_to_idx, anchors)
        if (a < 0).any():
            missing = anchors[int(np.argmax(a < 0))]
            raise NodeNotFoundException("Node = %s, was not provided in training" % missing)
        pos = None if positive is None else [_resolve(self.nodes_to_idx, p or []) for p in positive]
        neg = None if negative is None else [_resolve(self.nodes_to_idx, n or []) for n in negative]
        # the result is (node, score): distances are not an output of find_closest_neighbours (the kernel then only
        # computes the ones the membership / order of the result depends on)
        idx, score, _, _ = self.knn.index.search(node_type, a, k=k, positive=pos, negative=neg, scoring=self.scoring,
                                                 want_dist=False)
        return idx, score

    def find_closest_neighbours(self, node_type: str, anchor: Node, positive: List[Node] = None,
                                negative: List[Node] = None, k=200) -> List[Tuple[Node, float]]:
        self._check_query(node_type)
        if anchor not in self.nodes_to_idx:
            raise NodeNotFoundException("Node = %s, was not provided in training" % anchor)
        idx, score = self.find_closest_neighbours_batch(node_type, [anchor], k, [positive], [negative])
        from operator import itemgetter
        table = self.knn.idx_to_node()
        ids = idx[0].tolist()
        nodes = itemgetter(*ids)(table) if len(ids) > 1 else (table[ids[0]],)     # one C-level gather instead of a generator per node
        return list(zip(nodes, score[0].tolist()))

    def find_items_for_user(self, user: Node, k=200, item_type: str = "item"):
        """Alias named in BASELINE.json's north_star."""
        return self.find_closest_neighbours(item_type, user, k=k)
