"""ORACLE — CPU restatement of the hwer retrieval hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module.  The product package (`hybrid-weighted-embedding-recommender_amd/hwer`) never
does: it fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function below is checked against golden vectors captured by
running the *unmodified* reference (imported from /root/reference with the harness shims in
tools/ref_shims.py) — see tools/capture_goldens.py and tests/test_oracle_golden.py.

All `file:line` citations are relative to the reference tree (/root/reference).

The reference is numpy + scikit-learn; this restatement is numpy, so element-wise arithmetic is
the same IEEE arithmetic.  Three summation orders matter for bit-exactness and were verified
empirically against numpy 2.2.6 / scikit-learn 1.7.2 (tools/capture_goldens.py `arith` scenario):

  * `(a*b).sum(1)` and `np.linalg.norm(a, axis)`   -> numpy pairwise summation (`pairwise_sum`)
  * `KDTree.query` (sklearn.neighbors, euclidean)     -> sequential `d += (q_j-x_j)**2`, then sqrt
  * `ndarray.mean(0)` over rows                      -> sequential row adds, then / count
"""
from __future__ import annotations

import operator
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------------------
# scalar arithmetic orders
# --------------------------------------------------------------------------------------


def pairwise_sum(a: np.ndarray):
    """numpy's pairwise summation of a 1-D array (numpy/_core/src/umath/loops_utils.h.src
    `@TYPE@_pairwise_sum`, PW_BLOCKSIZE=128), i.e. the order behind `.sum(1)` used at
    hwer/recommendation_base.py:142 and inside `norm` at hwer/utils.py:44."""
    n = len(a)
    t = a.dtype.type
    if n < 8:
        r = t(0)
        for v in a:
            r = t(r + v)
        return r
    if n <= 128:
        r = [a[k] for k in range(8)]
        i = 8
        lim = n - (n % 8)
        while i < lim:
            for k in range(8):
                r[k] = t(r[k] + a[i + k])
            i += 8
        res = t(t(t(r[0] + r[1]) + t(r[2] + r[3])) + t(t(r[4] + r[5]) + t(r[6] + r[7])))
        while i < n:
            res = t(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return t(pairwise_sum(a[:n2]) + pairwise_sum(a[n2:]))


def unit_length(a: np.ndarray, axis: int = 0) -> np.ndarray:
    """hwer/utils.py:43-44 — `a / expand_dims(norm(a, axis), axis)`; no epsilon (zero row -> NaN)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        return a / np.expand_dims(np.linalg.norm(a, axis=axis), axis=axis)


def unit_length_violations(a: np.ndarray, axis: int = 0, epsilon: float = 1e-4):
    """hwer/utils.py:51-57 — returns (violations, mean|len-1|, positive, negative)."""
    lengths = np.expand_dims(np.linalg.norm(a, axis=axis), axis=axis)
    pos = np.sum(lengths > 1 + epsilon)
    neg = np.sum(lengths < 1 - epsilon)
    return pos + neg, np.mean(np.abs(lengths - 1)), pos, neg


# --------------------------------------------------------------------------------------
# array-level restatements (what the HIP kernels are compared with)
# --------------------------------------------------------------------------------------


def gather_rows(V: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """hwer/recommendation_base.py:146-151 on resolved indices: idx == -1 means "unknown node":
    row 0 clipped element-wise to [1e-6, 1e-5]."""
    idx = np.asarray(idx)
    mask = idx == -1
    E = V[np.where(idx >= 0, idx, 0)]
    E[mask] = np.clip(E[mask], 1e-6, 1e-5)
    return E


def pair_score(V: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """hwer/recommendation_base.py:141-144 (same body hwer/gcn_ncf.py:331-334):
    `((E_src * E_dst).sum(1) + 1) / 2` in the dtype of V."""
    r = (gather_rows(V, src) * gather_rows(V, dst)).sum(1)
    return (r + 1) / 2


def average_embedding(V: np.ndarray, idx: Sequence[int]) -> np.ndarray:
    """hwer/recommendation_base.py:153-155 — unit(mean_rows(get_embeddings(entities)))."""
    return unit_length(np.average(gather_rows(V, np.asarray(idx, dtype=np.int64)), axis=0))


def query_vector(V: np.ndarray, anchor: int, positive: Sequence[int] = (), negative: Sequence[int] = ()):
    """hwer/recommendation_base.py:164-170 — mean of [unit(anchor), unit(mean pos)?, -unit(mean neg)?];
    NOT re-normalised."""
    lst = [average_embedding(V, [anchor])]
    if positive is not None and len(positive) > 0:
        lst.append(average_embedding(V, positive))
    if negative is not None and len(negative) > 0:
        lst.append(-1 * average_embedding(V, negative))
    return np.average(lst, axis=0)


def dist2_sequential(q: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Squared Euclidean distance exactly as sklearn's `euclidean_rdist64`
    (sklearn/metrics/_dist_metrics.pyx.tp; call site hwer/recommendation_base.py:74,79):
    float64, `d += tmp*tmp` for j = 0..d-1 in order, no FMA."""
    q = np.asarray(q, dtype=np.float64)
    X = np.asarray(X)
    acc = np.zeros(X.shape[0], dtype=np.float64)
    for j in range(X.shape[1]):
        tmp = q[j] - X[:, j].astype(np.float64)
        acc += tmp * tmp
    return acc


def knn_bruteforce(q: np.ndarray, X: np.ndarray, k: int):
    """Exact k nearest rows of X to q, equal to `KDTree(X).query([q], k)` (hwer/recommendation_base.py:79)
    except for the order inside exact-distance ties, which this build defines as ascending row id
    (the KD-tree's tie order is a traversal artefact).
    Returns (local_idx int64 (k,), dist float64 (k,)) ascending by distance."""
    n = X.shape[0]
    if k > n:
        raise ValueError("k must be less than or equal to the number of training points")
    d2 = dist2_sequential(q, X)
    order = np.lexsort((np.arange(n), d2))[:k]
    return order.astype(np.int64), np.sqrt(d2[order])


def find_closest(V: np.ndarray, type_rows: np.ndarray, anchor: int, positive=(), negative=(), k: int = 200,
                 scoring: str = "base"):
    """hwer/recommendation_base.py:157-174 on resolved indices.
    `type_rows` = ascending-insertion global row ids of the retrieved node type (:68-73).
    scoring="base": score = predict(anchor, cand) (:172), stable sort desc (:173).
    scoring="gcn" : score = (2 - dist)/2 (hwer/gcn_ncf.py:379-383), stable sort desc.
    Returns (global_idx int64 (k,), score (k,), dist float64 (k,)) in final order."""
    q = query_vector(V, anchor, positive, negative)
    local, dist = knn_bruteforce(q, V[type_rows], k)
    cand = np.asarray(type_rows)[local]
    if scoring == "base":
        score = pair_score(V, np.full(len(cand), anchor, dtype=np.int64), cand)
    elif scoring == "gcn":
        score = (-1 * dist + 2) / 2
    else:
        raise ValueError(scoring)
    order = np.argsort(-score, kind="stable")
    return cand[order], score[order], dist[order]


def segment_mean(E: np.ndarray, indptr: np.ndarray, indices: np.ndarray) -> np.ndarray:
    """hwer/content_recommender.py:83 — A[n] = E[adj(n)].mean(0): sequential row adds in
    adjacency order, then / deg."""
    out = np.empty_like(E)
    for n in range(len(indptr) - 1):
        out[n] = E[indices[indptr[n]:indptr[n + 1]]].mean(0)
    return out


def adjacency_csr(n_nodes: int, src: np.ndarray, dst: np.ndarray):
    """hwer/content_recommender.py:77-81 — edges + one self-loop per node appended AFTER the real
    edges; for every (s, t): adj[s].append(t); adj[t].append(s).  Self loops therefore list the node
    twice, parallel edges repeat, weights are ignored.  Returns CSR (indptr int64, indices int32)
    preserving append order."""
    s = np.concatenate([np.asarray(src, dtype=np.int64), np.arange(n_nodes, dtype=np.int64)])
    t = np.concatenate([np.asarray(dst, dtype=np.int64), np.arange(n_nodes, dtype=np.int64)])
    owner = np.empty(2 * len(s), dtype=np.int64)
    other = np.empty(2 * len(s), dtype=np.int64)
    owner[0::2], other[0::2] = s, t
    owner[1::2], other[1::2] = t, s
    order = np.argsort(owner, kind="stable")
    indptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(owner, minlength=n_nodes), out=indptr[1:])
    return indptr, other[order].astype(np.int32)


def content_tail(E: np.ndarray, src: np.ndarray, dst: np.ndarray):
    """hwer/content_recommender.py:77-86 and :96 — returns (adjacent_vectors, unit_rows((E + A)/2))."""
    indptr, indices = adjacency_csr(E.shape[0], src, dst)
    A = segment_mean(E, indptr, indices)
    return A, unit_length((E + A) / 2.0, axis=1)



# --------------------------------------------------------------------------------------
# fit-side table construction: hwer/content_recommender.py:24-76
# --------------------------------------------------------------------------------------
def pca_batches(n: int, batch_size: int, min_batch_size: int):
    """sklearn.utils.gen_batches as IncrementalPCA.fit calls it (min_batch_size = n_components): a short last batch is
    folded into the one before it."""
    out, start = [], 0
    for _ in range(int(n // batch_size)):
        end = start + batch_size
        if end + min_batch_size > n:
            continue
        out.append((start, end))
        start = end
    if start < n:
        out.append((start, n))
    return out


def svd_flip_v(Vt: np.ndarray) -> np.ndarray:
    """sklearn.utils.extmath.svd_flip(u_based_decision=False): the entry of largest magnitude of every row is positive."""
    j = np.argmax(np.abs(Vt), axis=1)
    return Vt * np.sign(Vt[np.arange(Vt.shape[0]), j])[:, None]


def incremental_pca_fit_transform(X: np.ndarray, n_components: int, batch_size: int = 2 ** 16):
    """`IncrementalPCA(n_components, batch_size).fit_transform(X)` (hwer/content_recommender.py:68-69) restated from
    scikit-learn 1.7.2's published algorithm (sklearn/decomposition/_incremental_pca.py partial_fit, Ross et al. 2008):
    batch 0: SVD of the centred batch; batch b > 0: SVD of [S * V ; batch - its mean ; sqrt(n_seen n_b / n_tot) (mean - batch
    mean)], keeping n_components rows; transform = X V^T - mean V^T (sklearn/decomposition/_base.py _transform).
    Returns (scores, components, mean, singular_values)."""
    from scipy import linalg
    X = np.asarray(X, dtype=np.float64)
    n, w = X.shape
    if not n_components <= w:
        raise ValueError("n_components=%r invalid for n_features=%d, need more rows than columns for IncrementalPCA "
                         "processing" % (n_components, w))
    n_seen, mean, comps, sv = 0, np.zeros(w), None, None
    for a, b in pca_batches(n, batch_size, n_components):
        Xb = X[a:b].copy()
        nb = b - a
        if comps is None and n_components > nb:
            raise ValueError("n_components=%d must be less or equal to the batch number of samples %d for the first "
                             "partial_fit call." % (n_components, nb))
        n_tot = n_seen + nb
        col_mean = (mean * n_seen + Xb.sum(axis=0)) / n_tot       # _incremental_mean_and_var
        if n_seen == 0:
            Xb -= col_mean
            stack = Xb
        else:
            bm = np.mean(Xb, axis=0)
            Xb -= bm
            corr = np.sqrt((n_seen / n_tot) * nb) * (mean - bm)
            stack = np.vstack((sv.reshape((-1, 1)) * comps, Xb, corr))
        _, S, Vt = linalg.svd(stack, full_matrices=False, check_finite=False)
        Vt = svd_flip_v(Vt)
        comps, sv, mean, n_seen = Vt[:n_components], S[:n_components], col_mean, n_tot
    out = X @ comps.T
    out -= mean.reshape(1, -1) @ comps.T
    return out, comps, mean, sv


def feature_table(blocks: Sequence[np.ndarray], type_id: np.ndarray) -> np.ndarray:
    """The table IncrementalPCA sees (hwer/content_recommender.py:50-66): block-diagonal concat of the per-type blocks
    (block t = unit rows of type t in `nodes` order), rows put back into `nodes` order, then the one-hot node type
    (OneHotEncoder: categories sorted; here `type_id` already indexes the sorted type names).  Column order of the blocks =
    order of `blocks`."""
    type_id = np.asarray(type_id)
    N = len(type_id)
    T = len(blocks)
    width = sum(b.shape[1] for b in blocks)
    table = np.zeros((N, width + T))
    c = 0
    for t, b in enumerate(blocks):
        rows = np.flatnonzero(type_id == t)
        assert len(rows) == b.shape[0]
        table[rows, c:c + b.shape[1]] = b
        c += b.shape[1]
    table[np.arange(N), width + type_id] = 1.0
    return table


def fit_feature_table(blocks: Sequence[np.ndarray], type_id: np.ndarray, n_dims, block_order: Optional[Sequence[int]] = None):
    """hwer/content_recommender.py:50-76: feature table -> IncrementalPCA -> unit rows -> + one-hot type padded to a power
    of two.  `block_order`: the order in which the reference visited the node types (its `set` iteration order; only the
    column order of the table depends on it).  Returns (E, pca scores)."""
    T = len(blocks)
    order = list(range(T)) if block_order is None else list(block_order)
    type_id = np.asarray(type_id)
    width = sum(b.shape[1] for b in blocks)
    table = np.zeros((len(type_id), width + T))
    c = 0
    for t in order:
        rows = np.flatnonzero(type_id == t)
        table[rows, c:c + blocks[t].shape[1]] = blocks[t]
        c += blocks[t].shape[1]
    table[np.arange(len(type_id)), width + type_id] = 1.0
    if n_dims is None or np.isinf(n_dims):
        n_dims = 2 ** int(np.log2(table.shape[1]))
    scores, _, _, _ = incremental_pca_fit_transform(table, int(n_dims))
    unit = unit_length(scores, axis=1)
    ncol = 2 ** int(np.ceil(np.log2(T)))
    ohe = np.zeros((len(type_id), ncol))
    ohe[np.arange(len(type_id)), type_id] = 1.0
    return np.concatenate((unit, ohe), axis=1), scores

def blend_normalize(tables: Sequence[np.ndarray], row_idx: Sequence[Optional[np.ndarray]], alpha: np.ndarray):
    """Generalisation the build exposes: out[r] = unit(sum_t alpha[r,t] * T_t[i_t[r]]),
    accumulated left to right in float64.  With T=(E, A), alpha=(.5,.5) it equals
    hwer/content_recommender.py:85 + :96 ((E+A)/2 == .5E + .5A exactly)."""
    R = len(row_idx[0]) if row_idx[0] is not None else tables[0].shape[0]
    alpha = np.asarray(alpha, dtype=np.float64)
    acc = np.zeros((R, tables[0].shape[1]), dtype=np.float64)
    for t, T in enumerate(tables):
        rows = T if row_idx[t] is None else T[row_idx[t]]
        a = alpha[t] if alpha.ndim == 1 else alpha[:, t][:, None]
        acc = acc + a * rows.astype(np.float64)
    return unit_length(acc, axis=1)


def merge_topk(dist_lists: Sequence[np.ndarray], idx_lists: Sequence[np.ndarray], k: int):
    """Multi-GPU merge restated (no reference counterpart — SURVEY.md §8e): global k nearest =
    k smallest (dist2, idx) over the union of the per-shard lists."""
    d = np.concatenate(dist_lists)
    i = np.concatenate(idx_lists)
    order = np.lexsort((i, d))[:k]
    return d[order], i[order]


# --------------------------------------------------------------------------------------
# ranking metrics (hwer/utils.py) and the evaluation loop (hwer/validation.py)
# --------------------------------------------------------------------------------------


def reciprocal_rank(y_true, y_pred):
    """hwer/utils.py:71-78"""
    y_true = set(y_true)
    for i, e in enumerate(y_pred):
        if e in y_true:
            return 1.0 / (i + 1)
    return 0.0


def ndcg(y_true: Dict, y_pred: List):
    """hwer/utils.py:101-107"""
    ts = sorted(y_true.values(), reverse=True)[:len(y_pred)]
    idcg = np.sum((np.power(2, ts) - 1) / (np.log2(np.arange(len(ts)) + 2)))
    g = [y_true[i] if i in y_true else 0 for i in y_pred]
    dcg = np.sum((np.power(2, g) - 1) / (np.log2(np.arange(len(g)) + 2)))
    return dcg / (idcg + 1e-8)


def binary_ndcg(y_true: Dict, y_pred: List):
    """hwer/utils.py:110-111"""
    return ndcg({k: 1 for k in y_true}, y_pred)


def binary_ndcg_v2(y_true: List, y_pred: List):
    """hwer/utils.py:114-115"""
    return ndcg({k: 1 for k in y_true}, y_pred)


def recall(y_true, y_pred):
    """hwer/utils.py:118-121 — hits / max(min(|pred|, |true|), 1)"""
    norm = min(len(y_pred), len(y_true))
    hits = sum(1 if i in y_true else 0 for i in y_pred)
    return hits / max(norm, 1.0)


def retrieval_metrics(predictions: Dict[int, List[int]], train_edges, validation_edges, n_items_of_type: int):
    """hwer/validation.py:114-183 minus `ncf_eval` (which draws from `random`): the seen-item filter,
    the @10/@100 cuts and the means over validation users.  Entities are plain ints here.
    `predictions[u]` = ranked item ids (already sorted by score desc, as :132-133 would leave them)."""
    train_uid = defaultdict(set)
    for u, i, r in train_edges:
        train_uid[u].add(i)
    p10, p100 = {}, {}
    for u, ranked in predictions.items():
        filt = [x for x in ranked if x not in train_uid[u]]
        p10[u], p100[u] = filt[:10], filt[:100]
    val_act = defaultdict(list)
    for u, i, r in validation_edges:
        val_act[u].append((i, r))
    val_users = list(val_act.keys())
    score_dict = {}
    for u, lst in val_act.items():
        rem = sorted(lst, key=operator.itemgetter(1), reverse=True)
        rem = [x for x in rem if x[0] not in train_uid[u]]
        score_dict[u] = dict(rem)
    flat = set()
    for v in p100.values():
        flat.update(v)
    return {
        "recall@100": float(np.mean([recall(score_dict[u], p100[u]) for u in val_users])),
        "ndcg_b@100": float(np.mean([binary_ndcg(score_dict[u], p100[u]) for u in val_users])),
        "ndcg_b@10": float(np.mean([binary_ndcg(score_dict[u], p10[u]) for u in val_users])),
        "recall@10": float(np.mean([recall(score_dict[u], p10[u]) for u in val_users])),
        "diversity": len(flat) / n_items_of_type,
    }


# --------------------------------------------------------------------------------------
# object-level restatement (Node / registry / RecommendationBase) used by API-level goldens
# --------------------------------------------------------------------------------------


class Node:
    """hwer/recommendation_base.py:19-36 — identity is (node_type, str(external_id))."""

    def __init__(self, node_type, node_external_id):
        self.node_type = node_type
        self.node_external_id = str(node_external_id)

    def _key(self):
        return (self.node_type, self.node_external_id)

    def __hash__(self):
        return hash(self._key())

    def __eq__(self, other):
        return self._key() == other._key() if isinstance(other, Node) else NotImplemented

    def __repr__(self):
        return str(self._key())


class NodeNotFoundException(Exception):
    """hwer/utils.py:326-327"""


class OracleRecommender:
    """hwer/recommendation_base.py:86-174 with the KD-tree replaced by `knn_bruteforce`."""

    def __init__(self, node_types, scoring: str = "base"):
        self.node_types = set(node_types)
        self.nodes_to_idx: Dict[Node, int] = {}
        self.idx_to_node: List[Node] = []
        self.vectors = None
        self.type_rows: Dict[str, np.ndarray] = {}
        self.fit_done = False
        self.scoring = scoring

    def add_nodes(self, nodes):  # :96-103
        assert len(set(nodes)) == len(nodes)
        assert set(self.nodes_to_idx.keys()).isdisjoint(set(nodes))
        assert len(set(n.node_type for n in nodes) - self.node_types) == 0
        for n in nodes:
            self.nodes_to_idx[n] = len(self.idx_to_node)
            self.idx_to_node.append(n)
        return self

    def build_knn(self, vectors):  # :105-110 + MultiKNN.__init__ :64-76
        v, _, _, _ = unit_length_violations(vectors, axis=1)
        assert v == 0
        assert len(self.nodes_to_idx) == len(vectors)
        rows = defaultdict(list)
        for n, i in self.nodes_to_idx.items():
            rows[n.node_type].append(i)
        self.type_rows = {t: np.asarray(r, dtype=np.int64) for t, r in rows.items()}
        self.vectors = vectors
        self.fit_done = True
        return self

    def _idx(self, nodes):
        return np.array([self.nodes_to_idx.get(n, -1) for n in nodes], dtype=np.int64)

    def get_embeddings(self, nodes):  # :146-151
        return gather_rows(self.vectors, self._idx(nodes))

    def predict(self, pairs):  # :135-144
        src, dst = zip(*pairs)
        return pair_score(self.vectors, self._idx(src), self._idx(dst))

    def find_closest_neighbours(self, node_type, anchor, positive=None, negative=None, k=200):  # :157-174
        assert self.fit_done
        assert node_type in self.node_types and node_type in self.type_rows
        if anchor not in self.nodes_to_idx:
            raise NodeNotFoundException("Node = %s, was not provided in training" % anchor)
        idx, score, _ = find_closest(self.vectors, self.type_rows[node_type], self.nodes_to_idx[anchor],
                                     self._idx(positive or []), self._idx(negative or []), k, self.scoring)
        return [(self.idx_to_node[i], s) for i, s in zip(idx, score)]


def reference_shaped_sweep(V: np.ndarray, item_rows: np.ndarray, anchors, k: int = 200):
    """The reference's own retrieval loop, structure for structure, for TIMING the CPU path beside the GPU path (SURVEY.md
    §8d "reference-shaped loop"): one scikit-learn KD-tree over the item rows (hwer/recommendation_base.py:74,
    leaf_size=128), then per anchor - in a Python loop like hwer/validation.py:30-35 - unit(anchor) (:153-155,164),
    `tree.query(k)` (:79), the re-score of every candidate against the anchor (:141-143,172) and the stable sort by score
    (:173).  Single thread.  Returns (global idx (len(anchors), k), score).  Needs scikit-learn (test / bench hosts only)."""
    from sklearn.neighbors import KDTree
    X = V[item_rows]
    tree = KDTree(X, leaf_size=128)
    out_idx = np.empty((len(anchors), k), dtype=np.int64)
    out_sc = np.empty((len(anchors), k), dtype=V.dtype)
    for n, a in enumerate(anchors):
        emb = unit_length(np.average(V[[a]], axis=0))
        dist, loc = tree.query([emb], k=k)
        rows = item_rows[loc[0]]
        sc = ((V[[a] * k] * V[rows]).sum(1) + 1) / 2
        order = sorted(range(k), key=lambda i: sc[i], reverse=True)
        out_idx[n] = rows[order]
        out_sc[n] = sc[order]
    return out_idx, out_sc
