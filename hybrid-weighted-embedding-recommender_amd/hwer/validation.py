"""Evaluation callers of the hot path — drop-in for the retrieval part of the reference's hwer/validation.py.

`model_get_topk` (the hook extraction_efficiency takes, reference hwer/validation.py:30-38,100,111) is one
batched GPU call instead of a per-anchor Python loop; the ranking metrics are the reference's, computed on the
host.  The HPO / plotting / results-table drivers of the reference file are out of scope.
"""
from __future__ import annotations

import ctypes
import operator
import random
import time
from collections import defaultdict
from collections.abc import MutableMapping
from typing import Dict, List, Tuple

import numpy as np

from .recommendation_base import Edge, Node, NodeType
from .utils import binary_ndcg, binary_ndcg_v2, ndcg, recall, reciprocal_rank


def model_get_topk_arrays(model, anchors: List[Node], node_type: NodeType, k: int = 200):
    """-> (idx (U,k) int64 global row ids, score (U,k)); no Python objects per neighbour."""
    return model.find_closest_neighbours_batch(node_type, anchors, k=k)


class LazyRows(MutableMapping):
    """{key: list} with `defaultdict(list)` semantics (what the reference returns, hwer/validation.py:30-35) whose lists are built
    from array rows on first access (and cached).  The retrieval sweep returns U x 200 neighbours; as eager Python objects that is
    millions of tuples built after a sub-millisecond search - most callers only look at some users, or only iterate once.
    Full dict API: `[]` (an unknown key gives - and stores - an empty list, like the reference's defaultdict), `in`, `len`,
    iteration, keys / values / items, get, assignment, `del`, pop, update."""

    def __init__(self, keys, build_row):
        self._keys = keys        # the key -> row dictionary is built on first use (one hash per key: as much host time as resolving the anchors)
        self._pos_dict = None
        self._build = build_row
        self._cache = {}

    @property
    def _pos(self):              # keys whose list still lives in the arrays (or in the cache)
        if self._pos_dict is None:
            self._pos_dict = {k: j for j, k in enumerate(self._keys)}
            self._keys = None
        return self._pos_dict

    def __getitem__(self, key):
        got = self._cache.get(key)
        if got is None:
            j = self._pos.get(key)
            got = self._cache[key] = self._build(j) if j is not None else []   # defaultdict(list).__missing__
            if j is None:
                self._pos[key] = None
        return got

    def get(self, key, default=None):   # like dict.get: no insertion for unknown keys
        return self[key] if key in self._pos else default

    def __setitem__(self, key, value):
        self._pos.setdefault(key, None)
        self._cache[key] = value

    def __delitem__(self, key):
        del self._pos[key]              # KeyError for an unknown key, like a dict
        self._cache.pop(key, None)

    def __iter__(self):
        return iter(self._pos)

    def __len__(self):
        return len(self._pos)

    def __contains__(self, key):
        return key in self._pos


def model_get_topk_knn_per_anchor(model, anchors: List[Node], node_type: NodeType) -> Dict[Node, List[Tuple[Node, float]]]:
    """The reference's loop as it stands (hwer/validation.py:30-35): one `find_closest_neighbours` call per anchor, every list
    materialised.  Each call is one hipGraph replay (DeviceIndex: _OneAnchorSearch); `model_get_topk_knn` below is the batched form."""
    results = defaultdict(list)
    for u in anchors:
        results[u] = model.find_closest_neighbours(node_type, u)
    return results


def model_get_topk_knn(model, anchors: List[Node], node_type: NodeType) -> Dict[Node, List[Tuple[Node, float]]]:
    """{anchor: [(Node, score)] * 200} ordered by score desc (reference: hwer/validation.py:30-35).  ONE batched search;
    the per-anchor lists of (Node, score) tuples are materialised lazily (LazyRows)."""
    anchors = list(anchors)
    idx, score = model_get_topk_arrays(model, anchors, node_type)
    table = model.knn.idx_to_node()
    return LazyRows(anchors, lambda j: [(table[i], s) for i, s in zip(idx[j].tolist(), score[j].tolist())])


model_get_topk = model_get_topk_knn


def ncf_eval(model, train_edges: List[Edge], validation_edges: List[Edge], item_list: List[Node]):
    """Hit-rate / NDCG@10 of each validation edge against 100 sampled unseen items (reference:
    hwer/validation.py:68-97); all users' 101 candidates are scored by ONE pair_score launch."""
    pool = set(item_list)
    seen = defaultdict(set)
    for u, i, _ in list(train_edges) + list(validation_edges):
        seen[u].add(i)
    cases = {}
    for u, i, _ in validation_edges:
        cases[u] = [i, *random.sample(list(pool - seen[u]), 100)]
    users = list(cases)
    pairs = [(u, it) for u in users for it in cases[u]]
    scores = np.asarray(model.predict(pairs)).reshape(len(users), 101) if pairs else np.zeros((0, 101))
    hr, nd = [], []
    for u, row in zip(users, scores):
        order = np.argsort(-row, kind="stable")[:10]
        top = [cases[u][j] for j in order]
        target = cases[u][0]
        hr.append(target in top)
        nd.append(binary_ndcg_v2([target], top))
    return {"ncf_hr": float(np.mean(hr)) if hr else 0.0, "ncf_ndcg": float(np.mean(nd)) if nd else 0.0}


def link_prediction_accuracy(model, nodes: List[Node], train_edges: List[Edge], validation_edges: List[Edge], m: int = 10):
    """Average precision / accuracy of edge-vs-random-pair classification at threshold 0.5 (reference:
    hwer/validation.py:41-65).  Each split is one pair_score launch over |E| * (1 + m) pairs."""
    from sklearn.metrics import accuracy_score, average_precision_score, precision_recall_fscore_support
    out = {}
    for name, edges in (("train", train_edges), ("val", validation_edges)):
        n = len(edges)
        pairs = [(e.src, e.dst) for e in edges] + list(zip(random.choices(nodes, k=n * m), random.choices(nodes, k=n * m)))
        labels = [1] * n + [0] * (n * m)
        p = np.asarray(model.predict(pairs))
        prec, rec, _, _ = precision_recall_fscore_support(labels, p >= 0.5, average="binary")
        out.update({"lp_%s_ap" % name: average_precision_score(labels, p), "lp_%s_precision" % name: prec,
                    "lp_%s_recall" % name: rec, "lp_%s_accuracy" % name: accuracy_score(labels, p >= 0.5)})
    return out


def extraction_efficiency(model, train_edges: List[Edge], validation_edges: List[Edge], get_topk, node_type: NodeType):
    """Recall@K / NDCG / diversity of the retrieval sweep (reference: hwer/validation.py:100-187): anchors are
    every edge source; a user's train items are filtered out of its 200 neighbours before the @10/@20/@50/@100
    cuts; the truth is the user's validation items not seen in train.

    With the default retrieval hook the whole post-processing (seen-item filter, cuts, hit marking, coverage) runs on
    the device (hwer_rank_filter_seen) and only integers come back; a custom `get_topk` takes the host path below."""
    if get_topk in (model_get_topk, model_get_topk_knn) and getattr(getattr(model, "knn", None), "index", None) is not None:
        return _extraction_efficiency_device(model, train_edges, validation_edges, node_type)
    return _extraction_efficiency_host(model, train_edges, validation_edges, get_topk, node_type)


def _csr_rows(lists):
    """list of 1-D int arrays -> (ptr int64 (n+1), flat int64) with every row sorted ascending."""
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.concatenate([np.sort(np.asarray(l, dtype=np.int64)) for l in lists]) if len(lists) and ptr[-1] else np.zeros(0, dtype=np.int64)
    return ptr, flat


_CUTS = (10, 20, 50, 100)


def _unpack_hit_bits(words: np.ndarray, n_bits: int) -> np.ndarray:
    """(V, W) uint64 words -> (V, n_bits) int64 0/1, bit p of the row = bit p % 64 of word p // 64."""
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=1, bitorder="little")   # little-endian words
    return bits[:, :n_bits].astype(np.int64)


#: None until the first device evaluation of this process has compared hwer_rank_metrics' means with numpy's (see below)
_DEVICE_MEANS_MATCH_NUMPY = None


def recall_rows(hitbits, n_pred, n_true, c):
    """mean over users of utils.recall(truth, pred[:c]) from hit marks (reference: hwer/utils.py:118-121)."""
    if len(n_pred) == 0:
        return np.mean([])
    lp = np.minimum(n_pred, c)
    h = (hitbits[:, :c] * (np.arange(c)[None, :] < lp[:, None])).sum(1)
    return np.mean(h / np.maximum(np.minimum(lp, n_true), 1.0))


def binary_ndcg_rows(hitbits, n_pred, n_true, c):
    """mean over users of utils.binary_ndcg(truth, pred[:c]) from hit marks (reference: hwer/utils.py:101-111)."""
    V = len(n_pred)
    if V == 0:
        return np.mean([])
    lp = np.minimum(n_pred, c).astype(np.int64)
    vals = np.empty(V, dtype=np.float64)
    ideal_len = np.minimum(n_true, lp)
    idcg_of = np.asarray([np.sum((np.power(2, [1] * m) - 1) / np.log2(np.arange(m) + 2)) for m in range(c + 1)])
    for length in np.unique(lp):          # a handful of distinct lengths (nearly every user has c survivors)
        rows = np.flatnonzero(lp == length)
        length = int(length)
        gains = np.ascontiguousarray(hitbits[rows, :length])
        discounts = np.log2(np.arange(length) + 2)
        dcg = np.sum((np.power(2, gains) - 1) / discounts, axis=1) if length else np.zeros(len(rows))
        vals[rows] = dcg / (idcg_of[ideal_len[rows]] + 1e-8)
    return np.mean(vals)


def _extraction_efficiency_device(model, train_edges, validation_edges, node_type, k: int = 200):
    import torch
    from . import native
    lib = native.lib()
    ix = model.knn.index
    n2i = model.nodes_to_idx
    train_items_of = defaultdict(set)
    train_scores = defaultdict(dict)
    for u, i, r in train_edges:
        train_items_of[u].add(i)
        train_scores[u][i] = r
    train_users = list({u for u, _, _ in train_edges})
    validation_users = list({u for u, _, _ in validation_edges})
    all_users = list(set(train_users) | set(validation_users))
    all_items = [i for i in ({i for _, i, _ in train_edges} | {i for _, i, _ in validation_edges}) if i.node_type == node_type]
    val_pairs = defaultdict(list)
    for u, i, r in validation_edges:
        val_pairs[u].append((i, r))
    val_scores, val_actuals = {}, {}
    for u, lst in val_pairs.items():
        kept = [(i, r) for i, r in sorted(lst, key=operator.itemgetter(1), reverse=True) if i not in train_items_of[u]]
        val_scores[u] = dict(kept)
        val_actuals[u] = [i for i, _ in kept]

    # ---- retrieval: one batched search, results stay on the device ------------------------------------------------------
    model._check_query(node_type)
    for u in all_users:
        if u not in n2i:
            from .utils import NodeNotFoundException
            raise NodeNotFoundException("Node = %s, was not provided in training" % u)
    a = np.asarray([n2i[u] for u in all_users], dtype=np.int64)
    t0 = time.time()
    Q64, Q32, a_dev = ix.build_queries(a)
    idx_dev, score_dev, _, _ = ix.search_queries(node_type, Q64, Q32, k, a_dev, 2 if model.scoring == "gcn" else ix.arith,
                                                 want_dist=False)
    torch.cuda.synchronize()
    retrieval_time = time.time() - t0

    # ---- seen filter + cuts + hits + coverage on the device (K12) ------------------------------------------------------------
    B = len(all_users)
    unknown = lambda n: n2i.get(n, -1)   # noqa: E731  (items that were never registered cannot be retrieved)
    seen_ptr, seen_idx = _csr_rows([[x for x in (unknown(i) for i in train_items_of.get(u, ())) if x >= 0] for u in all_users])
    truth_ptr, truth_idx = _csr_rows([[x for x in (unknown(i) for i in val_actuals.get(u, ())) if x >= 0] for u in all_users])
    rows = ix.tables[node_type].rows
    cover_base, cover_n = int(rows.min()), int(rows.max() - rows.min() + 1)
    dev = ix.device
    T = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
    d_sp, d_si, d_tp, d_ti = T(seen_ptr), T(seen_idx if len(seen_idx) else np.zeros(1, np.int64)), T(truth_ptr), T(truth_idx if len(truth_idx) else np.zeros(1, np.int64))
    cmax = max(_CUTS)
    unseen = torch.empty((B, cmax), dtype=torch.int64, device=dev)
    ulen = torch.empty(B, dtype=torch.int32, device=dev)
    hits = torch.empty((B, (cmax + 63) // 64), dtype=torch.int64, device=dev)
    cover = torch.zeros((cover_n + 31) // 32, dtype=torch.int32, device=dev)
    with native.device_guard(dev):
        native.check(lib.hwer_rank_filter_seen(native.ptr(idx_dev), B, k, native.ptr(d_sp), native.ptr(d_si), native.ptr(d_tp), native.ptr(d_ti),
                                               cmax, native.ptr(unseen), native.ptr(ulen), native.ptr(hits), native.ptr(cover), cover_base,
                                               cover_n, native.stream_ptr(dev)), "rank_filter_seen")
    unseen_h, ulen_h = unseen.cpu().numpy(), ulen.cpu().numpy().astype(np.int64)
    hits_h = hits.cpu().numpy().view(np.uint64)
    n_covered = int(np.unpackbits(cover.cpu().numpy().view(np.uint8)).sum())
    idx_h = idx_dev[:, :100].cpu().numpy()

    # ---- float metrics from the integers, with the reference's own numpy expressions (utils.py:101-121), vectorised over the
    # users: element-wise operations are identical to the per-user ones, and every per-user np.sum is reproduced by a row sum
    # over a contiguous slice of exactly that user's length (users grouped by length; numpy reduces each contiguous row with
    # the same pairwise summation as a 1-D np.sum, tests/test_host_api.py::test_vectorised_metrics_match_reference_loops)
    pos_of = {u: j for j, u in enumerate(all_users)}
    vrow = np.asarray([pos_of[u] for u in validation_users], dtype=np.int64)
    hitbits = _unpack_hit_bits(hits_h[vrow], cmax)            # hitbits[v, p] = the p-th survivor of user v is truth
    n_true = np.asarray([len(val_scores.get(u, ())) for u in validation_users], dtype=np.int64)
    n_pred = ulen_h[vrow]

    # Round 4: the float metrics themselves come from the device too (hwer_rank_metrics: per-user recall / binary NDCG from the hit masks in
    # np.sum's order, the means over the validation users by np.add.reduce's pairwise recursion - bit-identical to the numpy expressions of
    # recall_rows / binary_ndcg_rows above, which stay as the host statement of the same arithmetic and as the tests' cross-check).
    dev_cuts = (100, 10, 20)
    if len(validation_users):
        log2_tab = np.zeros((cmax + 1, cmax), dtype=np.float64)      # row L: numpy's np.log2 over an array of L positions (see include/hwer_hip.h)
        for L in range(1, cmax + 1):
            log2_tab[L, :L] = np.log2(np.arange(L) + 2)
        idcg_tab = np.asarray([np.sum((np.power(2, [1] * m) - 1) / np.log2(np.arange(m) + 2)) for m in range(cmax + 1)], dtype=np.float64)
        d_vrow, d_ntrue, d_l2, d_id = T(vrow), T(n_true), T(log2_tab), T(idcg_tab)
        vals_ws = torch.empty(2 * len(dev_cuts) * len(vrow), dtype=torch.float64, device=dev)
        means_dev = torch.empty(2 * len(dev_cuts), dtype=torch.float64, device=dev)
        cuts_h = np.asarray(dev_cuts, dtype=np.int32)
        with native.device_guard(dev):
            native.check(lib.hwer_rank_metrics(native.ptr(hits), hits.shape[1], native.ptr(ulen), native.ptr(d_vrow), native.ptr(d_ntrue), len(vrow),
                                               cuts_h.ctypes.data_as(ctypes.c_void_p), len(dev_cuts), native.ptr(d_l2), cmax, native.ptr(d_id), native.ptr(vals_ws),
                                               native.ptr(means_dev), native.stream_ptr(dev)), "rank_metrics")
        means = means_dev.cpu().numpy()
        dev_recall = {c: means[j] for j, c in enumerate(dev_cuts)}
        dev_ndcg = {c: means[len(dev_cuts) + j] for j, c in enumerate(dev_cuts)}
        # The kernel restates numpy's reduce schedule (8 192-element pieces, 128-element pairwise leaves) as numpy 2.2 runs it.  On the
        # FIRST evaluation of a process the device means are held to this numpy's own expressions on the same integers (already on the
        # host): equal -> later calls trust the device; different (a numpy build that reduces in another order) -> the host values are
        # reported from then on, so "bit-identical to the reference's numpy" never silently stops holding.
        # (round 6, advisor: the verdict is LATCHED only by an evaluation large enough to discriminate - numpy's pairwise reduction differs from a
        # sequential one only beyond 8 terms, and a different leaf / unroll schedule needs thousands of users to show up; smaller evaluations are
        # compared every time, the host expressions are cheap there)
        global _DEVICE_MEANS_MATCH_NUMPY
        if _DEVICE_MEANS_MATCH_NUMPY is None:
            host = [recall_rows(hitbits, n_pred, n_true, c) for c in dev_cuts] + [binary_ndcg_rows(hitbits, n_pred, n_true, c) for c in dev_cuts]
            match = bool(np.array_equal(np.asarray(host, dtype=np.float64), means, equal_nan=True))
            if len(n_pred) > 8192 or not match:      # (equal on a small evaluation: not latched - too few users to tell the reduction orders apart)
                _DEVICE_MEANS_MATCH_NUMPY = match
            if not match:
                import warnings
                warnings.warn("hwer_rank_metrics: this numpy (%s) sums in another order than the kernel restates; extraction_efficiency reports the "
                              "host expressions instead of the device means" % np.__version__, RuntimeWarning, stacklevel=2)
        if _DEVICE_MEANS_MATCH_NUMPY is False:
            dev_recall = dev_ndcg = {}
    else:
        dev_recall = dev_ndcg = {}

    def recall_at(c):
        return dev_recall[c] if c in dev_recall else recall_rows(hitbits, n_pred, n_true, c)

    def binary_ndcg_at(c):
        return dev_ndcg[c] if c in dev_ndcg else binary_ndcg_rows(hitbits, n_pred, n_true, c)

    table = model.knn.idx_to_node()
    cuts100 = LazyRows(all_users, lambda j: [table[i] for i in unseen_h[j, :ulen_h[j]].tolist()])
    train_predictions = LazyRows(all_users, lambda j: [table[i] for i in idx_h[j].tolist()])
    train_actuals = {u: [i for i, _ in sorted(d.items(), key=operator.itemgetter(1), reverse=True)]
                     for u, d in train_scores.items()}

    def mean_over(fn, truth, c):
        return np.mean([fn(truth[u], cuts100[u][:c]) for u in validation_users])

    metrics = {"retrieval_time": retrieval_time,
               "recall@100": recall_at(100),
               "ndcg_b@100": binary_ndcg_at(100),
               "ndcg_b@10": binary_ndcg_at(10),
               "recall@10": recall_at(10),
               "diversity": n_covered / len(all_items),
               **ncf_eval(model, train_edges, validation_edges, all_items)}
    return {"actuals": val_actuals, "predictions": cuts100, "train_actuals": train_actuals,
            "train_predictions": train_predictions, "train_actuals_score_dict": dict(train_scores),
            "validation_actuals_score_dict": val_scores, "metrics": metrics,
            "extra": {"mrr": mean_over(reciprocal_rank, val_actuals, 100),
                      "ndcg@100": mean_over(ndcg, val_scores, 100),
                      "recall@20": recall_at(20)}}


def _extraction_efficiency_host(model, train_edges: List[Edge], validation_edges: List[Edge], get_topk, node_type: NodeType):
    """Host post-processing (any `get_topk` returning {anchor: [(Node, score)]}), the reference's own structure."""
    train_items_of = defaultdict(set)
    train_scores = defaultdict(dict)
    for u, i, r in train_edges:
        train_items_of[u].add(i)
        train_scores[u][i] = r
    train_users = list({u for u, _, _ in train_edges})
    validation_users = list({u for u, _, _ in validation_edges})
    all_users = list(set(train_users) | set(validation_users))
    all_items = [i for i in ({i for _, i, _ in train_edges} | {i for _, i, _ in validation_edges}) if i.node_type == node_type]

    t0 = time.time()
    predictions = get_topk(model, all_users, node_type)
    retrieval_time = time.time() - t0

    train_actuals = {u: [i for i, _ in sorted(d.items(), key=operator.itemgetter(1), reverse=True)]
                     for u, d in train_scores.items()}
    cuts = {10: {}, 20: {}, 50: {}, 100: {}}
    train_predictions = {}
    for u, ranked in predictions.items():
        ordered = [n for n, _ in sorted(ranked, key=operator.itemgetter(1), reverse=True)]
        unseen = [n for n in ordered if n not in train_items_of[u]]
        train_predictions[u] = ordered[:100]
        for c in cuts:
            cuts[c][u] = unseen[:c]

    val_pairs = defaultdict(list)
    for u, i, r in validation_edges:
        val_pairs[u].append((i, r))
    val_scores, val_actuals = {}, {}
    for u, lst in val_pairs.items():
        kept = [(i, r) for i, r in sorted(lst, key=operator.itemgetter(1), reverse=True) if i not in train_items_of[u]]
        val_scores[u] = dict(kept)
        val_actuals[u] = [i for i, _ in kept]

    def mean_over(fn, truth, pred):
        return np.mean([fn(truth[u], pred[u]) for u in validation_users])

    covered = set()
    for lst in cuts[100].values():
        covered.update(lst)
    metrics = {"retrieval_time": retrieval_time,
               "recall@100": mean_over(recall, val_scores, cuts[100]),
               "ndcg_b@100": mean_over(binary_ndcg, val_scores, cuts[100]),
               "ndcg_b@10": mean_over(binary_ndcg, val_scores, cuts[10]),
               "recall@10": mean_over(recall, val_scores, cuts[10]),
               "diversity": len(covered) / len(all_items),
               **ncf_eval(model, train_edges, validation_edges, all_items)}
    return {"actuals": val_actuals, "predictions": cuts[100], "train_actuals": train_actuals,
            "train_predictions": train_predictions, "train_actuals_score_dict": dict(train_scores),
            "validation_actuals_score_dict": val_scores, "metrics": metrics,
            "extra": {"mrr": mean_over(reciprocal_rank, val_actuals, cuts[100]),
                      "ndcg@100": mean_over(ndcg, val_scores, cuts[100]),
                      "recall@20": mean_over(recall, val_scores, cuts[20])}}
