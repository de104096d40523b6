"""Evaluation callers of the hot path — drop-in for the retrieval part of the reference's hwer/validation.py.

`model_get_topk` (the hook extraction_efficiency takes, reference hwer/validation.py:30-38,100,111) is one
batched GPU call instead of a per-anchor Python loop; the ranking metrics are the reference's, computed on the
host.  The HPO / plotting / results-table drivers of the reference file are out of scope.
"""
from __future__ import annotations

import operator
import random
import time
from collections import defaultdict
from typing import Dict, List, Tuple

import numpy as np

from .recommendation_base import Edge, Node, NodeType
from .utils import binary_ndcg, binary_ndcg_v2, ndcg, recall, reciprocal_rank


def model_get_topk_arrays(model, anchors: List[Node], node_type: NodeType, k: int = 200):
    """-> (idx (U,k) int64 global row ids, score (U,k)); no Python objects per neighbour."""
    return model.find_closest_neighbours_batch(node_type, anchors, k=k)


def model_get_topk_knn(model, anchors: List[Node], node_type: NodeType) -> Dict[Node, List[Tuple[Node, float]]]:
    """{anchor: [(Node, score)] * 200} ordered by score desc (reference: hwer/validation.py:30-35)."""
    anchors = list(anchors)
    idx, score = model_get_topk_arrays(model, anchors, node_type)
    table = model.knn.idx_to_node()
    out = defaultdict(list)
    for u, row, sc in zip(anchors, idx.tolist(), score.tolist()):
        out[u] = [(table[i], s) for i, s in zip(row, sc)]
    return out


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
    cuts; the truth is the user's validation items not seen in train."""
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
