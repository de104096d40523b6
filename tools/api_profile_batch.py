#!/usr/bin/env python3
"""Where does the batched retrieval of all anchors (validation.model_get_topk: the reference's retrieval_time call) spend its time?  C1 shape."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402


def main():
    from hwer import Node, RecommendationBase
    from hwer import validation
    U, I, d = 943, 1682, 64
    rng = np.random.default_rng(0)
    V = rng.standard_normal((U + I, d)).astype(np.float32).astype(np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    V = V.astype(np.float32)

    class M(RecommendationBase):
        def fit(self, *a, **k):
            pass
    m = M({"user", "item"}, d)
    nodes = [Node("user", i) for i in range(U)] + [Node("item", i) for i in range(I)]
    m.add_nodes(nodes)
    m._RecommendationBase__build_knn__(V) if hasattr(m, "_RecommendationBase__build_knn__") else m.__build_knn__(V)
    m.fit_done = True
    users = nodes[:U]
    for _ in range(5):
        validation.model_get_topk(m, users, "item")
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        validation.model_get_topk(m, users, "item")
    print("per call %.1f us" % ((time.perf_counter() - t0) / n * 1e6))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        validation.model_get_topk(m, users, "item")
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
