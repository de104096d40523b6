import sys, time, numpy as np
sys.path.insert(0, "hybrid-weighted-embedding-recommender_amd")
from hwer import Node, VectorIndex
U, N, d, K = 943, 1682, 64, 200
rng = np.random.default_rng(1)
V = rng.standard_normal((U + N, d)); V /= np.linalg.norm(V, axis=1, keepdims=True)
users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(N)]
m = VectorIndex(node_types={"user", "item"}, n_dims=d); m.fit(users + items, V)
ix = m.knn.index
for flag in (True, False):
    ix.one_anchor_graphs = flag
    for u in users[:20]: m.find_closest_neighbours("item", u, k=K)
    ts = []
    for u in users[:300]:
        t = time.perf_counter(); m.find_closest_neighbours("item", u, k=K); ts.append(time.perf_counter() - t)
    print("graphs", flag, "median ms", np.median(ts) * 1e3, "p99", np.percentile(ts, 99) * 1e3)
