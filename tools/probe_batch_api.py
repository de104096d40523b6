"""Wall time of find_closest_neighbours_batch / model_get_topk through the drop-in classes (results on the host): float32-representable and genuine float64 rows."""
import sys, time, numpy as np
sys.path.insert(0, "hybrid-weighted-embedding-recommender_amd")
from hwer import Node, VectorIndex, validation
U, N, d, K = 943, 1682, 64, 200
rng = np.random.default_rng(1)
V = rng.standard_normal((U + N, d)); V /= np.linalg.norm(V, axis=1, keepdims=True)
users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(N)]
for name, X in (("float32-representable rows", V.astype(np.float32).astype(np.float64)), ("genuine float64 rows", V)):
    m = VectorIndex(node_types={"user", "item"}, n_dims=d); m.fit(users + items, X)
    for fn_name, fn in (("find_closest_neighbours_batch", lambda: m.find_closest_neighbours_batch("item", users, k=K)), ("model_get_topk", lambda: validation.model_get_topk(m, users, "item"))):
        for _ in range(3): fn()
        ts = []
        for _ in range(15):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        print("%-28s %-30s %.4f ms (median of 15)" % (name, fn_name, np.median(ts) * 1e3))
