"""Scratch reproducer: test_extraction_efficiency_g9 followed by test_merge_topk_matches_single_shard, with a native
backtrace handler installed after the GPU runtime is up."""
import ctypes, os, sys, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"), os.path.join(ROOT, "tests")]
import torch
from conftest import unit_rows_f32
from hwer import Edge, Node, VectorIndex, validation, native
from hwer.device_index import DeviceIndex
from hwer.distributed import merge_on_device, shard_bounds
native.lib()
torch.zeros(1, device="cuda").item()
ctypes.CDLL(os.path.join(ROOT, "tools", "debug", "abort_bt.so"))   # constructor installs the SIGABRT/SIGSEGV handlers LAST
stage = os.environ.get("STAGE", "both")
_orig_check = native.check
def _check(rc, what=""):
    _orig_check(rc, what)
    if os.environ.get("TRACE"):
        torch.cuda.synchronize()
        print("  ok:", what, flush=True)
native.check = _check
import hwer.device_index as _di
_di.native.check = _check
if stage in ("both", "ee"):
    g = np.load(os.path.join(ROOT, "tests", "golden", "g9_metrics.npz"), allow_pickle=False)
    U, I = int(g["ee_n_users"]), int(g["ee_n_items"])
    users = [Node("user", i) for i in range(U)]
    items = [Node("item", i) for i in range(I)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=32)
    m.fit(users + items, g["ee_V32"].astype(np.float64))
    allv = users + items
    train = [Edge(allv[int(u)], allv[int(i)], float(r)) for u, i, r in g["ee_train"]]
    valid = [Edge(allv[int(u)], allv[int(i)], float(r)) for u, i, r in g["ee_valid"]]
    random.seed(0)
    res = validation.extraction_efficiency(m, train, valid, validation.model_get_topk, "item")
    print("ee done", res["metrics"]["recall@10"], flush=True)
d, N, B, k, G = 64, 4000, 50, 100, 4
V32 = np.concatenate([unit_rows_f32(71, B, d), unit_rows_f32(72, N, d)])
V = V32.astype(np.float64)
full = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
Q64, Q32, a = full.build_queries(np.arange(B))
ref = full.search_queries("item", Q64, Q32, k, a)
torch.cuda.synchronize(); print("full done", flush=True)
for r in range(G):
    lo, hi = shard_bounds(N, G, r)
    Vr = np.concatenate([V[:B], V[B + lo:B + hi]])
    ixr = DeviceIndex(Vr, {"user": np.arange(B), "item": np.arange(B, B + hi - lo)}, arith=0)
    torch.cuda.synchronize(); print("shard index", r, flush=True)
    i, s, d2, _ = ixr.search_queries("item", Q64, Q32, k, a, sort_by_score=False, id_base=B + lo)
    torch.cuda.synchronize(); print("shard search", r, flush=True)
print("ok")
