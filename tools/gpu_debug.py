"""Scratch diagnostics run on the GPU box (not a test)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"), os.path.join(ROOT, "tests")]
from conftest import unit_rows_f32
from oracle import hwer_oracle as orc
from hwer.device_index import DeviceIndex

def mk(V, U, I, arith=None):
    return DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, U + I)}, arith=arith)

g = np.load(os.path.join(ROOT, "tests/golden/g4_predict.npz"))
for flav in ("f64", "f32"):
    V = g["V32"].astype(np.float64) if flav == "f64" else g["V32"]
    ix = mk(V, 64, 512)
    r = ix.pair_score(g["src"], g["dst"])
    ref = g["score_" + flav]
    bad = np.nonzero(r != ref)[0]
    unk = (g["src"] < 0) | (g["dst"] < 0)
    print("pair", flav, "mismatch", len(bad), "of", len(r), "unknown among bad", unk[bad].sum(), "maxdiff", np.abs(r - ref).max())

g1 = np.load(os.path.join(ROOT, "tests/golden/g1_knn_basic.npz"))
for flav in ("f64", "f32"):
    V = g1["V32"].astype(np.float64) if flav == "f64" else g1["V32"]
    ix = mk(V, 64, 512)
    for k in (1, 10, 200):
        idx, score, dist, flags = ix.search("item", np.arange(64), k=k)
        print("knn", flav, k, "flags", np.bincount(flags, minlength=3), "idx eq", np.array_equal(idx, g1[f"idx_{flav}_k{k}"]),
              "score eq", np.array_equal(score, g1[f"score_{flav}_k{k}"]), "rows bad", (idx != g1[f"idx_{flav}_k{k}"]).any(1).sum())
    Q64, Q32, a = ix.build_queries(np.arange(64))
    ref = np.stack([orc.query_vector(V, u) for u in range(64)]).astype(np.float64)
    print("query", flav, "eq", np.array_equal(Q64.cpu().numpy(), ref), np.abs(Q64.cpu().numpy() - ref).max())
for (d, N, B, k) in [(64, 5000, 300, 200), (128, 9000, 257, 200), (256, 3000, 100, 50), (66, 2500, 70, 33)]:
    V32 = np.concatenate([unit_rows_f32(5 + d, B, d), unit_rows_f32(6 + d, N, d)])
    V = V32.astype(np.float64)
    ix = mk(V, B, N, 0)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    nb = 0
    for u in range(0, B, 10):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        nb += not (np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist))
    print("rand", d, N, B, k, "flags", np.bincount(flags, minlength=3), "bad queries", nb)
