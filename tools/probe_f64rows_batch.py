"""Batched search on GENUINE float64 rows (DeviceIndex._search_candidates): where the time goes.  Run under rocprofv3 --kernel-trace --stats:
    python tools/probe_f64rows_batch.py [B] [N] [d]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"))


def main():
    import torch
    from hwer.device_index import DeviceIndex
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 26744
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    rng = np.random.default_rng(3)
    V = rng.standard_normal((B + N, d))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    ix = DeviceIndex(V, {"user": np.arange(B), "item": np.arange(B, B + N)}, arith=0)
    assert ix.V64 is not None
    for _ in range(3):
        ix.search("item", np.arange(B), k=200, want_dist=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        ix.search("item", np.arange(B), k=200, want_dist=False)
        ts.append(time.perf_counter() - t)
    print("B=%d N=%d d=%d: search on float64 rows %.3f ms (median of 10, results on the host)" % (B, N, d, np.median(ts) * 1e3))


if __name__ == "__main__":
    main()
