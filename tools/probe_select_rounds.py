"""Dense select (wave per query) against the number of queries: is c2's 6 040-query launch (1.47 rounds of the 4 096 resident waves) paying for a
half-empty second round?  Prints the select stage's time per launch and per query for several batch sizes at c2's catalogue (3 706 x 128)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"))


def main():
    import torch
    import bench
    from hwer import native
    from hwer.device_index import DeviceIndex
    lib = native.lib()
    N, d, k = 3706, 128, 200
    rng = np.random.default_rng(5)
    for B in (2048, 3072, 4096, 5120, 6040, 6144, 8192, 12288, 16384):
        V = rng.standard_normal((B + N, d)).astype(np.float32)
        V /= np.linalg.norm(V, axis=1, keepdims=True)
        ix = DeviceIndex(V, {"user": np.arange(B), "item": np.arange(B, B + N)}, arith=0)
        Q64, Q32, a = ix.build_queries(np.arange(B), None, None)
        for _ in range(3):
            ix.search_queries("item", Q64, Q32, k, a, 0, True, want_dist=False)
        torch.cuda.synchronize()
        lib.hwer_profile_enable(1)
        for _ in range(20):
            ix.search_queries("item", Q64, Q32, k, a, 0, True, want_dist=False)
        torch.cuda.synchronize()
        lib.hwer_profile_enable(0)
        screen_ms, select_ms, n = bench.read_profile(lib)
        print("B = %5d  waves / 4096 = %.2f  screen %.4f ms  select %.4f ms  select per 1 000 queries %.4f ms" % (B, B / 4096.0, screen_ms, select_ms, select_ms / B * 1e3))
        del ix


if __name__ == "__main__":
    main()
