#!/usr/bin/env python3
"""Per-kernel view of small query batches against a 1.25 M x 128 shard (run under rocprofv3 --kernel-trace --stats):
    python tools/lat_probe.py [B] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    sys.path.insert(0, p)
import argparse  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    from hwer import native
    if os.environ.get("HWER_PROBE_K"):       # experiment: neighbours per query (bench.K is read when the workload is built)
        bench.K = int(os.environ["HWER_PROBE_K"])
    lib = native.lib()
    dev = torch.device("cuda", 0)
    w = bench.Workload(torch, lib, bench.CONFIGS[os.environ.get("HWER_PROBE_CFG", "c4")], dev)
    w.step()
    o = w.sets[0]
    ob = dict(idx=o["idx"][:B], score=o["score"][:B], dist2=None, flags=o["flags"][:B])
    opts = (w.opts if B > 1024 else 0) | int(os.environ.get("HWER_BENCH_OPTS", "0"))
    need = lib.hwer_knn_workspace_bytes_opts(B, w.I, w.d, bench.K, opts)      # (the plan - hence the layout - is a function of the options)
    if need > w.ws.numel():
        w.ws = torch.empty(need, dtype=torch.uint8, device=dev)
    ms = bench.cuda_time(torch, lambda: w.stage_search(ob, B=B, opts=opts), reps, warm=3)
    print("B=%d: %.4f ms per search" % (B, ms))


if __name__ == "__main__":
    main()
