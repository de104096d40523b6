#!/usr/bin/env python3
"""Time the step prologue (hwer_blend_prepare_search_f32) of bench configurations: python tools/prologue_probe.py "c4 c5 c4full"
With --libs "a b c": one child per library variant lib/libhwer_hip_<v>.so swapped in (same-box A/B)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    sys.path.insert(0, p)
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")


def child(cfgs):
    import torch
    import bench
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda", 0)
    out = []
    for c in cfgs:
        w = bench.Workload(torch, lib, bench.CONFIGS[c], dev)
        w.step()
        ms = bench.cuda_time(torch, w.stage_blend_prepare, 20, warm=3)
        out.append("%s %.4f" % (c, ms))
        del w
        torch.cuda.empty_cache()
    print(" | ".join(out), flush=True)


def main():
    if "--libs" in sys.argv:
        i = sys.argv.index("--libs")
        variants = sys.argv[i + 1].split()
        cfgs = sys.argv[1]
        keep = os.path.join(L, "libhwer_hip_keep.so")
        shutil.copy(os.path.join(L, "libhwer_hip.so"), keep)
        try:
            for rep in range(2):
                for v in variants:
                    shutil.copy(os.path.join(L, "libhwer_hip_%s.so" % v), os.path.join(L, "libhwer_hip.so"))
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), cfgs], capture_output=True, text=True, timeout=600)
                    print("lib %s rep%d: %s" % (v, rep, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1]), flush=True)
        finally:
            shutil.copy(keep, os.path.join(L, "libhwer_hip.so"))
        return
    child(sys.argv[1].split())


if __name__ == "__main__":
    main()
