#!/usr/bin/env python3
"""Ablation of the select / re-rank kernel inside real searches of a bench configuration (DIAGNOSTIC library: lib/libhwer_hip_diag.so copied
over lib/libhwer_hip.so by the caller - tools/ab_stage.sh shows the swap).  One process, one resident workload; HWER_SELECT_SKIP is read by the
library at every search call, so the masks are switched between repetitions:
    python tools/select_ablate.py c3 "0 8 16 24 32 64 96 1 2 4"
bits: 1 no re-rank rounds, 2 no sort, 4 no selection passes, 8 no distances, 16 no score arithmetic, 32 no row gathers, 64 no LDS staging.
Prints the event-bracketed kernel times (hwer_profile_enable) of the screen and the select stage per search call; results are WRONG under any mask."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from hwer import native
    name = sys.argv[1]
    masks = [int(x) for x in sys.argv[2].split()]
    var = sys.argv[3] if len(sys.argv) > 3 else "HWER_SELECT_SKIP"
    lib = native.lib()
    dev = torch.device("cuda:0")
    w = bench.Workload(torch, lib, bench.CONFIGS[name], dev)
    for _ in range(2):
        w.step()
    torch.cuda.synchronize()
    o = w.sets[0]
    for rep in range(2):
        for m in masks:
            os.environ[var] = str(m)
            w.stage_search(o)
            torch.cuda.synchronize()
            lib.hwer_profile_enable(1)
            ms = bench.cuda_time(torch, lambda: w.stage_search(o), 4, warm=0)
            lib.hwer_profile_enable(0)
            screen_ms, select_ms, n = bench.read_profile(lib)
            print("%s rep%d %s=%-4d search %.4f ms  screen/launch %.4f  select/launch %.4f  (%d bracketed launches)" % (name, rep, var, m, ms, screen_ms, select_ms, n), flush=True)
    os.environ[var] = "0"


if __name__ == "__main__":
    main()
