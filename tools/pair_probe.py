#!/usr/bin/env python3
"""pair_score throughput on a table beyond the Infinity Cache: d in (64, 128, 256), both flavours; HWER_PAIR_PB (diagnostic library) sweeps the pairs per trip.
    python tools/pair_probe.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"))
import numpy as np
import torch
from hwer import native
lib = native.lib(); P = native.ptr
dev = torch.device("cuda", 0)
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for d, N in ((64, 4000000), (128, 2000000), (256, 1381072)):
    V = torch.randn((N, d), device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(1)
    npairs = 16000000
    src = torch.randint(0, N, (npairs,), generator=g, device=dev, dtype=torch.int32)
    dst = torch.randint(0, N, (npairs,), generator=g, device=dev, dtype=torch.int32)
    for flav, fn, ob, dt in (("f64", lib.hwer_pair_score_f64, 8, torch.float64), ("f32", lib.hwer_pair_score_f32, 4, torch.float32)):
        out = torch.empty(npairs, dtype=dt, device=dev)
        for _ in range(3):
            native.check(fn(P(V), N, d, d, P(src), P(dst), npairs, P(out), S()), "pair_score")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            native.check(fn(P(V), N, d, d, P(src), P(dst), npairs, P(out), S()), "pair_score")
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        byts = npairs * (8.0 * d + 8 + ob)
        print(json.dumps({"d": d, "flavour": flav, "table_MB": round(N * d * 4 / 1e6), "ms": round(ms, 4), "GBs": round(byts / ms / 1e6, 1), "frac_of_8TBs": round(byts / ms / 1e6 / 8000, 3),
                          "pb": os.environ.get("HWER_PAIR_PB", "default")}))
    del V, src, dst
    torch.cuda.empty_cache()
