"""Scratch: hwer_merge_topk timing over (G, B, order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")]
import torch
from hwer.distributed import merge_on_device
dev = torch.device("cuda")
K = 200
def t_ms(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for G, B in ((2, 3020), (2, 1510), (2, 755), (3, 2013), (4, 1510), (8, 755), (8, 3020), (16, 755), (17, 755)):
    idx = torch.randint(0, 1 << 20, (G, B, K), device=dev)
    d2 = torch.rand((G, B, K), device=dev, dtype=torch.float64).sort(dim=2).values
    sc = torch.rand((G, B, K), device=dev, dtype=torch.float64)
    sc_mono = 1.0 - d2 / 4
    print("G=%2d B=%4d: by dist %.1f us, by score (random) %.1f us, by score (monotone in dist) %.1f us" % (
        G, B, t_ms(lambda: merge_on_device(idx, sc, d2, False)), t_ms(lambda: merge_on_device(idx, sc, d2, True)),
        t_ms(lambda: merge_on_device(idx, sc_mono, d2, True))))
print("strided (receive-buffer layout):")
for G, B in ((2, 3020), (4, 1510), (8, 755)):
    idx = torch.randint(0, 1 << 20, (G, B, K), device=dev)
    d2 = torch.rand((G, B, K), device=dev, dtype=torch.float64).sort(dim=2).values
    sc = torch.rand((G, B, K), device=dev, dtype=torch.float64)
    recv = torch.stack([idx, sc.view(torch.int64), d2.view(torch.int64)], dim=1).contiguous()
    v = (recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64))
    for rep in range(3):
        print("G=%2d B=%4d: by dist %.1f us, by score %.1f us" % (G, B, t_ms(lambda: merge_on_device(*v, False)), t_ms(lambda: merge_on_device(*v, True))))
