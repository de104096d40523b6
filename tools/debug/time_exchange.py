"""Scratch: single-GPU cost of everything in the N > 1 exchange except the collective itself (pack, in-place merge)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")]
import torch
from hwer.distributed import merge_on_device
dev = torch.device("cuda")
U, K = int(os.environ.get("U", 6040)), 200
for G in (2, 4, 8):
    Bs = (U + G - 1) // G
    idx = torch.randint(0, 1 << 20, (U, K), device=dev)
    d2 = torch.rand((U, K), device=dev, dtype=torch.float64).sort(dim=1).values
    sc = 1.0 - d2 / 4
    def pack():
        from hwer.distributed import pack_send_buffer
        return pack_send_buffer(idx, sc, d2, G)
    def pack_old():
        send = torch.empty((G, 3, Bs, K), dtype=torch.int64, device=dev)
        full = U // Bs; rem = U - full * Bs
        for c, t in enumerate((idx, sc.view(torch.int64), d2.view(torch.int64))):
            send[:full, c].copy_(t[:full * Bs].view(full, Bs, K))
            if rem: send[full, c, :rem].copy_(t[full * Bs:])
        if full < G:
            for c, v in enumerate((-1, 0, 0x7FF0000000000000)):
                send[full, c, rem:].fill_(v)
        return send
    recv = pack()
    def merge():
        return merge_on_device(recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64), True)
    for name, fn in (("pack", pack), ("merge", merge)):
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): fn()
        b.record(); torch.cuda.synchronize()
        print("G=%d U=%d %s: %.1f us" % (G, U, name, a.elapsed_time(b) / 20 * 1e3))
