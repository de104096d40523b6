"""Scratch: duration of hwer_knn_exact_device when one query of a large catalogue is uncertified (forced by duplicates)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")]
import torch
from hwer import native
lib = native.lib()
dev = torch.device("cuda")
for N, d in ((1250000, 128), (1250000, 256), (26744, 128)):
    K, B = 200, 4
    g = torch.Generator(device=dev).manual_seed(1)
    Xs = torch.randn((N, d), generator=g, device=dev); Xs /= Xs.norm(dim=1, keepdim=True)
    Xs[1000:1700] = Xs[5]                       # 700 copies: the query on row 5 cannot be certified
    ldx = lib.hwer_knn_padded_dim(d); Ip = lib.hwer_knn_padded_rows(N)
    X = torch.zeros((Ip, ldx), device=dev); hn = torch.empty(Ip, device=dev)
    P = native.ptr; S = native.stream_ptr
    rows = torch.arange(N, dtype=torch.int64, device=dev)
    native.check(lib.hwer_pack_rows_f32(P(Xs), N, d, d, P(rows), N, P(X), ldx, P(hn), S()), "pack")
    Vq = Xs[[5, 6, 7, 8]].contiguous()
    anchors = torch.arange(B, dtype=torch.int32, device=dev)
    Q64 = torch.empty((B, d), dtype=torch.float64, device=dev); Q32 = torch.empty((B, ldx), device=dev)
    native.check(lib.hwer_query_mix(P(Vq), B, d, d, P(anchors), None, None, None, None, B, 0, P(Q64), P(Q32), ldx, S()), "mix")
    oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float64, device=dev)
    od = torch.empty((B, K), dtype=torch.float64, device=dev); of = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.hwer_knn_workspace_bytes(B, N, d, K), dtype=torch.uint8, device=dev)
    wse = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(N, K), dtype=torch.uint8, device=dev)
    def search():
        native.check(lib.hwer_knn_search_f32(P(X), P(hn), N, d, ldx, P(Q32), P(Q64), P(Vq), d, P(anchors), B, K, 0, 1, None, 0, 0.51,
                                             P(oi), P(os_), P(od), P(of), P(ws), ws.numel(), S()), "search")
    def exact():
        native.check(lib.hwer_knn_exact_device(P(X), N, d, ldx, P(Q64), P(Vq), d, P(anchors), K, 0, 1, None, 0, P(oi), P(os_), P(od), P(of), P(ws), None,
                                               P(wse), wse.numel(), S()), "exact")
    search(); exact(); torch.cuda.synchronize()
    t = []
    for _ in range(5):
        search(); torch.cuda.synchronize()
        n_bad = int((of == 0).sum())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); exact(); b.record(); torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    print("N=%d d=%d: %d uncertified, exact_device %.3f ms (min %.3f)" % (N, d, n_bad, sum(t) / len(t), min(t)))
