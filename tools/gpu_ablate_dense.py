import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")]
import torch
from hwer import native
lib = native.lib()
B, I, d, K = int(os.environ.get("B", 6040)), int(os.environ.get("I", 3706)), 128, 200
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(1)
Ip = lib.hwer_knn_padded_rows(I)
X = torch.zeros((Ip, d), device=dev); X[:I] = torch.randn((I, d), generator=g, device=dev); X[:I] /= X[:I].norm(dim=1, keepdim=True)
Q = torch.randn((B, d), generator=g, device=dev); Q /= Q.norm(dim=1, keepdim=True)
hn = 0.5 * (X * X).sum(1); hn[I:] = float('inf')
Q64 = Q.double()
oi = torch.empty((B, K), dtype=torch.int64, device=dev); os_ = torch.empty((B, K), dtype=torch.float64, device=dev)
od = torch.empty((B, K), dtype=torch.float64, device=dev); of = torch.empty(B, dtype=torch.int32, device=dev)
ws = torch.empty(lib.hwer_knn_workspace_bytes(B, I, d, K), dtype=torch.uint8, device=dev)
P = native.ptr
def run():
    native.check(lib.hwer_knn_search_f32(P(X), P(hn), I, d, d, P(Q), P(Q64), P(Q), d, None, B, K, 0, 1, None, 0, 0.51,
                                         P(oi), P(os_), P(od), P(of), P(ws), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
run(); torch.cuda.synchronize()
lib.hwer_profile_enable(1)
for _ in range(10): run()
torch.cuda.synchronize()
tot = ctypes.c_double(); cnt = ctypes.c_int64()
lib.hwer_profile_read(0, ctypes.byref(tot), ctypes.byref(cnt)); sc = tot.value / cnt.value
lib.hwer_profile_read(1, ctypes.byref(tot), ctypes.byref(cnt)); se = tot.value / cnt.value
print("skip=%s B=%d I=%d: screen %.3f ms, select %.3f ms" % (os.environ.get("HWER_SELECT_SKIP", "0"), B, I, sc, se))
