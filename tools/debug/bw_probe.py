import torch, time
x = torch.randn(1250016, 128, device="cuda")
for fn, name in ((lambda: x.sum(), "sum"), (lambda: x.max(), "max"), (lambda: torch.mv(x, x[0]), "mv")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(name, "%.3f ms  %.2f TB/s" % (ms, x.numel() * 4 / ms / 1e9))
