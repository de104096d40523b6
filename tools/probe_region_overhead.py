"""What a timed region of K hipGraph replays of the c1 step costs beyond K x the step: the first launch's latency and the return of the closing
synchronisation.  `python tools/probe_region_overhead.py` on the GPU box; PROBE_SPIN=1 asks HIP for spinning synchronisations
(hipSetDeviceFlags(hipDeviceScheduleSpin)) before the context exists.  Prints ms per step for K = 20 / 100 / 1000 and several ways to wait."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"))


def main():
    if os.environ.get("PROBE_SPIN"):
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))      # hipDeviceScheduleSpin
        print("hipSetDeviceFlags(spin) ->", rc)
    import torch
    import bench
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)

    Workload = bench.make_workload_class()
    w = Workload(torch, lib, bench.CONFIGS["c1"], dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            w.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        w.step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    ev = torch.cuda.Event()

    def wait_dev():
        torch.cuda.synchronize()

    def wait_stream():
        st.synchronize()

    def wait_event_spin():
        ev.record()
        while not ev.query():
            pass

    def wait_spin_then_dev():
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()

    import gc
    for name, wait in (("torch.cuda.synchronize", wait_dev), ("stream.synchronize", wait_stream), ("event.query spin", wait_event_spin),
                       ("event spin + synchronize", wait_spin_then_dev)):
        for K in (20, 100, 1000):
            res = []
            for rep in range(7):
                gc.collect()
                gc.disable()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    g.replay()
                wait()
                el = time.perf_counter() - t0
                gc.enable()
                res.append(el / K * 1e3)
            print("%-26s K=%5d  ms/step first %.5f  median %.5f  min %.5f" % (name, K, res[0], float(np.median(res)), min(res)))
    # how long the device has been idle in front of the region: K = 20 right behind `pre` warm replays and an idle gap of `gap` seconds
    gc.collect()
    gc.disable()
    for pre, gap in ((200, 0.0), (200, 0.001), (200, 0.01), (200, 0.1), (5, 0.0), (0, 0.0), (20, 0.0), (50, 0.0)):
        res = []
        for rep in range(9):
            torch.cuda.synchronize()
            time.sleep(0.1)                       # start from an idle device
            for _ in range(pre):
                g.replay()
            torch.cuda.synchronize()
            if gap:
                time.sleep(gap)
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 20 * 1e3)
        print("idle 0.1 s, %3d warm replays, synchronize, gap %.3f s, K = 20: median %.5f  min %.5f  max %.5f ms/step" % (pre, gap, float(np.median(res)), min(res), max(res)))
    gc.enable()
    # empty-queue synchronisation cost and a single replay's round trip
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("synchronize on an idle device: median %.2f us" % (np.median(ts) * 1e6))
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("one replay + synchronize: median %.2f us" % (np.median(ts) * 1e6))
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); g.replay(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print("replay() host cost: median %.2f us" % (np.median(ts) * 1e6))


if __name__ == "__main__":
    main()
