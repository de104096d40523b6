#!/usr/bin/env python3
"""In-kernel phase stamps of the select kernels (diagnostic library swapped in): python tools/select_stamps.py "c1 c2" """
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")
keep = os.path.join(L, "libhwer_hip_keep.so")
shutil.copy(os.path.join(L, "libhwer_hip.so"), keep)
shutil.copy(os.path.join(L, "libhwer_hip_diag.so"), os.path.join(L, "libhwer_hip.so"))
try:
    for c in sys.argv[1].split():
        env = dict(os.environ, HWER_SELECT_STAMPS="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--no-cpu-baseline", "--no-graph", "--sweep-only", "--steps", "6", "--warmup", "2",
                            "--detail-out", os.path.join(ROOT, "gpurun_out", "stamps_%s.json" % c)], env=env, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stderr.splitlines() if l.startswith("[select stamps]")]
        print(c, "\n".join(lines[-6:]) if lines else "no stamps: " + r.stderr[-300:], flush=True)
finally:
    shutil.copy(keep, os.path.join(L, "libhwer_hip.so"))
