#!/usr/bin/env python3
"""Sweep a diagnostic-build env switch over the prologue probe (same box): python tools/sweep_prologue.py "c4 c5 c4full" HWER_ROW_BLOCKS "512 768 1024 2048"
(swaps lib/libhwer_hip_diag.so in; children are ordinary subprocesses)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")
cfgs, var, vals = sys.argv[1], sys.argv[2], sys.argv[3].split()
keep = os.path.join(L, "libhwer_hip_keep.so")
shutil.copy(os.path.join(L, "libhwer_hip.so"), keep)
shutil.copy(os.path.join(L, "libhwer_hip_diag.so"), os.path.join(L, "libhwer_hip.so"))
try:
    for rep in range(2):
        for v in vals:
            env = dict(os.environ)
            if v != "-":
                env[var] = v
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prologue_probe.py"), cfgs], env=env, capture_output=True, text=True, timeout=600)
            print("%s=%s rep%d: %s" % (var, v, rep, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1]), flush=True)
finally:
    shutil.copy(keep, os.path.join(L, "libhwer_hip.so"))
