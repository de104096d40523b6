#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/kernel_resources.py knn_select.hip [name filter] [extra hipcc flags ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "csrc")
src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
       "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", "/tmp/_kr.o"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = m.group(2)
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1)] = m.group(2)
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem:
        continue
    print("%-90s vgpr %4s agpr %3s  vspill %3s sspill %3s scratch %4s occ %s lds %s" % (dem[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
          r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
