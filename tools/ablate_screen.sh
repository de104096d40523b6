#!/bin/bash
# Ablation battery of the float16 streaming screen (diagnostic build swapped in as lib/libhwer_hip.so: HWER_LIB_NAME=libhwer_hip_diag.so HIPCC_EXTRA=-DHWER_DIAG_STAMPS python csrc/build.py):
# rocprofv3 kernel durations of one c4-shaped search (4 096 queries x 1.25 M rows) per HWER_SCREEN_NOFILTER bit mask.
# usage (GPU box, repo root): tools/ablate_screen.sh gpurun_out/abl "0 1 5 13 29"
out=${1:-gpurun_out/abl}; masks=${2:-"0 1"}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in $masks; do
  export HWER_SCREEN_NOFILTER=$m
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/m$m -o p --output-format csv -- python3 tools/lat_probe.py ${HWER_PROBE_B:-4096} 10 > $out/m$m.out 2> $out/m$m.err || { echo "mask $m failed" >> $out/summary.txt; break; }
  f=$(find $out/m$m -name 'p_kernel_stats.csv' | head -1)
  [ -n "$f" ] || { echo "mask $m: no stats file" >> $out/summary.txt; break; }
  python3 - "$f" "$m">> $out/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "screen" in r["Name"] and "kernel" in r["Name"]]
print("mask %s: %s | %s" % (sys.argv[2], " | ".join("%s calls=%s avg_us=%.1f min_us=%.1f" % (r["Name"][12:48], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3) for r in rows), open(sys.argv[1].replace("p_kernel_stats.csv", "") + "../m%s.out" % sys.argv[2]).read().strip()))
PY
  rm -rf $out/m$m
done
cat $out/summary.txt
