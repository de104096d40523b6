#!/bin/bash
# Which block kind of the fused prologue (blend_prepare_search_kernel) sets its duration: rocprofv3 kernel durations of bench steps with one
# kind of block switched off (diagnostic build swapped in: HWER_LIB_NAME=libhwer_hip_diag.so HIPCC_EXTRA=-DHWER_DIAG_STAMPS python csrc/build.py).
# usage (GPU box, repo root): tools/ablate_prologue.sh gpurun_out/ablp "0 1 2 4 6 8" c1
out=${1:-gpurun_out/ablp}; masks=${2:-"0 1"}; cfg=${3:-c1}; mkdir -p $out
L=hybrid-weighted-embedding-recommender_amd/lib
cp $L/libhwer_hip.so $L/libhwer_hip_keep.so && cp $L/libhwer_hip_diag.so $L/libhwer_hip.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in $masks; do
  export HWER_PROLOGUE_ABLATE=$m
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/m$m -o p --output-format csv -- python3 bench.py --config $cfg --no-cpu-baseline --sweep-only --no-graph --steps 40 --warmup 5 > $out/m$m.out 2> $out/m$m.err || echo "mask $m: bench exit $? (its oracle check fails when blocks are switched off)" >> $out/summary.txt
  f=$(find $out/m$m -name 'p_kernel_trace.csv' | head -1)
  [ -n "$f" ] || { echo "mask $m: no stats file" >> $out/summary.txt; break; }
  python3 - "$f" "$m" >> $out/summary.txt <<'PY'
import csv, sys, statistics
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(sys.argv[1])) if "blend_prepare_search" in r["Kernel_Name"]][4:]
print("mask %s: calls=%d median_us=%.2f min_us=%.2f p90_us=%.2f" % (sys.argv[2], len(d), statistics.median(d), min(d), sorted(d)[int(0.9 * len(d))]))
PY
  rm -rf $out/m$m
done
cp $L/libhwer_hip_keep.so $L/libhwer_hip.so
cat $out/summary.txt
