#!/bin/bash
# A/B of two builds of the library on the same GPU box: lib/libhwer_hip.so (new) vs lib/libhwer_hip_old.so, per-kernel durations of
# tools/lat_probe.py under rocprofv3.  usage: tools/ab_libs.sh gpurun_out/ab "c4 c5"
out=${1:-gpurun_out/ab}; cfgs=${2:-"c4"}; mkdir -p $out
L=hybrid-weighted-embedding-recommender_amd/lib
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp $L/libhwer_hip.so $L/libhwer_hip_new.so
for rep in 1 2; do
for which in new old; do
  cp $L/libhwer_hip_$which.so $L/libhwer_hip.so
  for c in $cfgs; do
    export HWER_PROBE_CFG=$c
    timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $out/$which$c -o p --output-format csv -- python3 tools/lat_probe.py 4096 10 > $out/$which$c.out 2> $out/$which$c.err || { echo "$which $c failed" >> $out/summary.txt; exit 1; }
    f=$(find $out/$which$c -name 'p_kernel_stats.csv' | head -1)
    [ -n "$f" ] || { echo "$which $c: no stats" >> $out/summary.txt; exit 1; }
    python3 - "$f" "$which $c" "$out/$which$c.out" >> $out/summary.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hwer" in r["Name"] and int(r["Calls"]) > 1 or "screen_kernel" in r["Name"]]
print("%s: %s | %s" % (sys.argv[2], " | ".join("%s n=%s avg=%.1f min=%.1f" % (r["Name"].replace("void hwer::", "").replace("hwer::", "")[:30], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3) for r in rows[:7]), open(sys.argv[3]).read().strip()))
PY
    rm -rf $out/$which$c
  done
done
done
cp $L/libhwer_hip_new.so $L/libhwer_hip.so
cat $out/summary.txt
