#!/bin/bash
# same-box A/B of alternative builds of the library on whole bench steps: tools/ab_libs2.sh OUT "c4 c4full" "libhwer_hip_notok.so ..."
# (each alternative is swapped in as lib/libhwer_hip.so for its runs; the production library is restored at the end)
out=${1:-gpurun_out/abl}; cfgs=${2:-"c4 c4full"}; libs=${3}; mkdir -p $out
L=hybrid-weighted-embedding-recommender_amd/lib
cp $L/libhwer_hip.so $L/libhwer_hip_prod_keep.so
trap "cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so" EXIT
for rep in 1 2; do
for c in $cfgs; do
  for lib in prod $libs; do
    if [ $lib = prod ]; then cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so; else cp $L/$lib $L/libhwer_hip.so; fi
    HWER_BENCH_OPTS=${AB_OPTS:-0} timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --sweep-only --steps 20 --warmup 5 --detail-out $out/x.json > $out/x.log 2>&1 || { echo "$c $lib FAILED"; tail -5 $out/x.log; continue; }
    python - <<PY
import json
r = json.load(open("$out/x.json"))
ks = {k["kernel"][:10]: round(k["ms_per_step"], 4) for k in r["kernels"]}
print("$c %-26s ms_median %.4f  ms_min %.4f  screen %.4f  select %.4f" % ("$lib", r["ms_median"], r["ms_min"], ks.get("knn_screen", 0), ks.get("knn_select", 0)))
PY
  done
done
done
