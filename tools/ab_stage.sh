#!/bin/bash
# same-box A/B of alternative library builds on ONE stage of the bench step (bench_detail's stages_ms): tools/ab_stage.sh OUT "c4 c5" "libX.so" blend_prepare_search
out=${1:-gpurun_out/abs}; cfgs=${2:-"c4"}; libs=${3}; stage=${4:-blend_prepare_search}; mkdir -p $out
L=hybrid-weighted-embedding-recommender_amd/lib
cp $L/libhwer_hip.so $L/libhwer_hip_prod_keep.so
trap "cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so" EXIT
for rep in 1 2; do
for c in $cfgs; do
  for lib in prod $libs; do
    if [ $lib = prod ]; then cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so; else cp $L/$lib $L/libhwer_hip.so; fi
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 5 --detail-out $out/x.json > $out/x.log 2>&1 || { echo "$c $lib FAILED"; tail -5 $out/x.log; continue; }
    python - <<PY
import json
r = json.load(open("$out/x.json"))
print("$c %-26s ms_median %.4f  $stage %.4f ms" % ("$lib", r["ms_median"], r["stages_ms"].get("$stage", 0)))
PY
  done
done
done
