#!/bin/bash
# same-box A/B of whole bench steps over (library build, environment) pairs:  tools/ab_lib_env.sh OUT "c3 c4" "libhwer_hip_base.so:HWER_BENCH_NO_TAIL_UNDER_SELECT=1 prod:X=1"
# each arm = <library file in lib/ | prod>:<env assignments separated by commas>; two repetitions, arms interleaved
out=${1:-gpurun_out/abl}; cfgs=${2:-"c3"}; arms=${3}; mkdir -p $out
L=hybrid-weighted-embedding-recommender_amd/lib
cp $L/libhwer_hip.so $L/libhwer_hip_prod_keep.so
trap "cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so" EXIT
for rep in 1 2; do
for c in $cfgs; do
  for arm in $arms; do
    lib=${arm%%:*}; envs=${arm#*:}
    if [ "$lib" = prod ]; then cp $L/libhwer_hip_prod_keep.so $L/libhwer_hip.so; else cp $L/$lib $L/libhwer_hip.so; fi
    env $(echo $envs | tr ',' ' ') timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --sweep-only --steps ${AB_STEPS:-20} --warmup 5 --detail-out $out/x.json > $out/x.log 2>&1 || { echo "$c $arm FAILED"; tail -5 $out/x.log; continue; }
    python - <<PY
import json
r = json.load(open("$out/x.json"))
ks = {k["kernel"].split(" ")[0]: round(k["ms_per_step"], 4) for k in r.get("kernels", [])}
print("$c %-60s ms_per_step %.4f median %.4f min %.4f exhaustive %s kernels %s" % ("$arm", r["ms_per_step"], r["ms_median"], r["ms_min"], r.get("exhaustive_fallback_queries"), ks), flush=True)
PY
  done
done
done
