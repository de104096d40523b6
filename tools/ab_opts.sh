#!/bin/bash
# same-box A/B of a library option on whole bench steps:  tools/ab_opts.sh OUT "c4 c5" 64   (64 = HWER_KNN_NO_PINGPONG)
out=${1:-gpurun_out/ab}; cfgs=${2:-"c4 c5 c3 c4full"}; opt=${3:-64}; mkdir -p $out
for c in $cfgs; do
  for o in 0 $opt 0 $opt; do
    HWER_BENCH_OPTS=$o timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --sweep-only --steps 20 --warmup 5 --detail-out $out/${c}_$o.json > $out/${c}_$o.log 2>&1 || { echo "$c opts=$o FAILED"; tail -5 $out/${c}_$o.log; exit 1; }
    python - <<PY
import json
r = json.load(open("$out/${c}_$o.json"))
ks = {k["kernel"][:10]: round(k["ms_per_step"], 4) for k in r["kernels"]}
print("$c opts=$o  ms_median %.4f  ms_min %.4f  screen %.4f  select %.4f  frac %.3f" % (r["ms_median"], r["ms_min"], ks.get("knn_screen", 0), ks.get("knn_select", 0), [k["frac"] for k in r["kernels"] if k["kernel"].startswith("knn_screen")][0]))
PY
  done
done
