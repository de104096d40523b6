#!/bin/bash
# same-box A/B of the c3 step's scheduling (round 6): one search after the other (the default) against the short last search on a second stream
# (HWER_BENCH_TAIL_STREAM=1 / HWER_TAIL_STREAM=1) and the pipelined search (HWER_BENCH_PIPELINE=1);  tools/ab_c3_pipeline.sh [steps]
steps=${1:-100}
for rep in 1 2; do
for arm in "c3:X=1" "c3:HWER_BENCH_TAIL_STREAM=1" "c3:HWER_BENCH_PIPELINE=1" "c3_f64rows:X=1" "c3_f64rows:HWER_TAIL_STREAM=1"; do
  c=${arm%%:*}; e=${arm#*:}
  env $e timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --sweep-only --steps $steps --warmup 5 --detail-out gpurun_out/x_pipe.json > gpurun_out/x_pipe.log 2>&1 || { echo "$arm FAILED"; tail -3 gpurun_out/x_pipe.log; continue; }
  python - <<PY
import json
r = json.load(open("gpurun_out/x_pipe.json"))
print("%-11s %-28s ms_per_step %.4f median %.4f min %.4f max %.4f" % ("$c", "$e", r["ms_per_step"], r["ms_median"], r["ms_min"], r["ms_max"]), flush=True)
PY
done
done
