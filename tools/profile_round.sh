#!/bin/bash
# Round profile: bench sweep (all configs, with CPU baselines), rocprofv3 kernel stats and PMC passes (c2, c3), micro-benchmarks.
# usage (on the GPU box, from the repo root):  tools/profile_round.sh gpurun_out/final
set -u
out=${1:-gpurun_out/final}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${BENCH_CONFIGS-c2 c1 c3 c4 c5}; do    # BENCH_CONFIGS="" skips the bench sweep (profiler passes only)
  echo "bench $c" >> $out/progress.log
  timeout -k 10 400 python3 bench.py --config $c > $out/bench_$c.json 2> $out/bench_$c.err || echo "bench $c failed" >> $out/progress.log
done
for c in c2 c3; do
  echo "rocprof stats $c" >> $out/progress.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_$c -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only > $out/stats_${c}_bench.json 2> $out/stats_$c.err
  python3 tools/prof_summary.py stats $out/stats_$c > $out/stats_$c.txt
  python3 tools/trace_timeline.py $out/stats_$c > $out/timeline_$c.txt 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "rocprof pmc $c $ctr" >> $out/progress.log
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $out/pmc_${c}_$ctr -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --steps 3 --warmup 1 > /dev/null 2> $out/pmc_${c}_$ctr.err
    python3 tools/prof_summary.py pmc $out/pmc_${c}_$ctr >> $out/pmc_$c.txt
  done
  rm -rf $out/stats_$c/*kernel_trace.csv $out/pmc_${c}_*/   # large raw files
done
if [ -z "${SKIP_MICRO-}" ]; then
  echo "micro" >> $out/progress.log
  timeout -k 10 300 python3 tools/bench_micro.py --what rowops,latency,fit > $out/micro.jsonl 2> $out/micro.err
fi
echo done >> $out/progress.log
