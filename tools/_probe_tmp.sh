set -e
python bench.py --config c3_f64rows --steps 10 --warmup 3 --detail-out gpurun_out/r6_c3f64_detail.json > gpurun_out/r6_c3f64.json 2> gpurun_out/r6_c3f64.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c3f64 -o c3f64 --output-format csv -- python3 bench.py --config c3_f64rows --steps 5 --warmup 2 > /dev/null 2> gpurun_out/prof_c3f64.err
python3 tools/prof_summary.py stats gpurun_out/prof_c3f64 > gpurun_out/r6_stats_c3f64.txt
rm -f gpurun_out/prof_c3f64/*/*kernel_trace.csv
