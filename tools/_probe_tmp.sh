set -e
python tools/probe_f64rows_latency.py
python tools/probe_f64rows_batch.py 943 1682 64
python tools/probe_f64rows_batch.py 4096 26744 128
HWER_F64ROWS_CANDIDATES=1 python tools/probe_f64rows_batch.py 943 1682 64
HWER_F64ROWS_CANDIDATES=1 python tools/probe_f64rows_batch.py 4096 26744 128
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_f64b -- python3 $GRAFT_REPO_ROOT/tools/probe_f64rows_batch.py 4096 26744 128 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_f64b -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {} | cut -c1-200'
