cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1 64; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/lat$B -o p --output-format csv -- python3 tools/lat_probe.py $B 50 > gpurun_out/lat$B.out 2>&1
f=$(find gpurun_out/lat$B -name 'p_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) >= 40]
for r in rows: print("%-70s n=%s avg=%.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat gpurun_out/lat$B.out | tail -1
rm -rf gpurun_out/lat$B
done
