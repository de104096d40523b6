cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HWER_PROBE_CFG=${1:-c4full}
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/kk -o p --output-format csv -- python3 tools/lat_probe.py ${2:-4096} 5 > gpurun_out/kk.out 2>&1
f=$(find gpurun_out/kk -name 'p_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "hwer" in r["Name"] and int(r["Calls"]) >= 5]
for r in rows: print("%-64s n=%s avg=%.1f us" % (r["Name"].replace("void hwer::","")[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/kk.out; rm -rf gpurun_out/kk
