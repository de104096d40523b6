#!/bin/bash
# usage: tools/sweep_env.sh OUTFILE CONFIG "ENV1=a ENV2=b" "ENV1=c" ...   one bench run per env set; appends the condensed line
out=$1; cfg=$2; shift 2
for e in "$@"; do
  echo "== $e" >> $out
  env $e timeout -k 10 120 python bench.py --config $cfg --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    print(json.dumps({'ms_per_step':round(j['ms_per_step'],4),'kernels_ms':j['kernels_ms'],'frac':round(j['roofline']['frac'],3)}))
" >> $out || { echo "FAILED/timeout" >> $out; exit 1; }
done
