for v in "" HWER_SELECT_SKIP=8 HWER_SELECT_SKIP=16 HWER_SELECT_SKIP=24; do env $v HWER_BENCH_NOSYNC=1 timeout 120 python bench.py --config c2 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('[$v]', 'ms %.3f screen %.3f select %.3f' % (j['ms_per_step'], j['kernels_ms']['knn_screen'], j['kernels_ms']['knn_select_rerank']))
    elif 'Error' in l or 'assert' in l: print(l.strip()[:200])
"; done
