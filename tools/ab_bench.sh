# same-box A/B of two library builds through bench.py stage times: lib/libhwer_hip.so vs lib/libhwer_hip_old.so (GPU box, repo root)
L=hybrid-weighted-embedding-recommender_amd/lib
cp $L/libhwer_hip.so $L/libhwer_hip_new.so
for rep in 1 2; do for which in new old; do cp $L/libhwer_hip_$which.so $L/libhwer_hip.so
for c in ${AB_CFGS:-c4full c5}; do python bench.py --config $c --no-cpu-baseline --sweep-only --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which $c', round(j['ms_per_step'],3), {k['kernel'][:14]: round(k['ms_per_step'],3) for k in j['kernels']})"
done; done; done
cp $L/libhwer_hip_new.so $L/libhwer_hip.so
