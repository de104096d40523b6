#!/bin/bash
# A/B of two library builds on small query batches against the 1.25 M-row shard (tools/lat_probe.py): lib/libhwer_hip.so vs lib/libhwer_hip_old.so
L=hybrid-weighted-embedding-recommender_amd/lib
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp $L/libhwer_hip.so $L/libhwer_hip_new.so
for rep in 1 2; do for which in new old; do cp $L/libhwer_hip_$which.so $L/libhwer_hip.so; for B in ${1:-1 8 64 512}; do echo "$which $(timeout -k 10 100 python3 tools/lat_probe.py $B 200 | tail -1)"; done; done; done
cp $L/libhwer_hip_new.so $L/libhwer_hip.so
