#!/bin/bash
# SQ instruction / activity counters of one configuration's kernels (rocprofv3 --pmc passes over `bench.py --config C --sweep-only --no-graph`):
#   tools/sq_probe.sh gpurun_out/sq c1        (GPU box, repo root)
out=${1:-gpurun_out/sq}; c=${2:-c1}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out/sq_$c.txt
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_${c}_$tag -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/sq_${c}_$tag.err || { echo "$tag failed" >> $out/sq_$c.txt; break; }
  python3 tools/prof_summary.py pmc $out/sq_${c}_$tag >> $out/sq_$c.txt
  rm -rf $out/sq_${c}_$tag
done
cat $out/sq_$c.txt
