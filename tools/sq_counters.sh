#!/bin/bash
# SQ counter groups (separate --pmc passes) of `bench.py --config <c>` for any configuration: tools/sq_counters.sh gpurun_out/sq "c1 c2"
set -u
out=${1:-gpurun_out/sq}; cfgs=${2:-"c1 c2"}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in $cfgs; do
  rm -f $out/sq_$c.txt
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    echo "rocprof sq $c $tag" >> $out/progress.log
    timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_${c}_$tag -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/sq_${c}_$tag.err
    python3 tools/prof_summary.py pmc $out/sq_${c}_$tag >> $out/sq_$c.txt
    rm -rf $out/sq_${c}_$tag
  done
done
echo done >> $out/progress.log
