#!/bin/bash
# SQ counters of the c2 sweep (one pass per counter group), condensed per kernel: what bounds select_dense / the screen.
# usage (GPU box, repo root): tools/pmc_select.sh gpurun_out/pmcsel [config]
out=${1:-gpurun_out/pmcsel}; cfg=${2:-c2}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/$tag -o c2 --output-format csv -- python3 bench.py --config $cfg --no-cpu-baseline --sweep-only --steps 3 --warmup 1 > /dev/null 2> $out/$tag.err
  python3 tools/prof_summary.py pmc $out/$tag >> $out/summary.txt
  rm -rf $out/$tag
done
grep -E "select_|screen" $out/summary.txt
