#!/bin/bash
# Texture-addresser / vector-L1 counters of one configuration's kernels (is the gather path the limiter of the select's re-rank rounds?):
#   tools/ta_probe.sh gpurun_out/ta c3        (GPU box, repo root)
out=${1:-gpurun_out/ta}; c=${2:-c3}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out/ta_$c.txt
for grp in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_TCC_NC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/ta_${c}_$tag -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/ta_${c}_$tag.err || { echo "$tag failed" >> $out/ta_$c.txt; tail -3 $out/ta_${c}_$tag.err >> $out/ta_$c.txt; continue; }
  python3 tools/prof_summary.py pmc $out/ta_${c}_$tag >> $out/ta_$c.txt
  rm -rf $out/ta_${c}_$tag
done
cat $out/ta_$c.txt
