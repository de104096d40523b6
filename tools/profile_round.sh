#!/bin/bash
# Per-round profiles: rocprofv3 kernel stats + timeline + FETCH_SIZE / WRITE_SIZE passes of `bench.py --config <c> --no-cpu-baseline
# --sweep-only --no-graph` for the configurations of the bench line (c1, c2, c3, c4), SQ counters of the c3 / c4 screen.
# usage (on the GPU box, from the repo root):  tools/profile_round.sh gpurun_out/prof "c1 c2 c3 c4"
set -u
out=${1:-gpurun_out/prof}; cfgs=${2:-"c1 c2 c3 c4"}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -c "import sys; sys.path.insert(0, 'hybrid-weighted-embedding-recommender_amd/csrc'); import build; print(build._digest())" > $out/library_digest.txt
for c in $cfgs; do
  steps=20; [ $c = c3 ] && steps=10; [ $c = c4 ] && steps=10; [ $c = c4full ] && steps=5
  echo "rocprof stats $c" >> $out/progress.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats_$c -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps $steps --warmup 5 --detail-out $out/stats_${c}_bench.json > /dev/null 2> $out/stats_$c.err
  python3 tools/prof_summary.py stats $out/stats_$c > $out/stats_$c.txt
  python3 tools/trace_timeline.py $out/stats_$c > $out/timeline_$c.txt 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "rocprof pmc $c $ctr" >> $out/progress.log
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $out/pmc_${c}_$ctr -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/pmc_${c}_$ctr.err
    python3 tools/prof_summary.py pmc $out/pmc_${c}_$ctr >> $out/pmc_$c.txt
  done
  rm -rf $out/stats_$c/*kernel_trace.csv $out/pmc_${c}_*/   # large raw files
done
for c in c3 c4 c4full; do
  case " $cfgs " in *" $c "*) ;; *) continue;; esac
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    echo "rocprof sq $c $tag" >> $out/progress.log
    timeout -k 10 200 rocprofv3 --pmc $grp -d $out/sq_${c}_$tag -o $c --output-format csv -- python3 bench.py --config $c --no-cpu-baseline --sweep-only --no-graph --steps 3 --warmup 1 > /dev/null 2> $out/sq_${c}_$tag.err
    python3 tools/prof_summary.py pmc $out/sq_${c}_$tag >> $out/sq_$c.txt
    rm -rf $out/sq_${c}_$tag
  done
done
echo done >> $out/progress.log
