#!/bin/bash
# Round-end evidence runs on ONE GPU box (tools/final_runs.sh OUT): the default bench twice + the driver's command, the 8-shard rehearsals, the row-kernel / latency
# micro-benchmarks, the N = 2 / N = 4 jobs over gloo on the one GPU and the job's own code over RCCL at world size 1.  Every step writes into OUT (progress in OUT/progress.log).
out=${1:-gpurun_out/final}; mkdir -p $out
say() { echo "$(date +%T) $*" >> $out/progress.log; }
say "default bench 1"; python bench.py > $out/bench_n1_final.json 2> $out/bench_n1_final.err && cp gpurun_out/bench_detail.json $out/bench_n1_final_detail.json
say "default bench 2"; python bench.py > $out/bench_n1_run2.json 2> $out/bench_n1_run2.err
say "driver command"; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_cmd.json 2> $out/bench_n1_driver_cmd.err
say "emulate 8 c4full"; python bench.py --emulate-shards 8 > $out/emulate8_final.json 2> $out/emulate8_final.err && cp gpurun_out/bench_detail.json $out/emulate8_final_detail.json
say "emulate 8 c5full"; python bench.py --emulate-shards 8 --config c5full > $out/emulate8_c5full.json 2> $out/emulate8_c5full.err && cp gpurun_out/bench_detail.json $out/emulate8_c5full_detail.json
say "bench_micro"; python tools/bench_micro.py --what rowops,latency > $out/bench_micro.txt 2> $out/bench_micro.err
say "pair probe"; python tools/pair_probe.py > $out/pair_probe.txt 2>&1
say "gloo n2"; HWER_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 > $out/bench_n2_gloo_rehearsal_one_gpu.json 2> $out/bench_n2_gloo.err
say "gloo n4"; HWER_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 3 --warmup 1 > $out/bench_n4_gloo_rehearsal_one_gpu.json 2> $out/bench_n4_gloo.err
say "rccl world1"; HWER_BENCH_SHARDED_WORLD1=1 python bench.py --gpus 1 --steps 3 --warmup 1 > $out/bench_rccl_world1_full_job.json 2> $out/bench_rccl_world1.err
say "f64 probes"; python tools/probe_f64rows_latency.py > $out/probe_f64rows.txt 2>&1; python tools/probe_f64rows_batch.py 943 1682 64 >> $out/probe_f64rows.txt 2>&1; python tools/probe_f64rows_batch.py 4096 26744 128 >> $out/probe_f64rows.txt 2>&1
say done
