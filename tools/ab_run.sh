#!/bin/bash
# one A/B round on the GPU box: parity tests with the new library, then tools/ab_libs.sh; prints a compact table
out=$1; cfgs=${2:-"c4 c5 c3"}
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
tools/ab_libs.sh $out "$cfgs" > /dev/null
python3 - $out/summary.txt <<'PY'
import re, sys
for line in open(sys.argv[1]):
    tag = line.split(":")[0]
    ks = re.findall(r"(screen_kernel<[^>]*>|select_rerank_kernel<0>|reduce_lists_wave_kernel|small_threshold[a-z_]*kernel<32>)\S* n=(\d+) avg=([\d.]+) min=([\d.]+)", line)
    print(tag, " ".join("%s:%s/%s" % (k.replace("screen_kernel", "scr").replace(" ", ""), a, m) for k, n, a, m in ks), line.strip().split("|")[-1])
PY
