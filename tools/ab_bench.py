#!/usr/bin/env python3
"""Same-box A/B of several builds of the library on whole bench steps: lib/libhwer_hip_<name>.so for every name given, swapped in turn over
lib/libhwer_hip.so (restored afterwards); one `bench.py --config C --no-cpu-baseline --sweep-only` child per (rep, config, build).
    python tools/ab_bench.py OUT.txt "c1 c2" name1 name2 ...      (AB_REPS, AB_STEPS)
The parent never touches the GPU."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")


def main():
    out, cfgs, names = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    prod, keep = os.path.join(L, "libhwer_hip.so"), os.path.join(L, "libhwer_hip_keep.so")
    shutil.copy(prod, keep)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "a") as fh:
            for rep in range(int(os.environ.get("AB_REPS", "2"))):
                for c in cfgs:
                    for n in names:
                        shutil.copy(os.path.join(L, "libhwer_hip_%s.so" % n), prod)
                        detail = os.path.join(ROOT, "gpurun_out", "ab_detail.json")
                        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--no-cpu-baseline", "--sweep-only", "--steps", os.environ.get("AB_STEPS", "40"),
                                            "--warmup", "10", "--detail-out", detail], capture_output=True, text=True, timeout=600)
                        try:
                            d = json.loads(r.stdout.strip().splitlines()[-1])
                            full = json.load(open(detail))
                            st = full.get("stages_ms") or full.get("kernels") or {}
                            line = "%s rep%d [%s]: ms_per_step %.5f median %.5f min %.5f stages %s" % (c, rep, n, d["ms_per_step"], d.get("ms_median", 0), d.get("ms_min", 0),
                                                                                                 json.dumps({k[:14]: round(v, 4) for k, v in st.items()} if isinstance(st, dict) else st))
                        except Exception as e:  # noqa: BLE001
                            line = "%s rep%d [%s]: FAILED %r %s" % (c, rep, n, e, r.stderr[-300:])
                        print(line, flush=True)
                        fh.write(line + "\n")
                        fh.flush()
    finally:
        shutil.copy(keep, prod)


if __name__ == "__main__":
    main()
