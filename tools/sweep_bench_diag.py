#!/usr/bin/env python3
"""Same-box sweep of diagnostic-build switches on whole bench steps: one `bench.py --config C --no-cpu-baseline --sweep-only` child per
(rep, config, env set) with the DIAGNOSTIC library (lib/libhwer_hip_diag.so) swapped in.
    python tools/sweep_bench_diag.py OUT.txt "c1 c2" "HWER_SELECT_SKIP=0" "HWER_SELECT_SKIP=4" ..."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")


def main():
    out, cfgs, sets = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    prod, diag, keep = os.path.join(L, "libhwer_hip.so"), os.path.join(L, "libhwer_hip_diag.so"), os.path.join(L, "libhwer_hip_keep.so")
    shutil.copy(prod, keep)
    shutil.copy(diag, prod)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "a") as fh:
            for rep in range(int(os.environ.get("AB_REPS", "2"))):
                for c in cfgs:
                    for e in sets:
                        env = dict(os.environ)
                        for kv in e.split():
                            k, v = kv.split("=", 1)
                            env[k] = v
                        detail = os.path.join(ROOT, "gpurun_out", "ab_detail.json")
                        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--no-cpu-baseline", "--sweep-only", "--steps", os.environ.get("AB_STEPS", "40"),
                                            "--warmup", "10", "--detail-out", detail], env=env, capture_output=True, text=True, timeout=600)
                        try:
                            d = json.loads(r.stdout.strip().splitlines()[-1])
                            st = json.load(open(detail)).get("stages_ms") or {}
                            line = "%s rep%d [%s]: ms_per_step %.5f median %.5f search %.4f parity %s" % (c, rep, e, d["ms_per_step"], d.get("ms_median", 0), st.get("knn_search", 0),
                                                                                                        d.get("parity_checked_queries"))
                        except Exception as ex:  # noqa: BLE001
                            line = "%s rep%d [%s]: FAILED %r %s" % (c, rep, e, ex, r.stderr[-400:])
                        print(line, flush=True)
                        fh.write(line + "\n")
                        fh.flush()
    finally:
        shutil.copy(keep, prod)


if __name__ == "__main__":
    main()
