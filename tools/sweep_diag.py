#!/usr/bin/env python3
"""Same-box sweep of diagnostic-build switches: one `tools/lat_probe.py` child per (config, env set) with the DIAGNOSTIC library
(lib/libhwer_hip_diag.so, built with HIPCC_EXTRA=-DHWER_DIAG_ABLATE: it honours the HWER_* tuning overrides) swapped in.
    python tools/sweep_diag.py OUT.txt "c4 c5" B "HWER_SEED_C=4" "HWER_SEED_C=8" ...
The parent never touches the GPU (children are ordinary subprocesses)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")


def main():
    out, cfgs, B = sys.argv[1], sys.argv[2].split(), sys.argv[3]
    sets = sys.argv[4:]
    prod, diag, keep = os.path.join(L, "libhwer_hip.so"), os.path.join(L, "libhwer_hip_diag.so"), os.path.join(L, "libhwer_hip_prod.so")
    shutil.copy(prod, keep)
    shutil.copy(diag, prod)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "a") as fh:
            for rep in range(2):
                for c in cfgs:
                    for e in sets:
                        env = dict(os.environ, HWER_PROBE_CFG=c)
                        for kv in e.split():
                            k, v = kv.split("=", 1)
                            env[k] = v
                        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lat_probe.py"), B, "20"], env=env, capture_output=True, text=True, timeout=300)
                        line = "%s rep%d [%s]: %s" % (c, rep, e, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1])
                        stamps = [l for l in r.stderr.splitlines() if l.startswith("[stamps]") or l.startswith("[select stamps]")]
                        if stamps:      # (a launch's stamps are printed by the NEXT launch: the last line is the steady state)
                            line += "\n    " + stamps[-1]
                        print(line, flush=True)
                        fh.write(line + "\n")
                        fh.flush()
    finally:
        shutil.copy(keep, prod)


if __name__ == "__main__":
    main()
