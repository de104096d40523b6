#!/usr/bin/env python3
"""Same-box A/B of several builds of the library: lib/libhwer_hip_<name>.so for every name given, swapped in turn over
lib/libhwer_hip.so (restored afterwards); one `tools/lat_probe.py` child per (rep, config, build) - ms per search.
    python tools/ab_multi.py OUT.txt "c4 c5 c4full" B name1 name2 ...
The parent never touches the GPU."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "lib")


def main():
    out, cfgs, B, names = sys.argv[1], sys.argv[2].split(), sys.argv[3], sys.argv[4:]
    prod, keep = os.path.join(L, "libhwer_hip.so"), os.path.join(L, "libhwer_hip_keep.so")
    shutil.copy(prod, keep)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        with open(out, "a") as fh:
            for rep in range(int(os.environ.get("AB_REPS", "3"))):
                for c in cfgs:
                    for n in names:
                        shutil.copy(os.path.join(L, "libhwer_hip_%s.so" % n), prod)
                        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lat_probe.py"), B, os.environ.get("AB_ITERS", "20")],
                                           env=dict(os.environ, HWER_PROBE_CFG=c), capture_output=True, text=True, timeout=300)
                        line = "%s rep%d [%s]: %s" % (c, rep, n, (r.stdout.strip().splitlines() or ["FAILED " + r.stderr[-300:]])[-1])
                        print(line, flush=True)
                        fh.write(line + "\n")
                        fh.flush()
    finally:
        shutil.copy(keep, prod)


if __name__ == "__main__":
    main()
