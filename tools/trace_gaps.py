#!/usr/bin/env python3
"""Timeline of one search from a rocprofv3 --kernel-trace CSV: the LAST repetition of the kernel sequence that starts with the
kernel named FIRST - per kernel its duration and the idle gap in front of it.
    python tools/trace_gaps.py kernel_trace.csv FIRST_KERNEL_SUBSTRING [n_last]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = sys.argv[2]
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for s_i in range(len(starts) - n_last, len(starts)):
        i0 = starts[s_i]
        i1 = starts[s_i + 1] if s_i + 1 < len(starts) else len(rows)
        t0 = int(rows[i0]["Start_Timestamp"])
        prev_end = t0
        print("search #%d" % s_i)
        for r in rows[i0:i1]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print("  +%8.2f us  gap %6.2f  dur %8.2f  grid %s wg %s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"),
                                                                       r["Kernel_Name"].replace("void hwer::", "").replace("hwer::", "")[:70]))
            prev_end = e
        print("  total %.2f us" % ((prev_end - t0) / 1e3))


if __name__ == "__main__":
    main()
