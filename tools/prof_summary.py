"""Condense rocprofv3 output (kernel-trace stats + PMC counter CSVs) into a small text summary for profiles/."""
import collections, csv, glob, sys

def short(name):
    name = name.split("(")[0]
    return name[-70:]

def stats(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    out = ["%-72s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct")]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        out.append("%-72s %8s %12.3f %12.2f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(out)

def pmc(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = []
    for k, dd in agg.items():
        if "hwer" in k:
            out.append("%-60s " % k + "  ".join("%s avg=%.4g n=%d" % (c, sum(v) / len(v), len(v)) for c, v in sorted(dd.items())))
    return "\n".join(out)

if __name__ == "__main__":
    kind, d = sys.argv[1], sys.argv[2]
    print(stats(d) if kind == "stats" else pmc(d))
