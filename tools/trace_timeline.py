"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: mean start offset / duration of every kernel of a step."""
import csv, glob, sys, collections
d = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else "blend_normalize"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
steps = []; cur = None
for s, e, n in rows:
    if anchor in n:
        cur = []; steps.append(cur)
    if cur is not None: cur.append((s, e, n))
steps = [st for st in steps[5:-1]]
agg = collections.OrderedDict()
tot = []
for i, st in enumerate(steps[:-1]):
    t0 = st[0][0]
    for j, (s, e, n) in enumerate(st):
        agg.setdefault((j, n), []).append(((s - t0) / 1e3, (e - s) / 1e3))
    tot.append((steps[i + 1][0][0] - t0) / 1e3)
for (j, n), v in agg.items():
    if len(v) < max(2, len(steps) // 4):   # one-off launches between the steps (set-up, result checks)
        continue
    print("%2d %-50s start %8.1f us  dur %8.1f us  (n=%d)" % (j, n, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v)))
print("step period %.1f us over %d steps" % (sum(tot) / len(tot), len(tot)))
