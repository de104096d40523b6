"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: mean start offset / duration of every kernel of a step."""
import csv, glob, sys, collections
d = sys.argv[1]; anchor = sys.argv[2] if len(sys.argv) > 2 else None
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
rows.sort()
if anchor is None:
    # a step starts with its prologue launch: the fused one-launch prologue (blend_prepare_search_kernel: c1, c2, c4, c5, c4full) or the separate
    # blend (blend_normalize_kernel: c3).  Whichever of the two was launched MORE OFTEN is the steps' first kernel - the other one, if it
    # appears at all, is the set-up blend of the workload (round 4's timelines anchored on "blend_normalize" and showed that set-up launch
    # instead of the step on every configuration with the fused prologue).
    n_fused = sum(1 for _, _, n in rows if "blend_prepare_search" in n)
    n_sep = sum(1 for _, _, n in rows if "blend_normalize" in n)
    anchor = "blend_prepare_search" if n_fused > n_sep else "blend_normalize"
print("# steps anchored on the first launch whose name contains %r" % anchor)
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
