#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run (gpurun_out/<dir>) into profiles/ under the round's names.

    python tools/collect_profiles.py gpurun_out/final2 v9
"""
import json
import shutil
import sys

src, ver = sys.argv[1], sys.argv[2]
for c in ["c1", "c2", "c3", "c4", "c5"]:
    shutil.copy(f"{src}/bench_{c}.json", f"profiles/r1_bench_{c}_{ver}.json")
for c in ["c2", "c3"]:
    j = json.load(open(f"{src}/stats_{c}_bench.json"))
    hdr = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --config {c} --no-cpu-baseline --sweep-only   (MI355X, round 1, final kernels; tools/profile_round.sh)",
           "# (55 calls per kernel = 5 warm-up + 50 timed steps; --sweep-only skips the one-anchor latency and pair-score loops after the timed region)",
           "# bench line under the profiler: ms_per_step=%.3f roofline=%s" % (j["ms_per_step"], json.dumps(j["roofline"]))]
    open(f"profiles/r1_rocprof_stats_{c}.txt", "w").write(
        "\n".join(hdr) + "\n" + open(f"{src}/stats_{c}.txt").read() +
        "\n# per-step timeline (tools/trace_timeline.py): mean start offset / duration of every kernel of a step\n" + open(f"{src}/timeline_{c}.txt").read())
    hdr = (f"# rocprofv3 --pmc <counter> -- python3 bench.py --config {c} --no-cpu-baseline --sweep-only --steps 3 --warmup 1  (one pass per counter; "
           "averages per launch; FETCH/WRITE_SIZE in KiB; final kernels)\n")
    open(f"profiles/r1_rocprof_pmc_{c}.txt", "w").write(hdr + open(f"{src}/pmc_{c}.txt").read())
shutil.copy(f"{src}/micro.jsonl", "profiles/r1_micro_rowops_latency.jsonl")


def grab(c, kern, ctr):
    best = None
    for line in open(f"{src}/pmc_{c}.txt"):
        if kern in line and ctr in line:
            v, n = float(line.split("avg=")[1].split()[0]), int(line.split("n=")[1].split()[0])
            if best is None or v > best:
                best = v
    return best


t = json.load(open("profiles/r1_pmc_traffic.json"))
# c2: the dense screen kernel of the full sweep (the largest-traffic screen_kernel<..., true> entry)
f2, w2 = grab("c2", "screen_kernel<16, ", "FETCH_SIZE"), grab("c2", "screen_kernel<16, ", "WRITE_SIZE")
t["c2"].update(knn_screen_bytes_per_launch=int((2 * f2 + w2) * 1024), fetch_size_kb=f2, write_size_kb=w2)
fs, ws = grab("c3", "hwer::screen_kernel<16, 2, false", "FETCH_SIZE"), grab("c3", "hwer::screen_kernel<16, 2, false", "WRITE_SIZE")
fd, wd = grab("c3", "hwer::small_screen", "FETCH_SIZE"), grab("c3", "hwer::small_screen", "WRITE_SIZE")
t["c3"]["streaming_launch"].update(fetch_size_kb=fs, write_size_kb=ws, bytes=int((2 * fs + ws) * 1024))
t["c3"]["dense_key_launch"].update(fetch_size_kb=fd, write_size_kb=wd, bytes=int((2 * fd + wd) * 1024))
t["c3"]["knn_screen_bytes_per_launch"] = int(((2 * fs + ws) + (2 * fd + wd)) * 1024 / 2)
json.dump(t, open("profiles/r1_pmc_traffic.json", "w"), indent=1)
for c in ["c1", "c2", "c3", "c4", "c5"]:
    j = json.load(open(f"{src}/bench_{c}.json"))
    cb = j.get("cpu_baseline", {})
    print("%s ms=%.4f value=%.3e frac=%.3f screen=%.4f select=%.4f cpu=%.3e x%.0f vec=%.3e lat=%.4f" % (
        c, j["ms_per_step"], j["value"], j["roofline"]["frac"], j["kernels_ms"]["knn_screen"], j["kernels_ms"]["knn_select_rerank"],
        cb.get("value", 0), j.get("speedup_vs_cpu", 0), cb.get("vectorised_f32", {}).get("value", 0), j["query_latency_ms"]["value"]))
