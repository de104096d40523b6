#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run (gpurun_out/<dir>) into profiles/ under this round's names (r<ROUND>_*) and
rebuild profiles/r<ROUND>_pmc_traffic.json (the HBM-side bytes of the screen stage per launch that bench.py labels `traffic`),
stamped with the digest of the library sources the counters were collected on (bench.py reports `traffic` only for that build).

    python tools/collect_profiles.py gpurun_out/prof "c1 c2 c3 c4" [ROUND=4]
"""
import json
import os
import re
import sys

src = sys.argv[1]
cfgs = (sys.argv[2] if len(sys.argv) > 2 else "c1 c2 c3 c4").split()
RND = "r%s" % (sys.argv[3] if len(sys.argv) > 3 else "4")
CMD = "python3 bench.py --config %s --no-cpu-baseline --sweep-only --no-graph"


def pmc_rows(path):
    """kernel -> {counter: (avg, n)} from a prof_summary.py pmc file"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(.{60}) (.*)", line.rstrip("\n"))
        if not m:
            continue
        kern = m.group(1).strip()
        for c in re.finditer(r"(\w+) avg=([0-9.e+\-]+) n=(\d+)", m.group(2)):
            out.setdefault(kern, {})[c.group(1)] = (float(c.group(2)), int(c.group(3)))
    return out


traffic = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC passes of `" + CMD % "<c>" + " --steps 3 --warmup 1` "
           "(profiles/" + RND + "_rocprof_pmc_<c>.txt): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE is doubled per MI355X_MICROARCH.md "
           "(gfx950 counts wide reads at half their bytes).  knn_screen_bytes_per_launch = the screen stage of the configuration's main search, "
           "per-launch averages of the kernels listed in knn_screen_kernels: the filtering screen and - streaming configurations - the seed "
           "pre-pass (dense-mode screen over the sample) and its threshold kernel."}
for c in cfgs:
    j = json.load(open(f"{src}/stats_{c}_bench.json"))
    hdr = [f"# rocprofv3 --kernel-trace --stats -- {CMD % c} --steps N --warmup 5   (MI355X, " + RND + "; tools/profile_round.sh)",
           "# bench line under the profiler: ms_per_step=%.4f kernels=%s" % (j["ms_per_step"], json.dumps({k["kernel"][:28]: round(k["ms_per_step"], 4) for k in j["kernels"]}))]
    open(f"profiles/{RND}_rocprof_stats_{c}.txt", "w").write(
        "\n".join(hdr) + "\n" + open(f"{src}/stats_{c}.txt").read() +
        "\n# per-step timeline (tools/trace_timeline.py): mean start offset / duration of every kernel of a step\n" + open(f"{src}/timeline_{c}.txt").read())
    hdr = (f"# rocprofv3 --pmc <counter> -- {CMD % c} --steps 3 --warmup 1  (one pass per counter; averages per launch; FETCH/WRITE_SIZE in KiB)\n")
    open(f"profiles/{RND}_rocprof_pmc_{c}.txt", "w").write(hdr + open(f"{src}/pmc_{c}.txt").read())
    if os.path.exists(f"{src}/sq_{c}.txt"):
        open(f"profiles/{RND}_rocprof_sq_{c}.txt", "w").write(f"# rocprofv3 --pmc <SQ counters> -- {CMD % c} --steps 3 --warmup 1 (averages per launch)\n" + open(f"{src}/sq_{c}.txt").read())
    rows = pmc_rows(f"{src}/pmc_{c}.txt")
    ent = {"kernels": {}}
    total = 0.0
    for kern, d in rows.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            b = (2 * d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024
            ent["kernels"][kern] = {"fetch_size_kb": d["FETCH_SIZE"][0], "write_size_kb": d["WRITE_SIZE"][0], "bytes_per_launch": int(b), "launches": d["FETCH_SIZE"][1]}
    # the screen stage of the configuration's MAIN search (per-launch averages): the filtering screen kernel with the largest
    # traffic, plus - streaming configurations - the seed pre-pass (dense-mode screen of the same template arguments) and the
    # threshold kernel that precede it in the same search
    def pick(pred):
        best = None
        for kern, e in ent["kernels"].items():
            if pred(kern) and (best is None or e["bytes_per_launch"] > ent["kernels"][best]["bytes_per_launch"]):
                best = kern
        return best
    # (round 4: the float16 streaming screen is screen16_pp_kernel<NG, TQ>; its seed pre-pass is still screen_kernel<NG, TQ, true, true>)
    main = (pick(lambda k: "screen16_pp_kernel<" in k) or pick(lambda k: "screen_kernel<" in k and ", false, " in k) or pick(lambda k: "screen_kernel<" in k))
    parts = [main]
    if main and ("screen16_pp_kernel<" in main or ", false, " in main):
        if "screen16_pp_kernel<" in main:
            seed = pick(lambda k: "screen_kernel<" in k and ", true, true" in k)
        else:
            seed = main.replace(", false, ", ", true, ")
        if seed in ent["kernels"]:
            parts.append(seed)
        thr = pick(lambda k: "small_threshold" in k or "seed_pick" in k)
        if thr:
            parts.append(thr)
    total = sum(ent["kernels"][k]["bytes_per_launch"] for k in parts if k)
    ent["knn_screen_kernels"] = [k for k in parts if k]
    ent["knn_screen_bytes_per_launch"] = int(total)
    traffic[c] = ent
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hybrid-weighted-embedding-recommender_amd", "csrc"))
import build as hip_build  # noqa: E402
dig_file = os.path.join(src, "library_digest.txt")      # written on the GPU box by tools/profile_round.sh next to the counters
traffic["library_digest"] = open(dig_file).read().strip() if os.path.exists(dig_file) else hip_build._digest()
json.dump(traffic, open("profiles/%s_pmc_traffic.json" % RND, "w"), indent=1)
for c in cfgs:
    print(c, traffic[c]["knn_screen_bytes_per_launch"] / 1e6, "MB per screen launch")
