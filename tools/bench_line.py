"""Condense bench.py JSON lines from stdin into one short line each."""
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l)
        r = j["roofline"]
        print("%s ms/step %.3f value %.3e screen %.3f ms x%.0f (%.1f TF frac %.3f) select %.3f ms fallback %d%s" % (
            j["config"]["workload"][:24], j["ms_per_step"], j["value"], r["avg_launch_ms"], r["launches_per_step"], r["achieved"], r["frac"],
            j["kernels_ms"]["knn_select_rerank"], j["exhaustive_fallback_queries"],
            (" cpu %.2e (%d cores) x%.0f" % (j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["speedup_vs_cpu"])) if "cpu_baseline" in j else ""))
    elif "Error" in l or "error" in l or "assert" in l:
        print(l.strip()[:300])
