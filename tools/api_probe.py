"""one-off: wall time of the per-anchor API at the c1 / c2 shapes (GPU box)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"))
import importlib.util
spec = importlib.util.spec_from_file_location("hwer_bench", os.path.join(ROOT, "bench.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
from oracle import c_oracle; c_oracle.lib()
import numpy as np, torch
for U, I, d in ((943, 1682, 64), (6040, 3706, 128)):
    Vh = torch.cat([mod.gen_unit_rows(torch, 1, 0, U, d, "cuda:0"), mod.gen_unit_rows(torch, 2, 0, I, d, "cuda:0")]).cpu().numpy()
    r = mod.api_latency(torch, Vh, U, I, d)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not k.endswith("note") and not isinstance(v, dict)}))
