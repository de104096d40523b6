import sys, os, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/hybrid-weighted-embedding-recommender_amd')
import importlib.util
spec = importlib.util.spec_from_file_location("hwer_bench", "bench.py"); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
from oracle import c_oracle; c_oracle.lib()
import torch
dev = torch.device("cuda", 0)
print(json.dumps(mod.wide_probe(torch, dev)))
