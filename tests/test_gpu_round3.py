"""GPU parity tests added in round 3: the full-scale 8-shard rehearsal of the multi-GPU job on one GPU, the
split screen / select entry points, the float64 re-rank table (G14), K-chunked wide tables.  Bit-exact for indices / scores /
distances unless an assert states a tolerance."""
import os
import sys

import numpy as np
import pytest

from conftest import unit_rows_f32
from oracle import c_oracle
from oracle import hwer_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hw():
    import hwer
    from hwer import native
    native.lib()
    return hwer


@pytest.fixture(scope="module")
def bench_mod():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hwer_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------- 8 shards on one GPU
@pytest.mark.parametrize("name,items,G", [("c4full", 10000000, 8), ("c4full", 1000003, 3)])
def test_emulated_shards_equal_single_table(hw, bench_mod, name, items, G):
    """SURVEY.md 8e at BASELINE.json configs[3]'s full size: the 10 M x 128 catalogue cut into the 8 `shard_bounds` ranges, every
    shard searched like rank g of the 8-GPU job (own blend, own table, ids = id_base + local row, (distance, id) order), packed
    by hwer_pack_exchange into slice g of a (8, 3, B, k) buffer laid out like the all-gather's receive buffer, merged in place by
    hwer_merge_topk_strided: ids, scores and squared distances must EQUAL the single-table result for all 4 096 queries and the
    oracle's brute force over the whole catalogue on 32 of them.  (A ragged 3-shard cut of 1 000 003 rows as well.)"""
    import torch
    from hwer import native
    cfg = dict(bench_mod.CONFIGS[name])
    cfg["I"] = items
    dev = torch.device("cuda", 0)
    r = bench_mod.emulate_shards(torch, native.lib(), cfg, dev, G, steps=1, warmup=1, n_oracle=32)
    assert r["merged_equals_single_table"] and r["queries_compared"] == 4096 and r["oracle_checked_queries"] == 32
    assert len(r["shard_ms"]) == G
