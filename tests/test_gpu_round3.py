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


# ------------------------------------------------------------------------------------------------- the step as library code
@pytest.mark.parametrize("layout,U,I,d,T,per_row", [("contiguous", 300, 1700, 64, 2, False), ("interleaved", 257, 2100, 128, 3, True),
                                                    ("contiguous", 90, 9001, 64, 4, True)])
def test_blend_search_step_equals_the_separate_calls(hw, layout, U, I, d, T, per_row):
    """ContentRecommendation.blend_search_step (hwer.pipeline.BlendSearchStep: the ONE-launch prologue hwer_blend_prepare_search_f32,
    HWER_KNN_QUERIES_READY, audit-derived float16 scale, hipGraph replay - what bench.py times) must return exactly what the
    ordinary library path returns (fit_tables -> find_closest_neighbours_batch: hwer_blend_normalize_f32, upload, hwer_unit_audit,
    hwer_pack_rows, hwer_query_mix, search) and what the oracle computes from the float64 blend of the same tables (ids ==,
    scores ==); contiguous and interleaved node types (id-list row sets), scalar and per-row alpha; a second run after the
    weights changed in place follows the new weights."""
    import torch
    from hwer import ContentRecommendation, Node
    rng = np.random.default_rng(7 + U)
    N = U + I
    if layout == "contiguous":
        types = ["user"] * U + ["item"] * I
    else:
        types = ["item" if (i % (N // U)) else "user" for i in range(N)]
        types = types[:N]
    nodes = [Node(t, i) for i, t in enumerate(types)]
    n_user = sum(1 for t in types if t == "user")
    tables = [unit_rows_f32(900 + 13 * t + d, N, d) for t in range(T)]
    if per_row:
        alpha = rng.dirichlet(np.ones(T), size=N).astype(np.float32)
    else:
        alpha = np.full(T, 1.0 / T, dtype=np.float32)
    users = [n for n in nodes if n.node_type == "user"]
    k = 200 if I >= 200 else 50

    def reference(al):
        m = ContentRecommendation({}, {"user", "item"}, n_dims=d)
        m.fit_tables(nodes, tables, al)
        return m.find_closest_neighbours_batch("item", users, k=k)

    model = ContentRecommendation({}, {"user", "item"}, n_dims=d)
    step = model.blend_search_step(nodes, tables, alpha, "item", users, k=k)
    assert step.fused_blend and (step._graph is not None) == (I <= 8192)
    step.run()
    idx, score = step.results()
    r_idx, r_score = reference(alpha)
    assert np.array_equal(idx, r_idx) and np.array_equal(score, r_score)
    assert step.audit()[0] == 0
    # new weights, written in place into the resident tensor: the replayed step follows them
    alpha2 = (rng.dirichlet(np.ones(T), size=N) if per_row else rng.dirichlet(np.ones(T))).astype(np.float32)
    step.alpha.copy_(torch.from_numpy(alpha2).to(step.alpha.device))
    step.run()
    idx2, score2 = step.results()
    r_idx2, r_score2 = reference(alpha2)
    assert np.array_equal(idx2, r_idx2) and np.array_equal(score2, r_score2)
    assert not np.array_equal(idx2, idx)


# ------------------------------------------------------------------------------------------------- seeded lists longer than one window
def test_seeded_lists_longer_than_the_merge_window(hw):
    """reduce_lists_wave_kernel merges ALL lists of a query (round 3: also when they hold more than its 4 096-key window - the
    workgroup rounds behind it are gone).  Forced here: ~6 300 near-duplicates of the queries' neighbourhood sit in rows that the
    seed pre-pass's strided sample never visits (sample tile p = catalogue tile 35 p at this shape), spread over the 32 item
    splits so that no single list compacts - every query then carries ~7 000 keys above its seeded threshold into the merge.
    ids / scores / distances == the oracle's brute force."""
    from hwer.device_index import DeviceIndex
    N, d, B, k = 300000, 64, 4096, 200
    rng = np.random.default_rng(77)
    X = unit_rows_f32(501, N, d)
    c = unit_rows_f32(502, 1, d)[0].astype(np.float64)
    tiles = np.arange(N) // 32
    rows = np.arange(N)[(tiles % 35 != 0) & (np.arange(N) % 46 == 7)]
    assert len(rows) >= 6000
    near = c[None, :] + 0.05 * rng.standard_normal((len(rows), d))
    X[rows] = (near / np.linalg.norm(near, axis=1, keepdims=True)).astype(np.float32)
    q = c[None, :] + 0.5 * rng.standard_normal((B, d)) / np.sqrt(d)
    users = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    V = np.concatenate([users, X]).astype(np.float64)
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all()
    sel = np.linspace(0, B - 1, 64).astype(np.int64)
    Qh = np.stack([orc.query_vector(V[:B], int(u)) for u in sel])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(X, Qh, V[sel], k, threads=c_oracle.effective_cpus())
    assert np.isin(o_idx[:, :50], rows).all()          # the planted rows are the neighbours
    assert np.array_equal(idx[sel], o_idx + B) and np.array_equal(score[sel], o_score) and np.array_equal(dist[sel], np.sqrt(o_d2))
