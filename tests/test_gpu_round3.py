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
@pytest.mark.parametrize("name,items,G,n_oracle", [("c4full", 10000000, 8, 256), ("c4full", 1000003, 3, 256), ("c5full", 10000000, 8, 64)])
def test_emulated_shards_equal_single_table(hw, bench_mod, name, items, G, n_oracle):
    """SURVEY.md 8e at BASELINE.json configs[3]'s full size: the 10 M x 128 catalogue cut into the 8 `shard_bounds` ranges, every
    shard searched like rank g of the 8-GPU job (own blend, own table, ids = id_base + local row, (distance, id) order), packed
    by hwer_pack_exchange into slice g of a (8, 3, B, k) buffer laid out like the all-gather's receive buffer, merged in place by
    hwer_merge_topk_strided: ids, scores and squared distances must EQUAL the single-table result for all 4 096 queries and the
    oracle's brute force over the whole catalogue on 256 of them (round 4; 32 before).  (A ragged 3-shard cut of 1 000 003 rows as well.)
    Round 6: the same for configs[4] at ITS full size - c5full: 4 tables, per-row alpha, d = 256, 131 072 user + 10 M item rows blended,
    8 shards of 1.25 M rows, oracle on 64 queries."""
    import torch
    from hwer import native
    cfg = dict(bench_mod.CONFIGS[name])
    cfg["I"] = items
    dev = torch.device("cuda", 0)
    r = bench_mod.emulate_shards(torch, native.lib(), cfg, dev, G, steps=1, warmup=1, n_oracle=n_oracle)
    assert r["merged_equals_single_table"] and r["queries_compared"] == 4096 and r["oracle_checked_queries"] == n_oracle
    assert len(r["shard_ms"]) == G
    # round 3: the shards sent shortened lists (72 of 200 entries for 8 shards, 136 for 3) and no list ran out
    from hwer.distributed import local_list_len
    assert r["local_list_len"] == local_list_len(200, G) < 200 and r["lists_ran_out_first_pass"] == 0


def _merge_lists_numpy(g_idx, g_score, g_d2, k, sort_by_score):
    G, B, k_in = g_idx.shape
    o_i = np.empty((B, k), np.int64); o_s = np.empty((B, k)); o_d = np.empty((B, k)); ex = np.zeros(B, np.int32)
    for b in range(B):
        dd, ii, ss = g_d2[:, b].reshape(-1), g_idx[:, b].reshape(-1), g_score[:, b].reshape(-1)
        order = np.lexsort((np.arange(G * k_in), ii, dd))[:k]      # (dist2, id), source position last (padding rows share id -1)
        taken = np.bincount(order // k_in, weights=(ii[order] >= 0), minlength=G)
        ex[b] = int(k_in < k and (taken >= k_in).any())
        if sort_by_score:
            order = order[np.argsort(-ss[order], kind="stable")]
        o_i[b], o_s[b], o_d[b] = ii[order], ss[order], dd[order]
    return o_i, o_s, o_d, ex


@pytest.mark.parametrize("G,B,k_in,k,skew", [(8, 300, 72, 200, 0.0), (8, 64, 25, 200, 0.0), (4, 257, 136, 200, 0.5), (2, 33, 100, 200, 0.0),
                                              (16, 40, 13, 200, 0.9), (3, 50, 7, 20, 0.0), (8, 31, 200, 200, 0.0), (5, 17, 1, 5, 0.0)])
def test_merge_topk_lists_shortened(hw, G, B, k_in, k, skew):
    """hwer_merge_topk_lists: G sorted lists of k_in <= k entries -> the k nearest + the per-query flag `a list ran out` - against a
    numpy merge of the same lists (ids, scores, distances ==, flags ==), both output orders; duplicate distances across lists (ties broken
    by id), padded tails (id -1, +inf) in a third of the lists, and `skew` = share of queries whose nearest all sit in list 0."""
    import torch
    from hwer.distributed import merge_on_device
    rng = np.random.default_rng(G * 1000 + k_in)
    g_idx = np.empty((G, B, k_in), np.int64); g_d2 = np.empty((G, B, k_in)); g_sc = np.empty((G, B, k_in))
    for g in range(G):
        for b in range(B):
            n_real = k_in if (g + b) % 3 else int(rng.integers(0, k_in + 1))
            d2 = np.round(rng.random(n_real) * 50) / 50.0 + 1.0              # few distinct values: ties within and across lists
            if g == 0 and rng.random() < skew:
                d2 = d2 * 1e-3                                                # list 0 holds this query's nearest
            ids = rng.choice(1000000, n_real, replace=False) * G + g          # distinct ids across lists
            o = np.lexsort((ids, d2))
            g_d2[g, b, :n_real], g_idx[g, b, :n_real] = d2[o], ids[o]
            g_d2[g, b, n_real:], g_idx[g, b, n_real:] = np.inf, -1
            g_sc[g, b] = rng.random(k_in)
            g_sc[g, b, n_real:] = 0.0
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x).to(dev) for x in (g_idx, g_sc, g_d2)]
    total_real = (g_idx >= 0).sum(axis=(0, 2))
    for sort_by_score in (True, False):
        m_i, m_s, m_d, ex = merge_on_device(*t, sort_by_score, k=k, want_exhausted=True)
        r_i, r_s, r_d, r_ex = _merge_lists_numpy(g_idx, g_sc, g_d2, k, sort_by_score)
        ok = total_real >= k        # (queries with fewer than k real entries in all lists end in padding rows: only the real part is compared)
        assert np.array_equal(ex.cpu().numpy(), r_ex)
        assert np.array_equal(m_i.cpu().numpy()[ok], r_i[ok]) and np.array_equal(m_s.cpu().numpy()[ok], r_s[ok]) and np.array_equal(m_d.cpu().numpy()[ok], r_d[ok])
        if not sort_by_score:
            short = ~ok
            for b in np.nonzero(short)[0]:
                n = int(total_real[b])
                assert np.array_equal(m_i.cpu().numpy()[b, :n], r_i[b, :n]) and np.array_equal(m_d.cpu().numpy()[b, :n], r_d[b, :n])
    if k_in < k and skew > 0:
        assert r_ex.sum() > 0
    if k_in == k:
        assert r_ex.sum() == 0


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


# ------------------------------------------------------------------------------------------------- split search = fused search
@pytest.mark.parametrize("B,N,d,k,opts", [(943, 1682, 64, 200, 0), (3000, 30000, 128, 200, 2), (40, 200000, 128, 100, 0), (1500, 16000, 256, 50, 2),
                                          (2100, 30000, 128, 200, 16)])
def test_screen_then_select_equals_the_one_call_search(hw, B, N, d, k, opts):
    """hwer_knn_screen_f32 + hwer_knn_select_f32 (round 3: the stages of a search as two stream-ordered calls, so that callers can
    pipeline batches) write exactly what hwer_knn_search_opts_f32 writes - ids, scores, distances, flags - on the dense, streaming
    (float32 / float16 / unseeded) and small-batch paths; the pipelined BlendSearchStep (three chunks over two streams) equals the
    one-call step."""
    import torch
    from hwer import native
    from hwer.device_index import DeviceIndex
    lib = native.lib()
    V = np.concatenate([unit_rows_f32(31 + d, B, d), unit_rows_f32(32 + d, N, d)]).astype(np.float64)
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    tt = ix.tables["item"]
    Q64, Q32, a = ix.build_queries(np.arange(B))
    x16, x16_exp = ix._x16(tt, B) if (opts & 2 or B <= 1024) else (None, 0)
    P, dev = native.ptr, ix.device
    wsb = lib.hwer_knn_workspace_bytes_opts(B, N, d, k, opts)
    out = []
    for split in (False, True):
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        idx = torch.full((B, k), -7, dtype=torch.int64, device=dev)
        score = torch.zeros((B, k), dtype=torch.float64, device=dev)
        d2 = torch.zeros((B, k), dtype=torch.float64, device=dev)
        flags = torch.full((B,), -7, dtype=torch.int32, device=dev)
        st = native.stream_ptr(dev)
        if split:
            native.check(lib.hwer_knn_screen_f32(P(tt.X), P(tt.hnorm), N, d, ix.ldx, P(Q32), B, k, P(ws), ws.numel(), opts, P(x16), x16_exp, st), "screen")
            native.check(lib.hwer_knn_select_f32(P(tt.X), N, d, ix.ldx, P(Q64), P(ix.V), ix.ldv, P(a), B, k, 0, 1, P(tt.rows_dev), 0, tt.hnorm_max,
                                                 P(idx), P(score), P(d2), P(flags), P(ws), ws.numel(), opts, P(x16), st), "select")
        else:
            native.check(lib.hwer_knn_search_opts_f32(P(tt.X), P(tt.hnorm), N, d, ix.ldx, P(Q32), P(Q64), P(ix.V), ix.ldv, P(a), B, k, 0, 1,
                                                      P(tt.rows_dev), 0, tt.hnorm_max, P(idx), P(score), P(d2), P(flags), P(ws), ws.numel(), opts,
                                                      P(x16), x16_exp, st), "search")
        torch.cuda.synchronize()
        out.append([t.cpu().numpy() for t in (idx, score, d2, flags)])
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    assert (out[0][3] > 0).all()


def test_pipelined_step_equals_the_one_call_step(hw):
    import torch
    from hwer.pipeline import BlendSearchStep
    dev = torch.device("cuda", 0)
    U, I, d = 3 * 700 + 55, 20000, 128
    tabs = [torch.from_numpy(np.concatenate([unit_rows_f32(71 + t, U, d), unit_rows_f32(81 + t, I, d)])).to(dev) for t in range(2)]
    alpha = torch.tensor([0.5, 0.5], dtype=torch.float32, device=dev)
    anchors = torch.arange(U, dtype=torch.int32, device=dev)
    import hwer.pipeline as pl
    res = []
    for pipe in (False, True):
        old = pl.PIPE_CHUNK
        pl.PIPE_CHUNK = 700 if pipe else old          # three full chunks + a short one over two workspaces
        try:
            step = BlendSearchStep(tabs, alpha, U, I, 0, U, anchors, k=200, want_dist=True, pipeline=pipe)
        finally:
            pl.PIPE_CHUNK = old
        assert step.pipeline == pipe and (not pipe or len(step.pipe_chunks) == 4)
        step.run()
        step.run()
        res.append(step.results())
    for x, y in zip(*res):
        assert np.array_equal(x, y)


# ------------------------------------------------------------------------------------------------- true float64 flavour (G14)
def g14_inputs():
    U, I, d = 64, 4096, 64
    x = np.random.default_rng(1400).standard_normal((U + I, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x, U, I, d


def test_float64_rows_g14(golden, hw):
    """north_star's "neighbour indices exactly" for the reference's float64 flavour with GENUINE float64 rows (G14, captured from the
    reference: hwer/content_recommender.py:96, hwer/recommendation_base.py:74,79,109,135-174): the float64 table stays resident beside
    its float32 rounding, the HIP search screens the rounding for k + 32 candidates, hwer_knn_cand_dist2 / hwer_pair_score_f64rows /
    hwer_query_mix_f64rows produce the values from the float64 rows.  ids ==, scores == (bit for bit), KD-tree distances ==, predict
    ==, positive / negative-list queries ==."""
    import hashlib
    from hwer import Node, VectorIndex
    g = golden("g14_float64_rows")
    V, U, I, d = g14_inputs()
    assert hashlib.sha256(np.ascontiguousarray(V).tobytes()).hexdigest() == str(g["input_sha256"])
    users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(I)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    m.fit(users + items, V)
    assert m.knn.index.V64 is not None and m.knn.index.arith == 0
    k = int(g["k"])
    idx, score = m.find_closest_neighbours_batch("item", users, k=k)
    assert np.array_equal(idx - U, g["idx"].astype(np.int64))
    assert np.array_equal(score, g["score"])
    one = m.find_closest_neighbours("item", users[5], k=k)
    assert [m.nodes_to_idx[n] - U for n, _ in one] == g["idx"][5].tolist() and [s for _, s in one] == g["score"][5].tolist()
    for u in range(8):
        res = m.knn.query(m.get_average_embeddings([users[u]]), "item", k=k)
        assert [m.nodes_to_idx[n] for n, _ in res] == g["query_idx"][u].tolist()
        assert np.array_equal(np.asarray([dd for _, dd in res]), g["query_dist"][u])
    assert np.array_equal(m.predict_indices(g["pair_src"], g["pair_dst"]), g["pair_score"])
    po, no = np.r_[0, np.cumsum(g["pn_pos_len"])], np.r_[0, np.cumsum(g["pn_neg_len"])]
    table = m.knn.idx_to_node()
    pos = [[table[int(j)] for j in g["pn_pos"][po[a]:po[a + 1]]] for a in range(16)]
    neg = [[table[int(j)] for j in g["pn_neg"][no[a]:no[a + 1]]] for a in range(16)]
    idx_pn, sc_pn = m.find_closest_neighbours_batch("item", users[:16], k=50, positive=pos, negative=neg)
    assert np.array_equal(idx_pn, g["pn_idx"]) and np.array_equal(sc_pn, g["pn_score"])


@pytest.mark.parametrize("N,d,k,B", [(20000, 96, 200, 300), (3000, 300, 393, 64), (1200, 64, 1200, 40)])
def test_float64_rows_vs_oracle(hw, N, d, k, B):
    """genuine float64 rows at other shapes, all against the numpy oracle on the float64 rows: a streaming-size catalogue, a wide
    table with k above the select kernels' limit (library-GEMM screen), and k = every row; near-duplicate rows (differing below
    float32 resolution) included - their order is decided by the float64 distances."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(90 + d)
    X = rng.standard_normal((B + N, d))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[B + 11] = X[B + 10] * (1.0 + 3e-9)            # the same float32 row, different float64 rows
    X[B + 12] = X[B + 10] * (1.0 - 2e-9)
    ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    assert ix.V64 is not None
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all()
    rows = np.arange(B, B + N)
    for u in list(range(0, B, max(1, B // 12))) + [B - 1]:
        o_idx, o_score, o_dist = orc.find_closest(X, rows, u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist), u
    # a query right at the planted rows
    q_near = np.where(np.arange(B) == 0, 0, 0)
    src, dst = rng.integers(-1, B + N, 500), rng.integers(-1, B + N, 500)
    assert np.array_equal(ix.pair_score(src, dst), orc.pair_score(X, src, dst))


# ------------------------------------------------------------------------------------------------- extreme shapes through the index
@pytest.mark.parametrize("U,I,d,k", [(1, 1, 1, 1), (3, 2, 2, 2), (5, 33, 3, 1), (5, 33, 3, 33), (2, 64, 64, 64), (7, 65, 17, 65), (1, 392, 8, 392),
                                      (4, 8193, 5, 1), (3, 300, 256, 299)])
def test_extreme_shapes_vs_oracle(hw, U, I, d, k):
    """Degenerate corners of the hot path through DeviceIndex.search (the call behind find_closest_neighbours_batch): a one-row catalogue, k = 1,
    k = every row of the type, d = 1, one column short of / one row past the tile and path thresholds (64-row tiles, 8 192-row dense limit,
    k = 392 = the fused search's largest) - ids, scores and distances == the oracle; an EMPTY anchor batch returns empty arrays."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(U * 1000 + I + d)
    V = rng.standard_normal((U + I, d))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    V = V.astype(np.float32)
    ix = DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, U + I)}, arith=0)
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
    assert idx.shape == (U, k) and (flags > 0).all()
    V64 = V.astype(np.float64)
    Qh = np.stack([orc.query_vector(V64, u) for u in range(U)])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(V[U:], Qh, V64[:U], k, threads=2)
    assert np.array_equal(idx, o_idx + U) and np.array_equal(score, o_score) and np.array_equal(dist, np.sqrt(o_d2))
    e_idx, e_score, e_dist, e_flags = ix.search("item", np.zeros(0, dtype=np.int64), k=k)
    assert e_idx.shape == (0, k) and e_score.shape == (0, k) and e_flags.shape == (0,)
    with pytest.raises(ValueError):
        ix.search("item", np.arange(U), k=I + 1)


# ------------------------------------------------------------------------------------------------- BASELINE configs at full size, as the bench runs them
@pytest.mark.parametrize("name,n_check", [("c1", 943), ("c2", 6040), ("c3", 138493), ("c5", 4096)])
def test_bench_step_at_full_size_vs_oracle(hw, bench_mod, name, n_check):
    """The step bench.py times (hwer.pipeline.BlendSearchStep on the synthetic tables of BASELINE.json's configs, full size: c1 943 x 1 682,
    c2 6 040 x 3 706, c3 138 493 x 26 744, c5 four tables / per-row alpha / d = 256 / 131 072 + 1.25 M blended rows / 4 096 queries): unit-norm
    audit clean, every query certified or recomputed, and `n_check` evenly spaced queries - round 4: ALL of c1 / c2 / c5, round 5: ALL 138 493 of c3 too (~20 s of the
    box's 16 CPU threads) - == the oracle's
    brute force (oracle/hwer_oracle.c, row-blocked loop) over the blended rows the step
    produced (ids and scores bit for bit; the blend itself is held to the oracle by ::test_blend_search_step_equals_the_separate_calls and
    test_gpu_round2.py::test_c5_blend_and_search_shard_size).  c4 / c4full at full size: ::test_emulated_shards_equal_single_table."""
    import torch
    from hwer import native
    cfg = dict(bench_mod.CONFIGS[name])
    dev = torch.device("cuda", 0)
    w = bench_mod.Workload(torch, native.lib(), cfg, dev)
    w.step()
    torch.cuda.synchronize()
    idx, score = w.results()[:2]
    U, B = w.U, w.B
    assert idx.shape == (B, bench_mod.K)
    Vh = w.V[:, :w.d].cpu().numpy()
    V64u = Vh[:U].astype(np.float64)
    sel = np.unique(np.linspace(0, B - 1, n_check).astype(np.int64))
    Qh = np.stack([orc.query_vector(V64u, int(q)) for q in sel[:4096]])
    if len(sel) > 4096:      # (plain anchor queries: the oracle's query_vector is unit_length of the row - batched for the 138 493 of c3, the first 4 096 compared with the loop)
        Qb = orc.unit_length(V64u[sel], axis=1)
        assert np.array_equal(Qb[:4096], Qh)
        Qh = Qb
    o_idx, o_score, _ = c_oracle.find_closest_f64(Vh[U:], Qh, V64u[sel], bench_mod.K, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx[sel], o_idx + U)
    assert np.array_equal(score[sel], o_score)


# ------------------------------------------------------------------------------------------------- one query: matrix-vector passes
@pytest.mark.parametrize("I,d,k", [(50000, 40, 200), (50000, 64, 1), (70001, 100, 200), (33000, 128, 50), (20000, 200, 200), (20000, 256, 392)])
def test_one_query_matrix_vector_passes(hw, I, d, k):
    """B = 1 on a catalogue beyond the dense limit (csrc/knn_small.hip: small_seedgemv_kernel + small_gemv16_kernel, every padded width 64 /
    128 / 256, a row count that is not a multiple of the 64-row padding): ids, scores and distances == the oracle for several anchors, one search
    each; two and three anchors at once (the MFMA ring) agree as well."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(I + d)
    U = 5
    V = rng.standard_normal((U + I, d)).astype(np.float32)
    V /= np.linalg.norm(V.astype(np.float64), axis=1, keepdims=True)
    V = V.astype(np.float32)
    ix = DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, U + I)}, arith=0)
    V64 = V.astype(np.float64)
    Qh = np.stack([orc.query_vector(V64, u) for u in range(U)])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(V[U:], Qh, V64[:U], k, threads=c_oracle.effective_cpus())
    for u in range(U):
        idx, score, dist, flags = ix.search("item", np.asarray([u]), k=k)
        assert flags[0] > 0
        assert np.array_equal(idx[0], o_idx[u] + U) and np.array_equal(score[0], o_score[u]) and np.array_equal(dist[0], np.sqrt(o_d2[u]))
    for sel in ([0, 3], [1, 2, 4]):
        idx, score, dist, flags = ix.search("item", np.asarray(sel), k=k)
        assert np.array_equal(idx, o_idx[sel] + U) and np.array_equal(score, o_score[sel])


# ------------------------------------------------------------------------------------------------- float16 fragments out of the prologue
@pytest.mark.parametrize("U,I,d,T,per_row", [(3000, 20000, 128, 2, False), (2500, 30000, 256, 3, True), (4096, 9000, 64, 2, False), (600, 3000, 128, 2, False)])
def test_fused_float16_fragments_equal_the_repacking_launch(hw, bench_mod, monkeypatch, U, I, d, T, per_row):
    """hwer_blend_prepare_search_f32 / hwer_prepare_search_f32 with bq < 0 write the float16 query fragments + per-query scales of the float16 screen;
    the search then runs with HWER_KNN_QUERIES_READY.  Same ids / scores / flags as the step whose search repacks the queries itself
    (HWER_NO_FUSED_FRAGMENTS=1), streaming and dense float16 shapes, d = 64 / 128 / 256."""
    import torch
    from hwer import native
    cfg = dict(U=U, I=I, d=d, T=T, per_row=per_row, name="test")
    dev = torch.device("cuda", 0)
    res = []
    for no_frag in ("", "1"):
        if no_frag:
            monkeypatch.setenv("HWER_NO_FUSED_FRAGMENTS", "1")
        else:
            monkeypatch.delenv("HWER_NO_FUSED_FRAGMENTS", raising=False)
        w = bench_mod.Workload(torch, native.lib(), cfg, dev, f16_dense=True)
        assert w.f16_screen
        assert bool(w.fused_fragments) == (not no_frag and (w.fused_blend or w.fused_prologue))
        w.step()
        torch.cuda.synchronize()
        res.append(w.results()[:2] + (w.sets[0]["flags"].cpu().numpy(),))
        del w
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
