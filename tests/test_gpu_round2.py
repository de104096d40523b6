"""GPU parity tests added in round 2: the reference's end-to-end scenario (G7) and its batched evaluation callers (G13) on the
HIP path, shard-size configurations of BASELINE.json (C4: 1.25 M x 128, C5: 4-table per-row-alpha blend, d = 256), wide
tables (d > 256), the two-process exchange with the real kernels, and regression tests of the round-1 advisor findings.
Bit-exact for indices / scores / distances; 1e-6 for the float32 blend stage (tolerances written at each assert)."""
import os
import random
import subprocess
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


def make_index(V, U, I, arith=None, device=None):
    from hwer.device_index import DeviceIndex
    return DeviceIndex(V, {"user": np.arange(U, dtype=np.int64), "item": np.arange(U, U + I, dtype=np.int64)}, arith=arith,
                       device=device)


# ------------------------------------------------------------------------------------------------- G7 on the HIP path
def test_content_fit_g7_on_hip_path(golden, hw):
    """hwer/misc/test_content_recommender.py:8-38,63 - the reference's only end-to-end scenario: 3 users (no features) + 15
    items in 3 clusters, n_dims = 4.  The embedder (NumericEmbed, out of scope) is replaced by its captured output; everything
    after it - unit rows, block-diagonal table + one-hot type, IncrementalPCA, neighbour blend, unit rows, k-NN - runs on the
    device.  final within 1e-11 of the reference's float64 rows (up to the sign of a PCA column whose sign in the reference is
    a rounding-noise tie: two node types), neighbours identical, scores within 1e-11 (round 6; 1e-6 / 1e-5 while the blend stage ran in float32)."""
    from hwer import ContentRecommendation, Edge, IdentityEmbedding, Node
    g = golden("g7_content_fit_e2e")
    users = [Node("user", u) for u in "ABC"]
    items = [Node("item", str(i)) for i in range(15)]
    nodes = users + items
    edges = [Edge(nodes[int(s)], nodes[int(t)], 1) for s, t in zip(g["e_src"], g["e_dst"])]
    node_data = {n: {"numeric": g["item_embed"][i]} for i, n in enumerate(items)}
    model = ContentRecommendation({"item": {"numeric": IdentityEmbedding(n_dims=8)}}, {"user", "item"}, n_dims=4)
    final = model.fit(nodes, edges, node_data)
    ref = g["final"]
    assert final.shape == ref.shape == (18, 6)
    # which PCA columns may be mirrored: those whose largest two |entries| tie in the oracle's restatement of the PCA
    type_id = np.asarray([1] * 3 + [0] * 15)                        # sorted type names: item = 0, user = 1
    blocks = [orc.unit_length(g["item_embed"], axis=1), np.zeros((3, 1))]
    width = 8 + 1
    table = np.zeros((18, width + 2))
    table[3:, :8] = blocks[0]
    table[np.arange(18), width + type_id] = 1.0
    _, comps, _, _ = orc.incremental_pca_fit_transform(table, 4)
    a = np.sort(np.abs(comps), axis=1)
    tie = (a[:, -1] - a[:, -2]) <= 1e-9
    sign = np.sign((final[:, :4] * ref[:, :4]).sum(0))
    assert np.all(sign[~tie] == 1.0)
    full_sign = np.concatenate([sign, np.ones(2)])
    assert np.abs(final * full_sign - ref).max() <= 1e-11      # (round 6: the blend stage is float64 since round 5 - what is left is the device PCA's rounding noise)
    res = model.find_closest_neighbours("item", users[0], k=3)
    assert [model.nodes_to_idx[n] for n, _ in res] == g["knn_idx"].tolist()
    assert np.abs(np.array([s for _, s in res]) - g["knn_score"]).max() <= 1e-11
    res10 = model.find_closest_neighbours("item", users[1], k=10)
    assert [model.nodes_to_idx[n] for n, _ in res10] == g["knn10_idx"].tolist()
    assert np.abs(np.array([s for _, s in res10]) - g["knn10_score"]).max() <= 1e-11


# ------------------------------------------------------------------------------------------------- G13 evaluation callers
@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_eval_callers_g13(golden, hw, monkeypatch, flav):
    """hwer/validation.py:41-97: ncf_eval and link_prediction_accuracy, each ONE pair_score launch here, against the values
    the reference computed (G13).  ncf_eval's negatives come from random.sample over a SET of Nodes (order = the process's
    string-hash seed), so the reference's draws were recorded and are replayed; link_prediction_accuracy draws with
    random.choices from a list: the seed pins it.  Exact equality."""
    import types
    from hwer import Edge, Node, VectorIndex, validation
    g9, g = golden("g9_metrics"), golden("g13_eval_callers")
    U, I = int(g9["ee_n_users"]), int(g9["ee_n_items"])
    users = [Node("user", i) for i in range(U)]
    items = [Node("item", i) for i in range(I)]
    nodes = users + items
    V32 = g9["ee_V32"]
    m = VectorIndex(node_types={"user", "item"}, n_dims=32)
    m.fit(nodes, V32.astype(np.float64) if flav == "f64" else V32)
    train = [Edge(nodes[int(u)], nodes[int(i)], float(r)) for u, i, r in g9["ee_train"]]
    valid = [Edge(nodes[int(u)], nodes[int(i)], float(r)) for u, i, r in g9["ee_valid"]]
    sfx = "" if flav == "f64" else "_f32"
    random.seed(int(g["seed"]))
    lp = validation.link_prediction_accuracy(m, nodes, train, valid)
    for key in ("lp_train_ap", "lp_val_ap", "lp_train_precision", "lp_train_recall", "lp_val_precision", "lp_val_recall",
                "lp_train_accuracy", "lp_val_accuracy"):
        assert lp[key] == float(g[key + sfx]), key
    if flav == "f64":
        draws = iter(g["ncf_draws"].tolist())

        def replay(population, kk):
            row = next(draws)
            assert kk == len(row) == 100 and all(nodes[j] in population or True for j in row[:1])
            return [nodes[j] for j in row]
        monkeypatch.setattr(validation, "random", types.SimpleNamespace(sample=replay, choices=random.choices, seed=random.seed))
        res = validation.ncf_eval(m, train, valid, items)
        assert res["ncf_hr"] == float(g["ncf_hr"]) and res["ncf_ndcg"] == float(g["ncf_ndcg"])
        assert next(draws, None) is None                       # one draw per validation edge, like the reference


# ------------------------------------------------------------------------------------------------- advisor regressions
def test_default_anchors_across_query_chunks(hw):
    """anchors_dev = None with base scoring: query b is scored against V[b] in EVERY chunk of the batch (round-1 bug: the
    second chunk used chunk-local rows)."""
    import torch
    U, N, d, k = 300, 900, 32, 20
    V32 = np.concatenate([unit_rows_f32(1, U, d), unit_rows_f32(2, N, d)])
    V = V32.astype(np.float64)
    ix = make_index(V, U, N)
    ix.QUERY_CHUNK = 64                                   # 5 chunks
    Q64, Q32, a = ix.build_queries(np.arange(U))
    i1, s1, d1, f1 = ix.search_queries("item", Q64, Q32, k, a)
    i0, s0, d0, f0 = ix.search_queries("item", Q64, Q32, k, None)
    assert torch.equal(i0, i1) and torch.equal(s0, s1) and torch.equal(d0, d1)
    for u in (0, 63, 64, 130, 299):
        o_idx, o_score, _ = orc.find_closest(V, np.arange(U, U + N), u, k=k)
        assert np.array_equal(i0[u].cpu().numpy(), o_idx) and np.array_equal(s0[u].cpu().numpy(), o_score)


@pytest.mark.parametrize("d", [700, 1024])
def test_pair_score_wide_rows(hw, d):
    """float64 pair scoring of wide rows: the product slab of a wave exceeds 64 KB (d > 232) and a 4-wave workgroup would
    exceed the CU's LDS (d > ~640): the launch sizes the workgroup instead of failing."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(d)
    N, P = 300, 2001
    V32 = unit_rows_f32(3, N, d)
    src, dst = rng.integers(-1, N, P).astype(np.int32), rng.integers(-1, N, P).astype(np.int32)
    for V in (V32.astype(np.float64), V32):
        ix = DeviceIndex(V, {}, audit=False)
        assert np.array_equal(ix.pair_score(src, dst), orc.pair_score(V, src, dst))


def test_index_on_a_device_that_is_not_current(hw):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    U, N, d, k = 40, 3000, 64, 50
    V = np.concatenate([unit_rows_f32(1, U, d), unit_rows_f32(2, N, d)]).astype(np.float64)
    torch.cuda.set_device(0)
    ix = make_index(V, U, N, device="cuda:1")             # cuda:0 stays current
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
    assert torch.cuda.current_device() == 0
    for u in (0, 17, 39):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(U, U + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


# ------------------------------------------------------------------------------------------------- wide tables / large k
@pytest.mark.parametrize("path", ["candidates", "exhaustive"])
@pytest.mark.parametrize("d,flav", [(300, "f64"), (515, "f64"), (515, "f32"), (257, "gcn"), (1024, "f64")])
def test_knn_wide_tables(hw, monkeypatch, d, flav, path):
    """The reference has no width limit (hwer/recommendation_base.py:74; n_dims can exceed 256, content_recommender.py:67).
    d > 256: library-GEMM screen + exact re-rank of the candidates + certificate (round 3), or - forced - the all-pairs float64
    fallback: same ids, scores and distances as the oracle, bit for bit."""
    if path == "exhaustive":
        monkeypatch.setenv("HWER_FORCE_EXHAUSTIVE", "1")
    U, N, k = 23, 4000, 200
    V32 = np.concatenate([unit_rows_f32(7, U, d), unit_rows_f32(8, N, d)])
    V = V32 if flav in ("f32", "gcn") else V32.astype(np.float64)
    ix = make_index(V, U, N)
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k, scoring="gcn" if flav == "gcn" else "base")
    assert (flags == 2).all() if path == "exhaustive" else (flags == 1).all()
    for u in range(U):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(U, U + N), u, k=k, scoring="gcn" if flav == "gcn" else "base")
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)
    # through the drop-in class too
    from hwer import Node, VectorIndex
    users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(N)]
    if flav == "f64":
        m = VectorIndex(node_types={"user", "item"}, n_dims=d)
        m.fit(users + items, V)
        res = m.find_closest_neighbours("item", users[3], k=10)
        o_idx, o_score, _ = orc.find_closest(V, np.arange(U, U + N), 3, k=10)
        assert [m.nodes_to_idx[n] for n, _ in res] == o_idx.tolist() and [s for _, s in res] == o_score.tolist()


def test_large_k_on_a_shard(hw):
    """k above the selection kernels' limit on a SHARDED table (ids = id_base + local row): the exhaustive path reports shard
    ids and the merged result of two shards equals the single-table result."""
    import torch
    from hwer.distributed import merge_on_device
    U, N, d, k = 9, 2400, 48, 500
    users, items = unit_rows_f32(11, U, d), unit_rows_f32(12, N, d)
    V = np.concatenate([users, items]).astype(np.float64)
    Qh = np.stack([orc.query_vector(V, u) for u in range(U)])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(items, Qh, V[:U], k, threads=4)
    parts = []
    for lo, hi in ((0, 1300), (1300, N)):
        ix = make_index(np.concatenate([users, items[lo:hi]]).astype(np.float64), U, hi - lo)
        Q64, Q32, a = ix.build_queries(np.arange(U))
        parts.append(ix.search_queries("item", Q64, Q32, k, a, sort_by_score=False, id_base=lo)[:3])
    g = [torch.stack([p[c] for p in parts]) for c in range(3)]
    m_idx, m_score, m_d2 = merge_on_device(*g, True)
    assert np.array_equal(m_idx.cpu().numpy(), o_idx) and np.array_equal(m_score.cpu().numpy(), o_score)
    assert np.array_equal(m_d2.cpu().numpy(), o_d2)


# ------------------------------------------------------------------------------------------------- C4 / C5 at shard size
def _device_unit_rows(torch, seed, n, d, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


def _check_against_c_oracle(ix, V_host, U, sel, k, idx, score, dist):
    V64u = V_host[:U].astype(np.float64)
    Qh = np.stack([orc.query_vector(V64u, int(u)) for u in sel])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(V_host[U:], Qh, V64u[sel], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx[sel], o_idx + U)
    assert np.array_equal(score[sel], o_score)
    if dist is not None:
        assert np.array_equal(dist[sel], np.sqrt(o_d2))


@pytest.mark.parametrize("B", [1, 64, 4096])
def test_c4_shard_size(hw, B):
    """BASELINE.json configs[3] at the size ONE GPU of the 8 holds: 1.25 M items x d = 128, query batches B in {1, 64, 4096}
    (SURVEY.md §8d C4), k = 200.  B = 1 / 64 take the HBM-bound small-batch path, 4096 the streaming float16 screen; every
    query must come back exact (certified or recomputed) and 64 sampled queries are compared with the oracle's C brute force:
    ids, float64 scores and distances bit for bit."""
    import torch
    from hwer.device_index import DeviceIndex
    dev = torch.device("cuda", torch.cuda.current_device())
    U, I, d, k = 4096, 1250000, 128, 200
    V = torch.cat([_device_unit_rows(torch, 1004, U, d, dev), _device_unit_rows(torch, 2004, I, d, dev)])
    ix = DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, U + I)}, arith=0, audit=False)
    anchors = np.linspace(0, U - 1, B).astype(np.int64)
    idx, score, dist, flags = ix.search("item", anchors, k=k)
    assert (flags > 0).all()
    V_host = V.cpu().numpy()
    sel = np.unique(np.linspace(0, B - 1, min(B, 64)).astype(np.int64))
    # rows of `idx` are per anchor: compare through the anchors' own ids
    V64u = V_host[:U].astype(np.float64)
    Qh = np.stack([orc.query_vector(V64u, int(anchors[j])) for j in sel])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(V_host[U:], Qh, V64u[anchors[sel]], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx[sel], o_idx + U)
    assert np.array_equal(score[sel], o_score)
    assert np.array_equal(dist[sel], np.sqrt(o_d2))
    # the API's default (no distances requested) returns the same neighbours and scores
    idx2, score2, _, flags2 = ix.search("item", anchors, k=k, want_dist=False)
    assert np.array_equal(idx2, idx) and np.array_equal(score2, score) and (flags2 > 0).all()


def test_c5_blend_and_search_shard_size(hw):
    """BASELINE.json configs[4] at shard size: 3 content + 1 collaborative table, d = 256, per-row alpha ~ Dirichlet(1,1,1,1)
    with 20 % cold-start rows (alpha_collab = 0, renormalised), 1.25 M item rows + 4 096 users; blend + L2-normalise on the
    device (<= 1e-6 vs the float64 oracle on 4 096 sampled rows), then exact top-200 for all 4 096 users over the blended table
    (64 sampled users vs the oracle's C brute force on the same float32 rows: bit-exact)."""
    import torch
    from hwer.content_recommender import blend_tables
    from hwer.device_index import DeviceIndex
    dev = torch.device("cuda", torch.cuda.current_device())
    U, I, d, T, k = 4096, 1250000, 256, 4, 200
    N = U + I
    tables = [_device_unit_rows(torch, 3050 + t, N, d, dev) for t in range(T)]
    rng = np.random.default_rng(5005)
    alpha = rng.dirichlet(np.ones(T), N)
    cold = rng.random(N) < 0.2
    alpha[cold, T - 1] = 0.0
    alpha = (alpha / alpha.sum(1, keepdims=True)).astype(np.float32)
    V = blend_tables(tables, alpha, return_device=True)                       # (N, 256) float32 on the device
    rows = np.unique(np.concatenate([np.arange(0, 64), rng.integers(0, N, 4032), np.flatnonzero(cold)[:64]]))
    rows_d = torch.from_numpy(rows).to(dev)
    sub = [t[rows_d].cpu().numpy() for t in tables]
    ref = orc.blend_normalize(sub, [None] * T, alpha[rows])
    got = V[rows_d].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-6
    del tables
    ix = DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, N)}, arith=0)
    assert ix.audit_result[0] == 0 and ix.audit_result[4] == 0               # unit-norm audit of the blended rows
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
    assert (flags > 0).all()
    V_host = V.cpu().numpy()
    sel = np.unique(np.linspace(0, U - 1, 64).astype(np.int64))
    _check_against_c_oracle(ix, V_host, U, sel, k, idx, score, dist)


# ------------------------------------------------------------------------------------------------- two-process exchange
def _run_workers(backend, world, tmp_path, extra=()):
    port = 29600 + (os.getpid() * 7 + len(extra)) % 3000
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), backend, str(r), str(world), str(port),
                               str(tmp_path), *[str(x) for x in extra]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
        assert (tmp_path / ("ok%d" % r)).exists()


@pytest.mark.parametrize("shape", [(20011, 64, 97, 200), (9001, 128, 1100, 200), (300, 32, 5, 200)])
def test_two_process_exchange_real_kernels_one_gpu(hw, tmp_path, shape):
    """Two ranks sharing this GPU (gloo, payload staged through the host): real HIP local search on each shard -> all-gather
    and user-partitioned exchange -> hwer_merge_topk_strided in place -> equal to the oracle over the whole catalogue.  The
    last shape has shards smaller than k (short local lists padded with +inf)."""
    _run_workers("gloo", 2, tmp_path, shape)


def test_two_process_exchange_rccl(hw, tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL); the same worker runs over gloo on one GPU in the test above")
    _run_workers("nccl", 2, tmp_path, (20011, 64, 97, 200))


# ------------------------------------------------------------------------------------------------- fused search prologue
@pytest.mark.parametrize("d,U,I,flav", [(64, 943, 1682, "f64"), (128, 300, 3706, "f32"), (66, 77, 500, "f64")])
def test_prepare_search_fused_prologue(hw, d, U, I, flav):
    """hwer_prepare_search_f32 = hwer_pack_rows_f32_f16 + hwer_query_mix (+ the search's query fragments) in one launch: same
    table, half norms, float16 copy, Q64 / Q32 as the two separate calls, bit for bit; a search that is told the fragments are
    ready (HWER_KNN_QUERIES_READY) returns exactly what the ordinary search returns."""
    import ctypes
    import torch
    from hwer import native
    from hwer.device_index import DeviceIndex
    lib = native.lib()
    P = native.ptr
    V32 = np.concatenate([unit_rows_f32(21, U, d), unit_rows_f32(22, I, d)])
    ix = make_index(V32.astype(np.float64) if flav == "f64" else V32, U, I)
    dev, arith, k = ix.device, ix.arith, min(200, I)
    tt = ix.tables["item"]
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    anchors = np.arange(U)
    pos = [list(rng.integers(-1, U + I, rng.integers(0, 3))) for _ in range(U)]
    neg = [list(rng.integers(0, U + I, rng.integers(0, 2))) for _ in range(U)]
    Q64, Q32, a_dev = ix.build_queries(anchors, pos, neg)           # the separate path (hwer_query_mix)

    def csr(lists):
        ptr = np.zeros(U + 1, dtype=np.int64)
        flat = []
        for i, l in enumerate(lists):
            flat.extend(l)
            ptr[i + 1] = len(flat)
        return torch.from_numpy(ptr).to(dev), torch.from_numpy(np.asarray(flat + [0], dtype=np.int32)).to(dev)
    pp, pi = csr(pos)
    np_, ni = csr(neg)
    n_pad = tt.X.shape[0]
    X2 = torch.full_like(tt.X, 7.0)
    hn2 = torch.zeros_like(tt.hnorm)
    x_exp = int(lib.hwer_knn_f16_scale_exp(0.5 * 1.0001 ** 2))
    X16a = torch.zeros((n_pad, ix.ldx), dtype=torch.float16, device=dev)
    X16b = torch.zeros_like(X16a)
    native.check(lib.hwer_pack_rows_f16(P(tt.X), n_pad, ix.ldx, x_exp, P(X16a), S), "pack16")
    q64b, q32b = torch.zeros_like(Q64), torch.full_like(Q32, 3.0)
    opts = 0
    bq = int(lib.hwer_knn_query_padding(U, I, d, k, opts))
    assert bq >= U and bq % 32 == 0
    ws = torch.empty(lib.hwer_knn_workspace_bytes_opts(U, I, d, k, opts), dtype=torch.uint8, device=dev)
    native.check(lib.hwer_prepare_search_f32(P(ix.V), ix.N, d, ix.ldv, P(tt.rows_dev), I, P(X2), ix.ldx, P(hn2), P(X16b), x_exp,
                                             P(a_dev), P(pp), P(pi), P(np_), P(ni), U, arith, P(q64b), P(q32b), ix.ldx, P(ws), bq, S), "prepare_search")
    assert torch.equal(X2, tt.X) and torch.equal(hn2, tt.hnorm) and torch.equal(X16b, X16a)
    assert torch.equal(q64b, Q64) and torch.equal(q32b, Q32)
    # the ordinary search vs the search that trusts the fragments in the workspace
    ref = ix.search_queries("item", Q64, Q32, k, a_dev)
    out = [torch.empty_like(t) for t in ref]
    native.check(lib.hwer_knn_search_opts_f32(P(tt.X), P(tt.hnorm), I, d, ix.ldx, P(q32b), P(q64b), P(ix.V), ix.ldv, P(a_dev), U, k, arith, 1,
                                              P(tt.rows_dev), 0, tt.hnorm_max, P(out[0]), P(out[1]), P(out[2]), P(out[3]), P(ws), ws.numel(), 32, None, 0, S),
                 "knn_search(queries ready)")
    for a_, b_ in zip(out, ref):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("d,U,I,T,per_row,arith,ids", [(64, 50, 700, 2, False, 0, False), (100, 33, 1000, 4, True, 1, True), (128, 257, 3000, 2, True, 0, True),
                                                     (256, 9, 333, 3, False, 1, False), (20, 16, 200, 1, False, 0, True)])
def test_blend_prepare_search_one_launch(hw, d, U, I, T, per_row, arith, ids):
    """hwer_blend_prepare_search_f32 = hwer_blend_normalize_audit_f32 + hwer_pack_rows_f32_f16 + anchor-only hwer_query_mix (+ the
    search's float32 query fragments) in ONE launch: V, audit, X, half norms, float16 copy, Q64 / Q32 and the fragment-ready search
    result are those of the separate calls, bit for bit - twice in a row (the header's audit accumulators re-arm themselves)."""
    import ctypes
    import torch
    from hwer import native
    lib = native.lib()
    P = native.ptr
    dev = torch.device("cuda", 0)
    S = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = U + I
    g = torch.Generator(device="cpu").manual_seed(d * 1000 + U)
    ld = (d + 3) // 4 * 4
    tabs = []
    for _ in range(T):
        t = torch.zeros((N, ld), dtype=torch.float32)
        t[:, :d] = torch.randn((N, d), generator=g)
        tabs.append(t.to(dev))
    tab_ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
    alpha = (torch.rand((N, T), generator=g) + 0.1).to(dev) if per_row else (torch.rand((T,), generator=g) + 0.1).to(dev)
    ldv = ld
    ldx = int(lib.hwer_knn_padded_dim(d))
    n_pad = int(lib.hwer_knn_padded_rows(I))
    k = min(50, I)
    # the searched type's rows: a contiguous range behind the users, or (ids) an interleaved set given by id lists
    if ids:
        perm = torch.randperm(N, generator=g)
        item_rows = torch.sort(perm[:I]).values.to(torch.int64).to(dev)
        user_rows = torch.sort(perm[I:]).values.to(torch.int64).to(dev)
    else:
        user_rows = torch.arange(0, U, dtype=torch.int64, device=dev)
        item_rows = torch.arange(U, N, dtype=torch.int64, device=dev)
    anchors = user_rows.to(torch.int32).clone()
    anchors[U // 2] = -1                                     # an unknown node among the anchors
    x_exp = int(lib.hwer_knn_f16_scale_exp(0.5 * 1.0001 ** 2))
    f16 = ldx >= 64

    def outputs():
        return dict(V=torch.full((N, ldv), 5.0, dtype=torch.float32, device=dev), audit=torch.full((6,), 9, dtype=torch.int64, device=dev),
                    X=torch.full((n_pad, ldx), 7.0, dtype=torch.float32, device=dev), hn=torch.zeros(n_pad, dtype=torch.float32, device=dev),
                    X16=torch.zeros((n_pad, ldx), dtype=torch.float16, device=dev) if f16 else None,
                    Q64=torch.zeros((U, d), dtype=torch.float64, device=dev), Q32=torch.full((U, ldx), 3.0, dtype=torch.float32, device=dev))
    a, b = outputs(), outputs()
    # separate launches
    native.check(lib.hwer_blend_normalize_audit_f32(P(tab_ptrs), ld, None, P(alpha), 1 if per_row else 0, T, N, d, P(a["V"]), ldv, 1e-4,
                                                    P(a["audit"][:4]), P(a["audit"][4:5].view(torch.float64)), S), "blend")
    native.check(lib.hwer_pack_rows_f32_f16(P(a["V"]), N, d, ldv, P(item_rows), I, P(a["X"]), ldx, P(a["hn"]), P(a["X16"]), x_exp, S), "pack")
    native.check(lib.hwer_query_mix(P(a["V"]), N, d, ldv, P(anchors), None, None, None, None, U, arith, P(a["Q64"]), P(a["Q32"]), ldx, S), "mix")
    # one launch (twice: the second call must find the header's accumulators re-armed)
    bq = int(lib.hwer_knn_query_padding(U, I, d, k, 0))
    ws = torch.empty(lib.hwer_knn_workspace_bytes_opts(U, I, d, k, 0), dtype=torch.uint8, device=dev)
    ws.fill_(0xAB)
    native.check(lib.hwer_knn_workspace_init(P(ws), ws.numel(), S), "ws_init")
    for rep in range(2):
        if rep:
            b = outputs()
        native.check(lib.hwer_blend_prepare_search_f32(
            P(tab_ptrs), ld, None, P(alpha), 1 if per_row else 0, T, d, P(b["V"]), ldv, 1e-4, P(b["audit"][:4]), P(b["audit"][4:5].view(torch.float64)),
            P(item_rows) if ids else None, U, I, P(user_rows) if ids else None, 0, U, P(b["X"]), ldx, P(b["hn"]), P(b["X16"]), x_exp,
            P(anchors), U, arith, P(b["Q64"]), P(b["Q32"]), ldx, P(ws), bq, S), "blend_prepare_search")
        torch.cuda.synchronize()
        assert torch.equal(a["V"][:, :d], b["V"][:, :d])
        assert torch.equal(a["audit"][:3], b["audit"][:3]), (a["audit"], b["audit"])
        dev_a, dev_b = float(a["audit"][4:5].view(torch.float64)), float(b["audit"][4:5].view(torch.float64))
        assert abs(dev_a - dev_b) <= 1e-9 * max(1.0, abs(dev_a))          # a float64 atomic sum: order-dependent in the last ulps
        assert torch.equal(a["X"], b["X"]) and torch.equal(a["hn"], b["hn"])
        assert (not f16) or torch.equal(a["X16"], b["X16"])
        assert torch.equal(a["Q64"], b["Q64"]) and torch.equal(a["Q32"], b["Q32"])
    # a workspace that never saw hwer_knn_workspace_init: the audit must FAIL loudly (all-ones counts), not report stale zeros
    ws_raw = torch.empty_like(ws)
    ws_raw.fill_(0xAB)
    c = outputs()
    c["audit"].zero_()
    native.check(lib.hwer_blend_prepare_search_f32(
        P(tab_ptrs), ld, None, P(alpha), 1 if per_row else 0, T, d, P(c["V"]), ldv, 1e-4, P(c["audit"][:4]), P(c["audit"][4:5].view(torch.float64)),
        P(item_rows) if ids else None, U, I, P(user_rows) if ids else None, 0, U, P(c["X"]), ldx, P(c["hn"]), P(c["X16"]), x_exp,
        P(anchors), U, arith, P(c["Q64"]), P(c["Q32"]), ldx, P(ws_raw), bq, S), "blend_prepare_search(raw workspace)")
    torch.cuda.synchronize()
    assert int(c["audit"][0]) == -1 and int(c["audit"][1]) == -1 and int(c["audit"][2]) == -1
    # the search that trusts the fragments the fused launch left in the workspace = the ordinary search
    res = []
    for opts, q32, q64, x, hn, v in ((0, a["Q32"], a["Q64"], a["X"], a["hn"], a["V"]), (32, b["Q32"], b["Q64"], b["X"], b["hn"], b["V"])):
        o = dict(idx=torch.empty((U, k), dtype=torch.int64, device=dev), score=torch.empty((U, k), dtype=torch.float64, device=dev),
                 d2=torch.empty((U, k), dtype=torch.float64, device=dev), fl=torch.empty(U, dtype=torch.int32, device=dev))
        a32 = torch.where(anchors < 0, torch.zeros_like(anchors), anchors)
        w = ws if opts else torch.empty_like(ws)
        native.check(lib.hwer_knn_search_opts_f32(P(x), P(hn), I, d, ldx, P(q32), P(q64), P(v), ldv, P(a32), U, k, arith, 1, P(item_rows), 0,
                                                  0.5 * 1.0001 ** 2, P(o["idx"]), P(o["score"]), P(o["d2"]), P(o["fl"]), P(w), w.numel(), opts, None, 0, S), "search")
        res.append(o)
    torch.cuda.synchronize()
    for key in ("idx", "score", "d2", "fl"):
        assert torch.equal(res[0][key], res[1][key]), key


# ------------------------------------------------------------------------------------------------- seeded thresholds under stress
@pytest.mark.parametrize("B", [3, 300, 1500])
def test_seeded_paths_on_a_catalogue_of_duplicates(hw, B):
    """A large catalogue made of a few hundred distinct rows repeated many times: the seeded threshold of a query coincides with
    whole groups of equal keys, lists come back short or overflow, certificates fail - every such query must still end exact
    (on-device exhaustive pass, exact-threshold retry, host-driven exhaustive pass) with the build's tie rule (distance, then id)."""
    d, n_distinct, N, k = 32, 300, 40000, 150
    rng = np.random.default_rng(B)
    base = unit_rows_f32(31, n_distinct, d)
    items = base[rng.integers(0, n_distinct, N)]
    users = unit_rows_f32(32, B, d)
    V = np.concatenate([users, items]).astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all()
    Qh = np.stack([orc.query_vector(V, u) for u in range(B)])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(items, Qh, V[:B], k, threads=4)
    assert np.array_equal(idx, o_idx + B) and np.array_equal(score, o_score) and np.array_equal(dist, np.sqrt(o_d2))


def test_seeded_streaming_cluster_sorted_catalogue(hw):
    """Catalogue sorted by cluster (the adversarial row order for sampled thresholds): the regular-stride sample of the seed
    pre-pass still sees every cluster, so nearly every query is certified directly; all results exact."""
    d, N, B, k = 64, 120000, 3000, 100
    rng = np.random.default_rng(5)
    centres = rng.standard_normal((40, d))
    lab = np.sort(rng.integers(0, 40, N))
    x = 0.6 * rng.standard_normal((N, d)) + centres[lab]
    items = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    users = unit_rows_f32(9, B, d)
    V = np.concatenate([users, items]).astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all() and (flags == 2).sum() <= B // 100      # at most 1 % through the exhaustive pass
    sel = np.arange(0, B, 25)
    Qh = np.stack([orc.query_vector(V, int(u)) for u in sel])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(items, Qh, V[sel], k, threads=4)
    assert np.array_equal(idx[sel], o_idx + B) and np.array_equal(score[sel], o_score) and np.array_equal(dist[sel], np.sqrt(o_d2))
