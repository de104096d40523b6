"""GPU parity tests added in round 6: the FUSED search on genuine float64 rows (hwer_knn_search_f64rows / hwer_knn_select_f64rows /
hwer_knn_exact_device_f64rows / hwer_knn_exact_f64rows, csrc/knn_select_f64.hip) - what ContentRecommendation serves since round 5's float64
blend stage (reference: hwer/recommendation_base.py:74,78-83,141-143,171-173 on the float64 `vectors` of hwer/content_recommender.py:96).
Every path the float32 flavour has (dense one-query-per-workgroup and wave-per-query selects, small-batch list select, streaming select,
on-device and host-driven exhaustive tiers) against the numpy oracle on the float64 rows, and against rounds 3-5's candidate scheme on every
query.  Bit-exact unless an assert states a tolerance."""
import os

import numpy as np
import pytest

from oracle import hwer_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hw():
    import hwer
    from hwer import native
    native.lib()
    return hwer


def _rows64(seed, n, d, scale_spread=False):
    x = np.random.default_rng(seed).standard_normal((n, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    if scale_spread:      # row norms within the unit audit's 1e-4
        x *= (1.0 + 8e-5 * np.random.default_rng(seed + 1).uniform(-1, 1, (n, 1)))
    return x


def _plant(X, B):
    """float32-colliding rows and exact duplicates next to each other: only the float64 distances / the id order separate them"""
    X[B + 11] = X[B + 10] * (1.0 + 3e-9)
    X[B + 12] = X[B + 10] * (1.0 - 2e-9)
    X[B + 13] = X[B + 10]
    return X


# (B, N, d, k): dense COOP select / dense wave-per-query / small-batch list select (COOP) / streaming select, d = 128 and 256, odd widths
SHAPES = [(40, 1682, 64, 200), (3000, 3706, 128, 200), (64, 20000, 96, 200), (1, 20000, 128, 50), (2, 9000, 66, 10), (2500, 20000, 128, 200),
          (1500, 12000, 256, 100), (300, 2000, 12, 392), (5, 300, 7, 300), (1100, 8193, 66, 64)]


@pytest.mark.parametrize("B,N,d,k", SHAPES)
@pytest.mark.parametrize("scoring", ["base", "gcn"])
def test_fused_float64_row_search_vs_oracle_and_candidate_scheme(hw, B, N, d, k, scoring):
    from hwer.device_index import DeviceIndex
    if scoring == "gcn" and (B, N) not in ((40, 1682), (64, 20000), (2500, 20000), (5, 300)):
        pytest.skip("the GcnNCF score flavour on one shape per path")
    X = _plant(_rows64(600 + d + B, B + N, d, scale_spread=True), B)
    ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    assert ix.V64 is not None and ix.ld64 % 8 == 0
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k, scoring=scoring)
    assert (flags > 0).all()
    # the lazy-distance variant of the same search (what find_closest_neighbours_batch asks for)
    idx_l, score_l, dist_l, flags_l = ix.search("item", np.arange(B), k=k, scoring=scoring, want_dist=False)
    assert np.array_equal(idx, idx_l) and np.array_equal(score, score_l) and (flags_l > 0).all()
    rows = np.arange(B, B + N)
    for u in sorted(set(list(range(0, B, max(1, B // 10))) + [B - 1])):
        o_idx, o_score, o_dist = orc.find_closest(X, rows, u, k=k, scoring=scoring)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist), u
    # every query against the candidate scheme of rounds 3-5 (an independent composition of other kernels)
    os.environ["HWER_F64ROWS_CANDIDATES"] = "1"
    try:
        idx_c, score_c, dist_c, flags_c = ix.search("item", np.arange(B), k=k, scoring=scoring)
    finally:
        del os.environ["HWER_F64ROWS_CANDIDATES"]
    assert np.array_equal(idx, idx_c) and np.array_equal(score, score_c) and np.array_equal(dist, dist_c)


def test_fused_float64_rows_query_at_the_planted_ties(hw):
    """anchors that ARE the planted rows' neighbours: the float32 screen cannot separate rows 10 .. 13 (equal after rounding), the float64
    re-rank must, in (distance, id) order, on the dense, list and streaming selects"""
    from hwer.device_index import DeviceIndex
    for B, N, d, k in [(8, 3000, 64, 4), (8, 20000, 64, 4), (1500, 20000, 64, 3)]:
        X = _rows64(77, B + N, d)
        X[0] = X[B + 10]                       # anchor 0 sits on the planted group
        X = _plant(X, B)
        ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
        o_idx, o_score, o_dist = orc.find_closest(X, np.arange(B, B + N), 0, k=k)
        assert np.array_equal(idx[0], o_idx) and np.array_equal(score[0], o_score) and np.array_equal(dist[0], o_dist), (B, N)
        assert set(idx[0].tolist()) >= {B + 10, B + 13} and flags[0] > 0


def test_float64_row_exhaustive_tiers(hw):
    """many exact duplicates at the cut: the certificate fails (flag 0) and the on-device / host-driven exhaustive passes - reading the
    float64 rows - answer; > 16 uncertified queries also go through the exact-threshold retry tier"""
    from hwer.device_index import DeviceIndex
    B, N, d, k = 40, 12000, 32, 20
    X = _rows64(5, B + N, d)
    X[B + 100:B + 700] = X[B + 50]            # 600 copies of one row: more near-ties than the re-rank set holds
    for b in range(B):
        X[b] = X[B + 50] * (1 - 1e-7 * (b + 1)) + 1e-4 * X[B + 1000 + b]
        X[b] /= np.linalg.norm(X[b])
    ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all() and (flags == 2).any()
    for u in (0, 7, B - 1):
        o_idx, o_score, o_dist = orc.find_closest(X, np.arange(B, B + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist), u


def test_fused_float64_rows_positive_negative_lists(hw):
    """queries that are NOT rows (hwer/recommendation_base.py:164-170: mean of anchor / positive / negative unit vectors, not re-normalised):
    candidates by distance to q, scores against the anchor row - on float64 rows"""
    from hwer.device_index import DeviceIndex
    B, N, d, k = 24, 5000, 48, 30
    X = _rows64(31, B + N, d)
    rng = np.random.default_rng(2)
    pos = [list(rng.integers(0, B + N, rng.integers(0, 4))) for _ in range(B)]
    neg = [list(rng.integers(-1, B + N, rng.integers(0, 3))) for _ in range(B)]
    ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k, positive=pos, negative=neg)
    assert (flags > 0).all()
    for u in range(B):
        o_idx, o_score, o_dist = orc.find_closest(X, np.arange(B, B + N), u, positive=pos[u], negative=neg[u], k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist), u


def test_pack_rows_f64rows_kernel(hw):
    """hwer_pack_rows_f64rows: the float32 rounding of listed float64 rows, half norms, float16 copy, pad rows, and the unit audit of the
    float64 rows (hwer/recommendation_base.py:106-107)"""
    import torch
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8)
    for N, d, n_rows in [(500, 66, 301), (310, 128, 300), (100, 7, 64)]:
        ld64 = (d + 7) // 8 * 8
        V = np.zeros((N, ld64))
        V[:, :d] = _rows64(9 + d, N, d)
        V[3, :d] *= 1.001
        V[4, :d] *= 0.998
        V[5, 0] = np.nan
        rows = np.concatenate([[3, 4, 5], 6 + np.sort(rng.choice(N - 6, n_rows - 3, replace=False))]).astype(np.int64)
        ldx = lib.hwer_knn_padded_dim(d)
        n_pad = lib.hwer_knn_padded_rows(n_rows)
        Vd, rd = torch.from_numpy(V).to(dev), torch.from_numpy(rows).to(dev)
        X = torch.full((n_pad, ldx), 7.0, dtype=torch.float32, device=dev)
        hn = torch.empty(n_pad, dtype=torch.float32, device=dev)
        X16 = torch.empty((n_pad, ldx), dtype=torch.float16, device=dev)
        counts = torch.zeros(4, dtype=torch.int64, device=dev)
        dev_sum = torch.zeros(1, dtype=torch.float64, device=dev)
        x_exp = int(lib.hwer_knn_f16_scale_exp(0.51))
        native.check(lib.hwer_pack_rows_f64rows(native.ptr(Vd), N, d, ld64, native.ptr(rd), n_rows, native.ptr(X), ldx, native.ptr(hn), native.ptr(X16),
                                                x_exp, 1e-4, native.ptr(counts), native.ptr(dev_sum), native.stream_ptr(dev)), "pack_rows_f64rows")
        Xh, hh, X16h = X.cpu().numpy(), hn.cpu().numpy(), X16.cpu().numpy()
        want = np.zeros((n_pad, ldx), np.float32)
        want[:n_rows, :d] = V[rows, :d].astype(np.float32)
        assert np.array_equal(Xh, want, equal_nan=True)
        hw_ = (0.5 * (want[:n_rows].astype(np.float64) ** 2).sum(1))
        ok = ~np.isnan(hw_)
        assert np.allclose(hh[:n_rows][ok], hw_[ok], rtol=2e-7, atol=0) and np.isinf(hh[n_rows:]).all()
        assert np.array_equal(X16h[:n_rows][ok], np.ldexp(want[:n_rows][ok], x_exp).astype(np.float16))
        c = counts.cpu().numpy()
        assert (c[0], c[1], c[2]) == (1, 1, 1)


# ------------------------------------------------------------------------------------------------- the N > 1 pipeline under allocator churn
def _run_pipe_workers(backend, world, tmp_path, exchange, steps):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29700 + (os.getpid() * 13 + world + len(exchange)) % 2500
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "pipe_stress_worker.py"), backend, str(r), str(world), str(port), str(tmp_path), exchange,
                               str(steps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
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


@pytest.mark.parametrize("backend,world,exchange", [("nccl", 1, "both"), ("gloo", 2, "both")])
def test_sharded_step_pipeline_under_allocator_churn(hw, tmp_path, backend, world, exchange):
    """hwer.distributed.ShardedStepPipeline (what `bench.py --gpus N` runs) owns every buffer its side stream touches (round 6: allocated once
    per result set in __init__; round 5 allocated the receive buffer and the merge outputs on the side stream inside run() without
    record_stream).  200 steps (50 over gloo) with the returned tensors dropped at once and allocator churn on the compute stream: every step's merged result
    == the single table's.  RCCL at world size 1 and two ranks over gloo sharing the GPU, both exchanges."""
    _run_pipe_workers(backend, world, tmp_path, exchange, 200 if backend == "nccl" else 50)      # (gloo stages every exchange through the host: fewer steps)


def test_blend_tables_explicit_dtype_and_mixed_warning(hw):
    """blend_tables(dtype=...): the arithmetic is the caller's choice (advisor, round 5); float32 and float64 tables mixed without it run the
    float32 kernel and say so"""
    from hwer.content_recommender import blend_tables
    rng = np.random.default_rng(3)
    a = rng.standard_normal((300, 24))
    b = rng.standard_normal((300, 24)).astype(np.float32)
    al = np.array([0.25, 0.75])
    with pytest.warns(RuntimeWarning, match="mixed"):
        mixed = blend_tables([a, b], al)
    assert mixed.dtype == np.float32
    out64 = blend_tables([a, b], al, dtype="float64")
    acc = np.zeros_like(a)
    for w_, t in zip(al, (a, b.astype(np.float64))):
        acc = acc + w_ * t
    assert out64.dtype == np.float64 and np.array_equal(out64, acc / np.expand_dims(np.linalg.norm(acc, axis=1), 1))
    out32 = blend_tables([a, a], al, dtype="float32")
    assert out32.dtype == np.float32
    with pytest.raises(ValueError):
        blend_tables([a, a], al, dtype="float16")


def test_persisted_model_keeps_genuine_float64_rows(hw, tmp_path):
    """hwer.persist: a model that serves genuine float64 rows - what ContentRecommendation.fit returns - comes back from disk
    with those rows (format hwer-index/2), not with their float32 rounding (rounds 3-5): same neighbours, scores and predict values bit for bit,
    float32-colliding rows at an anchor's cut included."""
    from hwer import Node, VectorIndex, persist
    U, I, d = 30, 2500, 40
    X = _rows64(91, U + I, d)
    X[0] = X[U + 10]
    X = _plant(X, U)
    nodes = [Node("user", i) for i in range(U)] + [Node("item", "i%d" % i) for i in range(I)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    m.fit(nodes, X)
    assert m.knn.index.V64 is not None
    path = str(tmp_path / "m64.npz")
    persist.save_model(path, m)
    assert persist.load_index(path)["vectors64"] is not None
    m2 = persist.restore_model(path)
    assert m2.knn.index.V64 is not None and np.array_equal(m2.knn.index.V64[:, :d].cpu().numpy(), X)
    for u in (0, 7, 29):
        a = m.find_closest_neighbours("item", nodes[u], k=60)
        b = m2.find_closest_neighbours("item", nodes[u], k=60)
        assert [n for n, _ in a] == [n for n, _ in b] and [s_ for _, s_ in a] == [s_ for _, s_ in b]
        o_idx, o_sc, _ = orc.find_closest(X, np.arange(U, U + I), u, k=60)
        assert [m2.nodes_to_idx[n] for n, _ in b] == o_idx.tolist() and [s_ for _, s_ in b] == o_sc.tolist()
    pairs = [(nodes[1], nodes[50]), (nodes[2], nodes[99])]
    assert np.array_equal(m.predict(pairs), m2.predict(pairs))


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_sharded_search_on_genuine_float64_rows(hw, tmp_path, backend, world):
    """the N > 1 search over GENUINE float64 rows: every rank's local lists come from the fused float64-row search on its shard (ids = id_base + local
    row, the whole float64 table addressed through the shard's row list), one exchange, hwer_merge_topk_lists - ids, scores, distances == the numpy oracle
    on the float64 rows over the whole catalogue; all-gather, user-partitioned exchange and shortened lists (tests/dist_worker.py, "f64rows")"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    from test_gpu_round2 import _run_workers
    _run_workers(backend, world, tmp_path, (6007, 48, 33, 100, "f64rows"))
    assert "float64 rows" in (tmp_path / "ok0").read_text()


@pytest.mark.parametrize("env", ["HWER_NO_DENSE", "HWER_NO_SMALL", "HWER_NO_PINGPONG", "HWER_F16_OFF"])
def test_fused_float64_rows_on_every_forced_path(hw, env, monkeypatch):
    """the path-forcing switches of the float32 flavour (results are identical on every path) applied to the fused float64-row search: a dense-size
    catalogue through the streaming screen, a small batch through the streaming screen instead of the list path, the two-workgroup float16 screen, and
    the float32 screen instead of the float16 one - ids, scores, distances == the default path's and the oracle's"""
    from hwer.device_index import DeviceIndex
    cases = {"HWER_NO_DENSE": (300, 3000, 64, 50), "HWER_NO_SMALL": (64, 20000, 96, 100), "HWER_NO_PINGPONG": (2100, 20000, 128, 200), "HWER_F16_OFF": (2100, 20000, 128, 200)}
    B, N, d, k = cases[env]
    X = _plant(_rows64(40 + d, B + N, d), B)
    rows = {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}
    ref = DeviceIndex(X, rows).search("item", np.arange(B), k=k)
    ix = DeviceIndex(X, rows)
    if env == "HWER_F16_OFF":
        ix.f16_screen = False
        ix.use_f16_storage = False
    else:
        monkeypatch.setenv(env, "1")
    got = ix.search("item", np.arange(B), k=k)
    assert (got[3] > 0).all()
    for a, b in zip(ref[:3], got[:3]):
        assert np.array_equal(a, b)
    o_idx, o_score, o_dist = orc.find_closest(X, np.arange(B, B + N), B - 1, k=k)
    assert np.array_equal(got[0][B - 1], o_idx) and np.array_equal(got[1][B - 1], o_score) and np.array_equal(got[2][B - 1], o_dist)


@pytest.mark.parametrize("d", [64, 66, 7, 128])
def test_anchor_query_kernel_on_float64_rows(hw, d):
    """hwer_query_mix_f64rows with >= 4 096 plain anchors takes the row-streaming kernel (anchor_query_kernel<double, double>): Q64 == numpy's
    unit(row) (hwer/recommendation_base.py:153-155,164-170 with only the anchor), Q32 == its float32 rounding zero-padded, unknown anchors (-1) == the
    clipped row 0 - and == the general kernel's output on a batch small enough to take that one"""
    import torch
    from hwer import native
    lib, P = native.lib(), native.ptr
    dev = torch.device("cuda:0")
    N, B = 3000, 5000
    ld64 = (d + 7) // 8 * 8
    V = np.zeros((N, ld64))
    V[:, :d] = _rows64(300 + d, N, d) * (1.0 + 5e-5 * np.random.default_rng(d).uniform(-1, 1, (N, 1)))
    an = np.random.default_rng(1).integers(-1, N, B).astype(np.int32)
    ldx = lib.hwer_knn_padded_dim(d)
    Vd, ad = torch.from_numpy(V).to(dev), torch.from_numpy(an).to(dev)

    def run(n):
        Q64 = torch.full((n, d), 7.0, dtype=torch.float64, device=dev)
        Q32 = torch.full((n, ldx), 7.0, dtype=torch.float32, device=dev)
        native.check(lib.hwer_query_mix_f64rows(P(Vd), N, d, ld64, P(ad), None, None, None, None, n, P(Q64), P(Q32), ldx, native.stream_ptr(dev)), "query_mix_f64rows")
        return Q64.cpu().numpy(), Q32.cpu().numpy()
    q64, q32 = run(B)              # the streaming kernel
    g64, g32 = run(1000)           # the general kernel (fewer than 4 096 anchors)
    assert np.array_equal(q64[:1000], g64) and np.array_equal(q32[:1000], g32)
    for b in list(range(0, B, 397)) + [B - 1]:
        row = V[an[b], :d] if an[b] >= 0 else np.clip(V[0, :d], 1e-6, 1e-5)
        want = (row / np.linalg.norm(row, axis=0)) / 1.0        # (axis given: numpy reduces pairwise, as utils.unit_length does - without it a 1-D norm is a BLAS dot)
        assert np.array_equal(q64[b], want), b
    want32 = np.zeros((B, ldx), np.float32)
    want32[:, :d] = q64.astype(np.float32)
    assert np.array_equal(q32, want32)


def test_audit_pack_rows_f64rows_one_launch(hw):
    """hwer_audit_pack_rows_f64rows == the two hwer_pack_rows_f64rows calls it replaces (audit of the other rows + audit and rounding of the searched range)"""
    import torch
    from hwer import native
    lib, P = native.lib(), native.ptr
    dev = torch.device("cuda:0")
    for N, d, row0, n in [(700, 66, 200, 301), (400, 128, 0, 400), (90, 7, 26, 64)]:
        ld64 = (d + 7) // 8 * 8
        V = np.zeros((N, ld64))
        V[:, :d] = _rows64(11 + d, N, d)
        V[3, :d] *= 1.001
        V[row0 + 1, :d] *= 0.998
        V[N - 1, 0] = np.nan
        Vd = torch.from_numpy(V).to(dev)
        ldx, n_pad = lib.hwer_knn_padded_dim(d), lib.hwer_knn_padded_rows(n)
        x_exp = int(lib.hwer_knn_f16_scale_exp(0.51))
        outs = []
        for fused in (False, True):
            X = torch.full((n_pad, ldx), 7.0, dtype=torch.float32, device=dev)
            hn = torch.full((n_pad,), 7.0, dtype=torch.float32, device=dev)
            X16 = torch.full((n_pad, ldx), 7.0, dtype=torch.float16, device=dev)
            cnt = torch.zeros(4, dtype=torch.int64, device=dev)
            dv = torch.zeros(1, dtype=torch.float64, device=dev)
            st = native.stream_ptr(dev)
            if fused:
                native.check(lib.hwer_audit_pack_rows_f64rows(P(Vd), N, d, ld64, row0, n, P(X), ldx, P(hn), P(X16), x_exp, 1e-4, P(cnt), P(dv), st), "audit_pack")
            else:
                for lo, hi in ((0, row0), (row0 + n, N)):
                    if hi > lo:
                        native.check(lib.hwer_pack_rows_f64rows(P(Vd[lo:hi]), hi - lo, d, ld64, None, hi - lo, None, 0, None, None, 0, 1e-4, P(cnt), P(dv), st), "audit")
                rows = torch.arange(row0, row0 + n, dtype=torch.int64, device=dev)
                native.check(lib.hwer_pack_rows_f64rows(P(Vd), N, d, ld64, P(rows), n, P(X), ldx, P(hn), P(X16), x_exp, 1e-4, P(cnt), P(dv), st), "pack")
            outs.append((X.cpu().numpy(), hn.cpu().numpy(), X16.cpu().numpy(), cnt.cpu().numpy()))
        for a, b in zip(*outs):
            assert np.array_equal(a, b, equal_nan=True)
        assert tuple(outs[1][3][:3]) == (1, 1, 1)


@pytest.mark.parametrize("U,I,d,T,graph", [(300, 3000, 64, 2, True), (2100, 20000, 128, 2, False), (40, 9000, 66, 3, False)])
def test_blend_search_step_f64(hw, U, I, d, T, graph, monkeypatch):
    """hwer.pipeline.BlendSearchStepF64 - the resident step in the reference's float64 flavour (what bench.py's c1_f64rows / c3_f64rows time): float64 tables
    -> blended float64 rows == the numpy expression -> audit -> fused float64-row search == the numpy oracle on those rows; the query launch that also writes
    the search's MFMA fragments (hwer_query_mix_frag_f64rows + HWER_KNN_QUERIES_READY; float32 fragments on the dense path, float16 ones under the float16
    screen, the small-batch path ignores them) gives the same results as the search's own repacking; a captured hipGraph replays the same step"""
    import torch
    from hwer.pipeline import BlendSearchStepF64
    dev = torch.device("cuda:0")
    N = U + I
    ld = (d + 1) // 2 * 2
    tabs_h = [np.zeros((N, ld)) for _ in range(T)]
    for t in range(T):
        tabs_h[t][:, :d] = _rows64(700 + 10 * t + d, N, d)
    al_h = np.random.default_rng(9).dirichlet(np.ones(T))
    tabs = [torch.from_numpy(x).to(dev)[:, :d] if ld != d else torch.from_numpy(x).to(dev) for x in tabs_h]
    alpha = torch.from_numpy(al_h).to(dev)
    anchors = torch.arange(U, dtype=torch.int32, device=dev)
    res = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("HWER_NO_FUSED_FRAGMENTS", "1")
        step = BlendSearchStepF64(tabs, alpha, search_row0=U, n_search=I, anchors=anchors, k=50, want_dist=True)
        assert step.fused_fragments == fused
        if graph and fused:
            step.capture()
        step.run()
        res[fused] = step.results()
        if fused:
            V = step.V64[:, :d].cpu().numpy()
            acc = np.zeros((N, d))
            for w_, x in zip(al_h, tabs_h):
                acc = acc + w_ * x[:, :d]
            assert np.array_equal(V, acc / np.expand_dims(np.linalg.norm(acc, axis=1), 1))
            assert step.audit_counts.tolist()[:3] == [0, 0, 0]
            for u in (0, U // 2, U - 1):
                o_idx, o_sc, o_dist = orc.find_closest(V, np.arange(U, N), u, k=50)
                assert np.array_equal(res[True][0][u], o_idx) and np.array_equal(res[True][1][u], o_sc) and np.array_equal(res[True][2][u], o_dist)
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b)



# Round 6 (second half): the re-rank rounds stage the candidates' rows by LDS-DMA into a lane-permuted image (csrc/knn_select_kernels.h: rr_reader_lane /
# rr_reader_gm) and their instruction stream was rebuilt (leaf hand-over as a branch, chains from zero, shared widening).  Shapes that walk the edges of that
# code: candidate counts around the 64-slot rounds (k = 50 ... 392), widths that are not a multiple of the 16-column chunk or end inside an 8-group (d = 7,
# 17, 100), two pairwise leaves (d = 200, 256), the one-query-per-workgroup, wave-per-query, list and streaming selects - EVERY query against the C oracle.
@pytest.mark.parametrize("B,N,d,k,arith", [(97, 1500, 64, 50, 0), (33, 1200, 17, 120, 0), (2100, 5000, 100, 200, 0), (3000, 9000, 7, 64, 0),
                                           (5, 20000, 200, 392, 0), (1500, 40000, 256, 200, 0), (70, 300, 128, 256, 0), (600, 2500, 66, 129, 1)])
def test_rebuilt_rerank_rounds_every_query_vs_c_oracle(hw, B, N, d, k, arith):
    from hwer.device_index import DeviceIndex
    from oracle import c_oracle
    rng = np.random.default_rng(B * 7 + d)
    V32 = rng.standard_normal((B + N, d), dtype=np.float32).astype(np.float64)
    V32 = (V32 / np.linalg.norm(V32, axis=1, keepdims=True)).astype(np.float32)
    V32[B + 5] = V32[B + 4]                      # an exact duplicate: ordered by id
    V = V32.astype(np.float64) if arith == 0 else V32
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, arith=arith, audit=False)
    for want in (False, True):                   # lazy distances (scores only) and the full re-rank
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k, want_dist=want)
        assert (flags > 0).all()
        Q64, _, _ = ix.build_queries(np.arange(B))
        o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(V32[B:], Q64.cpu().numpy(), V32[:B].astype(np.float64), k, threads=c_oracle.effective_cpus())
        if arith == 0:
            assert np.array_equal(idx, o_idx + B) and np.array_equal(score, o_sc)
            if want:
                assert np.array_equal(dist, np.sqrt(o_d2))
        else:                                    # float32 flavour: the order follows the float32 scores - same neighbour SETS, exact order on samples
            assert np.array_equal(np.sort(idx, axis=1), np.sort(o_idx + B, axis=1))
            for u in (0, B // 2, B - 1):
                n_idx, n_sc, _ = orc.find_closest(V, np.arange(B, B + N), u, k=k)
                assert np.array_equal(idx[u], n_idx) and np.array_equal(score[u], n_sc)
