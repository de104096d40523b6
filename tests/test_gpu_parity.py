"""GPU parity tests: the HIP path (through the C ABI / the hwer host API) against the oracle and the golden
vectors captured from the reference.  Bit-exact for indices, float64 and float32 scores and distances; 1e-6 for the
float32 blend stage (tolerances written at each assert)."""
import hashlib

import numpy as np
import pytest

from conftest import unit_rows_f32
from oracle import hwer_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["auto", "streaming"])
def knn_path(request, monkeypatch):
    """Small catalogues take the dense path by default; "streaming" forces the MFMA screen + streaming top-K' filter
    (the large-catalogue path) onto the same inputs so that both are held to the same goldens."""
    if request.param == "streaming":
        monkeypatch.setenv("HWER_NO_DENSE", "1")
    return request.param


@pytest.fixture(scope="module")
def hw():
    import hwer
    from hwer import native
    native.lib()
    return hwer


def make_index(V, U, I, arith=None):
    from hwer.device_index import DeviceIndex
    return DeviceIndex(V, {"user": np.arange(U, dtype=np.int64), "item": np.arange(U, U + I, dtype=np.int64)}, arith=arith)


def _flav(V32, f):
    return V32.astype(np.float64) if f == "f64" else V32


# ------------------------------------------------------------------------------------------------- pair score
@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_pair_score_golden_g4(golden, hw, flav):
    g = golden("g4_predict")
    V = _flav(g["V32"], flav)
    ix = make_index(V, 64, 512)
    r = ix.pair_score(g["src"], g["dst"])
    assert r.dtype == V.dtype
    assert np.array_equal(r, g[f"score_{flav}"])  # bit-exact vs the reference


@pytest.mark.parametrize("d", [1, 5, 8, 31, 64, 66, 128, 130, 200, 256, 300, 515])
def test_pair_score_dims_bit_exact(hw, d):
    rng = np.random.default_rng(d)
    N, P = 700, 5003
    V32 = unit_rows_f32(10 + d, N, d)
    src = rng.integers(-1, N, P).astype(np.int32)
    dst = rng.integers(-1, N, P).astype(np.int32)
    from hwer.device_index import DeviceIndex
    for arith, V in ((0, V32.astype(np.float64)), (1, V32)):
        ix = DeviceIndex(V, {}, arith=arith) if d > 256 else make_index(V, 100, 600, arith)
        got = ix.pair_score(src, dst)
        assert np.array_equal(got, orc.pair_score(V, src, dst)), (d, arith)


def test_pair_score_empty(hw):
    ix = make_index(unit_rows_f32(1, 40, 16), 8, 32)
    assert ix.pair_score(np.zeros(0, np.int32), np.zeros(0, np.int32)).shape == (0,)


# ------------------------------------------------------------------------------------------------- blend stage
def test_content_tail_golden_g6(golden, hw):
    from hwer.content_recommender import content_tail, segment_mean, adjacency_csr
    g = golden("g6_content_tail")
    E32 = g["E"].astype(np.float32)
    got = content_tail(E32, g["src"], g["dst"])
    ref_A, ref = orc.content_tail(E32.astype(np.float64), g["src"], g["dst"])
    assert np.abs(got - ref).max() <= 1e-6          # stage A tolerance
    assert np.abs(got - g["final"]).max() <= 1e-6   # and against the reference's own float64 output
    indptr, indices = adjacency_csr(300, g["src"], g["dst"])
    oi, ox = orc.adjacency_csr(300, g["src"], g["dst"])
    assert np.array_equal(indptr, oi) and np.array_equal(indices, ox)
    A = segment_mean(E32, indptr, indices)
    assert np.abs(A - ref_A).max() <= 1e-6
    # float32 oracle of the same sequential order: bit-exact
    assert np.array_equal(A, orc.segment_mean(E32, indptr, indices))


@pytest.mark.parametrize("d,T", [(128, 2), (256, 4), (66, 3), (8, 1)])
def test_blend_normalize(hw, d, T):
    from hwer.content_recommender import blend_tables
    rng = np.random.default_rng(d * T)
    R = 1500
    tables = [unit_rows_f32(30 + t, 900 + 50 * t, d) for t in range(T)]
    idx = [None if t == 0 else rng.integers(0, tables[t].shape[0], R).astype(np.int32) for t in range(T)]
    tables[0] = unit_rows_f32(29, R, d)
    alpha = rng.dirichlet(np.ones(T), R).astype(np.float32)
    if T > 1:
        alpha[::5, -1] = 0.0  # "cold-start" rows: the last (collaborative) table gets weight 0
    out, norm = blend_tables(tables, alpha, idx, return_norm=True)
    ref = orc.blend_normalize(tables, idx, alpha)
    assert np.abs(out - ref).max() <= 1e-6
    assert np.abs(np.linalg.norm(out.astype(np.float64), axis=1) - 1).max() <= 1e-6
    pre = sum(alpha[:, t:t + 1].astype(np.float64) * (tables[t] if idx[t] is None else tables[t][idx[t]]) for t in range(T))
    assert np.abs(norm - np.linalg.norm(pre, axis=1)).max() <= 1e-6
    # shared alpha, no index: reference special case (E + A)/2
    E, A = tables[0], unit_rows_f32(77, R, d)
    out2 = blend_tables([E, A], np.array([0.5, 0.5], np.float32))
    ref2 = orc.unit_length((E.astype(np.float64) + A) / 2.0, axis=1)
    assert np.abs(out2 - ref2).max() <= 1e-6
    # a zero row normalises to NaN (no epsilon), like hwer/utils.py:43-44
    Z = E.copy(); Z[3] = 0
    assert np.isnan(blend_tables([Z], np.array([1.0], np.float32))[3]).all()


def test_unit_audit_golden_g8(golden, hw):
    from hwer.device_index import DeviceIndex
    g = golden("g8_unit_audit")
    for name in ("rows", "rows_zero", "rows_nan"):
        ix = DeviceIndex(g[name].astype(np.float32), {}, audit=True)
        v, mean_dev, pos, neg, nan = ix.audit_result
        ref = g["viol32_" + name]
        assert (v, pos, neg) == (int(ref[0]), int(ref[2]), int(ref[3])), name
        if not np.isnan(ref[1]):
            assert abs(mean_dev - ref[1]) <= 1e-6
        assert nan == (1 if name == "rows_nan" else 0)


# ------------------------------------------------------------------------------------------------- queries
@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_query_mix_golden_g3(golden, hw, flav):
    g = golden("g3_posneg")
    V = _flav(g["V32"], flav)
    ix = make_index(V, 64, 512)
    pos = [[int(x) for x in row if x != -2] for row in g["pos"]]
    neg = [[int(x) for x in row if x != -2] for row in g["neg"]]
    Q64, Q32, _ = ix.build_queries(np.arange(32), pos, neg)
    assert np.array_equal(Q64.cpu().numpy(), g[f"q_{flav}"].astype(np.float64))  # bit-exact vs the reference
    assert np.array_equal(Q32.cpu().numpy()[:, :32], g[f"q_{flav}"].astype(np.float32))
    assert not Q32.cpu().numpy()[:, 32:].any()


@pytest.mark.parametrize("d", [5, 66, 128, 200, 256])
def test_query_mix_dims(hw, d):
    rng = np.random.default_rng(d)
    V32 = unit_rows_f32(d, 400, d)
    for arith, V in ((0, V32.astype(np.float64)), (1, V32)):
        ix = make_index(V, 100, 300, arith)
        anchors = rng.integers(0, 100, 50)
        pos = [list(rng.integers(-1, 400, rng.integers(0, 12))) for _ in range(50)]
        neg = [list(rng.integers(-1, 400, rng.integers(0, 4))) for _ in range(50)]
        Q64, _, _ = ix.build_queries(anchors, pos, neg)
        ref = np.stack([orc.query_vector(V, int(a), np.asarray(p, np.int64), np.asarray(n, np.int64))
                        for a, p, n in zip(anchors, pos, neg)])
        assert np.array_equal(Q64.cpu().numpy(), ref.astype(np.float64)), (d, arith)


# ------------------------------------------------------------------------------------------------- k-NN
@pytest.mark.parametrize("flav", ["f64", "f32"])
@pytest.mark.parametrize("k", [1, 10, 200])
def test_knn_golden_g1(golden, hw, knn_path, flav, k):
    g = golden("g1_knn_basic")
    V = _flav(g["V32"], flav)
    ix = make_index(V, 64, 512)
    idx, score, dist, flags = ix.search("item", np.arange(64), k=k)
    assert (flags > 0).all()
    assert np.array_equal(idx, g[f"idx_{flav}_k{k}"])
    assert score.dtype == V.dtype and np.array_equal(score, g[f"score_{flav}_k{k}"])  # bit-exact scores


@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_knn_query_order_and_gcn_scoring_g5(golden, hw, knn_path, flav):
    g = golden("g1_knn_basic")
    V = _flav(g["V32"], flav)
    ix = make_index(V, 64, 512)
    for k in (10, 200):
        idx, score, dist, _ = ix.search("item", np.arange(64), k=k, scoring="gcn", sort_by_score=False)
        assert np.array_equal(idx, g[f"knnq_idx_{flav}_k{k}"])
        assert np.array_equal(dist, g[f"knnq_dist_{flav}_k{k}"])       # scikit-learn's distances, bit-exact
        assert np.array_equal(score, (-1 * g[f"knnq_dist_{flav}_k{k}"] + 2) / 2)
    idx, score, _, _ = ix.search("user", np.array([3]), k=5)
    assert np.array_equal(idx[0], g[f"u2u_idx_{flav}"]) and np.array_equal(score[0], g[f"u2u_score_{flav}"])


def test_knn_golden_g2_ml100k_shape(golden, hw, knn_path):
    g = golden("g2_knn_ml100k_shape")
    U, I, d, k = int(g["n_users"]), int(g["n_items"]), int(g["d"]), int(g["k"])
    V32 = np.concatenate([unit_rows_f32(int(g["seed_users"]), U, d), unit_rows_f32(int(g["seed_items"]), I, d)])
    ix = make_index(V32.astype(np.float64), U, I)
    idx, score, _, flags = ix.search("item", np.arange(U), k=k)
    assert (flags == 1).all()
    assert np.array_equal(idx - U, g["idx"].astype(np.int64))
    assert hashlib.sha256(np.ascontiguousarray(idx).tobytes()).hexdigest() == str(g["idx_sha256"])
    assert hashlib.sha256(np.ascontiguousarray(score).tobytes()).hexdigest() == str(g["score_sha256"])


@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_knn_posneg_golden_g3(golden, hw, knn_path, flav):
    g = golden("g3_posneg")
    V = _flav(g["V32"], flav)
    ix = make_index(V, 64, 512)
    pos = [[int(x) for x in row if x != -2] for row in g["pos"]]
    neg = [[int(x) for x in row if x != -2] for row in g["neg"]]
    idx, score, _, _ = ix.search("item", np.arange(32), k=int(g["k"]), positive=pos, negative=neg)
    assert np.array_equal(idx, g[f"idx_{flav}"])
    assert np.array_equal(score, g[f"score_{flav}"])


def test_knn_ties_g10(golden, hw, knn_path):
    g = golden("g10_ties")
    U, I = int(g["n_users"]), int(g["n_items"])
    ix = make_index(g["V32"].astype(np.float64), U, I)
    idx, score, _, _ = ix.search("item", np.arange(U), k=10)
    assert np.array_equal(score, g["score"])
    for u in range(U):
        o_idx, _, _ = orc.find_closest(g["V32"].astype(np.float64), np.arange(U, U + I), u, k=10)
        assert np.array_equal(idx[u], o_idx)  # the build's tie rule == the oracle's (dist asc, id asc)
        for s in np.unique(score[u]):
            assert set(idx[u][score[u] == s].tolist()) == set(g["idx"][u][g["score"][u] == s].tolist())


@pytest.mark.parametrize("d,N,B,k", [(64, 5000, 300, 200), (128, 9000, 257, 200), (256, 3000, 100, 50), (66, 2500, 70, 33),
                                     (32, 700, 33, 392), (128, 150, 40, 150), (8, 64, 5, 64)])
def test_knn_random_vs_oracle(hw, knn_path, d, N, B, k):
    U = B
    V32 = np.concatenate([unit_rows_f32(5 + d, U, d), unit_rows_f32(6 + d, N, d)])
    rows = np.arange(U, U + N)
    for arith, V in ((0, V32.astype(np.float64)), (1, V32)):
        ix = make_index(V, U, N, arith)
        idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
        assert (flags == 1).all()   # random rows: every query must be CERTIFIED by the screen (2 = exhaustive pass)
        for u in range(0, U, max(1, U // 40)):
            o_idx, o_score, o_dist = orc.find_closest(V, rows, u, k=k)
            assert np.array_equal(idx[u], o_idx), (d, arith, u)
            assert np.array_equal(score[u], o_score)
            assert np.array_equal(dist[u], o_dist)
        assert (np.diff(score, axis=1) <= 0).all()


@pytest.mark.parametrize("d,N,B,k", [(64, 5000, 300, 200), (128, 9000, 257, 200), (32, 700, 33, 392), (128, 150, 40, 150), (66, 2500, 70, 33)])
def test_knn_without_distances_matches(hw, knn_path, d, N, B, k):
    """want_dist=False (find_closest_neighbours' own output: ids + scores): the kernel computes exact distances only for the
    candidates the float32 screen leaves undecided - indices and scores must not change."""
    U = B
    V32 = np.concatenate([unit_rows_f32(15 + d, U, d), unit_rows_f32(16 + d, N, d)])
    rows = np.arange(U, U + N)
    for arith, V in ((0, V32.astype(np.float64)), (1, V32)):
        ix = make_index(V, U, N, arith)
        idx, score, dist, flags = ix.search("item", np.arange(U), k=k, want_dist=False)
        assert dist is None and (flags == 1).all()
        idx_f, score_f, _, _ = ix.search("item", np.arange(U), k=k)
        assert np.array_equal(idx, idx_f) and np.array_equal(score, score_f)
        for u in range(0, U, max(1, U // 25)):
            o_idx, o_score, _ = orc.find_closest(V, rows, u, k=k)
            assert np.array_equal(idx[u], o_idx), (d, arith, u)
            assert np.array_equal(score[u], o_score)


def test_knn_without_distances_score_ties(hw, knn_path):
    """exact score ties between provable members need their distance order: the lazy path must detect them and redo
    the query with all distances.  Duplicate item rows give exact ties in both score and distance (order = row id)."""
    d, N, B, k = 32, 2000, 16, 50
    X = unit_rows_f32(77, N, d)
    X[1000:1400] = X[:400]            # every item in the first 400 has a twin 1000 rows later
    users = unit_rows_f32(78, B, d)
    for arith, V in ((0, np.concatenate([users, X]).astype(np.float64)), (1, np.concatenate([users, X]))):
        ix = make_index(V, B, N, arith)
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k, want_dist=False)
        idx_f, score_f, _, _ = ix.search("item", np.arange(B), k=k)
        assert np.array_equal(idx, idx_f) and np.array_equal(score, score_f)
        for u in range(B):
            o_idx, o_score, _ = orc.find_closest(V, np.arange(B, B + N), u, k=k)
            assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score)


@pytest.mark.parametrize("d,N,B,k", [(128, 20000, 1, 200), (64, 9000, 7, 200), (32, 30000, 33, 50), (256, 12000, 70, 200),
                                     (128, 16000, 200, 100), (66, 8500, 256, 392), (128, 70000, 5, 10)])
def test_knn_small_batch_path(hw, d, N, B, k):
    """a handful of queries over a catalogue beyond the dense select: barrier-free HBM-bound screen + sampled-threshold
    row select (knn_small.hip).  Bit-exact against the oracle, with and without the distances."""
    V32 = np.concatenate([unit_rows_f32(25 + d, B, d), unit_rows_f32(26 + d, N, d)])
    rows = np.arange(B, B + N)
    for arith, V in ((0, V32.astype(np.float64)), (1, V32)):
        ix = make_index(V, B, N, arith)
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
        assert (flags == 1).all()
        idx_l, score_l, _, flags_l = ix.search("item", np.arange(B), k=k, want_dist=False)
        assert np.array_equal(idx, idx_l) and np.array_equal(score, score_l) and (flags_l == 1).all()
        for u in range(0, B, max(1, B // 12)):
            o_idx, o_score, o_dist = orc.find_closest(V, rows, u, k=k)
            assert np.array_equal(idx[u], o_idx), (d, arith, u)
            assert np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


@pytest.mark.parametrize("d,N,B,k", [(128, 50000, 5, 200), (64, 20000, 64, 100), (256, 12000, 33, 200), (32, 40000, 200, 50)])
def test_knn_small_batch_f16_storage_is_invisible(hw, d, N, B, k):
    """the float16 storage of the small-batch screen (half the bytes streamed) must not change a single bit of the result:
    same ids / scores / distances with and without it, both certified.  Unit rows here; rows with a wide dynamic range and mixed
    norms: tests/test_gpu_round4.py::test_f16_screen_adversarial_rows."""
    from hwer.device_index import DeviceIndex
    X = unit_rows_f32(35 + d, N, d)
    users = unit_rows_f32(36 + d, B, d)
    V = np.concatenate([users, X]).astype(np.float64)
    type_rows = {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}
    res = []
    for f16 in (True, False):
        ix = DeviceIndex(V, type_rows, arith=0)
        ix.use_f16_storage = f16
        res.append(ix.search("item", np.arange(B), k=k))
        assert (ix.tables["item"].X16 is not None) == f16
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    assert (res[0][3] == 1).all()
    for u in range(0, B, max(1, B // 6)):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(res[0][0][u], o_idx) and np.array_equal(res[0][1][u], o_score) and np.array_equal(res[0][2][u], o_dist)


def test_knn_small_batch_duplicates(hw):
    """heavy duplication defeats the sampled threshold (list overflow / ties at the cut): the queries must come back
    through the exhaustive pass, still exact."""
    d, N, B, k = 64, 12000, 4, 30
    X = unit_rows_f32(88, N, d)
    X[2000:9000] = X[100]                       # 7000 copies of one row
    users = np.concatenate([X[[100, 101]], unit_rows_f32(89, 2, d)])
    V = np.concatenate([users, X]).astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all() and flags[0] == 2
    for u in range(B):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


def test_knn_cluster_sorted_catalogue_retry(hw):
    """a catalogue sorted by cluster defeats the sample-scaled thresholds of the streaming filter for a few per cent of the
    queries; they must come back exact through the exact-threshold retry (flag 1), not one by one through the exhaustive
    pass."""
    rng = np.random.default_rng(3)
    d, N, B, k = 64, 120000, 3000, 100
    x = rng.standard_normal((N, d))
    lab = np.sort(rng.integers(0, 40, N))
    x = 0.6 * x + rng.standard_normal((40, d))[lab]
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    V = np.concatenate([unit_rows_f32(401, B, d), x.astype(np.float32)]).astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all() and int((flags == 2).sum()) <= 16, np.bincount(flags)
    rows = np.arange(B, B + N)
    for u in range(0, B, 97):
        o_idx, o_score, o_dist = orc.find_closest(V, rows, u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


def test_knn_multi_split_lists(hw):
    """few queries x many items: the item range is split over many workgroups / lists (reduce path)."""
    d, N, B, k = 64, 60000, 3, 100
    V32 = np.concatenate([unit_rows_f32(91, B, d), unit_rows_f32(92, N, d)])
    V = V32.astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all()
    for u in range(B):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


def test_knn_exact_fallback_kernel(hw, knn_path):
    """drive hwer_knn_exact_f64 directly and through massive duplication (certificate must fail, fallback must fix it)."""
    d, N, B, k = 32, 3000, 6, 20
    X = unit_rows_f32(55, N, d)
    X[100:900] = X[50]  # 800 identical rows: the K' cut falls inside an exact tie group for queries near row 50
    V32 = np.concatenate([X[[50, 51, 52, 53, 54, 55]] , X])
    V = V32.astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert flags[0] == 2, flags          # query 0 == the duplicated row: uncertifiable by the screen -> exhaustive pass
    for u in range(B):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)


def test_knn_exact_device_pass(hw, knn_path):
    """the on-device exhaustive pass (hwer_knn_exact_device) fixes up to 8 uncertified queries per search without a host
    round trip; the host-driven pass takes the rest.  12 queries sit on an 800-fold duplicated row."""
    d, N, k = 32, 3000, 20
    X = unit_rows_f32(56, N, d)
    X[100:900] = X[50]
    users = np.concatenate([np.repeat(X[[50]], 12, axis=0), X[[51, 52, 53]]])
    B = len(users)
    V = np.concatenate([users, X]).astype(np.float64)
    ix = make_index(V, B, N)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags[:12] == 2).all() and (flags[12:] == 1).all(), flags
    for u in range(B):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)
    # the device pass alone (no host fallback): exactly 8 of the 12 get fixed, the rest stay flagged 0 and are counted
    import torch
    from hwer import native
    lib = native.lib()
    Q64, Q32, a = ix.build_queries(np.arange(B))
    tt = ix.tables["item"]
    o_i = torch.empty((B, k), dtype=torch.int64, device=ix.device)
    o_s = torch.empty((B, k), dtype=torch.float64, device=ix.device)
    o_d = torch.empty((B, k), dtype=torch.float64, device=ix.device)
    o_f = torch.empty(B, dtype=torch.int32, device=ix.device)
    ws = torch.empty(lib.hwer_knn_workspace_bytes(B, N, d, k), dtype=torch.uint8, device=ix.device)
    wse = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(N, k), dtype=torch.uint8, device=ix.device)
    missed = torch.zeros(1, dtype=torch.int32, device=ix.device)
    P, S = native.ptr, native.stream_ptr
    for rep in range(2):   # twice: the header (list, tickets) must be reusable
        native.check(lib.hwer_knn_search_f32(P(tt.X), P(tt.hnorm), N, d, ix.ldx, P(Q32), P(Q64), P(ix.V), ix.ldv, P(a), B, k, 0, 1,
                                             P(tt.rows_dev), 0, tt.hnorm_max, P(o_i), P(o_s), P(o_d), P(o_f), P(ws), ws.numel(), S()), "search")
        assert int((o_f == 0).sum()) == 12
        native.check(lib.hwer_knn_exact_device(P(tt.X), N, d, ix.ldx, P(Q64), P(ix.V), ix.ldv, P(a), k, 0, 1, P(tt.rows_dev), 0,
                                               P(o_i), P(o_s), P(o_d), P(o_f), P(ws), P(missed), P(wse), wse.numel(), S()), "exact_device")
        f = o_f.cpu().numpy()
        assert int((f == 2).sum()) == 8 and int((f == 0).sum()) == 4 and int(missed.item()) == 4 * (rep + 1)
        fixed = np.nonzero(f == 2)[0]
        assert np.array_equal(o_i.cpu().numpy()[fixed], idx[fixed]) and np.array_equal(o_s.cpu().numpy()[fixed], score[fixed])


def test_multiknn_query_arbitrary_vector(golden, hw):
    from hwer import MultiKNN, Node
    from hwer.recommendation_base import BiMap
    g = golden("g1_knn_basic")
    V = g["V32"].astype(np.float64)
    nodes = [Node("user", i) for i in range(64)] + [Node("item", i) for i in range(512)]
    m = MultiKNN(BiMap(zip(nodes, range(len(nodes)))), V, leaf_size=128)
    q = orc.average_embedding(V, [7])
    res = m.query(q, "item", k=10)
    assert [m.nodes_to_idx[n] for n, _ in res] == g["knnq_idx_f64_k10"][7].tolist()
    assert np.array_equal(np.array([dd for _, dd in res]), g["knnq_dist_f64_k10"][7])
    assert set(m.knn) == {"user", "item"} and m.idxs["item"][0] == 64 and m.idxs["item"].inverse[64] == 0


# ------------------------------------------------------------------------------------------------- API level
def test_api_drop_in_and_errors_g11(golden, hw):
    from hwer import Node, NodeNotFoundException, VectorIndex
    g = golden("g1_knn_basic")
    V = g["V32"].astype(np.float64)
    users = [Node("user", i) for i in range(64)]
    items = [Node("item", i) for i in range(512)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=32)
    with pytest.raises(AssertionError):
        m.find_closest_neighbours("item", users[0])        # not fitted
    m.fit(users + items, V)
    res = m.find_closest_neighbours("item", users[5], k=10)
    assert [m.nodes_to_idx[n] for n, _ in res] == g["idx_f64_k10"][5].tolist()
    assert [s for _, s in res] == g["score_f64_k10"][5].tolist()
    assert m.find_items_for_user(users[5], k=10) == res
    with pytest.raises(ValueError, match="k must be less than or equal to the number of training points"):
        m.find_closest_neighbours("item", users[0], k=513)
    with pytest.raises(NodeNotFoundException):
        m.find_closest_neighbours("item", Node("user", "zz"))
    with pytest.raises(AssertionError):
        m.find_closest_neighbours("movie", users[0])
    with pytest.raises(AssertionError):
        m.add_nodes([Node("user", 0)])
    # predict with unknown nodes (G4 semantics) and get_embeddings
    p = m.predict([(users[0], items[0]), (Node("user", "nope"), items[1])])
    assert np.array_equal(p, orc.pair_score(V, np.array([0, -1]), np.array([64, 65])))
    assert np.array_equal(m.get_embeddings([users[1], Node("x", 1)]), orc.gather_rows(V, np.array([1, -1])))
    # non-unit vectors are refused like the reference's assert
    bad = VectorIndex(node_types={"user", "item"}, n_dims=32)
    with pytest.raises(AssertionError):
        bad.fit(users + items, V * 1.01)


def test_content_recommendation_fit_g7(golden, hw):
    from hwer import ContentRecommendation, Edge, IdentityEmbedding, Node
    g = golden("g7_content_fit_e2e")
    g6 = golden("g6_content_tail")
    # G6 scenario end to end through the drop-in class (IdentityEmbedding features, 3 node types)
    rng = np.random.default_rng(7)
    types = ["user", "item", "tag"]
    nodes = [Node(types[i % 3], i) for i in range(300)]
    feats = rng.standard_normal((300, 5))
    node_data = {n: {"f": feats[i]} for i, n in enumerate(nodes)}
    edges = [Edge(nodes[s], nodes[t], 1.0) for s, t in zip(g6["src"], g6["dst"])]
    model = ContentRecommendation({t: {"f": IdentityEmbedding(n_dims=5)} for t in types}, set(types), n_dims=8)
    final = model.fit(nodes, edges, node_data)
    assert np.abs(final - g6["final"]).max() <= 1e-6   # float32 device rows vs the reference's float64 rows
    res = model.find_closest_neighbours("item", nodes[0], k=7)
    assert [model.nodes_to_idx[n] for n, _ in res] == g6["knn_idx"].tolist()
    assert np.abs(np.array([s for _, s in res]) - g6["knn_score"]).max() <= 1e-5
    assert final.shape == (300, 12) and g["final"].shape == (18, 6)


def test_extraction_efficiency_g9(golden, hw):
    import random
    from hwer import Edge, Node, VectorIndex, validation
    g = golden("g9_metrics")
    U, I = int(g["ee_n_users"]), int(g["ee_n_items"])
    users = [Node("user", i) for i in range(U)]
    items = [Node("item", i) for i in range(I)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=32)
    m.fit(users + items, g["ee_V32"].astype(np.float64))
    allv = users + items
    train = [Edge(allv[int(u)], allv[int(i)], float(r)) for u, i, r in g["ee_train"]]
    valid = [Edge(allv[int(u)], allv[int(i)], float(r)) for u, i, r in g["ee_valid"]]
    random.seed(0)
    res = validation.extraction_efficiency(m, train, valid, validation.model_get_topk, "item")
    for key in ("recall@100", "ndcg_b@100", "ndcg_b@10", "recall@10", "diversity"):
        assert res["metrics"][key] == pytest.approx(float(g["ee_" + key.replace("@", "_at_")]), rel=0, abs=1e-15), key
    assert 0 <= res["metrics"]["ncf_hr"] <= 1 and res["metrics"]["retrieval_time"] > 0


@pytest.mark.parametrize("B,k,cut", [(1, 200, 100), (37, 200, 100), (64, 50, 100), (9, 130, 128), (5, 3, 1)])
def test_rank_filter_seen_kernel(hw, B, k, cut):
    """hwer_rank_filter_seen against a plain-Python restatement of hwer/validation.py:133-141: survivors in order, lengths,
    hit bits, coverage bits; empty seen / truth rows, ranked lists shorter than the cut, negative (padding) ids."""
    import torch
    from hwer import native
    lib = native.lib()
    rng = np.random.default_rng(B * 1000 + k)
    n_ids, base = 5000, 1000
    ranked = np.stack([rng.permutation(n_ids)[:k] + base for _ in range(B)]).astype(np.int64)
    if k > 10:
        ranked[0, 7] = -1                                    # padding entries are skipped
    seen = [np.sort(rng.choice(ranked[b][ranked[b] >= 0], size=int(rng.integers(0, min(k, 60))), replace=False)) if b % 4 else
            np.zeros(0, np.int64) for b in range(B)]
    truth = [np.sort(rng.choice(np.arange(base, base + n_ids), size=int(rng.integers(0, 40)), replace=False)) if b % 3 else
             np.sort(ranked[b][ranked[b] >= 0][:5]) for b in range(B)]
    ptr = lambda rows: np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)   # noqa: E731
    cat = lambda rows: np.concatenate(rows + [np.zeros(1, np.int64)]).astype(np.int64)              # noqa: E731
    dev = torch.device("cuda")
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)   # noqa: E731
    d_r, d_sp, d_si, d_tp, d_ti = T(ranked), T(ptr(seen)), T(cat(seen)), T(ptr(truth)), T(cat(truth))
    nw = (cut + 63) // 64
    unseen = torch.empty((B, cut), dtype=torch.int64, device=dev)
    ulen = torch.empty(B, dtype=torch.int32, device=dev)
    hits = torch.empty((B, nw), dtype=torch.int64, device=dev)
    cover = torch.zeros((n_ids + 31) // 32, dtype=torch.int32, device=dev)
    native.check(lib.hwer_rank_filter_seen(native.ptr(d_r), B, k, native.ptr(d_sp), native.ptr(d_si), native.ptr(d_tp), native.ptr(d_ti), cut,
                                           native.ptr(unseen), native.ptr(ulen), native.ptr(hits), native.ptr(cover), base, n_ids,
                                           native.stream_ptr()), "rank_filter_seen")
    u_h, l_h, h_h = unseen.cpu().numpy(), ulen.cpu().numpy(), hits.cpu().numpy().view(np.uint64)
    covered = set()
    for b in range(B):
        want = [int(i) for i in ranked[b] if i >= 0 and i not in set(seen[b].tolist())][:cut]
        assert l_h[b] == len(want) and u_h[b, :len(want)].tolist() == want and (u_h[b, len(want):] == -1).all()
        tset = set(truth[b].tolist())
        for p, i in enumerate(want):
            assert bool((int(h_h[b, p // 64]) >> (p % 64)) & 1) == (i in tset), (b, p)
        assert all(((int(h_h[b, p // 64]) >> (p % 64)) & 1) == 0 for p in range(len(want), 64 * nw))
        covered.update(want)
    bits = np.unpackbits(cover.cpu().numpy().view(np.uint8), bitorder="little")[:n_ids]
    assert set((np.nonzero(bits)[0] + base).tolist()) == covered
    # no seen / truth CSR at all
    native.check(lib.hwer_rank_filter_seen(native.ptr(d_r), B, k, None, None, None, None, cut, native.ptr(unseen), native.ptr(ulen),
                                           native.ptr(hits), None, 0, 0, native.stream_ptr()), "rank_filter_seen")
    assert (hits.cpu().numpy() == 0).all()
    for b in range(B):
        want = [int(i) for i in ranked[b] if i >= 0][:cut]
        assert ulen.cpu().numpy()[b] == len(want) and unseen.cpu().numpy()[b, :len(want)].tolist() == want


def test_extraction_efficiency_device_matches_host(hw):
    """the on-device seen-filter / cuts / hit marking (hwer_rank_filter_seen) against the host post-processing on random
    interactions: identical prediction lists and bit-identical metrics."""
    import random
    from hwer import Edge, Node, VectorIndex, validation
    rng = np.random.default_rng(5)
    U, I, d = 90, 700, 24
    V = np.concatenate([unit_rows_f32(301, U, d), unit_rows_f32(302, I, d)]).astype(np.float64)
    users = [Node("user", i) for i in range(U)]
    items = [Node("item", i) for i in range(I)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    m.fit(users + items, V)
    train, valid = [], []
    for u in range(U):
        near = np.argsort(-(V[U:] @ V[u]))[:150]          # interactions among the nearest items: the seen filter has work to do
        picks = rng.choice(near, size=int(rng.integers(5, 120)), replace=False)
        for j, it in enumerate(picks):
            (valid if j % 7 == 0 and u % 3 != 0 else train).append(Edge(users[u], items[int(it)], float(rng.integers(1, 6))))
    random.seed(1)
    dev = validation.extraction_efficiency(m, train, valid, validation.model_get_topk, "item")
    random.seed(1)
    host = validation._extraction_efficiency_host(m, train, valid, validation.model_get_topk, "item")
    assert dev["predictions"] == host["predictions"] and dev["train_predictions"] == host["train_predictions"]
    assert dev["actuals"] == host["actuals"]
    for key in host["metrics"]:
        if key != "retrieval_time":
            assert dev["metrics"][key] == host["metrics"][key], key
    for key in host["extra"]:
        assert dev["extra"][key] == host["extra"][key], key


def test_persisted_model_roundtrip(hw, tmp_path):
    """save_model / restore_model (hwer.persist): the restored index returns the same neighbours and scores."""
    from hwer import Node, VectorIndex, GcnScoredIndex, persist
    U, I, d = 40, 900, 48
    V32 = np.concatenate([unit_rows_f32(501, U, d), unit_rows_f32(502, I, d)])
    nodes = [Node("user", i) for i in range(U)] + [Node("item", "i%d" % i) for i in range(I)]
    for cls, V in ((VectorIndex, V32.astype(np.float64)), (GcnScoredIndex, V32)):
        m = cls(node_types={"user", "item"}, n_dims=d)
        m.fit(nodes, V)
        path = str(tmp_path / ("m_%s.npz" % cls.__name__))
        persist.save_model(path, m)
        m2 = persist.restore_model(path)
        assert type(m2) is cls and m2.scoring == m.scoring
        for u in (0, 7, 39):
            a = m.find_closest_neighbours("item", nodes[u], k=60)
            b = m2.find_closest_neighbours("item", nodes[u], k=60)
            assert [n for n, _ in a] == [n for n, _ in b] and [s for _, s in a] == [s for _, s in b]
        assert np.array_equal(m.predict([(nodes[1], nodes[50]), (nodes[2], nodes[99])]), m2.predict([(nodes[1], nodes[50]), (nodes[2], nodes[99])]))


# ------------------------------------------------------------------------------------------------- merge (X1)
def test_merge_topk_matches_single_shard(hw):
    import torch
    from hwer.distributed import merge_on_device, shard_bounds
    d, N, B, k, G = 64, 4000, 50, 100, 4
    V32 = np.concatenate([unit_rows_f32(71, B, d), unit_rows_f32(72, N, d)])
    V = V32.astype(np.float64)
    full = make_index(V, B, N)
    Q64, Q32, a = full.build_queries(np.arange(B))
    ref = full.search_queries("item", Q64, Q32, k, a)
    parts = []
    for r in range(G):
        lo, hi = shard_bounds(N, G, r)
        from hwer.device_index import DeviceIndex
        Vr = np.concatenate([V[:B], V[B + lo:B + hi]])
        ixr = DeviceIndex(Vr, {"user": np.arange(B), "item": np.arange(B, B + hi - lo)}, arith=0)
        i, s, d2, _ = ixr.search_queries("item", Q64, Q32, k, a, sort_by_score=False, id_base=B + lo)
        parts.append((i, s, d2))
    g_idx = torch.stack([p[0] for p in parts]); g_s = torch.stack([p[1] for p in parts]); g_d = torch.stack([p[2] for p in parts])
    idx, score, dist2 = merge_on_device(g_idx, g_s, g_d, True)
    assert torch.equal(idx, ref[0]) and torch.equal(score, ref[1]) and torch.equal(dist2, ref[2])
    # the receive-buffer layout of the user-partitioned exchange, (rank, array, B, k), merged in place (strided lists)
    recv = torch.stack([g_idx, g_s.view(torch.int64), g_d.view(torch.int64)], dim=1).contiguous()
    v_idx, v_s, v_d = recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64)
    assert not v_idx.is_contiguous() and v_idx.stride(0) == 3 * B * k
    idx2, score2, dist22 = merge_on_device(v_idx, v_s, v_d, True)
    assert torch.equal(idx2, ref[0]) and torch.equal(score2, ref[1]) and torch.equal(dist22, ref[2])


@pytest.mark.parametrize("G,k,B", [(1, 200, 7), (2, 200, 33), (3, 1, 5), (8, 200, 20), (16, 512, 3), (17, 100, 4), (40, 200, 3),
                                   (4, 37, 9), (8192, 1, 2)])
def test_merge_topk_shapes(hw, G, k, B):
    """hwer_merge_topk against a lexsort merge: both kernel paths (rank merge for <= 16 lists, LDS sort beyond), exact
    distance ties across lists, padded short lists (id -1, dist2 +inf), both output orders."""
    import torch
    from hwer.distributed import merge_on_device
    rng = np.random.default_rng(1000 * G + k)
    n = G * k
    idx = np.empty((G, B, k), np.int64); d2 = np.empty((G, B, k)); sc = np.empty((G, B, k))
    for b in range(B):
        ids = rng.permutation(10 * n + 7)[:n].astype(np.int64) + 5
        dist = np.round(rng.random(n), 2 if b % 2 else 12)          # 2 decimals: many exact ties between lists
        score = np.round(rng.random(n), 1 if b % 3 == 0 else 9)     # and exact score ties
        for g in range(G):
            sl = slice(g * k, (g + 1) * k)
            i_g, d_g, s_g = ids[sl].copy(), dist[sl].copy(), score[sl].copy()
            n_pad = 0 if G == 1 else [0, 0, min(k - 1, 3), k // 2][(g + b) % 4]   # some shards are short (never the whole row: G*m >= k real entries not needed, pads just lose)
            if n_pad:
                i_g[k - n_pad:], d_g[k - n_pad:], s_g[k - n_pad:] = -1, np.inf, 0.0
            o = np.lexsort((i_g, d_g))
            idx[g, b], d2[g, b], sc[g, b] = i_g[o], d_g[o], s_g[o]
    dev = torch.device("cuda")
    t = [torch.from_numpy(x).to(dev) for x in (idx, sc, d2)]
    for by_score in (False, True):
        o_i, o_s, o_d = [x.cpu().numpy() for x in merge_on_device(*t, by_score)]
        for b in range(B):
            ii, dd, ss = idx[:, b].reshape(-1), d2[:, b].reshape(-1), sc[:, b].reshape(-1)
            pos = np.arange(n)
            order = np.lexsort((pos, ii, dd))[:k]
            if by_score:
                order = order[np.argsort(-ss[order], kind="stable")]
            assert np.array_equal(o_i[b], ii[order]), (G, k, b, by_score)
            assert np.array_equal(o_d[b], dd[order]) and np.array_equal(o_s[b], ss[order])


@pytest.mark.parametrize("B,k,G", [(6040, 200, 8), (10, 7, 4), (5, 3, 8), (33, 200, 2), (1, 1, 1)])
def test_pack_exchange_kernel_matches_host_packing(hw, B, k, G):
    import torch
    from hwer.distributed import pack_send_buffer
    g = torch.Generator().manual_seed(B * 131 + G)
    idx = torch.randint(0, 1 << 40, (B, k), generator=g)
    score = torch.rand((B, k), generator=g, dtype=torch.float64)
    dist2 = torch.rand((B, k), generator=g, dtype=torch.float64)
    host = pack_send_buffer(idx, score, dist2, G)
    dev = pack_send_buffer(idx.cuda(), score.cuda(), dist2.cuda(), G)
    assert torch.equal(dev.cpu(), host)


@pytest.mark.parametrize("path", ["candidates", "exhaustive"])
@pytest.mark.parametrize("k,flavour", [(393, "f64"), (600, "f64"), (1500, "f64"), (500, "f32"), (500, "gcn")])
def test_large_k_path_matches_oracle(hw, monkeypatch, k, flavour, path):
    """k above the selection kernels' limit (392); the reference takes any k <= N_t (hwer/recommendation_base.py:79).  Round 3: a
    library-GEMM screen + exact re-rank of k + 32 candidates + certificate (`_search_candidates`, flags 1; 2 for queries whose k-th
    distance ties across the bound); the all-pairs float64 pass of rounds 1-2 stays as the certified fallback (forced here by
    HWER_FORCE_EXHAUSTIVE: flags 2).  Same indices, scores and distances as the oracle on both, ties included."""
    if path == "exhaustive":
        monkeypatch.setenv("HWER_FORCE_EXHAUSTIVE", "1")
    d, N, B = 48, 1500, 9
    V32 = np.concatenate([unit_rows_f32(801, B, d), unit_rows_f32(802, N, d)])
    V32[B + 7] = V32[B + 3]            # duplicate rows: an exact distance tie
    V32[B + 900] = V32[B + 3]
    V = V32.astype(np.float64) if flavour != "f32" else V32
    ix = make_index(V, B, N)
    if flavour == "gcn":
        idx, sc, dist, fl = ix.search("item", np.arange(B), k=k, scoring="gcn")
    else:
        idx, sc, dist, fl = ix.search("item", np.arange(B), k=k)
    assert (fl == 2).all() if path == "exhaustive" else (fl > 0).all()
    for b in range(B):
        o_idx, o_sc, o_dist = orc.find_closest(V, np.arange(B, B + N), b, k=k, scoring="gcn" if flavour == "gcn" else "base")
        assert np.array_equal(idx[b], o_idx), (k, flavour, b)
        assert np.array_equal(sc[b], o_sc) and np.array_equal(dist[b], o_dist)
    with pytest.raises(ValueError):
        ix.search("item", np.arange(B), k=N + 1)


def test_blend_with_fused_audit(hw):
    """hwer_blend_normalize_audit_f32 == blend followed by the separate audit (counts exact)."""
    import torch
    from hwer import native
    from hwer.device_index import DeviceIndex
    lib = native.lib()
    dev = torch.device("cuda")
    R, d, T = 5000, 64, 2
    tabs = [torch.from_numpy(unit_rows_f32(300 + t, R, d)).to(dev) for t in range(T)]
    tabs[1][7] = -tabs[0][7]            # row 7 blends to zero -> NaN after normalisation
    ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
    alpha = torch.full((T,), 0.5, dtype=torch.float32, device=dev)
    out = torch.zeros((R, d), dtype=torch.float32, device=dev)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)
    devsum = torch.zeros(1, dtype=torch.float64, device=dev)
    native.check(lib.hwer_blend_normalize_audit_f32(native.ptr(ptrs), d, None, native.ptr(alpha), 0, T, R, d, native.ptr(out), d,
                                                    1e-4, native.ptr(counts), native.ptr(devsum), native.stream_ptr()))
    ix = DeviceIndex(out, {}, audit=True)
    v, mean_dev, pos, neg, nan = ix.audit_result
    assert counts.tolist()[:3] == [pos, neg, nan] == [0, 0, 1]
    ref = orc.blend_normalize([t.cpu().numpy() for t in tabs], [None, None], np.array([0.5, 0.5]))
    ok = ~np.isnan(ref).any(1)
    assert np.abs(out.cpu().numpy()[ok] - ref[ok]).max() <= 1e-6 and (~ok).sum() == 1


def test_prepare_for_knn_and_gcn_index(golden, hw):
    from hwer import GcnScoredIndex, Node, prepare_for_knn
    g = golden("g1_knn_basic")
    raw = (g["V32"] * np.linspace(0.5, 3.0, 576, dtype=np.float32)[:, None])      # un-normalised "collaborative" vectors
    unit = prepare_for_knn(raw, 32)
    assert unit.dtype == np.float32 and np.abs(unit - orc.unit_length(raw.astype(np.float64), axis=1)).max() <= 1e-6
    with pytest.raises(ValueError):
        prepare_for_knn(raw, 64)
    assert prepare_for_knn(np.concatenate([raw, raw], axis=1), 32).shape == (576, 32)   # PCA when wider
    nodes = [Node("user", i) for i in range(64)] + [Node("item", i) for i in range(512)]
    m = GcnScoredIndex(node_types={"user", "item"}, n_dims=32)
    m.fit(nodes, raw)
    res = m.find_closest_neighbours("item", nodes[3], k=10)
    V = unit                                                            # float32 flavour, score = (2 - dist)/2
    o_idx, o_score, _ = orc.find_closest(V, np.arange(64, 576), 3, k=10, scoring="gcn")
    assert [m.nodes_to_idx[n] for n, _ in res] == o_idx.tolist()
    assert [s for _, s in res] == o_score.tolist()


@pytest.mark.parametrize("N,k", [(5, 3), (1, 1), (33, 33), (31, 7)])
def test_knn_tiny_catalogues(hw, knn_path, N, k):
    d, U = 8, 3
    V32 = np.concatenate([unit_rows_f32(401, U, d), unit_rows_f32(402 + N, N, d)])
    V = V32.astype(np.float64)
    ix = make_index(V, U, N)
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
    for u in range(U):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(U, U + N), u, k=k)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_score) and np.array_equal(dist[u], o_dist)
    with pytest.raises(ValueError):
        ix.search("item", np.arange(U), k=N + 1)


# ------------------------------------------------------------------------------------------------- fit-side table (§8f-3)
def _gram_call(B, seg):
    import torch
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda:0")
    n_seg, D = len(seg) - 1, B.shape[1]
    Bd = torch.from_numpy(np.ascontiguousarray(B)).to(dev)
    sd = torch.from_numpy(np.asarray(seg, dtype=np.int64)).to(dev)
    mx = int(np.diff(seg).max())
    G = torch.full((n_seg, D + 1, D + 1), np.nan, dtype=torch.float64, device=dev)
    nb = lib.hwer_segment_gram_workspace_bytes(D, n_seg, mx)
    ws = torch.empty(nb // 8, dtype=torch.float64, device=dev)
    native.check(lib.hwer_segment_gram_f64(native.ptr(Bd), Bd.stride(0), D, native.ptr(sd), n_seg, mx, native.ptr(G),
                                           native.ptr(ws), nb, native.stream_ptr()), "segment_gram")
    return G.cpu().numpy()


@pytest.mark.parametrize("D", [1, 5, 15, 16, 63, 64, 130, 257])
def test_segment_gram_matches_numpy(hw, D):
    """[B | 1]^T [B | 1] per row segment (float64 MFMA, two-stage deterministic reduction) vs numpy; tolerance: 1e-12 of
    the entry scale (the sums run in a different order than BLAS)."""
    rng = np.random.default_rng(D)
    n = 21000
    B = rng.standard_normal((n, D))
    seg = np.asarray([0, 3, 3, 700, 700 + 8192, 700 + 8192 + 8193, n], dtype=np.int64)   # tiny, empty, exact chunk, chunk + 1
    got = _gram_call(B, seg)
    for s in range(len(seg) - 1):
        A = np.concatenate([B[seg[s]:seg[s + 1]], np.ones((seg[s + 1] - seg[s], 1))], axis=1)
        ref = A.T @ A
        assert np.abs(got[s] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (D, s)
        assert np.array_equal(got[s], got[s].T)                     # mirrored tiles / symmetric by construction
        assert got[s][D, D] == seg[s + 1] - seg[s]
    assert np.array_equal(got, _gram_call(B, seg))                  # deterministic


@pytest.mark.parametrize("D,n", [(1, 3), (5, 8), (13, 17), (64, 64), (130, 100), (200, 256)])
def test_pca_project_unit_matches_numpy(hw, D, n):
    import torch
    from hwer import native
    from hwer.device_index import roundup
    lib = native.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(100 * D + n)
    rows = 1000 + D
    B = rng.standard_normal((rows, D))
    C = rng.standard_normal((D, n))
    bias = rng.standard_normal(n)
    npad = lib.hwer_pca_padded_components(n)
    P = np.zeros((roundup(D, 4), npad))
    P[:D, :n] = C
    bp = np.zeros(npad)
    bp[:n] = bias
    perm = rng.permutation(rows + 7)[:rows].astype(np.int64)         # scatter into a larger table
    n_onehot, col = 4, 2
    ld = roundup(n + n_onehot, 4) + 4
    o32 = torch.full((rows + 7, ld), 7.0, dtype=torch.float32, device=dev)
    o64 = torch.full((rows + 7, ld), 7.0, dtype=torch.float64, device=dev)
    Bd, Pd, bd, pd = (torch.from_numpy(x).to(dev) for x in (B, P, bp, perm))
    native.check(lib.hwer_pca_project_unit_f64(native.ptr(Bd), D, rows, D, native.ptr(Pd), npad, native.ptr(bd), n, native.ptr(pd),
                                               col, n_onehot, native.ptr(o32), ld, native.ptr(o64), ld, native.stream_ptr()), "proj")
    y = B @ C + bias
    u = y / np.linalg.norm(y, axis=1, keepdims=True)
    ref = np.zeros((rows, ld))
    ref[:, :n] = u
    ref[:, n + col] = 1.0
    g64, g32 = o64.cpu().numpy(), o32.cpu().numpy()
    assert np.abs(g64[perm] - ref).max() <= 1e-13                     # float64 MFMA vs BLAS: rounding only
    assert np.abs(g32[perm] - ref).max() <= 6e-8                      # float32 rounding of unit rows
    untouched = np.setdiff1d(np.arange(rows + 7), perm)
    assert np.all(g32[untouched] == 7.0) and np.all(g64[untouched] == 7.0)


def _sign_ties(ref_components):
    """Components whose sign in the reference is a coin toss: svd_flip makes the entry of largest magnitude positive, and
    with two node types the two one-hot entries of a component are exact negatives of each other - when they are the
    largest, rounding noise picked the reference's sign."""
    a = np.sort(np.abs(ref_components), axis=1)
    return (a[:, -1] - a[:, -2]) <= 1e-9


@pytest.mark.parametrize("tag", ["small", "multi"])
def test_fit_feature_table_g12(golden, hw, tag):
    """hwer/content_recommender.py:50-76 on the device vs the reference (G12: one IncrementalPCA batch / two batches) and
    vs the oracle's SVD restatement.  Tolerance 1e-9 on the float64 rows (Gram + eigh instead of SVD), 1e-6 on the float32
    rows the path hands on."""
    from conftest import g12_scenario
    from hwer.content_recommender import content_tail
    from hwer.fit_pca import device_feature_table
    g = golden("g12_fit_pca")
    sc = g12_scenario(g, tag)
    blocks = [orc.unit_length(f, axis=1) if has else np.zeros((f.shape[0], 1)) for f, has in zip(sc["feats"], sc["with_data"])]
    det = {}
    E32, E64, D = device_feature_table(blocks, sc["type_id"], sc["n_dims"], block_order=sc["order"], want_f64=True, details=det)
    n = sc["n_dims"]
    assert D == n + (4 if tag == "small" else 2) and len(det["batches"]) == (1 if tag == "small" else 2)
    tie = _sign_ties(g[tag + "_components"])
    sign = np.sign((det["components"] * g[tag + "_components"]).sum(1))
    assert np.all(sign[~tie] == 1.0)                                   # flips only where the reference's sign is a tie
    assert np.abs(det["components"] * sign[:, None] - g[tag + "_components"]).max() <= 1e-9
    assert np.abs(det["mean"] - g[tag + "_mean"]).max() <= 1e-12
    assert np.abs(det["singular_values"] - g[tag + "_singular_values"]).max() <= 1e-8
    rows = g[tag + "_rows"]
    e64, e32 = E64.cpu().numpy(), E32.cpu().numpy()
    assert np.abs(e64[rows, :n] * sign - g[tag + "_pca_unit"]).max() <= 1e-9
    assert np.abs(e32[rows, :n] * sign - g[tag + "_pca_unit"]).max() <= 1e-6
    Eo, _ = orc.fit_feature_table(blocks, sc["type_id"], n, sc["order"])
    full_sign = np.concatenate([sign, np.ones(D - n)])
    assert np.abs(e64[:, :D] * full_sign - Eo).max() <= 1e-9            # every row, incl. the one-hot columns
    assert np.all(e32[:, D:] == 0)
    # the tail on top of it (neighbour mean + blend + unit rows) vs the reference's final vectors
    final = content_tail(E32, sc["src"], sc["dst"], d=D)
    assert np.abs(final[rows] * full_sign - g[tag + "_final"]).max() <= 1e-6


def test_content_recommendation_fit_g12_small(golden, hw):
    """ContentRecommendation.fit end to end (embedders on the host, everything after them on the device), including a node
    type without node_data."""
    from conftest import g12_scenario
    from hwer import ContentRecommendation, Edge, IdentityEmbedding, Node
    g = golden("g12_fit_pca")
    sc = g12_scenario(g, "small")
    types = sc["types"]
    nodes = [Node(types[t], i) for i, t in enumerate(sc["type_id"])]
    cursor = [0] * len(types)
    node_data = {}
    for nd, t in zip(nodes, sc["type_id"]):
        if sc["with_data"][t]:
            node_data[nd] = {"f": sc["feats"][t][cursor[t]]}
        cursor[t] += 1
    edges = [Edge(nodes[s], nodes[t], 1.0) for s, t in zip(sc["src"], sc["dst"])]
    mapper = {types[t]: {"f": IdentityEmbedding(n_dims=sc["feats"][t].shape[1])} for t in range(len(types)) if sc["with_data"][t]}
    model = ContentRecommendation(mapper, set(types), n_dims=sc["n_dims"])
    final = model.fit(nodes, edges, node_data)
    assert final.shape == g["small_final"].shape
    assert np.abs(final - g["small_final"]).max() <= 1e-6


@pytest.mark.parametrize("case", [
    dict(N=500, dims=[9], n=4, seed=1),                              # one node type (constant one-hot column)
    dict(N=3000, dims=[3, 1, 17, 6], n=None, seed=2, zero=[1]),      # four types, one without features, n_dims = inf
    dict(N=65536 + 5, dims=[4, 5], n=8, seed=3),                     # short last batch folded into the first (one batch)
    dict(N=65536 + 8, dims=[4, 5], n=8, seed=4),                     # ... and the smallest second batch (8 rows)
    dict(N=140000, dims=[20, 12, 7], n=16, seed=5),                  # three batches
    dict(N=20000, dims=[70, 40], n=64, seed=6),                      # wide blocks: two 64-column tiles, 4 MFMA column tiles
])
def test_fit_feature_table_vs_oracle_shapes(hw, case):
    """Device fit-side path vs the oracle's SVD restatement (itself pinned by G12) over table shapes / batch structures.
    Tolerance 1e-8 on the float64 rows, components compared up to the sign where the oracle's own sign is a near-tie."""
    from hwer.fit_pca import device_feature_table
    rng = np.random.default_rng(case["seed"])
    N, dims = case["N"], case["dims"]
    T = len(dims)
    tid = rng.integers(0, T, N)
    blocks = []
    for t in range(T):
        nt = int((tid == t).sum())
        if t in case.get("zero", []):
            blocks.append(np.zeros((nt, 1)))
        else:
            # anisotropic features: well-separated principal directions (isotropic noise has no preferred basis to compare)
            f = rng.standard_normal((nt, dims[t])) * (1.0 + np.arange(dims[t]))[None, :] ** -0.5
            blocks.append(orc.unit_length(f, axis=1))
    order = list(rng.permutation(T))
    det = {}
    E32, E64, D = device_feature_table(blocks, tid, case["n"], block_order=order, want_f64=True, details=det)
    W = sum(b.shape[1] for b in blocks) + T
    n = case["n"] if case["n"] is not None else 2 ** int(np.log2(W))
    table = np.zeros((N, W))
    c = 0
    for t in order:
        table[np.flatnonzero(tid == t), c:c + blocks[t].shape[1]] = blocks[t]
        c += blocks[t].shape[1]
    table[np.arange(N), W - T + tid] = 1.0
    scores, comps, mean, sv = orc.incremental_pca_fit_transform(table, n)
    assert det["batches"] == orc.pca_batches(N, 2 ** 16, n)
    keep = sv > 1e-6 * sv[0]                                          # directions with variance (a null direction has no defined basis)
    sign = np.sign((det["components"] * comps).sum(1))
    sign[sign == 0] = 1.0
    assert np.all(sign[keep & ~_sign_ties(comps)] == 1.0)
    assert np.abs(det["singular_values"] - sv).max() <= 1e-7 * sv[0]
    assert np.abs((det["components"] * sign[:, None] - comps)[keep]).max() <= 1e-8
    Eo, _ = orc.fit_feature_table(blocks, tid, n, order)
    e64 = E64.cpu().numpy()
    if keep.all():
        assert np.abs(e64[:, :n] * sign - Eo[:, :n]).max() <= 1e-8
    ncol = 2 ** int(np.ceil(np.log2(T)))
    assert D == n + ncol
    assert np.array_equal(e64[:, n:D], Eo[:, n:])                     # one-hot columns
    assert np.abs(E32.cpu().numpy()[:, :D] - e64[:, :D]).max() <= 6e-8


# ------------------------------------------------------------------------------------------------- float16 streaming screen
@pytest.mark.parametrize("d,N,B,k,splits", [
    (128, 30000, 2100, 200, "item splits, TQ = 1"),
    (64, 20000, 3000, 50, "d = 64"),
    (256, 16000, 1500, 100, "d = 256"),
    (100, 12000, 1100, 10, "padded dim"),
    (128, 9000, 70000, 200, "one list, TQ = 2"),
])
def test_knn_f16_streaming_screen_is_invisible(hw, monkeypatch, d, N, B, k, splits):
    """HWER_KNN_F16_SCREEN: the streaming screen on the float16 copy of the table (v_mfma_f32_32x32x16_f16) must return the
    same ids / scores / distances as the float32 screen and the oracle - bit for bit.  Row norms here are off 1 by at most 4e-5 (inside
    the audit); rows with a wide dynamic range and norms 1e-3 ... 1: tests/test_gpu_round4.py::test_f16_screen_adversarial_rows.
    (All five shapes hold more than 1 024 queries and therefore stream anyway; HWER_NO_SMALL - read by the host mirror,
    DeviceIndex._path_options -> HWER_KNN_NO_SMALL - only pins that choice against future changes of the path rule.)"""
    from hwer.device_index import DeviceIndex
    monkeypatch.setenv("HWER_NO_SMALL", "1")
    X = unit_rows_f32(135 + d, N, d)
    X[::7] *= np.float32(1.00004)                      # row norms off 1 by up to 4e-5 (inside the audit's 1e-4)
    users = unit_rows_f32(136 + d, B, d)
    V = np.concatenate([users, X]).astype(np.float64)
    type_rows = {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}
    res = []
    for f16 in (True, False):
        ix = DeviceIndex(V, type_rows, arith=0)
        ix.f16_screen = f16
        res.append(ix.search("item", np.arange(B), k=k))
        assert (ix.tables["item"].X16 is not None) == f16
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert (res[0][3] > 0).all()                       # certified (1) or recomputed exhaustively (2): always exact
    assert (res[0][3] == 1).mean() > 0.98
    from oracle import c_oracle
    Q = V[:B] / np.linalg.norm(V[:B], axis=1, keepdims=True)
    sel = np.arange(0, B, max(1, B // 64))
    Qs = np.stack([orc.query_vector(V, int(u)) for u in sel])
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(X, Qs, V[sel], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(res[0][0][sel], o_idx + B) and np.array_equal(res[0][1][sel], o_sc)
    assert np.array_equal(res[0][2][sel] ** 2, o_d2) or np.allclose(res[0][2][sel], np.sqrt(o_d2), rtol=0, atol=0)


@pytest.mark.parametrize("d,N,B,k", [(128, 3706, 700, 200), (64, 1682, 943, 200), (200, 8000, 300, 50)])
def test_knn_f16_dense_screen_opt_in_is_invisible(hw, d, N, B, k):
    """Opt-in: the DENSE screen (catalogues up to 8 192 rows) on a float16 copy of the table.  Same ids / scores / distances
    as the float32 screen (the default there) and the oracle."""
    from hwer.device_index import DeviceIndex
    X = unit_rows_f32(235 + d, N, d)
    users = unit_rows_f32(236 + d, B, d)
    V = np.concatenate([users, X]).astype(np.float64)
    type_rows = {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}
    res = []
    for f16 in (True, False):
        ix = DeviceIndex(V, type_rows, arith=0)
        ix.f16_screen = f16
        ix.F16_MIN_ROWS = 0 if f16 else DeviceIndex.F16_MIN_ROWS      # hand the dense path a float16 copy as well
        res.append(ix.search("item", np.arange(B), k=k))
        assert (ix.tables["item"].X16 is not None) == f16
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert (res[0][3] > 0).all()
    for u in range(0, B, max(1, B // 8)):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(B, B + N), u, k=k)
        assert np.array_equal(res[0][0][u], o_idx) and np.array_equal(res[0][1][u], o_score) and np.array_equal(res[0][2][u], o_dist)


@pytest.mark.parametrize("N,d,n", [(5000, 40, 16), (70000, 96, 64), (300, 130, 32)])
def test_prepare_for_knn_pca_if_wider(hw, N, d, n):
    """GcnNCF.prepare_for_knn (hwer/gcn_ncf.py:447-456): PCA to n_dims when the vectors are wider, then unit rows - on the device.
    Against scikit-learn's exact PCA on the float64 input (1e-6: float32 output rows) and on the float32 input the reference
    would hand it (5e-5: scikit-learn then works in float32)."""
    from sklearn.decomposition import PCA
    from hwer import prepare_for_knn
    rng = np.random.default_rng(N + d)
    X = (rng.standard_normal((N, d)) * np.linspace(3.0, 0.2, d)).astype(np.float32)
    got = prepare_for_knn(X, n)
    assert got.dtype == np.float32 and got.shape == (N, n)
    unit = lambda a: a / np.linalg.norm(a, axis=1, keepdims=True)   # noqa: E731
    ref64 = unit(PCA(n_components=n, svd_solver="full").fit_transform(X.astype(np.float64)))
    assert np.abs(got - ref64).max() <= 1e-6
    ref32 = unit(PCA(n_components=n, svd_solver="full").fit_transform(X))
    assert np.abs(got - ref32).max() <= 5e-5
    with pytest.raises(ValueError):
        prepare_for_knn(X[:, :n - 1], n)
    same = prepare_for_knn(X[:, :n], n)                               # equal width: unit rows only
    assert np.abs(same - unit(X[:, :n].astype(np.float64))).max() <= 1e-6
