"""The oracle (oracle/hwer_oracle.py) against golden vectors captured from the unmodified reference
(tools/capture_goldens.py).  CPU only.  These pin the oracle; the GPU parity tests then compare the
HIP path with the oracle."""
import hashlib

import numpy as np
import pytest

from oracle import hwer_oracle as orc
from conftest import g12_scenario, unit_rows_f32


def _flav(V32, f):
    return V32.astype(np.float64) if f == "f64" else V32


@pytest.mark.parametrize("flav", ["f64", "f32"])
@pytest.mark.parametrize("k", [1, 10, 200])
def test_g1_knn_basic(golden, flav, k):
    g = golden("g1_knn_basic")
    U, I = int(g["n_users"]), int(g["n_items"])
    V = _flav(g["V32"], flav)
    rows = np.arange(U, U + I)
    for u in range(U):
        idx, score, _ = orc.find_closest(V, rows, u, k=k)
        assert np.array_equal(idx, g[f"idx_{flav}_k{k}"][u])
        assert score.dtype == V.dtype
        assert np.array_equal(score, g[f"score_{flav}_k{k}"][u])  # bit-exact


@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_g5_knn_query_distances_and_gcn_scoring(golden, flav):
    g = golden("g1_knn_basic")
    U, I = int(g["n_users"]), int(g["n_items"])
    V = _flav(g["V32"], flav)
    rows = np.arange(U, U + I)
    for k in (10, 200):
        for u in range(0, U, 7):
            q = orc.average_embedding(V, [u])
            local, dist = orc.knn_bruteforce(q, V[rows], k)
            assert np.array_equal(rows[local], g[f"knnq_idx_{flav}_k{k}"][u])
            assert np.array_equal(dist, g[f"knnq_dist_{flav}_k{k}"][u])  # sklearn's distances, bit-exact
            # GcnNCF scoring restated from hwer/gcn_ncf.py:379-383 on the reference's own distances
            idx, score, _ = orc.find_closest(V, rows, u, k=k, scoring="gcn")
            ref = (-1 * g[f"knnq_dist_{flav}_k{k}"][u] + 2) / 2
            assert np.array_equal(score, ref) and np.array_equal(idx, g[f"knnq_idx_{flav}_k{k}"][u])
    # user -> user returns the anchor first, score 1.0
    idx, score, _ = orc.find_closest(V, np.arange(U), 3, k=5)
    assert np.array_equal(idx, g[f"u2u_idx_{flav}"]) and np.array_equal(score, g[f"u2u_score_{flav}"])
    assert idx[0] == 3 and abs(score[0] - 1.0) < 1e-6


def test_g2_ml100k_shape(golden):
    g = golden("g2_knn_ml100k_shape")
    U, I, d, k = int(g["n_users"]), int(g["n_items"]), int(g["d"]), int(g["k"])
    V32 = np.concatenate([unit_rows_f32(int(g["seed_users"]), U, d), unit_rows_f32(int(g["seed_items"]), I, d)])
    assert hashlib.sha256(V32.tobytes()).hexdigest() == str(g["input_sha256"])
    V = V32.astype(np.float64)
    rows = np.arange(U, U + I)
    idx = np.empty((U, k), np.int64)
    sc = np.empty((U, k), np.float64)
    for u in range(U):
        idx[u], sc[u], _ = orc.find_closest(V, rows, u, k=k)
    assert np.array_equal(idx - U, g["idx"].astype(np.int64))
    assert hashlib.sha256(idx.tobytes()).hexdigest() == str(g["idx_sha256"])
    assert hashlib.sha256(sc.tobytes()).hexdigest() == str(g["score_sha256"])
    assert np.array_equal(sc[g["score_rows"]], g["score_sel"])


@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_g3_posneg(golden, flav):
    g = golden("g3_posneg")
    U, I, k = int(g["n_users"]), int(g["n_items"]), int(g["k"])
    V = _flav(g["V32"], flav)
    rows = np.arange(U, U + I)
    for a in range(32):
        pos = [int(x) for x in g["pos"][a] if x != -2]
        neg = [int(x) for x in g["neg"][a] if x != -2]
        q = orc.query_vector(V, a, np.asarray(pos, np.int64), np.asarray(neg, np.int64))
        assert np.array_equal(q, g[f"q_{flav}"][a])
        idx, score, _ = orc.find_closest(V, rows, a, np.asarray(pos, np.int64), np.asarray(neg, np.int64), k=k)
        assert np.array_equal(idx, g[f"idx_{flav}"][a])
        assert np.array_equal(score, g[f"score_{flav}"][a])
    # the query is not re-normalised (hwer/recommendation_base.py:170)
    norms = np.linalg.norm(g["q_f64"], axis=1)
    assert norms.min() < 0.9


@pytest.mark.parametrize("flav", ["f64", "f32"])
def test_g4_predict(golden, flav):
    g = golden("g4_predict")
    V = _flav(g["V32"], flav)
    r = orc.pair_score(V, g["src"], g["dst"])
    assert r.dtype == V.dtype
    assert np.array_equal(r, g[f"score_{flav}"])
    unk = (g["src"] == -1) | (g["dst"] == -1)
    assert unk.sum() > 100 and np.all(np.abs(r[unk] - 0.5) < 1e-4)  # unknown nodes score ~0.5
    # scalar restatement of numpy's summation order
    E1, E2 = orc.gather_rows(V, g["src"][:64]), orc.gather_rows(V, g["dst"][:64])
    mine = np.array([orc.pairwise_sum(E1[i] * E2[i]) for i in range(64)], dtype=V.dtype)
    assert np.array_equal((mine + 1) / 2, r[:64])


def test_g6_content_tail(golden):
    g = golden("g6_content_tail")
    A, final = orc.content_tail(g["E"], g["src"], g["dst"])
    assert np.array_equal(A, g["adjacent"])          # diff 0.0
    assert np.array_equal(final, g["final"])
    # generalised blend, 2-table alpha=(.5,.5) special case reproduces :85 + :96 exactly
    b = orc.blend_normalize([g["E"], g["adjacent"]], [None, None], np.array([0.5, 0.5]))
    assert np.array_equal(b, g["final"])
    # adjacency: every node lists itself twice, duplicates kept
    indptr, indices = orc.adjacency_csr(300, g["src"], g["dst"])
    assert indptr[-1] == 2 * (len(g["src"]) + 300)
    for n in (0, 17, 299):
        assert (indices[indptr[n]:indptr[n + 1]] == n).sum() >= 2
    # retrieval on the fitted vectors
    rows = np.nonzero(g["node_type_id"] == 1)[0]
    idx, score, _ = orc.find_closest(g["final"], rows, 0, k=7)
    assert np.array_equal(idx, g["knn_idx"]) and np.array_equal(score, g["knn_score"])


def test_g7_content_fit_tail_consistency(golden):
    g = golden("g7_content_fit_e2e")
    final = g["final"]
    assert final.shape == (18, 6)
    assert np.allclose(np.linalg.norm(final, axis=1), 1.0, atol=1e-12)
    idx, score, _ = orc.find_closest(final, np.arange(3, 18), 0, k=3)
    assert np.array_equal(idx, g["knn_idx"]) and np.array_equal(score, g["knn_score"])


def test_g15_content_fit_6k(golden):
    """G15: the reference's ContentRecommendation.fit on 6 000 nodes.  The float64 restatement of hwer/content_recommender.py:77-96 on
    the captured table reproduces `final` bit for bit (sha256 of all 6 000 x 12 values), and the oracle's k-NN on those rows returns the
    reference's 200 neighbours and scores for every one of the 1 000 users."""
    import hashlib
    from conftest import g15_scenario
    g = golden("g15_content_fit_6k")
    sc = g15_scenario(g)
    _, final = orc.content_tail(sc["E"], sc["src"], sc["dst"])
    assert hashlib.sha256(np.ascontiguousarray(final).tobytes()).hexdigest() == str(g["final_sha"])
    assert np.array_equal(final[g["final_rows"]], g["final_sample"])
    k = sc["k"]
    scores = np.empty((len(sc["users"]), k))
    for i, u in enumerate(sc["users"]):
        idx, s_, _ = orc.find_closest(final, sc["items"], int(u), k=k)
        assert np.array_equal(idx, g["knn_idx"][i].astype(np.int64)), i
        scores[i] = s_
    assert hashlib.sha256(scores.tobytes()).hexdigest() == str(g["score_sha"])
    assert np.array_equal(scores[::125], g["score_sample"])


def test_g8_unit_audit(golden):
    g = golden("g8_unit_audit")
    for name in ("rows", "rows_zero", "rows_nan"):
        got = np.asarray(orc.unit_length_violations(g[name], axis=1), dtype=np.float64)
        assert np.array_equal(got, g["viol_" + name], equal_nan=True)
        got32 = np.asarray(orc.unit_length_violations(g[name].astype(np.float32), axis=1), dtype=np.float64)
        assert np.array_equal(got32, g["viol32_" + name], equal_nan=True)
    assert g["viol_rows"][0] == 3        # 1+2e-4, 1-2e-4, 1-1.01e-4
    assert g["viol_rows_nan"][0] == 3    # a NaN norm passes both comparisons
    assert np.array_equal(orc.unit_length(g["rows_zero"], axis=1), g["unit_zero"], equal_nan=True)
    assert np.isnan(g["unit_zero"][5]).all()
    assert list(g["build_ok"]) == [1, 0, -1]


def test_g9_metrics(golden):
    g = golden("g9_metrics")
    yt = dict(zip(g["y_true_keys"].tolist(), g["y_true_vals"].tolist()))
    yp = g["y_pred"].tolist()
    assert orc.recall(yt, yp) == float(g["recall"])
    assert orc.ndcg(yt, yp) == float(g["ndcg"])
    assert orc.binary_ndcg(yt, yp) == float(g["binary_ndcg"])
    assert orc.binary_ndcg_v2(list(yt), yp) == float(g["binary_ndcg_v2"])
    assert orc.reciprocal_rank(list(yt), yp) == float(g["reciprocal_rank"])
    assert orc.recall({}, yp) == float(g["recall_empty"]) == 0.0
    assert orc.reciprocal_rank([99], yp) == float(g["rr_miss"]) == 0.0
    assert orc.recall(yt, [3]) == float(g["recall_short"]) == 1.0
    # full evaluation loop
    U, I = int(g["ee_n_users"]), int(g["ee_n_items"])
    V = g["ee_V32"].astype(np.float64)
    rows = np.arange(U, U + I)
    train = [(int(u), int(i), float(r)) for u, i, r in g["ee_train"]]
    valid = [(int(u), int(i), float(r)) for u, i, r in g["ee_valid"]]
    users = sorted(set(u for u, _, _ in train + valid))
    preds = {u: orc.find_closest(V, rows, u, k=200)[0].tolist() for u in users}
    m = orc.retrieval_metrics(preds, train, valid, I_items_seen(train, valid))
    for key in ("recall@100", "ndcg_b@100", "ndcg_b@10", "recall@10", "diversity"):
        assert m[key] == pytest.approx(float(g["ee_" + key.replace("@", "_at_")]), rel=0, abs=1e-15), key
    assert 0.0 <= float(g["ee_ncf_hr"]) <= 1.0


def I_items_seen(train, valid):
    return len(set(i for _, i, _ in train) | set(i for _, i, _ in valid))


def test_g10_ties_as_sets(golden):
    g = golden("g10_ties")
    U, I = int(g["n_users"]), int(g["n_items"])
    V = g["V32"].astype(np.float64)
    rows = np.arange(U, U + I)
    saw_tie = False
    for u in range(U):
        idx, score, _ = orc.find_closest(V, rows, u, k=10)
        assert np.array_equal(score, g["score"][u])
        ref = g["idx"][u]
        if not np.array_equal(idx, ref):
            saw_tie = True
        # compare as sets inside equal-score groups
        for s in np.unique(score):
            sel = score == s
            assert set(idx[sel].tolist()) == set(ref[g["score"][u] == s].tolist())
            # the build's rule inside a group of identical rows: ascending id
            if sel.sum() > 1:
                assert list(idx[sel]) == sorted(idx[sel])
    assert saw_tie  # the KD-tree's tie order is NOT index order


def test_g11_errors(golden):
    g = golden("g11_errors")
    assert str(g["k_too_large"]).startswith("ValueError:k must be less than or equal to the number of training points")
    assert str(g["unknown_anchor"]) == "NodeNotFoundException"
    assert str(g["bad_type"]) == str(g["not_fitted"]) == str(g["dup_nodes"]) == str(g["bad_node_type"]) == "AssertionError"
    assert str(g["node_eq_int_str"]) == "True"
    m = orc.OracleRecommender({"user", "item"})
    users = [orc.Node("user", i) for i in range(4)]
    items = [orc.Node("item", i) for i in range(20)]
    m.add_nodes(users + items)
    V = np.concatenate([unit_rows_f32(1011, 4, 8), unit_rows_f32(2011, 20, 8)]).astype(np.float64)
    m.build_knn(V)
    with pytest.raises(ValueError, match="k must be less than or equal"):
        m.find_closest_neighbours("item", users[0], k=21)
    with pytest.raises(orc.NodeNotFoundException):
        m.find_closest_neighbours("item", orc.Node("user", "zz"))
    with pytest.raises(AssertionError):
        m.find_closest_neighbours("movie", users[0])
    with pytest.raises(AssertionError):
        m.add_nodes([orc.Node("user", 0)])
    assert orc.Node("item", 5) == orc.Node("item", "5")
    assert len(m.find_closest_neighbours("item", users[0], k=3)) == 3


# ---- the C part of the oracle (oracle/hwer_oracle.c) against the numpy oracle and the goldens ---------------------
def test_c_oracle_matches_golden_and_numpy(golden):
    from oracle import c_oracle
    g = golden("g1_knn_basic")
    U, I = int(g["n_users"]), int(g["n_items"])
    V32 = g["V32"]
    V = V32.astype(np.float64)
    Q = np.stack([orc.query_vector(V, u) for u in range(U)])
    for k in (1, 10, 200):
        idx, sc, d2 = c_oracle.find_closest_f64(V32[U:], Q, V[:U], k, threads=2)
        assert np.array_equal(idx + U, g[f"idx_f64_k{k}"])
        assert np.array_equal(sc, g[f"score_f64_k{k}"])
        # the row-blocked loop (default) and the one-pair-at-a-time restatement: the same bits
        for a, b in zip((idx, sc, d2), c_oracle.find_closest_f64(V32[U:], Q, V[:U], k, threads=2, scalar=True)):
            assert np.array_equal(a, b)
    idx, sc, d2 = c_oracle.find_closest_f64(V32[U:], Q, None, 10, sort_by_score=False)
    assert np.array_equal(idx + U, g["knnq_idx_f64_k10"]) and np.array_equal(np.sqrt(d2), g["knnq_dist_f64_k10"])
    g4 = golden("g4_predict")
    assert np.array_equal(c_oracle.pair_score_f64(g4["V32"], g4["src"], g4["dst"], threads=2), g4["score_f64"])
    for d in (5, 66, 130, 256, 300):
        Vd = unit_rows_f32(d, 300, d)
        rng = np.random.default_rng(d)
        s, t = rng.integers(-1, 300, 500), rng.integers(-1, 300, 500)
        assert np.array_equal(c_oracle.pair_score_f64(Vd, s, t), orc.pair_score(Vd.astype(np.float64), s, t))
        Qd = Vd[:7].astype(np.float64) * 0.9
        idx, sc, d2 = c_oracle.find_closest_f64(Vd[10:], Qd, Vd[:7].astype(np.float64), 20)
        for b in range(7):
            loc, dist = orc.knn_bruteforce(Qd[b], Vd[10:].astype(np.float64), 20)
            score = ((Vd[:7].astype(np.float64)[b] * Vd[10:].astype(np.float64)[loc]).sum(1) + 1) / 2
            order = np.argsort(-score, kind="stable")
            assert np.array_equal(idx[b], loc[order]) and np.array_equal(sc[b], score[order]) and np.array_equal(d2[b], (dist ** 2)[order]) or np.allclose(np.sqrt(d2[b]), dist[order], rtol=0, atol=0)


def test_reference_shaped_loop_agrees_with_oracle():
    """oracle.reference_shaped_sweep (scikit-learn KD-tree + per-anchor loop: the bench's secondary CPU baseline) returns
    the oracle's neighbours and scores on tie-free data."""
    pytest.importorskip("sklearn")
    rng = np.random.default_rng(11)
    V = rng.standard_normal((400, 24))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    rows = np.arange(30, 400)
    idx, sc = orc.reference_shaped_sweep(V, rows, list(range(8)), k=25)
    for u in range(8):
        o_idx, o_sc, _ = orc.find_closest(V, rows, u, k=25)
        assert np.array_equal(idx[u], o_idx) and np.array_equal(sc[u], o_sc)


@pytest.mark.parametrize("tag", ["small", "multi"])
def test_g12_fit_feature_table(golden, tag):
    """Fit-side table construction + IncrementalPCA restatement against the reference (one batch / two batches)."""
    g = golden("g12_fit_pca")
    sc = g12_scenario(g, tag)
    blocks = [orc.unit_length(f, axis=1) if has else np.zeros((f.shape[0], 1)) for f, has in zip(sc["feats"], sc["with_data"])]
    T = len(blocks)
    width = sum(b.shape[1] for b in blocks)
    # the table the reference handed to IncrementalPCA (block columns in the reference's set order)
    table = np.zeros((sc["N"], width + T))
    c = 0
    for t in sc["order"]:
        table[np.flatnonzero(sc["type_id"] == t), c:c + blocks[t].shape[1]] = blocks[t]
        c += blocks[t].shape[1]
    table[np.arange(sc["N"]), width + sc["type_id"]] = 1.0
    assert hashlib.sha256(np.ascontiguousarray(table).tobytes()).hexdigest() == str(g[tag + "_table_sha"])
    scores, comps, mean, sv = orc.incremental_pca_fit_transform(table, sc["n_dims"])
    assert len(orc.pca_batches(sc["N"], 2 ** 16, sc["n_dims"])) == (1 if tag == "small" else 2)
    assert np.abs(comps - g[tag + "_components"]).max() <= 1e-12
    assert np.abs(mean - g[tag + "_mean"]).max() <= 1e-15
    assert np.abs(sv - g[tag + "_singular_values"]).max() <= 1e-9
    rows = g[tag + "_rows"]
    assert np.abs(scores[rows] - g[tag + "_pca_out"]).max() <= 1e-12
    E, _ = orc.fit_feature_table(blocks, sc["type_id"], sc["n_dims"], sc["order"])
    assert np.abs(E[rows, :sc["n_dims"]] - g[tag + "_pca_unit"]).max() <= 1e-12
    _, final = orc.content_tail(E, sc["src"], sc["dst"])
    assert np.abs(final[rows] - g[tag + "_final"]).max() <= 1e-12
    # the column order of the blocks (the reference's set order) only permutes the table: same scores up to rounding
    E2, _ = orc.fit_feature_table(blocks, sc["type_id"], sc["n_dims"], None)
    assert np.abs(E2 - E).max() <= 1e-9


def test_g12_pca_errors():
    X = np.random.default_rng(0).standard_normal((5, 4))
    with pytest.raises(ValueError):
        orc.incremental_pca_fit_transform(X, 5)          # more components than columns
    with pytest.raises(ValueError):
        orc.incremental_pca_fit_transform(X[:3], 4)      # more components than rows of the first batch


def g14_inputs():
    """tools/capture_goldens.py g14_inputs: genuine float64 unit rows regenerated from the seed"""
    U, I, d = 64, 4096, 64
    x = np.random.default_rng(1400).standard_normal((U + I, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x, U, I, d


def test_g14_float64_rows(golden):
    """The reference fed GENUINE float64 rows (not float32 values widened): the oracle restatement reproduces its neighbours,
    scores, KD-tree distances, predict values and positive / negative-list queries bit for bit."""
    import hashlib
    g = golden("g14_float64_rows")
    V, U, I, d = g14_inputs()
    assert hashlib.sha256(np.ascontiguousarray(V).tobytes()).hexdigest() == str(g["input_sha256"])
    k = int(g["k"])
    rows = np.arange(U, U + I)
    for u in (0, 1, 17, 63):
        idx, sc, _ = orc.find_closest(V, rows, u, k=k)
        assert np.array_equal(idx - U, g["idx"][u].astype(np.int64)) and np.array_equal(sc, g["score"][u])
    for u in range(8):
        q = orc.query_vector(V, u)
        local, dist = orc.knn_bruteforce(q, V[rows], k)
        assert np.array_equal(rows[local], g["query_idx"][u]) and np.array_equal(dist, g["query_dist"][u])
    assert np.array_equal(orc.pair_score(V, g["pair_src"], g["pair_dst"]), g["pair_score"])
    po, no = np.r_[0, np.cumsum(g["pn_pos_len"])], np.r_[0, np.cumsum(g["pn_neg_len"])]
    for a in range(16):
        idx, sc, _ = orc.find_closest(V, rows, a, positive=g["pn_pos"][po[a]:po[a + 1]], negative=g["pn_neg"][no[a]:no[a + 1]], k=50)
        assert np.array_equal(idx, g["pn_idx"][a]) and np.array_equal(sc, g["pn_score"][a])


@pytest.mark.parametrize("B,N,d,k", [(70, 3001, 128, 200), (33, 1000, 37, 10), (5, 31, 1, 31), (3, 33, 300, 33), (65, 97, 64, 97)])
def test_c_oracle_blocked_equals_scalar(B, N, d, k):
    """orc_find_closest_f64 (32 rows x 32 queries in flight, SIMD lanes) against orc_find_closest_f64_scalar (one pair at a time): ids,
    scores, squared distances bit for bit - ragged row blocks and query groups, duplicate rows (ties -> ascending row id), both output
    orders, both score flavours."""
    from oracle import c_oracle
    rng = np.random.default_rng(B * 7 + N)
    X = rng.standard_normal((N, d)).astype(np.float32)
    X[N // 2] = X[N // 3]
    X[-1] = X[0]
    Q, A = rng.standard_normal((B, d)), rng.standard_normal((B, d))
    for anchors in (A, None):
        for sbs in (True, False):
            a = c_oracle.find_closest_f64(X, Q, anchors, k, sbs, threads=3)
            b = c_oracle.find_closest_f64(X, Q, anchors, k, sbs, threads=2, scalar=True)
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
