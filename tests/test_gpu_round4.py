"""GPU parity tests added in round 4: a seeded slice of tools/fuzz_parity.py inside the driver-run suite, adversarial rows for the
float16 screens (wide dynamic range, mixed norms, duplicates at the cut), the RCCL branch of hwer/distributed.py executed on one GPU
(backend nccl, world size 1).  Bit-exact for ids / scores / distances unless an assert states a tolerance."""
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


# ------------------------------------------------------------------------------------------------- seeded fuzz slice
def _fuzz_cases(n=48, seed=2026):
    """Deterministic (B, N, d, k, arith, scoring, want_dist, nonunit) cases around the thresholds between the search paths: 8 192 / 8 193
    rows (dense | streaming), 32 / 33 and 1 024 / 1 025 queries (cooperative select | wave per query; small-batch | streaming screen),
    k at both ends (1, 392 = the fused search's largest), widths that are not multiples of anything.  Work per case is bounded
    (B N d <= 1.2e9) so that the slice stays within seconds."""
    rng = np.random.default_rng(seed)
    Bs = [1, 2, 31, 32, 33, 64, 65, 255, 256, 257, 1000, 1024, 1025, 2100]
    Ns = [1, 5, 31, 32, 33, 64, 65, 1000, 4095, 8191, 8192, 8193, 8300, 12000, 20000, 40000]
    ds = [1, 3, 8, 17, 31, 32, 33, 64, 66, 100, 128, 129, 200, 256]
    ks = [1, 2, 10, 64, 100, 200, 256, 392]
    pinned = [(32, 8192, 64, 392), (33, 8193, 128, 1), (1024, 8192, 33, 200), (1025, 8193, 31, 392), (1, 8193, 256, 392), (33, 8192, 1, 1),
              (1025, 8192, 128, 1), (32, 8193, 17, 200), (1024, 8193, 64, 392), (1025, 8192, 200, 200)]
    cases = []
    for i in range(n):
        if i < len(pinned):
            B, N, d, k = pinned[i]
        else:
            while True:
                B, N, d = int(rng.choice(Bs)), int(rng.choice(Ns)), int(rng.choice(ds))
                if float(B) * N * d <= 1.2e9:
                    break
            k = int(rng.choice([x for x in ks if x <= N]))
        arith = int(rng.integers(0, 2))
        scoring = "gcn" if rng.random() < 0.2 else "base"
        want = bool(rng.integers(0, 2)) or scoring == "gcn"
        cases.append((B, N, d, k, arith, scoring, want, bool(d > 1 and rng.random() < 0.3), int(rng.integers(1 << 30))))
    return cases


@pytest.mark.parametrize("B,N,d,k,arith,scoring,want,nonunit,seed", _fuzz_cases())
def test_fuzz_slice_vs_oracle(hw, B, N, d, k, arith, scoring, want, nonunit, seed):
    """tools/fuzz_parity.py's check, one seeded case: every query of DeviceIndex.search against the C oracle's brute force - ids, scores
    (float64 flavour: bit for bit; float32 flavour: neighbour sets on every query, order and scores against the numpy oracle on three),
    distances where they were asked for; every query certified (1) or recomputed exhaustively (2)."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(seed)

    def unit(n):
        x = rng.standard_normal((n, d), dtype=np.float32).astype(np.float64)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        return x.astype(np.float32)
    V32 = np.concatenate([unit(B), unit(N)])
    if nonunit:      # non-unit rows are legal for the k-NN itself (MultiKNN.query on arbitrary vectors, recommendation_base.py:74-83)
        V32[B:] *= rng.uniform(0.5, 3.0, size=(N, 1)).astype(np.float32)
    V = V32.astype(np.float64) if arith == 0 else V32
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, arith=arith, audit=False)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k, scoring=scoring, want_dist=want)
    assert ((flags == 1) | (flags == 2)).all()
    Q64, _, _ = ix.build_queries(np.arange(B))
    A = V32[:B].astype(np.float64)
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(V32[B:], Q64.cpu().numpy(), None if scoring == "gcn" else A, k, threads=c_oracle.effective_cpus())
    if arith == 0 or scoring == "gcn":
        assert np.array_equal(idx, o_idx + B)
    else:
        assert np.array_equal(np.sort(idx, axis=1), np.sort(o_idx + B, axis=1))
        for u in rng.choice(B, size=min(B, 3), replace=False).tolist():
            n_idx, n_sc, _ = orc.find_closest(V, np.arange(B, B + N), int(u), k=k)
            assert np.array_equal(idx[u], n_idx) and np.array_equal(score[u], n_sc)
    if scoring == "gcn" and arith == 0:
        assert np.array_equal(score, (2.0 - np.sqrt(o_d2)) / 2.0)
    elif scoring != "gcn" and arith == 0:
        assert np.array_equal(score, o_sc)
    if want and (arith == 0 or scoring == "gcn"):
        assert np.array_equal(dist, np.sqrt(o_d2))


# ------------------------------------------------------------------------------------------------- adversarial rows for the float16 screens
def _heavy_tailed_rows(rng, n, d, norm_lo=1e-3, norm_hi=1.0):
    """rows whose elements span 30 binades (sign * 2^-U(0, 30)), row norms log-uniform in [norm_lo, norm_hi]"""
    x = np.exp2(-rng.uniform(0.0, 30.0, size=(n, d))) * rng.choice([-1.0, 1.0], size=(n, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    x *= np.exp(rng.uniform(np.log(norm_lo), np.log(norm_hi), size=(n, 1)))
    return x.astype(np.float32)


@pytest.mark.parametrize("B,N,d,k,tag", [(40, 12000, 128, 200, "small-batch float16 storage"), (1, 12000, 64, 100, "one query: matrix-vector passes"),
                                          (1500, 9000, 128, 200, "streaming float16 screen"), (1200, 10000, 256, 50, "streaming float16 screen, d = 256"),
                                          (3, 9000, 100, 392, "ring, padded width")])
def test_f16_screen_adversarial_rows(hw, B, N, d, k, tag):
    """The float16 screens' error bound (knn_select.hip: 1.02 * 2^-10 * ||q|| * ||x||max on top of the float32 bound) under rows the unit-norm
    suites never produce: elements spread over 30 binades (most of a row sits in float16's subnormal range after the table-wide scaling),
    row norms from 1e-3 to 1 in one table (the reference's MultiKNN takes any vectors, hwer/recommendation_base.py:64-83: the audit belongs to
    __build_knn__), heavy-tailed queries, and exact duplicates planted at every checked query's cut (rank k - 2 ... k + 2).  Through
    MultiKNN(nodes_to_idx, vectors) directly; ids, scores and distances == the oracle for every query, flags in {1, 2}, and the same
    bits with the float16 copies switched off."""
    from hwer import MultiKNN, Node
    rng = np.random.default_rng(B * 31 + N + d)
    items = _heavy_tailed_rows(rng, N, d)
    users = _heavy_tailed_rows(rng, B, d, 0.5, 1.0)
    # duplicates near the cut: for a few queries, copy the row at rank k-1 over rows chosen far down the ranking (exact ties at the cut)
    Xd, Ud = items.astype(np.float64), users.astype(np.float64)
    for b in range(0, B, max(1, B // 6)):
        q = Ud[b] / np.linalg.norm(Ud[b])
        d2 = ((Xd - q) ** 2).sum(1)
        order = np.argsort(d2, kind="stable")
        victims = order[N // 2 + 5 * b: N // 2 + 5 * b + 4]
        items[victims] = items[order[k - 1]]
        Xd[victims] = Xd[order[k - 1]]
    V = np.concatenate([users, items]).astype(np.float64)      # float32-representable float64 rows: stored as float32, float64 flavour
    nodes = [Node("user", i) for i in range(B)] + [Node("item", i) for i in range(N)]
    nodes_to_idx = {n: i for i, n in enumerate(nodes)}
    res = []
    for f16 in (True, False):
        m = MultiKNN(nodes_to_idx, V)
        m.index.use_f16_storage = f16
        m.index.f16_screen = f16
        res.append(m.index.search("item", np.arange(B), k=k))
        assert (m.index.tables["item"].X16 is not None) == f16, tag
    idx, score, dist, flags = res[0]
    assert ((flags == 1) | (flags == 2)).all(), tag
    for a, b_ in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b_), tag
    Qh = np.stack([orc.query_vector(V, u) for u in range(B)])
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(items, Qh, V[:B], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx, o_idx + B), tag
    assert np.array_equal(score, o_sc) and np.array_equal(dist, np.sqrt(o_d2)), tag
    # MultiKNN.query (arbitrary vector, ascending distance) on the same table
    got = m.query(Qh[0], "item", k=min(k, 50))
    r_idx, _, r_d2 = c_oracle.find_closest_f64(items, Qh[:1], None, min(k, 50), sort_by_score=False)
    assert [nodes_to_idx[n] for n, _ in got] == (r_idx[0] + B).tolist() and [x for _, x in got] == np.sqrt(r_d2[0]).tolist()


# ------------------------------------------------------------------------------------------------- RCCL, world size 1
def test_sharded_search_rccl_world1(hw, tmp_path):
    """The `nccl` branch of hwer/distributed.py on the one GPU of this box: RCCL initialised (backend nccl, world size 1),
    all_gather_into_tensor and all_to_all_single on DEVICE buffers (no host staging), hwer_pack_exchange -> the receive layout ->
    hwer_merge_topk_lists in place, results == the oracle - the same worker the two-rank tests run (tests/dist_worker.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_round2 import _run_workers
    _run_workers("nccl", 1, tmp_path, (20011, 64, 97, 200))
    assert "nccl world=1" in (tmp_path / "ok0").read_text()


# ------------------------------------------------------------------------------------------------- wide tables / large k: the round-4 kernels
@pytest.mark.parametrize("B,N,d", [(130, 1000, 515), (7, 129, 16), (300, 4100, 100), (64, 64, 1024)])
def test_knn_keys_kernel_vs_numpy(hw, B, N, d):
    """hwer_knn_keys_f32 (float32 MFMA, 16-column slabs, 128 x 128 tiles; ragged tiles on both sides): every key within the stated bound
    (2 d + 32) 2^-24 (||q|| ||x|| + 0.5 ||x||^2) of the float64 value; rows past n_items are not written."""
    import torch
    from hwer import native
    lib, P = native.lib(), native.ptr
    rng = np.random.default_rng(B + N + d)
    ldx = (d + 15) // 16 * 16
    n_pad = int(lib.hwer_knn_padded_rows(N))
    X = np.zeros((n_pad, ldx), np.float32)
    X[:N, :d] = rng.standard_normal((N, d)).astype(np.float32) * rng.uniform(0.1, 2.0, (N, 1)).astype(np.float32)
    Q = np.zeros((B, ldx), np.float32)
    Q[:, :d] = rng.standard_normal((B, d)).astype(np.float32)
    hn = np.full(n_pad, np.inf, np.float32)
    hn[:N] = (0.5 * (X[:N].astype(np.float64) ** 2).sum(1)).astype(np.float32)
    dev = torch.device("cuda", 0)
    Xd, Qd, hd = (torch.from_numpy(a).to(dev) for a in (X, Q, hn))
    keys = torch.full((B, n_pad), 7.0, dtype=torch.float32, device=dev)
    native.check(lib.hwer_knn_keys_f32(P(Xd), P(hd), n_pad, N, ldx, P(Qd), ldx, B, P(keys), n_pad, native.stream_ptr(dev)), "knn_keys")
    got = keys.cpu().numpy()
    ref = Q.astype(np.float64) @ X[:N].astype(np.float64).T - hn[:N].astype(np.float64)
    qn, xn = np.linalg.norm(Q.astype(np.float64), axis=1), np.linalg.norm(X[:N].astype(np.float64), axis=1)
    tol = (2 * d + 32) * 2.0 ** -24 * (qn[:, None] * xn[None, :] + 0.5 * xn[None, :] ** 2)
    assert (np.abs(got[:, :N] - ref) <= tol).all()
    assert (got[:, (N + 3) // 4 * 4:] == 7.0).all()


@pytest.mark.parametrize("B,n,k", [(50, 5000, 232), (3, 70, 70), (9, 1000, 1), (20, 26744, 1032), (4, 300, 299)])
def test_topk_rows_kernels_vs_numpy(hw, B, n, k):
    """hwer_knn_topk_rows_f32 / _f64: exactly the k largest keys (smallest distances) of every row, positions ascending, ties at the cut
    resolved towards the lowest positions; the k-th key returned.  Rows with many duplicates (quantised values) included."""
    import torch
    from hwer import native
    lib, P = native.lib(), native.ptr
    rng = np.random.default_rng(B * n + k)
    dev = torch.device("cuda", 0)
    keys = rng.standard_normal((B, n)).astype(np.float32)
    keys[1::3] = np.round(keys[1::3] * 4) / 4          # heavy ties
    keys[0, : n // 2] = keys[0, 0]
    kd = torch.from_numpy(keys).to(dev)
    out = torch.empty((B, k), dtype=torch.int64, device=dev)
    kth = torch.empty(B, dtype=torch.float32, device=dev)
    native.check(lib.hwer_knn_topk_rows_f32(P(kd), n, n, B, k, P(out), k, P(kth), native.stream_ptr(dev)), "topk32")
    got, gk = out.cpu().numpy(), kth.cpu().numpy()
    for b in range(B):
        order = np.lexsort((np.arange(n), -keys[b].astype(np.float64)))[:k]       # (key desc, position asc)
        assert np.array_equal(got[b], np.sort(order)) and gk[b] == keys[b][order[-1]]
    D = (keys.astype(np.float64) ** 2)
    Dd = torch.from_numpy(D).to(dev)
    native.check(lib.hwer_knn_topk_rows_f64(P(Dd), n, n, B, k, P(out), k, native.stream_ptr(dev)), "topk64")
    got = out.cpu().numpy()
    for b in range(B):
        order = np.lexsort((np.arange(n), D[b]))[:k]
        assert np.array_equal(got[b], np.sort(order))


@pytest.mark.parametrize("B,n_in,n_out,desc", [(40, 232, 200, False), (5, 1032, 1000, False), (3, 9000, 8500, True), (7, 200, 200, True), (2, 1, 1, False),
                                                (3, 8192, 100, False)])
def test_order_rows_kernel_vs_numpy(hw, B, n_in, n_out, desc):
    """hwer_knn_order_rows_f64: (value, id) order with ties by id = a stable sort of id-ordered input (LDS network up to 8 192 entries, rank
    sort beyond), the cut to n_out, the certificate flag, payload permutation (the score ordering of find_closest_neighbours,
    hwer/recommendation_base.py:172-173)."""
    import torch
    from hwer import native
    lib, P = native.lib(), native.ptr
    rng = np.random.default_rng(n_in * 7 + n_out)
    dev = torch.device("cuda", 0)
    vals = np.round(rng.standard_normal((B, n_in)) * 8) / 8      # many exact ties
    ids = np.stack([rng.permutation(10 * n_in)[:n_in] for _ in range(B)]).astype(np.int64)
    bound = np.sort(vals, axis=1)[:, n_out - 1] + rng.choice([0.0, 0.125], B)      # at the k-th value (not certified) or above it
    S = native.stream_ptr(dev)
    wsb = lib.hwer_knn_order_rows_workspace_bytes(B, n_in)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    vd, idd, bd = torch.from_numpy(vals).to(dev), torch.from_numpy(ids).to(dev), torch.from_numpy(bound).to(dev)
    o_v = torch.empty((B, n_out), dtype=torch.float64, device=dev)
    o_i = torch.empty((B, n_out), dtype=torch.int64, device=dev)
    fl = torch.empty(B, dtype=torch.int32, device=dev)
    if not desc:      # candidate ordering: ids given, certificate
        native.check(lib.hwer_knn_order_rows_f64(P(vd), n_in, None, 0, P(idd), n_in, B, n_in, n_out, 0, P(o_v), P(o_i), 5, None, None, None, None, None,
                                                 P(bd), P(fl), 0, P(ws) if wsb else None, wsb, S), "order")
        gv, gi, gf = o_v.cpu().numpy(), o_i.cpu().numpy(), fl.cpu().numpy()
        for b in range(B):
            order = np.lexsort((ids[b], vals[b]))[:n_out]
            assert np.array_equal(gv[b], vals[b][order]) and np.array_equal(gi[b], ids[b][order] + 5)
            assert gf[b] == int(vals[b][order[-1]] < bound[b])
    else:             # score ordering: descending, stable in the position, payloads follow
        pay_i = rng.integers(0, 1 << 40, (B, n_in)).astype(np.int64)
        pay_f = rng.standard_normal((B, n_in))
        pid, pfd = torch.from_numpy(pay_i).to(dev), torch.from_numpy(pay_f).to(dev)
        o_pi = torch.empty((B, n_out), dtype=torch.int64, device=dev)
        o_pf = torch.empty((B, n_out), dtype=torch.float64, device=dev)
        native.check(lib.hwer_knn_order_rows_f64(P(vd), n_in, None, 0, None, 0, B, n_in, n_out, 1, P(o_v), P(o_i), 0, None, P(pid), P(o_pi), P(pfd), P(o_pf),
                                                 None, None, 0, P(ws) if wsb else None, wsb, S), "order")
        for b in range(B):
            order = np.argsort(-vals[b], kind="stable")[:n_out]
            assert np.array_equal(o_v[b].cpu().numpy(), vals[b][order]) and np.array_equal(o_i[b].cpu().numpy(), order)
            assert np.array_equal(o_pi[b].cpu().numpy(), pay_i[b][order]) and np.array_equal(o_pf[b].cpu().numpy(), pay_f[b][order])


def test_large_k_beyond_the_lds_network(hw):
    """k = 9 000 of 9 500 rows (k + 32 candidates > 8 192 entries: the rank-sort path of hwer_knn_order_rows_f64), duplicates included:
    ids, scores and distances == the oracle."""
    from hwer.device_index import DeviceIndex
    U, N, d, k = 3, 9500, 24, 9000
    V32 = np.concatenate([unit_rows_f32(51, U, d), unit_rows_f32(52, N, d)])
    V32[U + 100] = V32[U + 7]
    V = V32.astype(np.float64)
    ix = DeviceIndex(V, {"user": np.arange(U), "item": np.arange(U, U + N)}, arith=0)
    idx, score, dist, flags = ix.search("item", np.arange(U), k=k)
    Qh = np.stack([orc.query_vector(V, u) for u in range(U)])
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(V32[U:], Qh, V[:U], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx, o_idx + U) and np.array_equal(score, o_sc) and np.array_equal(dist, np.sqrt(o_d2)) and (flags > 0).all()


# ------------------------------------------------------------------------------------------------- per-anchor calls: cached hipGraph
@pytest.mark.parametrize("U,N,d,k,flav", [(50, 1682, 64, 200, "f64"), (20, 3706, 128, 10, "f32"), (12, 20000, 100, 200, "f64"), (9, 500, 32, 300, "gcn")])
def test_one_anchor_graph_equals_the_general_path(hw, U, N, d, k, flav):
    """find_closest_neighbours called once per anchor (the reference's loop, hwer/validation.py:30-35) replays ONE cached hipGraph per
    (node type, k, score mode) - anchor id in pinned memory, query, search, on-device exhaustive pass, results back - and must return
    exactly what the general path returns and what the oracle says: every anchor in turn (the graph is reused), dense and streaming
    catalogues, both score flavours and the GcnNCF one, duplicates (ties at the cut -> flag 2 inside the graph), k = every row."""
    from hwer import Node, VectorIndex, validation
    V32 = np.concatenate([unit_rows_f32(61, U, d), unit_rows_f32(62, N, d)])
    V32[U + 5] = V32[U + 3]
    V32[U + N - 1] = V32[U + 3]
    V = V32 if flav in ("f32", "gcn") else V32.astype(np.float64)
    users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(N)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    if flav == "gcn":
        m.scoring = "gcn"
    m.fit(users + items, V)
    ix = m.knn.index
    per_anchor = validation.model_get_topk_knn_per_anchor(m, users, "item") if k == 200 else {u: m.find_closest_neighbours("item", u, k=k) for u in users}
    assert any(key[:3] == ("item", k, ix.arith if flav != "gcn" else 2) for key in ix.__dict__.get("_one_anchor", {}))       # the graph path was taken
    if flav == "gcn":     # GcnNCF scoring: one anchor or many, search() returns the distances with the scores
        one = ix.search("item", np.asarray([3]), k=k, scoring="gcn", want_dist=False)
        many = ix.search("item", np.asarray([3, 4]), k=k, scoring="gcn", want_dist=False)
        assert one[2] is not None and np.array_equal(one[2][0], many[2][0]) and np.array_equal(one[0][0], many[0][0])
    ix.one_anchor_graphs = False
    for j, u in enumerate(users):
        general = m.find_closest_neighbours("item", u, k=k)
        assert per_anchor[u] == general, j
        o_idx, o_sc, _ = orc.find_closest(V, np.arange(U, U + N), j, k=k, scoring="gcn" if flav == "gcn" else "base")
        assert [m.nodes_to_idx[n] for n, _ in general] == o_idx.tolist() and [s for _, s in general] == o_sc.tolist()


# ------------------------------------------------------------------------------------------------- ping-pong float16 streaming screen
@pytest.mark.parametrize("d,N,B,k,note", [(128, 30000, 2100, 200, "TQ = 1, item splits, odd number of 128-query blocks"), (64, 20000, 3000, 50, "d = 64"),
                                           (256, 16000, 1500, 100, "d = 256: 4 pieces per wave"), (100, 12000, 1100, 10, "padded width"),
                                           (128, 70000, 2500, 200, "TQ = 2: 64 queries per wave, 512 per workgroup"),
                                           (128, 9000, 70000, 200, "one list per query, TQ = 2"), (128, 66000, 2048, 392, "k = 392"),
                                           (32, 9000, 1300, 20, "d = 32: no float16 tiles - the float32 kernel under a ping-pong plan")])
def test_pingpong_screen_equals_the_two_workgroup_screen(hw, monkeypatch, d, N, B, k, note):
    """screen16_pp_kernel (round 4: one 512-thread workgroup per CU, SIMD partners alternating MFMA and filter segments, 3-slot tile
    ring) against screen_kernel<.., F16> (rounds 1-3; HWER_KNN_NO_PINGPONG): ids, scores, distances and flags identical for every
    query, and == the oracle on a sample; duplicates planted at a few queries' cut; a ragged last query block and a ragged last tile."""
    from hwer.device_index import DeviceIndex
    X = unit_rows_f32(235 + d, N, d)
    users = unit_rows_f32(236 + d, B, d)
    X[N // 3] = X[7]
    X[N - 1] = X[7]
    V = np.concatenate([users, X]).astype(np.float64)
    type_rows = {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}
    res = []
    for no_pp in ("", "1"):
        if no_pp:
            monkeypatch.setenv("HWER_NO_PINGPONG", "1")
        else:
            monkeypatch.delenv("HWER_NO_PINGPONG", raising=False)
        ix = DeviceIndex(V, type_rows, arith=0)
        res.append(ix.search("item", np.arange(B), k=k))
    for a, b_ in zip(res[0], res[1]):
        assert np.array_equal(a, b_), note
    assert (res[0][3] > 0).all()
    sel = np.unique(np.linspace(0, B - 1, 96).astype(np.int64))
    Qs = np.stack([orc.query_vector(V, int(u)) for u in sel])
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(X, Qs, V[sel], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(res[0][0][sel], o_idx + B) and np.array_equal(res[0][1][sel], o_sc) and np.array_equal(res[0][2][sel], np.sqrt(o_d2)), note


# ------------------------------------------------------------------------------------------------- float metrics on the device
@pytest.mark.parametrize("B,V,seed", [(300, 211, 1), (30000, 20011, 2), (5, 1, 3), (900, 129, 4)])
def test_rank_metrics_kernel_equals_the_numpy_expressions(hw, B, V, seed):
    """hwer_rank_metrics (round 4): mean recall@c and mean binary NDCG@c over the validation users from the hit masks of
    hwer_rank_filter_seen, against the host statement of the reference's numpy expressions (hwer/utils.py:101-121, hwer/validation.py:167-172:
    validation.recall_rows / binary_ndcg_rows, themselves held to the reference's per-user loops by tests/test_host_api.py) - BIT-identical:
    ragged prediction lengths (0 ... 100), users without truth, more than 128 users (the pairwise recursion of np.mean), cuts 100 / 10 / 20."""
    import ctypes
    import torch
    from hwer import native, validation
    lib, P = native.lib(), native.ptr
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda", 0)
    cmax, cuts = 100, (100, 10, 20)
    hit_words = (cmax + 63) // 64
    hits = (rng.integers(0, 1 << 62, (B, hit_words)).astype(np.uint64) & rng.integers(0, 1 << 62, (B, hit_words)).astype(np.uint64)
            & rng.integers(0, 1 << 62, (B, hit_words)).astype(np.uint64))                        # ~1/8 of the positions are hits
    n_pred_all = rng.integers(0, cmax + 1, B).astype(np.int32)
    n_pred_all[::7] = cmax
    vrow = np.sort(rng.choice(B, V, replace=False)).astype(np.int64)
    n_true = rng.integers(0, 40, V).astype(np.int64)
    hitbits = validation._unpack_hit_bits(hits[vrow], cmax)
    n_pred = n_pred_all[vrow].astype(np.int64)
    # (a hit beyond a user's prediction length cannot occur in the kernel's input: mask them as hwer_rank_filter_seen would)
    keep = (np.arange(cmax)[None, :] < n_pred_all[:, None])
    bits_all = validation._unpack_hit_bits(hits, cmax) * keep
    hits = np.packbits(np.pad(bits_all, ((0, 0), (0, hit_words * 64 - cmax))).astype(np.uint8), axis=1, bitorder="little").view(np.uint64).reshape(B, hit_words)
    hitbits = validation._unpack_hit_bits(hits[vrow], cmax)
    log2_tab = np.zeros((cmax + 1, cmax))
    for L in range(1, cmax + 1):
        log2_tab[L, :L] = np.log2(np.arange(L) + 2)      # numpy's values for a list of L predictions (vector body vs scalar remainder may differ in the last place)
    idcg_tab = np.asarray([np.sum((np.power(2, [1] * m) - 1) / np.log2(np.arange(m) + 2)) for m in range(cmax + 1)], dtype=np.float64)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)       # noqa: E731
    d_hits, d_np, d_vr, d_nt, d_l2, d_id = T(hits.view(np.int64)), T(n_pred_all), T(vrow), T(n_true), T(log2_tab), T(idcg_tab)
    vals = torch.empty(2 * len(cuts) * V, dtype=torch.float64, device=dev)
    means = torch.empty(2 * len(cuts), dtype=torch.float64, device=dev)
    cuts_h = np.asarray(cuts, dtype=np.int32)
    native.check(lib.hwer_rank_metrics(P(d_hits), hit_words, P(d_np), P(d_vr), P(d_nt), V, cuts_h.ctypes.data_as(ctypes.c_void_p), len(cuts), P(d_l2), cmax, P(d_id),
                                       P(vals), P(means), native.stream_ptr(dev)), "rank_metrics")
    got = means.cpu().numpy()
    for j, c in enumerate(cuts):
        assert got[j] == validation.recall_rows(hitbits, n_pred, n_true, c), ("recall", c)
        assert got[len(cuts) + j] == validation.binary_ndcg_rows(hitbits, n_pred, n_true, c), ("ndcg", c)


# ------------------------------------------------------------------------------------------------- round 4, second half: the select's new building blocks
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wave_primitives_selftest(hw, seed):
    """hwer_selftest_wave_ops: the DPP prefix scan, the one-key-per-lane register sort, the lower bound in a sorted run and the DPP max / min /
    float64 sum (csrc/hwer_common.h) on 512 waves of seeded lane values, each checked on the host against a plain loop (the float64 sum against
    the butterfly order the kernels' bit-exact results rely on)."""
    from hwer import native
    native.check(native.lib().hwer_selftest_wave_ops(512, seed), "selftest_wave_ops")


def _tie_heavy_rows(rng, n, d, kind):
    """item tables that stress the histogram rank search of the dense select (knn_select.hip): 'dups' = 20 prototypes repeated (exact ties fill
    single bins: more than 12 keys per bin -> the counting probes take over, then the certificate fails -> exhaustive pass); 'clusters' = the same
    prototypes + 1e-4 noise (hundreds of distinct keys inside one 1/4096 of the key range); 'outlier' = unit rows + ONE row of norm 40 (the key range
    is set by it: every other key lands in a handful of bins)"""
    if kind == "outlier":
        x = rng.standard_normal((n, d))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        x[n // 3] *= 40.0
        return x.astype(np.float32)
    proto = rng.standard_normal((20, d))
    proto /= np.linalg.norm(proto, axis=1, keepdims=True)
    x = proto[rng.integers(0, 20, n)]
    if kind == "clusters":
        x = x + 1e-4 * rng.standard_normal((n, d))
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


@pytest.mark.parametrize("B,N,d,k,kind", [(64, 2000, 64, 200, "dups"), (64, 2000, 64, 200, "clusters"), (300, 1682, 64, 200, "outlier"), (5, 8192, 128, 392, "clusters"),
                                          (2500, 1500, 32, 50, "clusters"), (2500, 2048, 64, 200, "outlier"), (1, 700, 16, 1, "dups"), (40, 4000, 128, 100, "dups")])
def test_dense_select_rank_search_under_ties_and_clusters(hw, B, N, d, k, kind):
    """The dense select's two-round histogram bracket (round 4) where it cannot finish by itself: exact duplicates, tight clusters, a key range
    blown up by one outlier - one query per workgroup (B <= 2 048) and the wave-per-query form (rows up to 2 048 keys), float64 flavour with
    distances: ids, scores and distances == the C oracle for every query, flags in {1, 2}."""
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(B * 7 + N + d + len(kind))
    items = _tie_heavy_rows(rng, N, d, kind)
    users = rng.standard_normal((B, d))
    users /= np.linalg.norm(users, axis=1, keepdims=True)
    users = users.astype(np.float32)
    if kind != "outlier":
        users[::3] = items[rng.integers(0, N, len(users[::3]))]      # queries that ARE catalogue rows: distance 0 to a whole cluster
    V = np.concatenate([users, items]).astype(np.float64)
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, arith=0, audit=False)
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k, want_dist=True)
    assert ((flags == 1) | (flags == 2)).all()
    Q64, _, _ = ix.build_queries(np.arange(B))
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(items, Q64.cpu().numpy(), V[:B], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx, o_idx + B)
    assert np.array_equal(score, o_sc) and np.array_equal(dist, np.sqrt(o_d2))


@pytest.mark.parametrize("B,N,d,k", [(1100, 9000, 256, 100), (1500, 9000, 128, 200), (2100, 8200, 64, 50)])
def test_streaming_screen_with_queries_of_mixed_magnitudes(hw, B, N, d, k):
    """The float16 streaming screen's per-wave key exponent (round 4: one query subtile per wave rescales its fragments to the smallest exponent of
    its 32 queries and sets the accumulators up from pre-scaled half norms; a wave whose exponents spread over more than 10 binades keeps the
    per-lane scaling) and the folded threshold, under queries the unit-norm callers never produce: norms log-uniform over 1e-6 ... 1, shuffled so
    that every 32-query block mixes them - DeviceIndex.search_queries on arbitrary query rows (the C ABI takes any Q), GcnNCF-style distance
    scoring: ids, scores and distances == the C oracle for every query, flags in {1, 2}."""
    import torch
    from hwer.device_index import DeviceIndex
    rng = np.random.default_rng(B + N + d)
    items = unit_rows_f32(3, N, d)
    Q = rng.standard_normal((B, d))
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    Q *= np.exp(rng.uniform(np.log(1e-6), 0.0, size=(B, 1)))
    Q[::5] = items[rng.integers(0, N, len(Q[::5]))]          # some queries ARE catalogue rows (distance 0)
    Q32 = Q.astype(np.float32)
    Q64 = Q32.astype(np.float64)
    V = np.concatenate([np.zeros((1, d), np.float32), items]).astype(np.float64)
    V[0, 0] = 1.0
    ix = DeviceIndex(V, {"user": np.arange(1, dtype=np.int64), "item": np.arange(1, 1 + N, dtype=np.int64)}, arith=0, audit=False)
    dev = ix.device
    idx, score, dist2, flags = ix.search_queries("item", torch.from_numpy(Q64).to(dev), torch.from_numpy(np.ascontiguousarray(Q32)).to(dev), k, score_mode=2)
    flags = flags.cpu().numpy()
    assert ((flags == 1) | (flags == 2)).all()
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(items, Q64, None, k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx.cpu().numpy(), o_idx + 1)
    assert np.array_equal(dist2.cpu().numpy(), o_d2)
    assert np.array_equal(score.cpu().numpy(), (2.0 - np.sqrt(o_d2)) / 2.0)
