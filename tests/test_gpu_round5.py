"""GPU parity tests added in round 5: the BLEND stage in the reference's own float64 arithmetic (hwer_segment_mean_f64,
hwer_content_tail_f64, hwer_blend_normalize_f64) - `final` of ContentRecommendation.fit identical to the reference bit for bit on the
captured table (G6, G15), the reference's neighbours from fit() end to end at 6 000 nodes (G15), the general T-table blend against the
numpy oracle.  Bit-exact unless an assert states a tolerance."""
import hashlib
import os

import numpy as np
import pytest

from conftest import g15_scenario, unit_rows_f32
from oracle import hwer_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hw():
    import hwer
    from hwer import native
    native.lib()
    return hwer


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------------- float64 blend stage
def test_content_tail_f64_golden_g6(golden, hw):
    """hwer/content_recommender.py:77-86 + :96 on the table the reference's fit() saw (captured E, float64): neighbour mean, (E + A)/2,
    unit rows - every bit of `adjacent` and `final` as the reference computed them."""
    from hwer.content_recommender import adjacency_csr, blend_tables, content_tail, segment_mean
    g = golden("g6_content_tail")
    E = g["E"]
    assert E.dtype == np.float64
    got = content_tail(E, g["src"], g["dst"])
    assert got.dtype == np.float64 and np.array_equal(got, g["final"])
    indptr, indices = adjacency_csr(300, g["src"], g["dst"])
    A = segment_mean(E, indptr, indices)
    assert A.dtype == np.float64 and np.array_equal(A, g["adjacent"])
    # the general blend with T = (E, A), alpha = (.5, .5) is the same numbers ((E + A)/2 == .5 E + .5 A exactly)
    b, nrm = blend_tables([E, g["adjacent"]], np.array([0.5, 0.5]), return_norm=True)
    assert b.dtype == np.float64 and np.array_equal(b, g["final"])
    assert np.array_equal(nrm, np.linalg.norm(g["blend"], axis=1))
    # the float32 flavour is still there for float32 tables (and is NOT the reference's bits)
    got32 = content_tail(E.astype(np.float32), g["src"], g["dst"])
    assert got32.dtype == np.float32 and np.abs(got32 - g["final"]).max() <= 1e-6


def test_content_tail_f64_golden_g15(golden, hw):
    """G15, 6 000 x 12: sha256 of all of `final` equals the reference's."""
    from hwer.content_recommender import content_tail
    g = golden("g15_content_fit_6k")
    sc = g15_scenario(g)
    final = content_tail(sc["E"], sc["src"], sc["dst"])
    assert _sha(final) == str(g["final_sha"])
    assert np.array_equal(final[g["final_rows"]], g["final_sample"])


@pytest.mark.parametrize("d,T", [(128, 2), (256, 4), (66, 3), (8, 1), (7, 2), (1, 2), (129, 2), (515, 3), (1024, 2)])
def test_blend_normalize_f64_vs_numpy(hw, d, T):
    """out[r] = unit(sum_t alpha[r, t] T_t[idx_t[r]]) on float64 tables: bit-identical to the numpy expression `acc = acc + a * rows`
    followed by hwer/utils.py:43-44 (oracle.blend_normalize); per-row weights with cold-start zeros, row gathers, shared weights; widths
    on both sides of numpy's 8 / 128-element pairwise-summation breaks."""
    from hwer.content_recommender import blend_tables
    rng = np.random.default_rng(1000 * d + T)
    R = 1237
    tables = [rng.standard_normal((900 + 50 * t, d)) * (0.3 + t) for t in range(T)]
    tables[0] = rng.standard_normal((R, d))
    idx = [None if t == 0 else rng.integers(0, tables[t].shape[0], R).astype(np.int32) for t in range(T)]
    alpha = rng.dirichlet(np.ones(T), R)
    if T > 1:
        alpha[::5, -1] = 0.0
    out, norm = blend_tables(tables, alpha, idx, return_norm=True)
    ref = orc.blend_normalize(tables, idx, alpha)
    assert out.dtype == np.float64 and np.array_equal(out, ref)
    pre = np.zeros((R, d))
    for t in range(T):
        pre = pre + alpha[:, t][:, None] * (tables[t] if idx[t] is None else tables[t][idx[t]])
    assert np.array_equal(norm, np.linalg.norm(pre, axis=1))
    shared = rng.dirichlet(np.ones(T))
    out2 = blend_tables([t[:900] for t in tables], shared)
    assert np.array_equal(out2, orc.blend_normalize([t[:900] for t in tables], [None] * T, shared))


def test_segment_mean_f64_ragged(hw):
    """neighbour lists of length 1 ... 300 (the 4-deep load pipeline and its tail), d on both sides of one 16-byte chunk per lane"""
    from hwer.content_recommender import segment_mean
    rng = np.random.default_rng(55)
    for d in (3, 12, 64, 130, 300):
        N = 500
        E = rng.standard_normal((N, d))
        deg = rng.integers(1, 12, N)
        deg[::50] = rng.integers(100, 300, len(deg[::50]))
        indptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(deg, out=indptr[1:])
        indices = rng.integers(0, N, int(indptr[-1])).astype(np.int32)
        got = segment_mean(E, indptr, indices)
        assert np.array_equal(got, orc.segment_mean(E, indptr, indices)), d


def test_zero_row_becomes_nan_like_the_reference(hw):
    """hwer/utils.py:43-44 has no epsilon: a zero blend row is 0 / 0 = NaN in the reference, and here"""
    from hwer.content_recommender import blend_tables
    E = np.random.default_rng(3).standard_normal((64, 10))
    E[5] = 0.0
    with np.errstate(invalid="ignore"):
        ref = orc.blend_normalize([E], [None], np.array([1.0]))
    got = blend_tables([E], np.array([1.0]))
    assert np.array_equal(got, ref, equal_nan=True) and np.isnan(got[5]).all()


# ------------------------------------------------------------------------------------------------- fit() end to end
def test_fit_6k_neighbours_equal_the_reference_g15(golden, hw):
    """ContentRecommendation.fit on the G15 scenario, everything after the (out-of-scope) embedders on the device in float64: the
    device PCA differs from scikit-learn's SVD by rounding noise (~1e-13 on E), the blend stage adds none - `final` within 1e-9 of the
    reference's rows - and for EVERY one of the 1 000 users the 200 nearest items are the reference's, in the reference's order, with
    scores within 1e-9.  Then the same model built on the reference's own `final` (content_tail_f64 of the captured table): ids AND
    scores identical to the reference for all 1 000 users (sha256 of the scores)."""
    from hwer import ContentRecommendation, Edge, IdentityEmbedding, Node, VectorIndex
    from hwer.content_recommender import content_tail
    g = golden("g15_content_fit_6k")
    sc = g15_scenario(g)
    types, tid, N, k = sc["types"], sc["tid"], sc["N"], sc["k"]
    nodes = [Node(types[tid[i]], i) for i in range(N)]
    cursor, node_data = [0, 0, 0], {}
    for i, n in enumerate(nodes):
        node_data[n] = {"f": sc["feats"][tid[i]][cursor[tid[i]]]}
        cursor[tid[i]] += 1
    edges = [Edge(nodes[s], nodes[t], 1.0) for s, t in zip(sc["src"], sc["dst"])]
    model = ContentRecommendation({t: {"f": IdentityEmbedding(n_dims=sc["dims"][ti])} for ti, t in enumerate(types)}, set(types),
                                  n_dims=sc["n_dims"])
    final = model.fit(nodes, edges, node_data)
    assert final.dtype == np.float64 and final.shape == (N, 12)
    assert np.abs(final[g["final_rows"]] - g["final_sample"]).max() <= 1e-9
    users = [nodes[i] for i in sc["users"]]
    ref_idx = g["knn_idx"].astype(np.int64)
    idx, score = model.find_closest_neighbours_batch("item", users, k=k)
    assert np.array_equal(idx, ref_idx)
    assert np.abs(score[::125] - g["score_sample"]).max() <= 1e-9
    # the per-anchor API, a few users
    for i in (0, 499, 999):
        res = model.find_closest_neighbours("item", users[i], k=k)
        assert [model.nodes_to_idx[n] for n, _ in res] == ref_idx[i].tolist()
    # the reference's own rows (blend stage on the captured table): everything identical
    exact = content_tail(sc["E"], sc["src"], sc["dst"])
    assert _sha(exact) == str(g["final_sha"])
    m2 = VectorIndex(node_types=set(types), n_dims=12)
    m2.fit(nodes, exact)
    idx2, score2 = m2.find_closest_neighbours_batch("item", users, k=k)
    assert np.array_equal(idx2, ref_idx)
    assert _sha(np.ascontiguousarray(score2, dtype=np.float64)) == str(g["score_sha"])


def test_fit_float32_flavour_still_available(golden, hw):
    """blend_dtype = "float32" keeps rounds 1-4's flavour: float32 rows widened, within 1e-6 of the reference"""
    from hwer import ContentRecommendation, Edge, IdentityEmbedding, Node
    g6 = golden("g6_content_tail")
    rng = np.random.default_rng(7)
    types = ["user", "item", "tag"]
    nodes = [Node(types[i % 3], i) for i in range(300)]
    feats = rng.standard_normal((300, 5))
    node_data = {n: {"f": feats[i]} for i, n in enumerate(nodes)}
    edges = [Edge(nodes[s], nodes[t], 1.0) for s, t in zip(g6["src"], g6["dst"])]
    model = ContentRecommendation({t: {"f": IdentityEmbedding(n_dims=5)} for t in types}, set(types), n_dims=8)
    model.blend_dtype = "float32"
    final = model.fit(nodes, edges, node_data)
    assert np.array_equal(final, final.astype(np.float32).astype(np.float64))
    assert np.abs(final - g6["final"]).max() <= 1e-6
    model64 = ContentRecommendation({t: {"f": IdentityEmbedding(n_dims=5)} for t in types}, set(types), n_dims=8)
    final64 = model64.fit(nodes, edges, node_data)
    assert np.abs(final64 - g6["final"]).max() <= 1e-11      # float64 blend: only the PCA's rounding noise is left
    res = model64.find_closest_neighbours("item", nodes[0], k=7)
    assert [model64.nodes_to_idx[n] for n, _ in res] == g6["knn_idx"].tolist()
    assert np.abs(np.array([s for _, s in res]) - g6["knn_score"]).max() <= 1e-11


# ------------------------------------------------------------------------------------------------- the N > 1 job, rehearsed on one GPU
def _bench_n(n, extra_env, timeout):
    import subprocess
    import sys
    import time
    env = dict(os.environ, HWER_BENCH_BACKEND="gloo", HWER_BENCH_ITEMS="400000", HWER_BENCH_NO_ANCHOR="1", HWER_BENCH_DIST_TIMEOUT_S="120", **extra_env)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=timeout)
    return r, time.time() - t0


def test_bench_n4_runs_the_library_pipeline_and_survives_a_dead_rank(hw):
    """`python bench.py --gpus 4` with its own launcher, four ranks over gloo on this one GPU (the payload staged through host memory; kernels,
    layouts, shortened lists and the merge are the job's): the step is hwer.distributed.ShardedStepPipeline, the line names the backend that
    really ran, carries `cpu_baseline`, and the merged result passed the oracle check inside the job.  Then the same job with rank 2 killed
    right after the rendezvous: the launcher ends the job with a non-zero exit code and that rank's stderr in well under a minute."""
    import json
    r, dt = _bench_n(4, {}, 400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 4 and line["scaling"] == "strong" and line["parity_checked_queries"] >= 8
    assert "backend gloo" in line["config"]["sharding"] and "ShardedStepPipeline" in line["config"]["sharding"] and "RCCL" not in line["config"]["sharding"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["exchange"]["lists_ran_out_first_pass"] == 0 and line["variants"]["by_query"]["ms_per_step"] > 0
    r, dt = _bench_n(4, {"HWER_BENCH_FAIL_RANK": "2"}, 200)
    assert r.returncode != 0 and dt < 60.0, (r.returncode, dt)
    assert "rank 2 of 4 exited with code 3" in r.stderr and "HWER_BENCH_FAIL_RANK" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_sharded_job_over_rccl_at_world_size_one(hw):
    """The N > 1 job's own code - rendezvous with a device id, ShardedStepPipeline, all-gather AND all-to-all exchanges, the merge, the parity
    gather, the all-reduced timings - over backend nccl (= RCCL), at the world size a one-GPU box has: 1.  gloo rehearsals cover N = 2 / 4 with
    the payload staged through the host; this is the branch the 8-GPU run takes, collectives on device buffers included."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HWER_BENCH_SHARDED_WORLD1="1", HWER_BENCH_ITEMS="600000", HWER_BENCH_DIST_TIMEOUT_S="120", MASTER_PORT="29591")
    env.pop("HWER_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and "backend nccl" in line["config"]["sharding"] and line["parity_checked_queries"] >= 8
    assert line["exchange"]["lists_ran_out_first_pass"] == 0 and line["variants"]["by_query"]["ms_per_step"] > 0 and line["cpu_baseline"]["value"] > 0


# ------------------------------------------------------------------------------------------------- bulk anchor-only queries
@pytest.mark.parametrize("d,arith", [(128, 0), (128, 1), (64, 0), (100, 0), (66, 1), (256, 0), (515, 0), (7, 1), (1, 0)])
def test_anchor_query_kernel_equals_query_mix(hw, d, arith):
    """hwer_query_mix with at least 4 096 anchor-only queries runs the row-streaming kernel (round 5); the same call with EMPTY positive
    lists runs the general 8-lanes-per-query kernel: Q64 and Q32 (padding columns included) must be identical bit for bit - unknown anchors
    (-1: row 0 clipped), repeated anchors, a ragged last trip - and equal the oracle's query_vector on a sample."""
    import torch
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda")
    rng = np.random.default_rng(900 + d)
    N, B = 3000, 4099
    ldv = (d + 3) // 4 * 4
    ldq = max(32, (d + 15) // 16 * 16)
    Vh = np.zeros((N, ldv), np.float32)
    Vh[:, :d] = rng.standard_normal((N, d)).astype(np.float32) * rng.uniform(0.1, 3.0, (N, 1)).astype(np.float32)
    V = torch.from_numpy(Vh).to(dev)
    an = rng.integers(0, N, B).astype(np.int32)
    an[::97] = -1
    anchors = torch.from_numpy(an).to(dev)
    empty_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    dummy_idx = torch.zeros(1, dtype=torch.int32, device=dev)
    P, S = native.ptr, native.stream_ptr()

    def run(with_lists):
        Q64 = torch.full((B, d), 7.0, dtype=torch.float64, device=dev)
        Q32 = torch.full((B, ldq), 7.0, dtype=torch.float32, device=dev)
        native.check(lib.hwer_query_mix(P(V), N, d, ldv, P(anchors), P(empty_ptr) if with_lists else None, P(dummy_idx) if with_lists else None,
                                        None, None, B, arith, P(Q64), P(Q32), ldq, S), "query_mix")
        return Q64.cpu().numpy(), Q32.cpu().numpy()
    a64, a32 = run(False)
    b64, b32 = run(True)
    assert np.array_equal(a64, b64) and np.array_equal(a32, b32)
    assert np.all(a32[:, d:] == 0.0)
    Vd = Vh[:, :d].astype(np.float64) if arith == 0 else Vh[:, :d]
    for b in (0, 1, 97, 2048, B - 1):
        ref = orc.query_vector(Vd, int(an[b]))
        assert np.array_equal(a64[b], np.asarray(ref, dtype=np.float64)), b


# ------------------------------------------------------------------------------------------------- group-maxima seed pre-pass
@pytest.mark.parametrize("B,N,d,k,layout", [(70000, 30000, 128, 200, "clustered"), (70000, 30000, 128, 200, "random"), (9000, 26744, 128, 200, "clustered"),
                                            (66000, 12000, 64, 50, "clustered")])
def test_group_maxima_prepass_on_clustered_catalogues(hw, B, N, d, k, layout):
    """The seed pre-pass of plans without lane candidates keeps ONE value per (sample tile, lane half, query) - the maximum of 16 keys - and takes
    tau0 from the group maxima (round 5).  A catalogue SORTED BY CLUSTER puts a query's good rows next to each other: several of them share a
    group, the threshold comes out lower, more keys pass (stream compactions inside the screen) - and the result must still be the oracle's,
    every query certified or recomputed.  One list per query (70 000 / 66 000 queries) and eight lists (9 000 queries: the ML-20M tail's plan)."""
    from hwer.device_index import DeviceIndex
    from oracle import c_oracle
    rng = np.random.default_rng(B + N)
    if layout == "clustered":
        n_c = 60
        centres = unit_rows_f32(71, n_c, d).astype(np.float64)
        lab = np.sort(rng.integers(0, n_c, N))                      # rows grouped by cluster
        X = centres[lab] + 0.35 * rng.standard_normal((N, d)) / np.sqrt(d)
        X = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
        ul = rng.integers(0, n_c, B)
        Q = centres[ul] + 0.35 * rng.standard_normal((B, d)) / np.sqrt(d)
        Q = (Q / np.linalg.norm(Q, axis=1, keepdims=True)).astype(np.float32)
    else:
        X, Q = unit_rows_f32(72, N, d), unit_rows_f32(73, B, d)
    V = np.concatenate([Q, X]).astype(np.float64)
    ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)})
    idx, score, dist, flags = ix.search("item", np.arange(B), k=k)
    assert (flags > 0).all()
    sel = np.unique(np.linspace(0, B - 1, 96).astype(np.int64))
    Qh = np.stack([orc.query_vector(V, int(u)) for u in sel])
    o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(X, Qh, V[sel], k, threads=c_oracle.effective_cpus())
    assert np.array_equal(idx[sel], o_idx + B) and np.array_equal(score[sel], o_sc) and np.array_equal(dist[sel], np.sqrt(o_d2))
    assert (flags == 2).mean() < 0.01          # the thresholds held: (almost) no query needed the exhaustive pass


# ------------------------------------------------------------------------------------------------- per-anchor calls on genuine float64 rows
@pytest.mark.parametrize("U,N,d,k", [(40, 1682, 64, 200), (10, 20000, 96, 50), (6, 300, 24, 268)])
def test_one_anchor_graph_on_float64_rows_equals_the_general_path(hw, U, N, d, k):
    """Rows that are NOT float32-representable (what ContentRecommendation.fit returns since the float64 blend stage): one
    find_closest_neighbours call replays the cached hipGraph of the candidate scheme (_OneAnchorSearchF64) and must return exactly the
    general path's and the oracle's (id, score) lists; exact duplicates and float32-colliding rows sit at a few anchors' cut, so some
    replays come back uncertified (flag 0) and are answered by the general path; k + 32 == every row in the last case."""
    from hwer import Node, VectorIndex
    rng = np.random.default_rng(700 + d)
    V = rng.standard_normal((U + N, d))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    V[U + 5] = V[U + 3]
    V[U + N - 1] = V[U + 3] * (1.0 + 2.0 ** -40)          # equal after rounding to float32, different in float64
    users, items = [Node("user", i) for i in range(U)], [Node("item", i) for i in range(N)]
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    m.fit(users + items, V)
    ix = m.knn.index
    assert ix.V64 is not None
    Vd = ix.V64.cpu().numpy()
    per_anchor = {u: m.find_closest_neighbours("item", u, k=k) for u in users}
    assert any(key[:3] == ("item", k, ("f64rows", 0)) for key in ix.__dict__.get("_one_anchor", {}))       # the graph path was taken
    again = m.find_closest_neighbours("item", users[1], k=k)                                           # the graph is reused
    assert again == per_anchor[users[1]]
    ix.one_anchor_graphs = False
    for j, u in enumerate(users):
        general = m.find_closest_neighbours("item", u, k=k)
        assert per_anchor[u] == general, j
        o_idx, o_sc, _ = orc.find_closest(Vd, np.arange(U, U + N), j, k=k)
        assert [m.nodes_to_idx[n] for n, _ in general] == o_idx.tolist() and [s for _, s in general] == o_sc.tolist()
