"""Worker of the two-process exchange tests (tests/test_gpu_round2.py): one rank of a row-sharded search that runs the REAL
HIP local search -> the real exchange (all-gather and user-partitioned) -> the real merge kernel, and checks the merged result
against the oracle's brute force over the WHOLE catalogue.

    python tests/dist_worker.py <backend: nccl|gloo> <rank> <world> <port> <out_dir> [n_items] [d] [B] [k] [f64rows]
(a 10th argument "f64rows": GENUINE float64 rows - every rank's shard goes through the fused float64-row search, round 6 - checked against the numpy oracle on those rows)

backend nccl: one GPU per rank (RCCL).  backend gloo: every rank uses cuda:0 and the payload is staged through host memory
(hwer.distributed does that for non-nccl backends) - the kernels, layouts and the merge are the same as under RCCL.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def unit_rows_f32(seed, n, d):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32).astype(np.float64)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def main():
    backend, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    n_items, d, B, k = (int(x) for x in (sys.argv[6:10] + ["20011", "64", "97", "200"][max(len(sys.argv) - 6, 0):])[:4])
    f64rows = len(sys.argv) > 10 and sys.argv[10] == "f64rows"
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from hwer.device_index import DeviceIndex
    from hwer.distributed import ShardedSearch, query_slice, shard_bounds, sharded_index_search
    from oracle import c_oracle
    from oracle import hwer_oracle as orc

    U = B
    users, items = unit_rows_f32(101, U, d), unit_rows_f32(202, n_items, d)
    lo, hi = shard_bounds(n_items, world, rank)
    if f64rows:
        return main_f64rows(backend, rank, world, out_dir, users, items, lo, hi, k, dev)
    # this rank's table: the (replicated) users followed by ITS item rows; reported ids are global (id_base = lo)
    V_local = np.concatenate([users, items[lo:hi]]).astype(np.float64)
    ix = DeviceIndex(V_local, {"user": np.arange(U), "item": np.arange(U, U + hi - lo)}, device=dev)
    Q64, Q32, a_dev = ix.build_queries(np.arange(U))
    V64 = np.concatenate([users, items]).astype(np.float64)
    Qh = np.stack([orc.query_vector(V64, u) for u in range(U)])
    o_idx, o_score, o_d2 = c_oracle.find_closest_f64(items, Qh, V64[:U], k, threads=4)

    results = {}
    for exchange in ("all_gather", "by_query"):
        sh = ShardedSearch(n_items, None, exchange=exchange)
        assert (sh.lo, sh.hi) == (lo, hi)
        idx, score, dist2 = sharded_index_search(ix, "item", sh, Q64, Q32, k, anchors_dev=a_dev)
        torch.cuda.synchronize()
        q_lo, q_hi = (0, U) if exchange == "all_gather" else query_slice(U, world, rank)
        assert idx.shape == (q_hi - q_lo, k)
        got_idx, got_score, got_d2 = idx.cpu().numpy(), score.cpu().numpy(), dist2.cpu().numpy()
        assert np.array_equal(got_idx, o_idx[q_lo:q_hi]), (exchange, rank)          # ids are item-local here: id_base = lo
        assert np.array_equal(got_score, o_score[q_lo:q_hi]), (exchange, rank)
        assert np.array_equal(got_d2, o_d2[q_lo:q_hi]), (exchange, rank)
        results[exchange] = got_idx
    # the distance-ordered variant (MultiKNN.query order) through the same exchange
    sh = ShardedSearch(n_items, None, exchange="all_gather")
    idx, score, dist2 = sharded_index_search(ix, "item", sh, Q64, Q32, k, anchors_dev=a_dev, sort_by_score=False)
    r_idx, _, r_d2 = c_oracle.find_closest_f64(items, Qh, V64[:U], k, sort_by_score=False, threads=4)
    assert np.array_equal(idx.cpu().numpy(), r_idx) and np.array_equal(dist2.cpu().numpy(), r_d2)
    # shortened local lists (round 3): every shard sends list_len < k entries, hwer_merge_topk_lists flags the queries for which a list
    # ran out, those - and only those - are repeated with full lists (a second, smaller exchange): same result, both orders
    repeated = []
    for list_len in sorted({-(-k // world), min(k, -(-k // world) + max(2, k // 10)), min(k, (3 * k) // 4)}):
        sh = ShardedSearch(n_items, None, exchange="all_gather", list_len=list_len)
        idx, score, dist2 = sharded_index_search(ix, "item", sh, Q64, Q32, k, anchors_dev=a_dev)
        assert np.array_equal(idx.cpu().numpy(), o_idx) and np.array_equal(score.cpu().numpy(), o_score) and np.array_equal(dist2.cpu().numpy(), o_d2), (rank, list_len)
        repeated.append((list_len, sh.repeated_queries))
        idx, score, dist2 = sharded_index_search(ix, "item", sh, Q64, Q32, k, anchors_dev=a_dev, sort_by_score=False)
        assert np.array_equal(idx.cpu().numpy(), r_idx) and np.array_equal(dist2.cpu().numpy(), r_d2), (rank, list_len)
    if world > 1 and k * world <= n_items and k >= 8:
        assert repeated[0][1] > 0, repeated   # lists of k / world entries: a list runs out unless the k nearest split exactly evenly
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as fh:
        fh.write("%s world=%d shard=[%d,%d) short lists (len, repeated queries): %s\n" % (backend, world, lo, hi, repeated))
    dist.barrier()
    dist.destroy_process_group()


def main_f64rows(backend, rank, world, out_dir, users, items, lo, hi, k, dev):
    """the same sharded search over GENUINE float64 rows: local lists from hwer_knn_search_f64rows (ids = id_base + local row of the shard), one exchange,
    the merge - ids, scores and distances == the numpy oracle on the float64 rows over the WHOLE catalogue, both exchanges, full and shortened lists"""
    import torch
    import torch.distributed as dist
    from hwer.device_index import DeviceIndex
    from hwer.distributed import ShardedSearch, query_slice, sharded_index_search
    from oracle import hwer_oracle as orc
    U, n_items = len(users), len(items)
    rng = np.random.default_rng(7)
    V = np.concatenate([users, items]).astype(np.float64) + 1e-9 * rng.standard_normal((U + n_items, users.shape[1]))
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    V[U + 11] = V[U + 10] * (1.0 + 3e-9)                     # float32-colliding rows (one shard holds both)
    V_local = np.concatenate([V[:U], V[U + lo:U + hi]])
    ix = DeviceIndex(V_local, {"user": np.arange(U), "item": np.arange(U, U + hi - lo)}, device=dev)
    assert ix.V64 is not None
    Q64, Q32, a_dev = ix.build_queries(np.arange(U))
    want = [orc.find_closest(V, np.arange(U, U + n_items), u, k=k) for u in range(U)]
    o_idx = np.stack([w[0] for w in want]) - U               # (ids are item-local here: id_base = lo)
    o_score, o_dist = np.stack([w[1] for w in want]), np.stack([w[2] for w in want])
    for exchange, list_len in (("all_gather", None), ("by_query", None), ("all_gather", -(-k // world) + max(2, k // 10))):
        sh = ShardedSearch(n_items, None, exchange=exchange, list_len=list_len)
        idx, score, dist2 = sharded_index_search(ix, "item", sh, Q64, Q32, k, anchors_dev=a_dev)
        torch.cuda.synchronize()
        q_lo, q_hi = (0, U) if exchange == "all_gather" else query_slice(U, world, rank)
        assert np.array_equal(idx.cpu().numpy(), o_idx[q_lo:q_hi]), (exchange, rank)
        assert np.array_equal(score.cpu().numpy(), o_score[q_lo:q_hi]), (exchange, rank)
        assert np.array_equal(np.sqrt(dist2.cpu().numpy()), o_dist[q_lo:q_hi]), (exchange, rank)
    with open(os.path.join(out_dir, "ok%d" % rank), "w") as fh:
        fh.write("%s world=%d float64 rows shard=[%d,%d)\n" % (backend, world, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
