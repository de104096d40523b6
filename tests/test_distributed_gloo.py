"""Multi-process (world_size 2 and 4, gloo, CPU) coverage of the N > 1 path: shard bounds, the single all-gather of packed
local results, global-id bookkeeping and the merge order.  The per-shard search and the merge are HIP kernels in the
product; here they are replaced by oracle-based stand-ins (tests may use the oracle) so that the collective plumbing
of hwer.distributed.ShardedSearch runs without a GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _unit_rows(seed, n, d):
    x = np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32).astype(np.float64)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def _worker(rank, world, port, n_items, k, out_dir, list_len=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hwer.distributed import ShardedSearch, shard_bounds
    from oracle import hwer_oracle as orc
    B, d, U = 6, 16, 6   # B is re-bound inside `merge` (the by_query exchange merges a slice)
    V = np.concatenate([_unit_rows(1, U, d), _unit_rows(2, n_items, d)]).astype(np.float64)
    Q = np.stack([orc.query_vector(V, u) for u in range(U)])
    lo, hi = shard_bounds(n_items, world, rank)

    def local_search(Q64, Q32, kl, rows=None):   # stand-in for DeviceIndex.search_queries(sort_by_score=False, id_base=U + lo)
        nb = Q64.shape[0]
        anchors = np.arange(nb) if rows is None else rows.numpy()   # query b of the batch is user b
        idx = np.empty((nb, kl), np.int64); sc = np.empty((nb, kl)); d2 = np.empty((nb, kl))
        for b in range(nb):
            loc, dist_ = orc.knn_bruteforce(Q64[b].numpy(), V[U + lo:U + hi], kl)
            idx[b] = loc + U + lo
            d2[b] = orc.dist2_sequential(Q64[b].numpy(), V[U + lo:U + hi])[loc]
            sc[b] = orc.pair_score(V, np.full(kl, anchors[b]), idx[b])
        return torch.from_numpy(idx), torch.from_numpy(sc), torch.from_numpy(d2)

    def merge(g_idx, g_score, g_dist2, sort_by_score, k=k, want_exhausted=False):   # stand-in for hwer_merge_topk_lists
        G, B, k_in = g_idx.shape
        o_i = np.empty((B, k), np.int64); o_s = np.empty((B, k)); o_d = np.empty((B, k)); ex = np.zeros(B, np.int32)
        for b in range(B):
            dd = np.concatenate([g_dist2[g, b].numpy() for g in range(G)])
            ii = np.concatenate([g_idx[g, b].numpy() for g in range(G)])
            ss = np.concatenate([g_score[g, b].numpy() for g in range(G)])
            order = np.lexsort((ii, dd))[:k]
            taken = np.bincount(order // k_in, weights=(ii[order] >= 0), minlength=G)   # real entries each list gave to the winners
            ex[b] = int(k_in < k and (taken >= k_in).any())
            if sort_by_score:
                order = order[np.argsort(-ss[order], kind="stable")]
            o_i[b], o_s[b], o_d[b] = ii[order], ss[order], dd[order]
        out = (torch.from_numpy(o_i), torch.from_numpy(o_s), torch.from_numpy(o_d))
        return out + (torch.from_numpy(ex),) if want_exhausted else out

    sh = ShardedSearch(n_items, local_search, merge, list_len=list_len)
    assert (sh.lo, sh.hi) == (lo, hi)
    idx, score, dist2 = sh.search(torch.from_numpy(Q), torch.from_numpy(Q.astype(np.float32)), k)
    if list_len is not None:   # shortened local lists: same result; the lists that ran out were repeated with full lists
        open(os.path.join(out_dir, "rep%d" % rank), "w").write(str(sh.repeated_queries))
    # every rank must hold the single-process result
    for b in range(B):
        o_idx, o_score, o_dist = orc.find_closest(V, np.arange(U, U + n_items), b, k=k)
        assert np.array_equal(idx[b].numpy(), o_idx), (rank, b)
        assert np.array_equal(score[b].numpy(), o_score)
        assert np.array_equal(np.sqrt(dist2[b].numpy()), o_dist)
    with pytest.raises(ValueError):
        sh.search(torch.from_numpy(Q), torch.from_numpy(Q.astype(np.float32)), n_items + 1)
    if list_len is not None:
        # deferred searches (round 5): no host synchronisation inside search(); the repeat of the queries whose list ran out happens in
        # settle() - called explicitly, or by the NEXT search one step late - and patches the tensors the search returned, in place
        first = sh.search(torch.from_numpy(Q), torch.from_numpy(Q.astype(np.float32)), k, deferred=True)
        second = sh.search(torch.from_numpy(Q), torch.from_numpy(Q.astype(np.float32)), k, deferred=True)   # settles `first`
        rep_first = sh.repeated_queries
        assert torch.equal(first[0], idx) and torch.equal(first[1], score) and torch.equal(first[2], dist2)
        assert sh.settle() == rep_first == int(open(os.path.join(out_dir, "rep%d" % rank)).read())   # the same batch again: the same repeats
        assert sh.settle() == 0                                                                       # nothing pending any more
        assert torch.equal(second[0], idx) and torch.equal(second[1], score) and torch.equal(second[2], dist2)
        # a pipelined caller REFILLS its resident query buffers between steps (advisor finding, round 5): the deferred repeat of search 1 runs one
        # search later and must still use search 1's queries (search() snapshots them) - with the buffers' new content it would patch wrong rows
        Qbuf, Q32buf = torch.from_numpy(Q.copy()), torch.from_numpy(Q.astype(np.float32))
        first = sh.search(Qbuf, Q32buf, k, deferred=True)
        perm = np.roll(np.arange(B), 1)
        Qbuf.copy_(torch.from_numpy(Q[perm]))
        Q32buf.copy_(torch.from_numpy(Q[perm].astype(np.float32)))
        sh.search(Qbuf, Q32buf, k, deferred=True)            # settles `first` - after the buffers changed
        sh.settle()
        assert torch.equal(first[0], idx) and torch.equal(first[1], score) and torch.equal(first[2], dist2)
    # user-partitioned exchange: this rank ends with the merged rows of its own query slice only
    from hwer.distributed import query_slice
    B = 6
    shq = ShardedSearch(n_items, local_search, merge, exchange="by_query")
    idx_q, score_q, dist2_q = shq.search(torch.from_numpy(Q), torch.from_numpy(Q.astype(np.float32)), k)
    q_lo, q_hi = query_slice(B, world, rank)
    assert idx_q.shape[0] == q_hi - q_lo
    assert torch.equal(idx_q, idx[q_lo:q_hi]) and torch.equal(score_q, score[q_lo:q_hi]) and torch.equal(dist2_q, dist2[q_lo:q_hi])
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items,k", [(101, 20), (7, 5), (3, 3)])
def test_sharded_search_world2_gloo(tmp_path, n_items, k):
    port = 29500 + (os.getpid() + n_items) % 2000
    mp.spawn(_worker, args=(2, port, n_items, k, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


@pytest.mark.parametrize("n_items,k,list_len", [(101, 20, 12), (101, 20, 10), (40, 30, 16)])
def test_sharded_search_short_lists_world2_gloo(tmp_path, n_items, k, list_len):
    """Shortened local lists (hwer.distributed.local_list_len / hwer_merge_topk_lists): every shard sends list_len < k entries, the
    merge flags the queries for which a list ran out, those are repeated with full lists - the result equals the single-process one
    (asserted inside the worker) and both ranks repeat the same number of queries."""
    port = 31500 + (os.getpid() + n_items + list_len) % 2000
    mp.spawn(_worker, args=(2, port, n_items, k, str(tmp_path), list_len), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()
    rep = [int((tmp_path / ("rep%d" % r)).read_text()) for r in range(2)]
    assert rep[0] == rep[1]
    if list_len * 2 == k:
        assert rep[0] > 0   # 2 x 10 = 20: a list is exhausted unless the two shards hold exactly 10 each


@pytest.mark.parametrize("n_items,k,list_len", [(101, 20, None), (101, 20, 8), (9, 8, 3)])
def test_sharded_search_world4_gloo(tmp_path, n_items, k, list_len):
    """Four ranks (round 4): ragged shards (101 = 26 + 26 + 26 + 23; 9 rows over 4 ranks leave the last shard short of the list length), both
    exchanges, full and shortened lists - every rank holds the single-process result (asserted inside the worker)."""
    port = 33500 + (os.getpid() + n_items + (list_len or 0)) % 2000
    mp.spawn(_worker, args=(4, port, n_items, k, str(tmp_path), list_len), nprocs=4, join=True)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(4))
    if list_len is not None:
        rep = [int((tmp_path / ("rep%d" % r)).read_text()) for r in range(4)]
        assert len(set(rep)) == 1


def test_local_search_without_rows_keeps_full_lists():
    """A local_search written against the rounds-1/2 contract (three arguments, no `rows=`) must never meet the data-dependent repeat of the
    shortened lists: ShardedSearch then keeps full lists whatever list_len says (advisor finding, round 3)."""
    from hwer.distributed import ShardedSearch

    class FakeDist:
        pass
    sh = ShardedSearch.__new__(ShardedSearch)
    sh.exchange, sh.world, sh.list_len = "all_gather", 8, 72
    from hwer.distributed import merge_on_device
    sh.merge = merge_on_device
    sh.local_search = lambda Q64, Q32, k: None
    assert sh._list_len(200) == 200
    sh.local_search = lambda Q64, Q32, k, rows=None: None
    assert sh._list_len(200) == 72
    sh.list_len = None
    assert sh._list_len(200) == 72       # the device merge's default: local_list_len(200, 8)
    sh.local_search = lambda Q64, Q32, k, **kw: None
    assert sh._list_len(200) == 72


def test_local_list_len_rule():
    from hwer.distributed import local_list_len
    assert local_list_len(200, 1) == 200 and local_list_len(200, 2) == 200
    assert local_list_len(200, 8) == 72 and local_list_len(200, 4) == 136
    for k in (1, 7, 20, 200, 392):
        for g in (1, 2, 3, 4, 8, 16):
            kin = local_list_len(k, g)
            assert 1 <= kin <= k and kin * g >= k


def test_shard_bounds_cover_and_order():
    from hwer.distributed import shard_bounds
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for g in (1, 2, 3, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= (n + g - 1) // g for lo, hi in spans)
