"""Item catalogue sharded row-wise over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm).  No reference counterpart — the reference is single process.

Each rank holds rows [lo, hi) of the retrieved type's table and answers every query against its shard with the
single-GPU kernels (exact local k nearest in (dist2, id) order).  ONE exchange step over xGMI follows — either an
all-gather of the packed local results (every rank ends with the full result) or the user-partitioned exchange
(every rank receives only the lists of ITS slice of the queries and merges those: 1/G of the bytes per link) — and
then a k-way merge kernel.  Because the local lists and the merge use the same total order (dist2 asc, global id
asc), the merged result is identical to the 1-GPU result.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of `rank`: ceil(n/G) rows per shard, the last ones may be short or empty."""
    per = (n_rows + world_size - 1) // world_size
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


KPRIME_SLACK = 56   # csrc/knn_internal.h SLACK: a local search keeps K' = roundup64(k + 56) keys per query


def local_list_len(k: int, world_size: int) -> int:
    """Entries per query a shard sends when the catalogue is split over `world_size` shards (round 3: shortened local lists).
    A shard holds ~k / G of a query's k nearest (Binomial(k, 1/G) for rows dealt at random): k_in = the largest list length
    with the same K' = roundup64(k_in + 56) as mean + 6 sigma + 4, capped at k.  Exactness never rests on that estimate - the
    merge reports every query for which a list ran out (hwer_merge_topk_lists: out_exhausted) and ShardedSearch.search repeats
    those with full lists - it only decides how often that happens.  G = 8, k = 200: 72 (K' = 128 instead of 256); G = 4: 136;
    G <= 2: k."""
    G = max(int(world_size), 1)
    if G <= 1 or k <= 8:
        return k
    mean = k / G
    need = mean + 6.0 * (mean * (1.0 - 1.0 / G)) ** 0.5 + 4.0
    kp = -(-(int(need + 0.999999) + KPRIME_SLACK) // 64) * 64          # K' that `need` entries cost
    k_in = kp - KPRIME_SLACK                                            # ... and the longest list of that cost
    k_in = max(k_in, -(-k // G))                                        # G * k_in >= k
    return k if k_in >= k else k_in


def all_gather_results(idx, score, dist2, group=None, send=None, recv=None):
    """ONE all_gather_into_tensor of the three (B, k) result arrays -> three (G, B, k) views of the receive buffer.
    Every rank packs (idx | score bits | dist2 bits) into one (3, B, k) int64 buffer (device tensors: one hwer_pack_exchange
    launch), the collective writes rank g's buffer to recv[g], and the returned tensors are STRIDED VIEWS of recv (list
    stride 3 B k): hwer_merge_topk_strided merges them in place - no unpack copies.
    24 bytes per entry: the exact float64 distance orders the merge, the id names the row, and the score has to travel
    with them - it is computed from the item's row, which only the owning rank holds."""
    import torch
    import torch.distributed as dist
    G = dist.get_world_size(group)
    B, k = idx.shape
    if B == 0:
        return idx.new_empty((G, 0, k)), score.new_empty((G, 0, k)), dist2.new_empty((G, 0, k))
    # send / recv (optional): preallocated (1, 3, B, k) / (G, 3, B, k) int64 buffers - a pipelined caller issues this on a side stream and must not
    # allocate there (ShardedStepPipeline owns one pair per result set)
    send = pack_send_buffer(idx, score, dist2, 1, out=send)[0]            # (3, B, k) int64
    if send.is_cuda and dist.get_backend(group) != "nccl":
        # test harness only (e.g. two ranks sharing one GPU over gloo): stage the payload through host memory
        host = torch.empty((G,) + tuple(send.shape), dtype=torch.int64)
        dist.all_gather_into_tensor(host.view(-1), send.cpu().view(-1), group=group)
        if recv is None:
            recv = host.to(send.device)
        else:
            recv.copy_(host)
    else:
        if recv is None:
            recv = torch.empty((G,) + tuple(send.shape), dtype=torch.int64, device=send.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    return recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64)


def query_slice(n_queries: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Queries whose merged result `rank` owns in the user-partitioned exchange (contiguous, ceil(B/G) per rank)."""
    return shard_bounds(n_queries, world_size, rank)


def pack_send_buffer(idx, score, dist2, G: int, out=None):
    """(G, 3, Bs, k) int64 send buffer of the user-partitioned exchange, Bs = ceil(B / G): send[g] = (idx, score bits,
    dist2 bits) of the queries of rank g's slice; rows past the batch are padding (id -1, score 0, dist2 +inf: they lose
    every merge).  Device tensors: one hwer_pack_exchange launch."""
    import torch
    B, k = idx.shape
    Bs = (B + G - 1) // G
    send = torch.empty((G, 3, Bs, k), dtype=torch.int64, device=idx.device) if out is None else out
    assert tuple(send.shape) == (G, 3, Bs, k) and send.dtype == torch.int64 and send.is_contiguous()
    if idx.is_cuda:
        from . import native
        idx, score, dist2 = idx.contiguous(), score.contiguous(), dist2.contiguous()
        with native.device_guard(idx.device):
            native.check(native.lib().hwer_pack_exchange(native.ptr(idx), native.ptr(score), native.ptr(dist2), B, k, G,
                                                         native.ptr(send), native.stream_ptr(idx.device)), "pack_exchange")
        return send
    # CPU tensors (the gloo tests of the exchange logic)
    full = B // Bs                      # slices that are completely inside the batch
    rem = B - full * Bs                 # rows of the partial slice `full` (0 when there is none)
    for c, t in enumerate((idx, score.view(torch.int64), dist2.view(torch.int64))):
        if full:
            send[:full, c].copy_(t[:full * Bs].view(full, Bs, k))
        if rem:
            send[full, c, :rem].copy_(t[full * Bs:])
    if full < G:
        inf_bits = int(np.array(np.inf, dtype=np.float64).view(np.int64))
        for c, v in enumerate((-1, 0, inf_bits)):
            send[full, c, rem:].fill_(v)
            if full + 1 < G:
                send[full + 1:, c].fill_(v)
    return send


def exchange_by_query(idx, score, dist2, group=None, send=None, recv=None):
    """User-partitioned exchange (SURVEY.md §8e variant): rank r receives, from every rank g, g's local results for the
    queries of r's slice -> three (G, Bs, k) tensors, Bs = ceil(B/G) (rows beyond the slice are padding: ids -1,
    dist2 +inf).  Per link this moves 1/G of what the all-gather moves: xGMI is point-to-point, so the exchange is G
    concurrent sends of B k 24 / G bytes instead of G broadcasts of B k 24 bytes.  RCCL: one all_to_all_single; other
    backends (gloo in the CPU tests): the same exchange as batched point-to-point operations."""
    import torch
    import torch.distributed as dist
    G = dist.get_world_size(group)
    r = dist.get_rank(group)
    B, k = idx.shape
    if B == 0:
        return idx.new_empty((G, 0, k)), score.new_empty((G, 0, k)), dist2.new_empty((G, 0, k))
    Bs = (B + G - 1) // G
    send = pack_send_buffer(idx, score, dist2, G, out=send)
    staged = send.is_cuda and dist.get_backend(group) != "nccl"        # test harness only: ranks sharing a GPU over gloo
    recv_dev = recv
    if staged:
        send = send.cpu()
        recv = None
    if recv is None:
        recv = torch.empty_like(send)
    if dist.get_backend(group) == "nccl":
        dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)   # equal splits: slice g <-> rank g
    else:
        recv[r] = send[r]
        ops = []
        for g in range(G):
            if g != r:
                peer = dist.get_global_rank(group, g) if group is not None else g
                ops.append(dist.P2POp(dist.isend, send[g], peer, group))
                ops.append(dist.P2POp(dist.irecv, recv[g], peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
    if staged:
        if recv_dev is None:
            recv = recv.to(idx.device)
        else:
            recv_dev.copy_(recv)
            recv = recv_dev
    # strided views of the receive buffer (list stride 3 Bs k): merge_on_device reads them in place
    return recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64)


def merge_on_device(g_idx, g_score, g_dist2, sort_by_score=True, k=None, want_exhausted=False, out=None):
    """hwer_merge_topk_lists over gathered (G, B, k_in) tensors -> (idx, score, dist2) (B, k) tensors, k >= k_in (default k_in);
    with want_exhausted a fourth (B,) int32 tensor: 1 where some list gave all its k_in entries to the winners (k_in < k only:
    that query has to be repeated with full lists)."""
    import torch
    from . import native
    lib = native.lib()
    G, B, k_in = g_idx.shape
    k = k_in if k is None else int(k)
    strides = {t.stride() for t in (g_idx, g_score, g_dist2)}
    if len(strides) != 1 or g_idx.stride()[1:] != (k_in, 1) or (G > 1 and g_idx.stride(0) < B * k_in):
        g_idx, g_score, g_dist2 = g_idx.contiguous(), g_score.contiguous(), g_dist2.contiguous()
    list_stride = g_idx.stride(0) if G > 1 else B * k_in
    if out is not None:      # preallocated (idx, score, dist2, exhausted) of at least B rows (a pipelined caller: no allocation on its side stream)
        o_idx, o_score, o_dist2 = out[0][:B], out[1][:B], out[2][:B]
        o_ex = out[3][:B] if want_exhausted else None
        assert o_idx.shape == (B, k) and o_idx.is_contiguous() and o_score.is_contiguous() and o_dist2.is_contiguous()
    else:
        o_idx = torch.empty((B, k), dtype=torch.int64, device=g_idx.device)
        o_score = torch.empty((B, k), dtype=torch.float64, device=g_idx.device)
        o_dist2 = torch.empty((B, k), dtype=torch.float64, device=g_idx.device)
        o_ex = torch.zeros((B,), dtype=torch.int32, device=g_idx.device) if want_exhausted else None
    with native.device_guard(g_idx.device):
        native.check(lib.hwer_merge_topk_lists(native.ptr(g_idx), native.ptr(g_score), native.ptr(g_dist2), list_stride, G, B, k_in, k,
                                               1 if sort_by_score else 0, native.ptr(o_idx), native.ptr(o_score),
                                               native.ptr(o_dist2), native.ptr(o_ex) if o_ex is not None else None,
                                               native.stream_ptr(g_idx.device)), "merge_topk")
    return (o_idx, o_score, o_dist2, o_ex) if want_exhausted else (o_idx, o_score, o_dist2)


class ShardedSearch:
    """k-NN over a row-sharded catalogue.

    local_search(Q64, Q32, k_local[, rows=positions]) -> (idx, score, dist2) (B, k_local) tensors of the LOCAL shard in ascending
    (dist2, global id) order with GLOBAL ids.  `rows` is passed only when a subset of the batch is repeated with full lists (the
    positions of those queries in the caller's batch, so that per-query state - anchors - follows); a callable WITHOUT a `rows`
    parameter is still accepted: the search then keeps full lists (no shortened lists, hence no repeat).
    merge(g_idx, g_score, g_dist2, sort_by_score) -> merged tensors.
    Both are injectable so that the collective plumbing is testable on CPU (gloo) without a GPU."""

    def __init__(self, n_rows: int, local_search: Callable, merge: Callable = merge_on_device, group=None,
                 pad_value: float = float("inf"), exchange: str = "all_gather", list_len: Optional[int] = None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows = n_rows
        self.lo, self.hi = shard_bounds(n_rows, self.world, self.rank)
        self.local_search, self.merge = local_search, merge
        self.pad_value = pad_value
        assert exchange in ("all_gather", "by_query")
        self.exchange = exchange   # "by_query": every rank ends up with the merged result of ITS query slice only
        # Shortened local lists (all-gather exchange only: every rank sees every list, so all ranks agree on the queries to repeat
        # without another collective).  None: local_list_len(k, world) when the merge is the device kernel, full lists for an
        # injected merge; an int forces that length (the merge must then accept k= / want_exhausted= like merge_on_device).
        self.list_len = list_len
        self.repeated_queries = 0   # queries of the last search() that were repeated with full lists

    @staticmethod
    def _takes_rows(fn) -> bool:
        """does the injected local_search accept `rows=` (the positions of a repeated subset)?  A 3-argument callable written against
        the rounds-1/2 contract does not: such a search keeps full lists (no data-dependent repeat can reach it)."""
        import inspect
        try:
            ps = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return False
        return "rows" in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())

    def _list_len(self, k: int) -> int:
        if self.exchange != "all_gather" or self.world > 16 or not self._takes_rows(self.local_search):
            return k
        if self.list_len is not None:
            return max(min(int(self.list_len), k), -(-k // self.world))
        return local_list_len(k, self.world) if self.merge is merge_on_device else k

    def search(self, Q64, Q32, k: int, sort_by_score: bool = True, _full: bool = False, _rows=None, deferred: bool = False):
        """-> merged (idx, score, dist2).  Shortened local lists (k_in < k): the merge kernel leaves, per query, whether a list ran out
        (`exhausted`, a device tensor); such queries are repeated with full lists - by `settle()`, which this call runs before it
        returns unless `deferred=True`.  A deferred search returns at once (no host synchronisation at all: the count of exhausted
        queries travels to pinned host memory behind an event) and its results are final after the caller's `settle()` - or after
        the NEXT `search()`, which settles its predecessor first, one step late, when that count has long arrived."""
        import torch
        if k > self.n_rows:
            raise ValueError("k must be less than or equal to the number of training points")
        if not _full:
            self.settle()                     # (a deferred predecessor: its count was copied out a whole search ago)
        n_local = self.hi - self.lo
        k_in = k if _full else self._list_len(k)
        k_local = min(k_in, n_local)
        B = Q64.shape[0]
        idx = torch.full((B, k_in), -1, dtype=torch.int64, device=Q64.device)
        score = torch.zeros((B, k_in), dtype=torch.float64, device=Q64.device)
        dist2 = torch.full((B, k_in), self.pad_value, dtype=torch.float64, device=Q64.device)  # short shards pad with +inf
        if k_local > 0:
            # (a repeat passes the positions of its queries in the caller's batch: per-query state of the local search - anchors - follows)
            li, ls, ld = self.local_search(Q64, Q32, k_local) if _rows is None else self.local_search(Q64, Q32, k_local, rows=_rows)
            idx[:, :k_local], score[:, :k_local], dist2[:, :k_local] = li, ls, ld
        if self.exchange == "by_query":
            g = exchange_by_query(idx, score, dist2, self.group)
            m_idx, m_score, m_dist2 = self.merge(*g, sort_by_score)
            lo, hi = query_slice(B, self.world, self.rank)
            return m_idx[:hi - lo], m_score[:hi - lo], m_dist2[:hi - lo]   # rows of queries [lo, hi)
        g = all_gather_results(idx, score, dist2, self.group)
        if k_in == k:
            if not _full:
                self.repeated_queries = 0
            return self.merge(*g, sort_by_score)
        m_idx, m_score, m_dist2, exhausted = self.merge(*g, sort_by_score, k=k, want_exhausted=True)
        # Every rank merged the same gathered lists with the same kernel: `exhausted` is identical on all ranks, so they all take (or
        # skip) the repeat - a second, full-list exchange for those queries only - together.  The mask stays on the device; only its
        # SUM goes to the host, asynchronously (device tensors: pinned memory + an event; the CPU tensors of the gloo tests: at once).
        count = exhausted.sum(dtype=torch.int64)
        if count.is_cuda:
            host = torch.zeros(1, dtype=torch.int64).pin_memory()
            host.copy_(count.view(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(count.device))
        else:
            host, ev = count.view(1).clone(), None
        if deferred:
            # the repeat runs one search later: a pipelined caller has refilled its resident query buffers by then - keep the queries of THIS
            # search (a device-side copy, no host synchronisation; advisor finding of round 5)
            Q64 = Q64.clone()
            Q32 = Q32.clone() if Q32 is not None else None
        self._pending = (ev, host, exhausted, Q64, Q32, k, sort_by_score, (m_idx, m_score, m_dist2))
        if not deferred:
            self.settle()
        return m_idx, m_score, m_dist2

    _pending = None

    def settle(self) -> int:
        """Finish the last shortened-list search: wait for its exhausted-query count (8 bytes, copied out behind an event when the
        search was issued), and if it is non-zero repeat exactly those queries with full lists and patch the result tensors the
        search returned, in place.  -> the number of repeated queries (also `repeated_queries`).  Every rank sees the same count."""
        import torch
        p, self._pending = self._pending, None
        if p is None:
            return 0
        ev, host, exhausted, Q64, Q32, k, sort_by_score, (m_idx, m_score, m_dist2) = p
        if ev is not None:
            ev.synchronize()
        self.repeated_queries = int(host[0])
        if self.repeated_queries:
            again = torch.nonzero(exhausted, as_tuple=False).flatten()
            r_idx, r_score, r_dist2 = self.search(Q64[again], Q32[again] if Q32 is not None else None, k, sort_by_score, _full=True, _rows=again)
            m_idx[again], m_score[again], m_dist2[again] = r_idx, r_score, r_dist2
        return self.repeated_queries


class ShardedStepPipeline:
    """The N-GPU step of the hot path as ONE library object (round 5; `bench.py --gpus N` times exactly this): a resident
    `BlendSearchStep` over this rank's shard (local (distance, id) order, shortened local lists of `k_in` entries, two result sets)
    followed by ONE exchange - `all_gather` (every rank ends with every query's result) or `by_query` (the user-partitioned
    all-to-all: a rank ends with its slice of the queries) - and the merge kernel, the exchange + merge of step i on a side stream
    under the search of step i + 1.  Nothing in `run()` synchronises with the host: queries whose merge saw a shortened list run
    out are OR-ed into a device mask over all steps (`exhausted_queries()` reads it - after the job, or whenever the caller can
    afford a synchronisation); such a job is repeated with full lists (`k_in = k`), as `ShardedSearch.settle()` does per search.

        step = BlendSearchStep(tables, alpha, ..., k=k_in, want_dist=True, sort_by_score=False, n_sets=2)
        pipe = ShardedStepPipeline(step, k=200, exchange="all_gather")
        for _ in range(n): merged = pipe.run()          # (idx, score, dist2) of the step just issued; final after pipe.wait(),
                                                        # overwritten by the run() after next (the pipeline owns two sets of buffers)
        pipe.wait(); assert pipe.exhausted_queries() == 0
    """

    def __init__(self, step, k: int, exchange: str = "all_gather", group=None, sort_by_score: bool = True, time_events: bool = False):
        import torch
        assert exchange in ("all_gather", "by_query")
        assert len(step.sets) >= 2 and step.sets[0]["dist2"] is not None and not step.sort_by_score, \
            "the local step keeps two result sets in ascending (distance, id) order with distances (n_sets=2, want_dist=True, sort_by_score=False)"
        self.step, self.k, self.k_in, self.exchange, self.group = step, int(k), int(step.k), exchange, group
        self.sort_by_score = bool(sort_by_score)
        dev = self.dev = step.dev
        self.comm = torch.cuda.Stream(device=dev)
        self.ev_searched = [torch.cuda.Event() for _ in range(2)]
        self.ev_merged = [torch.cuda.Event() for _ in range(2)]
        self.n_steps = 0
        self.ex_accum = torch.zeros((step.B,), dtype=torch.int32, device=dev)
        # Every buffer the side stream touches is allocated HERE, once per result set (round 6; round 5 allocated the receive buffer and the merge
        # outputs inside run() under `torch.cuda.stream(comm)` and handed them to the caller's stream without record_stream: the caching allocator
        # could have given such a block back to the side stream while the caller still read it).  Lifetime rule instead of allocator bookkeeping:
        # the tensors run() returns for step i are overwritten by step i + 2 - stream-ordered behind everything the caller enqueued on its stream
        # before that run() call (the side stream waits for ev_searched of step i + 2, recorded on the caller's stream).
        import torch.distributed as dist
        G = dist.get_world_size(group)
        B, k_in, K = step.B, self.k_in, self.k
        Bs = (B + G - 1) // G
        Bm = Bs if exchange == "by_query" else B          # rows of the merged result a rank ends with
        n_send = G if exchange == "by_query" else 1
        self.send = [torch.empty((n_send, 3, Bs if exchange == "by_query" else B, k_in), dtype=torch.int64, device=dev) for _ in range(2)]
        self.recv = [torch.empty((G, 3, Bs if exchange == "by_query" else B, k_in), dtype=torch.int64, device=dev) for _ in range(2)]
        self.out = [(torch.empty((Bm, K), dtype=torch.int64, device=dev), torch.empty((Bm, K), dtype=torch.float64, device=dev),
                     torch.empty((Bm, K), dtype=torch.float64, device=dev), torch.zeros((Bm,), dtype=torch.int32, device=dev)) for _ in range(2)]
        self.merged = [tuple(o[:3]) for o in self.out]
        self.time_events = bool(time_events)
        self.event_log = []        # time_events: (exchange start, collective done, merge done) events of every step

    def run(self):
        """issue one step on the current stream (search) and the side stream (exchange + merge); -> this step's merged tensors"""
        torch = self.step.torch
        s = self.n_steps & 1
        self.n_steps += 1
        o = self.step.sets[s]
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(self.ev_merged[s])          # the exchange that read this result set two steps ago is done
        self.step.step(o)
        self.ev_searched[s].record(cur)
        te = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.time_events else None
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ev_searched[s])
            if te:
                te[0].record(self.comm)
            g = (exchange_by_query if self.exchange == "by_query" else all_gather_results)(o["idx"], o["score"], o["dist2"], self.group,
                                                                                           send=self.send[s], recv=self.recv[s])
            if te:
                te[1].record(self.comm)
            if self.k_in < self.k:
                *m3, ex = merge_on_device(*g, self.sort_by_score, k=self.k, want_exhausted=True, out=self.out[s])
                self.ex_accum[:ex.numel()].bitwise_or_(ex)
            else:
                m3 = merge_on_device(*g, self.sort_by_score, out=self.out[s])
            self.merged[s] = tuple(m3[:3])
            if te:
                te[2].record(self.comm)
                self.event_log.append(tuple(te))
            self.ev_merged[s].record(self.comm)
        return self.merged[s]

    def wait(self):
        """the current stream waits for every exchange + merge issued so far (no host synchronisation)"""
        torch = self.step.torch
        cur = torch.cuda.current_stream(self.dev)
        for e in self.ev_merged:
            cur.wait_event(e)

    def last_local_set(self):
        return self.step.sets[(self.n_steps - 1) & 1]

    def exhausted_queries(self) -> int:
        """queries for which some step's merge saw a shortened list run out (synchronises); non-zero: repeat the job with k_in = k"""
        return int(self.ex_accum.sum().item())


def sharded_index_search(index, node_type: str, shard: ShardedSearch, Q64, Q32, k, anchors_dev=None, score_mode=None,
                         sort_by_score=True):
    """Convenience: DeviceIndex whose `node_type` table holds only this rank's rows [shard.lo, shard.hi)."""
    def local(q64, q32, kl, rows=None):   # rows: positions of a repeated subset of the batch (shortened lists that ran out)
        an = anchors_dev if (rows is None or anchors_dev is None) else anchors_dev[rows]
        i, s, d2, _ = index.search_queries(node_type, q64, q32, kl, an, score_mode, sort_by_score=False,
                                           id_base=shard.lo)
        return i, s, d2
    shard.local_search = local
    return shard.search(Q64, Q32, k, sort_by_score)
