#!/usr/bin/env python3
"""Benchmark of the hwer retrieval hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  blend of the content + collaborative tables with alpha weights -> L2-normalise -> unit-norm audit ->
  per-type item table (+ half norms) -> query construction for every user -> exact top-K retrieval
  (float32 MFMA screen + float64 re-rank + re-score + sort), K = 200.
`value` = user-item pairs scored per second = users * items / step time (whole job, all GPUs).

Default (N = 1): BASELINE.json configs[1] — ML-1M-shaped synthetic, 6040 users x 3706 items, d = 128, two tables,
alpha = 0.5.  N > 1 (launched by torch.distributed.run, one rank per GPU): the item catalogue is row-sharded, every
rank holds a shard of the SAME size as the N = 1 catalogue (weak scaling), queries are replicated, and the local
top-K are exchanged in ONE step over RCCL - user-partitioned: every rank receives the lists of its slice of the users
from all ranks (all_to_all_single) and merges those; HWER_BENCH_EXCHANGE=all_gather selects the replicated variant.

Prints ONE JSON line (rank 0).  See DESIGN.md §6 for the roofline arithmetic.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

CONFIGS = {
    # name: users, items per GPU, d, tables, per-row alpha, queries per step (None = all users), description
    "c1": dict(U=943, I=1682, d=64, T=2, per_row=False, B=None, name="ML-100K-shaped synthetic (943x1682, d=64)"),
    "c2": dict(U=6040, I=3706, d=128, T=2, per_row=False, B=None, name="ML-1M-shaped synthetic (6040x3706, d=128, alpha=0.5)"),
    "c3": dict(U=138493, I=26744, d=128, T=2, per_row=False, B=None, name="ML-20M-shaped synthetic (138493x26744, d=128)"),
    "c4": dict(U=4096, I=1250000, d=128, T=2, per_row=False, B=None, name="10M-item catalogue shard (4096 queries x 1.25M items/GPU, d=128)"),
    "c5": dict(U=4096, I=1250000, d=256, T=4, per_row=True, B=None, name="multi-alpha hybrid (3 content + 1 collaborative, per-row alpha, d=256)"),
}
K = 200
PROFILE_EVERY = 5
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense BF16 / F16 (only priced against with --f16-screen)
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E 8 TB/s spec


def unit_rows(torch, gen, n, d, device):
    x = torch.randn((n, d), generator=gen, device=device, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)



def effective_cpus():
    """CPU threads this process can actually keep busy: the smallest of os.cpu_count(), the affinity mask and the cgroup CPU
    quota (a GPU box exposes all host CPUs but caps the container: more runnable threads than the quota get throttled and
    run SLOWER - measured 256 threads: 0.16x of 16 threads)."""
    n = os.cpu_count() or 1
    note = "os.cpu_count()=%d" % n
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                q = max(1, int(float(quota) / float(period) + 0.5))
                if q < n:
                    note += ", cgroup cpu.max=%s/%s -> %d" % (quota, period, q)
                    n = q
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0 and max(1, (q + per // 2) // per) < n:
            n = max(1, (q + per // 2) // per)
            note += ", cfs quota -> %d" % n
    except (OSError, ValueError):
        pass
    return n, note

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f32-screen", action="store_true",
                    help="switch HWER_KNN_F16_SCREEN off.  By default the streaming screen (catalogues above 8 192 rows: c3, c4, c5) runs "
                         "on a float16 copy of the table, rebuilt every step like the float32 one (results unchanged, exact float64 "
                         "re-rank); the dense path of the default configuration (c2) and c1 always screen in float32")
    ap.add_argument("--f16-dense", action="store_true",
                    help="experiment: also run the DENSE screen (catalogues up to 8 192 rows: c1, c2) on a float16 copy of the table; "
                         "not the default (DESIGN.md 4.2b)")
    ap.add_argument("--sweep-only", action="store_true",
                    help="profiling runs: skip the one-anchor latency and pair-score measurements after the timed region, so that every "
                         "launch of a kernel in a rocprofv3 summary is a launch of the timed sweep")
    ap.add_argument("--cpu-sample", type=int, default=0, help="users in the CPU-baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hwer import native
    from hwer.device_index import roundup
    from hwer.distributed import all_gather_results, exchange_by_query, merge_on_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    backend = os.environ.get("HWER_BENCH_BACKEND", "nccl")     # "gloo" only to exercise the N > 1 code path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = native.lib()
    cfg = CONFIGS[args.config]
    U, I, d, T = cfg["U"], cfg["I"], cfg["d"], cfg["T"]
    N = U + I                      # rows held by this rank: replicated users + the local item shard
    ldv = roundup(d, 4)
    ldx = lib.hwer_knn_padded_dim(d)
    S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    P = native.ptr

    # ---------------- synthetic inputs, resident in HBM ------------------------------------------------------
    gen = torch.Generator(device=dev)
    tables = []
    for t in range(T):
        gen.manual_seed(3000 + 10 * 2 + t)                       # users: identical on every rank
        users = unit_rows(torch, gen, U, d, dev)
        gen.manual_seed(100003 * (rank + 1) + 3000 + t)          # items: this rank's shard
        items = unit_rows(torch, gen, I, d, dev)
        tables.append(torch.cat([users, items]).contiguous())
    if cfg["per_row"]:
        g = torch.Generator(device="cpu").manual_seed(5005)
        # Dirichlet(1, .., 1) = normalised unit exponentials, drawn from the seeded generator (torch.distributions has no generator argument)
        alpha = -torch.log(torch.rand((N, T), generator=g, dtype=torch.float64).clamp_min(1e-300))
        alpha = (alpha / alpha.sum(1, keepdim=True)).to(torch.float32)
        cold = torch.rand(N, generator=g) < 0.2
        alpha[cold, T - 1] = 0.0
        alpha = (alpha / alpha.sum(1, keepdim=True)).to(dev).contiguous()
    else:
        alpha = torch.full((T,), 1.0 / T, dtype=torch.float32, device=dev)
    tab_ptrs = torch.tensor([t.data_ptr() for t in tables], dtype=torch.int64, device=dev)
    item_rows = torch.arange(U, N, dtype=torch.int64, device=dev)
    anchors = torch.arange(0, U, dtype=torch.int32, device=dev)

    # ---------------- preallocated outputs / workspace ----------------------------------------------------------
    V = torch.zeros((N, ldv), dtype=torch.float32, device=dev)
    norm_ws = torch.empty(lib.hwer_unit_audit_ws_floats(N), dtype=torch.float32, device=dev)
    audit_buf = torch.zeros(6, dtype=torch.int64, device=dev)      # counts[4] | sum |norm-1| (f64) | pad: one 48-byte memset
    audit_counts = audit_buf[:4]
    audit_dev = audit_buf[4:5].view(torch.float64)
    I_pad = lib.hwer_knn_padded_rows(I)
    X = torch.empty((I_pad, ldx), dtype=torch.float32, device=dev)
    hnorm = torch.empty(I_pad, dtype=torch.float32, device=dev)
    f16_screen = ((not args.f32_screen) and I > 8192) or (args.f16_dense and I <= 8192)
    X16 = torch.empty((I_pad, ldx), dtype=torch.float16, device=dev) if f16_screen else None
    Q64 = torch.empty((U, d), dtype=torch.float64, device=dev)
    Q32 = torch.empty((U, ldx), dtype=torch.float32, device=dev)
    # N > 1: two result sets, so that the exchange + merge of step i (side stream) overlaps the search of step i + 1
    n_sets = 2 if world > 1 else 1
    sets = [dict(idx=torch.empty((U, K), dtype=torch.int64, device=dev), score=torch.empty((U, K), dtype=torch.float64, device=dev),
                 dist2=torch.empty((U, K), dtype=torch.float64, device=dev), flags=torch.empty(U, dtype=torch.int32, device=dev))
            for _ in range(n_sets)]
    o_idx, o_score, o_dist2, o_flags = sets[0]["idx"], sets[0]["score"], sets[0]["dist2"], sets[0]["flags"]
    CH = 131072   # = DeviceIndex.QUERY_CHUNK: one full round of workgroups per launch
    ws_bytes = lib.hwer_knn_workspace_bytes(min(U, CH), I, d, K)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    # rows passed the 1e-4 unit-norm audit (checked after the timed region) => 0.5||x||^2 <= 0.5 (1 + 1e-4)^2
    hnorm_max = 0.5 * (1.0 + 1e-4) ** 2
    x16_exp = int(lib.hwer_knn_f16_scale_exp(hnorm_max)) if f16_screen else 0
    id_base = U + rank * I          # global row id of this rank's first item (users first, then shards in rank order)
    arith = 0                       # float64 score arithmetic (ContentRecommendation flavour)
    # find_closest_neighbours returns (node, score): distances are an output only where the shards' results are merged
    want_dist = world > 1

    def search_into(o):
        # blend + L2-normalise with the unit-norm audit of the output rows fused into the same pass
        native.check(lib.hwer_blend_normalize_audit_f32(P(tab_ptrs), d, None, P(alpha), 1 if cfg["per_row"] else 0, T, N, d,
                                                        P(V), ldv, 1e-4, P(audit_counts), P(audit_dev), S()), "blend+audit")
        native.check(lib.hwer_pack_rows_f32(P(V), N, d, ldv, P(item_rows), I, P(X), ldx, P(hnorm), S()), "pack")
        if f16_screen:
            native.check(lib.hwer_pack_rows_f16(P(X), I_pad, ldx, x16_exp, P(X16), S()), "pack16")
        native.check(lib.hwer_query_mix(P(V), N, d, ldv, P(anchors), None, None, None, None, U, arith, P(Q64), P(Q32),
                                        ldx, S()), "query_mix")
        oi, osc, od, of = o["idx"], o["score"], o["dist2"], o["flags"]
        for b0 in range(0, U, CH):
            b1 = min(U, b0 + CH)
            native.check(lib.hwer_knn_search_opts_f32(P(X), P(hnorm), I, d, ldx, P(Q32[b0:b1]), P(Q64[b0:b1]), P(V), ldv,
                                                      P(anchors[b0:b1]), b1 - b0, K, arith, 1 if world == 1 else 0, None, id_base,
                                                      hnorm_max, P(oi[b0:b1]), P(osc[b0:b1]), P(od[b0:b1]) if want_dist else None,
                                                      P(of[b0:b1]), P(ws), ws.numel(), 2 if f16_screen else 0,
                                                      P(X16) if f16_screen else None, x16_exp, S()), "knn_search")
            # queries the screen could not certify (rare: near-ties at the cut, or a sample-scaled threshold that was too
            # tight) are recomputed by the exhaustive float64 pass INSIDE the step, on the device (no host round trip):
            # the step always returns exact results.  `missed` counts listed queries beyond the 8 one call handles
            # (checked after the timed region: must stay 0).
            native.check(lib.hwer_knn_exact_device(P(X), I, d, ldx, P(Q64[b0:b1]), P(V), ldv, P(anchors[b0:b1]), K, arith,
                                                   1 if world == 1 else 0, None, id_base, P(oi[b0:b1]), P(osc[b0:b1]),
                                                   P(od[b0:b1]) if want_dist else None, P(of[b0:b1]), P(ws), P(missed), P(ws_exact), ws_exact.numel(), S()),
                         "knn_exact_device")

    step_no = [0]

    def step():
        if world == 1:
            search_into(sets[0])
            return o_idx, o_score, o_dist2
        # N > 1.  ONE exchange step per search: user-partitioned (every rank receives the local lists of ITS slice of the
        # users from all ranks and merges those - 1/N of the all-gather's bytes per xGMI link), or the all-gather variant.
        # The exchange + merge run on a side stream: while RCCL moves the lists of step i over xGMI, the compute stream is
        # already searching step i + 1 into the other result set.  Every step's merge still completes inside the timed
        # region (the closing barrier synchronises all streams).
        s = step_no[0] & 1
        step_no[0] += 1
        o = sets[s]
        cur = torch.cuda.current_stream()
        cur.wait_event(ev_merged[s])          # the exchange that read this result set two steps ago is done
        search_into(o)
        ev_searched[s].record(cur)
        with torch.cuda.stream(comm):
            comm.wait_event(ev_searched[s])
            g = exchange_by_query(o["idx"], o["score"], o["dist2"]) if exchange == "by_query" else all_gather_results(o["idx"], o["score"], o["dist2"])
            merged[s] = merge_on_device(*g, True)
            ev_merged[s].record(comm)
        return merged[s]

    comm = torch.cuda.Stream(device=dev) if world > 1 else None
    ev_searched = [torch.cuda.Event() for _ in range(n_sets)]
    ev_merged = [torch.cuda.Event() for _ in range(n_sets)]
    merged = [None] * n_sets
    missed = torch.zeros(1, dtype=torch.int32, device=dev)
    exchange = os.environ.get("HWER_BENCH_EXCHANGE", "by_query")
    ws_exact = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    assert d == ldv, "bench tables are passed with row stride d"
    for _ in range(args.warmup):
        res = step()
    barrier()
    lib.hwer_profile_enable(PROFILE_EVERY)   # HIP events around every PROFILE_EVERY-th search (each pair costs ~10 us of stream time)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.hwer_profile_enable(0)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tot = ctypes.c_double()
    cnt = ctypes.c_int64()
    lib.hwer_profile_read(0, ctypes.byref(tot), ctypes.byref(cnt))
    screen_ms = tot.value / max(cnt.value, 1)
    searches_per_step = (U + CH - 1) // CH
    screen_launches_per_step = float(searches_per_step)
    lib.hwer_profile_read(1, ctypes.byref(tot), ctypes.byref(cnt))
    select_ms = tot.value / max(cnt.value, 1)

    # ---------------- post-hoc checks (outside the timed region) ------------------------------------------------
    diag = bool(os.environ.get("HWER_BENCH_NOSYNC"))   # diagnosis runs (ablation switches) skip the result checks
    pos, neg, nan = audit_counts.tolist()[:3]
    assert pos + neg + nan == 0, "unit-norm audit failed: %s" % ((pos, neg, nan),)
    last = sets[(step_no[0] - 1) & 1] if world > 1 else sets[0]   # the result set of the last step
    n_uncert = int((last["flags"] == 0).sum().item())
    assert diag or n_uncert == 0, "%d queries left without an exact result" % n_uncert
    assert diag or int(missed.item()) == 0, "the on-device exhaustive pass left %d queries to the host path" % int(missed.item())
    n_exhaustive = int((last["flags"] == 2).sum().item())
    idx_h = res[0].cpu().numpy()
    score_h = res[1].cpu().numpy()
    parity_n = 0
    if world == 1 and not diag:
        from oracle import c_oracle
        Vh = V[:, :d].cpu().numpy()
        sel = np.unique(np.linspace(0, U - 1, 16).astype(np.int64))
        V64 = Vh.astype(np.float64)
        Qh = V64[sel] / np.sqrt(np.stack([(V64[s] * V64[s]).sum() for s in sel]))[:, None]
        # oracle query = unit(anchor) in numpy's own arithmetic
        from oracle import hwer_oracle as orc
        Qh = np.stack([orc.query_vector(V64, int(s)) for s in sel])
        oi, osc, _ = c_oracle.find_closest_f64(Vh[U:], Qh, V64[sel], K, threads=effective_cpus()[0])
        assert np.array_equal(idx_h[sel], oi + U), "GPU indices differ from the oracle"
        assert np.array_equal(score_h[sel], osc), "GPU scores differ from the oracle"
        parity_n = len(sel)
    elif world > 1 and not diag and I * world <= 400000:
        # N > 1: rank 0 collects every rank's item shard (rank order = global id order) and checks the merged lists of
        # sampled users of ITS slice against the oracle's brute force over the whole catalogue
        shard = V[U:, :d].contiguous()
        if backend == "nccl":
            allv = torch.empty((world * I, d), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(allv, shard)
            items_h = allv.cpu().numpy()
        else:
            parts = [torch.empty((I, d), dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, shard.cpu())
            items_h = torch.cat(parts).numpy()
        if rank == 0:
            from oracle import c_oracle
            from oracle import hwer_oracle as orc
            n_mine = idx_h.shape[0] if exchange != "by_query" else (U + world - 1) // world   # rank 0 owns users [0, ceil(U/N))
            sel = np.unique(np.linspace(0, min(n_mine, U) - 1, 16).astype(np.int64))
            V64 = V[:U, :d].cpu().numpy().astype(np.float64)
            Qh = np.stack([orc.query_vector(V64, int(q)) for q in sel])
            oi, osc, _ = c_oracle.find_closest_f64(items_h, Qh, V64[sel], K, threads=effective_cpus()[0])
            assert np.array_equal(idx_h[sel], oi + U), "merged indices differ from the oracle (N = %d)" % world
            assert np.array_equal(score_h[sel], osc), "merged scores differ from the oracle (N = %d)" % world
            parity_n = len(sel)

    total_items = I * world
    pairs_per_step = float(U) * float(total_items)
    ms_per_step = elapsed / args.steps * 1e3
    value = pairs_per_step / (elapsed / args.steps)
    # dominant kernel: the float32 MFMA screen.  Algorithmic work per launch = 2 * B * N_t * d flops
    launches = max(1.0, screen_launches_per_step)
    flops_per_launch = 2.0 * U * I * d / launches
    achieved = flops_per_launch / (screen_ms * 1e-3) / 1e12 if screen_ms > 0 else 0.0
    # HBM traffic of the screen kernel per launch, measured with rocprofv3 PMC passes for the SAME command and kept in
    # profiles/r1_pmc_traffic.json: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes (FETCH_SIZE reads half the bytes of wide
    # loads on gfx950 - MI355X_MICROARCH.md §HBM).  None when no profile exists for this config / GPU count.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as fh:
            traffic = json.load(fh).get(args.config if world == 1 else "", {}).get("knn_screen_bytes_per_launch")
    except OSError:
        pass
    out = {
        "metric": "user-item pairs/sec scored (blend + L2-normalise + exact top-%d retrieval)" % K,
        "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if f16_screen else "f32", "data": "synthetic",
        "config": {"workload": cfg["name"], "users": U, "items_per_gpu": I, "items_total": total_items, "d": d, "k": K,
                   "tables": T, "alpha": "per-row" if cfg["per_row"] else "scalar", "f16_screen": f16_screen, "sharding": ("items row-sharded x%d, %s + merge, exchange of step i overlapped with the search of step i+1" % (world, "user-partitioned exchange" if exchange == "by_query" else "all-gather")) if world > 1 else "none"},
        "roofline": {"kernel": ("knn_screen (v_mfma_f32_32x32x8_f16 on a float16 copy of the table + streaming top-K' filter; exact float64 re-rank)"
                                if f16_screen else "knn_screen (v_mfma_f32_32x32x2_f32 + streaming top-K' filter)"), "bound": "mfma",
                     "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS if f16_screen else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / (MFMA_F16_PEAK_TFLOPS if f16_screen else MFMA_F32_PEAK_TFLOPS),
                     **({"vs_fp32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
                         "note": "float16 screen: the kernel is bound by its filter / LDS-DMA phases, not by the float16 MFMA pipe; "
                                 "vs_fp32_mfma_peak > 1 = faster than any float32-MFMA screen could be"} if f16_screen else {}),
                     "traffic": traffic, "avg_launch_ms": screen_ms, "launches_per_step": launches},
        "kernels_ms": {"knn_screen": screen_ms, "knn_select_rerank": select_ms},
        "recall_latency_ms": ms_per_step, "certified_queries": U - n_exhaustive, "exhaustive_fallback_queries": n_exhaustive,
        "parity_checked_queries": parity_n,
    }

    # ---------------- per-query latency: one anchor against this rank's catalogue, outside the timed region ----
    if world == 1 and not args.sweep_only:
        ws1 = torch.empty(lib.hwer_knn_workspace_bytes(1, I, d, K), dtype=torch.uint8, device=dev)

        def one_query():
            native.check(lib.hwer_knn_search_opts_f32(P(X), P(hnorm), I, d, ldx, P(Q32), P(Q64), P(V), ldv, P(anchors), 1, K, arith, 1, None,
                                                      id_base, hnorm_max, P(o_idx), P(o_score), None, P(o_flags), P(ws1), ws1.numel(), 0,
                                                      P(X16) if f16_screen else None, x16_exp, S()), "knn_search(B=1)")
        for _ in range(3):
            one_query()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            one_query()
        e1.record()
        torch.cuda.synchronize()
        out["query_latency_ms"] = {"batch": 1, "value": e0.elapsed_time(e1) / 20.0, "note": "one anchor, exact top-%d, catalogue resident%s" % (K, ", float16 storage" if f16_screen else "")}

        # ---------------- pair scoring (P / t), outside the timed region: P random (user, item) pairs of this table ----
        n_pairs = 1000209 if args.config in ("c1", "c2") else 16000000
        gp = torch.Generator(device=dev).manual_seed(4001)
        p_src = torch.randint(0, U, (n_pairs,), generator=gp, device=dev, dtype=torch.int32)
        p_dst = torch.randint(U, N, (n_pairs,), generator=gp, device=dev, dtype=torch.int32)
        p_out = torch.empty(n_pairs, dtype=torch.float64, device=dev)

        def score_pairs():
            native.check(lib.hwer_pair_score_f64(P(V), N, d, ldv, P(p_src), P(p_dst), n_pairs, P(p_out), S()), "pair_score")
        for _ in range(3):
            score_pairs()
        e0.record()
        for _ in range(20):
            score_pairs()
        e1.record()
        torch.cuda.synchronize()
        ps_ms = e0.elapsed_time(e1) / 20.0
        ps_bytes = n_pairs * (8.0 * d + 8 + 8)     # two rows + two int32 ids + one float64 result per pair
        out["pair_score"] = {"pairs": n_pairs, "value": n_pairs / ps_ms * 1e3, "unit": "pairs/s", "ms": ps_ms, "dtype": "f64",
                             "roofline": {"bound": "hbm", "achieved": ps_bytes / ps_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": ps_bytes / ps_ms / 1e6 / HBM_PEAK_GBS}}

    # ---------------- CPU baseline: the oracle's C restatement on the host cores (rank 0, N = 1 only) -------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        from oracle import hwer_oracle as orc
        Vh = V[:, :d].cpu().numpy()
        V64 = Vh.astype(np.float64)
        cores, cores_note = effective_cpus()
        n_s = args.cpu_sample or int(min(U, max(64, 6.0e9 * cores / (float(I) * d * 8))))  # ~10-20 s of CPU work
        sel = np.linspace(0, U - 1, n_s).astype(np.int64)
        Qh = V64[sel] / np.linalg.norm(V64[sel], axis=1, keepdims=True)
        c_oracle.find_closest_f64(Vh[U:U + min(I, 2048)], Qh[:8], V64[sel[:8]], min(K, 64), threads=cores)  # warm up the thread pool
        # bounded sample of the same workload, repeated until ~8 s of CPU work have been timed (at least one full pass)
        passes, t_all = 0, 0.0
        while passes < 1 or (t_all < 8.0 and passes < 64):
            t1 = time.perf_counter()
            c_oracle.find_closest_f64(Vh[U:], Qh, V64[sel], K, threads=cores)
            t_all += time.perf_counter() - t1
            passes += 1
        t_all /= passes
        n_1 = max(8, n_s // (4 * cores))
        t1 = time.perf_counter()
        c_oracle.find_closest_f64(Vh[U:], Qh[:n_1], V64[sel[:n_1]], K, threads=1)
        t_one = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n_s * float(I) / t_all, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": "%d of %d users x %d items, %d passes, exact float64 brute force + re-score + sort (oracle/hwer_oracle.c, OpenMP)" % (n_s, U, I, passes),
                               "seconds": t_all, "single_thread_value": n_1 * float(I) / t_one, "cores_note": cores_note}
        out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        # best-effort VECTORISED CPU path: batched float32 Q @ X^T (BLAS, all cores) + argpartition + sort
        # + exact float64 re-score of the k kept per user.  Not the reference's algorithm (its KD-tree is exact in float64):
        # reported so that the GPU/CPU ratio can also be read against the strongest CPU formulation of the same sweep.
        try:
            n_v = int(min(U, 4096, max(64, 2.0e9 / (4.0 * I))))     # key matrix of the sample <= 2 GB
            v_sel = np.linspace(0, U - 1, n_v).astype(np.int64)
            Xh = np.ascontiguousarray(Vh[U:U + I])
            hn = 0.5 * (Xh.astype(np.float64) ** 2).sum(1).astype(np.float32)

            def vectorised():
                Qv = np.ascontiguousarray(Vh[v_sel])
                keys = Qv @ Xh.T
                keys -= hn[None, :]
                part = np.argpartition(-keys, K - 1, axis=1)[:, :K]
                ks = np.take_along_axis(keys, part, axis=1)
                order = np.argsort(-ks, axis=1, kind="stable")
                top = np.take_along_axis(part, order, axis=1)
                sc = (np.einsum("bkd,bd->bk", V64[U + top], V64[v_sel]) + 1.0) / 2.0
                return top, sc
            try:
                from threadpoolctl import threadpool_info, threadpool_limits
                limiter = threadpool_limits(limits=cores, user_api="blas")
            except Exception:
                threadpool_info = limiter = None
            n_runs = 5 if I <= 50000 else 2      # a pass over a 1.25 M-item shard takes seconds
            if n_runs == 5:
                vectorised()
            ts = []
            for _ in range(n_runs):
                t1 = time.perf_counter()
                v_top, _ = vectorised()
                ts.append(time.perf_counter() - t1)
            t_v = float(np.median(ts))
            blas = None
            if threadpool_info is not None:
                blas = [{"api": x.get("internal_api"), "threads": x.get("num_threads")} for x in threadpool_info() if x.get("user_api") == "blas"]
                limiter.restore_original_limits()
            agree = float(np.mean([len(np.intersect1d(v_top[i], idx_h[v_sel[i]] - U)) / float(K) for i in range(0, n_v, max(1, n_v // 256))]))
            out["cpu_baseline"]["vectorised_f32"] = {"value": n_v * float(I) / t_v, "unit": "pairs/s", "cores": cores, "blas": blas,
                                                     "sample": "%d users x %d items, median of %d" % (n_v, I, n_runs), "seconds": t_v,
                                                     "note": "float32 BLAS screen + argpartition + float64 re-score of k: approximate at the cut",
                                                     "neighbour_overlap_with_gpu": agree}
            out["speedup_vs_cpu_vectorised_f32"] = value / out["cpu_baseline"]["vectorised_f32"]["value"]
        except MemoryError:
            pass
        # the reference-SHAPED loop (scikit-learn KD-tree + per-anchor Python loop, one thread: SURVEY.md §8d), on a small sample
        if I <= 50000:
            try:
                n_ref = 48
                ref_sel = np.linspace(0, U - 1, n_ref).astype(np.int64)
                t1 = time.perf_counter()
                r_idx, r_sc = orc.reference_shaped_sweep(V64, np.arange(U, U + I), ref_sel.tolist(), K)
                t_ref = time.perf_counter() - t1
                same = bool(np.array_equal(r_idx, idx_h[ref_sel])) if world == 1 else None
                out["cpu_baseline"]["reference_shaped"] = {"value": n_ref * float(I) / t_ref, "unit": "pairs/s", "cores": 1,
                                                           "ms_per_anchor": t_ref / n_ref * 1e3, "includes": "KD-tree build + %d anchors" % n_ref,
                                                           "same_neighbours_as_gpu": same}
            except ImportError:
                pass
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
