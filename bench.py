#!/usr/bin/env python3
"""Benchmark of the hwer retrieval hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5|c4full]

One "step" = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  blend of the content + collaborative tables with alpha weights -> L2-normalise -> unit-norm audit ->
  per-type item table (+ half norms, + float16 copy above 8 192 rows) -> query construction for every user ->
  exact top-K retrieval (MFMA screen + float64 re-rank + re-score + sort; on-device exhaustive pass for any query the
  screen could not certify), K = 200.
`value` = user-item pairs scored per second = users * items / step time (whole job, all GPUs).

N = 1, no --config (what the driver runs): ONE JSON line whose top level is BASELINE.json's metric on the shape it is quoted
on - ML-100K-shaped synthetic (943 x 1682, d = 64) - with `cpu_baseline`, `roofline` (every kernel >= 10 % of the step),
`api_latency_ms` (the same sweep through the drop-in Python classes, results on the host), and a `configs` object holding the
same measurements for c2 (ML-1M shape), c3 (ML-20M shape: the roofline configuration), c4 (one 1.25 M-item shard, B in
{1, 64, 4096}) and c4full (the whole 10 M-item catalogue on this GPU: the 1-GPU anchor of the scaling curve below).

N > 1: STRONG scaling of one fixed 10 M-item catalogue (BASELINE.json configs[3]; d = 128, 4 096 queries per step): the
catalogue is row-sharded over the ranks (same rows for every N: generated per 65 536-row block from the block's seed),
queries are replicated, each rank searches its shard and the local top-K are exchanged in ONE step over RCCL - an all-gather
(north_star; every rank ends with the full result) - then merged by hwer_merge_topk_strided.  The user-partitioned
all-to-all variant is timed after it and reported under `variants`.  Launch: `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE from the environment), or plain
`python bench.py --gpus N`, which starts the N ranks itself as child processes (before anything touches a GPU).

Prints ONE JSON line (rank 0).  DESIGN.md §6 explains every field and the roofline arithmetic.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

CONFIGS = {
    # users (queries per step), items, d, tables, per-row alpha
    "c1": dict(U=943, I=1682, d=64, T=2, per_row=False, name="ML-100K-shaped synthetic (943x1682, d=64, alpha=0.5)"),
    "c2": dict(U=6040, I=3706, d=128, T=2, per_row=False, name="ML-1M-shaped synthetic (6040x3706, d=128, alpha=0.5)"),
    "c3": dict(U=138493, I=26744, d=128, T=2, per_row=False, name="ML-20M-shaped synthetic (138493x26744, d=128)"),
    "c4": dict(U=4096, I=1250000, d=128, T=2, per_row=False, name="one shard of the 10M-item catalogue (4096 queries x 1.25M items, d=128)"),
    "c5": dict(U=131072, B=4096, I=1250000, d=256, T=4, per_row=True,
               name="multi-alpha hybrid shard (3 content + 1 collaborative, per-row alpha, d=256; 131072 user + 1.25M item rows blended, 4096 queries)"),
    "c4full": dict(U=4096, I=10000000, d=128, T=2, per_row=False, name="10M-item catalogue (4096 queries x 10M items, d=128)"),
    "c5full": dict(U=131072, B=4096, I=10000000, d=256, T=4, per_row=True,
                   name="multi-alpha hybrid, whole catalogue (3 content + 1 collaborative, per-row alpha, d=256; 131072 user + 10M item rows blended, 4096 queries)"),
    # the reference's float64 flavour (ContentRecommendation: float64 tables -> float64 `vectors`, hwer/content_recommender.py:85,96): the SAME step in
    # float64 - hwer.pipeline.BlendSearchStepF64: float64 blend, fused search whose distances / scores come from the float64 rows (round 6)
    "c1_f64rows": dict(U=943, I=1682, d=64, T=2, per_row=False, f64rows=True, name="ML-100K-shaped synthetic, float64 tables and rows (943x1682, d=64, alpha=0.5)"),
    "c3_f64rows": dict(U=138493, I=26744, d=128, T=2, per_row=False, f64rows=True, name="ML-20M-shaped synthetic, float64 tables and rows (138493x26744, d=128)"),
}
K = 200
PROFILE_EVERY = 5
GEN_BLOCK = 65536              # rows per generation block: row r of a table depends only on (table seed, r // GEN_BLOCK, r % GEN_BLOCK)
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PF dense BF16 / F16
VALU_F64_PEAK_TFLOPS = 78.6    # MI355X data sheet: FP64 vector (the re-rank's float64 chains run on the VALU)
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E 8 TB/s spec
METRIC = "user-item pairs/sec scored (blend + L2-normalise + exact top-%d retrieval)" % K


# ====================================================================================================================
# host facts
# ====================================================================================================================
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
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
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


def host_description():
    """CPU model, sockets / threads, numpy BLAS build ("host description to report")."""
    out = {}
    try:
        models, phys = set(), set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                models.add(line.split(":", 1)[1].strip())
            elif line.startswith("physical id"):
                phys.add(line.split(":", 1)[1].strip())
        out["cpu_model"] = sorted(models)[0] if models else None
        out["sockets"] = len(phys) or None
    except OSError:
        pass
    out["logical_cpus"] = os.cpu_count()
    try:
        from threadpoolctl import threadpool_info
        out["blas"] = [{"api": x.get("internal_api"), "version": x.get("version"), "threads": x.get("num_threads"),
                        "arch": x.get("architecture")} for x in threadpool_info() if x.get("user_api") == "blas"]
    except Exception:
        pass
    out["numpy"] = np.__version__
    try:
        import sklearn
        out["sklearn"] = sklearn.__version__
    except ImportError:
        pass
    return out


# ====================================================================================================================
# self-launch (N > 1 without a launcher): start the ranks as child processes BEFORE anything touches a GPU
# ====================================================================================================================
def self_launch(args, argv, poll_s=0.2, grace_s=10.0, script=None):
    """Start the N ranks as child processes (fresh interpreters: nothing here has touched a GPU) and WATCH ALL OF THEM: the first rank
    that exits non-zero (a crash in RCCL initialisation, an out-of-memory kill, an assertion) ends the job - the other ranks are
    terminated (they would otherwise sit in a collective until somebody's timeout), the failing rank's stderr tail is printed and the
    launcher exits with that rank's code.  Rank 0's stdout (the ONE JSON line) is passed through."""
    import socket
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = tempfile.TemporaryFile()
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err))

    def tail(f, n=2000):
        f.seek(0, 2)
        size = f.tell()
        f.seek(max(0, size - n))
        return f.read().decode("utf-8", "replace")

    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    if failed is not None:
        r, c = failed
        for p in procs:                      # exactly the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + grace_s
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                p.kill()
                p.wait()
        sys.stderr.write("[bench] rank %d of %d exited with code %s; the other ranks were terminated.  Its stderr tail:\n%s\n" % (r, args.gpus, c, tail(errs[r])))
        sys.stderr.flush()
        return c if isinstance(c, int) and c > 0 else 1
    for r, err in enumerate(errs):           # a clean run: pass the ranks' stderr through (rank 0 first), then rank 0's line
        t = tail(err, 20000)
        if t.strip():
            sys.stderr.write(t if r == 0 else "".join("[rank %d] %s\n" % (r, ln) for ln in t.splitlines()))
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    return 0


# ====================================================================================================================
# synthetic inputs: unit rows, row r of a table depends only on (seed, r): any shard of the catalogue can be generated
# ====================================================================================================================
def gen_unit_rows(torch, seed, lo, hi, d, dev):
    out = torch.empty((hi - lo, d), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    for blk in range(lo // GEN_BLOCK, (hi + GEN_BLOCK - 1) // GEN_BLOCK):
        gen.manual_seed(seed * 1000003 + blk)
        x = torch.randn((GEN_BLOCK, d), generator=gen, device=dev, dtype=torch.float32)
        a, b = max(lo, blk * GEN_BLOCK), min(hi, (blk + 1) * GEN_BLOCK)
        x = x[a - blk * GEN_BLOCK:b - blk * GEN_BLOCK]
        out[a - lo:b - lo] = x / x.norm(dim=1, keepdim=True)
    return out


def make_workload_class():
    from hwer.pipeline import BlendSearchStep

    class Workload(BlendSearchStep):
        """One configuration on one GPU: the synthetic resident tables + the LIBRARY's step over them (hwer.pipeline.BlendSearchStep:
        the timed step is exactly what `ContentRecommendation.blend_search_step(...).run()` launches).  `lo:hi` = the item rows this
        GPU holds (the whole catalogue at N = 1); ids reported by the search are global: users 0..U-1, then the items."""

        def __init__(self, torch, lib, cfg, dev, lo=None, hi=None, f32_screen=False, f16_dense=False, want_dist=False, local_order=False,
                     n_sets=1, k=None):
            self.torch, self.cfg, self.dev = torch, cfg, dev
            U, I_total, d, T = cfg["U"], cfg["I"], cfg["d"], cfg["T"]
            lo = 0 if lo is None else lo
            hi = I_total if hi is None else hi
            I = hi - lo
            self.T, self.lo, self.hi = T, lo, hi
            B = int(cfg.get("B", U))     # queries per step = the first B users (default: every user)
            N = U + I
            tables = []
            for t in range(T):
                tab = torch.empty((N, d), dtype=torch.float32, device=dev)
                tab[:U] = gen_unit_rows(torch, 3000 + 20 * t, 0, U, d, dev)          # users: identical on every rank
                tab[U:] = gen_unit_rows(torch, 3001 + 20 * t, lo, hi, d, dev)        # items: rows lo..hi of the catalogue
                tables.append(tab)
            if cfg["per_row"]:
                # Dirichlet(1, .., 1) = normalised unit exponentials; 20 % cold-start rows (alpha_collab = 0, renormalised).
                # Row-wise deterministic like the tables: generated per global row block.
                al = torch.empty((N, T), dtype=torch.float32, device=dev)
                al[:U] = self._alpha_rows(5005, 0, U)
                al[U:] = self._alpha_rows(5006, lo, hi)
                alpha = al.contiguous()
            else:
                alpha = torch.full((T,), 1.0 / T, dtype=torch.float32, device=dev)
            env = os.environ.get
            super().__init__(tables, alpha, search_row0=U, n_search=I, other_row0=0, n_other=U,
                             anchors=torch.arange(0, B, dtype=torch.int32, device=dev), k=K if k is None else k, id_base=U + lo, arith=0,   # float64 score flavour
                             want_dist=want_dist, sort_by_score=not local_order, f16_screen=False if f32_screen else None, f16_dense=f16_dense,
                             n_sets=n_sets, extra_options=int(env("HWER_BENCH_OPTS", "0")),     # A/B runs: e.g. 16 = HWER_KNN_NO_SEED
                             fused_blend=True if env("HWER_BENCH_FORCE_FUSED_BLEND") else (False if env("HWER_BENCH_NO_FUSED_BLEND") else None),
                             fused_prologue=False if env("HWER_BENCH_NO_FUSED_PROLOGUE") else None, tail_stream=bool(env("HWER_BENCH_TAIL_STREAM")) and env("HWER_BENCH_TAIL_STREAM") != "0",
                             tail_under_select=bool(env("HWER_BENCH_TAIL_UNDER_SELECT")),
                             pipeline=bool(env("HWER_BENCH_PIPELINE")))

        def _alpha_rows(self, seed, lo, hi):
            torch = self.torch
            T = self.T
            out = torch.empty((hi - lo, T), dtype=torch.float32, device=self.dev)
            gen = torch.Generator(device=self.dev)
            for blk in range(lo // GEN_BLOCK, (hi + GEN_BLOCK - 1) // GEN_BLOCK):
                gen.manual_seed(seed * 1000003 + blk)
                e = -torch.log(torch.rand((GEN_BLOCK, T), generator=gen, device=self.dev, dtype=torch.float64).clamp_min(1e-300))
                cold = torch.rand((GEN_BLOCK,), generator=gen, device=self.dev) < 0.2
                e[cold, T - 1] = 0.0
                e = (e / e.sum(1, keepdim=True)).to(torch.float32)
                a, b = max(lo, blk * GEN_BLOCK), min(hi, (blk + 1) * GEN_BLOCK)
                out[a - lo:b - lo] = e[a - blk * GEN_BLOCK:b - blk * GEN_BLOCK]
            return out

        # ---- algorithmic work -----------------------------------------------------------------------------------
        def stage_bytes(self):
            d, T, N, U, I = self.d, self.T, self.N, self.U, self.I
            per_row = self.cfg["per_row"]
            return {
                "blend_normalize_audit": N * (4.0 * d * (T + 1) + (4.0 * T if per_row else 0.0)),       # T rows in, one out (+ per-row alpha)
                "pack_rows": I * (4.0 * d + 4.0 * self.ldx + 4.0 + 8.0) + (I * 2.0 * self.ldx if self.f16_screen else 0.0),
                "query_mix": self.B * (4.0 * d + 8.0 * d + 4.0 * self.ldx + 4.0),
            }

    return Workload


_WORKLOAD = None


def Workload(*a, **kw):
    global _WORKLOAD
    if _WORKLOAD is None:
        _WORKLOAD = make_workload_class()
    return _WORKLOAD(*a, **kw)


def cuda_time(torch, fn, reps, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def cuda_time_median(torch, fn, reps, warm=2):
    """median of per-repetition HIP-event intervals (latencies: one hiccup must not move the number)"""
    for _ in range(warm):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]))


def read_profile(lib):
    tot, cnt = ctypes.c_double(), ctypes.c_int64()
    lib.hwer_profile_read(0, ctypes.byref(tot), ctypes.byref(cnt))
    screen_ms, n0 = tot.value / max(cnt.value, 1), cnt.value
    lib.hwer_profile_read(1, ctypes.byref(tot), ctypes.byref(cnt))
    return screen_ms, tot.value / max(cnt.value, 1), n0


def library_digest():
    """sha256 over the library's sources + flags (csrc/build.py::_digest): what `lib/libhwer_hip.so.sha256` holds for the library that ran"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd", "csrc"))
        import build as hip_build
        return hip_build._digest()
    except Exception:
        return None
    finally:
        sys.path.pop(0)


def load_traffic(config):
    """HBM-side bytes of the screen stage per launch from the committed rocprofv3 PMC passes of THIS command
    ((2 * FETCH_SIZE + WRITE_SIZE) * 1024; gfx950 counts wide reads at half their bytes - MI355X_MICROARCH.md §HBM).  The
    counters cannot be read from inside the run: the value is labelled with the file it comes from, and it is only reported when
    that file carries the digest of the library that is running (`library_digest`; tools/pmc_summary.py writes it) - counters of
    another build of the kernels are not this run's traffic: then `traffic` is null and `traffic_source` says why."""
    dig = library_digest()
    for fn in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as fh:
                doc = json.load(fh)
            ent = doc.get(config)
            if ent and ent.get("knn_screen_bytes_per_launch"):
                if dig is not None and doc.get("library_digest") == dig:
                    return ent["knn_screen_bytes_per_launch"], "profiles/" + fn
                return None, "profiles/%s was collected on another build of the library (digest %s != %s)" % (fn, str(doc.get("library_digest"))[:12], str(dig)[:12])
        except (OSError, ValueError):
            pass
    return None, None


# ====================================================================================================================
# one configuration on one GPU
# ====================================================================================================================
def run_config(torch, lib, name, dev, steps, warmup, args, cpu_budget_s=0.0, api=False, latencies=(), stage_reps=20, f32_beside=False):
    from hwer import native
    cfg = CONFIGS[name]
    w = Workload(torch, lib, cfg, dev, f32_screen=args.f32_screen, f16_dense=args.f16_dense)
    U, I, d = w.B, w.I, w.d      # U = queries per step from here on (w.U user rows are blended; c5: 131 072 rows, 4 096 queries)
    # Small catalogues: a step is ~8 launches of a few microseconds each, so issuing them from Python costs as much as running
    # them.  The step is captured ONCE into a hipGraph (every buffer is preallocated, nothing in the step synchronises or
    # allocates) and the timed region replays it.
    graph = None
    if I <= 8192 and not args.no_graph:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    w.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                w.step()
            graph = g
        except Exception as e:   # capture refused: measure the eager step and say so
            graph = None
            torch.cuda.synchronize()
            sys.stderr.write("[bench] hipGraph capture failed (%s): timing the eager step\n" % (str(e).splitlines()[0] if str(e) else type(e).__name__))
    run_step = graph.replay if graph is not None else w.step
    profile_in_region = graph is None and not w.tail and not w.pipeline      # HIP events around every PROFILE_EVERY-th search (a pair costs ~10 us of stream time)
    # The timed region: EXACTLY `steps` steps between two synchronisations, wall clock (the contract's ms_per_step / value).  Eager
    # steps (>= ~0.2 ms each) also get one HIP event per step boundary inside the region - the per-step distribution
    # (median / min / max) costs nothing there; graph-replayed steps (c1 / c2: ~0.05-0.2 ms) get theirs from a second pass.
    in_loop = graph is None
    import gc

    def timed_region(n_steps=steps, n_warm=warmup, events=True):
        # Round 5: the W warm-up steps run HERE, directly in front of the region - everything the host has to do for the region (event objects, a
        # full garbage collection: 10 - 30 ms with torch imported) comes first.  With those between the warm-up and the region the device sat
        # idle for milliseconds, its clocks came down, and a region of 20 x 0.042 ms ran 7 - 9 % slower than steady state (tools/
        # probe_region_overhead.py, profiles/r5_probe_region_overhead.txt: K = 20 behind 5 warm replays 0.0438 ms/step without a gap, 0.0446 / 0.0459
        # behind 10 / 100 ms of idling; steady state 0.0422).  warm-up, synchronize, t0 - as the contract words it.
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)] if events else None
        gc.collect()
        gc.disable()        # (as `timeit` does: a collection inside K x 0.05 ms would be most of the region)
        try:
            for _ in range(n_warm):
                run_step()
            if profile_in_region:
                lib.hwer_profile_enable(PROFILE_EVERY)      # (a flag of the library: the timed steps' searches, not the warm-up's)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if in_loop and events:
                evs[0].record()
            for i in range(n_steps):
                run_step()
                if in_loop and events:
                    evs[i + 1].record()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        finally:
            gc.enable()
        if not events:
            return el, None
        if not in_loop:
            evs[0].record()
            for i in range(n_steps):
                run_step()
                evs[i + 1].record()
            torch.cuda.synchronize()
        return el, np.asarray([evs[i].elapsed_time(evs[i + 1]) for i in range(n_steps)])

    # THE timed region is the FIRST one, whatever it shows: `value` and `ms_per_step` always come from it (round 4: rounds 2-3 measured a
    # region again when its wall-clock mean was far above its own per-step median and reported the repeat - a best-of-3 with a rejection
    # filter).  A region whose mean exceeds 1.3x (graph replays) / 1.15x (eager steps) the median of its per-step event intervals - one
    # host or driver stall inside K x 0.05 ms - is still measured once more, but only as the SECONDARY field `timed_region.remeasured_ms_per_step`.
    elapsed, series = timed_region()
    limit = 1.3 if graph is not None else 1.15
    remeasured = None
    if elapsed / steps * 1e3 > limit * float(np.median(series)):
        el2, series2 = timed_region()
        remeasured = {"ms_per_step": el2 / steps * 1e3, "ms_median": float(np.median(series2))}
    lib.hwer_profile_enable(0)
    # graph-replayed configurations (c1, c2): the same step over a region long enough for the device's clocks to settle - a SECONDARY figure
    # (`timed_region.steady_ms_per_step`); with the driver's --steps 20 the contract's region is 1 - 3 ms long and still on the ramp
    steady = None
    if graph is not None and steps < 500:
        el_s, _ = timed_region(n_steps=500, n_warm=max(warmup, 100), events=False)
        steady = el_s / 500 * 1e3
    in_step = None
    if graph is None and not w.tail and not w.pipeline:
        # Eager steps: the two parts of a step measured IN the step (an event between the prologue and the search of real steps, right behind
        # the timed region: same clocks, same cache state).  A stage repeated on its own runs in another cache state - the search of c4 reads
        # the table the prologue has just written (the last ~256 MB of it still in the Infinity Cache) and is ~10 % faster there than back
        # to back with itself.
        n_split = max(5, min(steps, 10))
        marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_split)]
        for e0, e1, e2 in marks:
            e0.record()
            w.step(mark=e1)
            e2.record()
        torch.cuda.synchronize()
        in_step = (float(np.median([e0.elapsed_time(e1) for e0, e1, _ in marks])), float(np.median([e1.elapsed_time(e2) for _, e1, e2 in marks])), n_split)
    if graph is not None or w.tail or w.pipeline:    # kernel durations: an eager pass with event brackets after the timed region
        lib.hwer_profile_enable(1)     # (searches one after the other: overlapped searches would stretch each other's brackets)
        w.overlap_tail = False
        for _ in range(10 if graph is not None else 3):
            w.step()
        torch.cuda.synchronize()
        w.overlap_tail = True
        lib.hwer_profile_enable(0)
    screen_ms, select_ms, _ = read_profile(lib)
    o = w.sets[0]
    n_exhaustive = w.checks(o)
    ms_per_step = elapsed / steps * 1e3
    value = float(w.B) * float(I) / (elapsed / steps)
    idx_h, score_h = o["idx"].cpu().numpy(), o["score"].cpu().numpy()

    # ---- parity of the timed step's result: sampled users against the oracle's C brute force on the same float32 rows -------
    from oracle import c_oracle
    from oracle import hwer_oracle as orc
    cores, cores_note = effective_cpus()
    Vh = w.V[:, :d].cpu().numpy()
    n_par = 16 if I <= 2000000 else 8
    sel = np.unique(np.linspace(0, U - 1, n_par).astype(np.int64))
    V64u = Vh[:U].astype(np.float64)
    Qh = np.stack([orc.query_vector(V64u, int(s)) for s in sel])
    oi, osc, _ = c_oracle.find_closest_f64(Vh[w.U:], Qh, V64u[sel], K, threads=cores)
    if not os.environ.get("HWER_BENCH_DIAG_NO_ORACLE"):   # (timing experiments with a DIAGNOSTIC library whose results are wrong on purpose; never in a reported line)
        assert np.array_equal(idx_h[sel], oi + w.U), "GPU indices differ from the oracle (%s)" % name
        assert np.array_equal(score_h[sel], osc), "GPU scores differ from the oracle (%s)" % name

    # ---- per-stage times (outside the timed region), per-kernel rooflines ----------------------------------------------------
    searches_per_step = (U + w.CH - 1) // w.CH
    stages = {"blend_normalize_audit": cuda_time(torch, w.stage_blend, stage_reps), "pack_rows": cuda_time(torch, w.stage_pack, stage_reps),
              "query_mix": cuda_time(torch, w.stage_queries, stage_reps)}
    stages["knn_search"] = cuda_time(torch, lambda: w.stage_search(o), max(3, stage_reps // 2), warm=1)
    sbytes = w.stage_bytes()
    if w.fused_prologue:   # what the step actually runs instead of pack_rows + query_mix (+ the search's repacking launch)
        stages["prepare_search"] = cuda_time(torch, w.stage_prepare, stage_reps)
        sbytes["prepare_search"] = sbytes["pack_rows"] + sbytes["query_mix"] + (2.0 if w.bq_frag < 0 else 4.0) * abs(w.bq_frag) * w.ldx
    if w.fused_blend:      # ... and the blend with them: the step's whole prologue
        stages["blend_prepare_search"] = cuda_time(torch, w.stage_blend_prepare, stage_reps)
        sbytes["blend_prepare_search"] = (sbytes["blend_normalize_audit"] + sbytes["pack_rows"] + sbytes["query_mix"] + (2.0 if w.bq_frag < 0 else 4.0) * abs(w.bq_frag) * w.ldx
                                          - 4.0 * d * (I + w.B)    # the table pass and the queries read registers, not V
                                          - (4.0 * w.ldx * I if w.x_in_v else 0.0))   # ... and the table IS a slice of V: one store per row
    f16 = w.f16_screen
    flops = 2.0 * U * I * d
    mfma_peak = MFMA_F16_PEAK_TFLOPS if f16 else MFMA_F32_PEAK_TFLOPS
    screen_step_ms, select_step_ms = screen_ms * searches_per_step, select_ms * searches_per_step
    kernels = []
    for sname in (("blend_prepare_search",) if w.fused_blend else (("blend_normalize_audit", "prepare_search") if w.fused_prologue else ("blend_normalize_audit", "pack_rows", "query_mix"))):
        ms = stages[sname]
        kernels.append({"kernel": sname, "ms_per_step": ms, "share": ms / ms_per_step, "bound": "hbm", "achieved": sbytes[sname] / ms / 1e6,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sbytes[sname] / ms / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes": sbytes[sname]})
    traffic, traffic_src = load_traffic(name)
    kernels.append({"kernel": "knn_screen (%s)" % ("v_mfma_f32_32x32x16_f16 on the float16 copy + streaming top-K' filter" if f16 else "v_mfma_f32_32x32x2_f32 + fused top-K' filter"),
                    "ms_per_step": screen_step_ms, "share": screen_step_ms / ms_per_step, "bound": "mfma", "achieved": flops / (screen_step_ms * 1e-3) / 1e12 if screen_step_ms > 0 else 0.0,
                    "peak": mfma_peak, "unit": "TFLOP/s", "frac": (flops / (screen_step_ms * 1e-3) / 1e12 / mfma_peak) if screen_step_ms > 0 else 0.0,
                    "vs_fp32_mfma_peak": (flops / (screen_step_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if screen_step_ms > 0 else 0.0,
                    "launches_per_step": searches_per_step, "avg_launch_ms": screen_ms, "algorithmic_flops": flops,
                    "traffic": traffic, "traffic_source": traffic_src})
    # select / re-rank: every query gathers its ~k + 16 candidate rows (4 * ldx bytes each, float32: the exact re-rank reads the
    # float32 rows whatever the screen read) from a table that lives in L2, in the Infinity Cache or in HBM by its size; the
    # gathered-row rates are the measured ones of /opt/skills/guides/MI355X_MICROARCH.md "Indexed rows: gather" (L2-resident
    # rows 17.8 TB/s chip-wide, Infinity Cache 8.6 TB/s, HBM random rows 5.6 TB/s).  profiles/r2_rocprof_pmc_c3.txt: the
    # select kernel of c3 fetches 2 * 5.5 GB per launch = the candidates' rows.
    table_bytes = float(w.I_pad) * w.ldx * 4
    level, gather_peak = ("l2", 17800.0) if table_bytes <= 4.0e6 else (("infinity_cache", 8600.0) if table_bytes <= 2.56e8 else ("hbm_random_rows", 5600.0))
    gather_bytes = float(U) * (K + 16) * 4.0 * w.ldx
    kernels.append({"kernel": "knn_select_rerank (candidate-row gather + float64 re-rank + certificate + sort)", "ms_per_step": select_step_ms,
                    "share": select_step_ms / ms_per_step, "bound": "gather:" + level,
                    "achieved": gather_bytes / (select_step_ms * 1e-3) / 1e9 if select_step_ms > 0 else 0.0, "peak": gather_peak, "unit": "GB/s",
                    "frac": (gather_bytes / (select_step_ms * 1e-3) / 1e9 / gather_peak) if select_step_ms > 0 else 0.0, "algorithmic_bytes": gather_bytes,
                    "note": "one wave per query; beside the row gather the kernel is bound by its instruction count (rank search, register sorts: profiles/r2_select_stamps.txt)"})
    dominant = max(kernels, key=lambda kk: kk["ms_per_step"])
    stage_tf = flops / ((screen_step_ms + select_step_ms) * 1e-3) / 1e12 if screen_step_ms + select_step_ms > 0 else 0.0
    res = {
        "workload": cfg["name"], "users": U, "items": I, "d": d, "k": K, "tables": w.T, "alpha": "per-row" if cfg["per_row"] else "scalar",
        "f16_screen": f16, "dtype": "f16" if f16 else "f32", "steps": steps, "warmup": warmup, "hip_graph": graph is not None,
        "value": value, "unit": "pairs/s", "ms_per_step": ms_per_step,
        "ms_median": float(np.median(series)), "ms_min": float(series.min()), "ms_max": float(series.max()), "ms_mean_events": float(series.mean()),
        "timed_region": {"attempts": 1 if remeasured is None else 2, "reported": "first", "remeasured_ms_per_step": None if remeasured is None else round(remeasured["ms_per_step"], 5),
                         "steady_ms_per_step": None if steady is None else round(steady, 5),
                         "rule": "value / ms_per_step = the FIRST region of exactly K steps (W warm-up steps, synchronize, clock), always; a region whose wall-clock mean exceeds 1.3x (graph replays) / 1.15x (eager) the median of "
                                 "its per-step event intervals is measured once more and that number is reported here only"},
        "ms_series": [round(float(x), 4) for x in series[:256]], "ms_series_note": "HIP-event interval of every step" + ("" if in_loop else " (second pass: the timed steps are graph replays)"),
        "user_rows": w.U,
        "roofline": {k_: dominant[k_] for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac")},
        "stage_frac": stage_tf / MFMA_F32_PEAK_TFLOPS,
        "stage_frac_note": "2*B*N*d / (screen + select time) / 157.3 TF: the cosine_topk unit of SURVEY.md 8d against the float32-MFMA peak",
        "kernels": [kk for kk in kernels if kk["share"] >= 0.10 or kk["kernel"].startswith("knn_")],
        "stages_ms": stages,
        "certified_queries": U - n_exhaustive, "exhaustive_fallback_queries": n_exhaustive, "parity_checked_queries": int(len(sel)),
    }
    res["roofline"]["traffic"] = traffic
    res["roofline"]["traffic_source"] = traffic_src
    res["roofline"]["share_of_step"] = dominant["share"]
    res["dominant_kernel"] = {"name": dominant["kernel"][:40], "ms": _r(dominant["ms_per_step"], 4), "share": _r(dominant["share"], 3),
                              "bound": dominant["bound"], "frac": _r(dominant["frac"], 3)}
    if not dominant["kernel"].startswith("knn_screen") and not dominant["bound"].startswith("hbm"):
        # Dense path (c1 / c2): the dominant kernel is the select / re-rank stage, which has no "hbm" | "mfma" roofline of its own
        # (L2 row gathers + a serial selection chain).  The line's `roofline` is then SURVEY.md 8d's cosine_topk unit - 2 B N d flops
        # over the screen + select time against the float32-MFMA peak (= stage_frac); the kernel's own pricing stays in
        # `dominant_kernel` and `kernels`.
        res["roofline"] = {"kernel": "cosine_topk = knn_screen + knn_select (SURVEY 8d unit)", "bound": "mfma", "achieved": stage_tf, "peak": MFMA_F32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": stage_tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                           "share_of_step": (screen_step_ms + select_step_ms) / ms_per_step}
    prologue = "blend_prepare_search" if w.fused_blend else None
    res["stage_sum_ms"] = (stages[prologue] if prologue else (stages["blend_normalize_audit"] + (stages["prepare_search"] if w.fused_prologue else stages["pack_rows"] + stages["query_mix"]))) + stages["knn_search"]
    res["stage_sum_note"] = "prologue + search, each timed on its own (back-to-back repetitions of one stage)"
    if in_step is not None:
        pro, sea, n_split = in_step
        res["stages_in_step_ms"] = {"prologue": pro, "search": sea}
        res["stage_sum_separate_ms"] = res["stage_sum_ms"]
        res["stage_sum_ms"] = pro + sea
        res["stage_sum_note"] = "prologue + search inside real steps (median of %d steps, one HIP event between the two); stage_sum_separate_ms: each stage repeated on its own" % n_split

    # ---- the same step with the float32 screen, beside the float16 number (streaming configurations) -------------------------
    if f32_beside and f16:
        def step32():
            w.stage_blend(); w.stage_pack(); w.stage_queries(); w.stage_search(o, opts=0)   # noqa: E702
        ms32 = cuda_time(torch, step32, max(3, min(steps, 5)), warm=1)
        lib.hwer_profile_enable(1)     # kernel durations from a second pass, searches one after the other (see above)
        w.overlap_tail = False
        cuda_time(torch, step32, 3, warm=0)
        w.overlap_tail = True
        lib.hwer_profile_enable(0)
        s32, sel32, _ = read_profile(lib)
        res["f32_screen"] = {"ms_per_step": ms32, "value": float(U) * I / ms32 * 1e3, "knn_screen_ms": s32 * searches_per_step,
                             "knn_screen_frac_of_fp32_mfma_peak": flops / (s32 * searches_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS if s32 > 0 else None,
                             "knn_select_rerank_ms": sel32 * searches_per_step}
        w.stage_search(o)   # leave the default path's result in place

    # ---- query-batch latencies against this catalogue (B in {1, 64, 4096}) ---------------------------------
    for B in latencies:
        if B > U:
            continue
        ob = dict(idx=o["idx"][:B], score=o["score"][:B], dist2=None, flags=o["flags"][:B])
        opts_b = w.opts if B > 1024 else 0       # small batches take the HBM-bound path, which reads the float16 copy when there is one
        ms = cuda_time_median(torch, lambda: w.stage_search(ob, B=B, opts=opts_b), 21 if B <= 64 else 7, warm=2)
        row_bytes = (2.0 if w.X16 is not None else 4.0) * w.ldx
        ent = {"batch": B, "ms": ms, "pairs_per_s": float(B) * I / ms * 1e3}
        if B <= 64:
            ent["hbm_frac_of_6.3TBs"] = (I * row_bytes / (ms * 1e-3)) / 6.3e12
        res.setdefault("query_latency_ms", []).append(ent)

    # ---- pair scoring (P / t): random (user, item) pairs of this table -------------------------------------
    if not args.sweep_only:
        n_pairs = 1000209 if I < 100000 else 16000000
        gp = torch.Generator(device=dev).manual_seed(4001)
        p_src = torch.randint(0, w.U, (n_pairs,), generator=gp, device=dev, dtype=torch.int32)
        p_dst = torch.randint(w.U, w.N, (n_pairs,), generator=gp, device=dev, dtype=torch.int32)
        p_out = torch.empty(n_pairs, dtype=torch.float64, device=dev)
        P = native.ptr
        ps_ms = cuda_time(torch, lambda: native.check(lib.hwer_pair_score_f64(P(w.V), w.N, d, w.ldv, P(p_src), P(p_dst), n_pairs, P(p_out), w.S()), "pair_score"), 20, warm=3)
        ps_bytes = n_pairs * (8.0 * d + 8 + 8)     # two rows + two int32 ids + one float64 result per pair
        table_mb = w.N * d * 4 / 1e6
        res["pair_score"] = {"pairs": n_pairs, "value": n_pairs / ps_ms * 1e3, "unit": "pairs/s", "ms": ps_ms, "dtype": "f64", "table_MB": table_mb,
                             "roofline": {"bound": "hbm", "achieved": ps_bytes / ps_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": ps_bytes / ps_ms / 1e6 / HBM_PEAK_GBS,
                                          "note": "table resident in L2 / Infinity Cache: not an HBM measurement" if table_mb < 256 else "gathered rows from HBM"}}

    # ---- CPU baseline on the host cores: the oracle's C restatement (kind "port"), the vectorised float32 formulation, the
    # reference-shaped per-anchor loop (scikit-learn KD-tree + Python, one thread) ---------------------------------------------
    if cpu_budget_s > 0:
        res["cpu_baseline"] = cpu_baseline(w, Vh, idx_h, cpu_budget_s, cores, cores_note)
        res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
        if "reference_shaped" in res["cpu_baseline"]:
            res["speedup_vs_reference_shaped_loop"] = value / res["cpu_baseline"]["reference_shaped"]["value"]

    # ---- API-level latency: the same sweep through the drop-in Python classes, results on the host -----------------------------
    if api:
        res["api_latency_ms"] = api_latency(torch, Vh, U, I, d, arrays_only=(api == "arrays"))
    del w
    torch.cuda.empty_cache()
    return res


def gen_unit_rows_f64(torch, seed, n, d, dev):
    """genuine float64 unit rows (not float32-representable): N(0, 1) in float64, row-normalised in float64"""
    out = torch.empty((n, d), dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev)
    for lo in range(0, n, GEN_BLOCK):
        gen.manual_seed(seed * 1000003 + lo // GEN_BLOCK)
        x = torch.randn((min(GEN_BLOCK, n - lo), d), generator=gen, device=dev, dtype=torch.float64)
        out[lo:lo + x.shape[0]] = x / x.norm(dim=1, keepdim=True)
    return out


def run_config_f64rows(torch, lib, name, dev, steps, warmup, args):
    """One configuration in the reference's float64 flavour (hwer.pipeline.BlendSearchStepF64): float64 tables -> hwer_blend_normalize_f64 ->
    float64 `vectors` -> audit + float32 / float16 rounding of the item rows (screen only) -> queries from the float64 rows -> fused search on the
    float64 rows.  Same timed-region contract as run_config; parity: the blended rows against the numpy expression (array_equal) and sampled queries
    against the numpy oracle ON THE FLOAT64 ROWS (ids, scores: bit for bit)."""
    import gc
    from hwer.pipeline import BlendSearchStepF64
    from oracle import hwer_oracle as orc
    cfg = CONFIGS[name]
    U, I, d, T = cfg["U"], cfg["I"], cfg["d"], cfg["T"]
    N = U + I
    tables = [gen_unit_rows_f64(torch, 3100 + 20 * t, N, d, dev) for t in range(T)]
    alpha = torch.full((T,), 1.0 / T, dtype=torch.float64, device=dev)
    w = BlendSearchStepF64(tables, alpha, search_row0=U, n_search=I, anchors=torch.arange(0, U, dtype=torch.int32, device=dev), k=K)
    graph = None
    if I <= 8192 and not args.no_graph:
        w.capture()
        graph = w._graph
    run_step = graph.replay if graph is not None else w.step

    def region(n_steps, n_warm):
        gc.collect()
        gc.disable()
        try:
            for _ in range(n_warm):
                run_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                run_step()
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        finally:
            gc.enable()
    elapsed = region(steps, warmup)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    evs[0].record()
    for i in range(steps):
        run_step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    series = np.asarray([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    steady = region(500, max(warmup, 100)) / 500 * 1e3 if (graph is not None and steps < 500) else None
    # the two parts of a step in real (eager) steps, and the search's kernels bracketed by HIP events on the launch stream
    n_split = max(5, min(steps, 10))
    marks = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_split)]
    lib.hwer_profile_enable(1)
    for e0, e1, e2 in marks:
        e0.record()
        w.step(mark=e1)
        e2.record()
    torch.cuda.synchronize()
    lib.hwer_profile_enable(0)
    screen_ms, select_ms, _ = read_profile(lib)
    searches_per_step = (U + w.CH - 1) // w.CH
    pro, sea = float(np.median([a.elapsed_time(b) for a, b, _ in marks])), float(np.median([b.elapsed_time(c) for _, b, c in marks]))
    stages = {"blend_normalize_f64": cuda_time(torch, w.stage_blend, 10), "audit+pack_rows_f64rows": cuda_time(torch, w.stage_pack, 10),
              "query_mix_f64rows": cuda_time(torch, w.stage_queries, 10)}
    run_step()
    torch.cuda.synchronize()
    o = w.sets[0]
    n_exhaustive = w.checks(o)
    ms_per_step = elapsed / steps * 1e3
    value = float(U) * float(I) / (elapsed / steps)
    # ---- parity: blend stage == numpy on sampled rows; sampled queries == the numpy oracle on the float64 rows ------------------------------
    idx_h, score_h = o["idx"].cpu().numpy(), o["score"].cpu().numpy()
    Vh = w.V64[:, :d].cpu().numpy()
    rows_chk = np.unique(np.linspace(0, N - 1, 64).astype(np.int64))
    acc = np.zeros((len(rows_chk), d))
    for t in range(T):
        acc = acc + (1.0 / T) * tables[t][torch.from_numpy(rows_chk).to(dev)].cpu().numpy()
    ref_rows = acc / np.expand_dims(np.linalg.norm(acc, axis=1), 1)
    assert np.array_equal(Vh[rows_chk], ref_rows), "float64 blend differs from the numpy expression (%s)" % name
    sel = np.unique(np.linspace(0, U - 1, 16).astype(np.int64))
    for u in sel:
        o_idx, o_sc, _ = orc.find_closest(Vh, np.arange(U, N), int(u), k=K)
        assert np.array_equal(idx_h[u], o_idx), "GPU indices differ from the oracle on float64 rows (%s, query %d)" % (name, u)
        assert np.array_equal(score_h[u], o_sc), "GPU scores differ from the oracle on float64 rows (%s, query %d)" % (name, u)
    flops = 2.0 * U * I * d
    sc_ms, se_ms = screen_ms * searches_per_step, select_ms * searches_per_step
    stage_tf = flops / ((sc_ms + se_ms) * 1e-3) / 1e12 if sc_ms + se_ms > 0 else 0.0
    gather_bytes = float(U) * (K + 16) * 8.0 * w.ld64
    table_bytes = float(N) * w.ld64 * 8
    level, gather_peak = ("l2", 17800.0) if table_bytes <= 4.0e6 else (("infinity_cache", 8600.0) if table_bytes <= 2.56e8 else ("hbm_random_rows", 5600.0))
    res = {"workload": cfg["name"], "users": U, "items": I, "d": d, "k": K, "tables": T, "alpha": "scalar", "f16_screen": w.f16_screen, "dtype": "f64",
           "dtype_note": "float64 tables, float64 blend, float64 rows: distances and scores from the float64 rows; the MFMA screen runs on their %s rounding (superset only)" % ("float16" if w.f16_screen else "float32"),
           "steps": steps, "warmup": warmup, "hip_graph": graph is not None, "value": value, "unit": "pairs/s", "ms_per_step": ms_per_step,
           "ms_median": float(np.median(series)), "ms_min": float(series.min()), "ms_max": float(series.max()), "ms_mean_events": float(series.mean()),
           "timed_region": {"attempts": 1, "reported": "first", "remeasured_ms_per_step": None, "steady_ms_per_step": None if steady is None else round(steady, 5)},
           "stages_in_step_ms": {"prologue": pro, "search": sea}, "stage_sum_ms": pro + sea, "stages_ms": stages,
           "knn_screen_ms": sc_ms, "knn_select_f64rows_ms": se_ms,
           "roofline": {"kernel": "cosine_topk = knn_screen + knn_select_f64rows (SURVEY 8d unit)", "bound": "mfma", "achieved": stage_tf, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": stage_tf / MFMA_F32_PEAK_TFLOPS, "traffic": None, "traffic_source": "not collected for this configuration"},
           "select_gather": {"bound": "gather:" + level, "achieved": gather_bytes / (se_ms * 1e-3) / 1e9 if se_ms > 0 else 0.0, "peak": gather_peak, "unit": "GB/s",
                             "algorithmic_bytes": gather_bytes, "note": "(k + 16) float64 rows of 8 ld64 bytes per query"},
           "certified_queries": U - n_exhaustive, "exhaustive_fallback_queries": n_exhaustive, "parity_checked_queries": int(len(sel)),
           "parity_note": "ids and scores == the numpy oracle on the float64 rows; blended rows == numpy (array_equal, %d rows)" % len(rows_chk)}
    blend_bytes, pack_bytes = N * 8.0 * d * (T + 1), N * 8.0 * d + I * (4.0 * w.ldx + (2.0 * w.ldx if w.f16_screen else 0.0) + 4.0)
    qmix_bytes = U * (8.0 * d + 8.0 * d + 4.0 * w.ldx + 4.0)
    res["kernels"] = [
        {"kernel": "blend_normalize_f64", "ms_per_step": stages["blend_normalize_f64"], "share": stages["blend_normalize_f64"] / ms_per_step, "bound": "hbm",
         "achieved": blend_bytes / stages["blend_normalize_f64"] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": blend_bytes / stages["blend_normalize_f64"] / 1e6 / HBM_PEAK_GBS},
        {"kernel": "audit + pack_rows_f64rows", "ms_per_step": stages["audit+pack_rows_f64rows"], "share": stages["audit+pack_rows_f64rows"] / ms_per_step, "bound": "hbm",
         "achieved": pack_bytes / stages["audit+pack_rows_f64rows"] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": pack_bytes / stages["audit+pack_rows_f64rows"] / 1e6 / HBM_PEAK_GBS},
        {"kernel": "query_mix_f64rows", "ms_per_step": stages["query_mix_f64rows"], "share": stages["query_mix_f64rows"] / ms_per_step, "bound": "hbm",
         "achieved": qmix_bytes / stages["query_mix_f64rows"] / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": qmix_bytes / stages["query_mix_f64rows"] / 1e6 / HBM_PEAK_GBS},
        {"kernel": "knn_screen (on the %s rounding of the float64 rows)" % ("float16" if w.f16_screen else "float32"), "ms_per_step": sc_ms, "share": sc_ms / ms_per_step, "bound": "mfma",
         "achieved": flops / (sc_ms * 1e-3) / 1e12 if sc_ms > 0 else 0.0, "peak": MFMA_F16_PEAK_TFLOPS if w.f16_screen else MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": (flops / (sc_ms * 1e-3) / 1e12 / (MFMA_F16_PEAK_TFLOPS if w.f16_screen else MFMA_F32_PEAK_TFLOPS)) if sc_ms > 0 else 0.0},
        {"kernel": "knn_select_f64rows", "ms_per_step": se_ms, "share": se_ms / ms_per_step, "bound": res["select_gather"]["bound"], "achieved": res["select_gather"]["achieved"],
         "peak": gather_peak, "unit": "GB/s", "frac": res["select_gather"]["achieved"] / gather_peak},
    ]
    if se_ms > sc_ms and I > 8192:
        # streaming shapes: the dominant kernel is the fused float64-row select, bound by the gather of (k + boundary) float64 rows per query from the
        # level the float64 table lives in (MI355X_MICROARCH.md "Indexed rows: gather") - priced like knn_select_rerank of the float32 flavour
        g = res["select_gather"]
        res["roofline"] = {"kernel": "knn_select_f64rows (gather of the candidates' float64 rows + float64 re-rank + certificate + sort)", "bound": g["bound"],
                           "achieved": g["achieved"], "peak": g["peak"], "unit": "GB/s", "frac": g["achieved"] / g["peak"], "traffic": None,
                           "traffic_source": "not collected for this configuration", "share_of_step": se_ms / ms_per_step}
    del w, tables
    torch.cuda.empty_cache()
    return res


def cpu_baseline(w, Vh, idx_h, budget_s, cores, cores_note):
    from oracle import c_oracle
    from oracle import hwer_oracle as orc
    U, I, d = w.U, w.I, w.d
    V64 = Vh[:U].astype(np.float64)
    n_s = int(min(U, max(64, 0.7e9 * budget_s * cores / (float(I) * d * 8))))
    sel = np.linspace(0, U - 1, n_s).astype(np.int64)
    Qh = V64[sel] / np.linalg.norm(V64[sel], axis=1, keepdims=True)
    c_oracle.find_closest_f64(Vh[U:U + min(I, 2048)], Qh[:8], V64[sel[:8]], min(K, 64), threads=cores)  # warm up the thread pool
    passes, t_all = 0, 0.0
    while passes < 1 or (t_all < budget_s and passes < 4096):   # ~budget_s seconds of CPU work whatever a pass costs (c1: ~5 ms per pass)
        t1 = time.perf_counter()
        c_oracle.find_closest_f64(Vh[U:], Qh, V64[sel], K, threads=cores)
        t_all += time.perf_counter() - t1
        passes += 1
    t_pass = t_all / passes
    n_1 = max(8, n_s // (4 * cores))
    t1 = time.perf_counter()
    c_oracle.find_closest_f64(Vh[U:], Qh[:n_1], V64[sel[:n_1]], K, threads=1)
    t_one = time.perf_counter() - t1
    out = {"value": n_s * float(I) / t_pass, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "%d of %d users x %d items, %d passes (%.1f s of CPU work), exact float64 brute force + re-score + sort (oracle/hwer_oracle.c, OpenMP)" % (n_s, U, I, passes, t_all),
           "seconds_per_pass": t_pass, "single_thread_value": n_1 * float(I) / t_one, "cores_note": cores_note, "host": host_description()}
    # best-effort VECTORISED CPU path: batched float32 Q @ X^T (BLAS, all cores) + argpartition + sort +
    # exact float64 re-score of the k kept per user.  Not the reference's algorithm (its KD-tree is exact in float64).
    try:
        n_v = int(min(U, 4096, max(64, 2.0e9 / (4.0 * I))))     # key matrix of the sample <= 2 GB
        v_sel = np.linspace(0, U - 1, n_v).astype(np.int64)
        Xh = np.ascontiguousarray(Vh[U:U + I])
        X64 = None
        hn = 0.5 * (Xh.astype(np.float64) ** 2).sum(1).astype(np.float32)

        def vectorised():
            Qv = np.ascontiguousarray(Vh[v_sel])
            keys = Qv @ Xh.T
            keys -= hn[None, :]
            part = np.argpartition(-keys, K - 1, axis=1)[:, :K]
            ks = np.take_along_axis(keys, part, axis=1)
            order = np.argsort(-ks, axis=1, kind="stable")
            top = np.take_along_axis(part, order, axis=1)
            sc = (np.einsum("bkd,bd->bk", Xh[top].astype(np.float64), V64[v_sel]) + 1.0) / 2.0
            return top, sc
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=cores, user_api="blas")
        except Exception:
            limiter = None
        n_runs = 3 if I <= 50000 else 1
        if n_runs > 1:
            vectorised()
        ts = []
        for _ in range(n_runs):
            t1 = time.perf_counter()
            v_top, _ = vectorised()
            ts.append(time.perf_counter() - t1)
        t_v = float(np.median(ts))
        if limiter is not None:
            limiter.restore_original_limits()
        agree = float(np.mean([len(np.intersect1d(v_top[i], idx_h[v_sel[i]] - U)) / float(K) for i in range(0, n_v, max(1, n_v // 256))]))
        out["vectorised_f32"] = {"value": n_v * float(I) / t_v, "unit": "pairs/s", "cores": cores,
                                 "sample": "%d users x %d items, median of %d" % (n_v, I, n_runs), "seconds": t_v,
                                 "note": "float32 BLAS screen + argpartition + float64 re-score of k: approximate at the cut",
                                 "neighbour_overlap_with_gpu": agree}
        del X64
    except MemoryError:
        pass
    # the reference-SHAPED loop (scikit-learn KD-tree + per-anchor Python loop, one thread: SURVEY.md §8d (1))
    if I <= 50000:
        try:
            n_ref = 48
            ref_sel = np.linspace(0, U - 1, n_ref).astype(np.int64)
            Vall = Vh.astype(np.float64)
            t1 = time.perf_counter()
            r_idx, _ = orc.reference_shaped_sweep(Vall, np.arange(U, U + I), ref_sel.tolist(), K)
            t_ref = time.perf_counter() - t1
            out["reference_shaped"] = {"value": n_ref * float(I) / t_ref, "unit": "pairs/s", "cores": 1, "ms_per_anchor": t_ref / n_ref * 1e3,
                                       "includes": "KD-tree build + %d anchors" % n_ref,
                                       "same_neighbours_as_gpu": bool(np.array_equal(r_idx, idx_h[ref_sel]))}
        except ImportError:
            pass
    return out


def api_latency(torch, Vh, U, I, d, arrays_only=False):
    """Wall time of the retrieval sweep THROUGH the drop-in Python API (reference: `retrieval_time` = the wall clock of
    get_topk for all anchors, hwer/validation.py:110-113, :30-35), results on the host: query construction, search, D2H copy
    and Python object construction included."""
    from hwer import Edge, Node, VectorIndex, validation
    out = {}
    users = [Node("user", i) for i in range(U)]
    items = [Node("item", i) for i in range(I)]
    t1 = time.perf_counter()
    m = VectorIndex(node_types={"user", "item"}, n_dims=d)
    m.fit(users + items, Vh.astype(np.float64))
    torch.cuda.synchronize()
    out["index_build_ms"] = (time.perf_counter() - t1) * 1e3

    def med(fn, n=5):
        fn()
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts)) * 1e3
    out["find_closest_neighbours_batch"] = med(lambda: m.find_closest_neighbours_batch("item", users, k=K), 5 if U < 50000 else 2)
    out["find_closest_neighbours_batch_note"] = "all %d anchors -> (idx, score) numpy arrays on the host" % U
    if arrays_only:
        return out
    out["model_get_topk"] = med(lambda: validation.model_get_topk(m, users, "item"))
    out["model_get_topk_note"] = "the reference's retrieval_time call: {anchor: [(Node, score)] * 200}; lists are materialised on access"

    def materialise():
        r = validation.model_get_topk(m, users, "item")
        for u in users:
            r[u]
    out["model_get_topk_all_lists_materialised"] = med(materialise, 3)
    n_loop = min(U, 64)
    m.find_closest_neighbours("item", users[0], k=K)      # (first call: builds the cached hipGraph of this (node type, k))
    t1 = time.perf_counter()
    for u in users[:n_loop]:
        m.find_closest_neighbours("item", u, k=K)
    out["find_closest_neighbours_per_anchor"] = (time.perf_counter() - t1) / n_loop * 1e3
    out["find_closest_neighbours_per_anchor_note"] = "the reference-shaped loop (one call per anchor) through the drop-in class, ms per anchor (each call = one cached hipGraph replay)"
    m.find_closest_neighbours("item", users[0], k=K)
    ts = []
    for u in users[:min(U, 256)]:
        t = time.perf_counter()
        m.find_closest_neighbours("item", u, k=K)
        ts.append(time.perf_counter() - t)
    out["find_closest_neighbours"] = float(np.median(ts)) * 1e3
    out["find_closest_neighbours_note"] = "median wall time of ONE find_closest_neighbours call (hwer/recommendation_base.py:157-174): list of 200 (Node, score) on the host"
    import gc
    gc.collect()
    gc.disable()      # (timeit's convention: the loop keeps 943 x 200 tuples alive, and full collections over a process that has torch imported
    try:              #  cost tens of milliseconds each - measured 305 ms with, 70 ms without)
        t1 = time.perf_counter()
        validation.model_get_topk_knn_per_anchor(m, users, "item")
        out["model_get_topk_knn_per_anchor"] = (time.perf_counter() - t1) * 1e3
    finally:
        gc.enable()
    out["model_get_topk_knn_per_anchor_note"] = "the reference's retrieval loop as it stands (hwer/validation.py:30-35: one call per anchor, all %d anchors, every list materialised), beside the batched model_get_topk above" % U
    # the ContentRecommendation flavour since round 5: GENUINE float64 rows (the float64 blend stage returns them) - the index screens on their
    # float32 rounding and answers from the float64 table (DeviceIndex._search_candidates; per-anchor calls: the cached hipGraph of _OneAnchorSearchF64)
    try:
        rng64 = np.random.default_rng(64)
        V64 = Vh.astype(np.float64) + 1e-9 * rng64.standard_normal(Vh.shape)
        V64 /= np.linalg.norm(V64, axis=1, keepdims=True)
        m64 = VectorIndex(node_types={"user", "item"}, n_dims=d)
        m64.fit(users + items, V64)
        assert m64.knn.index.V64 is not None
        m64.find_closest_neighbours("item", users[0], k=K)
        ts = []
        for u in users[:min(U, 64)]:
            t = time.perf_counter()
            m64.find_closest_neighbours("item", u, k=K)
            ts.append(time.perf_counter() - t)
        out["find_closest_neighbours_f64rows"] = float(np.median(ts)) * 1e3
        m64.find_closest_neighbours_batch("item", users, k=K)      # (first call: the caching allocator's blocks)
        out["find_closest_neighbours_batch_f64rows"] = med(lambda: m64.find_closest_neighbours_batch("item", users, k=K), 7)
        out["f64rows_note"] = "genuine float64 `vectors` (what ContentRecommendation.fit returns since the float64 blend stage): one call / all %d anchors in one call" % U
        del m64
    except Exception as e:      # (a probe: never fails the bench line)
        out["find_closest_neighbours_f64rows_error"] = repr(e)[:200]
    # extraction_efficiency end to end on synthetic edges: per user 20 train items + 1 validation item
    rng = np.random.default_rng(4001)
    train, valid = [], []
    for ui, u in enumerate(users):
        its = rng.choice(I, 21, replace=False)
        for j in its[:20]:
            train.append(Edge(u, items[int(j)], float(1 + (ui + int(j)) % 5)))
        valid.append(Edge(u, items[int(its[20])], 5.0))
    import random
    random.seed(0)
    t1 = time.perf_counter()
    res = validation.extraction_efficiency(m, train, valid, validation.model_get_topk, "item")
    t_ee = time.perf_counter() - t1
    random.seed(0)
    t1 = time.perf_counter()
    validation.ncf_eval(m, train, valid, items)
    t_ncf = time.perf_counter() - t1
    out["extraction_efficiency"] = t_ee * 1e3
    out["extraction_efficiency_parts"] = {"retrieval_time": res["metrics"]["retrieval_time"] * 1e3, "ncf_eval (host sampling, reference semantics)": t_ncf * 1e3}
    return out


# ====================================================================================================================
# N = 1
# ====================================================================================================================
def wide_probe(torch, dev):
    """Shapes outside the fused HIP search through the host mirror (results left on the device): a 515-column table (ML-20M-sized
    catalogue, 4 096 anchors, k = 200) and k = 1 000 on a 128-column one - round 4: hwer_knn_keys_f32 (float32 MFMA, 16-column slabs) +
    hwer_knn_topk_rows_f32 + exact candidate re-rank + hwer_knn_order_rows_f64 (`DeviceIndex._search_candidates`; rounds 1-3: a library
    GEMM / top-k / sorts) next to the all-pairs float64 pass it replaced (`_search_exhaustive`, forced)."""
    from hwer.device_index import DeviceIndex
    out = {}
    for tag, d, k, B, N in (("d515_k200", 515, 200, 4096, 26744), ("d128_k1000", 128, 1000, 4096, 26744)):
        V = gen_unit_rows(torch, 7000 + d, 0, B + N, d, dev)
        ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, device=dev, arith=0, audit=False)
        Q64, Q32, a = ix.build_queries(np.arange(B))
        ent = {"users": B, "items": N, "d": d, "k": k}
        for name, force in (("candidates_ms", None), ("exhaustive_ms", "1")):
            if force:
                os.environ["HWER_FORCE_EXHAUSTIVE"] = force
            try:
                fn = lambda: ix.search_queries("item", Q64, Q32, k, a)       # noqa: E731
                ent[name] = cuda_time(torch, fn, 3, warm=1)
            finally:
                os.environ.pop("HWER_FORCE_EXHAUSTIVE", None)
        ent["speedup"] = ent["exhaustive_ms"] / ent["candidates_ms"]
        out[tag] = ent
        del ix, V
        torch.cuda.empty_cache()
    return out


def run_single(args):
    import torch
    from hwer import native
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = native.lib()
    base = {"metric": METRIC, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic"}
    cpu = (lambda s_: 0.0 if args.no_cpu_baseline else s_)
    if args.config and CONFIGS[args.config].get("f64rows"):
        return finish_line(base, run_config_f64rows(torch, lib, args.config, dev, args.steps, args.warmup, args))
    if args.config:      # one configuration only (profiling runs)
        r = run_config(torch, lib, args.config, dev, args.steps, args.warmup, args, cpu_budget_s=cpu(8.0),
                       latencies=() if args.sweep_only else ((1,) if args.config != "c4" else (1, 64, 4096)), f32_beside=not args.sweep_only)
        return finish_line(base, r)
    t_start = time.perf_counter()
    top = run_config(torch, lib, "c1", dev, args.steps, args.warmup, args, cpu_budget_s=cpu(8.0), api=True, latencies=(1,))
    configs = {}
    configs["c2"] = run_config(torch, lib, "c2", dev, args.steps, args.warmup, args, cpu_budget_s=cpu(5.0), api=True, latencies=(1,))
    configs["c3"] = run_config(torch, lib, "c3", dev, max(10, min(args.steps, 20)), max(3, min(args.warmup, 5)), args, cpu_budget_s=cpu(4.0),
                               api="arrays", stage_reps=5, f32_beside=True)
    # the big shapes: >= 5 warm-ups and >= 20 (c4, c5) / >= 10 (c4full) steps whatever --steps says - a 2-warm-up, 10-step wall-clock
    # mean right after the previous configuration's tensors were freed was what the round-2 driver run caught at 4.41 ms
    configs["c4"] = run_config(torch, lib, "c4", dev, max(20, min(args.steps, 40)), max(5, min(args.warmup, 10)), args, latencies=(1, 64, 4096), stage_reps=10, f32_beside=True)
    configs["c5"] = run_config(torch, lib, "c5", dev, max(20, min(args.steps, 40)), max(5, min(args.warmup, 10)), args, stage_reps=10, f32_beside=True)
    configs["c4full"] = run_config(torch, lib, "c4full", dev, max(10, min(args.steps, 20)), max(5, min(args.warmup, 5)), args, stage_reps=3)
    # the reference's float64 flavour of the headline shape and of the roofline shape (round 6: fused float64-row search)
    configs["c1_f64rows"] = run_config_f64rows(torch, lib, "c1_f64rows", dev, args.steps, args.warmup, args)
    configs["c3_f64rows"] = run_config_f64rows(torch, lib, "c3_f64rows", dev, max(10, min(args.steps, 20)), max(3, min(args.warmup, 5)), args)
    configs["c4full"]["note"] = ("the whole 10M-item catalogue on ONE GPU: the 1-GPU value of the workload that `bench.py --gpus N` (N > 1) strong-scales; "
                                 "use this value as the N = 1 point of the scaling curve")
    out = finish_line(base, top)
    out["configs"] = configs
    out["wide_or_large_k"] = wide_probe(torch, dev)
    out["strong_scaling_headroom"] = {"c4full_over_c4_median": configs["c4full"]["ms_median"] / configs["c4"]["ms_median"],
                                      "note": "one GPU's step on the whole 10M-row catalogue / its step on one 1.25M-row shard: the 8-GPU speedup before exchange effects"}
    out["bench_wall_s"] = time.perf_counter() - t_start
    return out


def finish_line(base, r):
    out = dict(base)
    out.update({"value": r["value"], "ms_per_step": r["ms_per_step"], "dtype": r["dtype"], "steps": r["steps"], "warmup": r["warmup"],
                "config": {"workload": r["workload"], "users": r["users"], "items_per_gpu": r["items"], "items_total": r["items"], "d": r["d"],
                           "k": r["k"], "tables": r["tables"], "alpha": r["alpha"], "f16_screen": r["f16_screen"], "sharding": "none"}})
    for k_ in ("ms_median", "ms_min", "ms_max", "ms_mean_events", "timed_region", "ms_series", "ms_series_note", "roofline", "dominant_kernel", "stage_frac", "stage_frac_note",
               "kernels", "stages_ms", "stage_sum_ms", "stage_sum_note", "stages_in_step_ms", "stage_sum_separate_ms", "certified_queries", "exhaustive_fallback_queries",
               "parity_checked_queries", "hip_graph", "f32_screen", "query_latency_ms", "pair_score", "cpu_baseline", "speedup_vs_cpu",
               "speedup_vs_reference_shaped_loop", "api_latency_ms", "knn_screen_ms", "knn_select_f64rows_ms", "select_gather", "dtype_note", "parity_note"):
        if k_ in r:
            out[k_] = r[k_]
    if "api_latency_ms" in r and "model_get_topk" in r["api_latency_ms"]:
        out["recall_latency_ms"] = r["api_latency_ms"]["model_get_topk"]
        out["recall_latency_note"] = "wall time of model_get_topk for all anchors through the drop-in API (the reference's retrieval_time), results on the host"
    return out


def _r(x, sig=5):
    """float -> `sig` significant digits (the compact line is capped at ~3 KB)"""
    if x is None or isinstance(x, (bool, int, str)):
        return x
    x = float(x)
    if x == 0.0 or x != x or x in (float("inf"), float("-inf")):
        return x
    from math import floor, log10
    return round(x, sig - 1 - int(floor(log10(abs(x)))))


def compact_line(full, detail_path):
    """The LAST stdout line: <= 3 KB, everything the driver / judge reads first; the full object goes to `detail_path`."""
    def roof(r):
        if not r:
            return None
        out = {k_: (_r(r[k_]) if k_ in ("achieved", "peak", "frac") else r.get(k_)) for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source") if k_ in r}
        out["kernel"] = str(out.get("kernel", ""))[:48]
        return out
    out = {k_: full.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_median", "ms_min", "ms_max", "higher_is_better",
                                      "scaling", "vs_baseline", "dtype", "data")}
    for k_ in ("value", "ms_per_step", "ms_median", "ms_min", "ms_max"):
        out[k_] = _r(out.get(k_), 6)
    c = full.get("config", {})
    out["config"] = {k_: c.get(k_) for k_ in ("workload", "users", "items_total", "d", "k", "tables", "alpha", "sharding") if k_ in c}
    out["roofline"] = roof(full.get("roofline"))
    if "dominant_kernel" in full:
        out["dominant_kernel"] = full["dominant_kernel"]
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:72],
                               "cpu_model": (cb.get("host") or {}).get("cpu_model"),
                               "reference_shaped_value": _r((cb.get("reference_shaped") or {}).get("value"))}
        out["speedup_vs_cpu"] = _r(full.get("speedup_vs_cpu"), 4)
        out["speedup_vs_reference_shaped_loop"] = _r(full.get("speedup_vs_reference_shaped_loop"), 4)
    if "recall_latency_ms" in full:
        out["recall_latency_ms"] = _r(full["recall_latency_ms"], 4)
    al = full.get("api_latency_ms") or {}
    if "find_closest_neighbours" in al:      # one call of the reference's per-anchor API, and its loop over every anchor, beside the batched sweep
        out["api_latency_ms"] = {"find_closest_neighbours": _r(al["find_closest_neighbours"], 4), "per_anchor_loop_all": _r(al.get("model_get_topk_knn_per_anchor"), 4),
                                 "find_closest_neighbours_f64rows": _r(al.get("find_closest_neighbours_f64rows"), 4),
                                 "batch_all_anchors": _r(al.get("find_closest_neighbours_batch"), 4), "batch_all_anchors_f64rows": _r(al.get("find_closest_neighbours_batch_f64rows"), 4)}
    tr = full.get("timed_region")
    if tr:      # always in the line: which region is reported, and whether a second one was taken
        out["timed_region"] = {"attempts": tr["attempts"], "reported": tr.get("reported"), "remeasured_ms_per_step": tr.get("remeasured_ms_per_step")}
        if tr.get("steady_ms_per_step") is not None:     # graph-replayed steps timed with fewer than 500 steps: the same step over 500 (clocks settled)
            out["timed_region"]["steady_ms_per_step"] = tr["steady_ms_per_step"]
    if full.get("query_latency_ms"):
        out["query_latency_ms"] = {str(e["batch"]): _r(e["ms"], 4) for e in full["query_latency_ms"]}
    cfgs = {}
    for name, r in (full.get("configs") or {}).items():
        e = {"ms_per_step": _r(r["ms_per_step"]), "ms_median": _r(r["ms_median"]), "ms_min": _r(r["ms_min"]), "value": _r(r["value"]), "dtype": r["dtype"],
             "frac": _r(r["roofline"]["frac"], 3), "bound": r["roofline"]["bound"], "stage_sum_ms": _r(r.get("stage_sum_ms"), 4),
             "parity_checked_queries": r["parity_checked_queries"]}
        if "f32_screen" in r:
            e["f32_screen_ms"] = _r(r["f32_screen"]["ms_per_step"], 4)
            e["f32_screen_frac"] = _r(r["f32_screen"].get("knn_screen_frac_of_fp32_mfma_peak"), 3)
        if r.get("query_latency_ms"):
            e["query_latency_ms"] = {str(q["batch"]): _r(q["ms"], 4) for q in r["query_latency_ms"]}
        if "speedup_vs_cpu" in r:
            e["speedup_vs_cpu"] = _r(r["speedup_vs_cpu"], 4)
        e["timed_region_attempts"] = (r.get("timed_region") or {}).get("attempts")
        cfgs[name] = e
    if cfgs:
        out["configs"] = cfgs
    if "wide_or_large_k" in full:
        out["wide_or_large_k"] = {t: {"ms": _r(e["candidates_ms"], 4), "exhaustive_ms": _r(e["exhaustive_ms"], 4)} for t, e in full["wide_or_large_k"].items()}
    for k_ in ("strong_scaling_headroom", "emulated_shards"):
        if k_ in full:
            v = full[k_]
            out[k_] = {kk: (_r(vv) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk != "note" and not isinstance(vv, (list, dict))}
    out["bench_wall_s"] = _r(full.get("bench_wall_s"), 4)
    out["detail"] = detail_path
    return out


def write_detail(full, path):
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh)
        return path
    except OSError as e:
        sys.stderr.write("[bench] could not write %s: %s\n" % (path, e))
        return None


# ====================================================================================================================
# one GPU rehearses the G-GPU job (--emulate-shards G): same shards, same pack / merge kernels, same buffer layout - no collective
# ====================================================================================================================
def emulate_shards(torch, lib, cfg, dev, G, steps=3, warmup=2, n_oracle=32, f32_screen=False, log=None, list_len=None):
    """-> dict.  The catalogue of `cfg` on ONE GPU, three ways: (1) as a single table (the N = 1 job); (2) cut into the G
    `shard_bounds` ranges, each searched with ids = id_base + local row exactly as rank g of `bench.py --gpus G` does (its own blend,
    its own per-type table, local (distance, id) order), packed by hwer_pack_exchange into slice g of ONE (G, 3, B, k) int64 buffer -
    the layout `all_gather_into_tensor` gives the receive buffer - and merged in place by hwer_merge_topk_strided; (3) the oracle's
    brute force over the whole catalogue on `n_oracle` sampled queries.  Asserts (2) == (1) for every query (ids, scores, squared
    distances) and (2) == (3) on the sample.
    Round 3: the shards send shortened lists (list_len entries, default hwer.distributed.local_list_len(K, G); K = full lists) and
    hwer_merge_topk_lists proves per query that none ran out; if one did the rehearsal is repeated with full lists, as the job would."""
    from hwer import native
    from hwer.distributed import local_list_len, merge_on_device, pack_send_buffer, shard_bounds
    from oracle import c_oracle
    from oracle import hwer_oracle as orc
    I_total, d = cfg["I"], cfg["d"]
    say = log or (lambda *_: None)
    # (1) the single table
    w1 = Workload(torch, lib, cfg, dev, f32_screen=f32_screen, want_dist=True)
    B, U = w1.B, w1.U
    for _ in range(warmup):
        w1.step()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    e[0].record()
    for i in range(steps):
        w1.step()
        e[i + 1].record()
    torch.cuda.synchronize()
    single_ms = float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(steps)]))
    n_repaired_single = w1.repair(w1.sets[0])     # (the reference result of the comparison: queries beyond the on-device pass's 8 per search, if any)
    n_exh_single = w1.checks(w1.sets[0])
    one = {k_: w1.sets[0][k_].clone() for k_ in ("idx", "score", "dist2")}
    items_h = w1.V[U:, :d].cpu().numpy() if n_oracle else None
    users_h = w1.V[:U, :d].cpu().numpy().astype(np.float64)
    del w1
    torch.cuda.empty_cache()
    say("single table: %.3f ms / step" % single_ms)
    # (2) the G shards, one after the other, into the all-gather's receive layout
    k_in = local_list_len(K, G) if list_len is None else max(min(int(list_len), K), -(-K // G))
    recv = torch.empty((G, 3, B, k_in), dtype=torch.int64, device=dev)
    shard_ms, n_exh = [], 0
    for g in range(G):
        lo, hi = shard_bounds(I_total, G, g)
        w = Workload(torch, lib, cfg, dev, lo=lo, hi=hi, f32_screen=f32_screen, want_dist=True, local_order=True, k=k_in)
        for _ in range(warmup):
            w.step()
        e[0].record()
        for i in range(steps):
            w.step()
            e[i + 1].record()
        torch.cuda.synchronize()
        shard_ms.append(float(np.median([e[i].elapsed_time(e[i + 1]) for i in range(steps)])))
        n_exh += w.checks(w.sets[0])
        o = w.sets[0]
        recv[g].copy_(pack_send_buffer(o["idx"], o["score"], o["dist2"], 1)[0])
        say("shard %d rows [%d, %d): %.3f ms / step" % (g, lo, hi, shard_ms[-1]))
        del w, o
        torch.cuda.empty_cache()
    views = (recv[:, 0], recv[:, 1].view(torch.float64), recv[:, 2].view(torch.float64))     # strided: list stride 3 B k_in
    *m, ran_out = merge_on_device(*views, True, k=K, want_exhausted=True)
    n_ran_out = int(ran_out.sum().item())
    if n_ran_out and k_in < K:   # a list ran out: the job repeats with full lists - so does the rehearsal (its numbers are then the full-list ones)
        say("%d queries saw a %d-entry list run out: repeating with full lists" % (n_ran_out, k_in))
        del recv, views, m
        r = emulate_shards(torch, lib, cfg, dev, G, steps, warmup, n_oracle, f32_screen, log, list_len=K)
        r["lists_ran_out_first_pass"] = n_ran_out
        return r
    merge_ms = cuda_time(torch, lambda: merge_on_device(*views, True, k=K, want_exhausted=True), 5, warm=1)
    m_idx, m_score, m_d2 = (t.cpu().numpy() for t in m)
    same = {"idx": bool(np.array_equal(m_idx, one["idx"].cpu().numpy())), "score": bool(np.array_equal(m_score, one["score"].cpu().numpy())),
            "dist2": bool(np.array_equal(m_d2, one["dist2"].cpu().numpy()))}
    assert all(same.values()), "merged shards differ from the single table: %s" % same
    # (3) oracle on sampled queries
    n_par = 0
    if n_oracle:
        cores, _ = effective_cpus()
        sel = np.unique(np.linspace(0, B - 1, n_oracle).astype(np.int64))
        Qh = np.stack([orc.query_vector(users_h, int(q)) for q in sel])
        oi, osc, od2 = c_oracle.find_closest_f64(items_h, Qh, users_h[sel], K, threads=cores)
        assert np.array_equal(m_idx[sel], oi + U), "merged indices differ from the oracle"
        assert np.array_equal(m_score[sel], osc), "merged scores differ from the oracle"
        assert np.array_equal(m_d2[sel], od2), "merged distances differ from the oracle"
        n_par = int(len(sel))
    return {"shards": G, "queries": B, "items_total": I_total, "single_table_ms": single_ms, "shard_ms": shard_ms, "shard_ms_max": max(shard_ms),
            "merge_ms": merge_ms, "speedup_before_exchange": single_ms / (max(shard_ms) + merge_ms), "merged_equals_single_table": True,
            "queries_compared": B, "oracle_checked_queries": n_par, "exhaustive_fallback_queries": {"single": n_exh_single, "shards": n_exh, "single_host_repaired": n_repaired_single},
            "local_list_len": k_in, "lists_ran_out_first_pass": 0, "exchange_bytes_per_rank": 24 * B * k_in,
            "recv_layout": "(G, 3, B, k_in) int64 = all_gather_into_tensor's receive buffer; merged in place by hwer_merge_topk_lists (list stride 3*B*k_in)"}


def run_emulated_shards(args):
    import torch
    from hwer import native
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    lib = native.lib()
    name = args.config or "c4full"
    cfg = dict(CONFIGS[name])
    if os.environ.get("HWER_BENCH_ITEMS"):
        cfg["I"] = int(os.environ["HWER_BENCH_ITEMS"])
        cfg["name"] += " [catalogue cut to %d items]" % cfg["I"]
    t0 = time.perf_counter()
    r = emulate_shards(torch, lib, cfg, dev, args.emulate_shards, steps=max(3, min(args.steps, 10)), warmup=max(2, min(args.warmup, 3)),
                       f32_screen=args.f32_screen, log=lambda m: sys.stderr.write("[emulate] %s\n" % m),
                       list_len=K if os.environ.get("HWER_BENCH_FULL_LISTS") else (int(os.environ["HWER_BENCH_LIST_LEN"]) if os.environ.get("HWER_BENCH_LIST_LEN") else None))
    B, I = r["queries"], r["items_total"]
    # top level = what this ONE GPU measured on the whole catalogue (the single-table step); the would-be G-GPU rate lives under
    # `emulated_shards` only (`rehearsed_step_ms`, `rehearsed_value`): n_gpus is 1 and `value` must be a 1-GPU number
    step_ms = r["single_table_ms"]
    r["rehearsed_step_ms"] = r["shard_ms_max"] + r["merge_ms"]
    r["rehearsed_value"] = float(B) * I / (r["rehearsed_step_ms"] * 1e-3)
    return {"metric": METRIC, "value": float(B) * I / (step_ms * 1e-3), "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "ms_median": step_ms, "ms_min": step_ms, "ms_max": step_ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.f32_screen else "f16", "data": "synthetic",
            "config": {"workload": cfg["name"] + ": the single table on one GPU (top level) + the %d-GPU job REHEARSED shard by shard on it (emulated_shards; no collective)" % r["shards"], "users": B,
                       "items_total": I, "d": cfg["d"], "k": K, "tables": cfg["T"], "alpha": "per-row" if cfg["per_row"] else "scalar",
                       "sharding": "none (emulated x%d under emulated_shards)" % r["shards"]},
            "emulated_shards": r, "bench_wall_s": time.perf_counter() - t0}


# ====================================================================================================================
# N > 1: strong scaling of the 10 M-item catalogue
# ====================================================================================================================
def run_sharded(args, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    from hwer import native
    import datetime
    from hwer.distributed import ShardedStepPipeline, local_list_len, shard_bounds
    backend = os.environ.get("HWER_BENCH_BACKEND", "nccl")     # "gloo" only to exercise the N > 1 code path on a 1-GPU box
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a rank that never arrives (crashed before the rendezvous, wrong port) must not hold the others for torch's default 10-30 minutes:
    # the rendezvous and every collective give up after HWER_BENCH_DIST_TIMEOUT_S (default 600 s: ranks 1 .. N-1 wait in the closing barrier while rank 0 runs
    # the CPU baseline, the oracle check and the one-GPU anchor - tens of seconds on a busy host; advisor, round 5) and the rank exits non-zero
    # (self_launch / torchrun then end the job)
    tmo = datetime.timedelta(seconds=float(os.environ.get("HWER_BENCH_DIST_TIMEOUT_S", "600")))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    if os.environ.get("HWER_BENCH_FAIL_RANK") == str(rank):     # rehearsal of a dead rank (tests/test_bench_launcher.py): die after the rendezvous
        sys.stderr.write("[bench] rank %d: HWER_BENCH_FAIL_RANK set - exiting with code 3\n" % rank)
        sys.stderr.flush()
        os._exit(3)
    lib = native.lib()
    name = args.config or "c4full"
    cfg = dict(CONFIGS[name])
    if os.environ.get("HWER_BENCH_ITEMS"):       # rehearsals on small boxes: a smaller catalogue, same code path
        cfg["I"] = int(os.environ["HWER_BENCH_ITEMS"])
        cfg["name"] += " [catalogue cut to %d items]" % cfg["I"]
    I_total, U, d = cfg["I"], cfg["U"], cfg["d"]
    assert cfg.get("B", U) == U, "the strong-scaling job expects every user row to be a query (c4full, c4)"
    lo, hi = shard_bounds(I_total, world, rank)
    # Shortened local lists (hwer.distributed.local_list_len): a shard sends its k_in < K nearest - ~K / N of a query's K nearest live on
    # it - and the merge proves per query that no list ran out (hwer_merge_topk_lists: out_exhausted).  A run in which any did is
    # repeated as a whole with full lists (below); HWER_BENCH_FULL_LISTS=1 forces full lists.
    k_in = K if os.environ.get("HWER_BENCH_FULL_LISTS") else local_list_len(K, world)
    if os.environ.get("HWER_BENCH_LIST_LEN"):     # rehearsals: force a list length (a 2-rank job keeps full lists by itself)
        k_in = max(min(int(os.environ["HWER_BENCH_LIST_LEN"]), K), -(-K // world))
    w = Workload(torch, lib, cfg, dev, lo=lo, hi=hi, f32_screen=args.f32_screen, want_dist=True, local_order=True, n_sets=2, k=k_in)
    # The step itself is the LIBRARY's: hwer.distributed.ShardedStepPipeline = the resident BlendSearchStep over this rank's shard, ONE
    # exchange per search (all-gather or the user-partitioned all-to-all) + hwer_merge_topk_lists on a side stream, the exchange of step i
    # under the search of step i + 1, the "a list ran out" mask OR-ed on the device over every step of the job (no host sync in a step).
    pipes = {}
    ev_log = []      # (start, collective done, merge done) events of every step, read after the closing barrier

    def make_step(exchange):
        pipe = pipes[exchange] = ShardedStepPipeline(w, K, exchange=exchange, sort_by_score=True, time_events=True)
        pipe.event_log = ev_log
        return pipe.run

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup, profile=False):
        for _ in range(warmup):
            res = step()
        barrier()
        ev_log.clear()
        if profile:
            lib.hwer_profile_enable(PROFILE_EVERY)
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        barrier()
        el = time.perf_counter() - t0
        if profile:
            lib.hwer_profile_enable(0)
        tt = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        xs = [(a.elapsed_time(b), b.elapsed_time(c)) for a, b, c in ev_log]
        return float(tt.item()), res, xs

    main_exchange = os.environ.get("HWER_BENCH_EXCHANGE", "all_gather")
    elapsed, res, xs = timed(make_step(main_exchange), args.steps, args.warmup, profile=True)
    screen_ms, select_ms, _ = read_profile(lib)
    last = pipes[main_exchange].last_local_set()
    n_exhaustive = w.checks(last)
    other = "by_query" if main_exchange == "all_gather" else "all_gather"
    el2, _, xs2 = timed(make_step(other), args.steps, min(args.warmup, 2))
    # any list that ran out, on any rank, in any step of either variant: the shortened lists are not valid for this catalogue -
    # the whole measurement is repeated with full lists (every rank takes the same branch: the count is all-reduced)
    n_ran_out = torch.tensor([sum(p_.exhausted_queries() for p_ in pipes.values())], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(n_ran_out, op=dist.ReduceOp.MAX)
    lists_ran_out = int(n_ran_out.item())
    if lists_ran_out and k_in < K:
        if rank == 0:
            sys.stderr.write("[bench] %d queries saw a local list of %d run out: repeating the job with full lists\n" % (lists_ran_out, k_in))
        del w
        torch.cuda.empty_cache()
        k_in = K
        w = Workload(torch, lib, cfg, dev, lo=lo, hi=hi, f32_screen=args.f32_screen, want_dist=True, local_order=True, n_sets=2, k=k_in)
        elapsed, res, xs = timed(make_step(main_exchange), args.steps, args.warmup, profile=True)
        screen_ms, select_ms, _ = read_profile(lib)
        last = pipes[main_exchange].last_local_set()
        n_exhaustive = w.checks(last)
        el2, _, xs2 = timed(make_step(other), args.steps, min(args.warmup, 2))
    idx_h, score_h = res[0].cpu().numpy(), res[1].cpu().numpy()

    # ---- parity: rank 0 collects every rank's blended item rows (rank order = global id order) and checks the merged lists of
    # sampled users against the oracle's brute force over the WHOLE catalogue ----------------------------------------------
    per = (I_total + world - 1) // world
    shard = torch.zeros((per, d), dtype=torch.float32, device=dev)
    shard[:hi - lo] = w.V[U:, :d]
    parity_n, result_sha, cpu_base = 0, None, None
    if backend == "nccl":
        allv = torch.empty((world * per, d), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allv, shard)
        items_h = allv[:I_total].cpu().numpy() if rank == 0 else None
        del allv
    else:
        parts = [torch.empty((per, d), dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, shard.cpu())
        items_h = torch.cat(parts)[:I_total].numpy() if rank == 0 else None
    if rank == 0:
        import hashlib
        from oracle import c_oracle
        from oracle import hwer_oracle as orc
        cores, _ = effective_cpus()
        n_mine = idx_h.shape[0]      # all-gather: every query; by_query: rank 0 owns users [0, ceil(U/N))
        sel = np.unique(np.linspace(0, n_mine - 1, 16 if I_total <= 2000000 else 8).astype(np.int64))
        V64 = w.V[:U, :d].cpu().numpy().astype(np.float64)
        Qh = np.stack([orc.query_vector(V64, int(q)) for q in sel])
        oi, osc, _ = c_oracle.find_closest_f64(items_h, Qh, V64[sel], K, threads=cores)
        assert np.array_equal(idx_h[sel], oi + U), "merged indices differ from the oracle (N = %d)" % world
        assert np.array_equal(score_h[sel], osc), "merged scores differ from the oracle (N = %d)" % world
        parity_n = len(sel)
        result_sha = hashlib.sha256(np.ascontiguousarray(idx_h[:64]).tobytes()).hexdigest()[:16]
        # CPU baseline of the SAME workload (the whole catalogue), the oracle's C restatement on rank 0's host cores, bounded sample -
        # measured after the timed region, beside the N-GPU value it is compared with (the N = 1 line carries the same leg)
        if not args.no_cpu_baseline:
            n_s = int(min(U, max(16, 0.7e9 * 4.0 * cores / (float(I_total) * d * 8))))
            cs = np.linspace(0, U - 1, n_s).astype(np.int64)
            Qc = V64[cs] / np.linalg.norm(V64[cs], axis=1, keepdims=True)
            t1 = time.perf_counter()
            c_oracle.find_closest_f64(items_h, Qc, V64[cs], K, threads=cores)
            t_cpu = time.perf_counter() - t1
            cpu_base = {"value": n_s * float(I_total) / t_cpu, "unit": "pairs/s", "cores": cores, "kind": "port",
                        "sample": "%d of %d queries x %d items, one pass (%.1f s): exact float64 brute force + re-score + sort (oracle/hwer_oracle.c, OpenMP), rank 0's host"
                                  % (n_s, U, I_total, t_cpu)}
        del items_h

    # ---- the same workload on ONE GPU, measured in this job by rank 0 (the denominator of the strong-scaling speedup) --------
    n1 = None
    del w
    torch.cuda.empty_cache()
    if rank == 0 and not os.environ.get("HWER_BENCH_NO_ANCHOR"):
        w1 = Workload(torch, lib, cfg, dev, f32_screen=args.f32_screen)
        for _ in range(2):
            w1.step()
        torch.cuda.synchronize()
        n1_steps = max(3, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n1_steps):
            w1.step()
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / n1_steps
        w1.checks(w1.sets[0])
        same = bool(np.array_equal(w1.sets[0]["idx"].cpu().numpy(), idx_h)) if main_exchange == "all_gather" else None
        n1 = {"ms_per_step": t1 * 1e3, "value": float(U) * I_total / t1, "steps": n1_steps, "same_result_as_sharded": same,
              "note": "the whole catalogue on rank 0's GPU alone, same step, measured after the timed region of this job"}
        del w1
    barrier()

    value = float(U) * float(I_total) / (elapsed / args.steps)
    ms_per_step = elapsed / args.steps * 1e3
    f16 = (not args.f32_screen) and (hi - lo) > 8192
    flops_local = 2.0 * U * (hi - lo) * d
    achieved = flops_local / (screen_ms * 1e-3) / 1e12 if screen_ms > 0 else 0.0
    peak = MFMA_F16_PEAK_TFLOPS if f16 else MFMA_F32_PEAK_TFLOPS
    xa = np.asarray(xs) if xs else np.zeros((0, 2))
    xb = np.asarray(xs2) if xs2 else np.zeros((0, 2))
    out = {
        "metric": METRIC, "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16" if f16 else "f32", "data": "synthetic",
        "config": {"workload": cfg["name"] + ", strong scaling: the SAME catalogue for every N", "users": U, "items_per_gpu": hi - lo, "items_total": I_total,
                   "d": d, "k": K, "tables": cfg["T"], "alpha": "per-row" if cfg["per_row"] else "scalar", "f16_screen": f16,
                   "sharding": "items row-sharded x%d, queries replicated, ONE %s (backend %s%s) of the local top-%d per step + hwer_merge_topk_lists "
                               "(hwer.distributed.ShardedStepPipeline); the exchange of step i overlaps the search of step i+1"
                               % (world, "all-gather" if main_exchange == "all_gather" else "user-partitioned all-to-all", backend,
                                  " = RCCL over xGMI" if backend == "nccl" else ": a rehearsal, payload staged through host memory", k_in),
                   "backend": backend, "ranks": dist.get_world_size()},
        "roofline": {"kernel": "knn_screen on this rank's shard", "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "vs_fp32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": screen_ms},
        "kernels_ms": {"knn_screen": screen_ms, "knn_select_rerank": select_ms},
        "exchange": {"kind": main_exchange, "bytes_sent_per_rank_per_step": 24 * U * k_in if main_exchange == "all_gather" else 24 * U * k_in * (world - 1) // world,
                     "local_list_len": k_in, "lists_ran_out_first_pass": lists_ran_out,
                     "collective_ms": float(xa[:, 0].mean()) if len(xa) else None, "merge_ms": float(xa[:, 1].mean()) if len(xa) else None,
                     "note": "side-stream times (HIP events around the collective and around the merge kernel), overlapped with the next step's search"},
        "variants": {other: {"ms_per_step": el2 / args.steps * 1e3, "value": float(U) * I_total / (el2 / args.steps),
                             "collective_ms": float(xb[:, 0].mean()) if len(xb) else None, "merge_ms": float(xb[:, 1].mean()) if len(xb) else None}},
        "certified_queries": U - n_exhaustive, "exhaustive_fallback_queries": n_exhaustive, "parity_checked_queries": parity_n,
        "result_sha16_first64_queries": result_sha,
        "cpu_baseline": cpu_base, "speedup_vs_cpu": (value / cpu_base["value"]) if cpu_base else None,
        "one_gpu_same_workload": n1, "speedup_vs_one_gpu": (value / n1["value"]) if n1 else None,
        "scaling_note": "N = 1 point of this curve: `configs.c4full` of the `--gpus 1` line, or one_gpu_same_workload here",
    }
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps of the top-level configuration (c1: 1 000 x 0.05 ms; the larger configurations clamp it to 20-40)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="N = 1: run only this configuration (profiling); default = c1 at the top level + c2, c3, c4, c4full under `configs`.  "
                         "N > 1: the catalogue to strong-scale (default c4full)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--f32-screen", action="store_true",
                    help="switch HWER_KNN_F16_SCREEN off.  By default the streaming screen (catalogues above 8 192 rows: c3, c4, c5) runs "
                         "on a float16 copy of the table, rebuilt every step like the float32 one (results unchanged, exact float64 "
                         "re-rank); the dense path (c1, c2) always screens in float32")
    ap.add_argument("--f16-dense", action="store_true", help="experiment: also run the small-catalogue screen (c1, c2) on a float16 copy of the table")
    ap.add_argument("--no-graph", action="store_true", help="time the eager step of the small configurations instead of a hipGraph replay of it")
    ap.add_argument("--sweep-only", action="store_true",
                    help="profiling runs: skip the latency / pair-score / float32-screen measurements after the timed region, so that "
                         "the launches in a rocprofv3 summary are (almost only) those of the timed sweep")
    ap.add_argument("--emulate-shards", type=int, default=0, metavar="G",
                    help="one GPU rehearses the G-GPU job: the 10M-row catalogue cut into the G shard_bounds ranges, every shard searched in turn "
                         "(ids = id_base + local row), packed by hwer_pack_exchange into a (G, 3, B, k) buffer laid out like the all-gather's "
                         "receive buffer, merged by hwer_merge_topk_strided and compared with the single-table result")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the full (~25 KB) result object goes; stdout carries ONE compact (<= 3 KB) JSON line")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        if world != 1:
            raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
        sys.exit(self_launch(args, sys.argv[1:]))      # nothing has touched a GPU in this process
    # the oracle's checker library is built / loaded BEFORE anything touches a GPU (its `make` is a fork + exec)
    from oracle import c_oracle
    c_oracle.lib()
    if world > 1 or os.environ.get("HWER_BENCH_SHARDED_WORLD1"):
        # (HWER_BENCH_SHARDED_WORLD1=1: the N > 1 job's code - rendezvous, ShardedStepPipeline, both exchanges, parity gather - over the REAL
        # backend (nccl = RCCL) at world size 1: what a one-GPU box can execute of it; tests/test_gpu_round5.py)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:      # a free port (two such rehearsals may share a host)
                import socket
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        run_sharded(args, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world)
        return
    full = run_emulated_shards(args) if args.emulate_shards else run_single(args)
    path = write_detail(full, args.detail_out)
    line = json.dumps(compact_line(full, os.path.relpath(path, ROOT) if path else None), separators=(",", ":"))
    print(line)
    sys.stdout.flush()


if __name__ == "__main__":
    main()
