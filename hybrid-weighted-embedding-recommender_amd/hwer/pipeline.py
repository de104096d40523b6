"""The hot path as ONE resident, allocation-free step (north_star): weighted blend of T embedding tables -> L2-normalise ->
unit-norm audit -> per-type table (+ half norms, + float16 copy) -> anchor queries -> exact top-k retrieval (MFMA screen +
float64 re-rank + re-score + sort; on-device exhaustive pass for any query the screen cannot certify).

Reference lines covered, in one object: hwer/content_recommender.py:77-96 (blend + unit_length), hwer/recommendation_base.py:
105-110 (audit + index build), :64-76 (per-type table), :153-155,164-170 (anchor query), :78-83,171-173 (k-NN + re-score +
sort).  `DeviceIndex` answers queries against a STATIC table; `BlendSearchStep` is for callers whose tables or weights change
between retrievals (new collaborative vectors after a training epoch, per-user alpha updates): every buffer is preallocated at
construction, `run()` only launches kernels on the current stream - no allocation, no host synchronisation - so a step can be
captured into a hipGraph and replayed (`capture()`).  It is what `bench.py` times, and what `ContentRecommendation.
blend_search_step` hands to library users.

Launch plan (chosen at construction, the same rules `bench.py` measured - DESIGN.md 4.6 / 6):
  * few anchors (8 B <= rows of the searched type) or a small catalogue: the whole prologue is ONE launch,
    hwer_blend_prepare_search_f32 (blend + audit + table + float16 copy + queries, and on float32 dense screens the search's
    MFMA query fragments: HWER_KNN_QUERIES_READY);
  * more anchors than rows: hwer_blend_normalize_audit_f32, hwer_pack_rows_f32_f16, hwer_query_mix (three launches);
  * then hwer_knn_search_opts_f32 per 131 072 queries (a short last chunk runs beside the first on a second stream) and
    hwer_knn_exact_device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

from . import native
from .device_index import roundup

KNN_F16_SCREEN, KNN_QUERIES_READY = 2, 32
QUERY_CHUNK = 131072       # = DeviceIndex.QUERY_CHUNK: one full round of workgroups per search launch
PIPE_CHUNK = 65536         # queries per chunk of a pipelined search: one full round of TQ = 1 screen workgroups (512 x 128)


class BlendSearchStep:
    """tables: T float32 device tensors (N, d) with row stride d (d % 4 == 0); alpha: (T,) or (N, T) float32 device tensor.
    Rows [search_row0, search_row0 + n_search) are the searched node type, rows [other_row0, other_row0 + n_other) everything
    else (both contiguous: the layout `RecommendationBase.add_nodes` gives when the nodes of a type are added together).
    anchors: int32 device tensor of B row ids (-1 = unknown node).  Reported ids = id_base + local row of the searched range
    (default: the row id in V).  search_ids / other_ids (int64 device tensors, ascending): the same two row sets as id lists, for
    registries whose types are interleaved (then reported ids = the listed row ids).  arith 0 / 1: the reference's float64 / float32 score flavour.  sort_by_score False: ascending
    (distance, id) order - what the multi-GPU merge consumes (then want_dist is forced)."""

    def __init__(self, tables: Sequence, alpha, search_row0: int, n_search: int, other_row0: int, n_other: int, anchors, k: int = 200,
                 id_base: Optional[int] = None, arith: int = 0, want_dist: bool = False, sort_by_score: bool = True,
                 f16_screen: Optional[bool] = None, f16_dense: bool = False, n_sets: int = 1, extra_options: int = 0,
                 fused_blend: Optional[bool] = None, fused_prologue: Optional[bool] = None, tail_stream: bool = False, eps: float = 1e-4,
                 search_ids=None, other_ids=None, pipeline: Optional[bool] = None, tail_under_select: Optional[bool] = None):
        import torch
        lib = self.lib = native.lib()
        self.torch, self.native = torch, native
        self.tables = list(tables)
        T = self.T = len(self.tables)
        dev = self.dev = self.tables[0].device
        N, d = self.tables[0].shape
        self.N, self.d = int(N), int(d)
        for t in self.tables:
            assert t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (N, d), "tables: contiguous float32 (N, d)"
        self.ldv = roundup(d, 4)
        assert self.ldv == d, "tables are passed with row stride d (d % 4 == 0)"
        self.ldx = lib.hwer_knn_padded_dim(d)
        if self.ldx < 0:
            raise native.HwerNativeError("BlendSearchStep: d=%d unsupported (1..%d)" % (d, lib.hwer_knn_max_dim()))
        self.alpha = alpha
        self.per_row = alpha.dim() == 2
        assert alpha.dtype == torch.float32 and alpha.is_contiguous() and tuple(alpha.shape) in ((T,), (N, T))
        self.search_ids, self.other_ids = search_ids, other_ids
        if search_ids is not None:
            n_search, search_row0 = int(search_ids.numel()), 0
            assert search_ids.dtype == torch.int64 and search_ids.is_contiguous()
        if other_ids is not None:
            n_other, other_row0 = int(other_ids.numel()), 0
            assert other_ids.dtype == torch.int64 and other_ids.is_contiguous()
        self.U, self.I = int(n_other), int(n_search)        # (bench vocabulary: U user rows, I item rows)
        self.search_row0, self.other_row0 = int(search_row0), int(other_row0)
        assert self.I >= 1 and self.search_row0 + self.I <= N and self.other_row0 + self.U <= N
        I, K = self.I, int(k)
        self.k = K
        self.anchors = anchors
        B = self.B = int(anchors.numel())
        assert anchors.dtype == torch.int32 and B >= 1
        self.eps = float(eps)
        with native.device_guard(dev):
            self.tab_ptrs = torch.tensor([t.data_ptr() for t in self.tables], dtype=torch.int64, device=dev)
            self.item_rows = search_ids if search_ids is not None else torch.arange(self.search_row0, self.search_row0 + I, dtype=torch.int64, device=dev)
            I_pad = self.I_pad = lib.hwer_knn_padded_rows(I)
            # The searched type's padded table X is a SLICE of V when that type's rows are the last rows of V and the padded width
            # equals the row stride (d = 32 / 64 / 128 / 256): V is allocated with the <= 31 pad rows behind it, and the fused prologue
            # stores each such row once (hwer_blend_prepare_search_f32 recognises X == V + row0 * ldv) - 22 % fewer bytes at d = 128.
            self.x_in_v = search_ids is None and self.search_row0 + I == N and self.ldv == self.ldx and not os.environ.get("HWER_NO_X_IN_V")
            self._Vbuf = torch.zeros((N + (I_pad - I if self.x_in_v else 0), self.ldv), dtype=torch.float32, device=dev)
            self.V = self._Vbuf[:N]
            self.audit_buf = torch.zeros(6, dtype=torch.int64, device=dev)     # counts[4] | sum |norm-1| (f64) | pad
            self.audit_counts = self.audit_buf[:4]
            self.audit_dev = self.audit_buf[4:5].view(torch.float64)
            self.X = self._Vbuf[self.search_row0:self.search_row0 + I_pad] if self.x_in_v else torch.empty((I_pad, self.ldx), dtype=torch.float32, device=dev)
            self.hnorm = torch.empty(I_pad, dtype=torch.float32, device=dev)
            # float16 screen wherever it pays (hwer_knn_f16_screen_pays: every streamed catalogue, dense ones from ~1e9 multiply-adds)
            if f16_screen is None:
                f16_screen = bool(lib.hwer_knn_f16_screen_pays(min(B, QUERY_CHUNK), I, d))
            # (both ways in: the float16 MFMA tiles start at a padded width of 64 - below that the library screens in float32)
            self.f16_screen = bool((f16_screen or (f16_dense and I <= 8192)) and lib.hwer_knn_padded_dim(d) >= 64)
            self.X16 = torch.empty((I_pad, self.ldx), dtype=torch.float16, device=dev) if self.f16_screen else None
            self.Q64 = torch.empty((B, d), dtype=torch.float64, device=dev)
            self.Q32 = torch.empty((B, self.ldx), dtype=torch.float32, device=dev)
            if not sort_by_score:
                want_dist = True
            self.want_dist = want_dist
            self.sort_by_score = 1 if sort_by_score else 0
            self.sets = [dict(idx=torch.empty((B, K), dtype=torch.int64, device=dev), score=torch.empty((B, K), dtype=torch.float64, device=dev),
                              dist2=torch.empty((B, K), dtype=torch.float64, device=dev) if want_dist else None,
                              flags=torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(n_sets)]
            self.CH = int(os.environ.get("HWER_QUERY_CHUNK", QUERY_CHUNK))     # (the override is for chunk-size experiments)
            self.opts = (KNN_F16_SCREEN if self.f16_screen else 0) | int(extra_options)
            ws_bytes = max(lib.hwer_knn_workspace_bytes_opts(min(B, self.CH), I, d, K, self.opts),
                           lib.hwer_knn_workspace_bytes_opts(1, I, d, K, 0), lib.hwer_knn_workspace_bytes_opts(min(B, 64), I, d, K, 0))
            if ws_bytes == 0:
                raise native.HwerNativeError("BlendSearchStep: unsupported shape (B=%d, n=%d, d=%d, k=%d; k <= %d)" % (B, I, d, K, lib.hwer_knn_max_k()))
            self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            native.check(lib.hwer_knn_workspace_init(native.ptr(self.ws), self.ws.numel(), self.S()), "knn_workspace_init")   # once per allocation
            self.ws_exact = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)
            self.missed = torch.zeros(1, dtype=torch.int32, device=dev)
            # More anchors than one search takes (ML-20M shape: 131 072 + 7 421).  tail_stream=True runs the short last search on a
            # second stream with its own workspaces, beside the first one (round 2: 6.02 -> 5.76 ms).  OFF by default since round 3: the
            # main screen is ONE round of 512 workgroups that need two 241-VGPR waves per SIMD, and any wave of the tail's kernels
            # sitting on a SIMD keeps the second one out - a workgroup that starts late still needs its full 2.1 ms, so the step was
            # bimodal (rocprofv3: the main screen 2.19 ... 3.97 ms; steps of 4.96 or 6.2 ms, with or without stream priorities).
            # One search after the other: 5.05 ms, every step.
            # Round 6, re-measured with the ping-pong screen of rounds 4-5 and the rebuilt select (profiles/r6_ab_c3_tail_stream.txt): in a process that runs
            # only this step the side-by-side form is 3 - 4 % faster than one search after the other and no longer bimodal (c3 over 300 steps, same box:
            # 4.29 - 4.35 ms, max 4.52, against 4.48, max 4.68) - but as the third workload of a process (bench.py's full run: c1 and c2 before it) the SAME
            # step is 8 % SLOWER than the sequential one (4.89 - 4.95 against 4.38 - 4.60 ms), whichever search is submitted first: how the runtime maps
            # streams onto hardware queues depends on the process's history.  It stays OFF by default: the sequential step costs the same everywhere.
            # Round 5 experiment, OFF by default: tail_under_select=True makes the side stream wait for the main chunk's screen launch
            # (hwer_knn_screen_f32 / hwer_knn_select_f32: the search as two stream-ordered calls) and runs the short last search beside
            # the main SELECT, whose workgroups are small (256 threads, 128 VGPRs).  Measured at c3 (profiles/r5_ab_c3.txt, rocprofv3
            # timeline): the tail's 512-thread / 169-VGPR screen workgroups are not dispatched while the select's 34 000 workgroups keep
            # every SIMD's register file full - the 158 us launch stretches to the select's whole 1.96 ms whatever the stream priorities -
            # and its list merge and select still run behind the main select: 4.81 -> 4.76 ms, inside the noise.
            self.tail_under_select = bool(tail_under_select) and B > self.CH and B % self.CH != 0 and not tail_stream and not pipeline
            self.tail = B % self.CH if (B > self.CH and (tail_stream or self.tail_under_select)) else 0
            self.overlap_tail = True
            if self.tail_under_select:
                # (high stream priority: the tail's few workgroups are dispatched as soon as slots free up instead of queueing behind the
                # main select's 34 000 - measured below)
                self.side = torch.cuda.Stream(device=dev, priority=int(os.environ.get("HWER_TAIL_PRIO", "-1")))
                self.ev_main_screen = torch.cuda.Event()
                self.ws_tail = torch.empty(lib.hwer_knn_workspace_bytes_opts(self.tail, I, d, K, self.opts), dtype=torch.uint8, device=dev)
                native.check(lib.hwer_knn_workspace_init(native.ptr(self.ws_tail), self.ws_tail.numel(), self.S()), "knn_workspace_init")
                self.ws_exact_tail = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)
            elif self.tail:
                # the tail search runs BESIDE the main one, at a lower stream priority: without it the two searches' kernels compete for
                # the CUs in whatever order the queues drain, and a c3 step was bimodal (4.96 or 6.2 ms: the main screen's 241-VGPR waves
                # waiting for register space behind the tail's list-merge waves)
                self.side = torch.cuda.Stream(device=dev, priority=0)
                self.main_hi = torch.cuda.Stream(device=dev, priority=-1)
                self.ws_tail = torch.empty(lib.hwer_knn_workspace_bytes_opts(self.tail, I, d, K, self.opts), dtype=torch.uint8, device=dev)
                native.check(lib.hwer_knn_workspace_init(native.ptr(self.ws_tail), self.ws_tail.numel(), self.S()), "knn_workspace_init")
                self.ws_exact_tail = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)
        # rows pass the eps unit-norm audit (checked by `check()` after the step) => 0.5||x||^2 <= 0.5 (1 + eps)^2
        self.hnorm_max = 0.5 * (1.0 + self.eps) ** 2
        self.x16_exp = int(lib.hwer_knn_f16_scale_exp(self.hnorm_max)) if self.f16_screen else 0
        self.id_base = 0 if search_ids is not None else (self.search_row0 if id_base is None else int(id_base))
        self.arith = int(arith)
        # Small catalogues (dense path): the prologue's two latency-floor kernels run as ONE launch that also writes the search's
        # query fragments (hwer_prepare_search_f32 + HWER_KNN_QUERIES_READY); one search call per step only.
        self.fused_prologue = (I <= 8192 and B <= self.CH) if fused_prologue is None else bool(fused_prologue)
        self.fused_fragments = self.fused_prologue     # (re-decided below once fused_blend is known)
        # ... and the blend + audit with them (hwer_blend_prepare_search_f32: the whole prologue of the step is one kernel).  Large
        # catalogues too: the per-type table and its float16 copy are written from the blended row in registers instead of being
        # re-read from V - one pass over the rows less.  NOT where the anchors outnumber the rows (every anchor row is blended a
        # second time for its query): that shape keeps the separate launches.
        self.fused_blend = (self.fused_prologue or (self.d <= 256 and 8 * B <= I)) if fused_blend is None else bool(fused_blend)
        # The search's query fragments come out of the prologue launch whenever the step is ONE search call behind a fused prologue
        # (float32 fragments, or - round 3 - the float16 ones with their per-query exponents: bq < 0 in the prologue call); the search
        # then skips its repacking launch (HWER_KNN_QUERIES_READY).
        self.fused_fragments = (self.fused_prologue or self.fused_blend) and B <= self.CH and not os.environ.get("HWER_NO_FUSED_FRAGMENTS")
        self.bq_frag = int(lib.hwer_knn_query_padding(B, I, d, K, self.opts)) if self.fused_fragments else 0
        if self.f16_screen:
            self.bq_frag = -self.bq_frag
        self._graph = None
        # pipeline=True: the search runs as chunks of PIPE_CHUNK queries whose two stages are staggered over two
        # streams - hwer_knn_screen_f32 of chunk i + 1 (MFMA / issue bound) on the step's stream beside hwer_knn_select_f32 of
        # chunk i (bound by the gather of the candidates' rows) on a side stream; two workspaces alternate.  Same kernels, same
        # results as the one-call search.
        # MEASURED (round 3, c3 = 138 493 anchors x 26 744 rows, same box): 6.97 ms pipelined vs 5.72 ms as one search + tail stream -
        # the select needs four waves per SIMD in flight to saturate the Infinity-Cache gather and finds one free slot beside a
        # resident screen, and 65 536-query chunks run the screen at TQ = 1 (twice the per-tile overhead per query).  Off by default.
        self.pipeline = bool(pipeline) and B > 1
        if self.pipeline:
            with native.device_guard(dev):
                self.tail = 0
                ch = min(PIPE_CHUNK, B)
                self.pipe_chunks = [(b0, min(B, b0 + ch)) for b0 in range(0, B, ch)]
                wsb = max(lib.hwer_knn_workspace_bytes_opts(b1 - b0, I, d, K, self.opts) for b0, b1 in self.pipe_chunks)
                self.pipe_ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in range(min(2, len(self.pipe_chunks)))]
                for w_ in self.pipe_ws:
                    native.check(lib.hwer_knn_workspace_init(native.ptr(w_), w_.numel(), self.S()), "knn_workspace_init")
                self.pipe_side = torch.cuda.Stream(device=dev)
                self.pipe_ev_screen = [torch.cuda.Event() for _ in self.pipe_chunks]
                self.pipe_ev_select = [torch.cuda.Event() for _ in self.pipe_chunks]

    # ---- stages of a step (S() = the current torch stream of the step's device) -----------------------------------------------
    def S(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def stage_blend(self):
        P, lib = native.ptr, self.lib
        native.check(lib.hwer_blend_normalize_audit_f32(P(self.tab_ptrs), self.d, None, P(self.alpha), 1 if self.per_row else 0, self.T, self.N,
                                                        self.d, P(self.V), self.ldv, self.eps, P(self.audit_counts), P(self.audit_dev), self.S()), "blend+audit")

    def stage_pack(self):
        P, lib = native.ptr, self.lib
        # per-type table + half norms (+ the float16 copy in the same pass: its scale follows from the unit-norm audit bound)
        native.check(lib.hwer_pack_rows_f32_f16(P(self.V), self.N, self.d, self.ldv, P(self.item_rows), self.I, P(self.X), self.ldx, P(self.hnorm),
                                                P(self.X16) if self.f16_screen else None, self.x16_exp, self.S()), "pack")

    def stage_queries(self):
        P, lib = native.ptr, self.lib
        native.check(lib.hwer_query_mix(P(self.V), self.N, self.d, self.ldv, P(self.anchors), None, None, None, None, self.B, self.arith,
                                        P(self.Q64), P(self.Q32), self.ldx, self.S()), "query_mix")

    def stage_search(self, o, B=None, opts=None):
        """exact top-k of queries 0..B-1 (default: all) into result set `o`; uncertified queries (rare: near-ties at the cut, or a
        sampled threshold that was too tight) are recomputed by the exhaustive float64 pass INSIDE the step, on the device (no host
        round trip): up to 8 per search call; anything beyond is counted in `self.missed` and reported by check()."""
        P, lib, K = native.ptr, self.lib, self.k
        U = self.B if B is None else B
        opts = self.opts if opts is None else opts
        oi, osc, od, of = o["idx"], o["score"], o["dist2"], o["flags"]

        def search(b0, b1, ws, ws_exact):   # on the CURRENT torch stream
            native.check(lib.hwer_knn_search_opts_f32(
                P(self.X), P(self.hnorm), self.I, self.d, self.ldx, P(self.Q32[b0:b1]), P(self.Q64[b0:b1]), P(self.V), self.ldv,
                P(self.anchors[b0:b1]), b1 - b0, K, self.arith, self.sort_by_score, P(self.search_ids), self.id_base, self.hnorm_max, P(oi[b0:b1]),
                P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None, P(of[b0:b1]), P(ws), ws.numel(), opts,
                P(self.X16) if self.f16_screen else None, self.x16_exp, self.S()), "knn_search")
            native.check(lib.hwer_knn_exact_device(
                P(self.X), self.I, self.d, self.ldx, P(self.Q64[b0:b1]), P(self.V), self.ldv, P(self.anchors[b0:b1]), K, self.arith,
                self.sort_by_score, P(self.search_ids), self.id_base, P(oi[b0:b1]), P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None, P(of[b0:b1]),
                P(ws), P(self.missed), P(ws_exact), ws_exact.numel(), self.S()), "knn_exact_device")
        if self.pipeline and U == self.B and self.overlap_tail:
            return self._search_pipelined(o, opts)
        if self.tail_under_select and U == self.B and self.overlap_tail:
            torch = self.torch
            cur = torch.cuda.current_stream(self.dev)
            x16 = P(self.X16) if self.f16_screen else None
            n_main = U - self.tail
            for b0 in range(0, n_main, self.CH):
                b1 = min(n_main, b0 + self.CH)
                native.check(lib.hwer_knn_screen_f32(P(self.X), P(self.hnorm), self.I, self.d, self.ldx, P(self.Q32[b0:b1]), b1 - b0, K, P(self.ws), self.ws.numel(),
                                                     opts & ~KNN_QUERIES_READY, x16, self.x16_exp, self.S()), "knn_screen")
                if b1 == n_main:      # the last full chunk's screen is on its way: the tail search goes beside its select
                    self.ev_main_screen.record(cur)
                    self.side.wait_event(self.ev_main_screen)
                    with torch.cuda.stream(self.side):
                        search(n_main, U, self.ws_tail, self.ws_exact_tail)
                native.check(lib.hwer_knn_select_f32(
                    P(self.X), self.I, self.d, self.ldx, P(self.Q64[b0:b1]), P(self.V), self.ldv, P(self.anchors[b0:b1]), b1 - b0, K, self.arith,
                    self.sort_by_score, P(self.search_ids), self.id_base, self.hnorm_max, P(oi[b0:b1]), P(osc[b0:b1]),
                    P(od[b0:b1]) if od is not None else None, P(of[b0:b1]), P(self.ws), self.ws.numel(), opts & ~KNN_QUERIES_READY, x16, self.S()), "knn_select")
                native.check(lib.hwer_knn_exact_device(
                    P(self.X), self.I, self.d, self.ldx, P(self.Q64[b0:b1]), P(self.V), self.ldv, P(self.anchors[b0:b1]), K, self.arith,
                    self.sort_by_score, P(self.search_ids), self.id_base, P(oi[b0:b1]), P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None, P(of[b0:b1]),
                    P(self.ws), P(self.missed), P(self.ws_exact), self.ws_exact.numel(), self.S()), "knn_exact_device")
            cur.wait_stream(self.side)
            return
        side_tail = self.tail and not self.tail_under_select and U == self.B and self.overlap_tail
        if side_tail:
            # the main chunks on the high-priority stream, SUBMITTED FIRST (round 6: when the runtime maps both streams onto one hardware queue - it does
            # in a process that has used several streams before, e.g. the bench's full run - the searches execute in submission order, and the short one
            # in front of the long one cost 9 %: c3 4.93 against 4.53 ms; main first, that case is the sequential step), the short last search beside them
            torch = self.torch
            cur = torch.cuda.current_stream(self.dev)
            self.main_hi.wait_stream(cur)
            with torch.cuda.stream(self.main_hi):
                for b0 in range(0, U - self.tail, self.CH):
                    search(b0, min(U, b0 + self.CH), self.ws, self.ws_exact)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                search(U - self.tail, U, self.ws_tail, self.ws_exact_tail)
            cur.wait_stream(self.main_hi)
            cur.wait_stream(self.side)
        else:
            for b0 in range(0, U, self.CH):
                search(b0, min(U, b0 + self.CH), self.ws, self.ws_exact)

    def _search_pipelined(self, o, opts):
        P, lib, K, torch = native.ptr, self.lib, self.k, self.torch
        oi, osc, od, of = o["idx"], o["score"], o["dist2"], o["flags"]
        x16 = P(self.X16) if self.f16_screen else None
        cur = torch.cuda.current_stream(self.dev)
        for i, (b0, b1) in enumerate(self.pipe_chunks):
            ws = self.pipe_ws[i % len(self.pipe_ws)]
            if i >= len(self.pipe_ws):
                cur.wait_event(self.pipe_ev_select[i - len(self.pipe_ws)])     # the select that read this workspace is done
            native.check(lib.hwer_knn_screen_f32(P(self.X), P(self.hnorm), self.I, self.d, self.ldx, P(self.Q32[b0:b1]), b1 - b0, K, P(ws), ws.numel(),
                                                 opts & ~KNN_QUERIES_READY, x16, self.x16_exp, self.S()), "knn_screen")
            self.pipe_ev_screen[i].record(cur)
            with torch.cuda.stream(self.pipe_side):
                self.pipe_side.wait_event(self.pipe_ev_screen[i])
                st = ctypes.c_void_p(self.pipe_side.cuda_stream)
                native.check(lib.hwer_knn_select_f32(
                    P(self.X), self.I, self.d, self.ldx, P(self.Q64[b0:b1]), P(self.V), self.ldv, P(self.anchors[b0:b1]), b1 - b0, K, self.arith,
                    self.sort_by_score, P(self.search_ids), self.id_base, self.hnorm_max, P(oi[b0:b1]), P(osc[b0:b1]),
                    P(od[b0:b1]) if od is not None else None, P(of[b0:b1]), P(ws), ws.numel(), opts & ~KNN_QUERIES_READY, x16, st), "knn_select")
                native.check(lib.hwer_knn_exact_device(
                    P(self.X), self.I, self.d, self.ldx, P(self.Q64[b0:b1]), P(self.V), self.ldv, P(self.anchors[b0:b1]), K, self.arith,
                    self.sort_by_score, P(self.search_ids), self.id_base, P(oi[b0:b1]), P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None,
                    P(of[b0:b1]), P(ws), P(self.missed), P(self.ws_exact), self.ws_exact.numel(), st), "knn_exact_device")
                self.pipe_ev_select[i].record(self.pipe_side)
        cur.wait_stream(self.pipe_side)

    def stage_prepare(self):
        """per-type table + queries (+ the search's float32 query fragments) in ONE launch: hwer_prepare_search_f32"""
        P, lib = native.ptr, self.lib
        native.check(lib.hwer_prepare_search_f32(
            P(self.V), self.N, self.d, self.ldv, P(self.item_rows), self.I, P(self.X), self.ldx, P(self.hnorm),
            P(self.X16) if self.f16_screen else None, self.x16_exp, P(self.anchors), None, None, None, None, self.B, self.arith,
            P(self.Q64), P(self.Q32), self.ldx, P(self.ws) if self.fused_fragments else None, self.bq_frag, self.S()), "prepare_search")

    def stage_blend_prepare(self):
        """blend + audit + per-type table + queries (+ query fragments) in ONE launch: hwer_blend_prepare_search_f32"""
        P, lib = native.ptr, self.lib
        native.check(lib.hwer_blend_prepare_search_f32(
            P(self.tab_ptrs), self.d, None, P(self.alpha), 1 if self.per_row else 0, self.T, self.d, P(self.V), self.ldv, self.eps,
            P(self.audit_counts), P(self.audit_dev), P(self.search_ids), self.search_row0, self.I, P(self.other_ids), self.other_row0, self.U, P(self.X), self.ldx, P(self.hnorm),
            P(self.X16) if self.f16_screen else None, self.x16_exp, P(self.anchors), self.B, self.arith, P(self.Q64), P(self.Q32), self.ldx,
            P(self.ws), self.bq_frag, self.S()), "blend_prepare_search")

    def step(self, o=None, mark=None):
        """launch one whole step on the current stream into result set `o` (default: set 0); returns the result set.
        mark: optional torch.cuda.Event recorded between the prologue and the search (measurement: the two parts of a step IN the step)"""
        o = self.sets[0] if o is None else o
        if self.fused_blend:          # the whole prologue is one launch
            self.stage_blend_prepare()
        elif self.fused_prologue:     # blend, then the fused table + queries launch
            self.stage_blend()
            self.stage_prepare()
        else:
            self.stage_blend()
            self.stage_pack()
            self.stage_queries()
        if mark is not None:
            mark.record()
        self.stage_search(o, opts=self.opts | (KNN_QUERIES_READY if (self.fused_fragments and (self.fused_blend or self.fused_prologue)) else 0))
        return o

    # ---- library-facing helpers -------------------------------------------------------------------------------------------------
    def run(self, o=None):
        """one step with the step's device current; replays the captured hipGraph when capture() was called (set 0 only)"""
        with native.device_guard(self.dev):
            if self._graph is not None and o is None:
                self._graph.replay()
                return self.sets[0]
            return self.step(o)

    def capture(self, warm_steps: int = 2):
        """capture the step into a hipGraph (small, launch-bound shapes: 4-6 launches of a few microseconds each); run() then
        replays it.  Every buffer is preallocated and nothing in the step synchronises, so the capture is legal."""
        torch = self.torch
        with native.device_guard(self.dev):
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(warm_steps):
                    self.step()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self._graph = g
        return self

    def audit(self):
        """(violations, mean |len - 1|, positive, negative, nan_rows) of the blended rows - utils.unit_length_violations
        (hwer/utils.py:51-57) as RecommendationBase.__build_knn__ uses it; synchronises."""
        pos, neg, nan = (int(x) for x in self.audit_counts.tolist()[:3])
        return pos + neg, float(self.audit_dev.item()) / (self.I + self.U), pos, neg, nan

    def checks(self, o=None):
        """host-side verdict on a finished step (synchronises): the unit-norm audit passed (the reference asserts it,
        hwer/recommendation_base.py:106-107), every query has an exact result.  -> queries that took the exhaustive pass"""
        o = self.sets[0] if o is None else o
        pos, neg, nan = self.audit_counts.tolist()[:3]
        assert pos + neg + nan == 0, "unit-norm audit failed: %s" % ((pos, neg, nan),)
        n_uncert = int((o["flags"] == 0).sum().item())
        assert n_uncert == 0, "%d queries left without an exact result" % n_uncert
        assert int(self.missed.item()) == 0, "the on-device exhaustive pass left %d queries to the host path" % int(self.missed.item())
        return int((o["flags"] == 2).sum().item())

    def repair(self, o=None):
        """Host-driven last tier (synchronises; NOT part of the resident step): queries a step left uncertified beyond what its on-device
        exhaustive pass takes (8 per search call) are recomputed exhaustively in float64 - hwer_knn_exact_f64, the tier DeviceIndex runs
        behind every search.  -> the number of repaired queries (0: nothing to do)."""
        import numpy as np
        torch, lib, P = self.torch, self.lib, native.ptr
        o = self.sets[0] if o is None else o
        with native.device_guard(self.dev):
            bad = torch.nonzero(o["flags"] == 0).flatten()
            if bad.numel() == 0:
                return 0
            qsel = np.ascontiguousarray(bad.cpu().numpy(), dtype=np.int32)
            ws = torch.empty(lib.hwer_knn_exact_workspace_bytes(self.I, self.k), dtype=torch.uint8, device=self.dev)
            od = o["dist2"]
            native.check(lib.hwer_knn_exact_f64(P(self.X), self.I, self.d, self.ldx, P(self.Q64), P(self.V), self.ldv, P(self.anchors),
                                                qsel.ctypes.data_as(ctypes.c_void_p), len(qsel), self.k, self.arith, self.sort_by_score, P(self.search_ids),
                                                self.id_base, P(o["idx"]), P(o["score"]), P(od) if od is not None else None, P(o["flags"]), P(ws), ws.numel(),
                                                self.S()), "knn_exact")
            self.missed.zero_()
            torch.cuda.synchronize(self.dev)
        return int(len(qsel))

    def results(self, o=None):
        """-> numpy (idx (B, k) int64 ids, score (B, k) in the reference's dtype [, dist (B, k) float64]) after checks()"""
        import numpy as np
        o = self.sets[0] if o is None else o
        self.checks(o)
        sc = o["score"].cpu().numpy()
        if self.arith == 1:
            sc = sc.astype(np.float32)      # exact: the kernel widened float32 values
        out = (o["idx"].cpu().numpy(), sc)
        return out + (np.sqrt(o["dist2"].cpu().numpy()),) if o["dist2"] is not None else out


class BlendSearchStepF64:
    """The same resident step in the reference's own float64 arithmetic (round 6): T FLOAT64 tables -> hwer_blend_normalize_f64 (the bits of
    hwer/content_recommender.py:85,96 / hwer/utils.py:43-44) -> float64 `vectors` V64 -> unit-norm audit of the float64 rows + the float32 / float16
    rounding of the searched type's rows for the MFMA screen (hwer_pack_rows_f64rows; hwer/recommendation_base.py:105-110, :64-76) -> anchor queries
    from the float64 rows (hwer_query_mix_f64rows; :153-155,164-170) -> hwer_knn_search_f64rows + hwer_knn_exact_device_f64rows: exact top-k whose
    distances and scores are computed from the float64 rows (:74,78-83,141-143,171-173) - what ContentRecommendation serves, as one
    allocation-free, capturable step.  tables: (N, d) float64 device tensors with row stride ld (even, >= d); alpha (T,) or (N, T) float64.
    Rows [search_row0, + n_search) are the searched type, ids reported = id_base + local row (default: the row id)."""

    def __init__(self, tables: Sequence, alpha, search_row0: int, n_search: int, anchors, k: int = 200, id_base: Optional[int] = None,
                 want_dist: bool = False, sort_by_score: bool = True, f16_screen: Optional[bool] = None, eps: float = 1e-4, score_mode: int = 0):
        import torch
        lib = self.lib = native.lib()
        self.torch = torch
        self.tables = list(tables)
        T = self.T = len(self.tables)
        dev = self.dev = self.tables[0].device
        N, d = self.tables[0].shape
        self.N, self.d = int(N), int(d)
        self.ld_in = int(self.tables[0].stride(0))
        for t in self.tables:
            assert t.dtype == torch.float64 and tuple(t.shape) == (N, d) and t.stride(1) == 1 and t.stride(0) == self.ld_in and self.ld_in % 2 == 0
        self.ldx = lib.hwer_knn_padded_dim(d)
        if self.ldx < 0:
            raise native.HwerNativeError("BlendSearchStepF64: d=%d unsupported (1..%d)" % (d, lib.hwer_knn_max_dim()))
        self.alpha, self.per_row = alpha, alpha.dim() == 2
        assert alpha.dtype == torch.float64 and alpha.is_contiguous() and tuple(alpha.shape) in ((T,), (N, T))
        self.I, self.search_row0 = int(n_search), int(search_row0)
        assert self.I >= 1 and self.search_row0 + self.I <= N
        I, K = self.I, int(k)
        self.k, self.anchors = K, anchors
        B = self.B = int(anchors.numel())
        assert anchors.dtype == torch.int32 and B >= 1
        self.eps, self.score_mode = float(eps), int(score_mode)
        with native.device_guard(dev):
            self.tab_ptrs = torch.tensor([t.data_ptr() for t in self.tables], dtype=torch.int64, device=dev)
            self.ld64 = roundup(d, 8)
            self.V64 = torch.zeros((N, self.ld64), dtype=torch.float64, device=dev)      # (pad columns stay zero: the blend writes d of them)
            # the searched type's rows are a RANGE of the float64 table: the search addresses them as X64 = V64 + search_row0 * ld64 with no row map
            # (a row map costs one dependent load per gathered candidate row)
            self.X64 = self.V64[self.search_row0:self.search_row0 + I]
            I_pad = self.I_pad = lib.hwer_knn_padded_rows(I)
            self.X = torch.empty((I_pad, self.ldx), dtype=torch.float32, device=dev)
            self.hnorm = torch.empty(I_pad, dtype=torch.float32, device=dev)
            if f16_screen is None:
                f16_screen = bool(lib.hwer_knn_f16_screen_pays(min(B, QUERY_CHUNK), I, d))
            self.f16_screen = bool(f16_screen and self.ldx >= 64)
            self.X16 = torch.empty((I_pad, self.ldx), dtype=torch.float16, device=dev) if self.f16_screen else None
            self.audit_buf = torch.zeros(6, dtype=torch.int64, device=dev)
            self.audit_counts, self.audit_dev = self.audit_buf[:4], self.audit_buf[4:5].view(torch.float64)
            self.Q64 = torch.empty((B, d), dtype=torch.float64, device=dev)
            self.Q32 = torch.empty((B, self.ldx), dtype=torch.float32, device=dev)
            if not sort_by_score or score_mode == 2:
                want_dist = True
            self.want_dist, self.sort_by_score = want_dist, 1 if sort_by_score else 0
            self.sets = [dict(idx=torch.empty((B, K), dtype=torch.int64, device=dev), score=torch.empty((B, K), dtype=torch.float64, device=dev),
                              dist2=torch.empty((B, K), dtype=torch.float64, device=dev) if want_dist else None, flags=torch.empty(B, dtype=torch.int32, device=dev))]
            self.CH = QUERY_CHUNK
            self.opts = KNN_F16_SCREEN if self.f16_screen else 0
            ws_bytes = lib.hwer_knn_workspace_bytes_opts(min(B, self.CH), I, d, K, self.opts)
            if ws_bytes == 0:
                raise native.HwerNativeError("BlendSearchStepF64: unsupported shape (B=%d, n=%d, d=%d, k=%d; k <= %d)" % (B, I, d, K, lib.hwer_knn_max_k()))
            self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            native.check(lib.hwer_knn_workspace_init(native.ptr(self.ws), self.ws.numel(), self.S()), "knn_workspace_init")
            self.ws_exact = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)
            self.missed = torch.zeros(1, dtype=torch.int32, device=dev)
            # more anchors than one search takes (ML-20M shape: 131 072 + 7 421): HWER_TAIL_STREAM=1 runs the short last search on a second stream, beside the
            # main one, with its own workspaces (BlendSearchStep's tail_stream; c3_f64rows 6.70 -> 6.66 ms alone, 7.17 -> 7.02 in the bench's full run)
            self.tail = B % self.CH if (B > self.CH and B % self.CH != 0 and os.environ.get("HWER_TAIL_STREAM")) else 0     # (opt-in: BlendSearchStep's tail_stream note)
            if self.tail:
                self.side = torch.cuda.Stream(device=dev, priority=0)
                self.main_hi = torch.cuda.Stream(device=dev, priority=-1)
                self.ws_tail = torch.empty(lib.hwer_knn_workspace_bytes_opts(self.tail, I, d, K, self.opts), dtype=torch.uint8, device=dev)
                native.check(lib.hwer_knn_workspace_init(native.ptr(self.ws_tail), self.ws_tail.numel(), self.S()), "knn_workspace_init")
                self.ws_exact_tail = torch.empty(lib.hwer_knn_exact_device_workspace_bytes(I, K), dtype=torch.uint8, device=dev)
        self.hnorm_max = 0.5 * (1.0 + self.eps) ** 2
        self.x16_exp = int(lib.hwer_knn_f16_scale_exp(self.hnorm_max)) if self.f16_screen else 0
        self.id_base = self.search_row0 if id_base is None else int(id_base)
        # one search call per step: its query fragments come out of the query launch (hwer_query_mix_frag_f64rows + HWER_KNN_QUERIES_READY)
        self.fused_fragments = B <= self.CH and not os.environ.get("HWER_NO_FUSED_FRAGMENTS")
        self.bq_frag = int(lib.hwer_knn_query_padding(B, I, d, K, self.opts)) * (-1 if self.f16_screen else 1) if self.fused_fragments else 0
        self._graph = None

    def S(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def stage_blend(self):
        P, lib = native.ptr, self.lib
        native.check(lib.hwer_blend_normalize_f64(P(self.tab_ptrs), self.ld_in, None, P(self.alpha), 1 if self.per_row else 0, self.T, self.N, self.d,
                                                  P(self.V64), self.ld64, None, self.S()), "blend_normalize_f64")

    def stage_pack(self):
        """unit-norm audit of every float64 row + the screen's float32 / float16 rounding of the searched type's rows"""
        P, lib = native.ptr, self.lib
        self.audit_buf.zero_()
        # ONE launch: every row audited, the searched range also rounded into X / hnorm / X16 (hwer_audit_pack_rows_f64rows)
        native.check(lib.hwer_audit_pack_rows_f64rows(P(self.V64), self.N, self.d, self.ld64, self.search_row0, self.I, P(self.X), self.ldx, P(self.hnorm),
                                                      P(self.X16), self.x16_exp, self.eps, P(self.audit_counts), P(self.audit_dev), self.S()), "audit_pack_rows_f64rows")

    def stage_queries(self):
        P, lib = native.ptr, self.lib
        if self.fused_fragments:
            native.check(lib.hwer_query_mix_frag_f64rows(P(self.V64), self.N, self.d, self.ld64, P(self.anchors), None, None, None, None, self.B, P(self.Q64), P(self.Q32),
                                                         self.ldx, P(self.ws), self.bq_frag, self.S()), "query_mix_frag_f64rows")
            return
        native.check(lib.hwer_query_mix_f64rows(P(self.V64), self.N, self.d, self.ld64, P(self.anchors), None, None, None, None, self.B, P(self.Q64), P(self.Q32),
                                                self.ldx, self.S()), "query_mix_f64rows")

    def stage_search(self, o=None):
        P, lib, K = native.ptr, self.lib, self.k
        o = self.sets[0] if o is None else o
        oi, osc, od, of = o["idx"], o["score"], o["dist2"], o["flags"]
        A64 = P(self.V64) if self.score_mode != 2 else None

        def search(b0, b1, ws, ws_exact):   # on the CURRENT torch stream
            native.check(lib.hwer_knn_search_f64rows(
                P(self.X), P(self.hnorm), self.I, self.d, self.ldx, P(self.Q32[b0:b1]), P(self.Q64[b0:b1]), P(self.X64), self.ld64, None, A64,
                self.ld64, P(self.anchors[b0:b1]), b1 - b0, K, self.score_mode, self.sort_by_score, None, self.id_base, self.hnorm_max, P(oi[b0:b1]),
                P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None, P(of[b0:b1]), P(ws), ws.numel(),
                self.opts | (KNN_QUERIES_READY if self.fused_fragments else 0), P(self.X16), self.x16_exp, self.S()), "knn_search_f64rows")
            native.check(lib.hwer_knn_exact_device_f64rows(
                P(self.X64), self.ld64, None, self.I, self.d, P(self.Q64[b0:b1]), A64, self.ld64, P(self.anchors[b0:b1]), K, self.score_mode,
                self.sort_by_score, None, self.id_base, P(oi[b0:b1]), P(osc[b0:b1]), P(od[b0:b1]) if od is not None else None, P(of[b0:b1]), P(ws),
                P(self.missed), P(ws_exact), ws_exact.numel(), self.S()), "knn_exact_device_f64rows")

        if self.tail:   # the short last search on the second stream, the main chunks beside it on the high-priority one
            torch = self.torch
            cur = torch.cuda.current_stream(self.dev)
            self.main_hi.wait_stream(cur)      # (main first: BlendSearchStep.stage_search)
            with torch.cuda.stream(self.main_hi):
                for b0 in range(0, self.B - self.tail, self.CH):
                    search(b0, min(self.B - self.tail, b0 + self.CH), self.ws, self.ws_exact)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                search(self.B - self.tail, self.B, self.ws_tail, self.ws_exact_tail)
            cur.wait_stream(self.main_hi)
            cur.wait_stream(self.side)
            return
        for b0 in range(0, self.B, self.CH):
            search(b0, min(self.B, b0 + self.CH), self.ws, self.ws_exact)

    def step(self, o=None, mark=None):
        self.stage_blend()
        self.stage_pack()
        self.stage_queries()
        if mark is not None:
            mark.record()
        self.stage_search(o)
        return self.sets[0] if o is None else o

    def run(self, o=None):
        with native.device_guard(self.dev):
            if self._graph is not None and o is None:
                self._graph.replay()
                return self.sets[0]
            return self.step(o)

    def capture(self, warm_steps: int = 2):
        torch = self.torch
        with native.device_guard(self.dev):
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(warm_steps):
                    self.step()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self._graph = g
        return self

    def checks(self, o=None):
        o = self.sets[0] if o is None else o
        pos, neg, nan = self.audit_counts.tolist()[:3]
        assert pos + neg + nan == 0, "unit-norm audit failed: %s" % ((pos, neg, nan),)
        n_uncert = int((o["flags"] == 0).sum().item())
        assert n_uncert == 0, "%d queries left without an exact result" % n_uncert
        assert int(self.missed.item()) == 0, "the on-device exhaustive pass left %d queries to the host path" % int(self.missed.item())
        return int((o["flags"] == 2).sum().item())

    def results(self, o=None):
        import numpy as np
        o = self.sets[0] if o is None else o
        self.checks(o)
        out = (o["idx"].cpu().numpy(), o["score"].cpu().numpy())
        return out + (np.sqrt(o["dist2"].cpu().numpy()),) if o["dist2"] is not None else out
