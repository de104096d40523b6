"""Device-resident index of the retrieval hot path (array level, no Node objects).

HBM layout (all float32, row-major):
  V      (N, ldv)      the `vectors` table, ldv = roundup4(d), zero padded         — pair scoring, anchors, queries
  X_t    (N_t, ldx)    per node type copy of its rows, ldx = 32/64/128/256         — MFMA screen + re-rank
  hnorm_t(N_t)         0.5 * ||x||^2                                               — screen key offset
  rows_t (N_t) int64   local -> global row id (ascending insertion order)          — K6
Replaces MultiKNN's per-type KDTree copies (reference hwer/recommendation_base.py:64-76).

PyTorch is used for device memory and streams only; every computation is a call into libhwer_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence

import os

import numpy as np

from . import native


def _torch():
    import torch
    return torch


def _require_gpu(device):
    torch = _torch()
    if not torch.cuda.is_available():
        raise native.HwerNativeError("hwer (MI355X build) needs a ROCm GPU: torch.cuda.is_available() is False; "
                                     "there is no CPU fallback for the retrieval path")
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())


def roundup(x, m):
    return (x + m - 1) // m * m


def _on_device(fn):
    """Run a DeviceIndex method with the index's device current (native launches go to the CURRENT HIP device and to its
    current torch stream - an index on cuda:1 used while cuda:0 is current would otherwise launch on the wrong GPU).
    Re-entrant calls skip the guard when the device IS current (a query of the runtime, not a flag on the instance: a second
    thread entering while the first is inside must not inherit the first one's guard)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        torch = _torch()
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *a, **kw)
        with native.device_guard(self.device):
            return fn(self, *a, **kw)
    return wrapped


class _TypeTable:
    __slots__ = ("rows", "rows_dev", "X", "hnorm", "hnorm_max", "n", "id_base", "X16", "x16_exp", "row0")

    def __init__(self):
        self.rows = None
        self.rows_dev = None
        self.X = None
        self.hnorm = None
        self.hnorm_max = 0.0
        self.n = 0
        self.id_base = 0
        self.X16 = None      # float16 storage of X for the HBM-bound small-batch screen (built on first use)
        self.x16_exp = 0
        self.row0 = -1       # >= 0: the type's rows are the RANGE [row0, row0 + n) of the vector table (float64 rows: addressed without the row map)


class _OneAnchorSearch:
    """The reference calls find_closest_neighbours once per anchor (hwer/validation.py:30-35; hwer/recommendation_base.py:157-174): one
    anchor -> query -> k nearest of one node type -> scores.  Per (node type, k, score mode) this object keeps every buffer of that call
    resident and the whole device chain - anchor id H2D, hwer_query_mix, hwer_knn_search_opts_f32, hwer_knn_exact_device, results D2H -
    captured ONCE as a hipGraph: a call writes the anchor id into pinned host memory, replays the graph, waits for the stream and reads
    k ids + k scores + the flag from pinned host memory.  No allocation, no per-call stream / pointer plumbing in Python."""

    def __init__(self, ix, node_type, k, mode, sort_by_score=True):
        import threading
        torch = _torch()
        lib, dev = ix.lib, ix.device
        tt = ix.tables[node_type]
        self.k, self.mode = int(k), int(mode)
        # ONE pinned anchor word, ONE pinned result buffer and ONE workspace per cached graph: a call owns them from the anchor
        # write to the copy-out of its results (run() holds this lock; a second thread calling search() with one anchor waits)
        self.lock = threading.Lock()
        self.h_anchor = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.h_anchor_np = self.h_anchor.numpy()
        self.d_anchor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.Q64 = torch.empty((1, ix.d), dtype=torch.float64, device=dev)
        self.Q32 = torch.empty((1, ix.ldx), dtype=torch.float32, device=dev)
        # one result buffer, one D2H copy: [k ids | k score bit patterns | flag | GcnNCF flavour: k squared-distance bit patterns]
        # (the GcnNCF score is a function of the distance and the general path returns the distances with it: so does this one)
        nw = 2 * k + 1 + (k if mode == 2 else 0)
        self.buf = torch.zeros(nw, dtype=torch.int64, device=dev)
        self.h_buf = torch.zeros(nw, dtype=torch.int64).pin_memory()
        self.h_np = self.h_buf.numpy()
        idx = self.buf[:k].view(1, k)
        score = self.buf[k:2 * k].view(torch.float64).view(1, k)
        flags = self.buf[2 * k:2 * k + 1].view(torch.int32)[:1]
        d2 = self.buf[2 * k + 1:].view(torch.float64).view(1, k) if mode == 2 else None
        path_opt = ix._path_options()
        x16, x16_exp = ix._x16(tt, 1)
        opt = (ix.KNN_F16_SCREEN if (ix.f16_screen and x16 is not None) else 0) | path_opt
        self.opt = opt
        wsb = lib.hwer_knn_workspace_bytes_opts(1, tt.n, ix.d, k, opt)
        if wsb == 0:
            raise native.HwerNativeError("knn: unsupported shape (n=%d, d=%d, k=%d)" % (tt.n, ix.d, k))
        self.ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
        self.wse = torch.empty(int(lib.hwer_knn_exact_device_workspace_bytes(tt.n, k)), dtype=torch.uint8, device=dev)
        P = native.ptr
        A32 = P(ix.V) if mode != 2 else None
        use_rows = True

        def launch():      # on the CURRENT torch stream of the index's device
            st = native.stream_ptr(dev)
            self.d_anchor.copy_(self.h_anchor, non_blocking=True)
            native.check(lib.hwer_query_mix(P(ix.V), ix.N, ix.d, ix.ldv, P(self.d_anchor), None, None, None, None, 1, ix.arith, P(self.Q64), P(self.Q32),
                                            ix.ldx, st), "query_mix")
            native.check(lib.hwer_knn_search_opts_f32(
                P(tt.X), P(tt.hnorm), tt.n, ix.d, ix.ldx, P(self.Q32), P(self.Q64), A32, ix.ldv, P(self.d_anchor), 1, k, mode, 1 if sort_by_score else 0,
                P(tt.rows_dev) if use_rows else None, 0, tt.hnorm_max, P(idx), P(score), P(d2), P(flags), P(self.ws), self.ws.numel(), opt, P(x16), x16_exp, st),
                "knn_search")
            native.check(lib.hwer_knn_exact_device(
                P(tt.X), tt.n, ix.d, ix.ldx, P(self.Q64), A32, ix.ldv, P(self.d_anchor), k, mode, 1 if sort_by_score else 0, P(tt.rows_dev), 0, P(idx), P(score),
                P(d2), P(flags), P(self.ws), None, P(self.wse), self.wse.numel(), st), "knn_exact_device")
            self.h_buf.copy_(self.buf, non_blocking=True)
        with native.device_guard(dev):
            native.check(lib.hwer_knn_workspace_init(P(self.ws), self.ws.numel(), native.stream_ptr(dev)), "knn_workspace_init")
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    launch()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launch()
            self.graph = g
        self._keep = (tt, x16, ix.V)      # the captured launches hold raw pointers: keep their tensors alive as long as the graph
        self.dev = dev
        self.dev_index = dev.index
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._sync = lib.hwer_stream_synchronize

    def run(self, anchor: int):
        """-> (ids (k,) int64, scores (k,) float64, flag, squared distances (k,) float64 or None) as fresh numpy arrays.  Thread-safe: the pinned anchor word, the replay and
        the pinned result buffer belong to one caller at a time."""
        with self.lock:
            self.h_anchor_np[0] = anchor
            self.graph.replay()          # on the current stream of the index's device
            raw = self._raw_stream
            if raw is not None:          # the raw hipStream_t of that stream (one C call) -> hwer_stream_synchronize
                rc = self._sync(raw(self.dev_index))
                if rc:
                    native.check(rc, "stream_synchronize")
            else:
                _torch().cuda.current_stream(self.dev).synchronize()
            k = self.k
            h = self.h_np
            d2 = h[2 * k + 1:].view(np.float64).copy() if self.mode == 2 else None
            return h[:k].copy(), h[k:2 * k].view(np.float64).copy(), int(h[2 * k] & 0xFFFFFFFF), d2


class _OneAnchorSearchF64:
    """The per-anchor call on GENUINE float64 rows (what ContentRecommendation.fit returns since round 5's float64 blend stage): anchor id H2D ->
    hwer_query_mix_f64rows -> hwer_knn_search_f64rows (MFMA screen on the float32 rounding, fused select on the float64 rows) ->
    hwer_knn_exact_device_f64rows -> ids / scores / flag D2H, captured ONCE as a hipGraph per (node type, k, score mode) - the float32 flavour's
    launch chain (round 5 captured the 12-launch candidate scheme here).  A replay that comes back uncertified beyond the on-device pass (flag 0)
    is answered by the general, synchronising path instead."""

    def __init__(self, ix, node_type, k, mode=0):
        import threading
        torch = _torch()
        lib, dev = ix.lib, ix.device
        self.k, self.mode = int(k), int(mode)
        self.lock = threading.Lock()
        tt = ix.tables[node_type]
        self.h_anchor = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.h_anchor_np = self.h_anchor.numpy()
        self.d_anchor = torch.zeros(1, dtype=torch.int32, device=dev)
        self.Q64 = torch.empty((1, ix.d), dtype=torch.float64, device=dev)
        self.Q32 = torch.empty((1, ix.ldx), dtype=torch.float32, device=dev)
        nw = 2 * k + 1 + (k if mode == 2 else 0)
        self.buf = torch.zeros(nw, dtype=torch.int64, device=dev)
        self.h_buf = torch.zeros(nw, dtype=torch.int64).pin_memory()
        self.h_np = self.h_buf.numpy()
        idx = self.buf[:k].view(1, k)
        score = self.buf[k:2 * k].view(torch.float64).view(1, k)
        flags = self.buf[2 * k:2 * k + 1].view(torch.int32)[:1]
        d2 = self.buf[2 * k + 1:].view(torch.float64).view(1, k) if mode == 2 else None
        x16, x16_exp = ix._x16(tt, 1)
        opt = (ix.KNN_F16_SCREEN if (ix.f16_screen and x16 is not None) else 0) | ix._path_options()
        wsb = lib.hwer_knn_workspace_bytes_opts(1, tt.n, ix.d, k, opt)
        if wsb == 0:
            raise native.HwerNativeError("knn: unsupported shape (n=%d, d=%d, k=%d)" % (tt.n, ix.d, k))
        self.ws = torch.empty(int(wsb), dtype=torch.uint8, device=dev)
        self.wse = torch.empty(int(lib.hwer_knn_exact_device_workspace_bytes(tt.n, k)), dtype=torch.uint8, device=dev)
        P = native.ptr
        A64 = P(ix.V64) if mode != 2 else None
        X64p, xmap = ix._x64(tt)

        def launch():
            st = native.stream_ptr(dev)
            self.d_anchor.copy_(self.h_anchor, non_blocking=True)
            native.check(lib.hwer_query_mix_f64rows(P(ix.V64), ix.N, ix.d, ix.ld64, P(self.d_anchor), None, None, None, None, 1, P(self.Q64), P(self.Q32),
                                                    ix.ldx, st), "query_mix_f64rows")
            native.check(lib.hwer_knn_search_f64rows(
                P(tt.X), P(tt.hnorm), tt.n, ix.d, ix.ldx, P(self.Q32), P(self.Q64), X64p, ix.ld64, xmap, A64, ix.ld64, P(self.d_anchor), 1, k,
                mode, 1, P(tt.rows_dev), 0, tt.hnorm_max, P(idx), P(score), P(d2), P(flags), P(self.ws), self.ws.numel(), opt, P(x16), x16_exp, st),
                "knn_search_f64rows")
            native.check(lib.hwer_knn_exact_device_f64rows(
                X64p, ix.ld64, xmap, tt.n, ix.d, P(self.Q64), A64, ix.ld64, P(self.d_anchor), k, mode, 1, P(tt.rows_dev), 0, P(idx), P(score),
                P(d2), P(flags), P(self.ws), None, P(self.wse), self.wse.numel(), st), "knn_exact_device_f64rows")
            self.h_buf.copy_(self.buf, non_blocking=True)
        with native.device_guard(dev):
            native.check(lib.hwer_knn_workspace_init(P(self.ws), self.ws.numel(), native.stream_ptr(dev)), "knn_workspace_init")
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    launch()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                launch()
            self.graph = g
        self._keep = (tt, x16, ix.V64, ix.V)
        self.dev, self.dev_index = dev, dev.index
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._sync = lib.hwer_stream_synchronize

    def run(self, anchor: int):
        """-> (ids (k,) int64, scores (k,) float64, flag, squared distances (k,) float64 or None) as fresh numpy arrays; thread-safe like _OneAnchorSearch.run"""
        with self.lock:
            self.h_anchor_np[0] = anchor
            self.graph.replay()
            raw = self._raw_stream
            if raw is not None:
                rc = self._sync(raw(self.dev_index))
                if rc:
                    native.check(rc, "stream_synchronize")
            else:
                _torch().cuda.current_stream(self.dev).synchronize()
            k, h = self.k, self.h_np
            d2 = h[2 * k + 1:].view(np.float64).copy() if self.mode == 2 else None
            return h[:k].copy(), h[k:2 * k].view(np.float64).copy(), int(h[2 * k] & 0xFFFFFFFF), d2


class DeviceIndex:
    #: queries per launch of the k-NN kernels = 512 workgroup slots (256 CUs x 2) x 256 queries: one full round of
    #: workgroups without item splits; also bounds the candidate workspace (8 KiB per query and list)
    QUERY_CHUNK = 131072

    def __init__(self, vectors, type_rows: Dict[str, np.ndarray], device=None, arith: Optional[int] = None,
                 audit: bool = True):
        """vectors: (N, d) float32/float64 numpy array or float32 torch tensor already on the device.
        type_rows: node type -> ascending int64 array of global row ids of that type.
        arith: 0 = score / query arithmetic of the reference when its `vectors` are float64 (default for float64
        input), 1 = float32 (default for float32 input)."""
        torch = _torch()
        self.lib = native.lib()
        self.device = _require_gpu(device)
        with native.device_guard(self.device):
            self._init(vectors, type_rows, arith, audit)

    #: keep genuine float64 rows on the device and answer from them (False: the stage-wise contract of rounds 1-2 - rows are rounded
    #: to float32 at upload, results are exact for the rounded table)
    keep_f64 = True

    def _init(self, vectors, type_rows, arith, audit):
        torch = _torch()
        self.V64 = None
        self.ld64 = 0
        if isinstance(vectors, np.ndarray):
            if arith is None:
                arith = 0 if vectors.dtype == np.float64 else 1
            host = np.ascontiguousarray(vectors, dtype=np.float32)
            N, d = host.shape
            src = torch.from_numpy(host).to(self.device)
            # Genuine float64 rows (not float32 values widened): the reference's KD-tree and predict work on them as they are
            # (hwer/recommendation_base.py:74,79,109,135-144).  The float64 table stays resident beside its float32 rounding: the MFMA
            # screen runs on the rounding (a superset is all it has to keep), values come from the float64 rows (_search_candidates).
            if vectors.dtype == np.float64 and self.keep_f64 and not np.array_equal(host.astype(np.float64), vectors, equal_nan=True):
                # rows padded with zeros to a multiple of 8 columns: every 8-column chunk of a row is one aligned 64-byte segment (what the
                # fused float64-row select gathers, hwer_knn_search_f64rows)
                self.ld64 = roundup(d, 8)
                if self.ld64 == d:
                    self.V64 = torch.from_numpy(np.ascontiguousarray(vectors)).to(self.device)
                else:
                    self.V64 = torch.zeros((N, self.ld64), dtype=torch.float64, device=self.device)
                    self.V64[:, :d] = torch.from_numpy(np.ascontiguousarray(vectors)).to(self.device)
        else:
            if arith is None:
                arith = 1
            src = vectors.to(self.device, dtype=torch.float32)
            N, d = src.shape
        self.N, self.d, self.arith = int(N), int(d), int(arith)
        self.ldv = roundup(self.d, 4)
        if src.shape[1] == self.ldv and src.is_contiguous():
            self.V = src
        else:
            self.V = torch.zeros((self.N, self.ldv), dtype=torch.float32, device=self.device)
            self.V[:, :self.d] = src
        # d <= 256: the MFMA screen's padded widths.  Wider tables (the reference has no width limit,
        # hwer/recommendation_base.py:74) keep rows padded to 16 columns and take the exhaustive float64 path.
        self.wide = self.d > self.lib.hwer_knn_max_dim()
        if self.d > 1024:
            raise native.HwerNativeError("k-NN index supports 1 <= d <= 1024 (got d=%d)" % self.d)
        self.ldx = roundup(self.d, 16) if self.wide else self.lib.hwer_knn_padded_dim(self.d)
        self.audit_result = self.unit_audit() if audit else None
        self.tables: Dict[str, _TypeTable] = {}
        for t, rows in type_rows.items():
            self.add_type(t, rows)
        self._ws = None

    # ------------------------------------------------------------------------------------------------
    @_on_device
    def unit_audit(self, eps: float = 1e-4):
        """(violations, mean |len - 1|, positive, negative, nan_rows) — utils.unit_length_violations on device."""
        torch = _torch()
        ws = torch.empty(self.lib.hwer_unit_audit_ws_floats(self.N), dtype=torch.float32, device=self.device)
        counts = torch.zeros(3, dtype=torch.int64, device=self.device)
        dev = torch.zeros(1, dtype=torch.float64, device=self.device)
        native.check(self.lib.hwer_unit_audit_f32(native.ptr(self.V), self.N, self.d, self.ldv, eps, native.ptr(ws),
                                                  native.ptr(counts), native.ptr(dev), native.stream_ptr(self.device)), "unit_audit")
        pos, neg, nan = (int(x) for x in counts.tolist())
        return pos + neg, float(dev.item()) / self.N, pos, neg, nan

    @_on_device
    def add_type(self, node_type: str, rows: np.ndarray):
        torch = _torch()
        tt = _TypeTable()
        tt.rows = np.ascontiguousarray(rows, dtype=np.int64)
        tt.n = len(tt.rows)
        tt.rows_dev = torch.from_numpy(tt.rows).to(self.device)
        if tt.n and int(tt.rows[-1]) - int(tt.rows[0]) == tt.n - 1 and bool((np.diff(tt.rows) == 1).all()):
            tt.row0 = int(tt.rows[0])
        n_pad = self.lib.hwer_knn_padded_rows(tt.n)   # whole 32-row tiles: pad rows are zero with a +inf half norm
        tt.X = torch.empty((n_pad, self.ldx), dtype=torch.float32, device=self.device)
        tt.hnorm = torch.empty(n_pad, dtype=torch.float32, device=self.device)
        native.check(self.lib.hwer_pack_rows_f32(native.ptr(self.V), self.N, self.d, self.ldv, native.ptr(tt.rows_dev), tt.n,
                                                 native.ptr(tt.X), self.ldx, native.ptr(tt.hnorm), native.stream_ptr(self.device)),
                     "pack_rows")
        tt.hnorm_max = float(tt.hnorm[:tt.n].max().item()) if tt.n else 0.0
        self.tables[node_type] = tt
        # cached one-anchor graphs of this type hold raw pointers into the table they were captured on: drop them
        cache = self.__dict__.get("_one_anchor")
        if cache:
            for key in [k_ for k_ in cache if k_[0] == node_type]:
                del cache[key]
        return tt

    # ------------------------------------------------------------------------------------------------
    @_on_device
    def pair_score(self, src: np.ndarray, dst: np.ndarray, arith: Optional[int] = None) -> np.ndarray:
        """RecommendationBase.predict on resolved indices (-1 = unknown node)."""
        torch = _torch()
        arith = self.arith if arith is None else arith
        s = torch.from_numpy(np.ascontiguousarray(src, dtype=np.int32)).to(self.device)
        t = torch.from_numpy(np.ascontiguousarray(dst, dtype=np.int32)).to(self.device)
        P = s.numel()
        if arith == 0 and self.V64 is not None:     # float64 rows: the reference's float64 products
            out = torch.empty(P, dtype=torch.float64, device=self.device)
            native.check(self.lib.hwer_pair_score_f64rows(native.ptr(self.V64), self.N, self.d, self.ld64, native.ptr(s), native.ptr(t), P,
                                                          native.ptr(out), native.stream_ptr(self.device)), "pair_score_f64rows")
            return out.cpu().numpy()
        if arith == 0:
            out = torch.empty(P, dtype=torch.float64, device=self.device)
            fn = self.lib.hwer_pair_score_f64
        else:
            out = torch.empty(P, dtype=torch.float32, device=self.device)
            fn = self.lib.hwer_pair_score_f32
        native.check(fn(native.ptr(self.V), self.N, self.d, self.ldv, native.ptr(s), native.ptr(t), P, native.ptr(out),
                        native.stream_ptr(self.device)), "pair_score")
        return out.cpu().numpy()

    # ------------------------------------------------------------------------------------------------
    @_on_device
    def build_queries(self, anchors: np.ndarray, positive: Optional[Sequence[Sequence[int]]] = None,
                      negative: Optional[Sequence[Sequence[int]]] = None):
        """-> (Q64 (B, d) float64, Q32 (B, ldx) float32, anchors_dev int32) on the device."""
        torch = _torch()
        anchors = np.ascontiguousarray(anchors, dtype=np.int32)
        if anchors.size == 1 and 0 <= int(anchors.flat[0]) < self.N:
            # one anchor (the reference-shaped per-anchor loop): a one-element VIEW of a resident arange - no host-to-device copy
            if getattr(self, "_row_ids_dev", None) is None:
                self._row_ids_dev = torch.arange(self.N, dtype=torch.int32, device=self.device)
            a0 = int(anchors.flat[0])
            a = self._row_ids_dev[a0:a0 + 1]
        else:
            a = torch.from_numpy(anchors).to(self.device)
        B = a.numel()

        def csr(lists):
            if lists is None:
                return None, None
            ptrs = np.zeros(B + 1, dtype=np.int64)
            flat = []
            for i, l in enumerate(lists):
                l = [] if l is None else list(l)
                flat.extend(l)
                ptrs[i + 1] = len(flat)
            if not flat:
                return None, None
            return (torch.from_numpy(ptrs).to(self.device),
                    torch.from_numpy(np.asarray(flat, dtype=np.int32)).to(self.device))

        pp, pi = csr(positive)
        np_, ni = csr(negative)
        Q64 = torch.empty((B, self.d), dtype=torch.float64, device=self.device)
        Q32 = torch.empty((B, self.ldx), dtype=torch.float32, device=self.device)
        if self.V64 is not None and self.arith == 0:
            native.check(self.lib.hwer_query_mix_f64rows(native.ptr(self.V64), self.N, self.d, self.ld64, native.ptr(a), native.ptr(pp), native.ptr(pi),
                                                         native.ptr(np_), native.ptr(ni), B, native.ptr(Q64), native.ptr(Q32), self.ldx,
                                                         native.stream_ptr(self.device)), "query_mix_f64rows")
            return Q64, Q32, a
        native.check(self.lib.hwer_query_mix(native.ptr(self.V), self.N, self.d, self.ldv, native.ptr(a), native.ptr(pp),
                                             native.ptr(pi), native.ptr(np_), native.ptr(ni), B, self.arith,
                                             native.ptr(Q64), native.ptr(Q32), self.ldx, native.stream_ptr(self.device)), "query_mix")
        return Q64, Q32, a

    #: single-anchor searches without positive / negative lists replay a cached hipGraph (see _OneAnchorSearch); False: the general path
    one_anchor_graphs = True

    @staticmethod
    def _has_lists(lists):
        return lists is not None and any(l is not None and len(l) > 0 for l in lists)

    @property
    def _max_k(self):
        mk = self.__dict__.get("_max_k_cached")
        if mk is None:
            mk = self.__dict__["_max_k_cached"] = int(self.lib.hwer_knn_max_k())
        return mk

    #: catalogues above this many rows keep a float16 copy of their table for small query batches
    F16_MIN_ROWS = 8193
    use_f16_storage = True
    #: HWER_KNN_F16_SCREEN: larger query batches also screen against the float16 copy (streaming path,
    #: v_mfma_f32_32x32x8_f16: 1.8-2.5x faster on 26 K ... 1.25 M rows); results unchanged, see include/hwer_hip.h.
    #: HWER_F16_SCREEN=0 switches it off.
    f16_screen = bool(int(os.environ.get("HWER_F16_SCREEN", "1") or 0))
    KNN_F16_SCREEN = 2

    def _x16(self, tt, nb=0):
        """float16 copy of the type's table, built on first use (half the bytes streamed by the small-batch screen; the float16
        MFMA screen of larger batches).  Small catalogues (dense path) get one when a batch of `nb` queries makes the float16
        screen pay (hwer_knn_f16_screen_pays)."""
        if not self.use_f16_storage:
            return None, 0
        if tt.n < self.F16_MIN_ROWS and not (self.f16_screen and nb and self.lib.hwer_knn_f16_screen_pays(int(nb), int(tt.n), int(self.d))):
            return None, 0
        if tt.X16 is None:
            torch = _torch()
            tt.x16_exp = int(self.lib.hwer_knn_f16_scale_exp(float(tt.hnorm_max)))
            tt.X16 = torch.empty(tt.X.shape, dtype=torch.float16, device=self.device)
            native.check(self.lib.hwer_pack_rows_f16(native.ptr(tt.X), tt.X.shape[0], self.ldx, tt.x16_exp, native.ptr(tt.X16),
                                                     native.stream_ptr(self.device)), "pack_rows_f16")
        return tt.X16, tt.x16_exp

    def _x64(self, tt):
        """(float64 table pointer, row map pointer) the float64-row kernels address `tt`'s rows with: the table sliced at the type's first row and NO map when
        the rows are a range of the table (one dependent load per gathered row less), else the whole table + the type's row list"""
        if tt.row0 >= 0:
            return native.ptr(self.V64[tt.row0:tt.row0 + tt.n]), None
        return native.ptr(self.V64), native.ptr(tt.rows_dev)

    def _workspace(self, nbytes):
        torch = _torch()
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            # every (re)allocation: zero the header once (the fused prologue keeps its audit accumulators there and fails loudly on a
            # workspace that never went through this call)
            native.check(self.lib.hwer_knn_workspace_init(native.ptr(self._ws), self._ws.numel(), native.stream_ptr(self.device)), "knn_workspace_init")
        return self._ws

    def _exact_workspace(self, nbytes):
        torch = _torch()
        if getattr(self, "_ws_exact", None) is None or self._ws_exact.numel() < nbytes:
            self._ws_exact = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws_exact

    @staticmethod
    def _path_options():
        """Path-forcing switches for tests (results are identical on every path): HWER_NO_DENSE / HWER_NO_SMALL / HWER_NO_PINGPONG in the
        environment of the HOST mirror map to the library's HWER_KNN_NO_DENSE / HWER_KNN_NO_SMALL / HWER_KNN_NO_PINGPONG options."""
        return ((4 if os.environ.get("HWER_NO_DENSE") else 0) | (8 if os.environ.get("HWER_NO_SMALL") else 0) |
                (64 if os.environ.get("HWER_NO_PINGPONG") else 0))     # HWER_KNN_NO_PINGPONG: rounds 1-3's float16 streaming kernel

    @_on_device
    def search_queries(self, node_type: str, Q64, Q32, k: int, anchors_dev=None, score_mode: Optional[int] = None,
                       sort_by_score: bool = True, id_base: Optional[int] = None, want_dist: bool = True, _flags_host_out: Optional[list] = None):
        """Exact k nearest rows of `node_type` for B query vectors already on the device.
        -> torch tensors (idx (B,k) int64 global ids, score (B,k) f64, dist2 (B,k) f64 or None, flags (B) int32).
        want_dist=False (score order, base scoring only) skips the distances the result does not depend on."""
        torch = _torch()
        tt = self.tables[node_type]
        B = Q64.shape[0]
        if k > tt.n:
            raise ValueError("k must be less than or equal to the number of training points")
        if score_mode is None:
            score_mode = self.arith
        idx = torch.empty((B, k), dtype=torch.int64, device=self.device)
        score = torch.empty((B, k), dtype=torch.float64, device=self.device)
        if not want_dist and (score_mode == 2 or not sort_by_score):
            want_dist = True
        dist2 = torch.empty((B, k), dtype=torch.float64, device=self.device) if want_dist else None
        flags = torch.empty(B, dtype=torch.int32, device=self.device)
        use_rows = id_base is None
        if anchors_dev is None and score_mode != 2:
            # anchor row of query b defaults to row b of the table - materialised ONCE, so that every chunk and every
            # fallback tier scores query b against V[b] (a NULL a_idx means "chunk-local index" inside the library)
            anchors_dev = torch.arange(B, dtype=torch.int32, device=self.device)
        if self.V64 is not None and not self.wide and k <= self.lib.hwer_knn_max_k() and score_mode in (0, 2) and not os.environ.get("HWER_F64ROWS_CANDIDATES"):
            # genuine float64 rows inside the fused search's limits (round 6): the float32 flavour's launches, values from the float64 rows
            fh = self._search_f32(tt, Q64, Q32, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, want_dist, idx, score, dist2, flags, f64rows=True)
            if _flags_host_out is not None:
                _flags_host_out.append(fh)
            return idx, score, dist2, flags
        if self.V64 is not None or k > self.lib.hwer_knn_max_k() or self.wide:
            if os.environ.get("HWER_FORCE_EXHAUSTIVE") and self.V64 is None:     # tests: the certified fallback on its own
                self._search_exhaustive(tt, Q64, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2, flags)
            else:
                self._search_candidates(tt, Q64, Q32, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2, flags)
            return idx, score, dist2, flags
        fh = self._search_f32(tt, Q64, Q32, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, want_dist, idx, score, dist2, flags)
        if _flags_host_out is not None:
            _flags_host_out.append(fh)      # the host copy of `flags` the small-batch path already fetched (None otherwise)
        return idx, score, dist2, flags

    def _search_f32(self, tt, Q64, Q32, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, want_dist, idx, score, dist2, flags, no_sync=False, workspaces=None,
                    f64rows=False):
        """the fused HIP search (d <= 256, k <= 392): MFMA screen -> float64 re-rank -> certificate, then the
        exhaustive tiers for uncertified queries; fills idx / score / dist2 / flags.  -> the host copy of the final flags when the
        small-batch branch fetched one and nothing changed them afterwards (else None).
        f64rows (round 6): GENUINE float64 rows - the screen runs on their float32 rounding (tt.X / tt.X16), the select stage and every
        exhaustive tier read the float64 rows of V64 (hwer_knn_search_f64rows, hwer_knn_exact_device_f64rows, hwer_knn_exact_f64rows):
        the same launches as the float32 flavour; score_mode 0 or 2."""
        torch = _torch()
        P = native.ptr
        if f64rows:
            assert self.V64 is not None and score_mode in (0, 2)
            A64 = P(self.V64) if score_mode != 2 else None
            X64p, xmap = self._x64(tt)
        B = Q64.shape[0]
        path_opt = self._path_options()
        st = native.stream_ptr(self.device)       # one stream lookup per call (torch.cuda.current_stream costs ~4 us)
        sizes = self.__dict__.setdefault("_ws_sizes", {})
        for b0 in range(0, B, self.QUERY_CHUNK):
            b1 = min(B, b0 + self.QUERY_CHUNK)
            nb = b1 - b0
            # the small-batch screen always reads the float16 copy; the streaming screen only with f16_screen
            x16, x16_exp = self._x16(tt, nb) if (nb <= 1024 or self.f16_screen) else (None, 0)
            opt = (self.KNN_F16_SCREEN if (self.f16_screen and x16 is not None) else 0) | path_opt
            key = (nb, tt.n, k, opt)      # (the plan - and with it the workspace layout - is a function of the options of the call)
            if key not in sizes:
                sizes[key] = (self.lib.hwer_knn_workspace_bytes_opts(nb, tt.n, self.d, k, opt), self.lib.hwer_knn_exact_device_workspace_bytes(tt.n, k))
            wsb, wseb = sizes[key]
            if wsb == 0:
                raise native.HwerNativeError("knn: unsupported shape (B=%d, n=%d, d=%d, k=%d; k <= %d)" %
                                             (nb, tt.n, self.d, k, self.lib.hwer_knn_max_k()))
            if workspaces is not None:
                # a captured caller (_OneAnchorSearchF64) owns its workspaces: the shared ones may be re-allocated by a later, larger search
                if "ws" not in workspaces:
                    workspaces["ws"] = torch.empty(int(wsb), dtype=torch.uint8, device=self.device)
                    native.check(self.lib.hwer_knn_workspace_init(native.ptr(workspaces["ws"]), int(wsb), st), "knn_workspace_init")
                    workspaces["wse"] = torch.empty(max(int(wseb), 1), dtype=torch.uint8, device=self.device)
                ws = workspaces["ws"]
            else:
                ws = self._workspace(wsb)
            a_sub = anchors_dev[b0:b1] if anchors_dev is not None else None
            wse = workspaces["wse"] if workspaces is not None else self._exact_workspace(wseb)
            if f64rows:
                native.check(self.lib.hwer_knn_search_f64rows(
                    P(tt.X), P(tt.hnorm), tt.n, self.d, self.ldx, P(Q32[b0:b1]), P(Q64[b0:b1]), X64p, self.ld64, xmap, A64, self.ld64,
                    P(a_sub), nb, k, score_mode, 1 if sort_by_score else 0, P(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base),
                    tt.hnorm_max, P(idx[b0:b1]), P(score[b0:b1]), P(dist2[b0:b1]) if want_dist else None, P(flags[b0:b1]), P(ws), ws.numel(), opt,
                    P(x16), x16_exp, st), "knn_search_f64rows")
                native.check(self.lib.hwer_knn_exact_device_f64rows(
                    X64p, self.ld64, xmap, tt.n, self.d, P(Q64[b0:b1]), A64, self.ld64, P(a_sub), k, score_mode, 1 if sort_by_score else 0,
                    P(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base), P(idx[b0:b1]), P(score[b0:b1]),
                    P(dist2[b0:b1]) if want_dist else None, P(flags[b0:b1]), P(ws), None, P(wse), wse.numel(), st), "knn_exact_device_f64rows")
                continue
            native.check(self.lib.hwer_knn_search_opts_f32(
                native.ptr(tt.X), native.ptr(tt.hnorm), tt.n, self.d, self.ldx, native.ptr(Q32[b0:b1]), native.ptr(Q64[b0:b1]),
                native.ptr(self.V) if score_mode != 2 else None, self.ldv, native.ptr(a_sub), nb, k, score_mode,
                1 if sort_by_score else 0, native.ptr(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base),
                tt.hnorm_max, native.ptr(idx[b0:b1]), native.ptr(score[b0:b1]), native.ptr(dist2[b0:b1]) if want_dist else None,
                native.ptr(flags[b0:b1]), native.ptr(ws), ws.numel(), opt, native.ptr(x16), x16_exp, st), "knn_search")
            # queries the screen could not certify (rare) are recomputed exhaustively on the device, stream-ordered, up
            # to 8 per call; anything beyond that is picked up by the host-driven pass below
            native.check(self.lib.hwer_knn_exact_device(
                native.ptr(tt.X), tt.n, self.d, self.ldx, native.ptr(Q64[b0:b1]), native.ptr(self.V) if score_mode != 2 else None,
                self.ldv, native.ptr(a_sub), k, score_mode, 1 if sort_by_score else 0,
                native.ptr(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base), native.ptr(idx[b0:b1]),
                native.ptr(score[b0:b1]), native.ptr(dist2[b0:b1]) if want_dist else None, native.ptr(flags[b0:b1]), native.ptr(ws), None,
                native.ptr(wse), wse.numel(), st), "knn_exact_device")
        if no_sync:      # (hipGraph capture: the launches only - queries the on-device pass left uncertified keep flag 0 for the caller)
            return None
        if B <= 4096:
            # small batches: the flags themselves come to the host (one copy; `search` reuses it) instead of a count kernel + its result
            flags_host = flags.cpu()
            n_bad_host = int((flags_host == 0).sum())
        else:
            flags_host = None
            n_bad = torch.zeros(1, dtype=torch.int32, device=self.device)
            native.check(self.lib.hwer_knn_count_uncertified(native.ptr(flags), B, native.ptr(n_bad), st), "count_uncertified")
            n_bad_host = int(n_bad.item())
        if n_bad_host:
            flags_host = None
            bad = torch.nonzero(flags == 0).flatten()
            if bad.numel() > self.RETRY_MIN:
                # many uncertified queries = the sampled thresholds were defeated (e.g. a catalogue sorted by cluster): run
                # them again with exact thresholds before falling back to the exhaustive pass query by query
                self._retry_exact_thresholds(tt, bad, Q64, Q32, anchors_dev, k, score_mode, sort_by_score, use_rows, id_base,
                                             idx, score, dist2, flags, f64rows=f64rows)
                bad = torch.nonzero(flags == 0).flatten()
            if bad.numel():
                self._exact_fallback(tt, bad, Q64, anchors_dev, k, score_mode, sort_by_score, use_rows, id_base, idx, score,
                                     dist2, flags, f64rows=f64rows)
        return flags_host

    #: extra candidates asked of the screen by _search_candidates (the certificate needs a gap between the k-th exact distance and
    #: the bound on everything that was not a candidate)
    CAND_MARGIN = 32

    def _order_rows(self, values, n_in, n_out, descending=False, gather=None, ids=None, id_add=0, id_map=None, payload_i64=None, payload_f64=None,
                    bound=None, every_row=False, want_flags=False):
        """hwer_knn_order_rows_f64 over the rows of `values`: -> (out_values (B, n_out), out_ids (B, n_out) int64, out_payload_i64, out_payload_f64,
        flags) (None where not asked).  Rows longer than the LDS network get the rank-sort workspace."""
        torch = _torch()
        lib, dev = self.lib, self.device
        B = values.shape[0]
        o_val = torch.empty((B, n_out), dtype=torch.float64, device=dev)
        o_ids = torch.empty((B, n_out), dtype=torch.int64, device=dev)
        o_pi = torch.empty((B, n_out), dtype=torch.int64, device=dev) if payload_i64 is not None else None
        o_pf = torch.empty((B, n_out), dtype=torch.float64, device=dev) if payload_f64 is not None else None
        fl = torch.empty(B, dtype=torch.int32, device=dev) if want_flags else None
        wsb = lib.hwer_knn_order_rows_workspace_bytes(B, n_in)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        native.check(lib.hwer_knn_order_rows_f64(
            native.ptr(values), values.stride(0), native.ptr(gather), gather.stride(0) if gather is not None else 0, native.ptr(ids),
            ids.stride(0) if ids is not None else 0, B, n_in, n_out, 1 if descending else 0, native.ptr(o_val), native.ptr(o_ids), int(id_add),
            native.ptr(id_map), native.ptr(payload_i64), native.ptr(o_pi), native.ptr(payload_f64), native.ptr(o_pf), native.ptr(bound), native.ptr(fl),
            1 if every_row else 0, native.ptr(ws), wsb, native.stream_ptr(dev)), "knn_order_rows")
        return o_val, o_ids, o_pi, o_pf, fl

    def _nearest_of_all(self, D, n, k):
        """(nq, n) float64 squared distances -> the k smallest per row in (distance, row) order: (d2 (nq, k), local rows (nq, k)).
        hwer_knn_topk_rows_f64 (exact radix select + ordered compaction) then hwer_knn_order_rows_f64 on the k selected."""
        torch = _torch()
        nq = D.shape[0]
        sel = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        native.check(self.lib.hwer_knn_topk_rows_f64(native.ptr(D), D.stride(0), n, nq, k, native.ptr(sel), k, native.stream_ptr(self.device)), "knn_topk_rows_f64")
        d2, local, _, _, _ = self._order_rows(D, k, k, gather=sel)
        return d2, local

    def _score_and_emit(self, pos, nb, d2, local, tt, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2):
        """scores of the k nearest (hwer_pair_score*: numpy's pairwise order; GcnNCF flavour: from the distance), then the reference's final
        order - a stable sort by score descending (hwer/recommendation_base.py:172-173) = hwer_knn_order_rows_f64 on (score desc, position asc)
        with the ids and distances as payloads - into the rows `pos` (a slice, or an index tensor of `nb` positions) of the outputs."""
        torch = _torch()
        lib, dev = self.lib, self.device
        stream = native.stream_ptr(dev)
        rows = tt.rows_dev[local]                                      # rows of the vector table (pair scoring)
        ids = rows if use_rows else local + int(id_base)             # reported ids
        if score_mode == 2:
            sc = (-1.0 * torch.sqrt(d2) + 2.0) / 2.0                 # hwer/gcn_ncf.py:379-383
        else:
            a = anchors_dev[pos].to(torch.int32)
            src = a.view(nb, 1).expand(nb, k).contiguous().view(-1)
            dst = rows.to(torch.int32).contiguous().view(-1)
            if score_mode == 0 and self.V64 is not None:
                out = torch.empty(nb * k, dtype=torch.float64, device=dev)
                native.check(lib.hwer_pair_score_f64rows(native.ptr(self.V64), self.N, self.d, self.ld64, native.ptr(src), native.ptr(dst), nb * k,
                                                         native.ptr(out), stream), "pair_score_f64rows")
            else:
                out = torch.empty(nb * k, dtype=torch.float64 if score_mode == 0 else torch.float32, device=dev)
                fn = lib.hwer_pair_score_f64 if score_mode == 0 else lib.hwer_pair_score_f32
                native.check(fn(native.ptr(self.V), self.N, self.d, self.ldv, native.ptr(src), native.ptr(dst), nb * k, native.ptr(out), stream),
                             "pair_score")
            sc = out.view(nb, k).to(torch.float64)
        if sort_by_score:
            sc, _, ids, d2, _ = self._order_rows(sc.contiguous(), k, k, descending=True, payload_i64=ids.contiguous(), payload_f64=d2.contiguous())
        idx[pos], score[pos] = ids, sc
        if dist2 is not None:
            dist2[pos] = d2

    def _search_candidates(self, tt, Q64, Q32, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2, flags, no_sync=False, workspaces=None):
        """Exact k-NN as "screen a superset, re-rank it exactly, certify" for the shapes the fused HIP search does not take (since round 6 genuine
        float64 rows with d <= 256 and k <= 392 take the FUSED float64-row search, _search_f32(f64rows=True); this composition stays for their
        float32-arithmetic score flavour, as the independent implementation the tests hold the fused path to - HWER_F64ROWS_CANDIDATES=1 - and for
        wide tables / large k):
          * genuine float64 `vectors` (V64): the HIP search on the float32 rounding of the table returns the k + 32 nearest ROUNDED rows
            (exact for the rounded table); rounding moves a squared distance by at most eps(D) = 2 sqrt(D) delta + delta^2, delta =
            2^-24 ||x||max, so every row outside the candidates has a true squared distance >= T - eps(T), T = the largest candidate's
            rounded squared distance;
          * tables wider than hwer_knn_max_dim() or k above hwer_knn_max_k(): the screen is hwer_knn_keys_f32 - float32 MFMA keys
            Q X^T - half norms, the columns consumed in 16-column slabs (csrc/knn_wide.hip) - and hwer_knn_topk_rows_f32 keeps the k + 32
            largest keys of every query (exact radix select; ids ascending).  A float32 key is off by at most E = (2 d + 32) 2^-24
            (||q|| ||x||max + ||x||max^2) whatever the summation order, so every row outside the candidates has a true squared
            distance >= ||q||^2 - 2 (kmin + E).
        The candidates' exact squared distances come from hwer_knn_cand_dist2 (scikit-learn's summation order, on the float64 rows when
        there are any), hwer_knn_order_rows_f64 puts them in (distance, id) order, cuts to k and evaluates the certificate - the k-th exact
        distance strictly below the bound, or every row a candidate; the others - near-ties across the bound - are recomputed over ALL
        rows (hwer_knn_topk_rows_f64 + the same ordering).  hwer_pair_score* give the scores, one more hwer_knn_order_rows_f64 the
        reference's stable order by score.  Same values and order as the reference; every step is a kernel of libhwer_hip.so (rounds
        1-3: a library GEMM and library top-k / sorts)."""
        torch = _torch()
        lib, dev = self.lib, self.device
        B, n = Q64.shape[0], tt.n
        stream = native.stream_ptr(dev)
        kc = min(n, k + self.CAND_MARGIN)
        use_hip = self.V64 is not None and not self.wide and kc <= lib.hwer_knn_max_k()
        table, is64, ld = (self.V64, 1, self.ld64) if self.V64 is not None else (self.V, 0, self.ldv)
        xmax = float(np.sqrt(2.0 * max(tt.hnorm_max, 0.0))) * (1.0 + 1e-6)
        n_pad = tt.X.shape[0]
        per = max(1, min(B, 65535, (1 << 28) // max(n_pad if not use_hip else kc, 1)))       # <= 1 GiB of float32 keys at a time
        for b0 in range(0, B, per):
            b1 = min(B, b0 + per)
            nb = b1 - b0
            q64 = Q64[b0:b1]
            if use_hip:
                local = torch.empty((nb, kc), dtype=torch.int64, device=dev)
                c_sc = torch.empty((nb, kc), dtype=torch.float64, device=dev)
                c_d2 = torch.empty((nb, kc), dtype=torch.float64, device=dev)
                c_fl = torch.empty(nb, dtype=torch.int32, device=dev)
                # (launches only: a query the candidate search - on-device exhaustive pass included - leaves uncertified is treated like a failed
                # certificate below; its entries are clamped to valid rows so that the kernels behind this one read inside the table)
                self._search_f32(tt, q64, Q32[b0:b1], kc, None, 2, False, False, 0, True, local, c_sc, c_d2, c_fl, no_sync=True,
                                 workspaces=workspaces)   # ids = local rows
                local.clamp_(0, n - 1)
                T = c_d2[:, kc - 1]
                delta = (2.0 ** -24) * xmax
                bound = T - 2.0 * torch.sqrt(T) * delta - delta * delta - 1e-12 * torch.clamp(T, min=1.0)
            else:
                q32 = Q32[b0:b1]
                keys = torch.empty((nb, n_pad), dtype=torch.float32, device=dev)
                native.check(lib.hwer_knn_keys_f32(native.ptr(tt.X), native.ptr(tt.hnorm), n_pad, n, self.ldx, native.ptr(q32), q32.stride(0), nb,
                                                   native.ptr(keys), n_pad, stream), "knn_keys")
                local = torch.empty((nb, kc), dtype=torch.int64, device=dev)
                kmin = torch.empty(nb, dtype=torch.float32, device=dev)
                native.check(lib.hwer_knn_topk_rows_f32(native.ptr(keys), n_pad, n, nb, kc, native.ptr(local), kc, native.ptr(kmin), stream), "knn_topk_rows_f32")
                del keys
                qn2 = (q64 * q64).sum(1)
                E = (2.0 * self.d + 32.0) * (2.0 ** -24) * (torch.sqrt(qn2) * xmax + xmax * xmax)
                bound = qn2 - 2.0 * (kmin.to(torch.float64) + E) - 1e-12 * torch.clamp(qn2, min=1.0)
            # exact squared distances of the candidates, then (distance asc, row asc), cut to k, certificate
            rows = tt.rows_dev[local]
            D = torch.empty((nb, kc), dtype=torch.float64, device=dev)
            native.check(lib.hwer_knn_cand_dist2(native.ptr(table), is64, self.N, self.d, ld, native.ptr(rows), kc, nb, kc, native.ptr(q64),
                                                 native.ptr(D), stream), "knn_cand_dist2")
            d2, local, _, _, cert = self._order_rows(D, kc, k, ids=local, bound=bound.contiguous(), every_row=(kc >= n), want_flags=True)
            # Round 5: no host decision inside the loop.  Every query is scored and emitted from its candidates; one whose certificate failed (near-ties
            # across the bound) or whose candidate search stayed uncertified carries flag 0 and is repaired below, behind ONE synchronisation at
            # the end of the path (rounds 3-4: two synchronisations per chunk - the candidate search's flags and this certificate).
            ok = ((cert == 1) & (c_fl > 0)) if use_hip else (cert == 1)
            self._score_and_emit(slice(b0, b1), nb, d2, local, tt, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2)
            flags[b0:b1] = torch.where(ok, 1, 0).to(torch.int32)
        if no_sync:      # hipGraph capture (one-anchor calls): flag 0 goes back to the caller, which takes this path again outside the capture
            return
        bad = torch.nonzero(flags[:B] == 0).flatten()
        if bad.numel():     # all rows of the type, a few queries at a time (ties keep the row (= id) order); flag 2 like every exhaustive result
            step = max(1, min(65535, (1 << 26) // n))      # (hwer_knn_cand_dist2 takes at most 65 535 queries per call: the batch is its grid.y)
            for c0 in range(0, int(bad.numel()), step):
                sel = bad[c0:c0 + step]
                ns = int(sel.numel())
                allrows = tt.rows_dev.unsqueeze(0).expand(ns, n).contiguous()
                Da = torch.empty((ns, n), dtype=torch.float64, device=dev)
                native.check(lib.hwer_knn_cand_dist2(native.ptr(table), is64, self.N, self.d, ld, native.ptr(allrows), n, ns, n,
                                                     native.ptr(Q64.index_select(0, sel).contiguous()), native.ptr(Da), stream), "knn_cand_dist2(all)")
                d2s, locs = self._nearest_of_all(Da, n, k)
                self._score_and_emit(sel, ns, d2s, locs, tt, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2)
                flags[sel] = 2

    def _search_exhaustive(self, tt, Q64, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2, flags):
        """Shapes outside the selection kernels' limits - k above hwer_knn_max_k() or d above hwer_knn_max_dim(); the reference
        takes any k <= N_t and any width (hwer/recommendation_base.py:74,79): all float64 distances (hwer_knn_all_dist2_f64,
        scikit-learn's summation order), hwer_knn_topk_rows_f64 + hwer_knn_order_rows_f64 keep the k nearest in (distance, row) order,
        hwer_pair_score re-scores them against the anchors, one more ordering by score gives the reference's final order.  Same values and
        order as the regular path; works on a shard too (ids = id_base + local row).  The certified fallback of _search_candidates."""
        torch = _torch()
        B, n = Q64.shape[0], tt.n
        per = max(1, min(B, 65535, (1 << 30) // (8 * n)))   # <= 1 GiB of distances at a time
        stream = native.stream_ptr(self.device)
        for b0 in range(0, B, per):
            b1 = min(B, b0 + per)
            nb = b1 - b0
            D = torch.empty((nb, n), dtype=torch.float64, device=self.device)
            native.check(self.lib.hwer_knn_all_dist2_f64(native.ptr(tt.X), n, self.d, self.ldx, native.ptr(Q64[b0:b1]), nb,
                                                         native.ptr(D), stream), "knn_all_dist2")
            d2, local = self._nearest_of_all(D, n, k)                     # ties keep the row (= id) order
            del D
            self._score_and_emit(slice(b0, b1), b1 - b0, d2, local, tt, k, anchors_dev, score_mode, sort_by_score, use_rows, id_base, idx, score, dist2)
        flags.fill_(2)

    #: more uncertified queries than this go through the exact-threshold retry first
    RETRY_MIN = 16

    def _retry_exact_thresholds(self, tt, bad, Q64, Q32, anchors_dev, k, score_mode, sort_by_score, use_rows, id_base,
                                idx, score, dist2, flags, f64rows=False):
        torch = _torch()
        P = native.ptr
        nb = int(bad.numel())
        q64, q32 = Q64.index_select(0, bad).contiguous(), Q32.index_select(0, bad).contiguous()
        a_sub = anchors_dev.index_select(0, bad).contiguous() if anchors_dev is not None else None
        r_idx = torch.empty((nb, k), dtype=torch.int64, device=self.device)
        r_score = torch.empty((nb, k), dtype=torch.float64, device=self.device)
        r_dist2 = torch.empty((nb, k), dtype=torch.float64, device=self.device) if dist2 is not None else None
        r_flags = torch.empty(nb, dtype=torch.int32, device=self.device)
        opt = 1   # HWER_KNN_EXACT_THRESHOLDS
        for b0 in range(0, nb, self.QUERY_CHUNK):
            b1 = min(nb, b0 + self.QUERY_CHUNK)
            wsb = self.lib.hwer_knn_workspace_bytes_opts(b1 - b0, tt.n, self.d, k, opt)
            ws = self._workspace(wsb)
            if f64rows:
                native.check(self.lib.hwer_knn_search_f64rows(
                    P(tt.X), P(tt.hnorm), tt.n, self.d, self.ldx, P(q32[b0:b1]), P(q64[b0:b1]), self._x64(tt)[0], self.ld64, self._x64(tt)[1],
                    P(self.V64) if score_mode != 2 else None, self.ld64, P(a_sub[b0:b1]) if a_sub is not None else None, b1 - b0, k, score_mode,
                    1 if sort_by_score else 0, P(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base), tt.hnorm_max, P(r_idx[b0:b1]),
                    P(r_score[b0:b1]), P(r_dist2[b0:b1]) if r_dist2 is not None else None, P(r_flags[b0:b1]), P(ws), ws.numel(), opt, None, 0,
                    native.stream_ptr(self.device)), "knn_search_f64rows(exact thresholds)")
                continue
            native.check(self.lib.hwer_knn_search_opts_f32(
                native.ptr(tt.X), native.ptr(tt.hnorm), tt.n, self.d, self.ldx, native.ptr(q32[b0:b1]), native.ptr(q64[b0:b1]),
                native.ptr(self.V) if score_mode != 2 else None, self.ldv, native.ptr(a_sub[b0:b1]) if a_sub is not None else None,
                b1 - b0, k, score_mode, 1 if sort_by_score else 0, native.ptr(tt.rows_dev) if use_rows else None,
                0 if use_rows else int(id_base), tt.hnorm_max, native.ptr(r_idx[b0:b1]), native.ptr(r_score[b0:b1]),
                native.ptr(r_dist2[b0:b1]) if r_dist2 is not None else None, native.ptr(r_flags[b0:b1]), native.ptr(ws), ws.numel(),
                opt, None, 0, native.stream_ptr(self.device)), "knn_search(exact thresholds)")
        good = torch.nonzero(r_flags > 0).flatten()
        rows = bad.index_select(0, good)
        idx.index_copy_(0, rows, r_idx.index_select(0, good))
        score.index_copy_(0, rows, r_score.index_select(0, good))
        if dist2 is not None:
            dist2.index_copy_(0, rows, r_dist2.index_select(0, good))
        flags.index_copy_(0, rows, r_flags.index_select(0, good))

    def _exact_fallback(self, tt, bad, Q64, anchors_dev, k, score_mode, sort_by_score, use_rows, id_base, idx, score,
                        dist2, flags, f64rows=False):
        """Queries whose certificate failed (near-ties at the K'-cut, e.g. many duplicate rows): exhaustive float64
        pass on the device."""
        qsel = np.ascontiguousarray(bad.cpu().numpy(), dtype=np.int32)
        wsb = self.lib.hwer_knn_exact_workspace_bytes(tt.n, k)
        ws = self._workspace(wsb)
        if f64rows:
            P = native.ptr
            native.check(self.lib.hwer_knn_exact_f64rows(
                self._x64(tt)[0], self.ld64, self._x64(tt)[1], tt.n, self.d, P(Q64), P(self.V64) if score_mode != 2 else None, self.ld64, P(anchors_dev),
                qsel.ctypes.data_as(ctypes.c_void_p), len(qsel), k, score_mode, 1 if sort_by_score else 0, P(tt.rows_dev) if use_rows else None,
                0 if use_rows else int(id_base), P(idx), P(score), P(dist2) if dist2 is not None else None, P(flags), P(ws), ws.numel(),
                native.stream_ptr(self.device)), "knn_exact_f64rows")
            return
        native.check(self.lib.hwer_knn_exact_f64(
            native.ptr(tt.X), tt.n, self.d, self.ldx, native.ptr(Q64), native.ptr(self.V) if score_mode != 2 else None,
            self.ldv, native.ptr(anchors_dev), qsel.ctypes.data_as(ctypes.c_void_p), len(qsel), k, score_mode,
            1 if sort_by_score else 0, native.ptr(tt.rows_dev) if use_rows else None, 0 if use_rows else int(id_base),
            native.ptr(idx), native.ptr(score), native.ptr(dist2) if dist2 is not None else None, native.ptr(flags), native.ptr(ws), ws.numel(),
            native.stream_ptr(self.device)), "knn_exact")

    #: cached one-anchor graphs kept per index (each pins a search workspace, an exhaustive-pass workspace and a hipGraph): least recently
    #: used entries beyond this many are dropped
    ONE_ANCHOR_CACHE = 8

    def _one_anchor_search(self, node_type, k, mode):
        """The cached graph of (node type, k, score mode) UNDER THE CURRENT path switches: everything a captured graph bakes in is part of
        the key - the float16 switches, the HWER_NO_* path options of the environment and the identity of the float16 copy it reads."""
        import threading
        lock = self.__dict__.get("_one_anchor_lock")
        if lock is None:
            lock = self.__dict__.setdefault("_one_anchor_lock", threading.Lock())
        tt = self.tables[node_type]
        key = (node_type, int(k), mode if isinstance(mode, tuple) else int(mode), bool(self.f16_screen), bool(self.use_f16_storage), self._path_options(), id(tt.X16))
        with lock:      # (construction captures a hipGraph: one thread at a time)
            cache = self.__dict__.setdefault("_one_anchor", {})
            one = cache.pop(key, None)
            if one is None:
                one = _OneAnchorSearchF64(self, node_type, k, mode[1]) if isinstance(mode, tuple) else _OneAnchorSearch(self, node_type, k, mode)
                key = key[:-1] + (id(tt.X16),)       # (the constructor may have built the float16 copy)
            cache[key] = one                         # most recently used last
            while len(cache) > self.ONE_ANCHOR_CACHE:
                cache.pop(next(iter(cache)))
        return one

    @_on_device
    def search(self, node_type: str, anchors: np.ndarray, k: int = 200, positive=None, negative=None,
               scoring: str = "base", sort_by_score: bool = True, want_dist: bool = True):
        """Batched find_closest_neighbours on resolved indices.
        -> numpy (idx (B,k) int64, score (B,k) in the reference's dtype, dist (B,k) float64 or None, flags (B))."""
        mode = 2 if scoring == "gcn" else self.arith
        if (self.one_anchor_graphs and np.size(anchors) == 1 and sort_by_score and not want_dist and self.V64 is None and not self.wide
                and k <= self._max_k and k <= self.tables[node_type].n and not self._has_lists(positive) and not self._has_lists(negative)):
            # the reference-shaped per-anchor call (hwer/validation.py:30-35): one hipGraph replay per call (_OneAnchorSearch)
            a0 = int(np.asarray(anchors).reshape(-1)[0])
            if 0 <= a0 < self.N:
                one = self._one_anchor_search(node_type, k, mode)
                ids, sc, flag, d2 = one.run(a0)
                if flag > 0:
                    if mode == 1:
                        sc = sc.astype(np.float32)
                    # (GcnNCF scoring: the distances come back with the scores, as on the general path - the shape of the result does
                    # not depend on the number of anchors)
                    return ids.reshape(1, k), sc.reshape(1, k), (np.sqrt(d2).reshape(1, k) if d2 is not None else None), np.asarray([flag], dtype=np.int32)
                # (uncertified beyond the on-device exhaustive pass: the general path below runs the host-driven tiers)
        if (self.one_anchor_graphs and np.size(anchors) == 1 and sort_by_score and not want_dist and mode in (0, 2) and self.arith == 0 and self.V64 is not None
                and not self.wide and k <= self._max_k and k <= self.tables[node_type].n and not self._has_lists(positive)
                and not self._has_lists(negative) and not os.environ.get("HWER_FORCE_EXHAUSTIVE") and not os.environ.get("HWER_F64ROWS_CANDIDATES")):
            # the same call on genuine float64 rows (ContentRecommendation.fit since round 5): a cached hipGraph of the fused float64-row search
            a0 = int(np.asarray(anchors).reshape(-1)[0])
            if 0 <= a0 < self.N:
                one = self._one_anchor_search(node_type, k, ("f64rows", mode))
                ids, sc, flag, d2 = one.run(a0)
                if flag > 0:
                    return ids.reshape(1, k), sc.reshape(1, k), (np.sqrt(d2).reshape(1, k) if d2 is not None else None), np.asarray([flag], dtype=np.int32)
        Q64, Q32, a = self.build_queries(anchors, positive, negative)
        # (round 6, measured and not kept: every output of a small batch in ONE device buffer and one device-to-host copy instead of three - 943 anchors 0.208 ->
        # 0.176 ms in a long-lived process, but 0.3 - 0.7 ms for the first dozens of calls of a fresh one: a single fresh 3 MB host array per call sits on the host
        # allocator's mmap / page-fault path until its thresholds adapt, two 1.5 MB arrays do not; profiles/r6_probe_one_copy.txt)
        fh = []
        idx, score, dist2, flags = self.search_queries(node_type, Q64, Q32, k, a, mode, sort_by_score, want_dist=want_dist, _flags_host_out=fh)
        sc = score.cpu().numpy()
        if mode == 1:
            sc = sc.astype(np.float32)  # exact: the kernel widened float32 values
        flags_np = fh[0].numpy() if (fh and fh[0] is not None) else flags.cpu().numpy()   # (_search_f32 already fetched them)
        return idx.cpu().numpy(), sc, (np.sqrt(dist2.cpu().numpy()) if dist2 is not None else None), flags_np
