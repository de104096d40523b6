"""ContentRecommendation — drop-in for the reference's hwer/content_recommender.py.

Hot path (device): the tail of __build_content_embeddings__ (reference lines 77-86) and the unit_length of fit
(line 96) run as ONE fused HIP kernel (hwer_content_tail_f32: CSR neighbour mean -> (E + A)/2 -> L2 normalise),
or as the general multi-table, per-row alpha blend (blend_tables / hwer_blend_normalize_f32).
The table construction before it (reference lines 50-76: block-diagonal concat, one-hot node type, IncrementalPCA, unit
rows) runs on the device too (hwer/fit_pca.py); only the per-type feature embedders (lines 24-49, out of
scope) stay on the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Set

import numpy as np

from . import native
from .logging import getLogger
from .recommendation_base import Edge, FeatureName, Node, NodeType, RecommendationBase


def adjacency_csr(n_nodes: int, src: np.ndarray, dst: np.ndarray):
    """CSR of the reference's adjacency lists (hwer/content_recommender.py:77-81): the real edges first, then one
    self loop per node; every (s, t) appends t to adj[s] and s to adj[t] — self loops list the node twice,
    parallel edges repeat, weights are ignored.  Append order is preserved (stable sort by owner)."""
    loops = np.arange(n_nodes, dtype=np.int64)
    s = np.concatenate([np.asarray(src, dtype=np.int64), loops])
    t = np.concatenate([np.asarray(dst, dtype=np.int64), loops])
    owner = np.stack([s, t], axis=1).ravel()
    other = np.stack([t, s], axis=1).ravel()
    order = np.argsort(owner, kind="stable")
    indptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(owner, minlength=n_nodes), out=indptr[1:])
    return indptr, other[order].astype(np.int32)


def _dev_table(t, device, ld):
    import torch
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32))
    t = t.to(device=device, dtype=torch.float32)
    if t.shape[1] == ld and t.is_contiguous():
        return t
    out = torch.zeros((t.shape[0], ld), dtype=torch.float32, device=device)
    out[:, :t.shape[1]] = t
    return out


def _is_f64(t) -> bool:
    """float64 input (numpy or torch): the blend stage then runs in the reference's own float64 arithmetic (hwer_*_f64)."""
    if isinstance(t, np.ndarray):
        return t.dtype == np.float64
    return str(getattr(t, "dtype", "")) == "torch.float64"


def _dev_table64(t, device, ld):
    import torch
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.float64))
    t = t.to(device=device, dtype=torch.float64)
    if t.shape[1] == ld and t.is_contiguous():
        return t
    out = torch.zeros((t.shape[0], ld), dtype=torch.float64, device=device)
    out[:, :t.shape[1]] = t
    return out


@native.on_device_kw
def content_tail(E, src, dst, device=None, return_device=False, d=None):
    """unit_rows((E + mean_adj(E)) / 2) on the GPU.  E: (N, D) array; src/dst: edge endpoints as row ids.
    d: logical width when E is a device table whose rows are already padded to a multiple of 4.
    A float64 table runs hwer_content_tail_f64 - the reference's float64 arithmetic in its summation orders: the result equals
    hwer/content_recommender.py:77-86 + :96 bit for bit (float64 out); a float32 table runs the float32 kernel (float32 out)."""
    import torch
    from .device_index import _require_gpu, roundup
    lib = native.lib()
    dev = _require_gpu(device)
    N, D = E.shape
    if d is not None:
        assert roundup(d, 4) == D, "a pre-padded table must have roundup4(d) columns"
        D = d
    ld = roundup(D, 4)
    f64 = _is_f64(E)
    Ed = _dev_table64(E, dev, ld) if f64 else _dev_table(E, dev, ld)
    indptr, indices = adjacency_csr(N, src, dst)
    ip = torch.from_numpy(indptr).to(dev)
    ix = torch.from_numpy(indices).to(dev)
    out = torch.zeros((N, ld), dtype=torch.float64 if f64 else torch.float32, device=dev)
    fn = lib.hwer_content_tail_f64 if f64 else lib.hwer_content_tail_f32
    native.check(fn(native.ptr(Ed), ld, native.ptr(ip), native.ptr(ix), N, D, native.ptr(out), ld, native.stream_ptr()), "content_tail")
    return out if return_device else out[:, :D].cpu().numpy()


@native.on_device_kw
def segment_mean(E, indptr, indices, device=None):
    """A[n] = mean of E rows listed in CSR row n (reference: hwer/content_recommender.py:83)."""
    import torch
    from .device_index import _require_gpu, roundup
    lib = native.lib()
    dev = _require_gpu(device)
    N, D = E.shape
    ld = roundup(D, 4)
    f64 = _is_f64(E)       # float64 table: the reference's float64 mean(0), bit for bit (hwer_segment_mean_f64)
    Ed = _dev_table64(E, dev, ld) if f64 else _dev_table(E, dev, ld)
    ip = torch.from_numpy(np.ascontiguousarray(indptr, dtype=np.int64)).to(dev)
    ix = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32)).to(dev)
    out = torch.zeros((N, ld), dtype=torch.float64 if f64 else torch.float32, device=dev)
    fn = lib.hwer_segment_mean_f64 if f64 else lib.hwer_segment_mean_f32
    native.check(fn(native.ptr(Ed), ld, native.ptr(ip), native.ptr(ix), N, D, native.ptr(out), ld, native.stream_ptr()), "segment_mean")
    return out[:, :D].cpu().numpy()


@native.on_device_kw
def blend_tables(tables: Sequence, alpha, row_idx: Optional[Sequence] = None, device=None, return_device=False,
                 return_norm=False, dtype=None):
    """out[r] = unit(sum_t alpha[r, t] * tables[t][row_idx[t][r]]) — the weighted blend of content and
    collaborative tables (north_star; reference special case hwer/content_recommender.py:85 with alpha=(.5,.5)).
    alpha: (T,) shared weights or (R, T) per-row weights.  row_idx[t]: None (row r) or an int array (R,).
    float64 tables (all of them) run hwer_blend_normalize_f64: float64 weights, products and sums in the order of
    `acc = acc + alpha_t * T_t`, numpy's norm - float64 rows out, identical to that numpy expression.
    dtype ("float32" | "float64" | None): the arithmetic, explicitly (tables of the other dtype are converted); None infers it from the
    tables - float64 only when EVERY table is float64; a mix of float32 and float64 tables runs the float32 arithmetic and warns
    (round 6, advisor).  float64 rows are answered by the index from the float64 table (DeviceIndex: the fused float64-row search) -
    same speed class as float32 rows since round 6, twice the gathered bytes."""
    import torch
    from .device_index import _require_gpu, roundup
    lib = native.lib()
    dev = _require_gpu(device)
    T = len(tables)
    D = tables[0].shape[1]
    ld = roundup(D, 4)
    kinds = {bool(_is_f64(t)) for t in tables}
    if dtype is not None:
        f64 = np.dtype(dtype) == np.float64
        if not f64 and np.dtype(dtype) != np.float32:
            raise ValueError("blend_tables: dtype must be float32 or float64")
    else:
        f64 = kinds == {True}
        if len(kinds) > 1:
            import warnings
            warnings.warn("blend_tables: float32 and float64 tables mixed - running the float32 arithmetic; pass dtype='float64' (or 'float32') to choose",
                          RuntimeWarning, stacklevel=3)
    if f64:
        return _blend_tables_f64(tables, alpha, row_idx, dev, ld, return_device, return_norm)
    tabs = [_dev_table(t, dev, ld) for t in tables]
    alpha = np.asarray(alpha, dtype=np.float32)
    per_row = alpha.ndim == 2
    idx = [None] * T if row_idx is None else [None if r is None else
                                               torch.from_numpy(np.ascontiguousarray(r, dtype=np.int32)).to(dev) for r in row_idx]
    R = alpha.shape[0] if per_row else next((len(r) for r in (row_idx or []) if r is not None), tabs[0].shape[0])
    tab_ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
    has_idx = any(i is not None for i in idx)
    idx_ptrs = torch.tensor([0 if i is None else i.data_ptr() for i in idx], dtype=torch.int64, device=dev) if has_idx else None
    al = torch.from_numpy(np.ascontiguousarray(alpha)).to(dev)
    out = torch.zeros((R, ld), dtype=torch.float32, device=dev)
    norm = torch.empty(R, dtype=torch.float32, device=dev) if return_norm else None
    native.check(lib.hwer_blend_normalize_f32(native.ptr(tab_ptrs), ld, native.ptr(idx_ptrs), native.ptr(al), 1 if per_row else 0,
                                              T, R, D, native.ptr(out), ld, native.ptr(norm), native.stream_ptr()),
                 "blend_normalize")
    res = out if return_device else out[:, :D].cpu().numpy()
    return (res, norm if return_device else norm.cpu().numpy()) if return_norm else res


def _blend_tables_f64(tables, alpha, row_idx, dev, ld, return_device, return_norm):
    import torch
    lib = native.lib()
    T, D = len(tables), tables[0].shape[1]
    tabs = [_dev_table64(t, dev, ld) for t in tables]
    alpha = np.asarray(alpha, dtype=np.float64)
    per_row = alpha.ndim == 2
    idx = [None] * T if row_idx is None else [None if r is None else
                                               torch.from_numpy(np.ascontiguousarray(r, dtype=np.int32)).to(dev) for r in row_idx]
    R = alpha.shape[0] if per_row else next((len(r) for r in (row_idx or []) if r is not None), tabs[0].shape[0])
    tab_ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
    has_idx = any(i is not None for i in idx)
    idx_ptrs = torch.tensor([0 if i is None else i.data_ptr() for i in idx], dtype=torch.int64, device=dev) if has_idx else None
    al = torch.from_numpy(np.ascontiguousarray(alpha)).to(dev)
    out = torch.zeros((R, ld), dtype=torch.float64, device=dev)
    norm = torch.empty(R, dtype=torch.float64, device=dev) if return_norm else None
    native.check(lib.hwer_blend_normalize_f64(native.ptr(tab_ptrs), ld, native.ptr(idx_ptrs), native.ptr(al), 1 if per_row else 0,
                                              T, R, D, native.ptr(out), ld, native.ptr(norm), native.stream_ptr()),
                 "blend_normalize_f64")
    res = out if return_device else out[:, :D].cpu().numpy()
    return (res, norm if return_device else norm.cpu().numpy()) if return_norm else res


class ContentRecommendation(RecommendationBase):
    def __init__(self, embedding_mapper: Dict[NodeType, Dict[str, object]], node_types: Set[str], n_dims: int = 32):
        super().__init__(node_types=node_types, n_dims=n_dims)
        self.embedding_mapper = embedding_mapper
        self.log = getLogger(type(self).__name__)

    def __build_feature_blocks__(self, nodes: List[Node], node_data):
        """Host part of the fit (reference hwer/content_recommender.py:24-49): run the per-type embedders (out of scope,
        SURVEY.md §2) -> one float64 block of unit rows per node type, in SORTED type order (the OneHotEncoder's category
        order), rows in `nodes` order; plus the order in which the types were visited (the column order of the
        reference's block-diagonal table) and every node's type id."""
        from .utils import unit_length
        cats = sorted({n.node_type for n in nodes})
        cat_id = {t: i for i, t in enumerate(cats)}
        blocks, visit = [None] * len(cats), []
        for nt in self.node_types:
            members = [n for n in nodes if n.node_type == nt]
            with_data = sum(1 for n in members if n in node_data)
            assert with_data in (0, len(members))
            assert len(set(members)) == len(members)
            if with_data == 0:
                block = np.zeros((len(members), 1))
            else:
                feats = [node_data[n] for n in members]
                names = list(feats[0].keys())
                # like the reference, only the FIRST feature of a node type contributes (its concatenate result is
                # discarded at hwer/content_recommender.py:47); the other embedders are still fitted
                block = None
                for f in names:
                    emb = self.embedding_mapper[nt][f].fit_transform([x[f] for x in feats])
                    if block is None:
                        block = emb
                block = unit_length(block, axis=1)
            if nt in cat_id:
                blocks[cat_id[nt]] = np.asarray(block, dtype=np.float64)
                visit.append(cat_id[nt])
        type_id = np.fromiter((cat_id[n.node_type] for n in nodes), dtype=np.int64, count=len(nodes))
        return blocks, type_id, visit

    #: arithmetic of the blend stage behind fit(): "float64" = the reference's own (its table is float64 from the PCA on:
    #: hwer/content_recommender.py:68-96), rows come back as genuine float64 values and the index answers from them; "float32" = rounds
    #: 1-4's flavour (float32 rows widened: within 1e-6 of the reference, half the bytes).  Cost of the default: the index keeps the float64 table
    #: beside its float32 rounding and the search gathers float64 rows - since round 6 inside the same fused kernels (hwer_knn_search_f64rows): one
    #: find_closest_neighbours call 0.068 ms against 0.065, all 943 anchors of the ML-100K shape 0.26 ms against 0.28, a 138 493 x 26 744 sweep 1.5x the
    #: float32 flavour's time (twice the gathered bytes); rounds 3-5: ~10x per query.  Set it per instance (`model.blend_dtype = "float32"`) before fit().
    blend_dtype = "float64"

    def __build_feature_table__(self, nodes: List[Node], node_data, n_dims):
        """Reference hwer/content_recommender.py:50-76 on the GPU (hwer/fit_pca.py): the block-diagonal table is never
        built; IncrementalPCA's data passes, the row normalisation and the one-hot columns are HIP kernels.
        -> (device table (N, ld) in `blend_dtype`, its logical width)."""
        from .fit_pca import device_feature_table
        blocks, type_id, visit = self.__build_feature_blocks__(nodes, node_data)
        if self.blend_dtype == "float64":
            _, E64, D = device_feature_table(blocks, type_id, n_dims, block_order=visit, device=self.device, want_f64=True)
            return E64, D
        return device_feature_table(blocks, type_id, n_dims, block_order=visit, device=self.device)

    def __build_content_embeddings__(self, nodes, edges, node_data, n_dims):
        """-> unit-norm (N, D) rows: embedders on the host, PCA + neighbour blend + normalise on the GPU (float64 by default,
        like the reference)."""
        E, D = self.__build_feature_table__(nodes, node_data, n_dims)
        get = self.nodes_to_idx.__getitem__
        src = np.fromiter((get(e.src) for e in edges), dtype=np.int64, count=len(edges))
        dst = np.fromiter((get(e.dst) for e in edges), dtype=np.int64, count=len(edges))
        return content_tail(E, src, dst, device=self.device, d=D)

    def fit(self, nodes: List[Node], edges: List[Edge], node_data: Dict[Node, Dict[FeatureName, object]], **kwargs):
        super().fit(nodes, edges, node_data)
        embeddings = self.__build_content_embeddings__(nodes, edges, node_data, self.n_dims)
        # the reference returns float64 rows: blend_dtype "float64" computed them in float64 (nothing to widen); "float32": the
        # device's float32 rows widened
        embeddings = embeddings.astype(np.float64)
        self.__build_knn__(embeddings)
        self.fit_done = True
        return embeddings

    def fit_tables(self, nodes: List[Node], tables: Sequence, alpha, row_idx=None):
        """Index a weighted blend of precomputed embedding tables (content + collaborative, optionally per-row alpha)."""
        assert not self.fit_done
        self.add_nodes(nodes)
        # like fit(): the model's `vectors` are float64 (the reference's ContentRecommendation flavour: float64 score arithmetic) -
        # float64 tables are blended in float64 (hwer_blend_normalize_f64), float32 tables in float32 and widened
        vec = blend_tables(tables, alpha, row_idx, device=self.device).astype(np.float64)
        self.__build_knn__(vec)
        self.fit_done = True
        return vec


    def blend_search_step(self, nodes: List[Node], tables: Sequence, alpha, node_type: str, anchors: Sequence[Node], k: int = 200,
                          want_dist: bool = False, capture: Optional[bool] = None, **step_kwargs):
        """The whole hot path as ONE resident step (hwer.pipeline.BlendSearchStep): re-blend `tables` with `alpha` -> unit rows ->
        audit -> table of `node_type` -> anchor queries -> exact top-k, every buffer preallocated.  For callers whose tables or
        weights change between retrievals: write new values into `step.tables[t]` / `step.alpha` (device tensors) and call
        `step.run()` again - one fused prologue launch + the search, no allocation, no host round trip; small shapes are replayed as
        a hipGraph (`capture`, default: catalogues up to 8 192 rows).  Registers `nodes` like fit_tables (rows in `nodes` order).
        -> the step; `step.results()` gives (global row ids (B, k), scores (B, k)) in find_closest_neighbours' order and
        `self.knn.idx_to_node()`-style lookups map ids back to Nodes."""
        import torch
        from .device_index import _require_gpu
        from .pipeline import BlendSearchStep
        if not self.nodes_to_idx:
            self.add_nodes(nodes)
        assert len(nodes) == len(self.nodes_to_idx) and node_type in self.node_types
        dev = _require_gpu(self.device)
        with native.device_guard(dev):
            N = len(nodes)
            tabs = [t if (isinstance(t, torch.Tensor) and t.is_cuda) else _dev_table(t, dev, tables[0].shape[1]) for t in tables]
            al = alpha if isinstance(alpha, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(alpha, dtype=np.float32))
            al = al.to(device=dev, dtype=torch.float32).contiguous()
            is_t = np.fromiter((n.node_type == node_type for n in nodes), dtype=bool, count=N)
            rows_t, rows_o = np.flatnonzero(is_t), np.flatnonzero(~is_t)

            def rng(r):      # a contiguous range -> (row0, n, None); else an id list
                if len(r) == 0 or int(r[-1]) - int(r[0]) + 1 == len(r):
                    return (int(r[0]) if len(r) else 0), len(r), None
                return 0, len(r), torch.from_numpy(r.astype(np.int64)).to(dev)
            s0, ns, s_ids = rng(rows_t)
            o0, no, o_ids = rng(rows_o)
            a = np.fromiter((self.nodes_to_idx.get(n, -1) for n in anchors), dtype=np.int32, count=len(anchors))
            step = BlendSearchStep(tabs, al, s0, ns, o0, no, torch.from_numpy(a).to(dev), k=k, want_dist=want_dist, search_ids=s_ids,
                                   other_ids=o_ids, arith=0, **step_kwargs)
            if capture if capture is not None else ns <= 8192:
                step.capture()
        return step


ContentRecommender = ContentRecommendation  # the name BASELINE.json's north_star uses
