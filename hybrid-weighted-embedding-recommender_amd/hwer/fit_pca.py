"""Fit-side table construction on the device — the step of ContentRecommendation.fit right before the
neighbour blend: reference hwer/content_recommender.py:50-76.

The reference builds a dense block-diagonal table (N, sum_t D_t) plus one-hot node-type columns and hands it to
`IncrementalPCA(n_components=n_dims, batch_size=2**16).fit_transform`.  Here the table is never built.  A row of node type
t is non-zero only in block t's columns and in one-hot column t, so

  * the statistics IncrementalPCA needs from a batch - its column sums and X^T X - are block diagonal over the types: one
    (D_t + 1) x (D_t + 1) Gram matrix of [B_t | 1] per (batch, type), computed by `hwer_segment_gram_f64` (float64 MFMA)
    over the contiguous row range the batch covers in block t;
  * IncrementalPCA's recurrence (scikit-learn 1.7.2, sklearn/decomposition/_incremental_pca.py partial_fit) is then run on
    W x W matrices (W = table width) on the host: the right singular vectors of the matrix it stacks,
    [S V ; X_b - mean_b ; sqrt(n_seen n_b / n_tot) (mean - mean_b)], are the eigenvectors of
    V^T S^2 V + (X_b^T X_b - n_b mean_b mean_b^T) + corr corr^T, signs fixed like svd_flip(u_based_decision=False);
  * the projection, the row normalisation (hwer/content_recommender.py:70) and the one-hot columns padded to a power of
    two (:71-74) are one kernel per type, `hwer_pca_project_unit_f64`, writing float32 rows straight into the table the
    content tail (`hwer_content_tail_f32`) consumes.

There is no CPU fallback for the data passes: without the HIP library / a GPU these functions raise.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import native

BATCH_SIZE = 2 ** 16  # hwer/content_recommender.py:69


def pca_batches(n: int, batch_size: int, min_batch_size: int):
    """sklearn.utils.gen_batches(n, batch_size, min_batch_size=n_components): a short last batch joins the previous one."""
    out, start = [], 0
    for _ in range(int(n // batch_size)):
        end = start + batch_size
        if end + min_batch_size > n:
            continue
        out.append((start, end))
        start = end
    if start < n:
        out.append((start, n))
    return out


def _flip_rows(V: np.ndarray) -> np.ndarray:
    j = np.argmax(np.abs(V), axis=1)
    return V * np.sign(V[np.arange(V.shape[0]), j])[:, None]


def pca_from_grams(grams: Sequence[np.ndarray], col_of: Sequence[np.ndarray], onehot_col: Sequence[int], width: int,
                   n_components: int):
    """IncrementalPCA's recurrence on per-(batch, type) Gram matrices.
    grams[t]: (n_batches, D_t + 1, D_t + 1) from hwer_segment_gram_f64; col_of[t]: table columns of block t;
    onehot_col[t]: table column of type t's indicator.  Returns (components (n, W), mean (W), singular_values (n))."""
    n_batches = grams[0].shape[0]
    W = width
    if not n_components <= W:
        raise ValueError("n_components=%r invalid for n_features=%d, need more rows than columns for IncrementalPCA "
                         "processing" % (n_components, W))
    n_seen, mean, comps, sv = 0, np.zeros(W), None, None
    for b in range(n_batches):
        XtX = np.zeros((W, W))
        colsum = np.zeros(W)
        nb = 0
        for t, G in enumerate(grams):
            g = G[b]
            D = g.shape[0] - 1
            cols, oc = np.asarray(col_of[t]), onehot_col[t]
            XtX[np.ix_(cols, cols)] = g[:D, :D]
            XtX[cols, oc] = g[:D, D]
            XtX[oc, cols] = g[D, :D]
            XtX[oc, oc] = g[D, D]
            colsum[cols] = g[:D, D]
            colsum[oc] = g[D, D]
            nb += int(round(g[D, D]))
        if nb == 0:
            continue
        if comps is None and n_components > nb:
            raise ValueError("n_components=%d must be less or equal to the batch number of samples %d for the first "
                             "partial_fit call." % (n_components, nb))
        n_tot = n_seen + nb
        col_mean = (mean * n_seen + colsum) / n_tot
        if n_seen == 0:
            M = XtX - nb * np.outer(col_mean, col_mean)
        else:
            bm = colsum / nb
            corr = np.sqrt((n_seen / n_tot) * nb) * (mean - bm)
            M = (comps.T * (sv * sv)) @ comps + (XtX - nb * np.outer(bm, bm)) + np.outer(corr, corr)
        M = 0.5 * (M + M.T)
        lam, vec = np.linalg.eigh(M)
        top = np.argsort(-lam, kind="stable")[:n_components]
        # Rank-deficient tables (e.g. a zero block for a type without features): components beyond the numerical rank span
        # the null space, where ANY orthonormal basis is a valid answer - scikit-learn's SVD returns rounding noise there
        # too, so neither side defines those columns; the unit-row normalisation then amplifies ~1e-8 coordinates of rows
        # whose projection is (nearly) zero.  Parity with the reference only holds up to the numerical rank: say so.
        rank = int(np.sum(lam > lam.max() * W * np.finfo(np.float64).eps * 16.0)) if lam.max() > 0 else 0
        if n_components > rank:
            import warnings
            warnings.warn("device PCA: n_components=%d exceeds the numerical rank %d of the feature table (batch %d); the "
                          "trailing components are an arbitrary null-space basis, as in scikit-learn's IncrementalPCA"
                          % (n_components, rank, b), RuntimeWarning, stacklevel=2)
        comps = _flip_rows(vec[:, top].T)
        sv = np.sqrt(np.maximum(lam[top], 0.0))
        mean, n_seen = col_mean, n_tot
    return comps, mean, sv


@native.on_device_kw
def device_feature_table(blocks: Sequence[np.ndarray], type_id: np.ndarray, n_dims, block_order: Optional[Sequence[int]] = None,
                         device=None, want_f64: bool = False, details: Optional[dict] = None):
    """hwer/content_recommender.py:50-76 on the GPU.
    blocks[t]: (N_t, D_t) float64 unit rows of type t (rows in `nodes` order; a type without features: zeros (N_t, 1));
    type_id: (N,) index of every node's type in the SORTED type names (the OneHotEncoder's category order);
    block_order: order of the blocks' columns in the reference's table (its set iteration order; only rounding depends on it).
    Returns a float32 device tensor (N, roundup4(n_dims + ncol)); columns [0, n_dims) unit PCA rows, then the one-hot type
    padded to ncol = 2^ceil(log2 T) columns.  want_f64: also return the float64 rows (tests)."""
    import torch
    from .device_index import _require_gpu, roundup
    lib = native.lib()
    dev = _require_gpu(device)
    type_id = np.ascontiguousarray(type_id, dtype=np.int64)
    N, T = len(type_id), len(blocks)
    if N == 0 or T == 0:
        raise ValueError("device_feature_table: empty table (N=%d rows, %d node types)" % (N, T))
    order = list(range(T)) if block_order is None else list(block_order)
    dims = [int(b.shape[1]) for b in blocks]
    width = sum(dims)
    W = width + T
    col_of, c = [None] * T, 0
    for t in order:
        col_of[t] = np.arange(c, c + dims[t])
        c += dims[t]
    onehot_col = [width + t for t in range(T)]
    if n_dims is None or np.isinf(n_dims):
        n_dims = 2 ** int(np.log2(W))
    n = int(n_dims)
    if not n <= W:
        raise ValueError("n_components=%r invalid for n_features=%d, need more rows than columns for IncrementalPCA "
                         "processing" % (n, W))
    npad = lib.hwer_pca_padded_components(n)
    if npad < 0:
        raise native.HwerNativeError("device PCA supports 1..256 components, got %d" % n)
    batches = pca_batches(N, BATCH_SIZE, n)
    bounds = np.asarray([a for a, _ in batches] + [batches[-1][1]], dtype=np.int64)
    stream = native.stream_ptr()

    pos, Bd, grams = [], [], []
    for t in range(T):
        p = np.flatnonzero(type_id == t)
        assert len(p) == blocks[t].shape[0], "block %d has %d rows for %d nodes" % (t, blocks[t].shape[0], len(p))
        pos.append(p)
        b = torch.from_numpy(np.ascontiguousarray(blocks[t], dtype=np.float64)).to(dev)
        Bd.append(b)
        seg = np.searchsorted(p, bounds).astype(np.int64)       # rows of block t inside every batch: a contiguous range
        max_rows = int(np.diff(seg).max()) if len(p) else 0
        Da = dims[t] + 1
        G = torch.zeros((len(batches), Da, Da), dtype=torch.float64, device=dev)
        if max_rows > 0:
            seg_d = torch.from_numpy(seg).to(dev)
            ws_bytes = lib.hwer_segment_gram_workspace_bytes(dims[t], len(batches), max_rows)
            ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=dev)
            native.check(lib.hwer_segment_gram_f64(native.ptr(b), b.stride(0), dims[t], native.ptr(seg_d), len(batches), max_rows,
                                                   native.ptr(G), native.ptr(ws), ws_bytes, stream), "segment_gram")
        grams.append(G)
    grams_h = [G.cpu().numpy() for G in grams]
    comps, mean, sv = pca_from_grams(grams_h, col_of, onehot_col, W, n)
    if details is not None:
        details.update(components=comps, mean=mean, singular_values=sv, batches=batches, col_of=col_of, grams=grams_h)

    ncol = 2 ** int(np.ceil(np.log2(T)))
    D_out = n + ncol
    ld = roundup(D_out, 4)
    E32 = torch.empty((N, ld), dtype=torch.float32, device=dev)
    E64 = torch.empty((N, ld), dtype=torch.float64, device=dev) if want_f64 else None
    shift = comps @ mean                                        # (n,)  transform = X V^T - mean V^T
    for t in range(T):
        if len(pos[t]) == 0:
            continue
        d4 = roundup(dims[t], 4)
        P = np.zeros((d4, npad))
        P[:dims[t], :n] = comps[:, col_of[t]].T
        bias = np.zeros(npad)
        bias[:n] = comps[:, onehot_col[t]] - shift
        Pd = torch.from_numpy(P).to(dev)
        bd = torch.from_numpy(bias).to(dev)
        rows_d = torch.from_numpy(pos[t].astype(np.int64)).to(dev)
        native.check(lib.hwer_pca_project_unit_f64(native.ptr(Bd[t]), Bd[t].stride(0), len(pos[t]), dims[t], native.ptr(Pd), npad,
                                                   native.ptr(bd), n, native.ptr(rows_d), t, ncol, native.ptr(E32), ld,
                                                   native.ptr(E64), ld, stream), "pca_project")
    torch.cuda.current_stream().synchronize()                   # P / bias / rows_d go out of scope
    return (E32, E64, D_out) if want_f64 else (E32, D_out)


@native.on_device_kw
def device_pca_unit(vectors: np.ndarray, n_components: int, device=None) -> np.ndarray:
    """unit_length(PCA(n_components).fit_transform(vectors), axis=1) on the GPU - the PCA-if-wider step of
    GcnNCF.prepare_for_knn (reference hwer/gcn_ncf.py:447-456).  Exact PCA through the same two kernels as the fit-side table:
    one Gram matrix of [V | 1] over all rows (float64 MFMA), eigen-decomposition of the d x d covariance on the host, signs like
    scikit-learn's svd_flip(u_based_decision=False), fused projection + row normalisation.  scikit-learn's `full` and
    `covariance_eigh` solvers compute the same thing (in the input's precision: float32 vectors there carry ~1e-6 of rounding
    noise, the device path works in float64); its `randomized` solver (wide or short inputs) is an approximation of it.
    Returns float32 rows (N, n_components)."""
    import torch
    from .device_index import _require_gpu, roundup
    lib = native.lib()
    dev = _require_gpu(device)
    V = np.ascontiguousarray(vectors, dtype=np.float64)
    N, d = V.shape
    n = int(n_components)
    if not 1 <= n <= min(N, d):
        raise ValueError("n_components=%r must be between 1 and min(n_samples, n_features)=%r" % (n, min(N, d)))
    npad = lib.hwer_pca_padded_components(n)
    if npad < 0:
        raise native.HwerNativeError("device PCA supports 1..256 components, got %d" % n)
    stream = native.stream_ptr()
    Bd = torch.from_numpy(V).to(dev)
    seg = torch.tensor([0, N], dtype=torch.int64, device=dev)
    G = torch.zeros((1, d + 1, d + 1), dtype=torch.float64, device=dev)
    ws_bytes = lib.hwer_segment_gram_workspace_bytes(d, 1, N)
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=dev)
    native.check(lib.hwer_segment_gram_f64(native.ptr(Bd), Bd.stride(0), d, native.ptr(seg), 1, N, native.ptr(G), native.ptr(ws),
                                           ws_bytes, stream), "segment_gram")
    g = G[0].cpu().numpy()
    mean = g[:d, d] / N
    C = g[:d, :d] - N * np.outer(mean, mean)
    lam, vec = np.linalg.eigh(0.5 * (C + C.T))
    comps = _flip_rows(vec[:, np.argsort(-lam, kind="stable")[:n]].T)
    P = np.zeros((roundup(d, 4), npad))
    P[:d, :n] = comps.T
    bias = np.zeros(npad)
    bias[:n] = -(comps @ mean)
    Pd, bd = torch.from_numpy(P).to(dev), torch.from_numpy(bias).to(dev)
    ld = roundup(n, 4)
    out = torch.empty((N, ld), dtype=torch.float32, device=dev)
    native.check(lib.hwer_pca_project_unit_f64(native.ptr(Bd), Bd.stride(0), N, d, native.ptr(Pd), npad, native.ptr(bd), n, None, -1, 0,
                                               native.ptr(out), ld, None, 0, stream), "pca_project")
    return out[:, :n].cpu().numpy()
