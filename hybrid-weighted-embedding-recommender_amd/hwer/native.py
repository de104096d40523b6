"""ctypes binding of libhwer_hip.so (C ABI: include/hwer_hip.h).

The HIP library is the product path.  There is no CPU fallback: `lib()` raises if the shared object is
missing, and every device entry point raises on a non-zero status.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libhwer_hip.so")

HWER_OK, HWER_EINVAL, HWER_ELAUNCH, HWER_EWORKSPACE = 0, -1, -2, -3

_P = c_void_p
# name -> (restype, argtypes); mirrors include/hwer_hip.h one to one
PROTOTYPES = {
    "hwer_abi_version": (c_int, []),
    "hwer_last_error": (c_char_p, []),
    "hwer_stream_synchronize": (c_int, [_P]),
    "hwer_selftest_wave_ops": (c_int, [c_int, ctypes.c_uint]),
    "hwer_profile_enable": (c_int, [c_int]),
    "hwer_profile_read": (c_int, [c_int, _P, _P]),
    "hwer_pair_score_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, _P]),
    "hwer_pair_score_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, _P]),
    "hwer_segment_mean_f32": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, _P, c_int64, _P]),
    "hwer_blend_normalize_f32": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int64, c_int, _P, c_int64, _P, _P]),
    "hwer_blend_normalize_audit_f32": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int64, c_int, _P, c_int64, c_float, _P, _P, _P]),
    "hwer_content_tail_f32": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, _P, c_int64, _P]),
    "hwer_segment_mean_f64": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, _P, c_int64, _P]),
    "hwer_content_tail_f64": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, _P, c_int64, _P]),
    "hwer_blend_normalize_f64": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int64, c_int, _P, c_int64, _P, _P]),
    "hwer_unit_audit_ws_floats": (c_int64, [c_int64]),
    "hwer_knn_count_uncertified": (c_int, [_P, c_int64, _P, _P]),
    "hwer_unit_audit_f32": (c_int, [_P, c_int64, c_int, c_int64, c_float, _P, _P, _P, _P]),
    "hwer_knn_padded_dim": (c_int, [c_int]),
    "hwer_knn_padded_rows": (c_int64, [c_int64]),
    "hwer_pack_rows_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, _P, c_int64, _P, _P]),
    "hwer_pack_rows_f32_f16": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int, _P]),
    "hwer_query_mix": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, c_int, _P, _P, c_int64, _P]),
    "hwer_knn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "hwer_knn_query_padding": (c_int64, [c_int64, c_int64, c_int, c_int, ctypes.c_uint32]),
    "hwer_prepare_search_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int, _P, _P, _P, _P, _P, c_int64, c_int,
                                        _P, _P, c_int64, _P, c_int64, _P]),
    "hwer_knn_workspace_init": (c_int, [_P, c_size_t, _P]),
    # tables, ld, row_idx, alpha, per_row, T, d, V, ldv, eps, counts, absdev, pack_ids, pack_row0, n_rows, other_ids, other_row0, n_other,
    # X, ldx, hnorm, X16, x_exp, anchor, B, arith, Q64, Q32, ldq32, workspace, bq, stream
    "hwer_blend_prepare_search_f32": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_int, _P, c_int64, c_float, _P, _P, _P, c_int64, c_int64,
                                              _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int, _P, c_int64, c_int, _P, _P, c_int64, _P,
                                              c_int64, _P]),
    "hwer_knn_max_dim": (c_int, []),
    "hwer_knn_max_k": (c_int, []),
    "hwer_knn_search_f32": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                    c_int, _P, c_int64, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hwer_knn_workspace_bytes_opts": (c_size_t, [c_int64, c_int64, c_int, c_int, ctypes.c_uint32]),
    "hwer_knn_search_opts_f32": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int, c_int,
                                         c_int, _P, c_int64, c_float, _P, _P, _P, _P, _P, c_size_t, ctypes.c_uint32, _P, c_int, _P]),
    "hwer_knn_screen_f32": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, c_int64, c_int, _P, c_size_t, ctypes.c_uint32, _P, c_int, _P]),
    "hwer_knn_select_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, c_int64, c_float,
                                    _P, _P, _P, _P, _P, c_size_t, ctypes.c_uint32, _P, _P]),
    # X, hnorm, n, d, ldx, Q32, Q64, X64, ld64, x64_rows, A64, lda64, a_idx, B, k, mode, sort, row_ids, id_base, hnorm_max, idx, score, dist2, flags, ws, ws_bytes, opts, X16, x16_exp, stream
    "hwer_knn_search_f64rows": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, c_int64,
                                        c_float, _P, _P, _P, _P, _P, c_size_t, ctypes.c_uint32, _P, c_int, _P]),
    "hwer_knn_select_f64rows": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, c_int64,
                                        c_float, _P, _P, _P, _P, _P, c_size_t, ctypes.c_uint32, _P, _P]),
    # X64, ld64, x64_rows, n, d, Q64, A64, lda64, a_idx, k, mode, sort, row_ids, id_base, idx, score, dist2, flags, knn_ws, missed, ws, ws_bytes, stream
    "hwer_knn_exact_device_f64rows": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, c_int64, _P, c_int, c_int, c_int, _P, c_int64, _P, _P, _P, _P, _P, _P,
                                              _P, c_size_t, _P]),
    "hwer_knn_exact_f64rows": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, c_int64, _P, _P, c_int64, c_int, c_int, c_int, _P, c_int64, _P, _P, _P, _P,
                                       _P, c_size_t, _P]),
    "hwer_pack_rows_f64rows": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, _P, c_int64, _P, _P, c_int, c_float, _P, _P, _P]),
    "hwer_audit_pack_rows_f64rows": (c_int, [_P, c_int64, c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int, c_float, _P, _P, _P]),
    "hwer_query_mix_f64rows": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P]),
    "hwer_query_mix_frag_f64rows": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int64, _P, c_int64, _P]),
    "hwer_pair_score_f64rows": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, _P]),
    "hwer_knn_cand_dist2": (c_int, [_P, c_int, c_int64, c_int, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P]),
    "hwer_knn_f16_scale_exp": (c_int, [c_float]),
    "hwer_knn_f16_screen_pays": (c_int, [c_int64, c_int64, c_int]),
    "hwer_pack_rows_f16": (c_int, [_P, c_int64, c_int64, c_int, _P, _P]),
    "hwer_knn_exact_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "hwer_knn_exact_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int, c_int, c_int,
                                   _P, c_int64, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hwer_knn_exact_device_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "hwer_knn_exact_device": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_int64, _P, c_int, c_int, c_int,
                                      _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "hwer_rank_filter_seen": (c_int, [_P, c_int64, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int64, c_int64, _P]),
    "hwer_rank_metrics": (c_int, [_P, c_int, _P, _P, _P, c_int64, _P, c_int, _P, c_int, _P, _P, _P, _P]),
    "hwer_merge_topk": (c_int, [_P, _P, _P, c_int, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "hwer_knn_all_dist2_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, c_int64, _P, _P]),
    "hwer_pack_exchange": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P]),
    "hwer_gram_padded_dim": (c_int, [c_int]),
    "hwer_segment_gram_workspace_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "hwer_segment_gram_f64": (c_int, [_P, c_int64, c_int, _P, c_int, c_int64, _P, _P, c_size_t, _P]),
    "hwer_pca_padded_components": (c_int, [c_int]),
    "hwer_pca_project_unit_f64": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int64, _P, c_int, _P, c_int, c_int, _P, c_int64,
                                          _P, c_int64, _P]),
    "hwer_knn_keys_f32": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, _P]),
    "hwer_knn_topk_rows_f32": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P, _P]),
    "hwer_knn_topk_rows_f64": (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P]),
    "hwer_knn_order_rows_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    # values, ld, gather, ld_gather, ids, ld_ids, B, n_in, n_out, descending, out_values, out_ids, id_add, id_map, pay_i, out_pay_i, pay_f, out_pay_f,
    # bound, out_flags, every_row, workspace, workspace_bytes, stream
    "hwer_knn_order_rows_f64": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, c_int, _P, _P, c_int64, _P, _P, _P, _P, _P,
                                        _P, _P, c_int, _P, c_size_t, _P]),
    "hwer_merge_topk_strided": (c_int, [_P, _P, _P, c_int64, c_int, c_int64, c_int, c_int, _P, _P, _P, _P]),
    "hwer_merge_topk_lists": (c_int, [_P, _P, _P, c_int64, c_int, c_int64, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
}

_lib = None


class HwerNativeError(RuntimeError):
    pass


def lib():
    """Load the library (once).  Raises loudly when it has not been built — no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HwerNativeError(
                "libhwer_hip.so not found at %s — build it with "
                "`python hybrid-weighted-embedding-recommender_amd/csrc/build.py` (or __graft_entry__.build())" % LIB_PATH)
        # torch first: its wheel bundles the HIP runtime (libamdhip64.so.7) that owns every device pointer we are
        # handed; loading ours afterwards binds to that already-loaded soname instead of a second runtime.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        if handle.hwer_abi_version() != 1:
            raise HwerNativeError("libhwer_hip.so ABI version %d != 1" % handle.hwer_abi_version())
        _lib = handle
    return _lib


def check(rc, what=""):
    if rc == HWER_OK:
        return
    msg = lib().hwer_last_error().decode("utf-8", "replace")
    if rc == HWER_EINVAL and "k must be less than or equal" in msg:
        # same exception type and text as scikit-learn's KDTree.query (reference behaviour)
        raise ValueError("k must be less than or equal to the number of training points")
    kind = {HWER_EINVAL: "invalid argument", HWER_ELAUNCH: "HIP launch failure", HWER_EWORKSPACE: "workspace"}.get(rc, "error")
    raise HwerNativeError("%s: %s (%d): %s" % (what or "hwer native call", kind, rc, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Current torch stream of `device` (default: the current device).  Native calls launch on the CURRENT HIP device, so
    callers that were handed an explicit device wrap their calls in `device_guard(device)`."""
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def device_guard(device):
    """Context manager: make `device` the current HIP device for the native calls inside (kernels launch on the current
    device; a tensor on cuda:1 handed to a launch on cuda:0 faults or silently relies on peer access)."""
    import torch
    return torch.cuda.device(device)


def on_device_kw(fn):
    """Decorator for host-mirror functions with a `device=None` keyword: resolve it once and run the body with that device
    current, so that every allocation, launch and stream query inside refers to the same GPU."""
    import functools
    import inspect
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        import torch
        if not torch.cuda.is_available():
            raise HwerNativeError("hwer (MI355X build) needs a ROCm GPU: torch.cuda.is_available() is False; "
                                  "there is no CPU fallback for this path")
        bound = sig.bind_partial(*a, **kw)     # `device` may arrive positionally or by keyword
        device = bound.arguments.get("device")
        dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        bound.arguments["device"] = dev
        with torch.cuda.device(dev):
            return fn(*bound.args, **bound.kwargs)
    return wrapped
