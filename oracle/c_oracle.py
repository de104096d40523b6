"""ctypes access to the C part of the oracle (oracle/hwer_oracle.c).  TEST INFRASTRUCTURE ONLY (see that file)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libhwer_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            import sys
            torch = sys.modules.get("torch")
            if torch is not None and torch.cuda.is_initialized():
                # a fork + exec from a process that has initialised the GPU is refused on the GPU boxes (and takes the machine
                # down where it is not): build the checker BEFORE the first GPU call
                raise RuntimeError("oracle/_build/libhwer_oracle.so is missing and the GPU is already initialised: run `make -C oracle` "
                                   "(or __graft_entry__.build()) first, or call oracle.c_oracle.lib() before the first GPU call")
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = ctypes.CDLL(_LIB)
        P, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.orc_find_closest_f64.restype = i32
        L.orc_find_closest_f64.argtypes = [P, i64, i32, P, P, i64, i32, i32, i32, P, P, P]
        L.orc_find_closest_f64_scalar.restype = i32
        L.orc_find_closest_f64_scalar.argtypes = L.orc_find_closest_f64.argtypes
        L.orc_pair_score_f64.restype = i32
        L.orc_pair_score_f64.argtypes = [P, i64, i32, P, P, i64, i32, P]
        L.orc_num_threads.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def find_closest_f64(X32, Q64, A64, k, sort_by_score=True, threads=1, scalar=False):
    """X32 (N,d) float32 item rows; Q64 (B,d) queries; A64 (B,d) anchor rows or None (GcnNCF scoring).
    -> (local idx (B,k) int64, score (B,k) f64, dist2 (B,k) f64).  scalar=True: the one-pair-at-a-time restatement
    (orc_find_closest_f64_scalar) instead of the row-blocked loop - same values, bit for bit."""
    X32 = np.ascontiguousarray(X32, dtype=np.float32)
    Q64 = np.ascontiguousarray(Q64, dtype=np.float64)
    A = None if A64 is None else np.ascontiguousarray(A64, dtype=np.float64)
    B, d = Q64.shape
    idx = np.empty((B, k), np.int64)
    sc = np.empty((B, k), np.float64)
    d2 = np.empty((B, k), np.float64)
    fn = lib().orc_find_closest_f64_scalar if scalar else lib().orc_find_closest_f64
    rc = fn(_p(X32), X32.shape[0], d, _p(Q64), None if A is None else _p(A), B, k,
                                    1 if sort_by_score else 0, threads, _p(idx), _p(sc), _p(d2))
    if rc != 0:
        raise ValueError("k must be less than or equal to the number of training points")
    return idx, sc, d2


def pair_score_f64(V32, src, dst, threads=1):
    V32 = np.ascontiguousarray(V32, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    out = np.empty(len(src), np.float64)
    lib().orc_pair_score_f64(_p(V32), V32.shape[0], V32.shape[1], _p(src), _p(dst), len(src), threads, _p(out))
    return out


def num_threads():
    return lib().orc_num_threads()


def effective_cpus():
    """Threads worth starting: min(os.cpu_count(), affinity mask, cgroup CPU quota).  A GPU box shows all 256 host CPUs but
    caps the container at 16: OpenMP with 256 threads then runs several times SLOWER than with 16."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n
