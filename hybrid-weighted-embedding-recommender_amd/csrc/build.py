#!/usr/bin/env python3
"""Build libhwer_hip.so (gfx950) in-tree:  python build.py [--force]

hipcc cross-compiles without a GPU.  The library lands in ../lib/ (git-ignored, but it travels with gpurun).
-ffp-contract=off: the re-rank / pair-score kernels reproduce numpy's and scikit-learn's summation orders bit
for bit, which needs products and sums rounded separately; kernels that want an FMA call fmaf() explicitly.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIBDIR = os.path.join(os.path.dirname(HERE), "lib")
# HWER_LIB_NAME=libhwer_hip_diag.so HIPCC_EXTRA=-DHWER_DIAG_STAMPS python build.py: a diagnostic library beside the production one (own objects)
LIB = os.path.join(LIBDIR, os.environ.get("HWER_LIB_NAME", "libhwer_hip.so"))
OBJ_TAG = "" if os.path.basename(LIB) == "libhwer_hip.so" else "." + os.path.basename(LIB).replace(".so", "")
SOURCES = ["rowops.hip", "knn_screen.hip", "knn_small.hip", "knn_select.hip", "knn_select_f64.hip", "fit_pca.hip", "rows_f64.hip", "knn_wide.hip"]
HEADERS = ["hwer_common.h", "knn_internal.h", "knn_select_kernels.h", os.path.join(ROOT, "include", "hwer_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-I" + os.path.join(ROOT, "include"), "-I" + HERE, "-Wno-unused-result"] + os.environ.get("HIPCC_EXTRA", "").split()
# HIPCC_EXTRA=-DHWER_DIAG_STAMPS builds the screen kernel with its s_memtime phase stamps (diagnosis only: the stamps carry
# scheduling barriers; the production build has none)


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [os.path.abspath(__file__)]:
        with open(os.path.join(HERE, f) if not os.path.isabs(f) else f, "rb") as fh:
            h.update(fh.read())
    # (flags with the checkout's own location taken out: the digest of one source tree is the same wherever it is checked out - a GPU box runs
    # the snapshot from a scratch path - so that the build stamp and the profiles' `library_digest` hold there as well)
    h.update(" ".join(FLAGS).replace(ROOT, "<ROOT>").encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = LIB + ".sha256"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", OBJ_TAG + ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
