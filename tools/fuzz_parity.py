#!/usr/bin/env python3
"""Shape fuzz (GPU box only): random (B, N, d, k) including the thresholds between the search paths (8192 / 8193 items,
32 / 33 / 64 / 65 / 1024 / 1025 queries, k at both ends, dims that are not multiples of anything), both score flavours,
with and without distances, against the oracle's C restatement on every query.

    python tools/fuzz_parity.py [--cases 60] [--seed 0]
    python tools/fuzz_parity.py --f64rows [--cases 40]      # (round 6) GENUINE float64 rows: the fused float64-row search against the numpy oracle on the
                                                            # float64 rows (sampled queries) and against the candidate scheme of rounds 3-5 (every query)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32).astype(np.float64)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def fuzz_f64rows(args):
    from hwer.device_index import DeviceIndex
    from oracle import hwer_oracle as orc
    rng = np.random.default_rng(args.seed + 64)
    Bs = [1, 2, 31, 33, 64, 65, 128, 257, 1000, 1024, 1025, 2049, 3000]
    Ns = [1, 5, 32, 33, 1000, 4095, 8192, 8193, 12000, 20000, 40000]
    ds = [1, 3, 7, 8, 9, 17, 32, 33, 64, 66, 100, 128, 129, 200, 256]
    ks = [1, 2, 10, 64, 100, 200, 256, 392]
    bad = 0
    for case in range(args.cases):
        while True:
            B, N, d = int(rng.choice(Bs)), int(rng.choice(Ns)), int(rng.choice(ds))
            k = int(rng.choice([x for x in ks if x <= N]))
            if float(B) * N * d <= 2e9:
                break
        X = rng.standard_normal((B + N, d))
        X /= np.linalg.norm(X, axis=1, keepdims=True)
        if d > 1 and rng.random() < 0.3:
            X[B:] *= rng.uniform(0.5, 3.0, size=(N, 1))
        if N > 20 and rng.random() < 0.5:      # float32-colliding rows and an exact duplicate
            X[B + 11] = X[B + 10] * (1.0 + 3e-9)
            X[B + 12] = X[B + 10]
        ix = DeviceIndex(X, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, audit=False)
        if ix.V64 is None:      # (d = 1: unit rows are +-1, float32-representable - the index keeps float32 rows; covered by the other mode)
            print("case %3d  B=%5d N=%6d d=%3d: rows are float32-representable, skipped" % (case, B, N, d), flush=True)
            continue
        scoring = "gcn" if rng.random() < 0.25 else "base"
        want = bool(rng.integers(0, 2)) or scoring == "gcn"
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k, scoring=scoring, want_dist=want)
        ok = bool((flags > 0).all())
        for u in rng.choice(B, size=min(B, 4), replace=False).tolist():
            o_idx, o_sc, o_dist = orc.find_closest(X, np.arange(B, B + N), int(u), k=k, scoring=scoring)
            ok = ok and np.array_equal(idx[u], o_idx) and np.array_equal(score[u], o_sc) and (not want or np.array_equal(dist[u], o_dist))
        os.environ["HWER_F64ROWS_CANDIDATES"] = "1"
        try:
            c_idx, c_score, c_dist, _ = ix.search("item", np.arange(B), k=k, scoring=scoring, want_dist=True)
        finally:
            del os.environ["HWER_F64ROWS_CANDIDATES"]
        ok = ok and np.array_equal(idx, c_idx) and np.array_equal(score, c_score) and (not want or np.array_equal(dist, c_dist))
        print("case %3d  B=%5d N=%6d d=%3d k=%3d float64 rows %-4s dist=%d  exhaustive=%d  %s" % (case, B, N, d, k, scoring, want, int((flags == 2).sum()),
                                                                                                "OK" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        del ix
    if bad:
        raise SystemExit("%d case(s) differ" % bad)
    print("all %d float64-row cases agree with the numpy oracle (sampled queries) and the candidate scheme (every query)" % args.cases)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--f64rows", action="store_true")
    args = ap.parse_args()
    if args.f64rows:
        return fuzz_f64rows(args)
    from hwer.device_index import DeviceIndex
    from oracle import c_oracle
    rng = np.random.default_rng(args.seed)
    Bs = [1, 2, 31, 32, 33, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1024, 1025, 3000]
    Ns = [1, 5, 31, 32, 33, 1000, 4095, 8191, 8192, 8193, 8300, 12000, 20000, 40000, 70000]
    ds = [1, 3, 8, 17, 31, 32, 33, 64, 66, 100, 128, 129, 200, 256]
    ks = [1, 2, 10, 64, 100, 200, 256, 392]
    cores = c_oracle.effective_cpus()
    bad = 0
    for case in range(args.cases):
        while True:
            B, N, d = int(rng.choice(Bs)), int(rng.choice(Ns)), int(rng.choice(ds))
            k = int(rng.choice([x for x in ks if x <= N]))
            if float(B) * N * d <= 4e9:
                break
        arith = int(rng.integers(0, 2))
        V32 = np.concatenate([unit_rows(rng, B, d), unit_rows(rng, N, d)])
        if d > 1 and rng.random() < 0.3:      # non-unit rows are legal for the k-NN itself (MultiKNN.query on arbitrary vectors)
            V32[B:] *= rng.uniform(0.5, 3.0, size=(N, 1)).astype(np.float32)
        V = V32.astype(np.float64) if arith == 0 else V32
        ix = DeviceIndex(V, {"user": np.arange(B, dtype=np.int64), "item": np.arange(B, B + N, dtype=np.int64)}, arith=arith, audit=False)
        scoring = "gcn" if rng.random() < 0.2 else "base"
        want = bool(rng.integers(0, 2)) or scoring == "gcn"
        idx, score, dist, flags = ix.search("item", np.arange(B), k=k, scoring=scoring, want_dist=want)
        Q64, _, _ = ix.build_queries(np.arange(B))
        A = V32[:B].astype(np.float64)
        o_idx, o_sc, o_d2 = c_oracle.find_closest_f64(V32[B:], Q64.cpu().numpy(), None if scoring == "gcn" else A, k, threads=cores)
        from oracle import hwer_oracle as orc
        if arith == 0 or scoring == "gcn":
            ok = np.array_equal(idx, o_idx + B) and (flags > 0).all()
        else:
            # float32 flavour: the order follows the float32 scores (the C oracle scores in float64), so compare the
            # neighbour SETS on every query and the exact order / scores against the numpy oracle on a few
            ok = np.array_equal(np.sort(idx, axis=1), np.sort(o_idx + B, axis=1)) and (flags > 0).all()
            for u in rng.choice(B, size=min(B, 3), replace=False).tolist():
                n_idx, n_sc, n_dist = orc.find_closest(V, np.arange(B, B + N), int(u), k=k)
                ok = ok and np.array_equal(idx[u], n_idx) and np.array_equal(score[u], n_sc)
        if scoring == "gcn":
            ok = ok and np.array_equal(score, (2.0 - np.sqrt(o_d2)) / 2.0 if arith == 0 else score)   # float32 flavour checked by the unit tests
        elif arith == 0:
            ok = ok and np.array_equal(score, o_sc)
        if want and (arith == 0 or scoring == "gcn"):
            ok = ok and np.array_equal(dist, np.sqrt(o_d2))
        print("case %3d  B=%5d N=%6d d=%3d k=%3d arith=%d %-4s dist=%d  exhaustive=%d  %s" % (case, B, N, d, k, arith, scoring, want,
                                                                                             int((flags == 2).sum()), "OK" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
        del ix
    if bad:
        raise SystemExit("%d case(s) differ from the oracle" % bad)
    print("all %d cases agree with the oracle" % args.cases)


if __name__ == "__main__":
    main()
