#!/usr/bin/env python3
"""Micro-benchmarks of the secondary metrics of SURVEY.md §8(d): pair_score / blend_normalize / segment_mean throughput
against the HBM roofline, and per-batch retrieval latency on a C4 shard for B in {1, 8, 64, 512, 4096}.

    python tools/bench_micro.py [--what rowops,latency,fit] [--items 1250000] [--d 128]

Prints one JSON object per measurement.  Inputs are resident in HBM; timing with HIP events on the launch stream.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "hybrid-weighted-embedding-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0


def timeit(torch, fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters   # ms


def unit_rows(torch, n, d, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((n, d), generator=g, device=dev, dtype=torch.float32)
    return x / x.norm(dim=1, keepdim=True)


def hbm_roofline(byts, ms, working_set_bytes, gathered=False):
    """roofline entry of a row kernel.  A working set that fits the 256 MB Infinity Cache (or the 4 MB L2s) is NOT an HBM measurement: its rate is
    reported, a fraction of the HBM peak is not (round-5 review: profiles/r5_bench_micro.txt printed 1.157 of "8 000 GB/s" for an 85 MB table)."""
    rate = byts / ms / 1e6
    if working_set_bytes <= 2.56e8:
        return {"bound": "cache-resident (working set %.0f MB <= the 256 MB Infinity Cache): not an HBM measurement" % (working_set_bytes / 1e6),
                "achieved": rate, "unit": "GB/s", "peak": None, "frac": None}
    return {"bound": "hbm (gathered rows)" if gathered else "hbm", "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="rowops,latency")
    ap.add_argument("--items", type=int, default=1250000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--batches", default="1,8,64,512,4096")
    args = ap.parse_args()
    import torch
    from hwer import native
    lib = native.lib()
    dev = torch.device("cuda", 0)
    P = native.ptr
    S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    what = args.what.split(",")

    if "rowops" in what:
        # ---- pair_score: C2 (P = 1 000 209 pairs over 9 746 rows, d = 128) and a table too large for the caches ------
        for name, N, d, npairs in (("c2", 9746, 128, 1000209), ("c3", 165237, 128, 20000263), ("10M-shard", 2000000, 128, 16000000)):
            V = unit_rows(torch, N, d, dev, 1)
            g = torch.Generator(device=dev).manual_seed(2)
            src = torch.randint(0, N, (npairs,), generator=g, device=dev, dtype=torch.int32)
            dst = torch.randint(0, N, (npairs,), generator=g, device=dev, dtype=torch.int32)
            for flavour, fn, ob in (("f32", lib.hwer_pair_score_f32, 4), ("f64", lib.hwer_pair_score_f64, 8)):
                out = torch.empty(npairs, dtype=torch.float32 if ob == 4 else torch.float64, device=dev)
                ms = timeit(torch, lambda: native.check(fn(P(V), N, d, d, P(src), P(dst), npairs, P(out), S()), "pair_score"))
                byts = npairs * (8.0 * d + 8 + ob)
                print(json.dumps({"kernel": "pair_score_" + flavour, "case": name, "pairs": npairs, "ms": ms, "pairs_per_s": npairs / ms * 1e3,
                                  "roofline": hbm_roofline(byts, ms, 4.0 * N * d, gathered=True)}))
            del V, src, dst, out
        # ---- blend_normalize (+ fused audit): C3 (2 tables, scalar alpha) and C5 (4 tables, per-row alpha, d = 256) -------
        for name, N, d, T, per_row in (("c2", 9746, 128, 2, False), ("c3", 165237, 128, 2, False), ("c5", 1254096, 256, 4, True)):
            tabs = [unit_rows(torch, N, d, dev, 10 + t) for t in range(T)]
            tp = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
            if per_row:
                alpha = torch.rand((N, T), device=dev)
                alpha = (alpha / alpha.sum(1, keepdim=True)).contiguous()
            else:
                alpha = torch.full((T,), 1.0 / T, device=dev)
            out = torch.empty((N, d), dtype=torch.float32, device=dev)
            buf = torch.zeros(6, dtype=torch.int64, device=dev)
            ms = timeit(torch, lambda: native.check(lib.hwer_blend_normalize_audit_f32(P(tp), d, None, P(alpha), 1 if per_row else 0, T, N, d, P(out), d,
                                                                                       1e-4, P(buf[:4]), P(buf[4:5].view(torch.float64)), S()), "blend"))
            byts = N * (4.0 * d * (T + 1) + (4 * T if per_row else 0))
            print(json.dumps({"kernel": "blend_normalize_audit", "case": name, "rows": N, "ms": ms, "rows_per_s": N / ms * 1e3,
                              "roofline": hbm_roofline(byts, ms, 4.0 * N * d * (T + 1))}))
            del tabs, out, alpha
        # ---- segment_mean (content tail): 1 M nodes, 16 neighbours on average ----------------------------------------------
        N, d, deg = 1000000, 128, 16
        E = unit_rows(torch, N, d, dev, 20)
        g = torch.Generator(device=dev).manual_seed(21)
        indices = torch.randint(0, N, (N * deg,), generator=g, device=dev, dtype=torch.int32)
        indptr = (torch.arange(N + 1, device=dev, dtype=torch.int64) * deg).contiguous()
        out = torch.empty((N, d), dtype=torch.float32, device=dev)
        ms = timeit(torch, lambda: native.check(lib.hwer_segment_mean_f32(P(E), d, P(indptr), P(indices), N, d, P(out), d, S()), "segment_mean"))
        byts = 4.0 * d * (N * deg + N) + 8 * N + 4 * N * deg
        print(json.dumps({"kernel": "segment_mean", "rows": N, "nnz": N * deg, "ms": ms,
                          "roofline": {"bound": "hbm (gathered rows)", "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": byts / ms / 1e6 / HBM_PEAK_GBS}}))
        del E, indices, indptr, out
        # ---- round 5: the float64 flavour (the reference's own arithmetic) - same shapes, 8-byte elements --------------------------------
        for name, N, d, T, per_row in (("c3", 165237, 128, 2, False), ("1.25 M rows", 1250000, 128, 2, False), ("c5 shape", 1254096, 256, 4, True)):
            tabs = [unit_rows(torch, N, d, dev, 30 + t).to(torch.float64) for t in range(T)]
            tp = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64, device=dev)
            if per_row:
                alpha = torch.rand((N, T), device=dev, dtype=torch.float64)
                alpha = (alpha / alpha.sum(1, keepdim=True)).contiguous()
            else:
                alpha = torch.full((T,), 1.0 / T, device=dev, dtype=torch.float64)
            out = torch.empty((N, d), dtype=torch.float64, device=dev)
            ms = timeit(torch, lambda: native.check(lib.hwer_blend_normalize_f64(P(tp), d, None, P(alpha), 1 if per_row else 0, T, N, d, P(out), d, None, S()), "blend64"))
            byts = N * (8.0 * d * (T + 1) + (8 * T if per_row else 0))
            print(json.dumps({"kernel": "blend_normalize_f64", "case": name, "rows": N, "ms": ms, "rows_per_s": N / ms * 1e3,
                              "roofline": hbm_roofline(byts, ms, 8.0 * N * d * (T + 1))}))
            del tabs, out, alpha
        N, d, deg = 1000000, 128, 16
        E = unit_rows(torch, N, d, dev, 40).to(torch.float64)
        g = torch.Generator(device=dev).manual_seed(41)
        indices = torch.randint(0, N, (N * deg,), generator=g, device=dev, dtype=torch.int32)
        indptr = (torch.arange(N + 1, device=dev, dtype=torch.int64) * deg).contiguous()
        out = torch.empty((N, d), dtype=torch.float64, device=dev)
        for kname, fn in (("segment_mean_f64", lib.hwer_segment_mean_f64), ("content_tail_f64", lib.hwer_content_tail_f64)):
            ms = timeit(torch, lambda: native.check(fn(P(E), d, P(indptr), P(indices), N, d, P(out), d, S()), kname))
            byts = 8.0 * d * (N * deg + N + (N if kname.startswith("content") else 0)) + 8 * N + 4 * N * deg
            print(json.dumps({"kernel": kname, "rows": N, "nnz": N * deg, "ms": ms,
                              "roofline": {"bound": "hbm (gathered rows)", "achieved": byts / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": byts / ms / 1e6 / HBM_PEAK_GBS}}))
        del E, indices, indptr, out


    if "fit" in what:
        # ---- fit-side table construction: segmented Gram + projection, float64 MFMA -----------------
        # one node type, N rows of D features, 2**16-row batches (IncrementalPCA's), n components.  Roofline: float64 MFMA
        # (v_mfma_f64_16x16x4_f64).  The guide has no float64 MFMA row; tools/ubench/mfma_f64_rate.hip measures back-to-back
        # issue at ~128 cycles per MFMA per SIMD (profiles/r1_ubench_mfma_f64_rate.txt: 34-45 TFLOP/s over the chip), i.e.
        # 16 flop / clk / SIMD = 39.3 TF at 2.4 GHz - half of the 78.6 TF data-sheet figure.  39.3 TF is the peak used here.
        F64_PEAK_TF = 39.3
        for (N, D, n) in [(1 << 20, 64, 32), (1 << 22, 128, 64), (1 << 21, 256, 128)]:
            g = torch.Generator(device=dev).manual_seed(N % 1000 + D)
            B = torch.randn((N, D), generator=g, device=dev, dtype=torch.float64)
            B = B / B.norm(dim=1, keepdim=True)
            nseg = N >> 16
            seg = torch.arange(0, nseg + 1, dtype=torch.int64, device=dev) * 65536
            G = torch.empty((nseg, D + 1, D + 1), dtype=torch.float64, device=dev)
            wsb = lib.hwer_segment_gram_workspace_bytes(D, nseg, 65536)
            ws = torch.empty(wsb // 8, dtype=torch.float64, device=dev)
            fn = lambda: native.check(lib.hwer_segment_gram_f64(P(B), D, D, P(seg), nseg, 65536, P(G), P(ws), wsb, S()), "gram")  # noqa: E731
            ms = timeit(torch, fn, iters=5, warmup=1)
            # algorithmic work: a symmetric Gram of [B | 1] needs (D+1)(D+2)/2 multiply-adds per row; the kernel issues whole
            # 16 x 16 MFMA sub-tiles of the upper triangle (diagonal 64-tiles: 10 of 16, off-diagonal: 16) and re-reads the
            # rows once per 64-column tile pair
            flops = 1.0 * N * (D + 1) * (D + 2)
            dp = lib.hwer_gram_padded_dim(D)
            nt = (dp + 63) // 64
            sub = [min(4, (dp - 64 * t) // 16) for t in range(nt)]
            issued_sub = sum(x * (x + 1) // 2 for x in sub) + sum(sub[i] * sub[j] for i in range(nt) for j in range(i + 1, nt))
            issued = 2.0 * N * issued_sub * 256
            cols_read = sum(16 * x for x in sub) + sum(16 * (sub[i] + sub[j]) for i in range(nt) for j in range(i + 1, nt))
            print(json.dumps({"kernel": "segment_gram_f64 + reduce", "case": "%d rows x %d features, %d segments" % (N, D, nseg), "ms": ms,
                              "TFs_algorithmic": flops / ms / 1e9, "frac_f64_mfma": flops / ms / 1e9 / F64_PEAK_TF,
                              "TFs_issued": issued / ms / 1e9, "frac_issued": issued / ms / 1e9 / F64_PEAK_TF,
                              "hbm_GBs_algorithmic": 8.0 * N * D / ms / 1e6, "read_GBs_incl_rereads": 8.0 * N * cols_read / ms / 1e6,
                              "workspace_MB": wsb / 1e6}))
            npad = lib.hwer_pca_padded_components(n)
            Pm = torch.zeros((D, npad), dtype=torch.float64, device=dev)
            Pm[:, :n] = torch.randn((D, n), generator=g, device=dev, dtype=torch.float64)
            bias = torch.zeros(npad, dtype=torch.float64, device=dev)
            ld = n + 4
            out = torch.empty((N, ld), dtype=torch.float32, device=dev)
            fn = lambda: native.check(lib.hwer_pca_project_unit_f64(P(B), D, N, D, P(Pm), npad, P(bias), n, None, 1, 4, P(out), ld, None, 0, S()), "proj")  # noqa: E731
            ms = timeit(torch, fn, iters=5, warmup=1)
            byt = 8.0 * N * D + 4.0 * N * ld
            print(json.dumps({"kernel": "pca_project_unit_f64", "case": "%d rows x %d features -> %d components + 4 one-hot" % (N, D, n), "ms": ms,
                              "TFs": 2.0 * N * D * n / ms / 1e9, "frac_f64_mfma": 2.0 * N * D * n / ms / 1e9 / F64_PEAK_TF,
                              "hbm_GBs": byt / ms / 1e6, "hbm_frac": byt / ms / 1e6 / HBM_PEAK_GBS}))
            del B, G, ws, out

    if "latency" in what:
        I, d, K = args.items, args.d, args.k
        ldx = lib.hwer_knn_padded_dim(d)
        Ip = lib.hwer_knn_padded_rows(I)
        Xs = unit_rows(torch, I, d, dev, 30)
        X = torch.zeros((Ip, ldx), dtype=torch.float32, device=dev)
        hn = torch.empty(Ip, dtype=torch.float32, device=dev)
        rows = torch.arange(I, dtype=torch.int64, device=dev)
        native.check(lib.hwer_pack_rows_f32(P(Xs), I, d, d, P(rows), I, P(X), ldx, P(hn), S()), "pack")
        Bmax = 4096
        Vq = unit_rows(torch, Bmax, d, dev, 31)
        anchors = torch.arange(Bmax, dtype=torch.int32, device=dev)
        Q64 = torch.empty((Bmax, d), dtype=torch.float64, device=dev)
        Q32 = torch.empty((Bmax, ldx), dtype=torch.float32, device=dev)
        native.check(lib.hwer_query_mix(P(Vq), Bmax, d, d, P(anchors), None, None, None, None, Bmax, 0, P(Q64), P(Q32), ldx, S()), "mix")
        o_idx = torch.empty((Bmax, K), dtype=torch.int64, device=dev)
        o_sc = torch.empty((Bmax, K), dtype=torch.float64, device=dev)
        o_d2 = torch.empty((Bmax, K), dtype=torch.float64, device=dev)
        o_fl = torch.empty(Bmax, dtype=torch.int32, device=dev)
        hmax = 0.5 * (1 + 1e-4) ** 2
        x16_exp = lib.hwer_knn_f16_scale_exp(hmax)
        X16 = torch.empty((Ip, ldx), dtype=torch.float16, device=dev)
        native.check(lib.hwer_pack_rows_f16(P(X), Ip, ldx, x16_exp, P(X16), S()), "pack16")
        for B, f16 in [(int(x), f) for x in args.batches.split(",") for f in ((True, False) if int(x) <= int(os.environ.get("HWER_SMALL_MAXQ", "1024")) else (False,))]:
            wsb = lib.hwer_knn_workspace_bytes(B, I, d, K)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            fn = lambda: native.check(lib.hwer_knn_search_opts_f32(P(X), P(hn), I, d, ldx, P(Q32), P(Q64), P(Vq), d, P(anchors), B, K, 0, 1, None, 0,  # noqa: E731
                                                                   hmax, P(o_idx), P(o_sc), P(o_d2), P(o_fl), P(ws), wsb, 0,
                                                                   P(X16) if f16 else None, x16_exp, S()), "knn")
            ms = timeit(torch, fn, iters=10)
            bad = int((o_fl[:B] == 0).sum().item())
            x_bytes = 4.0 * ldx * I
            flops = 2.0 * B * I * d
            print(json.dumps({"kernel": "knn_search", "case": "%d queries x %d items, d=%d, k=%d%s" % (B, I, d, K, ", float16 storage" if f16 else ""), "ms": ms,
                              "latency_ms_per_batch": ms, "pairs_per_s": B * float(I) / ms * 1e3, "uncertified": bad,
                              "hbm_GBs_on_X": x_bytes / ms / 1e6, "hbm_frac": x_bytes / ms / 1e6 / HBM_PEAK_GBS,
                              "mfma_TFs": flops / ms / 1e9, "workspace_MB": wsb / 1e6}))
            del ws


if __name__ == "__main__":
    main()
