// Exact k-NN for the shapes the fused search (knn_screen / knn_select) does not take - tables wider than 256 columns, k above 392,
// genuine float64 rows - as "screen a superset, re-rank it exactly, certify" built from hand-written gfx950 kernels (round 4; rounds
// 1-3 used a library GEMM + library top-k / sorts here).  The reference accepts any width and any k <= N_t
// (hwer/recommendation_base.py:74,79; hwer/content_recommender.py:67 emits n_dims-wide rows, unbounded).
//
//   hwer_knn_keys_f32       key(q, i) = q . x_i - 0.5 ||x_i||^2 for every (query, row): float32 MFMA (v_mfma_f32_32x32x2_f32, an exact
//                           fmaf chain), the d columns consumed in 16-column slabs staged through LDS into the SAME accumulators
//                           (any d: the table's row stride is a multiple of 16), 128 x 128 keys per workgroup.  Bound: fp32 MFMA.
//   hwer_knn_topk_rows_f32  per query: the kc largest keys of its row - exact kc-th key by a 4-pass byte radix select over the row
//                           (L2 / Infinity-Cache resident), then an ORDERED compaction: ids ascending, ties at the cut by ascending id.
//   hwer_knn_topk_rows_f64  the same over a row of float64 squared distances, smallest first (the all-rows fallback).
//   hwer_knn_order_rows_f64 per query: order up to 8 192 (value, id) entries by (value asc | desc, id asc) - bitonic network in LDS,
//                           composite comparator, so the result is what a STABLE sort of id-ordered input gives - cut to the first
//                           n_out, permute payload arrays alongside, and evaluate the certificate (n_out-th value strictly below a
//                           per-query bound).  Longer rows: a rank sort through a caller workspace (quadratic; k in the thousands).
//   hwer_knn_cand_dist2     (rows_f64.hip) gets a 16-byte-load variant for float32 tables here (cand_dist2_f32_fast).
// Host side: hwer/device_index.py::_search_candidates / _search_exhaustive.
#include <algorithm>

#include "hwer_common.h"

namespace hwer {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// =====================================================================================================
// (A) key matrix: 128 rows x 128 queries per workgroup, 64 x 64 per wave (2 x 2 MFMA blocks), 16-column slabs
// =====================================================================================================
constexpr int WK_TILE = 128;
constexpr int WK_KC = 16;
// LDS row stride in floats: 20 -> the 16 lanes of a ds_read_b128 group (16 different rows, same column chunk) start on banks
// 4 (5 r mod 16): all different - conflict-free without a swizzle
constexpr int WK_RS = 20;

struct WideKeysArgs {
    const float* X;
    const float* hnorm;
    int64_t n_alloc;   // rows of X / hnorm that may be READ (the padded allocation)
    int64_t n_items;   // rows whose keys are stored
    int64_t ldx;
    const float* Q;
    int64_t ldq;
    int64_t B;
    int n_slabs;       // ldx / 16
    float* keys;
    int64_t ld_keys;
};

__global__ __launch_bounds__(256, 2) void wide_keys_kernel(const WideKeysArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][WK_TILE * WK_RS];
    __shared__ __attribute__((aligned(16))) float Bs[2][WK_TILE * WK_RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wi = wave & 1, wq = wave >> 1;
    const int64_t i0 = int64_t(blockIdx.x) * WK_TILE, q0 = int64_t(blockIdx.y) * WK_TILE;
    // loader: thread (lr, lc) moves the 16-byte chunk lc of rows lr and lr + 64 of both operands (rows past the end: the last row -
    // their keys are never stored)
    const int lr = tid >> 2, lc = tid & 3;
    const float* xa0 = a.X + std::min<int64_t>(i0 + lr, a.n_alloc - 1) * a.ldx + 4 * lc;
    const float* xa1 = a.X + std::min<int64_t>(i0 + lr + 64, a.n_alloc - 1) * a.ldx + 4 * lc;
    const float* qb0 = a.Q + std::min<int64_t>(q0 + lr, a.B - 1) * a.ldq + 4 * lc;
    const float* qb1 = a.Q + std::min<int64_t>(q0 + lr + 64, a.B - 1) * a.ldq + 4 * lc;
    float4 pa0 = *reinterpret_cast<const float4*>(xa0), pa1 = *reinterpret_cast<const float4*>(xa1);
    float4 pb0 = *reinterpret_cast<const float4*>(qb0), pb1 = *reinterpret_cast<const float4*>(qb1);
    auto stash = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][lr * WK_RS + 4 * lc]) = pa0;
        *reinterpret_cast<float4*>(&As[buf][(lr + 64) * WK_RS + 4 * lc]) = pa1;
        *reinterpret_cast<float4*>(&Bs[buf][lr * WK_RS + 4 * lc]) = pb0;
        *reinterpret_cast<float4*>(&Bs[buf][(lr + 64) * WK_RS + 4 * lc]) = pb1;
    };
    // accumulators start at -0.5 ||x||^2: C[m = row][n = query], lane (r, h) holds query r, rows (reg & 3) + 8 (reg >> 2) + 4 h
    f32x16 acc[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int64_t it = i0 + 64 * wi + 32 * ib + 8 * rg + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hv = a.hnorm[std::min<int64_t>(it + j, a.n_alloc - 1)];
                acc[ib][0][4 * rg + j] = -hv;
                acc[ib][1][4 * rg + j] = -hv;
            }
        }
    stash(0);
    __syncthreads();
    for (int c = 0; c < a.n_slabs; ++c) {
        if (c + 1 < a.n_slabs) {   // next slab's global loads fly under this slab's MFMAs
            const int o = WK_KC * (c + 1);
            pa0 = *reinterpret_cast<const float4*>(xa0 + o);
            pa1 = *reinterpret_cast<const float4*>(xa1 + o);
            pb0 = *reinterpret_cast<const float4*>(qb0 + o);
            pb1 = *reinterpret_cast<const float4*>(qb1 + o);
        }
        const float* Ab = As[c & 1];
        const float* Bb = Bs[c & 1];
#pragma unroll
        for (int g = 0; g < WK_KC / 8; ++g) {
            float4 av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = *reinterpret_cast<const float4*>(&Ab[(64 * wi + 32 * t + r) * WK_RS + 8 * g + 4 * h]);
                bv[t] = *reinterpret_cast<const float4*>(&Bb[(64 * wq + 32 * t + r) * WK_RS + 8 * g + 4 * h]);
            }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    acc[ib][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ib].x, bv[qb].x, acc[ib][qb], 0, 0, 0);
                    acc[ib][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ib].y, bv[qb].y, acc[ib][qb], 0, 0, 0);
                    acc[ib][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ib].z, bv[qb].z, acc[ib][qb], 0, 0, 0);
                    acc[ib][qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ib].w, bv[qb].w, acc[ib][qb], 0, 0, 0);
                }
        }
        if (c + 1 < a.n_slabs) stash((c + 1) & 1);   // the other ring slot: every wave finished reading it before the last barrier
        __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int64_t q = q0 + 64 * wq + 32 * qb + r;
        if (q >= a.B) continue;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int64_t it = i0 + 64 * wi + 32 * ib + 8 * rg + 4 * h;
                if (it < a.n_items)   // (ld_keys >= roundup4(n_items): the 16-byte store stays inside the row)
                    *reinterpret_cast<float4*>(a.keys + q * a.ld_keys + it) =
                        make_float4(acc[ib][qb][4 * rg], acc[ib][qb][4 * rg + 1], acc[ib][qb][4 * rg + 2], acc[ib][qb][4 * rg + 3]);
            }
    }
}

// =====================================================================================================
// (B) per row: the k best entries - exact k-th key by byte radix select, then ordered compaction
// =====================================================================================================
// keys are compared as unsigned integers, larger = better: float32 keys (largest first) through f32_orderable, float64 squared
// distances (smallest first) through the complement of the order-preserving map
__device__ __forceinline__ uint64_t f64_orderable(double v) {
    const uint64_t u = uint64_t(__double_as_longlong(v));
    return u ^ ((u >> 63) ? ~0ull : (1ull << 63));
}
__device__ __forceinline__ double f64_from_orderable(uint64_t o) {
    const uint64_t u = o ^ ((o >> 63) ? (1ull << 63) : ~0ull);
    return __longlong_as_double((long long)u);
}
template <typename KT>
__device__ __forceinline__ KT row_key(const void* row, int64_t i) {
    if constexpr (sizeof(KT) == 4) return KT(f32_orderable(static_cast<const float*>(row)[i]));
    else return KT(~f64_orderable(static_cast<const double*>(row)[i]));
}

struct RowSelectArgs {
    const void* vals;
    int64_t ld;
    int64_t n;
    int64_t k;
    int64_t* out_idx;   // (B, ld_out): the k selected positions, ascending
    int64_t ld_out;
    float* out_thr;     // nullable (float32 flavour): the k-th largest key of every row
};

template <typename KT>
__global__ __launch_bounds__(256) void row_select_kernel(const RowSelectArgs a) {
    __shared__ int hist[256];
    __shared__ int sh_bin, sh_rank;
    __shared__ int wcnt_gt[4], wcnt_eq[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const void* row = static_cast<const char*>(a.vals) + size_t(b) * size_t(a.ld) * sizeof(KT);
    const int n = int(a.n);
    constexpr int NB = int(sizeof(KT));
    KT prefix = 0;
    int rank = int(a.k);     // the rank-th best among the keys that share `prefix` above the current byte
    for (int p = 0; p < NB; ++p) {
        const int shift = 8 * (NB - 1 - p);
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const KT key = row_key<KT>(row, i);
            if (p == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[int((key >> shift) & KT(255))], 1);
        }
        __syncthreads();
        if (wave == 0) {   // lane l owns bins 4 l .. 4 l + 3; walk the bins from the top until `rank` keys are covered
            const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            const int s = c0 + c1 + c2 + c3;
            int suf = s;   // inclusive suffix sum over lanes >= l
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_down(suf, off, kWave);
                if (lane + off < 64) suf += o;
            }
            int cum = suf - s;   // keys in bins above this lane's
            const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (cum < rank && cum + cs[j] >= rank) { sh_bin = 4 * lane + j; sh_rank = rank - cum; }
                cum += cs[j];
            }
        }
        __syncthreads();
        prefix |= KT(sh_bin) << shift;
        rank = sh_rank;
        __syncthreads();
    }
    // prefix = T, the k-th best key; `rank` = how many of the keys == T belong to the k best (>= 1).  Ordered compaction: every wave
    // takes a contiguous quarter of the row, so positions come out ascending and the equals that are kept are the FIRST `rank` of them.
    const KT T = prefix;
    const int need = rank;
    const int seg = ((n + 3) / 4 + 63) / 64 * 64;
    const int lo = std::min(n, wave * seg), hi = std::min(n, lo + seg);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    int cgt = 0, ceq = 0;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < hi;
        const KT key = live ? row_key<KT>(row, i) : KT(0);
        cgt += __builtin_popcountll(__ballot(live && key > T));
        ceq += __builtin_popcountll(__ballot(live && key == T));
    }
    if (lane == 0) { wcnt_gt[wave] = cgt; wcnt_eq[wave] = ceq; }
    __syncthreads();
    int eq_base = 0, gt_base = 0;
    for (int w = 0; w < wave; ++w) { eq_base += wcnt_eq[w]; gt_base += wcnt_gt[w]; }
    int out_base = gt_base + std::min(need, eq_base);
    int64_t* out = a.out_idx + b * a.ld_out;
    for (int i0 = lo; i0 < hi; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < hi;
        const KT key = live ? row_key<KT>(row, i) : KT(0);
        const bool gt = live && key > T, eq = live && key == T;
        const uint64_t eb = __ballot(eq);
        const bool keep = gt || (eq && eq_base + __builtin_popcountll(eb & lt_mask) < need);
        const uint64_t kb = __ballot(keep);
        if (keep) out[out_base + __builtin_popcountll(kb & lt_mask)] = i;
        out_base += __builtin_popcountll(kb);
        eq_base += __builtin_popcountll(eb);
    }
    if constexpr (sizeof(KT) == 4) {
        if (tid == 0 && a.out_thr) a.out_thr[b] = f32_from_orderable(uint32_t(T));
    }
}

// =====================================================================================================
// (C) per row: order (value, id) entries, cut, permute payloads, certificate
// =====================================================================================================
struct OrderArgs {
    const double* prim;       // (B, ld_prim) values
    int64_t ld_prim;
    const int64_t* gather;    // nullable (B, ld_gather): entry c is prim[b][gather[b][c]], its id is gather[b][c]
    int64_t ld_gather;
    const int64_t* sec;       // nullable (B, ld_sec), only without gather: the id of entry c (default: its position c)
    int64_t ld_sec;
    int n_in, n_out, descending;
    double* out_prim;         // (B, n_out)
    int64_t* out_id;          // nullable (B, n_out): ids of the ordered entries + id_add
    int64_t id_add;
    const int64_t* id_map;    // nullable: out_id = id_map[id] instead of id + id_add (local row -> global row)
    const int64_t* pay_i_in;  // nullable (B, n_in) -> pay_i_out (B, n_out): payload of the ordered entries (entries identified by POSITION:
    int64_t* pay_i_out;       //   only without gather / sec)
    const double* pay_f_in;
    double* pay_f_out;
    const double* bound;      // nullable (B): flags[b] = (every_row || out_prim[b][n_out - 1] < bound[b]) ? 1 : 0
    int32_t* flags;
    int every_row;
    // rank-sort path (n_in > ORDER_LDS_MAX): the entries of all rows, written by order_prepare_kernel
    uint64_t* ws_a;
    uint32_t* ws_b;
};
constexpr int ORDER_LDS_MAX = 8192;   // entries the LDS network takes (96 KB)

__device__ __forceinline__ void order_load(const OrderArgs& a, int64_t b, int c, uint64_t& A, uint32_t& Bv) {
    int64_t id = c;
    double v;
    if (a.gather) {
        id = a.gather[b * a.ld_gather + c];
        v = a.prim[b * a.ld_prim + id];
    } else {
        v = a.prim[b * a.ld_prim + c];
        if (a.sec) id = a.sec[b * a.ld_sec + c];
    }
    const uint64_t o = f64_orderable(v + 0.0);   // -0.0 -> +0.0: the two compare equal, so they must order by id like any other tie
    A = a.descending ? ~o : o;
    Bv = uint32_t(id);
}
__device__ __forceinline__ void order_emit(const OrderArgs& a, int64_t b, int j, uint64_t A, uint32_t Bv) {
    const double v = f64_from_orderable(a.descending ? ~A : A);
    a.out_prim[b * a.n_out + j] = v;
    if (a.out_id) a.out_id[b * a.n_out + j] = a.id_map ? a.id_map[Bv] : int64_t(Bv) + a.id_add;
    if (a.pay_i_out) a.pay_i_out[b * a.n_out + j] = a.pay_i_in[b * a.n_in + Bv];
    if (a.pay_f_out) a.pay_f_out[b * a.n_out + j] = a.pay_f_in[b * a.n_in + Bv];
    if (a.flags && j == a.n_out - 1) a.flags[b] = (a.every_row || !a.bound || v < a.bound[b]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void order_rows_lds_kernel(const OrderArgs a, int np2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t* A = reinterpret_cast<uint64_t*>(smem_raw);
    uint32_t* Bv = reinterpret_cast<uint32_t*>(A + np2);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    for (int c = tid; c < np2; c += 256) {
        uint64_t x = ~0ull;
        uint32_t y = ~0u;
        if (c < a.n_in) order_load(a, b, c, x, y);
        A[c] = x;
        Bv[c] = y;
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < (np2 >> 1); i += 256) {
                const int p = ((i & ~(stride - 1)) << 1) | (i & (stride - 1));   // lower index of pair i at this stride
                const int q = p | stride;
                const bool asc = (p & size) == 0;
                const uint64_t ap = A[p], aq = A[q];
                const uint32_t bp = Bv[p], bq = Bv[q];
                const bool q_less = aq < ap || (aq == ap && bq < bp);
                if (q_less == asc) { A[p] = aq; A[q] = ap; Bv[p] = bq; Bv[q] = bp; }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < a.n_out; j += 256) order_emit(a, b, j, A[j], Bv[j]);
}

// rows longer than the LDS network: entries to the workspace, then every entry counts the entries that precede it (quadratic - k in
// the thousands with thousands of queries is the corner this serves, not a throughput path)
__global__ __launch_bounds__(256) void order_prepare_kernel(const OrderArgs a) {
    const int64_t b = blockIdx.y;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < a.n_in; c += gridDim.x * 256) {
        uint64_t x;
        uint32_t y;
        order_load(a, b, c, x, y);
        a.ws_a[b * a.n_in + c] = x;
        a.ws_b[b * a.n_in + c] = y;
    }
}
__global__ __launch_bounds__(256) void order_rank_kernel(const OrderArgs a) {
    __shared__ uint64_t ta[1024];
    __shared__ uint32_t tb[1024];
    const int64_t b = blockIdx.y;
    const uint64_t* A = a.ws_a + b * a.n_in;
    const uint32_t* Bv = a.ws_b + b * a.n_in;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < a.n_in;
    const uint64_t ae = live ? A[e] : 0ull;
    const uint32_t be = live ? Bv[e] : 0u;
    int rank = 0;
    for (int t0 = 0; t0 < a.n_in; t0 += 1024) {
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const bool in = t0 + i < a.n_in;
            ta[i] = in ? A[t0 + i] : ~0ull;
            tb[i] = in ? Bv[t0 + i] : ~0u;
        }
        __syncthreads();
        const int m = std::min(1024, a.n_in - t0);
        for (int i = 0; i < m; ++i) rank += (ta[i] < ae || (ta[i] == ae && tb[i] < be)) ? 1 : 0;
    }
    if (live && rank < a.n_out) order_emit(a, b, rank, ae, be);   // (value, id) pairs are distinct: ranks are a permutation
}

// =====================================================================================================
// (D) exact squared distances of listed rows of a float32 table, 16-byte loads
// =====================================================================================================
__global__ __launch_bounds__(256) void cand_dist2_f32x4_kernel(const float* __restrict__ table, int64_t ld, int d, const int64_t* __restrict__ cand,
                                                               int64_t ldc, int64_t kc, const double* __restrict__ Q64, double* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const double* q = Q64 + b * d;
    const int d4 = d >> 2;
    for (int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x; c < kc; c += int64_t(gridDim.x) * 256) {
        const int64_t row = cand[b * ldc + c];
        double acc = INFINITY;
        if (row >= 0) {
            const float4* x4 = reinterpret_cast<const float4*>(table + row * ld);
            acc = 0.0;
            float4 nx = d4 > 0 ? x4[0] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int j4 = 0; j4 < d4; ++j4) {    // scikit-learn's order: one dependent add per column (euclidean_rdist)
                const float4 x = nx;
                if (j4 + 1 < d4) nx = x4[j4 + 1];
                const double t0 = __dsub_rn(q[4 * j4], double(x.x)), t1 = __dsub_rn(q[4 * j4 + 1], double(x.y));
                const double t2 = __dsub_rn(q[4 * j4 + 2], double(x.z)), t3 = __dsub_rn(q[4 * j4 + 3], double(x.w));
                acc = __dadd_rn(acc, __dmul_rn(t0, t0));
                acc = __dadd_rn(acc, __dmul_rn(t1, t1));
                acc = __dadd_rn(acc, __dmul_rn(t2, t2));
                acc = __dadd_rn(acc, __dmul_rn(t3, t3));
            }
            const float* xs = table + row * ld;
            for (int j = 4 * d4; j < d; ++j) {
                const double t = __dsub_rn(q[j], double(xs[j]));
                acc = __dadd_rn(acc, __dmul_rn(t, t));
            }
        }
        out[b * kc + c] = acc;
    }
}

}  // namespace hwer

using namespace hwer;

extern "C" {

int hwer_knn_keys_f32(const float* X, const float* hnorm, int64_t n_rows_padded, int64_t n_items, int64_t ldx, const float* Q32, int64_t ldq,
                      int64_t B, float* keys, int64_t ld_keys, void* stream) {
    if (B == 0 || n_items == 0) return HWER_OK;
    if (!X || !hnorm || !Q32 || !keys) return fail(HWER_EINVAL, "knn_keys: null pointer");
    if (n_items < 1 || n_rows_padded < n_items || ldx < 16 || ldx % 16 != 0 || ldq < ldx || ldq % 4 != 0 || ld_keys % 4 != 0 || ld_keys < ((n_items + 3) / 4) * 4)
        return fail(HWER_EINVAL, "knn_keys: need n_rows_padded >= n_items >= 1, ldx %% 16 == 0, ldq >= ldx (queries zero-padded to ldx columns), ld_keys >= roundup4(n_items)");
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Q32) & 15) || (reinterpret_cast<uintptr_t>(keys) & 15))
        return fail(HWER_EINVAL, "knn_keys: X / Q32 / keys must be 16-byte aligned");
    const int64_t gy = cdiv64(B, WK_TILE);
    if (gy > 65535) return fail(HWER_EINVAL, "knn_keys: B <= %lld per call", (long long)(65535ll * WK_TILE));
    WideKeysArgs a{X, hnorm, n_rows_padded, n_items, ldx, Q32, ldq, B, int(ldx / WK_KC), keys, ld_keys};
    hipLaunchKernelGGL(wide_keys_kernel, dim3(unsigned(cdiv64(n_items, WK_TILE)), unsigned(gy)), dim3(256), 0, as_stream(stream), a);
    return check_launch("knn_keys");
}

int hwer_knn_topk_rows_f32(const float* keys, int64_t ld_keys, int64_t n, int64_t B, int64_t kc, int64_t* out_idx, int64_t ld_out, float* out_kth,
                           void* stream) {
    if (B == 0) return HWER_OK;
    if (!keys || !out_idx) return fail(HWER_EINVAL, "knn_topk_rows_f32: null pointer");
    if (n < 1 || n >= (int64_t(1) << 31) || kc < 1 || kc > n || ld_keys < n || ld_out < kc) return fail(HWER_EINVAL, "knn_topk_rows_f32: need 1 <= kc <= n < 2^31, ld_keys >= n, ld_out >= kc");
    RowSelectArgs a{keys, ld_keys, n, kc, out_idx, ld_out, out_kth};
    hipLaunchKernelGGL(row_select_kernel<uint32_t>, dim3(unsigned(B)), dim3(256), 0, as_stream(stream), a);
    return check_launch("knn_topk_rows_f32");
}

int hwer_knn_topk_rows_f64(const double* dist2, int64_t ld, int64_t n, int64_t B, int64_t k, int64_t* out_idx, int64_t ld_out, void* stream) {
    if (B == 0) return HWER_OK;
    if (!dist2 || !out_idx) return fail(HWER_EINVAL, "knn_topk_rows_f64: null pointer");
    if (n < 1 || n >= (int64_t(1) << 31) || k < 1 || k > n || ld < n || ld_out < k) return fail(HWER_EINVAL, "knn_topk_rows_f64: need 1 <= k <= n < 2^31, ld >= n, ld_out >= k");
    RowSelectArgs a{dist2, ld, n, k, out_idx, ld_out, nullptr};
    hipLaunchKernelGGL(row_select_kernel<uint64_t>, dim3(unsigned(B)), dim3(256), 0, as_stream(stream), a);
    return check_launch("knn_topk_rows_f64");
}

size_t hwer_knn_order_rows_workspace_bytes(int64_t B, int64_t n_in) {
    if (n_in <= ORDER_LDS_MAX) return 0;
    return size_t(B) * size_t(n_in) * 12 + 256;
}

int hwer_knn_order_rows_f64(const double* values, int64_t ld_values, const int64_t* gather, int64_t ld_gather, const int64_t* ids, int64_t ld_ids,
                            int64_t B, int64_t n_in, int64_t n_out, int descending, double* out_values, int64_t* out_ids, int64_t id_add,
                            const int64_t* id_map, const int64_t* payload_i64, int64_t* out_payload_i64, const double* payload_f64,
                            double* out_payload_f64, const double* bound, int32_t* out_flags, int every_row, void* workspace,
                            size_t workspace_bytes, void* stream) {
    if (B == 0 || n_out == 0) return HWER_OK;
    if (!values || !out_values) return fail(HWER_EINVAL, "knn_order_rows: null pointer");
    if (n_in < 1 || n_out < 1 || n_out > n_in || n_in >= (int64_t(1) << 31) || B > 65535)
        return fail(HWER_EINVAL, "knn_order_rows: need 1 <= n_out <= n_in < 2^31, B <= 65535");
    if ((gather || ids) && (payload_i64 || payload_f64)) return fail(HWER_EINVAL, "knn_order_rows: payloads follow POSITIONS - not available with gather / ids");
    if ((payload_i64 != nullptr) != (out_payload_i64 != nullptr) || (payload_f64 != nullptr) != (out_payload_f64 != nullptr))
        return fail(HWER_EINVAL, "knn_order_rows: payload in / out pointers come in pairs");
    if (gather && ld_gather < n_in) return fail(HWER_EINVAL, "knn_order_rows: ld_gather < n_in");
    if (!gather && ld_values < n_in) return fail(HWER_EINVAL, "knn_order_rows: ld_values < n_in");
    OrderArgs a{values, ld_values, gather, ld_gather, gather ? nullptr : ids, ld_ids, int(n_in), int(n_out), descending ? 1 : 0, out_values, out_ids, id_add, id_map,
                payload_i64, out_payload_i64, payload_f64, out_payload_f64, bound, out_flags, every_row ? 1 : 0, nullptr, nullptr};
    hipStream_t st = as_stream(stream);
    if (n_in <= ORDER_LDS_MAX) {
        int np2 = 64;
        while (np2 < n_in) np2 <<= 1;
        const size_t lds = size_t(np2) * 12;
        if (lds > size_t(64) * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&order_rows_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
            if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_order_rows: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(order_rows_lds_kernel, dim3(unsigned(B)), dim3(256), lds, st, a, np2);
        return check_launch("knn_order_rows");
    }
    const size_t need = hwer_knn_order_rows_workspace_bytes(B, n_in);
    if (!workspace || workspace_bytes < need) return fail(HWER_EWORKSPACE, "knn_order_rows: workspace %zu < %zu", workspace_bytes, need);
    a.ws_a = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(workspace) + 15) & ~uintptr_t(15));
    a.ws_b = reinterpret_cast<uint32_t*>(a.ws_a + size_t(B) * size_t(n_in));
    const unsigned gx = unsigned(cdiv64(n_in, 256));
    hipLaunchKernelGGL(order_prepare_kernel, dim3(std::min(gx, 1024u), unsigned(B)), dim3(256), 0, st, a);
    int rc;
    if ((rc = check_launch("knn_order_rows(prepare)"))) return rc;
    hipLaunchKernelGGL(order_rank_kernel, dim3(gx, unsigned(B)), dim3(256), 0, st, a);
    return check_launch("knn_order_rows(rank)");
}

}  // extern "C"

// used by rows_f64.hip's hwer_knn_cand_dist2 for float32 tables whose rows are 16-byte aligned
namespace hwer {
int launch_cand_dist2_f32x4(const float* table, int64_t ld, int d, const int64_t* cand, int64_t ld_cand, int64_t B, int64_t kc, const double* Q64,
                            double* out, hipStream_t st) {
    const dim3 grid(unsigned(std::min<int64_t>(cdiv64(kc, 256), 1024)), unsigned(B));
    hipLaunchKernelGGL(cand_dist2_f32x4_kernel, grid, dim3(256), 0, st, table, ld, d, cand, ld_cand, kc, Q64, out);
    return check_launch("knn_cand_dist2");
}
}  // namespace hwer
