// float64 `vectors` (round 3; SURVEY.md 8c "stage-wise" -> true float64 flavour).  The reference keeps ContentRecommendation's rows
// in float64 (hwer/content_recommender.py:96, hwer/recommendation_base.py:109) and both the KD-tree (:74, :79) and predict (:135-144)
// work on them.  The MFMA screen still runs on the float32 / float16 roundings of the rows (a screen only has to keep a superset);
// the kernels here produce the reference's VALUES from the float64 rows:
//   hwer_query_mix_f64rows   (rowops.hip)  queries from float64 rows, numpy's orders
//   hwer_pair_score_f64rows               predict on float64 rows, numpy's pairwise order
//   hwer_knn_cand_dist2                   exact squared distances of listed candidate rows (float32 or float64 table) in
//                                         scikit-learn's sequential order - the re-rank of the candidate sets that hwer_knn_keys_f32 +
//                                         hwer_knn_topk_rows_f32 (wider tables, larger k) hand over (d <= 256, k <= 392: the fused
//                                         float64-row select of knn_select_f64.hip, round 6)
// None of them is on the throughput path of the float32 flavour; they are written for clarity, HBM-coalescing where it is free.
#include <algorithm>

#include "hwer_common.h"

namespace hwer {

template <typename TV>
__device__ __forceinline__ double elem64(const TV* V, int64_t ld, int64_t row, int j, bool unknown) {
    const double v = double(V[(unknown ? 0 : row) * ld + j]);
    if (!unknown) return v;
    return v < 1e-6 ? 1e-6 : (v > 1e-5 ? 1e-5 : v);   // unknown node: row 0 clipped to [1e-6, 1e-5] (hwer/recommendation_base.py:147-150)
}

// 8 lanes per pair: products into an LDS row, then numpy's pairwise order (np_pairwise) - the arithmetic of pair_score_kernel<double>
__global__ __launch_bounds__(256) void pair_score_f64rows_kernel(const double* __restrict__ V, int d, int64_t ldv, const int32_t* __restrict__ src,
                                                                 const int32_t* __restrict__ dst, int64_t P, double* __restrict__ out, int stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g8 = lane >> 3, k = lane & 7;
    double* row = reinterpret_cast<double*>(smem_raw) + (size_t(wave) * 8 + g8) * stride;
    const int nw = int(blockDim.x) >> 6;
    for (int64_t p0 = (int64_t(blockIdx.x) * nw + wave) * 8; p0 < P; p0 += int64_t(gridDim.x) * nw * 8) {   // p0 wave-uniform
        const int64_t p = p0 + g8;
        const bool live = p < P;
        const int32_t s = live ? src[p] : 0, t = live ? dst[p] : 0;
        for (int j = k; j < d; j += 8) row[j] = elem64(V, ldv, s, j, s < 0) * elem64(V, ldv, t, j, t < 0);
        wave_lds_sync();
        const double r = np_pairwise<double, kPairwiseDepth>(row, d, k);
        if (k == 0 && live) out[p] = (r + 1.0) / 2.0;
        wave_lds_sync();
    }
}

// out[b * kc + c] = sum_j (Q64[b][j] - table[cand[b * ldc + c]][j])^2, sequential in j (scikit-learn's euclidean_rdist), float64;
// cand < 0 -> +inf.  One thread per (query, candidate); the query row is read through the scalar / L1 path (same for the block).
template <typename TV>
__global__ __launch_bounds__(256) void cand_dist2_kernel(const TV* __restrict__ table, int64_t ld, int d, const int64_t* __restrict__ cand, int64_t ldc,
                                                         int64_t kc, const double* __restrict__ Q64, double* __restrict__ out) {
    const int64_t b = blockIdx.y;
    const double* q = Q64 + b * d;
    for (int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x; c < kc; c += int64_t(gridDim.x) * 256) {
        const int64_t row = cand[b * ldc + c];
        double acc = INFINITY;
        if (row >= 0) {
            const TV* x = table + row * ld;
            acc = 0.0;
            for (int j = 0; j < d; ++j) {
                const double tmp = __dsub_rn(q[j], double(x[j]));
                acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
            }
        }
        out[b * kc + c] = acc;
    }
}

// The same for a float64 table whose rows are 16-byte aligned (ld even), COALESCED: a wave takes 64 candidates of one query; per 8-column
// chunk 4 lanes load the 64-byte segment of a row (16 rows per instruction, 4 instructions), the rows are transposed through wave-private LDS
// (row stride 80 bytes: conflict-free ds_read_b128) and every lane walks ITS candidate's 8 values in order - the data path of the select
// kernel's re-rank rounds, for 8-byte elements.  The chunk c + 1 is in flight while chunk c is reduced.  (One lane walking its own row with
// 8-byte loads touched 64 different 64-byte segments per instruction: 4 096 x 232 candidates of 128 columns took 0.61 ms, 41 % of a batched
// search on float64 rows.)
constexpr int CD_CW = 8;              // columns per chunk (one 64-byte segment of a float64 row)
constexpr int CD_RS = CD_CW + 2;      // LDS row stride in doubles
__global__ __launch_bounds__(256) void cand_dist2_f64_staged_kernel(const double* __restrict__ table, int64_t ld, int d, const int64_t* __restrict__ cand, int64_t ldc,
                                                                    int64_t kc, const double* __restrict__ Q64, double* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) double stage_all[4 * 64 * CD_RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.y;
    double* stage = stage_all + wave * (64 * CD_RS);
    typedef const __attribute__((address_space(4))) double* cdouble_p;      // wave-uniform: scalar loads
    const cdouble_p q = (cdouble_p)(Q64 + b * d);
    const int n_chunks = (d + CD_CW - 1) / CD_CW;
    const int srow = lane >> 2, scol = 2 * (lane & 3);
    for (int64_t c0 = (int64_t(blockIdx.x) * 4 + wave) * 64; c0 < kc; c0 += int64_t(gridDim.x) * 256) {   // c0 wave-uniform
        const int64_t mine = c0 + lane < kc ? cand[b * ldc + c0 + lane] : -1;
        // rows staged by this lane: candidates c0 + srow + 16 i (dead / absent candidates read row 0: their result is +inf)
        const double* src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t ci = c0 + srow + 16 * i;
            const int64_t row = ci < kc ? cand[b * ldc + ci] : -1;
            src[i] = table + (row >= 0 ? row : 0) * ld + scol;
        }
        auto load2 = [&](const double* p_, int col) {      // columns col, col + 1 of a row (zero beyond d)
            double2 v = make_double2(0.0, 0.0);
            if (col + 1 < d) v = *reinterpret_cast<const double2*>(p_);
            else if (col < d) v.x = p_[0];
            return v;
        };
        double2 pre[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = load2(src[i], scol);
        double acc = 0.0;
        for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<double2*>(stage + (srow + 16 * i) * CD_RS + scol) = pre[i];
            if (c + 1 < n_chunks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) pre[i] = load2(src[i] + CD_CW * (c + 1), CD_CW * (c + 1) + scol);
            }
            wave_lds_sync();
            double xv[CD_CW];
#pragma unroll
            for (int t = 0; t < CD_CW / 2; ++t) {
                const double2 v = *reinterpret_cast<const double2*>(stage + lane * CD_RS + 2 * t);
                xv[2 * t] = v.x;
                xv[2 * t + 1] = v.y;
            }
            wave_lds_sync();
            const int j0 = CD_CW * c;
#pragma unroll
            for (int t = 0; t < CD_CW; ++t) {
                if (j0 + t < d) {   // uniform
                    const double tmp = __dsub_rn(q[j0 + t], xv[t]);
                    acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                }
            }
        }
        if (c0 + lane < kc) out[b * kc + c0 + lane] = mine >= 0 ? acc : INFINITY;
    }
}

// knn_wide.hip: the same arithmetic with 16-byte row loads (float32 tables whose rows are 16-byte aligned)
int launch_cand_dist2_f32x4(const float* table, int64_t ld, int d, const int64_t* cand, int64_t ld_cand, int64_t B, int64_t kc, const double* Q64,
                            double* out, hipStream_t st);


// =====================================================================================================================
// Round 5: the BLEND stage in float64 - the reference's own arithmetic.  hwer/content_recommender.py:83 (`E[adj].mean(0)`), :85
// (`(E + A)/2.0`) and :96 + hwer/utils.py:43-44 (`a / expand_dims(norm(a, 1), 1)`) all run on float64 arrays; rounds 1-4 had float32
// kernels only (within 1e-6 of the reference).  These produce the reference's bits from the same float64 table:
//   neighbour mean : sequential row adds in adjacency order, then one IEEE division by the count           (ndarray.mean(0))
//   blend          : (E + A) / 2.0 (content tail); sum_t alpha_t * T_t accumulated left to right, product and sum rounded separately (general)
//   norm           : squares summed in numpy's pairwise order (np_pairwise over an LDS row), sqrt, one IEEE division per element
// One wave works on `pb_n` rows per trip: L lanes per row load 16-byte chunks (coalesced), the blended row goes to an LDS slab, 8 lanes per
// row replay numpy's summation order on its squares, then the L lanes divide and store.  HBM-bound: 8 d (nnz + 2) bytes per row for the tail,
// 8 d (T + 1) for the blend.
// =====================================================================================================================
struct Rows64Args {
    const double* const* tables;   // blend
    const int32_t* const* row_idx;
    const double* alpha;
    int alpha_per_row, T;
    const double* E;               // neighbour mean / content tail (also: the table stride for the blend)
    int64_t lde;
    const int64_t* indptr;
    const int32_t* indices;
    int64_t R;
    int d, d2;
    double* out;
    int64_t ldo;
    double* out_norm;
    int stride, pb_n;
};

__device__ __forceinline__ int pow2ceil64(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// numpy's pairwise order over the SQUARES of p[0..n) (8 lanes per row, k = lane & 7) - np_pairwise with x*x formed on the way in
__device__ __forceinline__ double np_pairwise_sq_leaf(const double* p, int n, int k) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r = r + p[i] * p[i];
        return r;
    }
    const int lim = n - (n & 7);
    double r = p[k] * p[k];
    for (int i = 8; i < lim; i += 8) r = r + p[i + k] * p[i + k];
    r = r + shfl_xor_t(r, 1);
    r = r + shfl_xor_t(r, 2);
    r = r + shfl_xor_t(r, 4);
    for (int i = lim; i < n; ++i) r = r + p[i] * p[i];
    return r;
}
template <int DEPTH>
__device__ __forceinline__ double np_pairwise_sq(const double* p, int n, int k) {
    if constexpr (DEPTH == 0) {
        return np_pairwise_sq_leaf(p, n, k);
    } else {
        if (n <= 128) return np_pairwise_sq_leaf(p, n, k);
        int n2 = n / 2;
        n2 -= n2 & 7;
        const double a = np_pairwise_sq<DEPTH - 1>(p, n2, k);
        const double b = np_pairwise_sq<DEPTH - 1>(p + n2, n - n2, k);
        return a + b;
    }
}

// MODE 0: out = mean_adj(E)   1: out = unit((E + mean_adj(E)) / 2)   2: out = unit(sum_t alpha_t T_t[idx_t])
template <int MODE>
__global__ __launch_bounds__(256) void rows64_kernel(const Rows64Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = int(blockDim.x) >> 6;
    double* slab = reinterpret_cast<double*>(smem_raw) + size_t(wave) * (size_t(a.pb_n) * a.stride + 8);
    double* s_nrm = slab + size_t(a.pb_n) * a.stride;
    const int L = min(64, pow2ceil64(a.d2));   // lanes per row
    const int rpi = 64 / L;                    // rows per load instruction
    const int sub = lane / L, c0 = lane % L;
    const int64_t n_trips = cdiv64(a.R, a.pb_n);
    for (int64_t b = int64_t(blockIdx.x) * nw + wave; b < n_trips; b += int64_t(gridDim.x) * nw) {
        const int64_t p0 = b * a.pb_n;
        if constexpr (MODE == 2) {
            // TWO rows x up to FOUR tables requested before the first is consumed: with one 16-byte load per lane in flight the kernel was a chain of
            // HBM latencies (20 resident waves x 1 KB per CU: 3.9 TB/s by Little's law, measured 3.9 - 4.0).  The accumulation order is unchanged:
            // tables left to right, product and sum rounded separately (-ffp-contract=off): `acc = acc + alpha * rows`.
            for (int pb = 0; pb < a.pb_n; pb += 2 * rpi) {
                const int slot_a = pb + sub, slot_b = pb + rpi + sub;
                const int64_t r_a = p0 + slot_a, r_b = p0 + slot_b;
                const bool live_a = slot_a < a.pb_n && r_a < a.R, live_b = slot_b < a.pb_n && r_b < a.R;
                if (!live_a) continue;      // (slot_b > slot_a: nothing to do for this lane group)
                for (int c = c0; c < a.d2; c += L) {
                    double2 acc_a = make_double2(0.0, 0.0), acc_b = make_double2(0.0, 0.0);
                    for (int t0 = 0; t0 < a.T; t0 += 4) {
                        double2 va[4], vb[4];
                        double ala[4], alb[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            va[j] = vb[j] = make_double2(0.0, 0.0);
                            ala[j] = alb[j] = 0.0;
                            if (t0 + j >= a.T) continue;      // (uniform)
                            const int t = t0 + j;
                            const int32_t* ri = a.row_idx ? a.row_idx[t] : nullptr;
                            const int64_t row_a = ri ? int64_t(ri[r_a]) : r_a;
                            const int64_t row_b = live_b ? (ri ? int64_t(ri[r_b]) : r_b) : row_a;
                            ala[j] = a.alpha_per_row ? a.alpha[r_a * a.T + t] : a.alpha[t];
                            alb[j] = (a.alpha_per_row && live_b) ? a.alpha[r_b * a.T + t] : ala[j];
                            va[j] = *reinterpret_cast<const double2*>(a.tables[t] + row_a * a.lde + 2 * c);
                            vb[j] = *reinterpret_cast<const double2*>(a.tables[t] + row_b * a.lde + 2 * c);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (t0 + j < a.T) {      // uniform
                                acc_a.x = acc_a.x + ala[j] * va[j].x;
                                acc_a.y = acc_a.y + ala[j] * va[j].y;
                                acc_b.x = acc_b.x + alb[j] * vb[j].x;
                                acc_b.y = acc_b.y + alb[j] * vb[j].y;
                            }
                        }
                    }
                    *reinterpret_cast<double2*>(slab + slot_a * a.stride + 2 * c) = acc_a;
                    if (live_b) *reinterpret_cast<double2*>(slab + slot_b * a.stride + 2 * c) = acc_b;
                }
            }
        }
        for (int pb = 0; MODE != 2 && pb < a.pb_n; pb += rpi) {
            const int slot = pb + sub;
            const int64_t r = p0 + slot;
            if (slot >= a.pb_n || r >= a.R) continue;
            if constexpr (MODE == 2) {
            } else {
                const int64_t e0 = a.indptr[r], e1 = a.indptr[r + 1];
                const double deg = double(e1 - e0);
                for (int c = c0; c < a.d2; c += L) {
                    double2 acc = make_double2(0.0, 0.0);
                    int64_t e = e0;
                    for (; e + 4 <= e1; e += 4) {   // four neighbour rows in flight, added in adjacency order
                        const double2 v0 = *reinterpret_cast<const double2*>(a.E + int64_t(a.indices[e]) * a.lde + 2 * c);
                        const double2 v1 = *reinterpret_cast<const double2*>(a.E + int64_t(a.indices[e + 1]) * a.lde + 2 * c);
                        const double2 v2 = *reinterpret_cast<const double2*>(a.E + int64_t(a.indices[e + 2]) * a.lde + 2 * c);
                        const double2 v3 = *reinterpret_cast<const double2*>(a.E + int64_t(a.indices[e + 3]) * a.lde + 2 * c);
                        acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x;
                        acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
                    }
                    for (; e < e1; ++e) {
                        const double2 v = *reinterpret_cast<const double2*>(a.E + int64_t(a.indices[e]) * a.lde + 2 * c);
                        acc.x = acc.x + v.x;
                        acc.y = acc.y + v.y;
                    }
                    acc.x = acc.x / deg;              // IEEE division: mean(0) = add.reduce / count
                    acc.y = acc.y / deg;
                    if constexpr (MODE == 0) {
                        *reinterpret_cast<double2*>(a.out + r * a.ldo + 2 * c) = acc;
                    } else {
                        const double2 v = *reinterpret_cast<const double2*>(a.E + r * a.lde + 2 * c);
                        acc.x = (v.x + acc.x) / 2.0;
                        acc.y = (v.y + acc.y) / 2.0;
                        *reinterpret_cast<double2*>(slab + slot * a.stride + 2 * c) = acc;
                    }
                }
            }
        }
        if constexpr (MODE != 0) {
            wave_lds_sync();
            {
                const int p8 = lane >> 3, k = lane & 7;
                const double ss = np_pairwise_sq<kPairwiseDepth>(slab + min(p8, a.pb_n - 1) * a.stride, a.d, k);
                if (k == 0 && p8 < a.pb_n) s_nrm[p8] = sqrt(ss);   // no epsilon: a zero row divides by zero like hwer/utils.py:43-44
            }
            wave_lds_sync();
            for (int pb = 0; pb < a.pb_n; pb += rpi) {
                const int slot = pb + sub;
                const int64_t r = p0 + slot;
                if (slot >= a.pb_n || r >= a.R) continue;
                const double nrm = s_nrm[slot];
                for (int c = c0; c < a.d2; c += L) {
                    double2 v = *reinterpret_cast<const double2*>(slab + slot * a.stride + 2 * c);
                    v.x = v.x / nrm;
                    v.y = v.y / nrm;
                    *reinterpret_cast<double2*>(a.out + r * a.ldo + 2 * c) = v;
                }
                if (a.out_norm && c0 == 0) a.out_norm[r] = nrm;
            }
            wave_lds_sync();
        }
    }
}

// Round 6: the float32 ROUNDING of float64 rows as the screen's table - X (zero-padded to ldx columns, pad rows zero with a +inf half norm),
// hnorm = 0.5 ||x32||^2 (summed in float64 from the rounded elements, rounded once), optional float16 copy scaled by 2^x_exp - and the unit-norm
// audit of the float64 rows themselves (hwer/recommendation_base.py:106-107) in the same pass.  16 lanes per row, 16-byte loads.
struct Pack64Args {
    const double* V64;
    int64_t ld64;
    int d;
    const int64_t* row_ids;
    int64_t n_rows, n_out;
    float* X;
    int64_t ldx;
    float* hnorm;
    _Float16* X16;
    int x_exp;
    double eps;
    unsigned long long* counts;
    double* absdev;
    // audit_all (hwer_audit_pack_rows_f64rows): group g < n_table works on table row g - audited always, rounded into X[g - pack_row0] when it lies in the
    // searched range [pack_row0, pack_row0 + n_rows); groups n_table .. n_table + (n_out - n_rows) write the pad rows of X
    int audit_all;
    int64_t n_table, pack_row0;
};
__global__ __launch_bounds__(256) void pack_rows_f64rows_kernel(const Pack64Args a) {
    const int sub = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int x2 = int(a.ldx / 2), d2 = (a.d + 1) / 2;
    unsigned int n_pos = 0, n_neg = 0, n_nan = 0;   // this 16-lane group's audit tallies (lane 0), combined per block below: one atomic per block and counter
    double dev_sum = 0.0;
    const int64_t n_groups = a.audit_all ? a.n_table + (a.n_out - a.n_rows) : a.n_out;
    for (int64_t g = int64_t(blockIdx.x) * 16 + sub; g < n_groups; g += int64_t(gridDim.x) * 16) {   // (uniform per 16-lane group)
        // r: row of X this group writes (-1: none, audit only); live: the group reads a table row
        int64_t r = g;
        bool live = g < a.n_rows;
        const double* src = a.V64;
        if (a.audit_all) {
            live = g < a.n_table;
            r = live ? ((g >= a.pack_row0 && g < a.pack_row0 + a.n_rows) ? g - a.pack_row0 : -1) : a.n_rows + (g - a.n_table);
            src = a.V64 + (live ? g : 0) * a.ld64;
        } else {
            src = a.V64 + (live ? (a.row_ids ? a.row_ids[g] : g) : 0) * a.ld64;
        }
        const bool store = r >= 0;
        double ss32 = 0.0, ss64 = 0.0;
        for (int c = l; c < x2; c += 16) {
            double2 v = make_double2(0.0, 0.0);
            if (live && c < d2) {
                v = *reinterpret_cast<const double2*>(src + 2 * c);
                if (2 * c + 1 >= a.d) v.y = 0.0;
            }
            const float2 w = make_float2(float(v.x), float(v.y));
            ss64 += v.x * v.x + v.y * v.y;
            ss32 += double(w.x) * double(w.x) + double(w.y) * double(w.y);
            if (a.X && store) *reinterpret_cast<float2*>(a.X + r * a.ldx + 2 * c) = w;
            if (a.X16 && store) {
                typedef _Float16 half2_rt __attribute__((ext_vector_type(2)));
                *reinterpret_cast<half2_rt*>(a.X16 + r * a.ldx + 2 * c) = half2_rt{(_Float16)ldexpf(w.x, a.x_exp), (_Float16)ldexpf(w.y, a.x_exp)};
            }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            ss32 += shfl_xor_t(ss32, m);
            ss64 += shfl_xor_t(ss64, m);
        }
        if (l == 0) {
            if (a.hnorm && store) a.hnorm[r] = live ? float(0.5 * ss32) : INFINITY;
            if (live) {
                const double n = sqrt(ss64);
                n_pos += n > 1.0 + a.eps;
                n_neg += n < 1.0 - a.eps;
                n_nan += n != n;
                dev_sum += fabs(n - 1.0);
            }
        }
    }
    if (a.counts) {   // (kernel-uniform)
        __shared__ unsigned int s_c[3][16];
        __shared__ double s_d[16];
        if (l == 0) { s_c[0][sub] = n_pos; s_c[1][sub] = n_neg; s_c[2][sub] = n_nan; s_d[sub] = dev_sum; }
        __syncthreads();
        if (threadIdx.x < 3) {
            unsigned int t = 0;
            for (int i = 0; i < 16; ++i) t += s_c[threadIdx.x][i];
            if (t) atomicAdd(&a.counts[threadIdx.x], (unsigned long long)t);
        } else if (threadIdx.x == 3 && a.absdev) {
            double t = 0.0;
            for (int i = 0; i < 16; ++i) t += s_d[i];
            atomicAdd(a.absdev, t);
        }
    }
}

template <int MODE>
static int launch_rows64(const char* fn, Rows64Args a, void* stream) {
    a.d2 = (a.d + 1) / 2;
    int stride = 2 * a.d2;
    while (stride % 32 != 8) stride += 2;   // 4 rows x 8 chains of a ds_read_b64 group on 64 distinct banks; even: 16-byte slab rows
    int pb_n = 8;
    while (pb_n > 2 && size_t(pb_n) * stride * 8 > size_t(4800)) pb_n >>= 1;   // (pair_score's rule: narrow slabs keep more waves resident)
    a.stride = stride;
    a.pb_n = pb_n;
    const size_t per_wave = MODE == 0 ? 0 : (size_t(pb_n) * stride + 8) * 8;
    const int nw = 4;
    const size_t lds = per_wave * nw;
    if (lds > size_t(64) * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rows64_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "%s: hipFuncSetAttribute(%zu): %s", fn, lds, hipGetErrorString(e));
    }
    const int64_t n_trips = cdiv64(a.R, pb_n);
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(n_trips, nw), kMaxStreamBlocks * 2)));
    hipLaunchKernelGGL(rows64_kernel<MODE>, dim3(grid), dim3(64 * nw), lds, as_stream(stream), a);
    return check_launch(fn);
}

static int check_table64(const char* fn, const void* p, int64_t ld, int d) {
    if (!p) return fail(HWER_EINVAL, "%s: null table", fn);
    if (d < 1 || d > 1024) return fail(HWER_EINVAL, "%s: need 1 <= d <= 1024 (d=%d)", fn, d);
    if (ld % 2 != 0 || ld < 2 * ((d + 1) / 2)) return fail(HWER_EINVAL, "%s: the row stride must be even and >= roundup2(d) (ld=%lld, d=%d)", fn, (long long)ld, d);
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return fail(HWER_EINVAL, "%s: the table must be 16-byte aligned", fn);
    return HWER_OK;
}

}  // namespace hwer

using namespace hwer;

extern "C" {

int hwer_segment_mean_f64(const double* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N, int d, double* out, int64_t ldo,
                          void* stream) {
    if (N == 0) return HWER_OK;
    int rc;
    if ((rc = check_table64("segment_mean_f64(E)", E, lde, d))) return rc;
    if ((rc = check_table64("segment_mean_f64(out)", out, ldo, d))) return rc;
    if (!indptr || !indices) return fail(HWER_EINVAL, "segment_mean_f64: null CSR");
    Rows64Args a{};
    a.E = E; a.lde = lde; a.indptr = indptr; a.indices = indices; a.R = N; a.d = d; a.out = out; a.ldo = ldo;
    return launch_rows64<0>("segment_mean_f64", a, stream);
}

int hwer_content_tail_f64(const double* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N, int d, double* out, int64_t ldo,
                          void* stream) {
    if (N == 0) return HWER_OK;
    int rc;
    if ((rc = check_table64("content_tail_f64(E)", E, lde, d))) return rc;
    if ((rc = check_table64("content_tail_f64(out)", out, ldo, d))) return rc;
    if (!indptr || !indices) return fail(HWER_EINVAL, "content_tail_f64: null CSR");
    Rows64Args a{};
    a.E = E; a.lde = lde; a.indptr = indptr; a.indices = indices; a.R = N; a.d = d; a.out = out; a.ldo = ldo;
    return launch_rows64<1>("content_tail_f64", a, stream);
}

int hwer_blend_normalize_f64(const double* const* tables, int64_t ld_tables, const int32_t* const* row_idx, const double* alpha, int alpha_per_row,
                             int T, int64_t R, int d, double* out, int64_t ldo, double* out_norm, void* stream) {
    if (R == 0) return HWER_OK;
    if (!tables || !alpha || T < 1) return fail(HWER_EINVAL, "blend_normalize_f64: null tables/alpha or T < 1");
    int rc;
    if ((rc = check_table64("blend_normalize_f64(out)", out, ldo, d))) return rc;
    if (ld_tables % 2 != 0 || ld_tables < 2 * ((d + 1) / 2)) return fail(HWER_EINVAL, "blend_normalize_f64: ld_tables must be even and >= roundup2(d)");
    Rows64Args a{};
    a.tables = tables; a.row_idx = row_idx; a.alpha = alpha; a.alpha_per_row = alpha_per_row; a.T = T; a.lde = ld_tables; a.R = R; a.d = d;
    a.out = out; a.ldo = ldo; a.out_norm = out_norm;
    return launch_rows64<2>("blend_normalize_f64", a, stream);
}

int hwer_pack_rows_f64rows(const double* V64, int64_t N, int d, int64_t ld64, const int64_t* row_ids, int64_t n_rows, float* X, int64_t ldx,
                           float* hnorm, void* X16, int x_exp, float eps, uint64_t* audit_counts, double* audit_absdev_sum, void* stream) {
    int rc;
    if ((rc = check_table64("pack_rows_f64rows(V64)", V64, ld64, d))) return rc;
    const bool audit_only = !X && !hnorm && !X16;   // (only the unit-norm audit of the listed rows)
    if (audit_only && !audit_counts) return fail(HWER_EINVAL, "pack_rows_f64rows: nothing to do (no X / hnorm and no audit accumulators)");
    if (!audit_only && (!X || !hnorm)) return fail(HWER_EINVAL, "pack_rows_f64rows: null X / hnorm");
    if (audit_only) ldx = 2 * ((d + 1) / 2);
    if (ldx < d || ldx % 2 != 0 || (reinterpret_cast<uintptr_t>(X) & 7) || (X16 && (reinterpret_cast<uintptr_t>(X16) & 3)))
        return fail(HWER_EINVAL, "pack_rows_f64rows: ldx must be even and >= d, X 8-byte aligned");
    if (n_rows < 0 || (!row_ids && n_rows > N)) return fail(HWER_EINVAL, "pack_rows_f64rows: n_rows > N without row_ids");
    Pack64Args a{V64, ld64, d, row_ids, n_rows, audit_only ? n_rows : hwer_knn_padded_rows(n_rows), X, ldx, hnorm, static_cast<_Float16*>(X16), x_exp, double(eps),
                 reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev_sum, 0, 0, 0};
    if (a.n_out == 0) return HWER_OK;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(a.n_out, 16), kMaxStreamBlocks * 4)));
    hipLaunchKernelGGL(pack_rows_f64rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
    return check_launch("pack_rows_f64rows");
}

int hwer_audit_pack_rows_f64rows(const double* V64, int64_t N, int d, int64_t ld64, int64_t pack_row0, int64_t n_rows, float* X, int64_t ldx, float* hnorm,
                                 void* X16, int x_exp, float eps, uint64_t* audit_counts, double* audit_absdev_sum, void* stream) {
    int rc;
    if ((rc = check_table64("audit_pack_rows_f64rows(V64)", V64, ld64, d))) return rc;
    if (!X || !hnorm || !audit_counts) return fail(HWER_EINVAL, "audit_pack_rows_f64rows: null X / hnorm / audit_counts");
    if (ldx < d || ldx % 2 != 0 || (reinterpret_cast<uintptr_t>(X) & 7) || (X16 && (reinterpret_cast<uintptr_t>(X16) & 3)))
        return fail(HWER_EINVAL, "audit_pack_rows_f64rows: ldx must be even and >= d, X 8-byte aligned");
    if (N < 1 || n_rows < 1 || pack_row0 < 0 || pack_row0 + n_rows > N) return fail(HWER_EINVAL, "audit_pack_rows_f64rows: the searched rows must be a range of the table");
    Pack64Args a{V64, ld64, d, nullptr, n_rows, hwer_knn_padded_rows(n_rows), X, ldx, hnorm, static_cast<_Float16*>(X16), x_exp, double(eps),
                 reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev_sum, 1, N, pack_row0};
    const int64_t n_groups = N + (a.n_out - n_rows);
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(n_groups, 16), kMaxStreamBlocks * 4)));
    hipLaunchKernelGGL(pack_rows_f64rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
    return check_launch("audit_pack_rows_f64rows");
}

int hwer_pair_score_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst, int64_t P, double* out,
                            void* stream) {
    if (P == 0) return HWER_OK;
    if (!V64 || !src || !dst || !out) return fail(HWER_EINVAL, "pair_score_f64rows: null pointer");
    if (N < 1 || d < 1 || d > 1024 || ldv < d) return fail(HWER_EINVAL, "pair_score_f64rows: need N >= 1, 1 <= d <= 1024, ldv >= d");
    int stride = d;
    while (stride % 16 != 4) ++stride;   // the 8 rows of a wave start on different banks
    const int nw = size_t(4) * 8 * stride * 8 <= size_t(64) * 1024 ? 4 : 1;
    const size_t lds = size_t(nw) * 8 * stride * 8;
    if (lds > size_t(64) * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_score_f64rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "pair_score_f64rows: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(P, 8 * nw), kMaxStreamBlocks * 2)));
    hipLaunchKernelGGL(pair_score_f64rows_kernel, dim3(grid), dim3(64 * nw), lds, as_stream(stream), V64, d, ldv, src, dst, P, out, stride);
    return check_launch("pair_score_f64rows");
}

int hwer_knn_cand_dist2(const void* table, int table_is_f64, int64_t n_rows, int d, int64_t ld, const int64_t* cand, int64_t ld_cand, int64_t B,
                        int64_t kc, const double* Q64, double* out, void* stream) {
    if (B == 0 || kc == 0) return HWER_OK;
    if (!table || !cand || !Q64 || !out) return fail(HWER_EINVAL, "knn_cand_dist2: null pointer");
    if (n_rows < 1 || d < 1 || ld < d || ld_cand < kc || B > 65535) return fail(HWER_EINVAL, "knn_cand_dist2: need n_rows >= 1, ld >= d >= 1, ld_cand >= kc, B <= 65535");
    const dim3 grid(unsigned(std::min<int64_t>(cdiv64(kc, 256), 1024)), unsigned(B));
    if (!table_is_f64 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0)
        return launch_cand_dist2_f32x4(static_cast<const float*>(table), ld, d, cand, ld_cand, B, kc, Q64, out, as_stream(stream));
    if (table_is_f64 && ld % 2 == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0 && kc >= 32) {
        const dim3 grid_s(unsigned(std::min<int64_t>(cdiv64(kc, 256), 1024)), unsigned(B));
        hipLaunchKernelGGL(cand_dist2_f64_staged_kernel, grid_s, dim3(256), 0, as_stream(stream), static_cast<const double*>(table), ld, d, cand, ld_cand, kc, Q64, out);
    } else if (table_is_f64)
        hipLaunchKernelGGL(cand_dist2_kernel<double>, grid, dim3(256), 0, as_stream(stream), static_cast<const double*>(table), ld, d, cand, ld_cand, kc, Q64, out);
    else
        hipLaunchKernelGGL(cand_dist2_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(table), ld, d, cand, ld_cand, kc, Q64, out);
    return check_launch("knn_cand_dist2");
}

}  // extern "C"
