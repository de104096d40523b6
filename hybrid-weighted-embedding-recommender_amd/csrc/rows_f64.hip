// float64 `vectors` (round 3; SURVEY.md 8c "stage-wise" -> true float64 flavour).  The reference keeps ContentRecommendation's rows
// in float64 (hwer/content_recommender.py:96, hwer/recommendation_base.py:109) and both the KD-tree (:74, :79) and predict (:135-144)
// work on them.  The MFMA screen still runs on the float32 / float16 roundings of the rows (a screen only has to keep a superset);
// the kernels here produce the reference's VALUES from the float64 rows:
//   hwer_query_mix_f64rows   (rowops.hip)  queries from float64 rows, numpy's orders
//   hwer_pair_score_f64rows               predict on float64 rows, numpy's pairwise order
//   hwer_knn_cand_dist2                   exact squared distances of listed candidate rows (float32 or float64 table) in
//                                         scikit-learn's sequential order - the re-rank of the candidate sets that the float32 search
//                                         (d <= 256, k <= 392) or a library GEMM screen (wider tables, larger k) hands over
// None of them is on the throughput path of the float32 flavour; they are written for clarity, HBM-coalescing where it is free.
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

// knn_wide.hip: the same arithmetic with 16-byte row loads (float32 tables whose rows are 16-byte aligned)
int launch_cand_dist2_f32x4(const float* table, int64_t ld, int d, const int64_t* cand, int64_t ld_cand, int64_t B, int64_t kc, const double* Q64,
                            double* out, hipStream_t st);

}  // namespace hwer

using namespace hwer;

extern "C" {

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
    if (table_is_f64)
        hipLaunchKernelGGL(cand_dist2_kernel<double>, grid, dim3(256), 0, as_stream(stream), static_cast<const double*>(table), ld, d, cand, ld_cand, kc, Q64, out);
    else
        hipLaunchKernelGGL(cand_dist2_kernel<float>, grid, dim3(256), 0, as_stream(stream), static_cast<const float*>(table), ld, d, cand, ld_cand, kc, Q64, out);
    return check_launch("knn_cand_dist2");
}

}  // extern "C"
