// Fit-side table construction on the device: the data passes of
//   IncrementalPCA(n_components, batch_size=2**16).fit_transform(table)          hwer/content_recommender.py:68-69
//   unit_length(., axis=1) + the one-hot node-type columns                          hwer/content_recommender.py:70-74
// over the reference's block-diagonal feature table (hwer/content_recommender.py:50-66) WITHOUT materialising it: a row
// of node type t is non-zero only in the columns of block t and in one-hot column t, so
//   * the Gram matrix of any row range is block diagonal over the types: one (D_t+1) x (D_t+1) Gram of [B_t | 1] per
//     (batch, type) - `segment_gram_kernel`, float64 MFMA (v_mfma_f64_16x16x4_f64), deterministic two-stage reduction;
//   * the projection of a row is B_t[l] . P_t + bias_t with P_t = the block's columns of the components -
//     `pca_project_kernel`, float64 MFMA, fused with the row normalisation and the one-hot / padding columns.
// The O(W^3) eigen-recurrence between the two (W = table width) runs on the host (hwer/fit_pca.py).
#include <algorithm>

#include "hwer_common.h"

namespace hwer {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int kGramTile = 64;     // output tile of a workgroup: 4 x 4 MFMA tiles per wave
constexpr int kGramPF = 3;        // row steps in flight per wave (register ring of 4 operand sets)

__host__ __device__ static inline int gram_padded(int D) { return (D + 15) & ~15; }

// rows of one partial Gram (one workgroup): 8 192 (a 65 536-row batch -> 8 partials), fewer on small inputs so that the
// launch still has ~1 000 workgroups
static inline int gram_chunk_rows(int D, int n_seg, int64_t max_seg_rows) {
    const int nt = (gram_padded(D) + kGramTile - 1) / kGramTile;
    const int64_t npair = int64_t(nt) * (nt + 1) / 2;
    int rows = 8192;
    while (rows > 1024 && int64_t(n_seg) * ((max_seg_rows + rows - 1) / rows) * npair < 1024) rows >>= 1;
    return rows;
}

// Partial layout per (segment, chunk): Dp x Dp Gram (upper-triangle elements of the tile pairs ti <= tj), then Dp column sums.
// grid: x = chunk inside the segment, y = segment, z = upper-triangle tile pair.  DIAG (ti == tj): one operand set, only the
// sub-tiles x <= y, and the column sums of the tile's columns (VALU adds next to the MFMAs).
template <bool DIAG>
__device__ __forceinline__ void gram_tile(const double* __restrict__ B, int64_t ldb, int D, int64_t r0, int64_t r1, int ti, int tj,
                                          int Dp, double* __restrict__ out, double* tile) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kr = lane >> 4, c = lane & 15;
    const int na = min(4, (Dp - ti * kGramTile) / 16), nb = min(4, (Dp - tj * kGramTile) / 16);
    const int ca = ti * kGramTile + c, cb = tj * kGramTile + c;
    d4_t acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = d4_t{0.0, 0.0, 0.0, 0.0};
    double csum[4] = {0.0, 0.0, 0.0, 0.0};
    double fa[4][4], fb[DIAG ? 1 : 4][4];
    const int64_t first = r0 + wave * 4;
    const int nsteps = first < r1 ? int((r1 - first + 15) / 16) : 0;
    auto fetch = [&](int step, double (&va)[4], double (&vb)[4]) {
        const int64_t r = first + int64_t(step) * 16 + kr;
        const bool valid = r < r1;
        const double* p = B + (valid ? r : r0) * ldb;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            va[x] = (valid && x < na && ca + 16 * x < D) ? p[ca + 16 * x] : 0.0;
            if (!DIAG) vb[x] = (valid && x < nb && cb + 16 * x < D) ? p[cb + 16 * x] : 0.0;
        }
    };
#pragma unroll
    for (int u = 0; u < kGramPF; ++u)
        if (u < nsteps) fetch(u, fa[u], fb[DIAG ? 0 : u]);
    for (int base = 0; base < nsteps; base += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int step = base + u;
            if (step < nsteps) {   // wave-uniform
                if (step + kGramPF < nsteps) fetch(step + kGramPF, fa[(u + kGramPF) & 3], fb[DIAG ? 0 : ((u + kGramPF) & 3)]);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    if (x < na) {
                        if (DIAG) csum[x] += fa[u][x];
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            if (y < nb && (!DIAG || x <= y))
                                acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[u][x], DIAG ? fa[u][y] : fb[DIAG ? 0 : u][y], acc[x][y], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // fixed-order reduction over the 4 waves through LDS (deterministic)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * x + kr + 4 * r, j = 16 * y + c;   // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 reg
                        if (w == 0) tile[i * kGramTile + j] = acc[x][y][r];
                        else tile[i * kGramTile + j] += acc[x][y][r];
                    }
            if (DIAG) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    double v = csum[x];                  // rows kr, kr + 4, ... of this wave: add the 4 row groups in lane order
                    const double v1 = __shfl(v, c + 16, kWave), v2 = __shfl(v, c + 32, kWave), v3 = __shfl(v, c + 48, kWave);
                    v = ((__shfl(v, c, kWave) + v1) + v2) + v3;
                    if (kr == 0) {
                        double* sl = tile + kGramTile * kGramTile + 16 * x + c;
                        if (w == 0) *sl = v; else *sl += v;
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < kGramTile * kGramTile; e += 256) {
        const int i = e >> 6, j = e & 63, gi = ti * kGramTile + i, gj = tj * kGramTile + j;
        if (gi < Dp && gj < Dp && (!DIAG || (i >> 4) <= (j >> 4))) out[int64_t(gi) * Dp + gj] = tile[e];
    }
    if (DIAG && threadIdx.x < kGramTile && ti * kGramTile + threadIdx.x < Dp)
        out[int64_t(Dp) * Dp + ti * kGramTile + threadIdx.x] = tile[kGramTile * kGramTile + threadIdx.x];
}

// grid: x = tile pair (fastest: the workgroups that read the same rows are dispatched together, so all but the first read
// of a row chunk are served by the L2 / Infinity Cache), y = chunk inside the segment, z = segment
__global__ __launch_bounds__(256, 2) void segment_gram_kernel(const double* __restrict__ B, int64_t ldb, int D,
                                                              const int64_t* __restrict__ seg_ptr, int chunks_per_seg, int chunk_rows,
                                                              double* __restrict__ ws, int Dp, int ntile) {
    __shared__ double tile[kGramTile * kGramTile + kGramTile];
    const int seg = blockIdx.z, chunk = blockIdx.y;
    const int64_t s1 = seg_ptr[seg + 1];
    const int64_t r0 = seg_ptr[seg] + int64_t(chunk) * chunk_rows;
    if (r0 >= s1) return;
    const int64_t r1 = (r0 + chunk_rows < s1) ? r0 + chunk_rows : s1;
    int ti = 0, z = blockIdx.x;
    while (z >= ntile - ti) { z -= ntile - ti; ++ti; }
    const int tj = ti + z;
    double* out = ws + (int64_t(seg) * chunks_per_seg + chunk) * (int64_t(Dp) * Dp + Dp);
    if (ti == tj) gram_tile<true>(B, ldb, D, r0, r1, ti, tj, Dp, out, tile);
    else gram_tile<false>(B, ldb, D, r0, r1, ti, tj, Dp, out, tile);
}

// G[seg] (Da x Da, Da = D + 1): Gram entries = sum of the segment's partials in chunk order (upper-triangle element mirrored),
// row / column D = the column sums, G[D][D] = the row count
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ ws, const int64_t* __restrict__ seg_ptr,
                                                          int chunks_per_seg, int chunk_rows, int Dp, int Da, double* __restrict__ G) {
    const int seg = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Da * Da) return;
    const int D = Da - 1;
    const int i = e / Da, j = e % Da;
    const int64_t rows = seg_ptr[seg + 1] - seg_ptr[seg];
    double s;
    if (i == D && j == D) {
        s = double(rows);
    } else {
        const int nchunk = int((rows + chunk_rows - 1) / chunk_rows);
        const int64_t stride = int64_t(Dp) * Dp + Dp;
        const int lo = min(i, j), hi = max(i, j);
        const double* p = ws + int64_t(seg) * chunks_per_seg * stride + (hi == D ? int64_t(Dp) * Dp + lo : int64_t(lo) * Dp + hi);
        s = 0.0;
        for (int cidx = 0; cidx < nchunk; ++cidx) s += p[int64_t(cidx) * stride];
    }
    G[int64_t(seg) * Da * Da + e] = s;
}

// One wave = 16 rows of the block x all (padded) components.  A = B rows (16 x 4 per step), B operand = P rows.
template <int NT>
__global__ __launch_bounds__(256) void pca_project_kernel(const double* __restrict__ B, int64_t ldb, int64_t n_rows, int D,
                                                          const double* __restrict__ P, int64_t ldp,
                                                          const double* __restrict__ bias, int n,
                                                          const int64_t* __restrict__ out_rows, int onehot_col, int n_onehot,
                                                          float* __restrict__ out32, int64_t ldo32,
                                                          double* __restrict__ out64, int64_t ldo64) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kr = lane >> 4, c = lane & 15;
    const int64_t R0 = (int64_t(blockIdx.x) * 4 + wave) * 16;
    if (R0 >= n_rows) return;
    d4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = d4_t{0.0, 0.0, 0.0, 0.0};
    const int64_t arow = R0 + c;
    const bool avalid = arow < n_rows;
    const double* ap = B + (avalid ? arow : R0) * ldb;
    const int D4 = (D + 3) & ~3;
    for (int c0 = 0; c0 < D4; c0 += 4) {
        const int col = c0 + kr;
        const double a = (avalid && col < D) ? ap[col] : 0.0;
        const double* pp = P + int64_t(col) * ldp + c;     // P has D4 rows (zero padded) and 16 NT columns
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pp[16 * t], acc[t], 0, 0, 0);
    }
    // lane holds rows kr + 4 r (r = 0..3), components 16 t + c
    double nrm2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const double bj = bias[16 * t + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[t][r] += bj;
            nrm2[r] += acc[t][r] * acc[t][r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        for (int m = 1; m < 16; m <<= 1) nrm2[r] += __shfl_xor(nrm2[r], m, kWave);
        nrm2[r] = sqrt(nrm2[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t lrow = R0 + kr + 4 * r;
        if (lrow >= n_rows) continue;
        const int64_t orow = out_rows ? out_rows[lrow] : lrow;
        float* o32 = out32 + orow * ldo32;
        double* o64 = out64 ? out64 + orow * ldo64 : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = 16 * t + c;
            if (j < n) {
                const double u = acc[t][r] / nrm2[r];      // 0/0 = NaN like numpy (hwer/utils.py:44 has no epsilon)
                o32[j] = float(u);
                if (o64) o64[j] = u;
            }
        }
        for (int q = n + c; q < ldo32; q += 16) o32[q] = (q - n == onehot_col) ? 1.f : 0.f;
        if (o64) for (int q = n + c; q < ldo64; q += 16) o64[q] = (q - n == onehot_col) ? 1.0 : 0.0;
    }
}

}  // namespace hwer

using namespace hwer;

extern "C" {

int hwer_gram_padded_dim(int D) { return D < 1 ? -1 : gram_padded(D); }

size_t hwer_segment_gram_workspace_bytes(int D, int n_seg, int64_t max_seg_rows) {
    if (D < 1 || n_seg < 1 || max_seg_rows < 1) return 0;
    const size_t Dp = size_t(gram_padded(D));
    const int rows = gram_chunk_rows(D, n_seg, max_seg_rows);
    return size_t(n_seg) * size_t((max_seg_rows + rows - 1) / rows) * (Dp * Dp + Dp) * sizeof(double);
}

int hwer_segment_gram_f64(const double* B, int64_t ldb, int D, const int64_t* seg_ptr, int n_seg, int64_t max_seg_rows,
                          double* G, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_seg == 0) return HWER_OK;
    if (!B || !seg_ptr || !G || !workspace) return fail(HWER_EINVAL, "segment_gram: null pointer");
    if (D < 1 || D > 4095 || ldb < D) return fail(HWER_EINVAL, "segment_gram: bad D=%d / ldb=%lld", D, (long long)ldb);
    if (n_seg < 0 || n_seg > 65535 || max_seg_rows < 1) return fail(HWER_EINVAL, "segment_gram: bad n_seg=%d / max_seg_rows=%lld", n_seg, (long long)max_seg_rows);
    const int chunk_rows = gram_chunk_rows(D, n_seg, max_seg_rows);
    const int chunks = int(std::min<int64_t>((max_seg_rows + chunk_rows - 1) / chunk_rows, 1 << 20));
    if (chunks > 65535) return fail(HWER_EINVAL, "segment_gram: segment too long (%lld rows)", (long long)max_seg_rows);
    if (workspace_bytes < hwer_segment_gram_workspace_bytes(D, n_seg, max_seg_rows)) return fail(HWER_EWORKSPACE, "segment_gram: workspace too small");
    const int Dp = gram_padded(D), Da = D + 1;
    const int ntile = (Dp + kGramTile - 1) / kGramTile;
    hipLaunchKernelGGL(segment_gram_kernel, dim3(ntile * (ntile + 1) / 2, chunks, n_seg), dim3(256), 0, as_stream(stream), B, ldb, D, seg_ptr,
                       chunks, chunk_rows, static_cast<double*>(workspace), Dp, ntile);
    int rc = check_launch("segment_gram");
    if (rc) return rc;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((Da * Da + 255) / 256, n_seg), dim3(256), 0, as_stream(stream),
                       static_cast<const double*>(workspace), seg_ptr, chunks, chunk_rows, Dp, Da, G);
    return check_launch("gram_reduce");
}

int hwer_pca_padded_components(int n) {
    if (n < 1 || n > 256) return -1;
    int nt = 1;
    while (16 * nt < n) nt <<= 1;
    return 16 * nt;
}

int hwer_pca_project_unit_f64(const double* B, int64_t ldb, int64_t n_rows, int D, const double* P, int64_t ldp,
                              const double* bias, int n, const int64_t* out_rows, int onehot_col, int n_onehot, float* out32,
                              int64_t ldo32, double* out64, int64_t ldo64, void* stream) {
    if (n_rows == 0) return HWER_OK;
    if (!B || !P || !bias || !out32) return fail(HWER_EINVAL, "pca_project: null pointer");
    const int np = hwer_pca_padded_components(n);
    if (np < 0) return fail(HWER_EINVAL, "pca_project: n_components=%d unsupported (1..256)", n);
    if (D < 1 || ldb < D || ldp < np) return fail(HWER_EINVAL, "pca_project: bad D=%d ldb=%lld ldp=%lld (need ldp >= %d)", D, (long long)ldb, (long long)ldp, np);
    if (n_onehot < 0 || onehot_col < -1 || onehot_col >= n_onehot) return fail(HWER_EINVAL, "pca_project: bad one-hot column %d of %d", onehot_col, n_onehot);
    if (ldo32 < n + n_onehot || (out64 && ldo64 < n + n_onehot)) return fail(HWER_EINVAL, "pca_project: output rows too short");
    if (n_rows < 0 || n_rows > (int64_t(1) << 36)) return fail(HWER_EINVAL, "pca_project: bad n_rows");
    const unsigned grid = unsigned((n_rows + 63) / 64);
#define HWER_LAUNCH_PROJ(NT)                                                                                                  \
    hipLaunchKernelGGL(pca_project_kernel<NT>, dim3(grid), dim3(256), 0, as_stream(stream), B, ldb, n_rows, D, P, ldp, bias, n, \
                       out_rows, onehot_col, n_onehot, out32, ldo32, out64, ldo64)
    switch (np / 16) {
        case 1: HWER_LAUNCH_PROJ(1); break;
        case 2: HWER_LAUNCH_PROJ(2); break;
        case 4: HWER_LAUNCH_PROJ(4); break;
        case 8: HWER_LAUNCH_PROJ(8); break;
        default: HWER_LAUNCH_PROJ(16); break;
    }
#undef HWER_LAUNCH_PROJ
    return check_launch("pca_project");
}

}  // extern "C"
