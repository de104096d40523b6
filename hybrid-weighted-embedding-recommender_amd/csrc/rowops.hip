// HBM-bound row kernels of the hwer hot path for gfx950 (wave64): pair scoring, blend + L2-normalise,
// neighbour mean, fused content tail, unit-norm audit, per-type table packing, query construction.
// Each is bounded by HBM bandwidth; algorithmic bytes per unit are listed in DESIGN.md §4.
#include <algorithm>

#include "hwer_common.h"
#include "knn_internal.h"   // WS_HEADER_BYTES: where a search workspace keeps its query fragments (hwer_prepare_search_f32)

namespace hwer {

// =====================================================================================================
// error plumbing
// =====================================================================================================
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return HWER_OK;
}

// unknown node (index -1): row 0 clipped to [1e-6, 1e-5]   (hwer/recommendation_base.py:147-150).
// np.clip works in the dtype of `vectors`: the bounds are the float64 values 1e-6 / 1e-5 for float64 vectors and
// their float32 roundings for float32 vectors, so the clip happens AFTER the widening to T.
template <typename T>
__device__ __forceinline__ T clip_unknown(T v) {
    const T lo = T(1e-6), hi = T(1e-5);
    return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ float4 load_row4(const float* V, int64_t ldv, int32_t idx, int c) {
    return *reinterpret_cast<const float4*>(V + int64_t(idx < 0 ? 0 : idx) * ldv + 4 * c);
}
template <typename T>
__device__ __forceinline__ T widen(float v, bool unknown) {
    return unknown ? clip_unknown<T>(T(v)) : T(v);
}

__device__ __forceinline__ int pow2ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// =====================================================================================================
// K1+K2 pair score.  One wave = 8 pairs per step.
//   load phase : rows are read as float4, coalesced (L lanes per row, 64/L rows per instruction), the element
//                products are written to an LDS slab [8][stride];
//   chain phase: 8 lanes per pair replay numpy's pairwise order on the products (np_pairwise).
// Bytes per pair: 2*4*d (rows) + 8 (ids) + sizeof(T) (out).
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pair_score_kernel(const float* __restrict__ V, int d, int d4, int64_t ldv,
                                                         const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ dst, int64_t P,
                                                         T* __restrict__ out, int stride, int pb_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // pb_n = pairs per wave and trip: 8, or 4 / 2 for wide rows - the product slab of a trip then stays below ~9 KB per wave, and the CU
    // keeps 16 waves instead of 8 (d = 256, float64: round 4).  The kernel waits for gathered rows; waves in flight hide that, the chain
    // phase's idle lanes (8 lanes per pair, pb_n pairs) cost issue slots it does not need.
    T* slab = reinterpret_cast<T*>(smem_raw) + size_t(wave) * pb_n * stride;

    const int L = min(64, pow2ceil(d4));  // lanes per row in the load phase
    const int rows_per_instr = 64 / L;
    const int sub = lane / L;             // which of the rows_per_instr rows this lane loads
    const int c0 = lane % L;

    const int64_t n_batches = cdiv64(P, pb_n);
    const int nw = int(blockDim.x) >> 6;   // 4 waves, fewer when a wave's slab is large (wide float64 rows)
    const int64_t wave_global = int64_t(blockIdx.x) * nw + wave;
    const int64_t wave_stride = int64_t(gridDim.x) * nw;

    for (int64_t b = wave_global; b < n_batches; b += wave_stride) {
        const int64_t p0 = b * pb_n;
        // (round 3: issuing all 16 row loads of a step before the first product - 64 more VGPRs - measured SLOWER: 0.72 -> 0.50 of the
        // HBM roofline on the 1.25 M-row table, 0.40 -> 0.38 at d = 256; the loop below already keeps several waves' loads in flight)
        for (int pb = 0; pb < pb_n; pb += rows_per_instr) {
            const int64_t p = p0 + pb + sub;
            if (p < P) {
                const int32_t s = src[p], t = dst[p];
                for (int c = c0; c < d4; c += L) {
                    const float4 a = load_row4(V, ldv, s, c);
                    const float4 bb = load_row4(V, ldv, t, c);
                    T* w = slab + (pb + sub) * stride + 4 * c;
                    const bool us = s < 0, ut = t < 0;
                    w[0] = widen<T>(a.x, us) * widen<T>(bb.x, ut);
                    w[1] = widen<T>(a.y, us) * widen<T>(bb.y, ut);
                    w[2] = widen<T>(a.z, us) * widen<T>(bb.z, ut);
                    w[3] = widen<T>(a.w, us) * widen<T>(bb.w, ut);
                }
            }
        }
        wave_lds_sync();
        const int p8 = lane >> 3, k = lane & 7;
        const T r = np_pairwise<T, kPairwiseDepth>(slab + min(p8, pb_n - 1) * stride, d, k);
        if (k == 0 && p8 < pb_n && p0 + p8 < P) out[p0 + p8] = (r + T(1)) / T(2);
        wave_lds_sync();
    }
}

template <typename T>
static int launch_pair_score(const float* V, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst,
                             int64_t P, T* out, void* stream) {
    if (P == 0) return HWER_OK;
    if (!V || !src || !dst || !out) return fail(HWER_EINVAL, "pair_score: null pointer");
    if (N < 1 || d < 1 || d > 1024) return fail(HWER_EINVAL, "pair_score: need N >= 1 and 1 <= d <= 1024 (d=%d)", d);
    if (ldv % 4 != 0 || ldv < d) return fail(HWER_EINVAL, "pair_score: ldv must be a multiple of 4 and >= d");
    if ((reinterpret_cast<uintptr_t>(V) & 15) != 0) return fail(HWER_EINVAL, "pair_score: V must be 16-byte aligned");
    const int d4 = int((d + 3) / 4);
    int stride = 4 * d4;
    while (stride % 32 != 8) ++stride;  // chain reads of 4 pairs x 8 chains hit 32 distinct banks
    // one slab of 8 product rows per wave; 4 waves per workgroup while they fit the CU's 160 KB of LDS (d = 1024 in
    // float64: 66 KB per wave -> 2 waves)
    // pairs per wave and trip: 8 while a trip's slab stays below ~4.7 KB (d = 128 in float32, d = 64 in float64), else 4 or 2 - measured on tables
    // beyond the Infinity Cache (tools/pair_probe.py, fraction of 8 TB/s at 8 / 4 / 2 pairs): d = 256 float64 0.38 / 0.57 / 0.58, d = 256 float32 0.64 /
    // 0.73 / 0.65, d = 128 float64 0.61 / 0.67 / 0.58, d = 128 float32 0.77 / 0.72 / 0.66 (HWER_PAIR_PB overrides in diagnostic builds)
    int pb_n = 8;
    while (pb_n > 2 && size_t(pb_n) * stride * sizeof(T) > size_t(4800)) pb_n >>= 1;
    if (const char* e = diag_env("HWER_PAIR_PB")) pb_n = std::max(1, std::min(8, atoi(e)));
    const size_t slab = size_t(pb_n) * stride * sizeof(T);
    int nw = 4;
    while (nw > 1 && size_t(nw) * slab > size_t(160) * 1024) nw >>= 1;
    const size_t lds = size_t(nw) * slab;
    if (lds > size_t(160) * 1024) return fail(HWER_EINVAL, "pair_score: d=%d needs %zu bytes of LDS per wave", d, slab);
    if (lds > size_t(64) * 1024) {   // per device, cheap: the runtime caches the attribute
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_score_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "pair_score: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    const int64_t n_batches = cdiv64(P, pb_n);
    const int grid = int(std::min<int64_t>(cdiv64(n_batches, nw), kMaxStreamBlocks * 2));
    hipLaunchKernelGGL(pair_score_kernel<T>, dim3(grid), dim3(64 * nw), lds, as_stream(stream), V, d, d4, ldv, src, dst,
                       P, out, stride, pb_n);
    return check_launch("pair_score");
}

// =====================================================================================================
// row-group helpers: L = pow2 lanes cooperate on one row, each lane owns float4 chunks c = c0, c0+L, ...
// (at most MAXC chunks per lane: d <= 4*64*MAXC)
// =====================================================================================================
constexpr int MAXC = 4;  // d <= 1024

// butterfly sum over the L (power of two) lanes of a row group, result in every lane.  The exchanges are DPP / permlane moves
// (lane_xor32: one or two VALU instructions, ~10 cycles of latency) instead of ds_bpermute round trips through the LDS pipe (~100
// cycles each): a row kernel's trip is load -> reduce -> normalise -> store, and with 5-6 dependent exchanges per reduction and two
// or three reductions per row the bpermute chain was most of the time a wave spent with no load in flight (round 3).  Same lanes
// paired in the same order: bit-identical sums.
__device__ __forceinline__ float xor_lane_f(float v, int m) {
    const uint32_t u = __float_as_uint(v);
    switch (m) {
        case 1: return __uint_as_float(lane_xor32<1>(u));
        case 2: return __uint_as_float(lane_xor32<2>(u));
        case 4: return __uint_as_float(lane_xor32<4>(u));
        case 8: return __uint_as_float(lane_xor32<8>(u));
        case 16: return __uint_as_float(lane_xor32<16>(u));
        default: return __uint_as_float(lane_xor32<32>(u));
    }
}
__device__ __forceinline__ float group_sum(float v, int L) {
    if (L > 1) v += xor_lane_f(v, 1);
    if (L > 2) v += xor_lane_f(v, 2);
    if (L > 4) v += xor_lane_f(v, 4);
    if (L > 8) v += xor_lane_f(v, 8);
    if (L > 16) v += xor_lane_f(v, 16);
    if (L > 32) v += xor_lane_f(v, 32);
    return v;
}

struct RowGroup {
    int L, sub, c0, groups_per_wave;
    __device__ __forceinline__ RowGroup(int d4) {
        const int lane = threadIdx.x & 63;
        L = min(64, pow2ceil(d4));
        groups_per_wave = 64 / L;
        sub = lane / L;
        c0 = lane % L;
    }
};

// K9+K10 (+ optional norms): out[r] = unit(sum_t alpha * T_t[idx_t[r]])
// NC = float4 chunks per lane (1 for d <= 256, MAXC above), U = rows a row group works on per iteration: with one chunk per
// lane a wave had a single 16-byte load in flight per lane and table, and the kernel sat at ~4.5 TB/s on latency; four rows per
// iteration put 4 T loads in flight before the first reduction.  The arithmetic of a row is unchanged.
template <int NC, int U>
__global__ __launch_bounds__(256) void blend_normalize_kernel(const float* const* __restrict__ tables,
                                                              int64_t ldt, const int32_t* const* __restrict__ row_idx,
                                                              const float* __restrict__ alpha, int alpha_per_row,
                                                              int T, int64_t R, int d4, float* __restrict__ out,
                                                              int64_t ldo, float* __restrict__ out_norm, float audit_eps,
                                                              unsigned long long* __restrict__ audit_counts,
                                                              double* __restrict__ audit_absdev) {
    const RowGroup g(d4);
    unsigned long long a_pos = 0, a_neg = 0, a_nan = 0;   // fused K11 audit of the rows this thread finishes
    double a_dev = 0.0;
    const int64_t group_global = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * g.groups_per_wave + g.sub;
    const int64_t group_stride = int64_t(gridDim.x) * 4 * g.groups_per_wave;
    for (int64_t rb = group_global - g.sub; rb < R; rb += group_stride * U) {  // rb: wave-uniform base row of the first of U row groups
        float4 acc[U][NC];
        bool live[U];
        int64_t rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rr[u] = rb + int64_t(u) * group_stride + g.sub;
            live[u] = rr[u] < R;
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int t = 0; t < T; ++t) {
            const int32_t* ri = row_idx ? row_idx[t] : nullptr;
            const float* tab = tables[t];
            float4 v[U][NC];
            float al[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {   // all loads of this table first
                al[u] = 0.f;
                if (live[u]) {
                    al[u] = alpha_per_row ? alpha[rr[u] * T + t] : alpha[t];
                    const int64_t row = ri ? int64_t(ri[rr[u]]) : rr[u];
                    const float* base = tab + row * ldt;
#pragma unroll
                    for (int i = 0; i < NC; ++i) {
                        const int c = g.c0 + i * g.L;
                        v[u][i] = c < d4 ? *reinterpret_cast<const float4*>(base + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NC; ++i) v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    acc[u][i].x = fmaf(al[u], v[u][i].x, acc[u][i].x);
                    acc[u][i].y = fmaf(al[u], v[u][i].y, acc[u][i].y);
                    acc[u][i].z = fmaf(al[u], v[u][i].z, acc[u][i].z);
                    acc[u][i].w = fmaf(al[u], v[u][i].w, acc[u][i].w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rb + int64_t(u) * group_stride >= R) break;   // wave-uniform: this row group lies past the table
            const int64_t r = rr[u];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < NC; ++i) ss += acc[u][i].x * acc[u][i].x + acc[u][i].y * acc[u][i].y + acc[u][i].z * acc[u][i].z + acc[u][i].w * acc[u][i].w;
            ss = group_sum(ss, g.L);
            const float nrm = sqrtf(ss);  // no epsilon: a zero row becomes NaN like hwer/utils.py:43-44
            // one correctly rounded reciprocal per row, then multiplies: <= 1 ulp from x / nrm (the stage's contract is 1e-6 against
            // the float64 oracle) instead of ~10 VALU instructions per IEEE division and element; 0 * inf = NaN keeps a zero row NaN
            const float inv = 1.0f / nrm;
            float s2 = 0.f;
            if (live[u]) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int c = g.c0 + i * g.L;
                    if (c < d4) {
                        const float4 o = make_float4(acc[u][i].x * inv, acc[u][i].y * inv, acc[u][i].z * inv, acc[u][i].w * inv);
                        *reinterpret_cast<float4*>(out + r * ldo + 4 * c) = o;
                        s2 += o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;   // norm of the row as STORED (float32): what a separate audit pass would read back
                    }
                }
                if (out_norm && g.c0 == 0) out_norm[r] = nrm;
            }
            if (audit_counts) {
                s2 = group_sum(s2, g.L);
                if (live[u] && g.c0 == 0) {
                    const float n = sqrtf(s2);
                    a_pos += double(n) > 1.0 + double(audit_eps);
                    a_neg += double(n) < 1.0 - double(audit_eps);
                    a_nan += (n != n);
                    a_dev += fabs(double(n) - 1.0);
                }
            }
        }
    }
    if (audit_counts) {
        // One set of atomics per BLOCK: thousands of float64 atomics on a single address serialise in L2 and were the
        // whole cost of this kernel on small tables.  Counts are integers (deterministic); the |norm-1| sum is a float64
        // atomic (order-dependent in the last ulps).
        __shared__ unsigned long long s_cnt[4][3];
        __shared__ double s_dev[4];
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            a_pos += __shfl_xor(a_pos, off, kWave);
            a_neg += __shfl_xor(a_neg, off, kWave);
            a_nan += __shfl_xor(a_nan, off, kWave);
            a_dev += __shfl_xor(a_dev, off, kWave);
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            s_cnt[wave][0] = a_pos; s_cnt[wave][1] = a_neg; s_cnt[wave][2] = a_nan;
            s_dev[wave] = a_dev;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long p = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
            const unsigned long long n = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
            const unsigned long long q = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
            const double dv = (s_dev[0] + s_dev[1]) + (s_dev[2] + s_dev[3]);
            if (p) atomicAdd(&audit_counts[0], p);
            if (n) atomicAdd(&audit_counts[1], n);
            if (q) atomicAdd(&audit_counts[2], q);
            if (dv != 0.0) atomicAdd(audit_absdev, dv);
        }
    }
}


// K12: the post-filter of validation.extraction_efficiency (hwer/validation.py:133-141) on the device.
// One wave per user: walk its ranked list (score order), drop the items the user has seen in training (binary search in a
// sorted CSR row), keep the first `cut_max` survivors in order, mark which of them are validation truth, and set the
// coverage bit of every survivor ("diversity").  The float metrics (recall, NDCG) are finished on the host from these
// integers with the reference's own numpy expressions, so they stay bit-identical.
__device__ __forceinline__ bool csr_contains(const int64_t* __restrict__ v, int64_t lo, int64_t hi, int64_t x) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int64_t y = v[mid];
        if (y == x) return true;
        if (y < x) lo = mid + 1; else hi = mid;
    }
    return false;
}

__global__ __launch_bounds__(256) void rank_filter_seen_kernel(const int64_t* __restrict__ ranked, int64_t B, int k,
                                                              const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                                              const int64_t* __restrict__ truth_ptr, const int64_t* __restrict__ truth_idx,
                                                              int cut_max, int64_t* __restrict__ out_unseen, int32_t* __restrict__ out_len,
                                                              uint64_t* __restrict__ out_hits, int n_hit_words,
                                                              uint32_t* __restrict__ cover_bits, int64_t cover_base, int64_t cover_n) {
    const int lane = threadIdx.x & 63;
    const int64_t b = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t s0 = seen_ptr ? seen_ptr[b] : 0, s1 = seen_ptr ? seen_ptr[b + 1] : 0;
    const int64_t t0 = truth_ptr ? truth_ptr[b] : 0, t1 = truth_ptr ? truth_ptr[b + 1] : 0;
    int n_out = 0;
    uint64_t h0 = 0ull, h1 = 0ull;   // hit bits of survivor positions 0..63 and 64..127 (cut_max <= 128)
    for (int c0 = 0; c0 < k && n_out < cut_max; c0 += 64) {   // wave-uniform
        const int i = c0 + lane;
        const int64_t id = i < k ? ranked[b * k + i] : -1;
        const bool keep = id >= 0 && !csr_contains(seen_idx, s0, s1, id);
        const uint64_t bal = __ballot(keep);
        const int pos = n_out + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (keep && pos < cut_max) {
            out_unseen[b * cut_max + pos] = id;
            if (cover_bits && id >= cover_base && id - cover_base < cover_n)
                atomicOr(&cover_bits[(id - cover_base) >> 5], 1u << uint32_t((id - cover_base) & 31));
            if (csr_contains(truth_idx, t0, t1, id)) {
                if (pos < 64) h0 |= 1ull << uint32_t(pos);
                else h1 |= 1ull << uint32_t(pos - 64);
            }
        }
        n_out += __builtin_popcountll(bal);
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        h0 |= __shfl_xor(h0, off, kWave);
        h1 |= __shfl_xor(h1, off, kWave);
    }
    if (lane == 0) {
        out_hits[b * n_hit_words] = h0;
        if (n_hit_words > 1) out_hits[b * n_hit_words + 1] = h1;
    }
    n_out = min(n_out, cut_max);
    for (int i = n_out + lane; i < cut_max; i += 64) out_unseen[b * cut_max + i] = -1;
    if (lane == 0) out_len[b] = n_out;
}

// K8 and the fused tail.  FUSE = false: out = mean_adj(E);  FUSE = true: out = unit((E + mean_adj(E))/2)
template <bool FUSE>
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ E, int64_t lde,
                                                           const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices, int64_t N, int d4,
                                                           float* __restrict__ out, int64_t ldo) {
    const RowGroup g(d4);
    const int64_t group_global = (int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6)) * g.groups_per_wave + g.sub;
    const int64_t group_stride = int64_t(gridDim.x) * 4 * g.groups_per_wave;
    for (int64_t r0 = group_global - g.sub; r0 < N; r0 += group_stride) {
        const int64_t r = r0 + g.sub;
        const bool live = r < N;
        float4 acc[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float deg = 0.f;
        if (live) {
            const int64_t e0 = indptr[r], e1 = indptr[r + 1];
            deg = float(e1 - e0);
            for (int64_t e = e0; e < e1; ++e) {  // sequential adds in adjacency order (numpy mean(0))
                const float* base = E + int64_t(indices[e]) * lde;
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = g.c0 + i * g.L;
                    if (c < d4) {
                        const float4 v = *reinterpret_cast<const float4*>(base + 4 * c);
                        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                acc[i].x /= deg; acc[i].y /= deg; acc[i].z /= deg; acc[i].w /= deg;
            }
        }
        if constexpr (FUSE) {
            float ss = 0.f;
            if (live) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int c = g.c0 + i * g.L;
                    if (c < d4) {
                        const float4 v = *reinterpret_cast<const float4*>(E + r * lde + 4 * c);
                        acc[i].x = (v.x + acc[i].x) * 0.5f; acc[i].y = (v.y + acc[i].y) * 0.5f;
                        acc[i].z = (v.z + acc[i].z) * 0.5f; acc[i].w = (v.w + acc[i].w) * 0.5f;
                        ss += acc[i].x * acc[i].x + acc[i].y * acc[i].y + acc[i].z * acc[i].z + acc[i].w * acc[i].w;
                    }
                }
            }
            ss = group_sum(ss, g.L);
            const float nrm = sqrtf(ss);
            if (live) {
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    acc[i].x /= nrm; acc[i].y /= nrm; acc[i].z /= nrm; acc[i].w /= nrm;
                }
            }
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int c = g.c0 + i * g.L;
                if (c < d4) *reinterpret_cast<float4*>(out + r * ldo + 4 * c) = acc[i];
            }
        }
    }
}

// K11 pass 1: row norms;  pack_rows: gather rows of one node type into the padded table + half norms
typedef _Float16 half4_rt __attribute__((ext_vector_type(4)));
// NC / U as in blend_normalize_kernel: rows of up to 256 columns keep one chunk per lane and four rows in flight per row group
struct RowNormArgs {
    const float* V;
    int64_t ldv;
    int d, d4;
    const int64_t* row_ids;
    int64_t n_rows;
    float* X;
    int64_t ldx;
    int x4;
    float* out_norm;
    half4_rt* X16;
    int x_exp;
};
template <bool PACK, int NC, int U>
__device__ __forceinline__ void row_norm_body(const RowNormArgs& a, int64_t blk, int64_t nblk) {
    const float* __restrict__ V = a.V;
    const int64_t ldv = a.ldv, n_rows = a.n_rows, ldx = a.ldx;
    const int d = a.d, d4 = a.d4, x4 = a.x4, x_exp = a.x_exp;
    const int64_t* __restrict__ row_ids = a.row_ids;
    float* __restrict__ X = a.X;
    float* __restrict__ out_norm = a.out_norm;
    half4_rt* __restrict__ X16 = a.X16;
    const RowGroup g(PACK ? x4 : d4);
    const int64_t group_global = (blk * 4 + (threadIdx.x >> 6)) * g.groups_per_wave + g.sub;
    const int64_t group_stride = nblk * 4 * g.groups_per_wave;
    const int64_t n_out = PACK ? ((n_rows + ROW_PAD - 1) / ROW_PAD) * ROW_PAD : n_rows;   // PACK: pad rows (zeros, half norm +inf)
    for (int64_t rb = group_global - g.sub; rb < n_out; rb += group_stride * U) {
        float4 v[U][NC];
        int64_t rr[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // all loads first
            rr[u] = rb + int64_t(u) * group_stride + g.sub;
            live[u] = rr[u] < n_rows;
            const int64_t row = live[u] ? (row_ids ? row_ids[rr[u]] : rr[u]) : 0;
            const float* base = V + row * ldv;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = g.c0 + i * g.L;
                v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live[u] && c < d4) {
                    v[u][i] = *reinterpret_cast<const float4*>(base + 4 * c);
                    // columns >= d inside the last chunk are padding of the source table (must be zero there)
                    if (4 * c + 1 >= d) v[u][i].y = 0.f;
                    if (4 * c + 2 >= d) v[u][i].z = 0.f;
                    if (4 * c + 3 >= d) v[u][i].w = 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rb + int64_t(u) * group_stride >= n_out) break;   // wave-uniform
            const int64_t r = rr[u];
            float ss = 0.f;
            if (PACK && !live[u] && r < n_out) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int c = g.c0 + i * g.L;
                    if (c < x4) {
                        *reinterpret_cast<float4*>(X + r * ldx + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (X16) X16[(r * ldx) / 4 + c] = half4_rt{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    }
                }
                if (g.c0 == 0) out_norm[r] = INFINITY;
            }
            if (live[u]) {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int c = g.c0 + i * g.L;
                    const float4 w = v[u][i];
                    ss += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
                    if (PACK && c < x4) {
                        *reinterpret_cast<float4*>(X + r * ldx + 4 * c) = w;
                        if (X16)   // the float16 copy in the same pass (scaled by 2^x_exp: hwer_pack_rows_f16)
                            X16[(r * ldx) / 4 + c] = half4_rt{(_Float16)ldexpf(w.x, x_exp), (_Float16)ldexpf(w.y, x_exp), (_Float16)ldexpf(w.z, x_exp), (_Float16)ldexpf(w.w, x_exp)};
                    }
                }
            }
            ss = group_sum(ss, g.L);
            if (live[u] && g.c0 == 0) out_norm[r] = PACK ? 0.5f * ss : sqrtf(ss);
        }
    }
}
template <bool PACK, int NC, int U>
__global__ __launch_bounds__(256) void row_norm_kernel(const RowNormArgs a) {
    row_norm_body<PACK, NC, U>(a, blockIdx.x, gridDim.x);
}

// K11 pass 2: AUDIT_BLOCKS blocks write partials, the last block to finish (ticket) combines them in block order:
// deterministic, one launch.  `scratch` = AUDIT_BLOCKS * 4 doubles + 1 ticket word (zeroed by the previous call).
constexpr int AUDIT_BLOCKS = 64;
__global__ __launch_bounds__(256) void audit_reduce_kernel(const float* __restrict__ norms, int64_t N, float eps,
                                                           uint64_t* __restrict__ counts, double* __restrict__ absdev_sum,
                                                           double* __restrict__ partial, unsigned int* __restrict__ ticket) {
    __shared__ double s_v[4][256];
    __shared__ bool s_last;
    double pos = 0, neg = 0, nan = 0, dev = 0.0;
    // thresholds exactly as numpy evaluates `lengths > 1 + epsilon` with a float32 array and a python float epsilon:
    // the comparison is done in float64
    const double hi = 1.0 + double(eps), lo = 1.0 - double(eps);
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < N; i += int64_t(gridDim.x) * 256) {
        const float n = norms[i];
        pos += double(n) > hi;
        neg += double(n) < lo;
        nan += (n != n);
        dev += fabs(double(n) - 1.0);
    }
    s_v[0][threadIdx.x] = pos; s_v[1][threadIdx.x] = neg; s_v[2][threadIdx.x] = nan; s_v[3][threadIdx.x] = dev;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int c = 0; c < 4; ++c) s_v[c][threadIdx.x] += s_v[c][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        for (int c = 0; c < 4; ++c) __hip_atomic_store(&partial[blockIdx.x * 4 + c], s_v[c][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1);
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        double tot[4] = {0, 0, 0, 0};
        for (unsigned int bI = 0; bI < gridDim.x; ++bI)
            for (int c = 0; c < 4; ++c) tot[c] += __hip_atomic_load(&partial[bI * 4 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        counts[0] = uint64_t(tot[0]); counts[1] = uint64_t(tot[1]); counts[2] = uint64_t(tot[2]);
        absdev_sum[0] = tot[3];
    }
}

// =====================================================================================================
// K3+K4 query construction (hwer/recommendation_base.py:153-155,164-170), T = arithmetic type.
// 8 lanes per query; lane k owns columns j = k, k+8, ...  LDS per query: vec[d], sq[d], q[d] of T.
// =====================================================================================================
// (V64 != NULL: the rows come from the float64 table - the true float64 flavour of round 3, hwer_query_mix_f64rows; ldv is then ITS stride)
template <typename T>
__device__ __forceinline__ T load_elem(const float* V, int64_t ldv, int32_t idx, int j, const double* V64 = nullptr) {
    if (V64) return idx < 0 ? clip_unknown<T>(T(V64[j])) : T(V64[int64_t(idx) * ldv + j]);
    if (idx < 0) return clip_unknown<T>(T(V[j]));
    return T(V[int64_t(idx) * ldv + j]);
}

// vec <- unit(mean of the listed rows); returns nothing, result in `vec`
template <typename T>
__device__ __forceinline__ void unit_mean_rows(const float* V, int64_t ldv, int d, const int32_t* idx, int64_t n,
                                               int32_t single, T* vec, T* sq, int k, const double* V64 = nullptr) {
    for (int j = k; j < d; j += 8) {
        T acc;
        if (idx == nullptr) {
            acc = load_elem<T>(V, ldv, single, j, V64);  // mean of one row = row / 1
        } else {
            acc = load_elem<T>(V, ldv, idx[0], j, V64);
            for (int64_t m = 1; m < n; ++m) acc = acc + load_elem<T>(V, ldv, idx[m], j, V64);
            acc = acc / T(double(n));
        }
        vec[j] = acc;
        sq[j] = acc * acc;
    }
    wave_lds_sync();
    const T nrm = sqrt(np_pairwise<T, kPairwiseDepth>(sq, d, k));
    wave_lds_sync();
    for (int j = k; j < d; j += 8) vec[j] = vec[j] / nrm;
}

struct QueryMixArgs {
    const float* V;
    int64_t ldv;
    int d;
    const int32_t* anchor;
    const int64_t* pos_ptr;
    const int32_t* pos_idx;
    const int64_t* neg_ptr;
    const int32_t* neg_idx;
    int64_t B;
    double* Q64;
    float* Q32;
    int64_t ldq32;
    // optional: the float32 MFMA query fragments of the search that follows (knn_internal.h: repack_queries layout), rows [0, bq):
    //   Qf[((b / 32) * (ldq32 / 8) + j / 8) * 64 + (b % 32) + 32 * ((j % 8) / 4)][j % 4] = q_b[j], zero for b >= B and j >= d
    float* Qf;
    int64_t bq;
    const double* V64;   // nullable: read the rows from this float64 table (stride ldv) instead of V
    int gq;              // queries a wave builds per trip: 0 = 8 (one per group of 8 lanes); 4 / 2 / 1 for rows so wide that 8 x 3 d values
                         // do not fit the CU's LDS (float64 arithmetic above d = 853) - the other lane groups sit out
    int frag16;          // Qf holds the FLOAT16 fragments of the float16 screen instead (knn_small.hip: repack_queries16 layout): half8
                         // Qf16[((b / 32) * (ldq32 / 16) + j / 16) * 64 + (b % 32) + 32 * ((j % 16) / 8)][j % 8] = half(q_b[j] * 2^e_b), e_b = the power of
                         // two that puts the row's largest |element| in [2^13, 2^14), and behind them (byte offset bq * ldq32 * 2) the bq exponents
};

// One query row's fragments (float32 or float16 form), by the 8 lanes (k = 0 .. 7) that built the row: value(j) = float32 element j (j < d) of a LIVE
// row; dead / padding rows (b < bq) and the padding columns are zero-filled.  Same values as repack_queries / repack_queries16 produce from Q32.
template <class Value>
__device__ __forceinline__ void store_query_fragments(const QueryMixArgs& a, int64_t b, bool live, int k, Value value) {
    if (!a.Qf || b >= a.bq) return;
    const int ldq = int(a.ldq32), d = a.d;
    if (!a.frag16) {
        const int ngq = ldq / 8;
        for (int j = k; j < ldq; j += 8)
            a.Qf[(((b >> 5) * ngq + (j >> 3)) * 64 + (b & 31) + 32 * ((j & 7) >> 2)) * 4 + (j & 3)] = (live && j < d) ? value(j) : 0.f;
        return;
    }
    float m = 0.f;
    if (live)
        for (int j = k; j < d; j += 8) m = fmaxf(m, fabsf(value(j)));
    m = fmaxf(m, __shfl_xor(m, 1, kWave));   // the 8 lanes of the row's group
    m = fmaxf(m, __shfl_xor(m, 2, kWave));
    m = fmaxf(m, __shfl_xor(m, 4, kWave));
    int e = 0;
    if (m > 0.f && m < INFINITY) {
        int me;
        frexpf(m, &me);        // m = f * 2^me, f in [0.5, 1)  ->  m * 2^(14 - me) in [2^13, 2^14)
        e = 14 - me;
    }
    _Float16* Qh = reinterpret_cast<_Float16*>(a.Qf);
    const int ks = ldq / 16;
    for (int j = k; j < ldq; j += 8)
        Qh[(((b >> 5) * ks + (j >> 4)) * 64 + (b & 31) + 32 * ((j & 15) >> 3)) * 8 + (j & 7)] = (live && j < d) ? (_Float16)ldexpf(value(j), e) : (_Float16)0.f;
    if (k == 0) reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(a.Qf) + size_t(a.bq) * size_t(ldq) * 2)[b] = e;
}
template <typename T>
__device__ __forceinline__ void query_mix_body(const QueryMixArgs& a, unsigned char* smem_raw, int64_t blk, int64_t nblk, int wpb) {
    const float* __restrict__ V = a.V;
    const int64_t ldv = a.ldv, B = a.B, ldq32 = a.ldq32;
    const int d = a.d;
    const int32_t* __restrict__ anchor = a.anchor;
    const int64_t* __restrict__ pos_ptr = a.pos_ptr;
    const int32_t* __restrict__ pos_idx = a.pos_idx;
    const int64_t* __restrict__ neg_ptr = a.neg_ptr;
    const int32_t* __restrict__ neg_idx = a.neg_idx;
    double* __restrict__ Q64 = a.Q64;
    float* __restrict__ Q32 = a.Q32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= wpb) return;          // fused launches run 256-thread blocks whatever wpb is
    const int g8 = lane >> 3, k = lane & 7;
    const int gq = a.gq ? a.gq : 8;
    if (g8 >= gq) return;             // (no workgroup barrier below; the 8-lane groups never exchange data with each other)
    T* base = reinterpret_cast<T*>(smem_raw) + size_t(wave * gq + g8) * 3 * d;
    T* vec = base;
    T* sq = base + d;
    T* q = base + 2 * d;
    const int64_t q0 = (blk * wpb + wave) * gq;
    const int64_t qstride = nblk * wpb * gq;
    const int64_t n_rows_all = a.Qf ? (a.bq > B ? a.bq : B) : B;
    for (int64_t qb = q0; qb < n_rows_all; qb += qstride) {  // qb wave-uniform
        if (qb >= B) {   // wave-uniform: only fragment padding rows are left in this group of 8
            store_query_fragments(a, qb + g8, false, k, [](int) { return 0.f; });
            continue;
        }
        const int64_t b = qb + g8;
        const bool live = b < B;
        // every lane of the wave runs the same control flow; dead groups compute on anchor 0 and skip the store
        const int64_t bb = live ? b : 0;
        int cnt = 1;
        unit_mean_rows<T>(V, ldv, d, nullptr, 1, anchor[bb], vec, sq, k, a.V64);
        for (int j = k; j < d; j += 8) q[j] = vec[j];
        const int64_t np_ = pos_ptr ? pos_ptr[bb + 1] - pos_ptr[bb] : 0;
        const int64_t nn_ = neg_ptr ? neg_ptr[bb + 1] - neg_ptr[bb] : 0;
        // the two branches below are taken per 8-lane group; wave_lds_sync inside is a wave-level fence only
        // (no s_barrier), so divergence between groups is legal.
        if (np_ > 0) {
            unit_mean_rows<T>(V, ldv, d, pos_idx + pos_ptr[bb], np_, 0, vec, sq, k, a.V64);
            for (int j = k; j < d; j += 8) q[j] = q[j] + vec[j];
            ++cnt;
        }
        if (nn_ > 0) {
            unit_mean_rows<T>(V, ldv, d, neg_idx + neg_ptr[bb], nn_, 0, vec, sq, k, a.V64);
            for (int j = k; j < d; j += 8) q[j] = q[j] + (T(-1) * vec[j]);
            ++cnt;
        }
        if (live) {
            for (int j = k; j < d; j += 8) {
                const T v = q[j] / T(double(cnt));
                Q64[b * d + j] = double(v);
                Q32[b * ldq32 + j] = float(v);
            }
            for (int j = d + k; j < ldq32; j += 8) Q32[b * ldq32 + j] = 0.f;
        }
        store_query_fragments(a, b, live, k, [&](int j) { return float(q[j] / T(double(cnt))); });
        wave_lds_sync();
    }
}
// Anchor-only queries in bulk (round 5): q_b = unit(V[anchor_b]) with numpy's norm, for callers whose anchors outnumber what the 8-lanes-per-query
// kernel above does well (it reads a row as 32-byte pieces and spends its time in LDS bank conflicts: 145 us for the 138 493 queries of the
// ML-20M shape, 1.9 TB/s).  Same arithmetic, element for element - widen (clip an unknown anchor's row 0), squares summed in numpy's pairwise
// order, one IEEE division per element - on the layout of the float64 blend kernels (rows_f64.hip): L lanes load a row as 16-byte chunks
// (coalesced), the widened row waits in an LDS slab, 8 lanes replay the summation order, the L lanes divide and store Q64 / Q32.
template <typename T>
__device__ __forceinline__ T np_pairwise_sq_leaf_t(const T* p, int n, int k) {
    if (n < 8) {
        T r = T(0);
        for (int i = 0; i < n; ++i) r = r + p[i] * p[i];
        return r;
    }
    const int lim = n - (n & 7);
    T r = p[k] * p[k];
    for (int i = 8; i < lim; i += 8) r = r + p[i + k] * p[i + k];
    r = r + shfl_xor_t(r, 1);
    r = r + shfl_xor_t(r, 2);
    r = r + shfl_xor_t(r, 4);
    for (int i = lim; i < n; ++i) r = r + p[i] * p[i];
    return r;
}
template <typename T, int DEPTH>
__device__ __forceinline__ T np_pairwise_sq_t(const T* p, int n, int k) {
    if constexpr (DEPTH == 0) {
        return np_pairwise_sq_leaf_t(p, n, k);
    } else {
        if (n <= 128) return np_pairwise_sq_leaf_t(p, n, k);
        int n2 = n / 2;
        n2 -= n2 & 7;
        const T a = np_pairwise_sq_t<T, DEPTH - 1>(p, n2, k);
        const T b = np_pairwise_sq_t<T, DEPTH - 1>(p + n2, n - n2, k);
        return a + b;
    }
}
template <typename T, typename TV = float>      // TV = double (round 6): the rows of a float64 table (hwer_query_mix_f64rows), two 16-byte loads per 4-column chunk
__global__ __launch_bounds__(256) void anchor_query_kernel(const TV* __restrict__ V, int64_t ldv, int d, int d4, const int32_t* __restrict__ anchor, int64_t B,
                                                           double* __restrict__ Q64, float* __restrict__ Q32, int64_t ldq32, int stride, int pb_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* slab = reinterpret_cast<T*>(smem_raw) + size_t(wave) * (size_t(pb_n) * stride + 8);
    T* s_nrm = slab + size_t(pb_n) * stride;
    const int q4 = int(ldq32 / 4);                 // float4 chunks of a Q32 row (>= d4: the padding columns are zero)
    const int L = min(64, pow2ceil(q4));
    const int rpi = 64 / L, sub = lane / L, c0 = lane % L;
    const int64_t n_trips = cdiv64(B, pb_n);
    for (int64_t t = int64_t(blockIdx.x) * 4 + wave; t < n_trips; t += int64_t(gridDim.x) * 4) {
        const int64_t p0 = t * pb_n;
        for (int pb = 0; pb < pb_n; pb += rpi) {
            const int slot = pb + sub;
            const int64_t b = p0 + slot;
            if (slot >= pb_n || b >= B) continue;
            const int32_t an = anchor[b];
            const bool unknown = an < 0;
            for (int c = c0; c < d4; c += L) {
                T* w = slab + slot * stride + 4 * c;
                if constexpr (std::is_same<TV, double>::value) {
                    const double* p = V + int64_t(unknown ? 0 : an) * ldv + 4 * c;
                    const double2 lo = *reinterpret_cast<const double2*>(p), hi = *reinterpret_cast<const double2*>(p + 2);
                    w[0] = unknown ? clip_unknown<T>(T(lo.x)) : T(lo.x); w[1] = unknown ? clip_unknown<T>(T(lo.y)) : T(lo.y);
                    w[2] = unknown ? clip_unknown<T>(T(hi.x)) : T(hi.x); w[3] = unknown ? clip_unknown<T>(T(hi.y)) : T(hi.y);
                } else {
                const float4 v = load_row4(V, ldv, an, c);
                w[0] = widen<T>(v.x, unknown); w[1] = widen<T>(v.y, unknown); w[2] = widen<T>(v.z, unknown); w[3] = widen<T>(v.w, unknown);
                }
            }
        }
        wave_lds_sync();
        {
            const int p8 = lane >> 3, k = lane & 7;
            const T ss = np_pairwise_sq_t<T, kPairwiseDepth>(slab + min(p8, pb_n - 1) * stride, d, k);
            if (k == 0 && p8 < pb_n) s_nrm[p8] = sqrt(ss);
        }
        wave_lds_sync();
        for (int pb = 0; pb < pb_n; pb += rpi) {
            const int slot = pb + sub;
            const int64_t b = p0 + slot;
            if (slot >= pb_n || b >= B) continue;
            const T nrm = s_nrm[slot];
            for (int c = c0; c < q4; c += L) {
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = 4 * c + i;
                    if (j < d) {
                        const T v = (slab[slot * stride + j] / nrm) / T(1.0);   // (unit row) / (number of terms of the query mean = 1)
                        Q64[b * d + j] = double(v);
                        o[i] = float(v);
                    }
                }
                *reinterpret_cast<float4*>(Q32 + b * ldq32 + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        wave_lds_sync();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void query_mix_kernel(const QueryMixArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    query_mix_body<T>(a, smem_raw, blockIdx.x, gridDim.x, int(blockDim.x >> 6));
}

// Fused search prologue: the first `n_pack_blocks` workgroups build the per-type table (row_norm_body<PACK>), the others the
// queries (query_mix_body) - two latency-floor kernels of a small search in ONE launch, plus the search's query fragments.
template <typename T, int NC, int U>
__global__ __launch_bounds__(256) void prepare_search_kernel(const RowNormArgs pa, const QueryMixArgs qa, int n_pack_blocks, int wpb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (int(blockIdx.x) < n_pack_blocks) row_norm_body<true, NC, U>(pa, blockIdx.x, n_pack_blocks);
    else query_mix_body<T>(qa, smem_raw, int64_t(blockIdx.x) - n_pack_blocks, int64_t(gridDim.x) - n_pack_blocks, wpb);
}


// =====================================================================================================
// Fused step prologue (hwer_blend_prepare_search_f32): blend + unit-norm audit + per-type table + anchor queries in ONE launch.
// Three block roles; none waits for another:
//   PACK   : rows of the searched type - blended, written to V AND to the padded table X (+ half norms, + float16 copy);
//   OTHER  : the remaining rows - blended, written to V;
//   QUERY  : 8 anchors per wave - the anchor's row is blended AGAIN by the same code (bit-identical to the row a PACK / OTHER
//            block stores in V) into LDS, then the reference's query arithmetic runs on it (anchor-only queries).
// Row arithmetic = blend_normalize_kernel<1, U> / row_norm_body<true, 1, U> / query_mix_body: the outputs are bit-identical to
// the three separate launches (tests/test_gpu_round2.py).
// =====================================================================================================
struct BlendArgs {
    const float* const* tables;
    int64_t ldt;
    const int32_t* const* row_idx;
    const float* alpha;
    int alpha_per_row, T, d, d4;
    float* V;
    int64_t ldv;
    float audit_eps;
    WsHeader* hdr;
    unsigned long long* audit_counts;
    double* audit_absdev;
    unsigned int n_audit_blocks;
};
struct AuditAcc {
    unsigned long long pos = 0, neg = 0, nan = 0;
    double dev = 0.0;
};

// o[u] <- this lane's chunk (c0) of unit(sum_t alpha_t T_t[row]) for the U rows rr[u] of the lane's row group
template <int U>
__device__ __forceinline__ void blend_unit_rows(const BlendArgs& a, const RowGroup& g, const int64_t (&rr)[U], const bool (&live)[U], float4 (&o)[U]) {
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < a.T; ++t) {
        const int32_t* ri = a.row_idx ? a.row_idx[t] : nullptr;
        const float* tab = a.tables[t];
        float4 v[U];
        float al[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // all loads of this table first
            al[u] = 0.f;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[u]) {
                al[u] = a.alpha_per_row ? a.alpha[rr[u] * a.T + t] : a.alpha[t];
                const int64_t row = ri ? int64_t(ri[rr[u]]) : rr[u];
                if (g.c0 < a.d4) v[u] = *reinterpret_cast<const float4*>(tab + row * a.ldt + 4 * g.c0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc[u].x = fmaf(al[u], v[u].x, acc[u].x);
            acc[u].y = fmaf(al[u], v[u].y, acc[u].y);
            acc[u].z = fmaf(al[u], v[u].z, acc[u].z);
            acc[u].w = fmaf(al[u], v[u].w, acc[u].w);
        }
    }
    float ssv[U], inv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float ss = acc[u].x * acc[u].x + acc[u].y * acc[u].y + acc[u].z * acc[u].z + acc[u].w * acc[u].w;
        ssv[u] = group_sum(ss, g.L);  // lanes past d4 hold exact zeros: the sum is that of the pow2ceil(d4)-lane group
    }
    // An IEEE square root + division is ~45 vector instructions whatever the lanes hold, and this kernel is not far from
    // VALU-bound (SQ counters: ~830 wave-instructions per trip of 8 rows).  Every lane of a row group holds all four sums: lane
    // (lane & 3) == u takes row u's, ONE sqrt / division sequence serves the four rows, quad broadcasts hand the results back
    // (round 3).  Same operations on the same values: bit-identical rows.
    if (U == 4 && g.L >= 4) {
        const int l3 = int(threadIdx.x) & 3;
        const float sp = l3 == 0 ? ssv[0] : (l3 == 1 ? ssv[1] : (l3 == 2 ? ssv[2] : ssv[U - 1]));
        const float ip = 1.0f / sqrtf(sp);
        const int ipb = __float_as_int(ip);
        inv[0] = __int_as_float(__builtin_amdgcn_update_dpp(ipb, ipb, 0x00, 0xF, 0xF, true));   // quad_perm [0,0,0,0]
        if (U > 1) inv[1 % U] = __int_as_float(__builtin_amdgcn_update_dpp(ipb, ipb, 0x55, 0xF, 0xF, true));   // [1,1,1,1]
        if (U > 2) inv[2 % U] = __int_as_float(__builtin_amdgcn_update_dpp(ipb, ipb, 0xAA, 0xF, 0xF, true));   // [2,2,2,2]
        if (U > 3) inv[3 % U] = __int_as_float(__builtin_amdgcn_update_dpp(ipb, ipb, 0xFF, 0xF, 0xF, true));   // [3,3,3,3]
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) inv[u] = 1.0f / sqrtf(ssv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) o[u] = make_float4(acc[u].x * inv[u], acc[u].y * inv[u], acc[u].z * inv[u], acc[u].w * inv[u]);
}
// store the row in V; returns the squared norm of the row AS STORED (every lane of the row group)
__device__ __forceinline__ float store_row(const BlendArgs& a, const RowGroup& g, int64_t r, bool live, const float4& o) {
    float s2 = 0.f;
    if (live && g.c0 < a.d4) {
        *reinterpret_cast<float4*>(a.V + r * a.ldv + 4 * g.c0) = o;
        s2 = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    }
    return group_sum(s2, g.L);
}
__device__ __forceinline__ void audit_norm(const BlendArgs& a, float s2, bool count, AuditAcc& au) {
    const float n = sqrtf(s2);
    if (count) {
        au.pos += double(n) > 1.0 + double(a.audit_eps);
        au.neg += double(n) < 1.0 - double(a.audit_eps);
        au.nan += (n != n);
        au.dev += fabs(double(n) - 1.0);
    }
}
// store the row in V and audit its norm AS STORED
__device__ __forceinline__ void store_audit_row(const BlendArgs& a, const RowGroup& g, int64_t r, bool live, const float4& o, AuditAcc& au) {
    const float s2 = store_row(a, g, r, live, o);
    audit_norm(a, s2, live && g.c0 == 0, au);
}
// block partials -> header accumulators; the last PACK / OTHER block moves the totals out and re-zeroes the accumulators
__device__ __forceinline__ void audit_finish(const BlendArgs& a, AuditAcc au) {
    __shared__ unsigned long long s_cnt[4][3];
    __shared__ double s_dev[4];
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        au.pos += __shfl_xor(au.pos, off, kWave);
        au.neg += __shfl_xor(au.neg, off, kWave);
        au.nan += __shfl_xor(au.nan, off, kWave);
        au.dev += __shfl_xor(au.dev, off, kWave);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[wave][0] = au.pos; s_cnt[wave][1] = au.neg; s_cnt[wave][2] = au.nan;
        s_dev[wave] = au.dev;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        WsHeader* h = a.hdr;
        const unsigned long long p = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
        const unsigned long long n = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
        const unsigned long long q = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
        const double dv = (s_dev[0] + s_dev[1]) + (s_dev[2] + s_dev[3]);
        if (p) atomicAdd(&h->audit_acc[0], p);
        if (n) atomicAdd(&h->audit_acc[1], n);
        if (q) atomicAdd(&h->audit_acc[2], q);
        if (dv != 0.0) atomicAdd(&h->audit_dev, dv);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(&h->audit_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == a.n_audit_blocks - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int c = 0; c < 3; ++c) {
                a.audit_counts[c] = __hip_atomic_load(&h->audit_acc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&h->audit_acc[c], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a.audit_counts[3] = 0ull;
            a.audit_absdev[0] = __hip_atomic_load(&h->audit_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&h->audit_dev, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&h->audit_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct RowRange {          // rows of V: ids[i] (or row0 + i), i < n
    const int64_t* ids;
    int64_t row0, n;
};

template <bool PACK, int U>
__device__ __forceinline__ void blend_rows_body(const BlendArgs& a, const RowRange& rg, const RowNormArgs& pk, int64_t blk, int64_t nblk, AuditAcc& au) {
    const RowGroup g(PACK ? pk.x4 : a.d4);
    const int64_t group_global = (blk * 4 + (threadIdx.x >> 6)) * g.groups_per_wave + g.sub;
    const int64_t group_stride = nblk * 4 * g.groups_per_wave;
    const int64_t n_out = PACK ? ((rg.n + ROW_PAD - 1) / ROW_PAD) * ROW_PAD : rg.n;   // PACK: pad rows (zeros, half norm +inf)
    // The searched type's table IS a slice of V when its rows are contiguous, as wide as the padded table and the caller passed
    // X = V + row0 * ldv (V then has the <= 31 pad rows behind its last row): the row just stored in V is the table row - no second
    // 4 d-byte store per row (round 3: 22 % of the bytes this launch moves at d = 128, T = 2)
    const bool x_is_v = PACK && rg.ids == nullptr && pk.ldx == a.ldv && a.d == int(pk.ldx) && pk.X == a.V + rg.row0 * a.ldv;
    for (int64_t rb = group_global - g.sub; rb < n_out; rb += group_stride * U) {
        int64_t pos[U], rr[U];
        bool live[U];
        float4 o[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pos[u] = rb + int64_t(u) * group_stride + g.sub;
            live[u] = pos[u] < rg.n;
            rr[u] = live[u] ? (rg.ids ? rg.ids[pos[u]] : rg.row0 + pos[u]) : 0;
        }
        blend_unit_rows<U>(a, g, rr, live, o);
        if (U == 4 && g.L >= 4) {   // the four rows' audits through ONE square root: lane (c0 & 3) == u of a group audits row u
            float s2v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) s2v[u] = (rb + int64_t(u) * group_stride < n_out) ? store_row(a, g, rr[u], live[u], o[u]) : 0.f;   // (wave-uniform guard)
            const int l3 = g.c0 & 3;
            const float sp = l3 == 0 ? s2v[0] : (l3 == 1 ? s2v[1] : (l3 == 2 ? s2v[2] : s2v[U - 1]));
            const bool lv = l3 == 0 ? live[0] : (l3 == 1 ? live[1] : (l3 == 2 ? live[2] : live[U - 1]));
            audit_norm(a, sp, lv && g.c0 < 4, au);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (rb + int64_t(u) * group_stride >= n_out) break;   // wave-uniform
            if (!(U == 4 && g.L >= 4)) store_audit_row(a, g, rr[u], live[u], o[u], au);
            if constexpr (PACK) {
                const int64_t r = pos[u];
                const int c = g.c0;
                float ss = 0.f;
                if (!live[u] && r < n_out) {
                    if (c < pk.x4) {
                        *reinterpret_cast<float4*>(pk.X + r * pk.ldx + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (pk.X16) pk.X16[(r * pk.ldx) / 4 + c] = half4_rt{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    }
                    if (c == 0) pk.out_norm[r] = INFINITY;
                }
                if (live[u]) {
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < a.d4) {
                        w = o[u];   // = the row as stored in V; columns >= d of the last chunk are table padding
                        if (4 * c + 1 >= a.d) w.y = 0.f;
                        if (4 * c + 2 >= a.d) w.z = 0.f;
                        if (4 * c + 3 >= a.d) w.w = 0.f;
                    }
                    ss = w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
                    if (c < pk.x4) {
                        if (!x_is_v) *reinterpret_cast<float4*>(pk.X + r * pk.ldx + 4 * c) = w;
                        if (pk.X16)
                            pk.X16[(r * pk.ldx) / 4 + c] = half4_rt{(_Float16)ldexpf(w.x, pk.x_exp), (_Float16)ldexpf(w.y, pk.x_exp), (_Float16)ldexpf(w.z, pk.x_exp), (_Float16)ldexpf(w.w, pk.x_exp)};
                    }
                }
                ss = group_sum(ss, g.L);
                if (live[u] && c == 0) pk.out_norm[r] = 0.5f * ss;
            }
        }
    }
}

// anchor-only queries from rows blended in place (LDS): query_mix_body with np_ = nn_ = 0
template <typename T>
// G = queries per wave and trip (8, or fewer for wide rows: the launch has ONE dynamic-LDS size, and the row-streaming PACK / OTHER
// blocks need many workgroups per CU - 40 KB of query staging per block made the d = 256 launch 1.8x slower than three launches)
__device__ __forceinline__ void blend_query_body(const BlendArgs& a, const QueryMixArgs& qa, unsigned char* smem_raw, int64_t blk, int64_t nblk, int wpb, int G) {
    const int64_t B = qa.B, ldq32 = qa.ldq32;
    const int d = qa.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= wpb) return;
    const int g8 = lane >> 3, k = lane & 7;
    const bool grp = g8 < G;                 // lane groups beyond G idle (their pointers stay inside the wave's area)
    const int g8c = grp ? g8 : G - 1;
    // per wave: G x (vec, sq) of T, then G rows of 4 d4 floats
    const size_t per_wave = size_t(G) * 2 * d * sizeof(T) + size_t(G) * 4 * a.d4 * sizeof(float);
    unsigned char* wbase = smem_raw + size_t(wave) * per_wave;
    T* vec = reinterpret_cast<T*>(wbase) + size_t(g8c) * 2 * d;
    T* sq = vec + d;
    float* rows = reinterpret_cast<float*>(wbase + size_t(G) * 2 * d * sizeof(T));
    const RowGroup g(a.d4);
    const int64_t q0 = (blk * wpb + wave) * G;
    const int64_t qstride = nblk * wpb * G;
    const int64_t n_rows_all = qa.Qf ? (qa.bq > B ? qa.bq : B) : B;
    for (int64_t qb = q0; qb < n_rows_all; qb += qstride) {  // qb wave-uniform
        if (qb >= B) {   // only fragment padding rows are left in this group
            if (grp) store_query_fragments(qa, qb + g8, false, k, [](int) { return 0.f; });
            continue;
        }
        // the G anchor rows of this trip, blended exactly as the PACK / OTHER blocks blend them, into LDS
        for (int s0 = 0; s0 < G; s0 += g.groups_per_wave) {   // (L <= 64: at least one row per pass)
            const int s = s0 + g.sub;
            int64_t rr[1];
            bool live[1];
            float4 o[1];
            const int64_t bq_ = qb + s;
            const int32_t an = (s < G && bq_ < B) ? qa.anchor[bq_] : 0;
            rr[0] = an < 0 ? 0 : an;                 // unknown node: row 0 (clipped below)
            live[0] = s < G;
            blend_unit_rows<1>(a, g, rr, live, o);
            if (s < G && g.c0 < a.d4) *reinterpret_cast<float4*>(rows + size_t(s) * 4 * a.d4 + 4 * g.c0) = o[0];
        }
        wave_lds_sync();
        const int64_t b = qb + g8;
        const bool live = grp && b < B;
        const bool unknown = live && qa.anchor[b] < 0;
        const float* row = rows + size_t(g8c) * 4 * a.d4;
        // unit_mean_rows(idx == nullptr): mean of one row = the row, then unit length in T arithmetic
        if (grp)
        for (int j = k; j < d; j += 8) {
            const T acc = unknown ? clip_unknown<T>(T(row[j])) : T(row[j]);
            vec[j] = acc;
            sq[j] = acc * acc;
        }
        wave_lds_sync();
        const T nrm = sqrt(np_pairwise<T, kPairwiseDepth>(sq, d, k));
        wave_lds_sync();
        if (live) {
            for (int j = k; j < d; j += 8) {
                const T v = (vec[j] / nrm) / T(1.0);
                qa.Q64[b * d + j] = double(v);
                qa.Q32[b * ldq32 + j] = float(v);
            }
            for (int j = d + k; j < ldq32; j += 8) qa.Q32[b * ldq32 + j] = 0.f;
        }
        if (grp) store_query_fragments(qa, b, live, k, [&](int j) { return float((vec[j] / nrm) / T(1.0)); });
        wave_lds_sync();
    }
}

// (experiment switch: waves per SIMD the compiler must make room for.  5 / 6 instead of the 4 the kernel's 121 VGPRs allow: the QUERY role's float64
// arithmetic sets that number, and capping it spills in the row-streaming roles too - c4 0.47 -> 0.55 / 0.82 ms, profiles/r4_ab_pingpong.txt)
#ifndef HWER_PROLOGUE_MINWAVES
#define HWER_PROLOGUE_MINWAVES 1
#endif
template <typename T, int U>
__global__ __launch_bounds__(256, HWER_PROLOGUE_MINWAVES) void blend_prepare_search_kernel(const BlendArgs ba, const RowRange pack, const RowRange other, const RowNormArgs pk,
                                                                   const QueryMixArgs qa, int n_pack_blocks, int n_other_blocks, int wpb, int gq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int blk = int(blockIdx.x);
    if (blk == 0 && threadIdx.x == 0 && ba.hdr->init_magic != WS_INIT_MAGIC) {
        // the workspace never went through hwer_knn_workspace_init: its audit accumulators are garbage and no block may ever
        // see the last ticket - make the audit FAIL instead of leaving stale "all rows fine" counts in the caller's buffer
        ba.audit_counts[0] = ba.audit_counts[1] = ba.audit_counts[2] = ~0ull;
        ba.audit_counts[3] = ~0ull;
    }
#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
    const int ablate = wpb >> 8;   // diagnostic builds: HWER_PROLOGUE_ABLATE (1 no QUERY blocks, 2 no PACK, 4 no OTHER, 8 no audit atomics)
    wpb &= 0xff;
    if (blk >= n_pack_blocks + n_other_blocks ? (ablate & 1) : (blk < n_pack_blocks ? (ablate & 2) : (ablate & 4))) return;
#endif
    if (blk < n_pack_blocks + n_other_blocks) {
        AuditAcc au;
        if (blk < n_pack_blocks) blend_rows_body<true, U>(ba, pack, pk, blk, n_pack_blocks, au);
        else blend_rows_body<false, U>(ba, other, pk, blk - n_pack_blocks, n_other_blocks, au);
#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
        if (ablate & 8) return;
#endif
        audit_finish(ba, au);
    } else {
        blend_query_body<T>(ba, qa, smem_raw, int64_t(blk) - n_pack_blocks - n_other_blocks, int64_t(gridDim.x) - n_pack_blocks - n_other_blocks, wpb, gq);
    }
}

}  // namespace hwer

using namespace hwer;

// =====================================================================================================
// C ABI
// =====================================================================================================
// =====================================================================================================
// K12b  Recall@K / binary NDCG@K of the retrieval evaluation as floats, on the device (round 4) - the per-user values of
//       utils.recall (hwer/utils.py:118-121) and utils.binary_ndcg (:101-111) from the hit masks of hwer_rank_filter_seen, and their means
//       over the validation users (hwer/validation.py:167-172), in numpy's arithmetic: a per-user DCG is np.sum over a row of <= 128 terms
//       (8 strided chains, pairwise combine, sequential tail), a mean is np.add.reduce over the users (8192-element pieces in order, each by the pairwise recursion with leaves of <= 128) / V.
// =====================================================================================================
__device__ __forceinline__ double np_sum_leaf_seq(const double* p, int64_t n) {   // numpy's <= 128-element leaf, by one thread
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r = __dadd_rn(r, p[i]);
        return r;
    }
    double r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3], r4 = p[4], r5 = p[5], r6 = p[6], r7 = p[7];
    const int64_t lim = n - (n & 7);
    for (int64_t i = 8; i < lim; i += 8) {
        r0 = __dadd_rn(r0, p[i]); r1 = __dadd_rn(r1, p[i + 1]); r2 = __dadd_rn(r2, p[i + 2]); r3 = __dadd_rn(r3, p[i + 3]);
        r4 = __dadd_rn(r4, p[i + 4]); r5 = __dadd_rn(r5, p[i + 5]); r6 = __dadd_rn(r6, p[i + 6]); r7 = __dadd_rn(r7, p[i + 7]);
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r0, r1), __dadd_rn(r2, r3)), __dadd_rn(__dadd_rn(r4, r5), __dadd_rn(r6, r7)));
    for (int64_t i = lim; i < n; ++i) res = __dadd_rn(res, p[i]);
    return res;
}
__device__ double np_pairwise_sum_seq(const double* p, int64_t n) {   // numpy's recursion (loops_utils.h.src DOUBLE_pairwise_sum)
    if (n <= 128) return np_sum_leaf_seq(p, n);
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const double a = np_pairwise_sum_seq(p, n2);
    const double b = np_pairwise_sum_seq(p + n2, n - n2);
    return __dadd_rn(a, b);
}

struct RankMetricsArgs {
    const uint64_t* hits;     // (B, hit_words)
    int hit_words;
    const int32_t* n_pred;    // (B) survivors of every anchor (hwer_rank_filter_seen: out_len)
    const int64_t* vrow;      // (V) row of validation user v in the two arrays above
    const int64_t* n_true;    // (V) validation items of user v that are not training items
    int64_t V;
    int n_cuts;
    int cuts[8];
    const double* log2_tab;   // (cut_max + 1, ld_log2): row L = np.log2(np.arange(L) + 2) - numpy's values for a list of L predictions
    int ld_log2;
    const double* idcg_tab;   // (cut_max + 1) ideal DCG of m hits
    double* vals;             // (2 n_cuts, V): recall then ndcg of every cut
    double* out;              // (2 n_cuts) means
};

// one thread per (validation user, cut)
__global__ __launch_bounds__(256) void rank_metrics_rows_kernel(const RankMetricsArgs a) {
    const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int ci = blockIdx.y;
    if (v >= a.V) return;
    const int c = a.cuts[ci];
    const int64_t row = a.vrow[v];
    const int lp = min(a.n_pred[row], c);
    const int64_t nt = a.n_true[v];
    const uint64_t* hw = a.hits + row * a.hit_words;
    int h = 0;
    double terms[128];      // (gain / discount of position p: 1 / log2(p + 2) for a hit, 0.0 / log2(p + 2) = 0.0 otherwise)
    for (int pp = 0; pp < lp; ++pp) {
        const bool hit = (hw[pp >> 6] >> (pp & 63)) & 1ull;
        h += hit ? 1 : 0;
        terms[pp] = hit ? __ddiv_rn(1.0, a.log2_tab[int64_t(lp) * a.ld_log2 + pp]) : 0.0;
    }
    const double denom = fmax(double(min(int64_t(lp), nt)), 1.0);
    a.vals[int64_t(ci) * a.V + v] = __ddiv_rn(double(h), denom);
    const double dcg = lp ? np_sum_leaf_seq(terms, lp) : 0.0;
    const int64_t ideal = min(nt, int64_t(lp));
    a.vals[int64_t(a.n_cuts + ci) * a.V + v] = __ddiv_rn(dcg, __dadd_rn(a.idcg_tab[ideal], 1e-8));
}
// one thread per metric: np.mean over the validation users
__global__ void rank_metrics_mean_kernel(const RankMetricsArgs a) {
    const int m = blockIdx.x;
    if (threadIdx.x != 0) return;
    // np.add.reduce over a 1-D array runs its inner loop on the ufunc buffer's 8192-element pieces (numpy 2.2: the pairwise recursion is per
    // piece, the pieces are added to the running sum in order) - measured against np.sum / np.mean for V up to 100003 (tests/test_host_api.py)
    const double* p = a.vals + int64_t(m) * a.V;
    double r = 0.0;
    for (int64_t i = 0; i < a.V; i += 8192) r = __dadd_rn(r, np_pairwise_sum_seq(p + i, min(int64_t(8192), a.V - i)));
    a.out[m] = __ddiv_rn(r, double(a.V));
}

extern "C" {

int hwer_abi_version(void) { return HWER_ABI_VERSION; }
const char* hwer_last_error(void) { return g_err; }
int hwer_stream_synchronize(void* stream) {
    hipError_t e = hipStreamSynchronize(as_stream(stream));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "stream_synchronize: %s", hipGetErrorString(e));
    return HWER_OK;
}

int hwer_pair_score_f32(const float* V, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst,
                        int64_t P, float* out, void* stream) {
    return launch_pair_score<float>(V, N, d, ldv, src, dst, P, out, stream);
}
int hwer_pair_score_f64(const float* V, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst,
                        int64_t P, double* out, void* stream) {
    return launch_pair_score<double>(V, N, d, ldv, src, dst, P, out, stream);
}

// compute units of the current device (cached per device)
static int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

static int rows_grid(int64_t rows, int d4) {
    int L = 1;
    while (L < d4 && L < 64) L <<= 1;
    const int64_t rows_per_block = 4 * (64 / L);
    int64_t cap = kMaxStreamBlocks;
    if (const char* e = diag_env("HWER_ROW_BLOCKS")) cap = std::max(1, atoi(e));   // tuning override (diagnostic builds)
    return int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(rows, rows_per_block), cap)));
}

static int check_table(const char* fn, const void* p, int64_t ld, int d) {
    if (!p) return fail(HWER_EINVAL, "%s: null pointer", fn);
    if (d < 1 || d > 4 * 64 * MAXC) return fail(HWER_EINVAL, "%s: need 1 <= d <= %d (d=%d)", fn, 4 * 64 * MAXC, d);
    if (ld % 4 != 0 || ld < d) return fail(HWER_EINVAL, "%s: row stride must be a multiple of 4 and >= d", fn);
    if ((reinterpret_cast<uintptr_t>(p) & 15) != 0) return fail(HWER_EINVAL, "%s: table must be 16-byte aligned", fn);
    return HWER_OK;
}

int hwer_segment_mean_f32(const float* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N,
                          int d, float* out, int64_t ldo, void* stream) {
    if (N == 0) return HWER_OK;
    int rc;
    if ((rc = check_table("segment_mean(E)", E, lde, d))) return rc;
    if ((rc = check_table("segment_mean(out)", out, ldo, d))) return rc;
    if (!indptr || !indices) return fail(HWER_EINVAL, "segment_mean: null CSR");
    const int d4 = (d + 3) / 4;
    hipLaunchKernelGGL(segment_mean_kernel<false>, dim3(rows_grid(N, d4)), dim3(256), 0, as_stream(stream), E, lde,
                       indptr, indices, N, d4, out, ldo);
    return check_launch("segment_mean");
}

int hwer_content_tail_f32(const float* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N,
                          int d, float* out, int64_t ldo, void* stream) {
    if (N == 0) return HWER_OK;
    int rc;
    if ((rc = check_table("content_tail(E)", E, lde, d))) return rc;
    if ((rc = check_table("content_tail(out)", out, ldo, d))) return rc;
    if (!indptr || !indices) return fail(HWER_EINVAL, "content_tail: null CSR");
    const int d4 = (d + 3) / 4;
    hipLaunchKernelGGL(segment_mean_kernel<true>, dim3(rows_grid(N, d4)), dim3(256), 0, as_stream(stream), E, lde,
                       indptr, indices, N, d4, out, ldo);
    return check_launch("content_tail");
}

static int blend_impl(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx, const float* alpha,
                      int alpha_per_row, int T, int64_t R, int d, float* out, int64_t ldo, float* out_norm, float audit_eps,
                      uint64_t* audit_counts, double* audit_absdev, void* stream);

int hwer_blend_normalize_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx,
                             const float* alpha, int alpha_per_row, int T, int64_t R, int d, float* out,
                             int64_t ldo, float* out_norm, void* stream) {
    return blend_impl(tables, ld_tables, row_idx, alpha, alpha_per_row, T, R, d, out, ldo, out_norm, 0.f, nullptr, nullptr, stream);
}

int hwer_blend_normalize_audit_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx,
                                   const float* alpha, int alpha_per_row, int T, int64_t R, int d, float* out,
                                   int64_t ldo, float eps, uint64_t* audit_counts, double* audit_absdev_sum, void* stream) {
    if (!audit_counts || !audit_absdev_sum) return fail(HWER_EINVAL, "blend_normalize_audit: null audit output");
    // one memset node when the two outputs are laid out back to back (counts[4] then the sum), else two
    if (reinterpret_cast<const void*>(audit_absdev_sum) == reinterpret_cast<const void*>(audit_counts + 4)) {
        if (hipMemsetAsync(audit_counts, 0, 48, as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "blend_normalize_audit: memset failed");
    } else {
        if (hipMemsetAsync(audit_counts, 0, 32, as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "blend_normalize_audit: memset failed");
        if (hipMemsetAsync(audit_absdev_sum, 0, 8, as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "blend_normalize_audit: memset failed");
    }
    return blend_impl(tables, ld_tables, row_idx, alpha, alpha_per_row, T, R, d, out, ldo, nullptr, eps, audit_counts, audit_absdev_sum, stream);
}

static int blend_impl(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx, const float* alpha,
                      int alpha_per_row, int T, int64_t R, int d, float* out, int64_t ldo, float* out_norm, float audit_eps,
                      uint64_t* audit_counts, double* audit_absdev, void* stream) {
    if (R == 0) return HWER_OK;
    if (!tables || !alpha || T < 1) return fail(HWER_EINVAL, "blend_normalize: null tables/alpha or T < 1");
    int rc;
    if ((rc = check_table("blend_normalize(out)", out, ldo, d))) return rc;
    if (ld_tables % 4 != 0 || ld_tables < d) return fail(HWER_EINVAL, "blend_normalize: ld_tables must be a multiple of 4 and >= d");
    const int d4 = (d + 3) / 4;
    // every block ends with one float64 atomic on the audit sum (same address for all blocks: they serialise in L2), so small
    // tables use fewer, longer blocks: about 32 rows per block, at least one block per CU, at most kMaxStreamBlocks
    int grid = rows_grid(R, d4);
    if (audit_counts) grid = std::min(grid, int(std::max<int64_t>(256, cdiv64(R, 32))));
    if (d4 <= 64 && T <= 2)   // one float4 chunk per lane: four rows per iteration (2 x 4 loads in flight per lane)
        hipLaunchKernelGGL((blend_normalize_kernel<1, 4>), dim3(grid), dim3(256), 0, as_stream(stream), tables,
                       ld_tables, row_idx, alpha, alpha_per_row, T, R, d4, out, ldo, out_norm, audit_eps,
                       reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev);
    else if (d4 <= 64)        // more tables: the rows of one iteration (measured at T = 4: U = 4 is 17 % slower than U = 1)
        hipLaunchKernelGGL((blend_normalize_kernel<1, 1>), dim3(grid), dim3(256), 0, as_stream(stream), tables,
                       ld_tables, row_idx, alpha, alpha_per_row, T, R, d4, out, ldo, out_norm, audit_eps,
                       reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev);
    else
        hipLaunchKernelGGL((blend_normalize_kernel<MAXC, 1>), dim3(grid), dim3(256), 0, as_stream(stream), tables,
                       ld_tables, row_idx, alpha, alpha_per_row, T, R, d4, out, ldo, out_norm, audit_eps,
                       reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev);
    return check_launch("blend_normalize");
}

int hwer_rank_metrics(const uint64_t* hits, int hit_words, const int32_t* n_pred, const int64_t* vrow, const int64_t* n_true, int64_t V,
                      const int32_t* cuts_host, int n_cuts, const double* log2_table, int ld_log2, const double* idcg_table, double* vals_ws,
                      double* out_means, void* stream) {
    if (!hits || !n_pred || !vrow || !n_true || !cuts_host || !log2_table || !idcg_table || !vals_ws || !out_means)
        return fail(HWER_EINVAL, "rank_metrics: null pointer");
    if (V < 1 || n_cuts < 1 || n_cuts > 8 || hit_words < 1) return fail(HWER_EINVAL, "rank_metrics: need V >= 1, 1 <= n_cuts <= 8");
    RankMetricsArgs a{hits, hit_words, n_pred, vrow, n_true, V, n_cuts, {0, 0, 0, 0, 0, 0, 0, 0}, log2_table, ld_log2, idcg_table, vals_ws, out_means};
    for (int i = 0; i < n_cuts; ++i) {
        if (cuts_host[i] < 1 || cuts_host[i] > 128 || cuts_host[i] > hit_words * 64 || cuts_host[i] > ld_log2)
            return fail(HWER_EINVAL, "rank_metrics: cuts must lie in 1 .. min(128, 64 hit_words, ld_log2)");
        a.cuts[i] = cuts_host[i];
    }
    hipLaunchKernelGGL(rank_metrics_rows_kernel, dim3(unsigned(cdiv64(V, 256)), unsigned(n_cuts)), dim3(256), 0, as_stream(stream), a);
    int rc;
    if ((rc = check_launch("rank_metrics(rows)"))) return rc;
    hipLaunchKernelGGL(rank_metrics_mean_kernel, dim3(unsigned(2 * n_cuts)), dim3(64), 0, as_stream(stream), a);
    return check_launch("rank_metrics(mean)");
}

int hwer_rank_filter_seen(const int64_t* ranked, int64_t B, int k, const int64_t* seen_ptr, const int64_t* seen_idx,
                          const int64_t* truth_ptr, const int64_t* truth_idx, int cut_max, int64_t* out_unseen,
                          int32_t* out_len, uint64_t* out_hits, uint32_t* cover_bits, int64_t cover_base, int64_t cover_n,
                          void* stream) {
    if (B == 0) return HWER_OK;
    if (!ranked || !out_unseen || !out_len || !out_hits) return fail(HWER_EINVAL, "rank_filter_seen: null pointer");
    if (k < 1 || cut_max < 1 || cut_max > 128) return fail(HWER_EINVAL, "rank_filter_seen: need k >= 1 and 1 <= cut_max <= 128");
    if ((seen_ptr && !seen_idx) || (truth_ptr && !truth_idx)) return fail(HWER_EINVAL, "rank_filter_seen: ptr without idx");
    if (cover_bits && cover_n < 1) return fail(HWER_EINVAL, "rank_filter_seen: cover_n < 1");
    const int n_words = (cut_max + 63) / 64;
    hipLaunchKernelGGL(rank_filter_seen_kernel, dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, as_stream(stream), ranked, B, k, seen_ptr,
                       seen_idx, truth_ptr, truth_idx, cut_max, out_unseen, out_len, out_hits, n_words, cover_bits, cover_base, cover_n);
    return check_launch("rank_filter_seen");
}

int64_t hwer_unit_audit_ws_floats(int64_t N) { return ((N + 3) / 4) * 4 + (AUDIT_BLOCKS * 4 * 2) + 8; }

// X1 pack of the user-partitioned exchange: send[g][c][j][:] = array c (0 idx, 1 score bits, 2 dist2 bits) of query
// g * Bs + j, padding rows (query >= B) = (id -1, score 0, dist2 +inf).  One wave per 8-byte row of k elements.
__global__ __launch_bounds__(256) void pack_exchange_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ score,
                                                            const int64_t* __restrict__ dist2, int64_t B, int k, int64_t Bs, int64_t n_rows,
                                                            int64_t* __restrict__ send) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6); row < n_rows; row += int64_t(gridDim.x) * 4) {
        const int64_t j = row % Bs, gc = row / Bs;
        const int c = int(gc % 3);
        const int64_t q = (gc / 3) * Bs + j;
        const int64_t* src = (c == 0 ? idx : c == 1 ? score : dist2) + q * k;
        const int64_t pad = c == 0 ? int64_t(-1) : c == 1 ? int64_t(0) : int64_t(0x7FF0000000000000ll);
        int64_t* dst = send + row * k;
        for (int e = lane; e < k; e += 64) dst[e] = q < B ? src[e] : pad;
    }
}

// number of queries whose flag is 0 (uncertified) -> out[0]; lets the host decide on the exhaustive pass with one
// 4-byte read instead of a device-wide nonzero()
__global__ __launch_bounds__(256) void count_zero_flags_kernel(const int32_t* __restrict__ flags, int64_t B, int32_t* __restrict__ out) {
    int c = 0;
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < B; i += int64_t(gridDim.x) * 256) c += flags[i] == 0;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) c += __shfl_xor(c, off, kWave);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);   // uncertified queries are rare: normally no atomic at all
}
int hwer_pack_exchange(const int64_t* idx, const double* score, const double* dist2, int64_t B, int k, int n_ranks,
                       int64_t* send, void* stream) {
    if (!idx || !score || !dist2 || !send) return fail(HWER_EINVAL, "pack_exchange: null pointer");
    if (B < 1 || k < 1 || n_ranks < 1) return fail(HWER_EINVAL, "pack_exchange: need B, k, n_ranks >= 1");
    const int64_t Bs = cdiv64(B, n_ranks);
    const int64_t n_rows = int64_t(n_ranks) * 3 * Bs;
    const int grid = int(std::min<int64_t>(cdiv64(n_rows, 4), 8192));
    hipLaunchKernelGGL(pack_exchange_kernel, dim3(grid), dim3(256), 0, as_stream(stream), idx, reinterpret_cast<const int64_t*>(score),
                       reinterpret_cast<const int64_t*>(dist2), B, k, Bs, n_rows, send);
    return check_launch("pack_exchange");
}

int hwer_knn_count_uncertified(const int32_t* flags, int64_t B, int32_t* out_count, void* stream) {
    if (!flags || !out_count || B < 0) return fail(HWER_EINVAL, "count_uncertified: bad argument");
    if (hipMemsetAsync(out_count, 0, sizeof(int32_t), as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "count_uncertified: memset failed");
    if (B == 0) return HWER_OK;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(512, cdiv64(B, 1024))));
    hipLaunchKernelGGL(count_zero_flags_kernel, dim3(grid), dim3(256), 0, as_stream(stream), flags, B, out_count);
    return check_launch("count_uncertified");
}

int hwer_unit_audit_f32(const float* V, int64_t N, int d, int64_t ldv, float eps, float* row_norm_ws,
                        uint64_t* out_counts, double* out_absdev_sum, void* stream) {
    int rc;
    if (N < 1) return fail(HWER_EINVAL, "unit_audit: N < 1");
    if ((rc = check_table("unit_audit", V, ldv, d))) return rc;
    if (!row_norm_ws || !out_counts || !out_absdev_sum) return fail(HWER_EINVAL, "unit_audit: null output");
    const int d4 = (d + 3) / 4;
    const RowNormArgs ra{V, ldv, d, d4, nullptr, N, nullptr, 0, 0, row_norm_ws, nullptr, 0};
    if (d4 <= 64)
        hipLaunchKernelGGL((row_norm_kernel<false, 1, 4>), dim3(rows_grid(N, d4)), dim3(256), 0, as_stream(stream), ra);
    else
        hipLaunchKernelGGL((row_norm_kernel<false, MAXC, 1>), dim3(rows_grid(N, d4)), dim3(256), 0, as_stream(stream), ra);
    if ((rc = check_launch("unit_audit(norms)"))) return rc;
    // partials + ticket live behind the N row norms (callers size row_norm_ws with hwer_unit_audit_ws_floats(N))
    double* partial = reinterpret_cast<double*>(row_norm_ws + ((N + 3) / 4) * 4);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(partial + AUDIT_BLOCKS * 4);
    // zero the arrival ticket on every call (a memset node on the stream: the workspace may be freshly allocated)
    if (hipMemsetAsync(ticket, 0, 16, as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "unit_audit: memset failed");
    const int blocks = int(std::max<int64_t>(1, std::min<int64_t>(AUDIT_BLOCKS, cdiv64(N, 1024))));
    hipLaunchKernelGGL(audit_reduce_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), row_norm_ws, N, eps, out_counts,
                       out_absdev_sum, partial, ticket);
    return check_launch("unit_audit(reduce)");
}

int64_t hwer_knn_padded_rows(int64_t n_rows) { return ((n_rows + ROW_PAD - 1) / ROW_PAD) * ROW_PAD; }

int hwer_knn_padded_dim(int d) {
    if (d < 1 || d > 256) return -1;
    if (d <= 32) return 32;
    if (d <= 64) return 64;
    if (d <= 128) return 128;
    return 256;
}

int hwer_pack_rows_f32(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids, int64_t n_rows,
                       float* X, int64_t ldx, float* hnorm, void* stream) {
    return hwer_pack_rows_f32_f16(V, N, d, ldv, row_ids, n_rows, X, ldx, hnorm, nullptr, 0, stream);
}

int hwer_pack_rows_f32_f16(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids, int64_t n_rows,
                           float* X, int64_t ldx, float* hnorm, void* X16, int x_exp, void* stream) {
    if (n_rows == 0) return HWER_OK;
    int rc;
    if ((rc = check_table("pack_rows(V)", V, ldv, d))) return rc;
    if ((rc = check_table("pack_rows(X)", X, ldx, d))) return rc;
    if (!hnorm) return fail(HWER_EINVAL, "pack_rows: null hnorm");
    if (!row_ids && n_rows > N) return fail(HWER_EINVAL, "pack_rows: n_rows > N without row_ids");
    const int d4 = (d + 3) / 4, x4 = int(ldx / 4);
    if (x4 > 64 * MAXC) return fail(HWER_EINVAL, "pack_rows: ldx too large");
    if (X16 && ((reinterpret_cast<uintptr_t>(X16) & 7) != 0 || ldx % 4 != 0)) return fail(HWER_EINVAL, "pack_rows: X16 must be 8-byte aligned");
    const RowNormArgs ra{V, ldv, d, d4, row_ids, n_rows, X, ldx, x4, hnorm, reinterpret_cast<half4_rt*>(X16), x_exp};
    if (x4 <= 64)
        hipLaunchKernelGGL((row_norm_kernel<true, 1, 4>), dim3(rows_grid(n_rows, x4)), dim3(256), 0, as_stream(stream), ra);
    else
        hipLaunchKernelGGL((row_norm_kernel<true, MAXC, 1>), dim3(rows_grid(n_rows, x4)), dim3(256), 0, as_stream(stream), ra);
    return check_launch("pack_rows");
}

int hwer_query_mix(const float* V, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                   const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, int arith,
                   double* Q64, float* Q32, int64_t ldq32, void* stream) {
    if (B == 0) return HWER_OK;
    if (!V || !anchor || !Q64 || !Q32) return fail(HWER_EINVAL, "query_mix: null pointer");
    if (N < 1 || d < 1 || d > 1024 || ldv < d || ldq32 < d) return fail(HWER_EINVAL, "query_mix: bad shape");
    if ((pos_ptr && !pos_idx) || (neg_ptr && !neg_idx)) return fail(HWER_EINVAL, "query_mix: ptr without idx");
    if (arith != 0 && arith != 1) return fail(HWER_EINVAL, "query_mix: arith must be 0 (float64) or 1 (float32)");
    // anchor-only queries in bulk: the row-streaming kernel (same values; from 4 096 queries on it is the faster one)
    if (!pos_ptr && !neg_ptr && B >= 4096 && ldv % 4 == 0 && ldq32 % 4 == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(Q32) & 15) == 0 && !diag_env("HWER_NO_ANCHOR_QUERY_KERNEL")) {
        const size_t ts = arith == 0 ? sizeof(double) : sizeof(float);
        const int d4 = (d + 3) / 4;
        int stride = 4 * d4;
        if (arith == 0) { while (stride % 32 != 8) stride += 2; } else { while (stride % 32 != 8) ++stride; }   // the 8-lane chains of 4 rows on distinct banks
        int pb_n = 8;
        while (pb_n > 2 && size_t(pb_n) * stride * ts > size_t(4800)) pb_n >>= 1;
        const size_t lds = size_t(4) * (size_t(pb_n) * stride + 8) * ts;
        if (lds <= size_t(64) * 1024) {
            const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(cdiv64(B, pb_n), 4), kMaxStreamBlocks * 2)));
            if (arith == 0) hipLaunchKernelGGL(anchor_query_kernel<double>, dim3(grid), dim3(256), lds, as_stream(stream), V, ldv, d, d4, anchor, B, Q64, Q32, ldq32, stride, pb_n);
            else hipLaunchKernelGGL(anchor_query_kernel<float>, dim3(grid), dim3(256), lds, as_stream(stream), V, ldv, d, d4, anchor, B, Q64, Q32, ldq32, stride, pb_n);
            return check_launch("query_mix(anchors)");
        }
    }
    int gq = 8;
    while (gq > 1 && size_t(gq) * 3 * d * (arith == 0 ? sizeof(double) : sizeof(float)) > size_t(128) * 1024) gq >>= 1;
    const size_t per_wave = size_t(gq) * 3 * d * (arith == 0 ? sizeof(double) : sizeof(float));
    const int wpb = per_wave * 4 <= 64 * 1024 ? 4 : 1;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(B, gq * wpb), kMaxStreamBlocks * 4)));
    const QueryMixArgs qa{V, ldv, d, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, Q64, Q32, ldq32, nullptr, 0, nullptr, gq == 8 ? 0 : gq};
    if (arith == 0) hipLaunchKernelGGL(query_mix_kernel<double>, dim3(grid), dim3(64 * wpb), per_wave * wpb, as_stream(stream), qa);
    else hipLaunchKernelGGL(query_mix_kernel<float>, dim3(grid), dim3(64 * wpb), per_wave * wpb, as_stream(stream), qa);
    return check_launch("query_mix");
}

static int query_mix_f64rows_impl(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                                  const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                                  int64_t ldq32, void* knn_workspace, int64_t bq, void* stream);
int hwer_query_mix_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                           const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                           int64_t ldq32, void* stream) {
    return query_mix_f64rows_impl(V64, N, d, ldv, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, Q64, Q32, ldq32, nullptr, 0, stream);
}
// ... and, like hwer_prepare_search_f32 for float32 rows, the MFMA query fragments of the search that follows written straight into its workspace
// (float32 fragments for bq > 0, float16 fragments + exponents for bq < 0: |bq| = hwer_knn_query_padding(...)); search with HWER_KNN_QUERIES_READY
int hwer_query_mix_frag_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                                const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                                int64_t ldq32, void* knn_workspace, int64_t bq, void* stream) {
    if (!knn_workspace || bq == 0) return fail(HWER_EINVAL, "query_mix_frag_f64rows: workspace and bq required (hwer_query_mix_f64rows writes no fragments)");
    return query_mix_f64rows_impl(V64, N, d, ldv, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, Q64, Q32, ldq32, knn_workspace, bq, stream);
}
static int query_mix_f64rows_impl(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                                  const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                                  int64_t ldq32, void* knn_workspace, int64_t bq, void* stream) {
    if (B == 0) return HWER_OK;
    const int frag16 = bq < 0 ? 1 : 0;
    if (frag16) bq = -bq;
    if (knn_workspace) {
        if ((reinterpret_cast<uintptr_t>(knn_workspace) & 255) != 0 || bq < B || bq % 32 != 0)
            return fail(HWER_EINVAL, "query_mix_frag_f64rows: workspace must be 256-byte aligned and |bq| = hwer_knn_query_padding(...) >= B");
        if (ldq32 % 8 != 0 || (frag16 && ldq32 % 16 != 0) || d > 256) return fail(HWER_EINVAL, "query_mix_frag_f64rows: ldq32 must be the search's padded width (d <= 256)");
    }
    if (!V64 || !anchor || !Q64 || !Q32) return fail(HWER_EINVAL, "query_mix_f64rows: null pointer");
    if (N < 1 || d < 1 || d > 1024 || ldv < d || ldq32 < d) return fail(HWER_EINVAL, "query_mix_f64rows: bad shape");
    if ((pos_ptr && !pos_idx) || (neg_ptr && !neg_idx)) return fail(HWER_EINVAL, "query_mix_f64rows: ptr without idx");
    // anchor-only queries in bulk (round 6): the row-streaming kernel of hwer_query_mix on float64 rows - same values as the general kernel below
    if (!knn_workspace && !pos_ptr && !neg_ptr && B >= 4096 && ldv % 4 == 0 && ldq32 % 4 == 0 && (reinterpret_cast<uintptr_t>(V64) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(Q32) & 15) == 0 && !diag_env("HWER_NO_ANCHOR_QUERY_KERNEL")) {
        const int d4 = (d + 3) / 4;
        int stride = 4 * d4;
        while (stride % 32 != 8) stride += 2;
        int pb_n = 8;
        while (pb_n > 2 && size_t(pb_n) * stride * sizeof(double) > size_t(4800)) pb_n >>= 1;
        const size_t lds = size_t(4) * (size_t(pb_n) * stride + 8) * sizeof(double);
        if (lds <= size_t(64) * 1024) {
            const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(cdiv64(B, pb_n), 4), kMaxStreamBlocks * 2)));
            hipLaunchKernelGGL((anchor_query_kernel<double, double>), dim3(grid), dim3(256), lds, as_stream(stream), V64, ldv, d, d4, anchor, B, Q64, Q32, ldq32, stride, pb_n);
            return check_launch("query_mix_f64rows(anchors)");
        }
    }
    int gq = 8;
    while (gq > 1 && size_t(gq) * 3 * d * sizeof(double) > size_t(128) * 1024) gq >>= 1;
    const size_t per_wave = size_t(gq) * 3 * d * sizeof(double);
    const int wpb = per_wave * 4 <= 64 * 1024 ? 4 : 1;
    const int64_t q_rows = knn_workspace ? std::max(B, bq) : B;      // (fragment rows past the batch are written as zeros)
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(q_rows, gq * wpb), kMaxStreamBlocks * 4)));
    float* qf = knn_workspace ? reinterpret_cast<float*>(static_cast<unsigned char*>(knn_workspace) + WS_HEADER_BYTES) : nullptr;
    const QueryMixArgs qa{nullptr, ldv, d, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, Q64, Q32, ldq32, qf, knn_workspace ? bq : 0, V64, gq == 8 ? 0 : gq, frag16};
    hipLaunchKernelGGL(query_mix_kernel<double>, dim3(grid), dim3(64 * wpb), per_wave * wpb, as_stream(stream), qa);
    return check_launch("query_mix_f64rows");
}

int hwer_prepare_search_f32(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids, int64_t n_rows, float* X, int64_t ldx,
                            float* hnorm, void* X16, int x_exp, const int32_t* anchor, const int64_t* pos_ptr, const int32_t* pos_idx,
                            const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, int arith, double* Q64, float* Q32, int64_t ldq32,
                            void* knn_workspace, int64_t bq, void* stream) {
    if (n_rows == 0 || B == 0) {   // nothing to fuse: the two stages on their own
        int rc = hwer_pack_rows_f32_f16(V, N, d, ldv, row_ids, n_rows, X, ldx, hnorm, X16, x_exp, stream);
        if (rc) return rc;
        return hwer_query_mix(V, N, d, ldv, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, arith, Q64, Q32, ldq32, stream);
    }
    int rc;
    if ((rc = check_table("prepare_search(V)", V, ldv, d))) return rc;
    if ((rc = check_table("prepare_search(X)", X, ldx, d))) return rc;
    if (!hnorm || !anchor || !Q64 || !Q32) return fail(HWER_EINVAL, "prepare_search: null pointer");
    if (!row_ids && n_rows > N) return fail(HWER_EINVAL, "prepare_search: n_rows > N without row_ids");
    if (N < 1 || ldq32 < d || ldq32 % 8 != 0) return fail(HWER_EINVAL, "prepare_search: ldq32 must be a multiple of 8 and >= d");
    if ((pos_ptr && !pos_idx) || (neg_ptr && !neg_idx)) return fail(HWER_EINVAL, "prepare_search: ptr without idx");
    if (arith != 0 && arith != 1) return fail(HWER_EINVAL, "prepare_search: arith must be 0 (float64) or 1 (float32)");
    if (X16 && (reinterpret_cast<uintptr_t>(X16) & 7) != 0) return fail(HWER_EINVAL, "prepare_search: X16 must be 8-byte aligned");
    const int frag16 = bq < 0 ? 1 : 0;   // bq < 0: float16 fragments (the search then runs with HWER_KNN_F16_SCREEN | HWER_KNN_QUERIES_READY) for -bq rows
    if (frag16) bq = -bq;
    if (knn_workspace && ((reinterpret_cast<uintptr_t>(knn_workspace) & 255) != 0 || bq < B || bq % 32 != 0))
        return fail(HWER_EINVAL, "prepare_search: workspace must be 256-byte aligned and |bq| = hwer_knn_query_padding(...) >= B");
    if (frag16 && ldq32 % 16 != 0) return fail(HWER_EINVAL, "prepare_search: float16 fragments need ldq32 to be a multiple of 16");
    const int d4 = (d + 3) / 4, x4 = int(ldx / 4);
    if (x4 > 64) return fail(HWER_EINVAL, "prepare_search: d <= 256 only");
    const size_t per_wave = size_t(8) * 3 * d * (arith == 0 ? sizeof(double) : sizeof(float));
    const int wpb = per_wave * 4 <= 64 * 1024 ? 4 : 1;
    const int64_t q_rows = knn_workspace ? std::max(B, bq) : B;
    const int n_pack = rows_grid(n_rows, x4);
    const int n_mix = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(q_rows, 8 * wpb), kMaxStreamBlocks * 4)));
    const RowNormArgs pa{V, ldv, d, d4, row_ids, n_rows, X, ldx, x4, hnorm, reinterpret_cast<half4_rt*>(X16), x_exp};
    // right behind the workspace header: where hwer_knn_search_opts_f32 keeps its query fragments (ws_layout: w.qf)
    float* qf = knn_workspace ? reinterpret_cast<float*>(static_cast<unsigned char*>(knn_workspace) + WS_HEADER_BYTES) : nullptr;
    const QueryMixArgs qa{V, ldv, d, anchor, pos_ptr, pos_idx, neg_ptr, neg_idx, B, Q64, Q32, ldq32, qf, bq, nullptr, 0, frag16};
    if (arith == 0)
        hipLaunchKernelGGL((prepare_search_kernel<double, 1, 4>), dim3(n_pack + n_mix), dim3(256), per_wave * wpb, as_stream(stream), pa, qa, n_pack, wpb);
    else
        hipLaunchKernelGGL((prepare_search_kernel<float, 1, 4>), dim3(n_pack + n_mix), dim3(256), per_wave * wpb, as_stream(stream), pa, qa, n_pack, wpb);
    return check_launch("prepare_search");
}


int hwer_knn_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || workspace_bytes < size_t(WS_HEADER_BYTES)) return fail(HWER_EINVAL, "knn_workspace_init: workspace too small");
    if (hipMemsetAsync(workspace, 0, WS_HEADER_BYTES, as_stream(stream)) != hipSuccess) return fail(HWER_ELAUNCH, "knn_workspace_init: memset failed");
    // (a device-side fill: no pageable host source, valid under stream capture)
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&static_cast<WsHeader*>(workspace)->init_magic), int(WS_INIT_MAGIC), 1, as_stream(stream)) != hipSuccess)
        return fail(HWER_ELAUNCH, "knn_workspace_init: fill failed");
    return HWER_OK;
}

int hwer_blend_prepare_search_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx, const float* alpha,
                                  int alpha_per_row, int T, int d, float* V, int64_t ldv, float eps, uint64_t* audit_counts,
                                  double* audit_absdev_sum, const int64_t* pack_ids, int64_t pack_row0, int64_t n_rows,
                                  const int64_t* other_ids, int64_t other_row0, int64_t n_other, float* X, int64_t ldx, float* hnorm,
                                  void* X16, int x_exp, const int32_t* anchor, int64_t B, int arith, double* Q64, float* Q32,
                                  int64_t ldq32, void* knn_workspace, int64_t bq, void* stream) {
    if (!tables || !alpha || T < 1) return fail(HWER_EINVAL, "blend_prepare_search: null tables/alpha or T < 1");
    if (!audit_counts || !audit_absdev_sum) return fail(HWER_EINVAL, "blend_prepare_search: null audit output");
    int rc;
    if ((rc = check_table("blend_prepare_search(V)", V, ldv, d))) return rc;
    if ((rc = check_table("blend_prepare_search(X)", X, ldx, d))) return rc;
    if (ld_tables % 4 != 0 || ld_tables < d) return fail(HWER_EINVAL, "blend_prepare_search: ld_tables must be a multiple of 4 and >= d");
    if (n_rows < 1 || n_other < 0 || B < 1) return fail(HWER_EINVAL, "blend_prepare_search: need n_rows >= 1, n_other >= 0, B >= 1");
    if (!hnorm || !anchor || !Q64 || !Q32) return fail(HWER_EINVAL, "blend_prepare_search: null pointer");
    if (ldq32 < d || ldq32 % 8 != 0) return fail(HWER_EINVAL, "blend_prepare_search: ldq32 must be a multiple of 8 and >= d");
    if (arith != 0 && arith != 1) return fail(HWER_EINVAL, "blend_prepare_search: arith must be 0 (float64) or 1 (float32)");
    if (X16 && (reinterpret_cast<uintptr_t>(X16) & 7) != 0) return fail(HWER_EINVAL, "blend_prepare_search: X16 must be 8-byte aligned");
    if (!knn_workspace || (reinterpret_cast<uintptr_t>(knn_workspace) & 255) != 0)
        return fail(HWER_EINVAL, "blend_prepare_search: needs the (256-byte aligned, hwer_knn_workspace_init-ed) search workspace");
    const int frag16 = bq < 0 ? 1 : 0;   // bq < 0: float16 fragments for -bq rows (see hwer_prepare_search_f32)
    if (frag16) bq = -bq;
    if (bq != 0 && (bq < B || bq % 32 != 0)) return fail(HWER_EINVAL, "blend_prepare_search: |bq| must be 0 (no fragments) or hwer_knn_query_padding(...) >= B");
    if (frag16 && ldq32 % 16 != 0) return fail(HWER_EINVAL, "blend_prepare_search: float16 fragments need ldq32 to be a multiple of 16");
    const int d4 = (d + 3) / 4, x4 = int(ldx / 4);
    if (x4 > 64) return fail(HWER_EINVAL, "blend_prepare_search: d <= 256 only");
    // query staging: G queries per wave and trip, as many as keep the block's LDS (4 waves) within 32 KB - the same launch streams
    // the rows, and those blocks want many workgroups per CU
    const size_t per_query = size_t(2) * d * (arith == 0 ? sizeof(double) : sizeof(float)) + size_t(4) * d4 * sizeof(float);
    int gq = 8;
    while (gq > 1 && size_t(4) * gq * per_query > 32 * 1024) gq /= 2;
    const size_t per_wave = gq * per_query;
    const int wpb = 4;
    const int64_t q_rows = bq ? std::max(B, bq) : B;
    // about 32 rows per PACK / OTHER block, at least ... at most the streaming grid (every such block ends with one set of atomics)
    int n_pack = int(std::min<int64_t>(rows_grid(n_rows, x4), std::max<int64_t>(64, cdiv64(n_rows, 32))));
    int n_oth = n_other ? int(std::min<int64_t>(rows_grid(n_other, d4), std::max<int64_t>(64, cdiv64(n_other, 32)))) : 0;
    const int n_mix = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(q_rows, gq * wpb), kMaxStreamBlocks * 4)));
    // Large tables: ONE round of resident workgroups (4 per CU at this kernel's 111 VGPRs), every block a grid-stride loop over an
    // equal share of the rows - with 2 048 + 2 048 blocks the kernel ran 2.6 rounds and its last, 60 % filled round was a tail of
    // ~10 % (round 3, same-box sweep of the PACK blocks 2 048 -> 768 ... 1 024: c4 0.54 -> 0.49 ms, c5 1.65 -> 1.5 ms, c4full +-0).
    // The row-streaming blocks share what the QUERY blocks leave, in proportion to their rows.
    if (!diag_env("HWER_ROW_BLOCKS")) {
        const int64_t slots = int64_t(4) * device_cu_count();
        const int64_t budget = slots - std::min<int64_t>(n_mix, slots / 4);
        if (int64_t(n_pack) + n_oth > budget) {
            const double share = double(n_rows) / double(n_rows + n_other);
            int64_t np = std::max<int64_t>(1, std::min<int64_t>(n_pack, int64_t(share * double(budget) + 0.5)));
            int64_t no = n_other ? std::max<int64_t>(1, std::min<int64_t>(n_oth, budget - np)) : 0;
            np = std::max<int64_t>(1, std::min<int64_t>(n_pack, budget - no));
            n_pack = int(np);
            n_oth = int(no);
        }
    }
    WsHeader* hdr = static_cast<WsHeader*>(knn_workspace);
    const BlendArgs ba{tables, ld_tables, row_idx, alpha, alpha_per_row, T, d, d4, V, ldv, eps, hdr,
                       reinterpret_cast<unsigned long long*>(audit_counts), audit_absdev_sum, unsigned(n_pack + n_oth)};
    const RowRange pack{pack_ids, pack_row0, n_rows}, other{other_ids, other_row0, n_other};
    const RowNormArgs pk{V, ldv, d, d4, nullptr, n_rows, X, ldx, x4, hnorm, reinterpret_cast<half4_rt*>(X16), x_exp};
    float* qf = bq ? reinterpret_cast<float*>(static_cast<unsigned char*>(knn_workspace) + WS_HEADER_BYTES) : nullptr;
    const QueryMixArgs qa{V, ldv, d, anchor, nullptr, nullptr, nullptr, nullptr, B, Q64, Q32, ldq32, qf, bq, nullptr, 0, frag16};
    const dim3 grid(n_pack + n_oth + n_mix);
    // four rows in flight per row group whatever T is (measured at T = 4, d = 256: 1.63 ms, against 2.38 with one row in flight and
    // 1.84 for the three separate launches - the stand-alone blend kernel prefers one row there, this one does not)
    int wpb_arg = wpb;
    if (const char* e = diag_env("HWER_PROLOGUE_ABLATE")) {   // diagnostic builds only; the first launches run whole so that every buffer holds valid rows
        static int calls = 0;
        if (++calls > 3) wpb_arg |= atoi(e) << 8;
    }
    if (arith == 0) hipLaunchKernelGGL((blend_prepare_search_kernel<double, 4>), grid, dim3(256), per_wave * wpb, as_stream(stream), ba, pack, other, pk, qa, n_pack, n_oth, wpb_arg, gq);
    else hipLaunchKernelGGL((blend_prepare_search_kernel<float, 4>), grid, dim3(256), per_wave * wpb, as_stream(stream), ba, pack, other, pk, qa, n_pack, n_oth, wpb_arg, gq);
    return check_launch("blend_prepare_search");
}

}  // extern "C"
