// Stage 2 of the exact k-NN (K5-K7): candidate selection, float64 re-rank in the reference's exact summation
// orders, exactness certificate, re-score + final sort; plus the exhaustive float64 fallback and the multi-GPU
// merge (X1).  One 256-thread workgroup per query.
//
// Orders reproduced bit for bit (verified against numpy 2.2.6 / scikit-learn 1.7.2, see oracle/hwer_oracle.py):
//   dist2 : scikit-learn `euclidean_rdist64`  -> sequential  d += (q_j - x_j)^2, separate mul and add roundings
//   score : numpy `(a * b).sum(1)`             -> pairwise summation of the rounded products
#include <algorithm>
#include <string>
#include <cstdlib>
#include <type_traits>

#include "knn_internal.h"

namespace hwer {

// ---------------------------------------------------------------------------------------------------
// LDS bitonic sorts (256 threads, n2 = power of two)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_desc_u64(uint64_t* a, int n2) {
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));   // = (t / stride) * 2 stride + t % stride, stride a power of two
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t x = a[lo], y = a[hi];
                if ((x < y) == desc) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
    }
    __syncthreads();
}

struct Ent {  // sorted ascending by (hi, lo)
    uint64_t hi, lo;
};
__device__ __forceinline__ bool ent_less(const Ent& x, const Ent& y) { return x.hi < y.hi || (x.hi == y.hi && x.lo < y.lo); }

__device__ __forceinline__ void bitonic_asc_ent(Ent* a, int n2) {
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n2 >> 1); t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));   // = (t / stride) * 2 stride + t % stride, stride a power of two
                const int hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const Ent x = a[lo], y = a[hi];
                if (ent_less(y, x) == asc) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ uint64_t f64_orderable(double v) {
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return u ^ ((u >> 63) ? 0xFFFFFFFFFFFFFFFFull : 0x8000000000000000ull);
}

__device__ __forceinline__ int pow2ceil_i(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---------------------------------------------------------------------------------------------------
// shared tail: given `n_c` candidate items in LDS (c_item), compute exact dist2 / score, sort, certify, emit
// ---------------------------------------------------------------------------------------------------
struct EmitArgs {
    const float* X;
    uint32_t item_max;   // n_items - 1: item ids are clamped before they address X (a logic error must never fault)
    int64_t ldx;
    int d;
    const double* Q64;
    const float* A32;
    int64_t lda;
    const int32_t* a_idx;  // nullable: anchor row of query b is A32[a_idx[b]] (else A32[b])
    int score_mode;
    int sort_by_score;
    int debug_skip;   // ablation only (HWER_SELECT_SKIP bits: 1 = no re-rank rounds, 2 = no sort, 4 = no selection passes)
    const int64_t* row_ids;
    int64_t id_base;
    int k;
    int64_t* out_idx;
    double* out_score;
    double* out_dist2;
    int32_t* out_flags;
    WsHeader* hdr;    // nullable: uncertified queries (flag 0) are appended to the header's list
    unsigned long long* stamps;   // diagnostic build only (HWER_SELECT_STAMPS=1): per-query phase cycle counts, 8 per query
};
#ifdef HWER_DIAG_STAMPS
#define SEL_STAMP(slot) do { if (a_stamps && (threadIdx.x & 63) == 0) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); a_stamps[(slot)] = t_; \
    if ((slot) == 0) a_stamps[10] = __builtin_amdgcn_s_memrealtime(); if ((slot) == 6) a_stamps[11] = __builtin_amdgcn_s_memrealtime(); } } while (0)   /* 10 / 11: start / end in 10-ns ticks (one clock for the whole chip) */
#else
#define SEL_STAMP(slot) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------
// Wave-per-query back end.  One 64-lane wave owns one query from the candidate keys to the output rows; there
// is no workgroup barrier anywhere (a 256-thread block is just four independent queries).
// LDS per wave, for a candidate capacity kp (multiple of 64) and kp2 = pow2 >= kp:
//   t_d2[kp] f64 | t_sc[kp] f64 | cand[kp] u32 | stage 64 x 20 f32
// ---------------------------------------------------------------------------------------------------
constexpr int RR_CW = 16;            // columns per re-rank step
constexpr int RR_RS = RR_CW + 4;     // LDS row stride of the staging area in floats (5 x 16 B: odd -> conflict-free ds_read_b128)
__host__ __device__ inline size_t wave_lds_bytes(int d, int kp) {
    (void)d;
    return size_t(kp) * (8 + 8 + 4) + size_t(64) * RR_RS * 4;
}
// COOP variant: ONE query per 256-thread workgroup - the four waves share the query's tables and split the re-rank rounds
// (one staging area per wave); for launches with so few queries that a wave per query leaves most of the chip idle
__host__ __device__ inline size_t coop_lds_bytes(int d, int kp) { return wave_lds_bytes(d, kp) + size_t(3) * 64 * RR_RS * 4; }

__device__ __forceinline__ float e32_bound(int d, double Q2, float hnorm_max) {
    // |key32 - (q.x - 0.5||x||^2)| <= E32 for every item: fmaf chain of d+1 terms, float32 rounding of q and of the
    // half norm.  Generous constant factor; typical errors are ~100x smaller.
    // With M = ||q|| ||x||max + hnorm_max the first-order error of the plain chain is (d + 3) u M (d + 1 additions on partial sums <= M, the
    // roundings of q and of the half norm).  The ping-pong screen's folded form (knn_screen.hip: the accumulator runs tau' - key', the key is
    // recovered as tau' - accumulator) has partial sums <= |tau'| + M <= 2 M (tau' is a key of the same query) over d + 2 additions and one more
    // rounding in the recovery: (2 d + 7) u M - still inside the (2 d + 16) u M used here, whose second-order remainder (d u)^2 M is 1e-9 M.
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double xn = sqrt(2.0 * double(hnorm_max)) * (1.0 + 1e-6);
    return float((2.0 * d + 16.0) * u * (sqrt(Q2) * xn + double(hnorm_max)) * (1.0 + 1e-6));
}

// Re-rank + emit for ONE query by ONE wave.  cand[0..n_r) (wave-private LDS, part of `lds`): local item ids, n_r >= k.
//   dist2: scikit-learn euclidean_rdist64 — sequential, (q - x) in float64, mul and add rounded separately
//   score: numpy pairwise summation of the rounded products (8 chains per <=128-element leaf, then the tail)
// then: the k smallest (dist2, item), ordered by (score desc, distance rank) [= the reference's stable sort by score
// of the distance-ordered list] or left in (dist2, item) order when sort_by_score == 0.
// Lazy distances: when the caller does not ask for distances (out_dist2 == NULL, score order, score_mode 0/1), only the
// first n_b candidates - the BOUNDARY of the float32 screen, whose membership in the top-k is undecided - get their
// exact float64 distance; the others are provably members (their screen key clears the k-th key by more than the
// float32 error bound) and only need their score.  Distances are 60 % of the float64 work of a full re-rank.  Should two
// members tie exactly in score with an unknown distance between them, the lanes that hold them compute it on the spot.
template <int MODE, bool COOP = false>   // score_mode as a compile-time constant: the float64 and float32 score chains never share registers
__device__ __forceinline__ void wave_rerank_emit_t(const EmitArgs& a, int64_t b, unsigned char* lds, int kp, int n_r, int n_b, int flag) {
    const int d = a.d, lane = threadIdx.x & 63;
    const int coop_wave = COOP ? int(threadIdx.x >> 6) : 0;
    unsigned long long* const a_stamps = (a.stamps && coop_wave == 0) ? a.stamps + b * 16 : nullptr;
    (void)a_stamps;
    SEL_STAMP(3);
    double* t_d2 = reinterpret_cast<double*>(lds);
    double* t_sc = t_d2 + kp;
    uint32_t* cand = reinterpret_cast<uint32_t*>(t_sc + kp);
    float* stage = reinterpret_cast<float*>(cand + kp) + coop_wave * (64 * RR_RS);

    // The query and the anchor row are wave-uniform: they are read with SCALAR loads straight from global memory and
    // enter the float64 arithmetic as SGPR operands.  (Broadcasting them from LDS costs a full 1-KiB LDS return per
    // 16 bytes of payload and made the LDS pipe the bottleneck of the whole kernel.)
    const int64_t bu = (int64_t(__builtin_amdgcn_readfirstlane(int(uint64_t(b) >> 32))) << 32) |
                       int64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(b)))));
    // (constant address space: loads from it with a uniform address are selected as s_load; both arrays were written
    // by earlier kernels and are read-only here)
    typedef const __attribute__((address_space(4))) double* cdouble_p;
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cdouble_p q64 = (cdouble_p)(a.Q64 + bu * d);
    const int64_t arow = a.a_idx ? int64_t(__builtin_amdgcn_readfirstlane(a.a_idx[bu])) : bu;
    const cfloat_p a32 = (cfloat_p)(a.A32 ? a.A32 + arow * a.lda : a.X);   // mode 2 never reads it
    // numpy pairwise structure for this d (d <= 256: at most two leaves)
    const bool two = d > 128;
    int n2 = d / 2;
    n2 -= n2 & 7;
    const int gA = two ? n2 / 8 : -1;          // first 8-group of the second leaf
    const int lim = d - (d & 7);               // elements >= lim form the sequential tail of the last leaf
    const int n_groups8 = lim / 8;
    constexpr int mode = MODE;
    const int n_chunks = (d + RR_CW - 1) / RR_CW;
    const bool lazy = a.out_dist2 == nullptr && a.sort_by_score && mode != 2 && n_b < n_r;

#pragma unroll 1
    for (int g0 = 64 * coop_wave; g0 < ((a.debug_skip & 1) ? 0 : n_r); g0 += COOP ? 256 : 64) {   // wave-uniform rounds of 64 candidates
#ifdef HWER_DIAG_STAMPS   // ablation of the round's ingredients (HWER_SELECT_SKIP bits 8 no distances, 16 no score arithmetic, 32 no row gathers, 64 no LDS staging)
        const bool need_d2 = (!lazy || g0 < n_b) && !(a.debug_skip & 8);
        const bool abl_score = a.debug_skip & 16, abl_gather = a.debug_skip & 32, abl_stage = a.debug_skip & 64;
#else
        const bool need_d2 = !lazy || g0 < n_b;   // wave-uniform
        constexpr bool abl_score = false, abl_gather = false, abl_stage = false;
#endif
        const int ci = g0 + lane;
        const bool live = ci < n_r;
        const uint32_t it = min(cand[live ? ci : g0], a.item_max);
        double acc = 0.0;
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, sA = 0.0;
        float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0, fA = 0.f;
        // staging map: 4 lanes x 16 B per row (whole 64-byte segments: the L1 serves 64 B per clock, per-lane row walks
        // would use a quarter of each access), 16 rows per instruction, 4 instructions per 16-column chunk; the rows are
        // transposed through wave-private LDS (row stride 80 B: conflict-free ds_read_b128).
        // (named scalars, not arrays: hipcc leaves small arrays touched inside a rolled loop in scratch)
        const int srow = lane >> 2, scol = 4 * (lane & 3);
        const float* src0 = a.X + int64_t(min(cand[min(g0 + 0 + srow, n_r - 1)], a.item_max)) * a.ldx + scol;
        const float* src1 = a.X + int64_t(min(cand[min(g0 + 16 + srow, n_r - 1)], a.item_max)) * a.ldx + scol;
        const float* src2 = a.X + int64_t(min(cand[min(g0 + 32 + srow, n_r - 1)], a.item_max)) * a.ldx + scol;
        const float* src3 = a.X + int64_t(min(cand[min(g0 + 48 + srow, n_r - 1)], a.item_max)) * a.ldx + scol;
        float* dst = stage + srow * RR_RS + scol;
        // software pipeline: the gathers of chunk c + 1 are in flight while chunk c is reduced in float64
        // (round 3: a second chunk of look-ahead - +16 VGPRs, 128 with a few spilled - measured slower: c2 search 0.155 -> 0.160 ms, c3 4.80 -> 4.99)
        float4 pre0 = *reinterpret_cast<const float4*>(src0), pre1 = *reinterpret_cast<const float4*>(src1);
        float4 pre2 = *reinterpret_cast<const float4*>(src2), pre3 = *reinterpret_cast<const float4*>(src3);
        if (COOP && g0 == 0) SEL_STAMP(7);
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            if (!abl_stage) {
            *reinterpret_cast<float4*>(dst) = pre0;
            *reinterpret_cast<float4*>(dst + 16 * RR_RS) = pre1;
            *reinterpret_cast<float4*>(dst + 32 * RR_RS) = pre2;
            *reinterpret_cast<float4*>(dst + 48 * RR_RS) = pre3;
            }
            if (c + 1 < n_chunks && !abl_gather) {
                const int o = RR_CW * (c + 1);
                pre0 = *reinterpret_cast<const float4*>(src0 + o);
                pre1 = *reinterpret_cast<const float4*>(src1 + o);
                pre2 = *reinterpret_cast<const float4*>(src2 + o);
                pre3 = *reinterpret_cast<const float4*>(src3 + o);
            }
            if (!abl_stage) wave_lds_sync();
            if (COOP && g0 == 0 && c == 0) SEL_STAMP(8);
            float xv[RR_CW];
#pragma unroll
            for (int t = 0; t < RR_CW / 4; ++t) {
                const float4 v = abl_stage ? pre0 : *reinterpret_cast<const float4*>(stage + lane * RR_RS + 4 * t);
                xv[4 * t] = v.x; xv[4 * t + 1] = v.y; xv[4 * t + 2] = v.z; xv[4 * t + 3] = v.w;
            }
            if (!abl_stage) wave_lds_sync();
            const int j0 = RR_CW * c;
            if (!need_d2) {
            } else if (j0 + RR_CW <= d) {   // whole chunk (always, when d is a multiple of 16): straight-line code
#pragma unroll
                for (int t = 0; t < RR_CW; ++t) {
                    const double tmp = __dsub_rn(q64[j0 + t], double(xv[t]));
                    acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                }
            } else {
#pragma unroll
                for (int t = 0; t < RR_CW; ++t) {
                    if (j0 + t < d) {   // uniform
                        const double tmp = __dsub_rn(q64[j0 + t], double(xv[t]));
                        acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                    }
                }
            }
            if constexpr (mode != 2) {
#pragma unroll
                for (int gg = 0; gg < RR_CW / 8; ++gg) {
                    const int g8 = (RR_CW / 8) * c + gg;
                    if (g8 < n_groups8 && !abl_score) {   // uniform
                        const bool first = g8 == 0 || g8 == gA;
                        const int jj = 8 * g8;
                        if constexpr (mode == 0) {
                            // a and x are float32 values, so a * x is EXACT in float64 (48 significant bits): fma(a, x, r) rounds
                            // r + a x once, exactly like numpy's add of the (unrounded-because-exact) product - one instruction less
                            // per element than mul + add, bit-identical results.
                            // (round 5, measured and NOT kept: the anchor row widened to float64 once per search and read as scalar float64 operands -
                            // 2 instead of 3 vector instructions per element, but twice the scalar-load bytes and 32 SGPRs per 8-group in a kernel at
                            // 106 SGPRs: c3 select 2.19 -> 2.42 ms, c4 / c5 +3 - 4 %, profiles/r5_ab_a64_anchors.txt)
                            // (also measured and not kept: the chunk's 16 anchor floats requested at the top of the step, in front of the wait for the
                            // gathered rows, so that their scalar-cache latency hides under it - 105 more spilled SGPRs and a full lgkmcnt drain in front
                            // of the next gathers: c2 select 121 -> 125 us, c3 2.12 -> 2.16 ms, profiles/r5_ab_select_scalar_hoist.txt)
                            const double a0 = double(a32[jj + 0]), a1 = double(a32[jj + 1]), a2 = double(a32[jj + 2]), a3 = double(a32[jj + 3]);
                            const double a4 = double(a32[jj + 4]), a5 = double(a32[jj + 5]), a6 = double(a32[jj + 6]), a7 = double(a32[jj + 7]);
                            const double x0 = double(xv[8 * gg + 0]), x1 = double(xv[8 * gg + 1]), x2 = double(xv[8 * gg + 2]), x3 = double(xv[8 * gg + 3]);
                            const double x4 = double(xv[8 * gg + 4]), x5 = double(xv[8 * gg + 5]), x6 = double(xv[8 * gg + 6]), x7 = double(xv[8 * gg + 7]);
                            if (first) { r0 = __dmul_rn(a0, x0); r1 = __dmul_rn(a1, x1); r2 = __dmul_rn(a2, x2); r3 = __dmul_rn(a3, x3);
                                         r4 = __dmul_rn(a4, x4); r5 = __dmul_rn(a5, x5); r6 = __dmul_rn(a6, x6); r7 = __dmul_rn(a7, x7); }
                            else { r0 = __fma_rn(a0, x0, r0); r1 = __fma_rn(a1, x1, r1); r2 = __fma_rn(a2, x2, r2); r3 = __fma_rn(a3, x3, r3);
                                   r4 = __fma_rn(a4, x4, r4); r5 = __fma_rn(a5, x5, r5); r6 = __fma_rn(a6, x6, r6); r7 = __fma_rn(a7, x7, r7); }
                            if (g8 == gA - 1) sA = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                        } else {
                            const float p0 = __fmul_rn((a32[jj + 0]), xv[8 * gg + 0]), p1 = __fmul_rn((a32[jj + 1]), xv[8 * gg + 1]);
                            const float p2 = __fmul_rn((a32[jj + 2]), xv[8 * gg + 2]), p3 = __fmul_rn((a32[jj + 3]), xv[8 * gg + 3]);
                            const float p4 = __fmul_rn((a32[jj + 4]), xv[8 * gg + 4]), p5 = __fmul_rn((a32[jj + 5]), xv[8 * gg + 5]);
                            const float p6 = __fmul_rn((a32[jj + 6]), xv[8 * gg + 6]), p7 = __fmul_rn((a32[jj + 7]), xv[8 * gg + 7]);
                            if (first) { f0 = p0; f1 = p1; f2 = p2; f3 = p3; f4 = p4; f5 = p5; f6 = p6; f7 = p7; }
                            else { f0 = __fadd_rn(f0, p0); f1 = __fadd_rn(f1, p1); f2 = __fadd_rn(f2, p2); f3 = __fadd_rn(f3, p3);
                                   f4 = __fadd_rn(f4, p4); f5 = __fadd_rn(f5, p5); f6 = __fadd_rn(f6, p6); f7 = __fadd_rn(f7, p7); }
                            if (g8 == gA - 1) fA = ((f0 + f1) + (f2 + f3)) + ((f4 + f5) + (f6 + f7));
                        }
                    }
                }
            }
        }
        if (COOP && g0 == 0) SEL_STAMP(9);
        double sco;
        const float* xr = a.X + int64_t(it) * a.ldx;
        if constexpr (mode == 0) {
            double res = n_groups8 > 0 ? ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) : 0.0;
            for (int j = lim; j < d; ++j) res = __dadd_rn(res, __dmul_rn(double(a32[j]), double(xr[j])));   // <= 7 tail elements
            if (two) res = sA + res;
            sco = (res + 1.0) / 2.0;
        } else if constexpr (mode == 1) {
            float res = n_groups8 > 0 ? ((f0 + f1) + (f2 + f3)) + ((f4 + f5) + (f6 + f7)) : 0.f;
            for (int j = lim; j < d; ++j) res = __fadd_rn(res, __fmul_rn(a32[j], xr[j]));
            if (two) res = fA + res;
            sco = double((res + 1.0f) / 2.0f);
        } else {
            sco = (-1.0 * sqrt(acc) + 2.0) / 2.0;  // hwer/gcn_ncf.py:381
        }
        if (live) {
            t_d2[ci] = need_d2 ? acc : -1.0;   // -1: distance not computed (a provable member)
            t_sc[ci] = sco;
        }
    }
    // COOP, usual (lazy, one-round boundary, <= 256 candidates) case: the whole workgroup puts the candidates in score order - one
    // candidate per thread, position = number of keys below its own, counted over the 256 keys in LDS (independent broadcast
    // reads instead of the 36 dependent exchange stages of a register sort by one wave: 12.4 K -> ~4 K cycles) - and wave 0 only
    // resolves colliding approximations and writes the rows.
    bool presorted = false;
    const uint64_t* s_sorted = nullptr;
    if constexpr (COOP) {   // all four waves' rounds are in the shared tables
        SEL_STAMP(12);
        __syncthreads();
        SEL_STAMP(13);
        if (lazy && n_b <= 64 && n_r <= 256 && !(a.debug_skip & 6)) {   // block-uniform
            uint64_t* skeys = reinterpret_cast<uint64_t*>(cand + kp);     // the four staging areas (20 KB) are free now
            uint64_t* sfinal = skeys + 256;
            const int i = 64 * coop_wave + lane;
            bool member = i < n_r && i >= n_b;
            if (coop_wave == 0 && lane < n_b) {   // boundary candidates: the k - n_sure nearest are in (counting rank by distance)
                const uint64_t dm = f64_orderable(t_d2[lane]);
                const uint32_t im = cand[lane];
                int rank = 0;
#pragma unroll 4
                for (int j = 0; j < n_b; ++j) {
                    const uint64_t dj = f64_orderable(t_d2[j]);
                    const uint32_t ij = cand[j];
                    rank += (dj < dm || (dj == dm && ij < im)) ? 1 : 0;
                }
                member = rank < a.k - (n_r - n_b);
            }
            SEL_STAMP(14);
            // unique keys: members by (approximate score, slot), everything else behind them by slot
            const uint64_t mykey = (uint64_t(member ? ~f32_orderable(float(t_sc[min(i, kp - 1)])) : 0xFFFFFFFFu) << 32) | uint64_t(uint32_t(i));
            // Round 4: every wave sorts its 64 keys in registers (21 DPP / permlane-swap stages), the four ascending runs go to LDS and a key's
            // position is its lane + its lower bounds in the three other runs (7 dependent probes each, the three searches interleaved): ~250
            // instructions per wave where counting the smaller ones among all 256 keys took ~900 (128 broadcast reads, 256 64-bit compares) -
            // stamps: 8.0 K -> ~2.5 K cycles of wave 0's chain per query.
            uint64_t skey = mykey;
            wave_sort1_u64(skey);
            skeys[i] = skey;      // run w = skeys[64 w .. 64 w + 63]
            __syncthreads();
            int pos = lane;
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) pos += run64_count_less(skeys + 64 * ((coop_wave + w2) & 3), skey);
            sfinal[pos] = skey;
            __syncthreads();
            SEL_STAMP(15);
            presorted = true;
            s_sorted = sfinal;
        }
        if (coop_wave != 0) return;
    } else {
        wave_lds_sync();
    }
    SEL_STAMP(4);
    // ---- membership and output order, two register sorts (NE keys per lane) ------------------------------------------------
    // (1) by (dist2, item): positions < k are the members and the position is the KD-tree's return order;
    // (2) by (score desc, position): the reference's stable sort by score.  Sort keys are 64-bit:
    //     float32 rounding of the exact key (monotone) << 32 | payload, finished exactly where roundings collide.
    const int k = a.k;
    // distance of ONE candidate by ONE lane (divergent, rare): exact score ties between members whose distance was not
    // computed are ordered by (distance, item) like every other tie
    auto ensure_d2 = [&](uint32_t slot) -> double {
        double d2 = t_d2[slot];
        if (d2 < 0.0) {
            const float* xr = a.X + int64_t(min(cand[slot], a.item_max)) * a.ldx;
            d2 = 0.0;
            for (int j = 0; j < d; ++j) {
                const double tmp = __dsub_rn(q64[j], double(xr[j]));
                d2 = __dadd_rn(d2, __dmul_rn(tmp, tmp));
            }
            t_d2[slot] = d2;
        }
        return d2;
    };
    auto tail = [&](auto ne_tag) {
        constexpr int NE = decltype(ne_tag)::value;
        uint64_t key[NE];
        // Lazy mode with a one-round boundary (the usual case): no distance sort at all.  Every provable member is in;
        // of the n_b <= 64 boundary candidates (slots 0 .. n_b - 1, one per lane) the k - n_sure nearest are in, found
        // by counting, per candidate, the boundary candidates that beat it.
        const bool rank_boundary = lazy && n_b <= 64 && !(a.debug_skip & 2);
        if (presorted) {   // (NE == 4) the workgroup's counting sort left sorted position p in s_sorted[p]
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const uint64_t kk = s_sorted[(NE * lane + j) & 255];
                key[j] = (kk >> 32) == 0xFFFFFFFFull ? ~0ull : kk;   // non-members: the padding pattern of the register path
            }
            wave_sort_fixup<NE>(key, [&](uint64_t x, uint64_t y) {
                const uint32_t sx = uint32_t(x) & 0xFFFFu, sy = uint32_t(y) & 0xFFFFu;
                const uint64_t ax = ~f64_orderable(t_sc[sx]), ay = ~f64_orderable(t_sc[sy]);
                if (ax != ay) return ax < ay;
                const uint64_t dx = f64_orderable(ensure_d2(sx)), dy = f64_orderable(ensure_d2(sy));
                return dx < dy || (dx == dy && cand[sx] < cand[sy]);
            });
        } else if (rank_boundary) {
            const int n_keep = k - (n_r - n_b);
            const bool mine = lane < n_b;
            const uint64_t dm = mine ? f64_orderable(t_d2[lane]) : ~0ull;
            const uint32_t im = mine ? cand[lane] : 0xFFFFFFFFu;
            int rank = 0;
            // (round 5) candidate j's (distance, id) comes out of lane j's registers through v_readlane - three scalar reads and four compares
            // per boundary candidate, nothing waits: the LDS broadcast reads this loop made before were a dependent ~300-cycle round trip per
            // candidate (c2 stamps: 8.9 K cycles for ~30 candidates)
            for (int j = 0; j < n_b; ++j) {   // wave-uniform
                const uint32_t dlo = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(dm)), j)), dhi = uint32_t(__builtin_amdgcn_readlane(int(uint32_t(dm >> 32)), j));
                const uint64_t dj = (uint64_t(dhi) << 32) | dlo;
                const uint32_t ij = uint32_t(__builtin_amdgcn_readlane(int(im), j));
                rank += (dj < dm || (dj == dm && ij < im)) ? 1 : 0;
            }
            const bool member_b = mine && rank < n_keep;
            SEL_STAMP(7);
            bool ordered = false;
            if constexpr (NE == 4) {
            if (!(a.debug_skip & 256)) {
                ordered = true;
                // (round 5) score order from four register-sorted runs and their merge ranks - the one-workgroup-per-query select's method
                // (round 4) inside one wave: every register is sorted across the lanes (21 DPP / permlane exchange stages), the four ascending
                // runs go to the (now idle) staging area, a key's position is its lane + its lower bounds in the three other runs (7 probes
                // each, twelve independent searches per lane), and the keys come back in the blocked layout the tie fix-up and the emission
                // expect.  Non-members carry (0xFFFFFFFF, slot): distinct keys, behind every member.  ~600 instructions where the 36-stage
                // bitonic network over 256 keys took ~1 300 and 300 ds_bpermute round trips.
                uint64_t* skeys = reinterpret_cast<uint64_t*>(stage);      // 4 runs x 64 keys, then sfinal: 256 keys (4 KB of the 5 KB staging area)
                uint64_t* sfinal = skeys + 256;
                static_assert(size_t(64) * RR_RS * 4 >= size_t(256) * 8, "the staging area holds the 256 keys");
                uint64_t sk[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 64 * j + lane;
                    const bool member = i < n_r && (i >= n_b || (j == 0 && member_b));
                    sk[j] = (uint64_t(member ? ~f32_orderable(float(t_sc[min(i, kp - 1)])) : 0xFFFFFFFFu) << 32) | uint64_t(uint32_t(i));
                    wave_sort1_u64(sk[j]);
                }
                SEL_STAMP(8);
                // the staging area holds 64 x RR_RS floats = 5 120 bytes: the four runs (2 KB) first, the merged order after a second pass
                uint64_t* runs = skeys;
#pragma unroll
                for (int j = 0; j < 4; ++j) runs[64 * j + lane] = sk[j];
                wave_lds_sync();
                int pos[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pos[j] = lane;
#pragma unroll
                    for (int w2 = 1; w2 < 4; ++w2) pos[j] += run64_count_less(runs + 64 * ((j + w2) & 3), sk[j]);
                }
                wave_lds_sync();
                uint64_t* merged = skeys;        // every lane has read what it needs of the runs: the merged order overwrites them
#pragma unroll
                for (int j = 0; j < 4; ++j) merged[pos[j]] = sk[j];
                wave_lds_sync();
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const uint64_t kk = merged[(NE * lane + j) & 255];
                    key[j] = (kk >> 32) == 0xFFFFFFFFull ? ~0ull : kk;
                }
                (void)sfinal;
                SEL_STAMP(9);
            }
            }
            if (!ordered) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int i = 64 * j + lane;
                const bool member = i < n_r && (i >= n_b || (j == 0 && member_b));
                key[j] = member ? (uint64_t(~f32_orderable(float(t_sc[i]))) << 32) | uint64_t(uint32_t(i)) : ~0ull;
            }
            SEL_STAMP(8);
            wave_sort_u64<NE>(key);
            SEL_STAMP(9);
            }
            wave_sort_fixup<NE>(key, [&](uint64_t x, uint64_t y) {
                const uint32_t sx = uint32_t(x) & 0xFFFFu, sy = uint32_t(y) & 0xFFFFu;
                const uint64_t ax = ~f64_orderable(t_sc[sx]), ay = ~f64_orderable(t_sc[sy]);
                if (ax != ay) return ax < ay;
                const uint64_t dx = f64_orderable(ensure_d2(sx)), dy = f64_orderable(ensure_d2(sy));
                return dx < dy || (dx == dy && cand[sx] < cand[sy]);
            });
        } else {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = 64 * j + lane;   // any initial placement will do
            key[j] = ~0ull;
            if (i < n_r) {
                const double d2 = t_d2[i];
                // unknown distances (provable members) sort first, with distinct approximations (their slot: below the
                // orderable bits of any non-negative float) so that they never look like a tie
                const uint32_t ap = d2 < 0.0 ? uint32_t(i) : f32_orderable(float(d2));
                key[j] = (uint64_t(ap) << 32) | uint64_t(uint32_t(i));
            }
        }
        if (!(a.debug_skip & 2)) {
            wave_sort_u64<NE>(key);
            wave_sort_fixup<NE>(key, [&](uint64_t x, uint64_t y) {
                const uint32_t sx = uint32_t(x) & 0xFFFFu, sy = uint32_t(y) & 0xFFFFu;
                const uint64_t dx = f64_orderable(t_d2[sx]), dy = f64_orderable(t_d2[sy]);
                return dx < dy || (dx == dy && cand[sx] < cand[sy]);
            });
            if (a.sort_by_score) {
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const int p = NE * lane + j;
                    const uint32_t slot = uint32_t(key[j]) & 0xFFFFu;
                    if (p < k) {
                        const uint32_t ap = ~f32_orderable(float(t_sc[slot]));
                        key[j] = (uint64_t(ap) << 32) | (uint64_t(uint32_t(p)) << 16) | uint64_t(slot);
                    } else {
                        key[j] = ~0ull;
                    }
                }
                wave_sort_u64<NE>(key);
                wave_sort_fixup<NE>(key, [&](uint64_t x, uint64_t y) {
                    const uint32_t sx = uint32_t(x) & 0xFFFFu, sy = uint32_t(y) & 0xFFFFu;
                    const uint64_t ax = ~f64_orderable(t_sc[sx]), ay = ~f64_orderable(t_sc[sy]);
                    if (ax != ay) return ax < ay;
                    if (!lazy) return (uint32_t(x) >> 16) < (uint32_t(y) >> 16);   // position in the distance order
                    const uint64_t dx = f64_orderable(ensure_d2(sx)), dy = f64_orderable(ensure_d2(sy));
                    return dx < dy || (dx == dy && cand[sx] < cand[sy]);
                });
            }
        }
        }   // !rank_boundary
        SEL_STAMP(5);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int p = NE * lane + j;
            if (p < k) {
                const uint32_t slot = uint32_t(key[j]) & 0xFFFFu;
                const uint32_t it = min(cand[slot], a.item_max);
                a.out_idx[b * k + p] = a.row_ids ? a.row_ids[it] : a.id_base + int64_t(it);
                a.out_score[b * k + p] = t_sc[slot];
                if (a.out_dist2) a.out_dist2[b * k + p] = t_d2[slot];
            }
        }
    };
    if (presorted) tail(std::integral_constant<int, 4>{});
    else if (n_r <= 128) tail(std::integral_constant<int, 2>{});
    else if (n_r <= 256) tail(std::integral_constant<int, 4>{});
    else tail(std::integral_constant<int, 8>{});
    SEL_STAMP(6);
    if (lane == 0) {
        a.out_flags[b] = flag;
        if (flag == 0 && a.hdr) {   // rare: hand the query to the exhaustive pass (hwer_knn_exact_device / _f64)
            const int p = atomicAdd(&a.hdr->uncert_count, 1);
            if (p < UNCERT_LIST_MAX) a.hdr->uncert_list[p] = int(b);
        }
    }
}

__device__ __forceinline__ void wave_rerank_emit(const EmitArgs& a, int64_t b, unsigned char* lds, int kp, int n_r, int n_b, int flag) {
    switch (a.score_mode) {   // wave-uniform
        case 0: wave_rerank_emit_t<0>(a, b, lds, kp, n_r, n_b, flag); break;
        case 1: wave_rerank_emit_t<1>(a, b, lds, kp, n_r, n_b, flag); break;
        default: wave_rerank_emit_t<2>(a, b, lds, kp, n_r, n_b, flag); break;
    }
}

// ---------------------------------------------------------------------------------------------------
constexpr int RR_MAXR = (KPRIME_MAX + 63) / 64;   // candidate keys per lane (7): a list holds at most K' keys
// select: one wave per query.  Input: ONE candidate list per query (<= K' keys, unsorted, split over the two lane
// streams).  Finds the k-th largest screen key by an in-register radix select, derives the re-rank set and the
// exactness certificate from the float32 error bound, then re-ranks.
// ---------------------------------------------------------------------------------------------------
struct SelectArgs {
    const uint64_t* cand;
    const int32_t* cand_cnt;
    int64_t bq;
    int64_t B;
    int kprime;
    int kcap;     // LDS capacity of the re-rank tables (= K')
    float hnorm_max;
    float key_rel_err;   // extra |key error| / (||q|| ||x||max) of the screen that produced the keys (float16 screen: ~2^-10)
    EmitArgs emit;
};

template <int MODE>
__global__ __launch_bounds__(256, 4) void select_rerank_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = int64_t(blockIdx.x) * 4 + wave;
    if (b >= a.B) return;   // no workgroup barrier below
    const int d = a.emit.d, k = a.emit.k;
    const int kp = a.kcap;                                             // candidate capacity (multiple of 64)
    unsigned char* lds_emit = smem_raw + size_t(wave) * wave_lds_bytes(d, kp);
    uint32_t* cand_item = reinterpret_cast<uint32_t*>(lds_emit + size_t(kp) * 16);   // = `cand` of wave_rerank_emit

    const int c0r = a.cand_cnt[b * 2], c1r = a.cand_cnt[b * 2 + 1];
    const int c0 = c0r & ~CNT_FLAGS, c1 = c1r & ~CNT_FLAGS;
    const int M = c0 + c1;                                             // <= K' (final compaction of the screen)
    // a list that kept only the keys above the query's seeded threshold is complete as long as the query's lists together
    // hold K' keys (every one of them above the threshold, hence above everything that was dropped)
    const bool seeded_short = ((c0r | c1r) & CNT_SEEDED_SHORT) != 0;
    const bool underfull = ((c0r | c1r) & CNT_UNDERFULL) != 0 || (seeded_short && M < a.kprime);
    const uint64_t* s0 = a.cand + (b * 2) * CAPH;
    const uint64_t* s1 = s0 + CAPH;
    uint64_t key[RR_MAXR];
#pragma unroll
    for (int i = 0; i < RR_MAXR; ++i) {
        const int e = 64 * i + lane;
        key[i] = e < c0 ? s0[e] : (e - c0 < c1 ? s1[e - c0] : 0ull);
    }
    // ---- (approximately) the k-th largest screen key: float bisection on the register-resident scores -------------------
    float sv[RR_MAXR];
#pragma unroll
    for (int i = 0; i < RR_MAXR; ++i) sv[i] = key[i] != 0ull ? key_score(key[i]) : -INFINITY;
    int c_ge;
    const bool lazy_ok = a.emit.out_dist2 == nullptr && a.emit.sort_by_score && a.emit.score_mode != 2;
    float t_hi;   // count(key32 >= t_hi) < k (and >= k - 12 when lazy_ok)
    const float t_k = wave_find_rank<RR_MAXR>(sv, k, 12, c_ge, &t_hi, lazy_ok);   // <= k-th largest key, <= 12 extra keys above it
    // ---- re-rank set + certificate (float32 only) ------------------------------------------------------------------
    // Every item has |key32 - key| <= E32.  An item can belong to the true top-k only if key32 >= t_k - 2 E32: those
    // candidates form the re-rank set.  Items that never became candidates have key32 <= min candidate key; if some
    // candidate already falls below t_k - 2 E32 (n_r < M), none of them can enter the top-k: provably exact.
    double Q2 = 0.0;
    {
        const double* q = a.emit.Q64 + b * d;
        for (int j = lane; j < d; j += 64) Q2 += q[j] * q[j];
        Q2 = wave_sum_f64(Q2);
    }
    const float e32 = e32_bound(d, Q2, a.hnorm_max) + a.key_rel_err * float(sqrt(Q2)) * sqrtf(2.0f * a.hnorm_max) * 1.0001f;
    const float thr = t_k - 2.0f * e32;
    // provable members ("sure"): key32 >= s with count(key32 >= s - 2 E32) < k - such an item cannot be beaten by k others
    const float s_sure = lazy_ok ? t_hi + 2.0f * e32 : INFINITY;
    int n_r = 0, n_b = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {   // boundary candidates first, then the sure ones
#pragma unroll
        for (int i = 0; i < RR_MAXR; ++i) {
            const bool in = pass == 0 ? (sv[i] >= thr && !(sv[i] >= s_sure)) : (sv[i] >= s_sure);
            const uint64_t bal = __ballot(in);
            if (in) cand_item[n_r + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = key_item(key[i]);
            n_r += __builtin_popcountll(bal);
        }
        if (pass == 0) n_b = n_r;
    }
    wave_lds_sync();
    const bool exhaustive = M < a.kprime && !underfull;   // no stream ever compacted: the candidates are ALL the items
    const int flag = (!underfull && (exhaustive || n_r < M)) ? 1 : 0;
    wave_rerank_emit_t<MODE>(a.emit, b, lds_emit, kp, n_r, n_b, flag);
}


// ---------------------------------------------------------------------------------------------------
// dense select (small catalogues): one wave per query reads its row of fp32 keys (written by the DENSE screen),
// radix-selects the k-th largest, collects the re-rank set and re-ranks.  Every item is a candidate, so the result is
// exact by construction unless the re-rank set overflows its capacity (massive ties) -> flag 0.
// ---------------------------------------------------------------------------------------------------
struct DenseSelectArgs {
    const float* keys;   // (B, ld)
    int64_t ld;          // multiple of 32; entries >= n_items hold -inf (pad rows have a +inf half norm)
    int64_t B;
    int kp;              // re-rank capacity (multiple of 64)
    float hnorm_max;
    EmitArgs emit;
    // LIST variant (small-batch path): the "row" of query b is its candidate list of packed keys
    const uint64_t* list;        // (B, list_cap)
    const int32_t* list_count;   // (B x COUNT_STRIDE) appended keys of query b at [b * COUNT_STRIDE] (may exceed the capacity: overflow)
    int list_cap;
    int list_min;                // K': a list shorter than this came from a threshold that was too high
    float key_rel_err;           // extra |key error| / (||q|| ||x||max) of the screen that produced the keys (float16 screen: ~2^-10)
};

// LIST = false: dense select over the key row written by the DENSE screen (every item is a candidate).
// LIST = true : the same selection over a candidate LIST of packed keys (knn_small.hip); items that are not in the list
//               have keys below every list key, so the usual certificate applies (some candidate below t_k - 2 E32).
// COOP: one query per workgroup, for launches of so few queries that a wave per query leaves most of the chip idle and the
// kernel's time is the serial chain of ONE wave.  The four waves split the query: every wave keeps a quarter of the key
// slices in registers (a quarter of the compares per counting probe; counts, extrema and write offsets are combined through a
// few LDS words and workgroup barriers - all waves run the same probes on the same totals, so their control flow is identical),
// then they split the re-rank rounds, and wave 0 finishes (sorts, output).
template <int MAXV, int MODE, bool LIST, bool COOP = false>   // float4 slices of the key row per lane: rows up to 256 * MAXV keys; score mode
__global__ __launch_bounds__(256, MAXV <= 16 ? 4 : 2) void select_dense_kernel(const DenseSelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = COOP ? int64_t(blockIdx.x) : int64_t(blockIdx.x) * 4 + wave;
    if (b >= a.B) return;   // block-uniform with COOP; otherwise no workgroup barrier below
    const int d = a.emit.d, k = a.emit.k, kp = a.kp;
    unsigned long long* const a_stamps = (a.emit.stamps && (!COOP || wave == 0)) ? a.emit.stamps + b * 16 : nullptr;
    (void)a_stamps;
    SEL_STAMP(0);
    unsigned char* lds_emit = smem_raw + (COOP ? size_t(0) : size_t(wave) * wave_lds_bytes(d, kp));
    uint32_t* cand_item = reinterpret_cast<uint32_t*>(lds_emit + size_t(kp) * 16);
    const float4* row = reinterpret_cast<const float4*>(a.keys + b * a.ld);
    const int n4 = int(a.ld / 4);
    // COOP: wave w owns slices w, w + 4, ...: MV = MAXV / 4 per lane; s_x = the workgroup's exchange words
    constexpr int MV = COOP ? (MAXV + 3) / 4 : MAXV;
    __shared__ float s_mm[COOP ? 8 : 1];
    __shared__ int s_cnt[COOP ? 8 : 1];
    __shared__ int s_tot[COOP ? 4 : 1];
    auto slice = [&](int mm) { return COOP ? wave + 4 * mm : mm; };
    int probe_parity = 0;
    (void)probe_parity;
    // The lane's slice of the key row lives in registers (rows up to 256 * MAXV keys): every load is in flight at once,
    // and the k-th key is bracketed by a float BISECTION on those registers (per-lane counting + one wave reduction per
    // step).  A byte-radix histogram in LDS is the wrong tool here: thousands of keys share their top bytes, so every
    // LDS atomic would serialise 64 lanes on one bin.
    float kv[MV][4];   // slices beyond the row hold -inf (below every real key; pad items are -inf too)
    const uint64_t* lst = nullptr;
    int n_list = 0, cnt_raw = 0;
    if constexpr (LIST) {
        lst = a.list + b * a.list_cap;
        cnt_raw = a.list_count[b * COUNT_STRIDE];
        n_list = min(cnt_raw, a.list_cap);
#pragma unroll
        for (int mm = 0; mm < MV; ++mm) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = 4 * (lane + 64 * slice(mm)) + c;
                kv[mm][c] = (slice(mm) < MAXV && j < n_list) ? key_score(lst[j]) : -INFINITY;
            }
        }
    } else {
#pragma unroll
        for (int mm = 0; mm < MV; ++mm) {
            const int i = lane + 64 * slice(mm);
            float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (slice(mm) < MAXV && i < n4) v = row[i];
            kv[mm][0] = v.x; kv[mm][1] = v.y; kv[mm][2] = v.z; kv[mm][3] = v.w;
        }
    }
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int m = 0; m < MV; ++m) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            vmax = fmaxf(vmax, kv[m][c]);
            if (kv[m][c] > -INFINITY) vmin = fminf(vmin, kv[m][c]);
        }
    }
    vmax = wave_max_f32(vmax);
    vmin = wave_min_f32(vmin);
    // 2 x 64 counters of the histogram rank search below, in the (not yet used) distance table of the re-rank; zeroed ahead of the barrier
    int* const hist = reinterpret_cast<int*>(lds_emit);
    if constexpr (COOP) {   // extrema of the whole row: the four waves' values through LDS
        if (wave < 2) hist[threadIdx.x] = 0;
        if (lane == 0) { s_mm[wave] = vmax; s_mm[4 + wave] = vmin; }
        __syncthreads();
        vmax = fmaxf(fmaxf(s_mm[0], s_mm[1]), fmaxf(s_mm[2], s_mm[3]));
        vmin = fminf(fminf(s_mm[4], s_mm[5]), fminf(s_mm[6], s_mm[7]));
    } else {
        hist[lane] = 0;
        hist[64 + lane] = 0;
        wave_lds_sync();
    }
    SEL_STAMP(1);
    auto count_ge = [&](float t) {
        int c = 0;   // ballot counting: one v_cmp per key, the popcount and the sum run on the scalar unit
#pragma unroll
        for (int m = 0; m < MV; ++m) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) c += __builtin_popcountll(__ballot(kv[m][cc] >= t));
        }
        if constexpr (COOP) {   // + the other waves' counts: one barrier per probe (two alternating sets of words)
            int* slot = s_cnt + 4 * probe_parity;
            probe_parity ^= 1;
            if (lane == 0) slot[wave] = c;
            __syncthreads();
            c = (slot[0] + slot[1]) + (slot[2] + slot[3]);
        }
        return c;
    };
    int c_lo;
    const bool lazy_ok = a.emit.out_dist2 == nullptr && a.emit.sort_by_score && a.emit.score_mode != 2;
    float t_hi;   // count(key32 >= t_hi) < k (and >= k - 12 when lazy_ok)
    // (round 3, measured and not kept, c1 = 943 queries x 1 682 keys in the one-query-per-workgroup form, phase stamps + same-box step times:
    //  three thresholds per workgroup exchange - 9.5 -> 4.3 barrier rounds, the same ~9.7 K cycles of rank search; the whole search by ONE
    //  wave on the whole row in its registers, ballot counts 13.0 K cycles, per-lane VALU counts + one wave reduction per probe ~ the same
    //  step time; a probe's cost follows the compares a SIMD has to issue for its four resident workgroups, not the barrier)
    // Round 4: the bracket comes from TWO histogram rounds instead of ~10 counting probes.  Every key gets a 12-bit bin number
    // bin(key) = int((key - vmin) * 4096 / (vmax - vmin)) - monotone in the key, so "bin(key) >= B" is "key >= the smallest key with bin >= B" and
    // the counts above a bin edge are exact.  Round 1 counts the keys by the upper 6 bits (64 LDS counters, one ds_add per key; linear bins
    // over the row's own range keep the lanes of an instruction on different counters, unlike a radix over raw float bits), a DPP suffix scan
    // finds the coarse bin that holds the k-th largest key; round 2 does the same with the lower 6 bits of the keys of that bin.  The fine bin
    // B* found holds pop >= 1 keys (about N / 4096 x the local density: one or two); t_k = the smallest key with bin >= B*, t_hi = the smallest
    // key above the bin: k <= count(>= t_k) = k - 1 + ... <= k + pop - 1 and k - pop <= count(>= t_hi) < k.  With pop <= 12 that is the
    // bracket the probes would have delivered (ties, clusters: the probes take over on the narrowed interval).  ~165 instead of ~450
    // instructions on the one-query-per-workgroup path, 3 workgroup barriers instead of ~10, and the SAME cost for every query: the probe
    // count (6 ... 13) was what spread the workgroups' end times (c1: rank search 6.8 K ... 19 K cycles, select launch 26.5 us).
    // (the wave-per-query form with 64 or 128 keys per lane keeps the probes: three more passes over that many registers spill)
    float t_k = 0.f;
    bool bracketed = false;
    if constexpr (COOP || MAXV <= 8) {
        const float range = vmax - vmin;
        const float scale = 4096.0f / range;
        if (range > 0.f && scale < 1e30f && vmin > -INFINITY && !(a.emit.debug_skip & 128)) {   // (workgroup-uniform)
            // (recomputed in each of the three passes - 3 VALU - rather than kept: the wave-per-query form holds up to 128 keys per lane)
            // -inf pads: (-inf) * scale = -inf -> INT_MIN (v_cvt_i32_f32 saturates): negative = "no key"
            const float vmin0 = vmin;
            auto bin_of = [&](float key) { return min(int((key - vmin0) * scale), 4095); };
#pragma unroll
            for (int m = 0; m < MV; ++m) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int bb = bin_of(kv[m][c]);
                    if (bb >= 0) atomicAdd(&hist[63 - (bb >> 6)], 1);   // lane l of the scan <-> coarse bin 63 - l
                }
            }
            if constexpr (COOP) __syncthreads();
            else wave_lds_sync();
            int s1 = wave_incl_scan_i32(hist[lane]);   // keys in coarse bins >= 63 - lane
            const uint64_t m1 = __ballot(s1 >= k);
            if (m1 != 0ull) {   // (else: fewer than k keys - the probes' own handling below)
                const int l1 = __builtin_ctzll(m1);
                const int c_above = l1 ? __builtin_amdgcn_readlane(s1, l1 - 1) : 0;   // keys in coarser bins above the one that holds the k-th
                const int coarse = 63 - l1;
#pragma unroll
                for (int m = 0; m < MV; ++m) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int bb = bin_of(kv[m][c]);
                        if ((bb >> 6) == coarse) atomicAdd(&hist[64 + 63 - (bb & 63)], 1);
                    }
                }
                if constexpr (COOP) __syncthreads();
                else wave_lds_sync();
                int s2 = c_above + wave_incl_scan_i32(hist[64 + lane]);
                const uint64_t m2 = __ballot(s2 >= k);   // never empty: the coarse bin holds the k-th key
                const int l2 = __builtin_ctzll(m2);
                const int n_lo = __builtin_amdgcn_readlane(s2, l2);                     // count(bin >= B*)
                const int n_hi = l2 ? __builtin_amdgcn_readlane(s2, l2 - 1) : c_above;   // count(bin > B*) < k
                const int bstar = coarse * 64 + 63 - l2;
                float lo = INFINITY, hi = INFINITY;
#pragma unroll
                for (int m = 0; m < MV; ++m) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int bb = bin_of(kv[m][c]);
                        lo = fminf(lo, bb >= bstar ? kv[m][c] : INFINITY);
                        hi = fminf(hi, bb > bstar ? kv[m][c] : INFINITY);
                    }
                }
                lo = wave_min_f32(lo);
                hi = wave_min_f32(hi);
                if constexpr (COOP) {
                    if (lane == 0) { s_mm[wave] = lo; s_mm[4 + wave] = hi; }
                    __syncthreads();
                    lo = fminf(fminf(s_mm[0], s_mm[1]), fminf(s_mm[2], s_mm[3]));
                    hi = fminf(fminf(s_mm[4], s_mm[5]), fminf(s_mm[6], s_mm[7]));
                }
                if (n_lo - n_hi <= 12 || !lazy_ok) {
                    if (n_lo <= k + 12) {
                        t_k = lo;
                        t_hi = hi;   // +inf when no key lies above the bin (k-th key in the top bin): no provable members, all boundary
                        c_lo = n_lo;
                        bracketed = true;
                    }
                }
                if (!bracketed) {   // more than 12 keys in one 1 / 4096 of the range: the probes finish on the narrowed interval
                    vmin = lo;
                    if (hi < INFINITY) vmax = hi;
                }
            }
        }
    }
    if (!bracketed) t_k = wave_find_rank_fn(count_ge, vmin, vmax, k, 12, c_lo, &t_hi, lazy_ok);   // <= k-th largest key, at most c_lo - k extra keys above it
    SEL_STAMP(2);
    double Q2 = 0.0;
    {
        const double* q = a.emit.Q64 + b * d;
        for (int j = lane; j < d; j += 64) Q2 += q[j] * q[j];
        Q2 = wave_sum_f64(Q2);
    }
    const float e32 = e32_bound(d, Q2, a.hnorm_max) + a.key_rel_err * float(sqrt(Q2)) * sqrtf(2.0f * a.hnorm_max) * 1.0001f;
    const float thr = t_k - 2.0f * e32;
    // provable members ("sure"): key32 >= s with count(key32 >= s - 2 E32) < k - such an item cannot be beaten by k others
    const float s_sure = lazy_ok ? t_hi + 2.0f * e32 : INFINITY;
    // re-rank set = keys >= thr (a few per lane): per-lane pass masks (2 VALU per key and class: the sign of key - thr is
    // funnel-shifted into the mask), wave scans for the write offsets, then every lane pops its set bits.  The BOUNDARY
    // candidates (thr <= key < s) go first, the sure ones after them.
    constexpr int NW = (4 * MV + 31) / 32;
    constexpr int KW = 4 * MV < 32 ? 4 * MV : 32;   // keys per mask word (4 slices: one half-filled word)
    constexpr uint32_t WMASK = KW == 32 ? 0xFFFFFFFFu : ((1u << KW) - 1u);
    uint32_t fail[NW], fail_s[NW];   // bit (KW - 1 - (e % KW)) of word e / KW <-> key e = 4 m + c of this lane; set = below the bound
#pragma unroll
    for (int w = 0; w < NW; ++w) fail[w] = fail_s[w] = 0u;
#pragma unroll
    for (int m = 0; m < MV; ++m) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = 4 * m + c;
            // key - bound has its sign bit set exactly when key < bound (-inf keys included; key == bound gives +0)
            fail[e / 32] = __builtin_amdgcn_alignbit(fail[e / 32], __float_as_uint(kv[m][c] - thr), 31);
            fail_s[e / 32] = __builtin_amdgcn_alignbit(fail_s[e / 32], __float_as_uint(kv[m][c] - s_sure), 31);
        }
    }
    int mine_b = 0, mine_s = 0;
    uint32_t pass_b[NW], pass_s[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        pass_s[w] = ~fail_s[w] & WMASK;
        pass_b[w] = ~fail[w] & fail_s[w] & WMASK;
        mine_b += __builtin_popcount(pass_b[w]);
        mine_s += __builtin_popcount(pass_s[w]);
    }
    const int incl = wave_incl_scan_i32((mine_b << 16) | mine_s);   // inclusive scan over lanes of both counts at once (each < 2^15)
    int tot = __builtin_amdgcn_readlane(incl, 63);
    int wave_off = 0;   // COOP: boundary / sure candidates of the waves before this one (packed like `incl`)
    if constexpr (COOP) {
        if (lane == 0) s_tot[wave] = tot;
        __syncthreads();
        tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            if (w2 < wave) wave_off += s_tot[w2];
            tot += s_tot[w2];
        }
    }
    int n_b = tot >> 16;
    int n_r = n_b + (tot & 0xFFFF);
    bool overflow = false;
    int pos_b = (wave_off >> 16) + (incl >> 16) - mine_b, pos_s = n_b + (wave_off & 0xFFFF) + (incl & 0xFFFF) - mine_s;
    auto pop = [&](uint32_t (&pm)[NW], int& pos) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            while (__any(pm[w] != 0u)) {
                if (pm[w] != 0u) {
                    const int bit = __builtin_ctz(pm[w]);
                    pm[w] &= pm[w] - 1u;
                    const int e = KW * w + KW - 1 - bit;
                    const uint32_t j = uint32_t(4 * (lane + 64 * slice(e >> 2)) + (e & 3));   // position in the row / list
                    if (pos < kp) cand_item[pos] = LIST ? key_item(lst[j]) : j;
                    ++pos;
                }
            }
        }
    };
    pop(pass_b, pos_b);
    pop(pass_s, pos_s);
    if (n_r > kp) {   // more near-ties than the re-rank capacity: not certifiable here
        overflow = true;
        n_r = kp;
        n_b = min(n_b, kp);
    }
    if constexpr (COOP) __syncthreads();   // every wave's candidates are in the shared list before any wave's rounds read it
    else wave_lds_sync();
    int flag = overflow ? 0 : 1;
    if constexpr (LIST) {
        // the list must be complete (no overflow, not cut short by a threshold above the K'-th key) and reach below the
        // re-rank bound; with fewer than k valid keys there is nothing to select from (k-th key = -inf)
        if (cnt_raw < a.list_min || cnt_raw > a.list_cap || n_r >= n_list) flag = 0;
        if (n_list < k) { n_r = 0; n_b = 0; }
    }
    wave_rerank_emit_t<MODE, COOP>(a.emit, b, lds_emit, kp, n_r, n_b, flag);
}

// pre-reduction when a query has more than MAX_LISTS_DIRECT lists: groups of 8 lists -> one list of K' keys
struct ReduceArgs {
    const uint64_t* cand_in;
    const int32_t* cnt_in;
    int n_lists_in;
    uint64_t* cand_out;
    int32_t* cnt_out;
    int64_t bq;
    int kprime;
    const int32_t* todo;   // nullable: reduce only the queries with todo[b] != 0 (the others were merged by reduce_lists_wave_kernel)
};

__global__ __launch_bounds__(256) void reduce_lists_kernel(const ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_off = reinterpret_cast<int*>(smem_raw);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);
    const int64_t b = blockIdx.x;
    if (a.todo && a.todo[b] == 0) return;   // block-uniform
    const int group = blockIdx.y;
    const int l0 = group * MAX_LISTS_DIRECT;
    const int nl = min(MAX_LISTS_DIRECT, a.n_lists_in - l0);
    const int n_streams = nl * 2;
    if (threadIdx.x == 0) {
        int off = 0, under = 0;
        for (int s = 0; s < n_streams; ++s) {
            s_off[s] = off;
            const int c = a.cnt_in[(int64_t(l0 + (s >> 1)) * a.bq + b) * 2 + (s & 1)];
            under |= c & CNT_FLAGS;
            off += c & ~CNT_FLAGS;
        }
        s_off[n_streams] = off;
        s_off[25] = under;
    }
    __syncthreads();
    const int M = s_off[n_streams];
    const int m2 = pow2ceil_i(max(M, 2));
    for (int s = 0; s < n_streams; ++s) {
        const uint64_t* src = a.cand_in + ((int64_t(l0 + (s >> 1)) * a.bq + b) * 2 + (s & 1)) * CAPH;
        const int n = s_off[s + 1] - s_off[s];
        for (int i = threadIdx.x; i < n; i += blockDim.x) keys[s_off[s] + i] = src[i];
    }
    for (int i = M + threadIdx.x; i < m2; i += blockDim.x) keys[i] = 0ull;
    bitonic_desc_u64(keys, m2);
    const int n_c = min(M, a.kprime);
    uint64_t* dst = a.cand_out + ((int64_t(group) * a.bq + b) * 2) * CAPH;  // streams 0 and 1 are contiguous
    for (int i = threadIdx.x; i < n_c; i += blockDim.x) dst[i] = keys[i];
    if (threadIdx.x == 0) {
        a.cnt_out[(int64_t(group) * a.bq + b) * 2 + 0] = min(n_c, CAPH) | s_off[25];
        a.cnt_out[(int64_t(group) * a.bq + b) * 2 + 1] = max(n_c - CAPH, 0);
    }
}

// All lists of a query in ONE pass by ONE wave, for the short lists seeded thresholds leave behind (a query's 16 - 32 lists
// then hold 4 - 8 K' keys together): the keys are gathered into wave-private LDS (up to REDW_MAX, scores in registers), the K'-th
// largest is bracketed by counting probes (wave_find_rank, exact: slack 0) and the K' best are written out as the query's
// single list - what two rounds of reduce_lists_kernel (workgroup per query and group of 8 lists, LDS bitonic sorts) did before.
// A query whose lists hold more than REDW_MAX keys is left to those kernels (todo[b] = 1).
constexpr int REDW_NV = 64;                 // keys per lane (their scores live in registers, the packed keys stay in LDS)
constexpr int REDW_MAX = REDW_NV * 64;      // 4 096 keys per query
constexpr int REDW_WAVES = 2;               // waves (= queries) per workgroup: 32 KB of LDS each
constexpr int REDW_MAX_LISTS = 32;          // one lane per stream
struct ReduceWaveArgs {
    const uint64_t* cand_in;
    const int32_t* cnt_in;
    int n_lists;
    uint64_t* cand_out;
    int32_t* cnt_out;
    int64_t bq;
    int64_t B;
    int kprime;
    int32_t* todo;
    int win;        // keys of the wave's LDS array (1 024 ... REDW_MAX): sized for the lists the seeded thresholds are EXPECTED to leave (c K' keys
                    // per query), so that more waves fit a CU - the kernel is a chain of load latencies (round 3: 4 -> 10 resident waves per CU at c4)
};

// NVMAX = keys per lane of the largest selection the launch can need (window / 64): 32 for windows up to 2 048 keys - the usual case,
// and the one that must not carry the 64-key variant's registers: compiled together the kernel took 256 VGPRs + 90 AGPRs = ONE wave per
// SIMD, and its ~24 us per query ran in five rounds (round 3: 120 -> 70 us at c4, 91 -> 57 at c5 with three waves per SIMD)
template <int NVMAX>
__global__ __launch_bounds__(64 * REDW_WAVES, NVMAX <= 32 ? 3 : 1) void reduce_lists_wave_kernel(const ReduceWaveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = int64_t(blockIdx.x) * REDW_WAVES + wave;
    if (b >= a.B) return;   // no workgroup barrier below
    uint64_t* lds = reinterpret_cast<uint64_t*>(smem_raw) + size_t(wave) * a.win;
    const int n_streams = 2 * a.n_lists;   // <= 64: lane s owns stream s = (list s / 2, half s % 2)
    const int craw = lane < n_streams ? a.cnt_in[(int64_t(lane >> 1) * a.bq + b) * 2 + (lane & 1)] : 0;
    const int n = craw & ~CNT_FLAGS;
    int flags = craw & CNT_FLAGS;
    flags = (__ballot((flags & CNT_UNDERFULL) != 0) ? CNT_UNDERFULL : 0) | (__ballot((flags & CNT_SEEDED_SHORT) != 0) ? CNT_SEEDED_SHORT : 0);   // OR over the lanes
    const int incl = wave_incl_scan_i32(n);
    const int M = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) a.todo[b] = 0;   // (round 3: every query is finished here - the workgroup rounds are no longer launched behind this kernel)
    const int excl = incl - n;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint64_t* dst = a.cand_out + (b * 2) * CAPH;   // streams 0 and 1 of the output list are contiguous
    // The query's keys in stream-major order form one sequence of M entries.  It is consumed in windows: the window's keys go to
    // the wave's LDS array behind the `kept` best keys of the windows before it (none in the first window), then the K' best of
    // the array are selected - into the output list after the last window, to the front of the array otherwise.  M <= REDW_MAX
    // (always, but for a sampled threshold that came out far too low: ~1e-4 of the queries at c = 8) is ONE window: the loop
    // body runs once, exactly the single-pass kernel of round 2.
    int kept = 0, n_c = 0;
    for (int w0 = 0; w0 < M || w0 == 0; ) {   // (wave-uniform)
        const int w1 = min(M, w0 + a.win - kept);
        // coalesced copy of every stream's part inside [w0, w1), eight streams' first 64 entries in flight at a time (seeded streams
        // are short: a stream-by-stream loop is a chain of dependent global-load latencies)
        for (int s0 = 0; s0 < n_streams; s0 += 8) {
            uint64_t v[8];
            int ns[8], os[8], fs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s2 = min(s0 + j, n_streams - 1);
                const int ex = __shfl(excl, s2, kWave), in = s0 + j < n_streams ? __shfl(incl, s2, kWave) : ex;
                const int lo = max(ex, w0), hi = min(in, w1);
                ns[j] = max(hi - lo, 0);       // entries of this stream inside the window
                fs[j] = lo - ex;               // ... starting at this entry of the stream
                os[j] = kept + lo - w0;        // ... going to this LDS slot
                const uint64_t* src = a.cand_in + ((int64_t(s2 >> 1) * a.bq + b) * 2 + (s2 & 1)) * CAPH + fs[j];
                v[j] = lane < ns[j] ? src[lane] : 0ull;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (lane < ns[j]) lds[os[j] + lane] = v[j];
                if (ns[j] > 64) {   // wave-uniform, rare: the rest of a long stream
                    const int s2 = s0 + j;
                    const uint64_t* src = a.cand_in + ((int64_t(s2 >> 1) * a.bq + b) * 2 + (s2 & 1)) * CAPH + fs[j];
                    for (int i = 64 + lane; i < ns[j]; i += 64) lds[os[j] + i] = src[i];
                }
            }
        }
        wave_lds_sync();
        const int n_now = kept + (w1 - w0);
        const bool last = w1 >= M;
        // the selection over the array's n_now keys with NV keys per lane: 16 when they fit (seeded lists at c = 4: ~1 000 keys per
        // query - a counting probe is then 16 compares per lane instead of 64), else 32 or 64
        auto select_window = [&](auto nv_tag) {
            constexpr int NV = decltype(nv_tag)::value;
            float sv[NV];
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int e = 64 * r + lane;
                const uint64_t kk = e < n_now ? lds[e] : 0ull;
                sv[r] = kk != 0ull ? key_score(kk) : -INFINITY;
            }
            if (n_now > a.kprime) {
                // exactly the K' best: every key above the K'-th largest score plus as many of the keys equal to it as fit
                int c_ge;
                const float t = wave_find_rank<NV>(sv, a.kprime, 0, c_ge);
                int c_gt = 0;
#pragma unroll
                for (int r = 0; r < NV; ++r) c_gt += __builtin_popcountll(__ballot(sv[r] > t));
                const int want_eq = min(c_ge - c_gt, a.kprime - c_gt);
                int eq_seen = 0, out = 0;
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const bool gt = sv[r] > t, eq = sv[r] == t;
                    const uint64_t be = __ballot(eq);
                    const bool keep = gt || (eq && eq_seen + __builtin_popcountll(be & lt_mask) < want_eq);
                    const uint64_t bk = __ballot(keep);
                    // (in-place compaction to the array's front, not-last windows: a kept key moves DOWN, to a slot whose own key was read in
                    // this or an earlier round - the read of round r precedes its write in program order)
                    const uint64_t kk = lds[64 * r + lane];
                    if (keep) {
                        if (last) dst[out + __builtin_popcountll(bk & lt_mask)] = kk;
                        else lds[out + __builtin_popcountll(bk & lt_mask)] = kk;
                    }
                    out += __builtin_popcountll(bk);
                    eq_seen += __builtin_popcountll(be);
                }
                n_c = out;   // == K'
            } else {
                n_c = n_now;
                if (last) {
#pragma unroll
                    for (int r = 0; r < NV; ++r) {
                        const int e = 64 * r + lane;
                        if (e < n_now) dst[e] = lds[e];
                    }
                }
            }
        };
        if (n_now <= 16 * 64) select_window(std::integral_constant<int, 16>{});
        else if (NVMAX <= 32 || n_now <= 32 * 64) select_window(std::integral_constant<int, 32>{});
        else select_window(std::integral_constant<int, NVMAX>{});
        if (last) break;
        wave_lds_sync();
        kept = n_c;
        w0 = w1;
    }
    if (lane == 0) {
        a.cnt_out[b * 2 + 0] = min(n_c, CAPH) | flags;
        a.cnt_out[b * 2 + 1] = max(n_c - CAPH, 0);
    }
}

// ---------------------------------------------------------------------------------------------------
// exhaustive float64 pass (exact_device_kernel below; the host-driven entry point feeds it query lists of 8)
// ---------------------------------------------------------------------------------------------------
constexpr int EX_CHUNK = 4096;

// ---------------------------------------------------------------------------------------------------
// On-device exhaustive pass: recomputes the queries listed in the workspace header (appended by the select kernels
// when a certificate failed) without a host round trip.  Grid (item chunks, EXACT_DEVICE_MAXQ): block (c, qi) computes
// the float64 distances of chunk c for listed query qi, sorts them and publishes its k best; the LAST block to arrive
// (ticket) merges the per-chunk winners and emits.  With an empty list every block exits at once.
// ---------------------------------------------------------------------------------------------------
struct ExactDevArgs {
    const float* X;
    int64_t n_items;
    int64_t ldx;
    const double* Q64;
    WsHeader* hdr;
    Ent* cand;       // [EXACT_DEVICE_MAXQ][n_chunks][k]
    int n_chunks;
    int chunk;       // items per stage-A block (power of two, k <= chunk <= EX_CHUNK)
    int chunks_per_group, n_groups;   // first-level merge groups (n_groups <= EXACT_DEVICE_GROUPS)
    Ent* cand2;      // [EXACT_DEVICE_MAXQ][n_groups][k]
    int32_t* missed_accum;   // nullable device counter: += listed queries beyond EXACT_DEVICE_MAXQ (left flagged 0)
    EmitArgs emit;
};

__global__ __launch_bounds__(256) void exact_device_kernel(const ExactDevArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int s_last;
    Ent* ent = reinterpret_cast<Ent*>(smem_raw);
    unsigned char* lds_emit = smem_raw + size_t(EX_CHUNK) * sizeof(Ent);
    const int qi = blockIdx.y, tid = threadIdx.x;
    const int count = a.hdr->uncert_count;
    if (blockIdx.x == 0 && qi == 0 && tid == 0 && count > EXACT_DEVICE_MAXQ && a.missed_accum) atomicAdd(a.missed_accum, count - EXACT_DEVICE_MAXQ);
    if (qi >= count) return;   // block-uniform
    const int64_t b = a.hdr->uncert_list[qi];
    const int d = a.emit.d, k = a.emit.k;
    const int64_t base = int64_t(blockIdx.x) * a.chunk;
    {
        // Stage A: float64 distances of the chunk's items in scikit-learn's sequential order.  Each wave takes rounds of 64
        // items: rows are gathered 64 B per 4 lanes (coalesced segments) and transposed through wave-private LDS, every
        // lane then walks ITS item - the same data path as the re-rank; the query comes in through scalar loads.
        const int lane = tid & 63, wave = tid >> 6;
        float* stage = reinterpret_cast<float*>(lds_emit) + wave * (64 * RR_RS);
        typedef const __attribute__((address_space(4))) double* cdouble_p;
        const int64_t bu = (int64_t(__builtin_amdgcn_readfirstlane(int(uint64_t(b) >> 32))) << 32) |
                           int64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(b)))));
        const cdouble_p q64 = (cdouble_p)(a.Q64 + bu * d);
        const int n_steps = (d + RR_CW - 1) / RR_CW;
        const int srow = lane >> 2, scol = 4 * (lane & 3);
        const int64_t last = a.n_items - 1;
#pragma unroll 1
        for (int r0 = wave * 64; r0 < a.chunk; r0 += 256) {
            const int64_t item = base + r0 + lane;
            const float* src0 = a.X + min(base + r0 + 0 + srow, last) * a.ldx + scol;
            const float* src1 = a.X + min(base + r0 + 16 + srow, last) * a.ldx + scol;
            const float* src2 = a.X + min(base + r0 + 32 + srow, last) * a.ldx + scol;
            const float* src3 = a.X + min(base + r0 + 48 + srow, last) * a.ldx + scol;
            float* dst = stage + srow * RR_RS + scol;
            float4 pre0 = *reinterpret_cast<const float4*>(src0), pre1 = *reinterpret_cast<const float4*>(src1);
            float4 pre2 = *reinterpret_cast<const float4*>(src2), pre3 = *reinterpret_cast<const float4*>(src3);
            double acc = 0.0;
#pragma unroll 1
            for (int c = 0; c < n_steps; ++c) {
                *reinterpret_cast<float4*>(dst) = pre0;
                *reinterpret_cast<float4*>(dst + 16 * RR_RS) = pre1;
                *reinterpret_cast<float4*>(dst + 32 * RR_RS) = pre2;
                *reinterpret_cast<float4*>(dst + 48 * RR_RS) = pre3;
                if (c + 1 < n_steps) {
                    const int o = RR_CW * (c + 1);
                    pre0 = *reinterpret_cast<const float4*>(src0 + o);
                    pre1 = *reinterpret_cast<const float4*>(src1 + o);
                    pre2 = *reinterpret_cast<const float4*>(src2 + o);
                    pre3 = *reinterpret_cast<const float4*>(src3 + o);
                }
                wave_lds_sync();
                float xv[RR_CW];
#pragma unroll
                for (int t = 0; t < RR_CW / 4; ++t) {
                    const float4 v = *reinterpret_cast<const float4*>(stage + lane * RR_RS + 4 * t);
                    xv[4 * t] = v.x; xv[4 * t + 1] = v.y; xv[4 * t + 2] = v.z; xv[4 * t + 3] = v.w;
                }
                wave_lds_sync();
                const int j0 = RR_CW * c;
#pragma unroll
                for (int t = 0; t < RR_CW; ++t) {
                    if (j0 + t < d) {   // uniform
                        const double tmp = __dsub_rn(q64[j0 + t], double(xv[t]));
                        acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                    }
                }
            }
            Ent e;
            e.hi = ~0ull;
            e.lo = ~0ull;
            if (item < a.n_items) {
                e.hi = f64_orderable(acc);
                e.lo = uint64_t(item);
            }
            ent[r0 + lane] = e;
        }
    }
    __syncthreads();
    bitonic_asc_ent(ent, a.chunk);
    // ---- two-level merge: the last block of every GROUP of chunks merges the group's winners, the last of those group
    // mergers merges the groups' winners.  (One flat merge by the last block costs ~2 ms on a 1.25 M-row catalogue.)
    // merge_rounds: n sorted-by-block entries at `src` -> the k smallest at src[0..k) and in ent[0..k), by this block.
    auto merge_rounds = [&](Ent* src, int64_t n) {
        while (n > k) {
            const int64_t chunks = cdiv64(n, EX_CHUNK);
            for (int64_t c = 0; c < chunks; ++c) {
                __syncthreads();
                for (int i = tid; i < EX_CHUNK; i += 256) {
                    Ent e;
                    e.hi = ~0ull;
                    e.lo = ~0ull;
                    if (c * EX_CHUNK + i < n) {   // written by other blocks: read through L2 (agent scope), never a stale L1 line
                        const Ent* p = &src[c * EX_CHUNK + i];
                        e.hi = __hip_atomic_load(&p->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        e.lo = __hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ent[i] = e;
                }
                bitonic_asc_ent(ent, EX_CHUNK);
                for (int i = tid; i < k; i += 256) src[c * k + i] = ent[i];   // c * k + k <= (c + 1) * EX_CHUNK: already consumed
                __threadfence();
            }
            n = chunks * k;
        }
        __syncthreads();
    };
    const int grp = blockIdx.x / a.chunks_per_group;
    const int g_first = grp * a.chunks_per_group;
    const int g_chunks = min(a.chunks_per_group, a.n_chunks - g_first);
    Ent* all = a.cand + int64_t(qi) * a.n_chunks * k;        // level 1: k entries per chunk
    Ent* all2 = a.cand2 + int64_t(qi) * a.n_groups * k;      // level 2: k entries per group
    for (int i = tid; i < k; i += 256) all[int64_t(blockIdx.x) * k + i] = ent[i];
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.hdr->group_tickets[qi][grp], 1) == g_chunks - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (g_chunks > 1) merge_rounds(all + int64_t(g_first) * k, int64_t(g_chunks) * k);   // else ent still holds this block's sorted chunk
    for (int i = tid; i < k; i += 256) all2[int64_t(grp) * k + i] = ent[i];
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&a.hdr->tickets[qi], 1) == a.n_groups - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (a.n_groups > 1) merge_rounds(all2, int64_t(a.n_groups) * k);
    __syncthreads();
    // ent[0..k) = the k nearest items (n_items >= k), ascending (dist2, item)
    const int kp = ((k + 63) / 64) * 64;
    uint32_t* c_item = reinterpret_cast<uint32_t*>(lds_emit + size_t(kp) * 16);   // `cand` of wave_rerank_emit
    for (int i = tid; i < k; i += 256) c_item[i] = uint32_t(ent[i].lo);
    __syncthreads();
    if (tid < 64) wave_rerank_emit(a.emit, b, lds_emit, kp, k, k, 2);
}

// host-driven entry (hwer_knn_exact_f64): up to 8 query ids arrive as kernel arguments and become the header's list
struct ExactIds { int32_t id[EXACT_DEVICE_MAXQ]; };
__global__ __launch_bounds__(256) void exact_list_init_kernel(WsHeader* hdr, int n, ExactIds ids) {
    hdr_reset(hdr, threadIdx.x, 256);
    __syncthreads();
    if (threadIdx.x < n) hdr->uncert_list[threadIdx.x] = ids.id[threadIdx.x];
    if (threadIdx.x == 0) hdr->uncert_count = n;
}

// ---------------------------------------------------------------------------------------------------
// X1 merge of per-shard results
// ---------------------------------------------------------------------------------------------------
struct MergeArgs {
    const int64_t* idx;
    const double* score;
    const double* dist2;
    int n_lists;
    int64_t B;
    int k;
    int sort_by_score;
    int64_t* out_idx;
    double* out_score;
    double* out_dist2;
    int64_t list_stride;   // elements between list l and l + 1 of the same array (B * kin when the lists are packed)
    int kin;               // entries per input list (<= k: shortened local lists, hwer_merge_topk_lists); n_lists * kin >= k
    int32_t* out_exhausted;   // nullable (B): 1 when some list gave ALL its kin (real) entries to the k winners - the shard behind it may
                              // hold more that belong (only meaningful when kin < k); 0 otherwise
};

// Stable order of the k winners by score, descending, by ONE wave: win_key[r] = ~orderable(score of the winner of rank r).
// Register bitonic sort on (32-bit approximation << 32 | rank) + exact fix-up of equal approximations; ord[p] = rank of the
// winner that goes to output position p.
template <int NE>
__device__ __forceinline__ void merge_score_order(const uint64_t* win_key, int k, uint32_t* ord) {
    const int lane = threadIdx.x & 63;
    uint64_t key[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int r = NE * lane + j;
        // padding has approximation 0xFFFFFFFF; real approximations are clamped below it (still monotone)
        key[j] = r < k ? (uint64_t(min(uint32_t(win_key[r] >> 32), 0xFFFFFFFEu)) << 32) | uint64_t(uint32_t(r)) : ~0ull;
    }
    wave_sort_u64<NE>(key);
    wave_sort_fixup<NE>(key, [&](uint64_t x, uint64_t y) {
        const uint32_t rx = uint32_t(x), ry = uint32_t(y);
        const uint64_t kx = win_key[rx], ky = win_key[ry];
        return kx < ky || (kx == ky && rx < ry);
    });
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int p = NE * lane + j;
        if (p < k) ord[p] = uint32_t(key[j]);
    }
}

// One workgroup per query.  Entries: hi = orderable(dist2), lo = global id << 22 | list << 9 | position in the list, sorted
// ascending by (hi, lo) = (dist2, id) with the source position as the last resort (padding rows share id -1).
//   1. T = max over the lists of their m-th entry, m = ceil(k / G): at least G m >= k entries are <= T, so nothing above T
//      can be among the k smallest.  Lists are sorted, so the survivors of a list are a prefix: each wave scans its lists in
//      64-entry pieces and stops at the first piece without a survivor (~k + 32 G entries read instead of G k).
//   2. n_lists <= 16: every survivor finds its final rank directly - own position + one binary search per other list's
//      survivor prefix (a sort of the survivors through LDS moves ~10x the bytes and was LDS-bandwidth-bound) - and, when
//      the reference's order is wanted, one wave orders the k winners by score in registers.
//      More lists: bitonic sort of the survivors in LDS, then of the k winners by score.
constexpr int MERGE_POS_BITS = 22, MERGE_J_BITS = 9;   // k <= 512, n_lists * k <= 8192
constexpr int MERGE_RANK_LISTS = 16;
__host__ __device__ inline size_t merge_lds_bytes(int n_lists, int k) {
    int n2 = 2;
    while (n2 < n_lists * k) n2 <<= 1;
    const size_t sort_path = size_t(n2 > 512 ? n2 : 512) * sizeof(Ent);
    // rank path: entries at their list position | winners' score keys | their source positions | output order
    const size_t rank_path = size_t(n_lists) * k * sizeof(Ent) + size_t(512) * (8 + 4 + 4);
    return n_lists <= MERGE_RANK_LISTS ? rank_path : sort_path;
}
__global__ __launch_bounds__(256) void merge_topk_kernel(const MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Ent* ent = reinterpret_cast<Ent*>(smem_raw);
    __shared__ Ent s_wave_max[4];
    __shared__ int s_count;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k, kin = a.kin, G = a.n_lists;   // kin entries per list in, k winners out
    const int64_t row = b * k, row_in = b * kin;
    auto load_ent = [&](int l, int j) {
        const int64_t src = int64_t(l) * a.list_stride + row_in + j;
        Ent e;
        e.hi = f64_orderable(a.dist2[src]);
        e.lo = (uint64_t(a.idx[src]) << MERGE_POS_BITS) | uint64_t(uint32_t((l << MERGE_J_BITS) | j));
        return e;
    };
    // ---- 1. bound T ---------------------------------------------------------------------------------------------------
    const int m = (k + G - 1) / G;
    Ent t{0ull, 0ull};
    for (int l = tid; l < G; l += 256) {
        const Ent e = load_ent(l, m - 1);
        if (ent_less(t, e)) t = e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Ent o;
        o.hi = __shfl_xor((unsigned long long)t.hi, off, kWave);
        o.lo = __shfl_xor((unsigned long long)t.lo, off, kWave);
        if (ent_less(t, o)) t = o;
    }
    if (tid == 0) s_count = 0;
    if (lane == 0) s_wave_max[wave] = t;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (ent_less(t, s_wave_max[w])) t = s_wave_max[w];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t pos_mask = (1u << MERGE_POS_BITS) - 1u;
    auto src_of_pos = [&](uint32_t pos) {
        return int64_t(pos >> MERGE_J_BITS) * a.list_stride + row_in + int64_t(pos & ((1u << MERGE_J_BITS) - 1u));
    };
    if (G <= MERGE_RANK_LISTS) {
        __shared__ int s_cnt[MERGE_RANK_LISTS];
        __shared__ int s_win[MERGE_RANK_LISTS];   // winners per list (out_exhausted)
        if (tid < MERGE_RANK_LISTS) s_win[tid] = 0;
        uint64_t* win_key = reinterpret_cast<uint64_t*>(ent + size_t(G) * kin);
        uint32_t* win_pos = reinterpret_cast<uint32_t*>(win_key + 512);
        uint32_t* ord = win_pos + 512;
        // ---- 2a. survivors stay at their list position; s_cnt[l] = length of list l's surviving prefix ---------------------
        for (int l = wave; l < G; l += 4) {
            int c = 0;
            for (int j0 = 0; j0 < kin; j0 += 64) {
                const int j = j0 + lane;
                bool keep = false;
                Ent e{};
                if (j < kin) {
                    e = load_ent(l, j);
                    keep = !ent_less(t, e);
                }
                const uint64_t bal = __ballot(keep);
                if (keep) ent[l * kin + j] = e;
                c += __builtin_popcountll(bal);
                if (bal != ~0ull) break;   // sorted list: the survivors are a prefix, and it ended inside this piece
            }
            if (lane == 0) s_cnt[l] = c;
        }
        __syncthreads();
        int n_s = 0;
        for (int l = 0; l < G; ++l) n_s += s_cnt[l];
        // ---- 2b. final rank of every survivor ---------------------------------------------------------------------------------
        for (int i = tid; i < n_s; i += 256) {
            int l = 0, j = i;
            for (; l < G - 1; ++l) {
                const int c = s_cnt[l];
                if (j < c) break;
                j -= c;
            }
            const Ent e = ent[l * kin + j];
            int rank = j;
            for (int l2 = 0; l2 < G && rank < k; ++l2) {
                if (l2 == l) continue;
                const Ent* lst = ent + l2 * kin;
                int lo = 0, hi = s_cnt[l2];
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ent_less(lst[mid], e)) lo = mid + 1;
                    else hi = mid;
                }
                rank += lo;
            }
            if (rank < k) {
                const uint32_t pos = uint32_t(e.lo) & pos_mask;
                const int64_t src = src_of_pos(pos);
                if (a.out_exhausted && a.idx[src] >= 0) atomicAdd(&s_win[l], 1);   // (padding rows carry id -1)
                if (a.sort_by_score) {
                    win_key[rank] = ~f64_orderable(a.score[src]);
                    win_pos[rank] = pos;
                } else {
                    a.out_idx[row + rank] = a.idx[src];
                    a.out_score[row + rank] = a.score[src];
                    a.out_dist2[row + rank] = a.dist2[src];
                }
            }
        }
        if (a.out_exhausted) {   // a list whose kin entries ALL made it: its shard may hold further members (kin < k only)
            __syncthreads();
            if (tid == 0) {
                int ex = 0;
                for (int l = 0; l < G; ++l) ex |= (kin < k && s_win[l] >= kin) ? 1 : 0;
                a.out_exhausted[b] = ex;
            }
        }
        if (!a.sort_by_score) return;
        __syncthreads();
        // ---- 3a. the reference's final order: stable sort by score, descending (one wave, registers) -------------------------
        if (wave == 0) {
            if (k <= 128) merge_score_order<2>(win_key, k, ord);
            else if (k <= 256) merge_score_order<4>(win_key, k, ord);
            else merge_score_order<8>(win_key, k, ord);
        }
        __syncthreads();
        for (int i = tid; i < k; i += 256) {
            const int64_t src = src_of_pos(win_pos[ord[i]]);
            a.out_idx[row + i] = a.idx[src];
            a.out_score[row + i] = a.score[src];
            a.out_dist2[row + i] = a.dist2[src];
        }
        return;
    }
    // ---- 2. survivors -> LDS ------------------------------------------------------------------------------------------
    for (int l = wave; l < G; l += 4) {
        for (int j0 = 0; j0 < kin; j0 += 64) {
            const int j = j0 + lane;
            bool keep = false;
            Ent e{};
            if (j < kin) {
                e = load_ent(l, j);
                keep = !ent_less(t, e);
            }
            const uint64_t bal = __ballot(keep);
            if (bal == 0ull) break;   // sorted list: no later entry can be <= T
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_count, __builtin_popcountll(bal));
            base = __shfl(base, 0, kWave);
            if (keep) ent[base + __builtin_popcountll(bal & lt_mask)] = e;
        }
    }
    __syncthreads();
    const int n_s = s_count;                      // k <= n_s <= G k
    const int n2 = pow2ceil_i(max(n_s, 2));
    for (int i = n_s + tid; i < n2; i += 256) ent[i] = Ent{~0ull, ~0ull};
    bitonic_asc_ent(ent, n2);
    auto src_of = [&](const Ent& e) { return src_of_pos(uint32_t(e.lo) & pos_mask); };
    // ---- 3. the reference's final order: stable sort by score, descending ---------------------------------------------------
    if (a.sort_by_score) {
        const int k2 = pow2ceil_i(max(k, 2));
        auto make = [&](int i) {
            Ent e;
            if (i < k) {
                e.hi = ~f64_orderable(a.score[src_of(ent[i])]);
                e.lo = (uint64_t(uint32_t(i)) << 32) | (ent[i].lo & ((1ull << MERGE_POS_BITS) - 1ull));
            } else {
                e.hi = ~0ull;
                e.lo = ~0ull;
            }
            return e;
        };
        const int i0 = tid, i1 = tid + 256;  // k2 <= 512, 256 threads
        Ent e0{}, e1{};
        if (i0 < k2) e0 = make(i0);
        if (i1 < k2) e1 = make(i1);
        __syncthreads();
        if (i0 < k2) ent[i0] = e0;
        if (i1 < k2) ent[i1] = e1;
        bitonic_asc_ent(ent, k2);
    }
    for (int i = tid; i < k; i += 256) {
        const int64_t src = src_of(ent[i]);
        a.out_idx[row + i] = a.idx[src];
        a.out_score[row + i] = a.score[src];
        a.out_dist2[row + i] = a.dist2[src];
    }
}

// ---------------------------------------------------------------------------------------------------
// All float64 squared distances of a few queries (the k > K_MAX path of the host mirror: distances here, selection by
// a device sort).  Grid (row blocks, queries); same data path as stage A of the exhaustive pass: rows gathered 64 B per
// 4 lanes, transposed through wave-private LDS, every lane walks ITS row in scikit-learn's sequential order.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void all_dist2_kernel(const float* __restrict__ X, int64_t n_items, int d, int64_t ldx,
                                                        const double* __restrict__ Q64, double* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float stage_all[4 * 64 * RR_RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* stage = stage_all + wave * (64 * RR_RS);
    typedef const __attribute__((address_space(4))) double* cdouble_p;
    const cdouble_p q64 = (cdouble_p)(Q64 + int64_t(blockIdx.y) * d);
    double* orow = out + int64_t(blockIdx.y) * n_items;
    const int n_steps = (d + RR_CW - 1) / RR_CW;
    const int srow = lane >> 2, scol = 4 * (lane & 3);
    const int64_t last = n_items - 1;
    for (int64_t r0 = (int64_t(blockIdx.x) * 4 + wave) * 64; r0 < n_items; r0 += int64_t(gridDim.x) * 256) {
        const float* src0 = X + min(r0 + 0 + srow, last) * ldx + scol;
        const float* src1 = X + min(r0 + 16 + srow, last) * ldx + scol;
        const float* src2 = X + min(r0 + 32 + srow, last) * ldx + scol;
        const float* src3 = X + min(r0 + 48 + srow, last) * ldx + scol;
        float* dst = stage + srow * RR_RS + scol;
        float4 pre0 = *reinterpret_cast<const float4*>(src0), pre1 = *reinterpret_cast<const float4*>(src1);
        float4 pre2 = *reinterpret_cast<const float4*>(src2), pre3 = *reinterpret_cast<const float4*>(src3);
        double acc = 0.0;
#pragma unroll 1
        for (int c = 0; c < n_steps; ++c) {
            *reinterpret_cast<float4*>(dst) = pre0;
            *reinterpret_cast<float4*>(dst + 16 * RR_RS) = pre1;
            *reinterpret_cast<float4*>(dst + 32 * RR_RS) = pre2;
            *reinterpret_cast<float4*>(dst + 48 * RR_RS) = pre3;
            if (c + 1 < n_steps) {
                const int o = RR_CW * (c + 1);
                pre0 = *reinterpret_cast<const float4*>(src0 + o);
                pre1 = *reinterpret_cast<const float4*>(src1 + o);
                pre2 = *reinterpret_cast<const float4*>(src2 + o);
                pre3 = *reinterpret_cast<const float4*>(src3 + o);
            }
            wave_lds_sync();
            float xv[RR_CW];
#pragma unroll
            for (int t = 0; t < RR_CW / 4; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(stage + lane * RR_RS + 4 * t);
                xv[4 * t] = v.x; xv[4 * t + 1] = v.y; xv[4 * t + 2] = v.z; xv[4 * t + 3] = v.w;
            }
            wave_lds_sync();
            const int j0 = RR_CW * c;
#pragma unroll
            for (int t = 0; t < RR_CW; ++t) {
                if (j0 + t < d) {   // uniform
                    const double tmp = __dsub_rn(q64[j0 + t], double(xv[t]));
                    acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                }
            }
        }
        if (r0 + lane < n_items) orow[r0 + lane] = acc;
    }
}

static int set_max_lds(const void* fn, size_t bytes, const char* what) {
    if (bytes <= 64 * 1024) return HWER_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
    return HWER_OK;
}

}  // namespace hwer

using namespace hwer;

// ---- optional per-kernel timing with HIP events recorded on the launch stream (bench.py roofline) -----------
#include <vector>
namespace {
struct ProfSlot {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
};
int g_prof_period = 0;      // 0 = off, n = bracket every n-th hwer_knn_search_f32 call
long g_prof_calls = 0;
bool g_prof_on = false;    // decision for the current call
ProfSlot g_prof[2];  // 0: knn_screen, 1: knn_select (+ reduce)
void prof_next_call() { g_prof_on = g_prof_period > 0 && (g_prof_calls++ % g_prof_period) == 0; }

struct ProfScope {
    int which;
    hipStream_t st;
    bool on;
    ProfScope(int w, hipStream_t s) : which(w), st(s), on(g_prof_on) {
        if (!on) return;
        ProfSlot& p = g_prof[which];
        if (p.used == p.start.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }   // no timing for this call
            p.start.push_back(a);
            p.stop.push_back(b);
        }
        if (hipEventRecord(p.start[p.used], st) != hipSuccess) on = false;
    }
    ~ProfScope() {
        if (!on) return;
        ProfSlot& p = g_prof[which];
        if (hipEventRecord(p.stop[p.used], st) == hipSuccess) ++p.used;
    }
};
}  // namespace

extern "C" int hwer_profile_enable(int on) {
    g_prof_period = on > 0 ? on : 0;
    g_prof_calls = 0;
    g_prof_on = false;
    return HWER_OK;
}
// Sum of the event-bracketed durations since the last read.  Synchronises on the recorded events (call it outside
// the timed region).
extern "C" int hwer_profile_read(int which, double* total_ms, int64_t* launches) {
    if (which < 0 || which > 1 || !total_ms || !launches) return fail(HWER_EINVAL, "profile_read: bad argument");
    ProfSlot& p = g_prof[which];
    double sum = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i < p.used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(p.stop[i]) != hipSuccess || hipEventElapsedTime(&ms, p.start[i], p.stop[i]) != hipSuccess) continue;
        sum += ms;
        ++n;
    }
    *total_ms = sum;
    *launches = n;
    p.used = 0;
    return HWER_OK;
}

// Diagnostic builds (HWER_DIAG_STAMPS) with HWER_SELECT_STAMPS=1: a device buffer of 8 cycle stamps per query; the previous
// launch's stamps are summarised on stderr when the next search starts.
static unsigned long long* sel_stamps_buffer(int64_t B) {
#ifdef HWER_DIAG_STAMPS
    static unsigned long long* buf = nullptr;
    static int64_t buf_B = 0;
    if (!getenv("HWER_SELECT_STAMPS")) return nullptr;
    if (buf) {
        std::vector<unsigned long long> h(size_t(buf_B) * 16);
        (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[6] = {0, 0, 0, 0, 0, 0}, tail[4] = {0, 0, 0, 0};
        int64_t n = 0, nt = 0;
        for (int64_t q = 0; q < buf_B; ++q) {
            const unsigned long long* t = &h[size_t(q) * 16];
            if (t[6] <= t[0] || t[0] == 0) continue;
            for (int c = 0; c < 6; ++c) sum[c] += double(t[c + 1] - t[c]);
            if (t[7] && t[8] && t[9]) { tail[0] += double(t[7] - t[4]); tail[1] += double(t[8] - t[7]); tail[2] += double(t[9] - t[8]); tail[3] += double(t[5] - t[9]); ++nt; }
            ++n;
        }
        {   // when do the workgroups of the launch start and end? (100-MHz clock)
            std::vector<double> st, en;
            for (int64_t q = 0; q < buf_B; ++q) {
                const unsigned long long* t = &h[size_t(q) * 16];
                if (t[10] && t[11] > t[10]) { st.push_back(double(t[10])); en.push_back(double(t[11])); }
            }
            if (!st.empty()) {
                const double t0 = *std::min_element(st.begin(), st.end());
                std::vector<double> s2(st), e2(en);
                std::sort(s2.begin(), s2.end());
                std::sort(e2.begin(), e2.end());
                auto pc = [&](const std::vector<double>& v, double f) { return (v[size_t(f * double(v.size() - 1))] - t0) * 0.01; };
                double life = 0;
                for (size_t i = 0; i < st.size(); ++i) life += en[i] - st[i];
                fprintf(stderr, "[select stamps]   launch timeline, us after the first start: starts p50 %.2f p90 %.2f max %.2f | ends p10 %.2f p50 %.2f p90 %.2f max %.2f | mean lifetime %.2f us\n",
                        pc(s2, 0.5), pc(s2, 0.9), pc(s2, 1.0), pc(e2, 0.1), pc(e2, 0.5), pc(e2, 0.9), pc(e2, 1.0), life / double(st.size()) * 0.01);
            }
        }
        {   // spread of every phase over the queries
            static const char* names[6] = {"load+minmax", "rank-search", "collect", "rerank-rounds", "sorts", "emit"};
            std::string line = "[select stamps]   phase cycles p10 / p50 / p90 / max:";
            for (int c = 0; c < 6; ++c) {
                std::vector<double> v;
                for (int64_t q = 0; q < buf_B; ++q) {
                    const unsigned long long* t = &h[size_t(q) * 16];
                    if (t[6] <= t[0] || t[0] == 0) continue;
                    v.push_back(double(t[c + 1] - t[c]));
                }
                if (v.empty()) break;
                std::sort(v.begin(), v.end());
                char b[160];
                snprintf(b, sizeof b, " %s %.0f / %.0f / %.0f / %.0f;", names[c], v[v.size() / 10], v[v.size() / 2], v[(v.size() * 9) / 10], v.back());
                line += b;
            }
            fprintf(stderr, "%s\n", line.c_str());
        }
        {   // one-query-per-workgroup launches: inside "rerank-rounds" (wave 0's clock)
            double r[5] = {0, 0, 0, 0, 0}, rr[4] = {0, 0, 0, 0};
            int64_t nr = 0;
            for (int64_t q = 0; q < buf_B; ++q) {
                const unsigned long long* t = &h[size_t(q) * 16];
                if (!(t[12] && t[13] && t[14] && t[15]) || t[6] <= t[0]) continue;
                if (t[7] && t[8] && t[9]) { rr[0] += double(t[7] - t[3]); rr[1] += double(t[8] - t[7]); rr[2] += double(t[9] - t[8]); rr[3] += double(t[12] - t[9]); }
                r[0] += double(t[12] - t[3]); r[1] += double(t[13] - t[12]); r[2] += double(t[14] - t[13]); r[3] += double(t[15] - t[14]); r[4] += double(t[4] - t[15]);
                ++nr;
            }
            if (nr) fprintf(stderr, "[select stamps]   inside rerank-rounds (%lld queries): wave 0's rounds=%.0f wait for the other waves=%.0f boundary rank=%.0f counting order=%.0f rest=%.0f\n",
                            (long long)nr, r[0] / nr, r[1] / nr, r[2] / nr, r[3] / nr, r[4] / nr);
            if (nr && rr[1] > 0) fprintf(stderr, "[select stamps]   inside wave 0's rounds: set-up + first gathers issued=%.0f until the first chunk is staged=%.0f rest of the chunk loop=%.0f score + table writes=%.0f\n",
                                         rr[0] / nr, rr[1] / nr, rr[2] / nr, rr[3] / nr);
        }
        if (nt) fprintf(stderr, "[select stamps]   tail of %lld lazy queries: boundary-rank=%.0f keys=%.0f sort=%.0f fixup=%.0f\n", (long long)nt, tail[0] / nt, tail[1] / nt, tail[2] / nt, tail[3] / nt);
        if (n) fprintf(stderr, "[select stamps] queries=%lld cycles/query: load+minmax=%.0f rank-search=%.0f collect=%.0f rerank-rounds=%.0f sorts=%.0f emit=%.0f total=%.0f\n",
                       (long long)n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / n);
        (void)hipFree(buf);
        buf = nullptr;
    }
    if (hipMalloc(&buf, size_t(B) * 128) != hipSuccess || hipMemset(buf, 0, size_t(B) * 128) != hipSuccess) buf = nullptr;
    buf_B = B;
    return buf;
#else
    (void)B;
    return nullptr;
#endif
}

// workspace layout: [cand A][cnt A][cand B][cnt B]   (B only when lists must be pre-reduced)
static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

struct WsLayout {
    size_t qf, cand_a, cnt_a, cand_b, cnt_b, seed_keys, seed_tau, seed_cnt, todo, total;
};
static WsLayout ws_layout(const ScreenPlan& p, int64_t B) {
    WsLayout w{};
    size_t off = WS_HEADER_BYTES;   // WsHeader first
    w.qf = off;                     // queries in fragment order
    off += align256(repacked_queries_bytes(p.bq, p.ng * 8));
    if (p.dense) {   // one (B, ld_dense) float matrix of keys
        w.cand_a = off;
        w.total = off + align256(size_t(p.bq) * size_t(p.ld_dense) * 4);   // bq >= B rows: the screen stores unconditionally
        return w;
    }
    w.cand_a = off;
    off += align256(size_t(p.n_lists) * p.bq * 2 * CAPH * 8);
    w.cnt_a = off;
    off += align256(size_t(p.n_lists) * p.bq * 2 * 4);
    int groups = p.n_lists > 1 ? int(cdiv64(p.n_lists, MAX_LISTS_DIRECT)) : 0;
    w.cand_b = off;
    off += align256(size_t(groups) * p.bq * 2 * CAPH * 8);
    w.cnt_b = off;
    off += align256(size_t(groups) * p.bq * 2 * 4);
    w.todo = off;   // per query: lists left to the workgroup reduce kernels
    off += align256(size_t(p.bq) * 4);
    if (p.seed) {   // seed pre-pass: sample keys (bq, seed_ld), thresholds, a scratch counter row
        w.seed_keys = off;
        off += align256(seed_keys_bytes(p));   // float16 sample keys, or the lanes' best candidates (float32)
        w.seed_tau = off;
        off += align256(size_t(p.bq) * 4);
        w.seed_cnt = off;
        off += align256(size_t(p.bq) * 4 * COUNT_STRIDE);   // (the threshold kernels zero a strided counter per query: knn_small.hip)
    }
    w.total = off;
    return w;
}

// Up to this many queries a launch gives every query a whole workgroup (COOP): with fewer queries than ~2 waves per SIMD the
// kernel's time is the serial chain of one wave, which the four waves of a workgroup split.
#ifndef HWER_COOP_MAX_QUERIES
#define HWER_COOP_MAX_QUERIES 2048
#endif
constexpr int64_t COOP_MAX_QUERIES_DEFAULT = HWER_COOP_MAX_QUERIES;
static int64_t coop_max_queries() {   // (diagnostic builds: HWER_COOP_MAX overrides)
    const char* e = diag_env("HWER_COOP_MAX");
    return e ? atoll(e) : COOP_MAX_QUERIES_DEFAULT;
}

static int launch_select_list(const DenseSelectArgs& da, int d, int score_mode, hipStream_t st) {
    static_assert(SMALL_LIST_CAP == 256 * 16, "the list select keeps 16 float4-equivalents of the list per lane");
    const bool coop = da.B <= coop_max_queries();
    const size_t lds = coop ? coop_lds_bytes(d, da.kp) : size_t(4) * wave_lds_bytes(d, da.kp);
    auto launch = [&](auto md) -> int {
        constexpr int MD = decltype(md)::value;
        if (coop) {
            if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<16, MD, true, true>), lds, "knn_select_list")) return e;
            hipLaunchKernelGGL((select_dense_kernel<16, MD, true, true>), dim3(unsigned(da.B)), dim3(256), lds, st, da);
            return HWER_OK;
        }
        if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<16, MD, true>), lds, "knn_select_list")) return e;
        hipLaunchKernelGGL((select_dense_kernel<16, MD, true>), dim3(unsigned(cdiv64(da.B, 4))), dim3(256), lds, st, da);
        return HWER_OK;
    };
    int rc;
    switch (score_mode) {
        case 0: rc = launch(std::integral_constant<int, 0>{}); break;
        case 1: rc = launch(std::integral_constant<int, 1>{}); break;
        default: rc = launch(std::integral_constant<int, 2>{}); break;
    }
    if (rc) return rc;
    return check_launch("knn_select_list");
}

static int launch_select_rerank(const SelectArgs& sa, int d, int score_mode, hipStream_t st) {
    const size_t lds = size_t(4) * wave_lds_bytes(d, sa.kcap);
    auto launch_select = [&](auto md) -> int {
        constexpr int MD = decltype(md)::value;
        if (int e = set_max_lds(reinterpret_cast<const void*>(&select_rerank_kernel<MD>), lds, "knn_select")) return e;
        hipLaunchKernelGGL(select_rerank_kernel<MD>, dim3(unsigned(cdiv64(sa.B, 4))), dim3(256), lds, st, sa);
        return HWER_OK;
    };
    int rc;
    switch (score_mode) {
        case 0: rc = launch_select(std::integral_constant<int, 0>{}); break;
        case 1: rc = launch_select(std::integral_constant<int, 1>{}); break;
        default: rc = launch_select(std::integral_constant<int, 2>{}); break;
    }
    if (rc) return rc;
    return check_launch("knn_select");
}

// ---- self-test of the wave primitives (hwer_selftest_wave_ops) -------------------------------------------------------------------------
namespace hwer {
__global__ __launch_bounds__(64) void selftest_wave_ops_kernel(const int* in_i, const uint64_t* in_k, const float* in_f, const double* in_d, int* out_scan,
                                                               uint64_t* out_sorted, int* out_lb, float* out_mm, double* out_sum) {
    __shared__ uint64_t run[64];
    const int w = blockIdx.x, lane = threadIdx.x, i = w * 64 + lane;
    out_scan[i] = wave_incl_scan_i32(in_i[i]);
    uint64_t key = in_k[i];
    wave_sort1_u64(key);
    out_sorted[i] = key;
    run[lane] = key;
    __syncthreads();
    out_lb[i] = run64_count_less(run, in_k[(w ^ 1) * 64 + lane]);      // lower bounds of the NEIGHBOUR wave's keys in this wave's sorted run
    out_mm[2 * i] = wave_max_f32(in_f[i]);
    out_mm[2 * i + 1] = wave_min_f32(in_f[i]);
    out_sum[i] = wave_sum_f64(in_d[i]);
}
}  // namespace hwer

extern "C" {

int hwer_selftest_wave_ops(int n_waves, unsigned seed) {
    if (n_waves < 2 || (n_waves & 1) || n_waves > 4096) return fail(HWER_EINVAL, "selftest_wave_ops: n_waves must be even, 2 .. 4096");
    const size_t n = size_t(n_waves) * 64;
    std::vector<int> hi(n), sc(n), lb(n);
    std::vector<uint64_t> hk(n), so(n);
    std::vector<float> hf(n), mm(2 * n);
    std::vector<double> hd(n), su(n);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (size_t i = 0; i < n; ++i) {
        hi[i] = int(rnd() % 2001) - 1000;
        hk[i] = rnd();
        hf[i] = float(int(rnd() % 200001) - 100000) * 1e-3f;
        hd[i] = double(int64_t(rnd() % 2000001) - 1000000) * 1e-7;
    }
    for (size_t i = 0; i < n; ++i) hk[i] = (hk[i] & ~uint64_t(127)) | uint64_t(i & 127);      // distinct keys inside every pair of waves (128 lanes)
    void *d_i, *d_k, *d_f, *d_d, *o_sc, *o_so, *o_lb, *o_mm, *o_su;
    bool ok = hipMalloc(&d_i, n * 4) == hipSuccess && hipMalloc(&d_k, n * 8) == hipSuccess && hipMalloc(&d_f, n * 4) == hipSuccess && hipMalloc(&d_d, n * 8) == hipSuccess &&
              hipMalloc(&o_sc, n * 4) == hipSuccess && hipMalloc(&o_so, n * 8) == hipSuccess && hipMalloc(&o_lb, n * 4) == hipSuccess && hipMalloc(&o_mm, n * 8) == hipSuccess &&
              hipMalloc(&o_su, n * 8) == hipSuccess;
    if (!ok) return fail(HWER_ELAUNCH, "selftest_wave_ops: hipMalloc failed");
    (void)hipMemcpy(d_i, hi.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_k, hk.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_f, hf.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_d, hd.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(selftest_wave_ops_kernel, dim3(unsigned(n_waves)), dim3(64), 0, nullptr, static_cast<const int*>(d_i), static_cast<const uint64_t*>(d_k),
                       static_cast<const float*>(d_f), static_cast<const double*>(d_d), static_cast<int*>(o_sc), static_cast<uint64_t*>(o_so), static_cast<int*>(o_lb),
                       static_cast<float*>(o_mm), static_cast<double*>(o_su));
    int rc = check_launch("selftest_wave_ops");
    if (rc == HWER_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(HWER_ELAUNCH, "selftest_wave_ops: kernel failed");
    if (rc == HWER_OK) {
        (void)hipMemcpy(sc.data(), o_sc, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(so.data(), o_so, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(lb.data(), o_lb, n * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(mm.data(), o_mm, n * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(su.data(), o_su, n * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < n_waves && rc == HWER_OK; ++w) {
            const size_t b = size_t(w) * 64;
            std::vector<uint64_t> ref(hk.begin() + b, hk.begin() + b + 64);
            std::sort(ref.begin(), ref.end());
            int run_sum = 0;
            float fmx = hf[b], fmn = hf[b];
            for (int l = 0; l < 64; ++l) { fmx = std::max(fmx, hf[b + l]); fmn = std::min(fmn, hf[b + l]); }
            // the butterfly order of the float64 sum: pairs at distance 1, 2, 4, ... (every lane ends with the same value)
            double bf[64];
            for (int l = 0; l < 64; ++l) bf[l] = hd[b + l];
            for (int off = 1; off < 64; off <<= 1) {
                double nx[64];
                for (int l = 0; l < 64; ++l) nx[l] = bf[l] + bf[l ^ off];
                for (int l = 0; l < 64; ++l) bf[l] = nx[l];
            }
            for (int l = 0; l < 64 && rc == HWER_OK; ++l) {
                run_sum += hi[b + l];
                const uint64_t other = hk[size_t(w ^ 1) * 64 + l];
                const int want_lb = int(std::lower_bound(ref.begin(), ref.end(), other) - ref.begin());
                if (sc[b + l] != run_sum) rc = fail(HWER_EINVAL, "selftest_wave_ops: scan wave %d lane %d: %d != %d", w, l, sc[b + l], run_sum);
                else if (so[b + l] != ref[l]) rc = fail(HWER_EINVAL, "selftest_wave_ops: sort wave %d lane %d", w, l);
                else if (lb[b + l] != want_lb) rc = fail(HWER_EINVAL, "selftest_wave_ops: lower bound wave %d lane %d: %d != %d", w, l, lb[b + l], want_lb);
                else if (mm[2 * (b + l)] != fmx || mm[2 * (b + l) + 1] != fmn) rc = fail(HWER_EINVAL, "selftest_wave_ops: max / min wave %d lane %d", w, l);
                else if (su[b + l] != bf[l]) rc = fail(HWER_EINVAL, "selftest_wave_ops: float64 sum wave %d lane %d: %.17g != %.17g", w, l, su[b + l], bf[l]);
            }
        }
    }
    for (void* p_ : {d_i, d_k, d_f, d_d, o_sc, o_so, o_lb, o_mm, o_su}) (void)hipFree(p_);
    return rc;
}


int hwer_knn_max_k(void) { return K_MAX; }
int hwer_knn_max_dim(void) { return 256; }

int hwer_knn_all_dist2_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, int64_t nq, double* out,
                           void* stream) {
    if (nq == 0) return HWER_OK;
    if (!X || !Q64 || !out) return fail(HWER_EINVAL, "knn_all_dist2: null pointer");
    if (n_items < 1 || d < 1 || d > 1024 || ldx < ((d + 15) / 16) * 16 || nq < 0 || nq > 65535)
        return fail(HWER_EINVAL, "knn_all_dist2: need n_items >= 1, 1 <= d <= 1024, ldx >= roundup16(d), nq <= 65535");
    const unsigned gx = unsigned(std::min<int64_t>(cdiv64(n_items, 256), 4096));
    hipLaunchKernelGGL(all_dist2_kernel, dim3(gx, unsigned(nq)), dim3(256), 0, as_stream(stream), X, n_items, d, ldx, Q64, out);
    return check_launch("knn_all_dist2");
}

int hwer_knn_f16_scale_exp(float hnorm_max) { return f16_scale_exp(hnorm_max); }
int hwer_knn_f16_screen_pays(int64_t B, int64_t n_items, int d) {
    if (B < 1 || n_items < 1 || d < 1 || hwer_knn_padded_dim(d) < 64) return 0;   // the float16 MFMA tiles start at 64 columns
    if (n_items > DENSE_MAX_ITEMS) return 1;
    return double(B) * double(n_items) * double(d) >= 1e9 ? 1 : 0;
}

int hwer_pack_rows_f16(const float* X, int64_t n_rows_padded, int64_t ldx, int x_exp, void* X16, void* stream) {
    if (!X || !X16) return fail(HWER_EINVAL, "pack_rows_f16: null pointer");
    if (n_rows_padded < 0 || n_rows_padded % 32 != 0 || hwer_knn_padded_dim(int(ldx)) != ldx)
        return fail(HWER_EINVAL, "pack_rows_f16: expects the padded table of hwer_pack_rows_f32 (rows % 32 == 0, ldx in {32,64,128,256})");
    return pack_rows16(X, n_rows_padded, ldx, x_exp, X16, as_stream(stream));
}

size_t hwer_knn_workspace_bytes_opts(int64_t B, int64_t n_items, int d, int k, uint32_t options) {
    const int ldx = hwer_knn_padded_dim(d);
    if (ldx < 0 || B < 1 || n_items < 1 || k < 1 || k > K_MAX) return 0;
    const bool conservative = (options & HWER_KNN_EXACT_THRESHOLDS) != 0;
    if (!conservative && use_small_path(B, n_items, options)) {
        const ScreenPlan sp = make_screen_plan(B, n_items, ldx, k, options);
        return WS_HEADER_BYTES + small_ws_bytes(make_small_plan(B, n_items, ldx, k), (sp.seed && sp.seed_frac <= 0.01) ? &sp : nullptr);
    }
    return ws_layout(make_screen_plan(B, n_items, ldx, k, options), B).total;
}
size_t hwer_knn_workspace_bytes(int64_t B, int64_t n_items, int d, int k) { return hwer_knn_workspace_bytes_opts(B, n_items, d, k, 0); }
int64_t hwer_knn_query_padding(int64_t B, int64_t n_items, int d, int k, uint32_t options) {
    const int ldx = hwer_knn_padded_dim(d);
    if (ldx < 0 || B < 1 || n_items < 1 || k < 1 || k > K_MAX) return 0;
    return make_screen_plan(B, n_items, ldx, k, options).bq;
}

int hwer_knn_search_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                        const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k, int score_mode,
                        int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                        double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return hwer_knn_search_opts_f32(X, hnorm, n_items, d, ldx, Q32, Q64, A32, lda, a_idx, B, k, score_mode, sort_by_score, row_ids,
                                    id_base, hnorm_max, out_idx, out_score, out_dist2, out_flags, workspace, workspace_bytes, 0, nullptr, 0, stream);
}

// stages: 1 = screen (query fragments, seed pre-pass, MFMA screen -> candidate keys in the workspace), 2 = select (candidate keys ->
// re-rank -> outputs), 3 = both.  The plan and the workspace layout are functions of (B, n_items, d, k, options, X16 != NULL) only,
// so the two stages of a split search agree on where the candidates live.
static int knn_search_stages(int stages, const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                             const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k, int score_mode,
                             int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                             double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                             size_t workspace_bytes, uint32_t options, const void* X16, int x16_exp, void* stream) {
    if (B == 0) return HWER_OK;
    const bool do_screen = (stages & 1) != 0, do_select = (stages & 2) != 0;
    const bool conservative = (options & HWER_KNN_EXACT_THRESHOLDS) != 0;
    if (!X || !workspace || (do_screen && (!hnorm || !Q32)) || (do_select && (!Q64 || !out_idx || !out_score || !out_flags)))
        return fail(HWER_EINVAL, "knn_search: null pointer");
    if (do_select && !out_dist2 && (score_mode == 2 || !sort_by_score))
        return fail(HWER_EINVAL, "knn_search: out_dist2 may only be NULL with score_mode 0/1 and sort_by_score");
    const int dp = hwer_knn_padded_dim(d);
    if (dp < 0) return fail(HWER_EINVAL, "knn_search: d=%d unsupported (1..256)", d);
    if (ldx != dp) return fail(HWER_EINVAL, "knn_search: ldx must be hwer_knn_padded_dim(d)=%d (got %lld)", dp, (long long)ldx);
    if (k < 1 || k > K_MAX) return fail(HWER_EINVAL, "knn_search: k=%d outside 1..%d", k, K_MAX);
    if (k > n_items) return fail(HWER_EINVAL, "knn_search: k must be less than or equal to the number of training points");
    if (n_items >= (int64_t(1) << 32)) return fail(HWER_EINVAL, "knn_search: n_items must be < 2^32 per shard");
    if (do_select && (score_mode < 0 || score_mode > 2)) return fail(HWER_EINVAL, "knn_search: score_mode must be 0, 1 or 2");
    if (do_select && score_mode != 2 && !A32) return fail(HWER_EINVAL, "knn_search: A32 required for score_mode 0/1");
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (do_screen && (reinterpret_cast<uintptr_t>(Q32) & 15)) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(HWER_EINVAL, "knn_search: X/Q32 must be 16-byte and workspace 256-byte aligned");
    hipStream_t st = as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    if (!conservative && use_small_path(B, n_items, options)) {   // a handful of queries over a large catalogue: HBM-bound path (knn_small.hip)
        const SmallPlan sp = make_small_plan(B, n_items, int(ldx), k);
        // seeded thresholds (the streaming screen's sample pre-pass): ONE filtering pass over the catalogue instead of a key
        // matrix + threshold + collect passes
        const ScreenPlan seed_plan = make_screen_plan(B, n_items, int(ldx), k, options);
        // (only where few keys pass - above ~1 % of the catalogue per query the filtering pass spends its time parking keys and the
        // key-matrix passes are faster: small catalogues)
        const ScreenPlan* seed = (seed_plan.seed && seed_plan.seed_frac <= 0.01) ? &seed_plan : nullptr;
        if (seed && X16 && !screen_f16_supported(seed_plan)) X16 = nullptr;   // d <= 32: the pre-pass has no float16 variant, use the float32 rows
        const size_t need = WS_HEADER_BYTES + small_ws_bytes(sp, seed);
        if (workspace_bytes < need) return fail(HWER_EWORKSPACE, "knn_search: workspace %zu < %zu", workspace_bytes, need);
        WsHeader* hdr = reinterpret_cast<WsHeader*>(ws);
        uint64_t* list = nullptr;
        int32_t* cnt = nullptr;
        int rc = HWER_OK;
        prof_next_call();
        if (do_screen) {
            ProfScope ps(0, st);
            rc = launch_small(sp, X, hnorm, n_items, ldx, Q32, B, ws + WS_HEADER_BYTES, &list, &cnt, hdr, X16, x16_exp, st, seed);
        } else {
            small_outputs(sp, seed, ws + WS_HEADER_BYTES, &list, &cnt);   // where the screen stage left the lists
        }
        if (rc || !do_select) return rc;
        ProfScope ps_select(1, st);
        DenseSelectArgs da{};
        da.B = B;
        da.kp = sp.kprime;
        da.hnorm_max = hnorm_max;
        da.emit = EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, 0, row_ids, id_base, k,
                           out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)};
        da.list = list;
        da.list_count = cnt;
        da.list_cap = SMALL_LIST_CAP;
        da.list_min = sp.kprime;
        // float16 screen: two float16 roundings per product (2^-11 each) on top of the float32 accumulation error
        da.key_rel_err = X16 ? 1.02f * 0.0009765625f : 0.f;
        return launch_select_list(da, d, score_mode, st);
    }
    ScreenPlan plan = make_screen_plan(B, n_items, int(ldx), k, options);
    if (conservative) plan.scaled_first = 0;   // every threshold is an exact K'-th key: a list can only fail on genuine ties
    const WsLayout w = ws_layout(plan, B);
    if (workspace_bytes < w.total) return fail(HWER_EWORKSPACE, "knn_search: workspace %zu < %zu", workspace_bytes, w.total);
    uint64_t* cand = reinterpret_cast<uint64_t*>(ws + w.cand_a);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + w.cnt_a);
    WsHeader* hdr = reinterpret_cast<WsHeader*>(ws);
    int rc;
    prof_next_call();
    float* qf = reinterpret_cast<float*>(ws + w.qf);
    // float16 screen (option, streaming mode): float16 query fragments in the first half of the fragment area, the per-query
    // exponents in the second half (bq ints <= bq * ng * 16 bytes)
    const bool f16_screen = (options & HWER_KNN_F16_SCREEN) != 0 && X16 != nullptr && screen_f16_supported(plan);
    ScreenF16 f16{X16, qf, reinterpret_cast<const int32_t*>(ws + w.qf + size_t(plan.bq) * size_t(plan.ng) * 16), x16_exp};
    if (!do_screen) {
    } else if (f16_screen) {
        // (HWER_KNN_QUERIES_READY: the prologue - hwer_prepare_search_f32 / hwer_blend_prepare_search_f32 with bq < 0 - wrote the float16 fragments and exponents)
        if (!(options & HWER_KNN_QUERIES_READY))
        if ((rc = repack_queries16(Q32, B, plan.bq, int(ldx), qf, reinterpret_cast<int32_t*>(ws + w.qf + size_t(plan.bq) * size_t(plan.ng) * 16), st))) return rc;
    } else if (!(options & HWER_KNN_QUERIES_READY) || (options & HWER_KNN_F16_SCREEN)) {
        // (QUERIES_READY alone: hwer_prepare_search_f32 wrote the float32 fragments already.  QUERIES_READY together with F16_SCREEN when
        // the float16 screen does NOT run here - no X16, or a padded width below 64 - means the prologue may have left float16
        // fragments where this screen reads float32 ones: repack from Q32 rather than trust them)
        if ((rc = repack_queries(Q32, B, plan.bq, int(ldx), qf, st))) return rc;
    }
    rc = HWER_OK;
    if (do_screen) {
        ProfScope ps(0, st);
        const float* tau0 = nullptr;
        if (plan.seed) {   // sample pre-pass -> per-query initial thresholds
            float* seed_keys = reinterpret_cast<float*>(ws + w.seed_keys);
            float* tau = reinterpret_cast<float*>(ws + w.seed_tau);
            if ((rc = launch_seed_screen(plan, X, hnorm, n_items, ldx, qf, B, seed_keys, hdr, st, f16_screen ? &f16 : nullptr))) return rc;
            if ((rc = launch_seed_tau0(plan, seed_keys, B, tau, reinterpret_cast<int32_t*>(ws + w.seed_cnt), st))) return rc;
            tau0 = tau;
        }
        rc = launch_screen(plan, X, hnorm, n_items, ldx, qf, B, cand, cnt, reinterpret_cast<float*>(ws + w.cand_a), hdr, st,
                           f16_screen ? &f16 : nullptr, tau0);
    }
    if (rc || !do_select) return rc;
    ProfScope ps_select(1, st);
    if (plan.dense) {
        DenseSelectArgs da{};
        da.keys = reinterpret_cast<const float*>(ws + w.cand_a);
        da.ld = plan.ld_dense;
        da.B = B;
        da.kp = plan.kprime;
        da.hnorm_max = hnorm_max;
        da.key_rel_err = f16_screen ? 1.02f * 0.0009765625f : 0.f;
        da.emit = EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, diag_env("HWER_SELECT_SKIP") ? atoi(diag_env("HWER_SELECT_SKIP")) : 0, row_ids, id_base, k,
                           out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)};
        const bool coop = B <= coop_max_queries();
        const size_t lds = coop ? coop_lds_bytes(d, plan.kprime) : size_t(4) * wave_lds_bytes(d, plan.kprime);
        static_assert(DENSE_MAX_ITEMS <= 256 * 32, "dense select keeps at most 32 float4 of the row per lane");
        auto launch_dense2 = [&](auto mv, auto md) -> int {
            constexpr int MV = decltype(mv)::value, MD = decltype(md)::value;
            if (coop) {
                if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<MV, MD, false, true>), lds, "knn_select_dense")) return e;
                hipLaunchKernelGGL((select_dense_kernel<MV, MD, false, true>), dim3(unsigned(B)), dim3(256), lds, st, da);
                return HWER_OK;
            }
            if (int e = set_max_lds(reinterpret_cast<const void*>(&select_dense_kernel<MV, MD, false>), lds, "knn_select_dense")) return e;
            hipLaunchKernelGGL((select_dense_kernel<MV, MD, false>), dim3(unsigned(cdiv64(B, 4))), dim3(256), lds, st, da);
            return HWER_OK;
        };
        auto launch_dense = [&](auto mv) -> int {
            switch (score_mode) {
                case 0: return launch_dense2(mv, std::integral_constant<int, 0>{});
                case 1: return launch_dense2(mv, std::integral_constant<int, 1>{});
                default: return launch_dense2(mv, std::integral_constant<int, 2>{});
            }
        };
        if (plan.ld_dense <= 256 * 4) rc = launch_dense(std::integral_constant<int, 4>{});
        else if (plan.ld_dense <= 256 * 8) rc = launch_dense(std::integral_constant<int, 8>{});
        else if (plan.ld_dense <= 256 * 16) rc = launch_dense(std::integral_constant<int, 16>{});
        else rc = launch_dense(std::integral_constant<int, 32>{});
        if (rc) return rc;
        return check_launch("knn_select_dense");
    }

    int n_lists = plan.n_lists;
    uint64_t* cand_other = reinterpret_cast<uint64_t*>(ws + w.cand_b);
    int32_t* cnt_other = reinterpret_cast<int32_t*>(ws + w.cnt_b);
    const int32_t* todo = nullptr;
    if (n_lists > 1 && n_lists <= REDW_MAX_LISTS && plan.seed) {
        // seeded thresholds: the lists of a query are short - one wave merges ALL of them (up to 4 096 keys) in one pass and
        // writes where the rounds below would leave their result; only queries with more than 4 096 keys go through the rounds
        int rounds = 0;
        for (int nl = n_lists; nl > 1; nl = int(cdiv64(nl, MAX_LISTS_DIRECT))) ++rounds;
        uint64_t* cand_final = (rounds & 1) ? cand_other : cand;
        int32_t* cnt_final = (rounds & 1) ? cnt_other : cnt;
        int32_t* todo_w = reinterpret_cast<int32_t*>(ws + w.todo);
        // window: 1.5x the keys the seeded thresholds are expected to leave a query (c K' = seed_frac x n_items; a query with more takes
        // further windows), 1 024 ... 4 096
        int win = 1024;
        while (win < REDW_MAX && double(win) < 1.5 * plan.seed_frac * double(n_items)) win <<= 1;
        ReduceWaveArgs rw{cand, cnt, n_lists, cand_final, cnt_final, plan.bq, B, plan.kprime, todo_w, win};
        const size_t lds_w = size_t(REDW_WAVES) * size_t(win) * 8;
        if (win <= 2048) {
            hipLaunchKernelGGL(reduce_lists_wave_kernel<32>, dim3(unsigned(cdiv64(B, REDW_WAVES))), dim3(64 * REDW_WAVES), lds_w, st, rw);
        } else {
            if ((rc = set_max_lds(reinterpret_cast<const void*>(&reduce_lists_wave_kernel<REDW_NV>), lds_w, "knn_reduce_wave"))) return rc;
            hipLaunchKernelGGL(reduce_lists_wave_kernel<REDW_NV>, dim3(unsigned(cdiv64(B, REDW_WAVES))), dim3(64 * REDW_WAVES), lds_w, st, rw);
        }
        if ((rc = check_launch("knn_reduce_wave"))) return rc;
        // the wave kernel finishes every query (windows of 4 096 keys when a threshold came out far too low): the two workgroup
        // rounds that used to follow it - 2 launches of ~40 us of empty blocks at c4 - are gone
        cand = cand_final;
        cnt = cnt_final;
        n_lists = 1;
        (void)todo;
    }
    while (n_lists > 1) {   // item splits / shared-query waves: merge the lists of a query into one list of K' keys
        const int groups = int(cdiv64(n_lists, MAX_LISTS_DIRECT));
        ReduceArgs ra{cand, cnt, n_lists, cand_other, cnt_other, plan.bq, plan.kprime, todo};
        const size_t lds = 128 + size_t(MAX_LISTS_DIRECT) * 512 * 8;  // 8 lists x K' <= 8 x 448 keys -> 4096 slots
        if ((rc = set_max_lds(reinterpret_cast<const void*>(&reduce_lists_kernel), lds, "knn_reduce"))) return rc;
        hipLaunchKernelGGL(reduce_lists_kernel, dim3(unsigned(B), unsigned(groups)), dim3(256), lds, st, ra);
        if ((rc = check_launch("knn_reduce"))) return rc;
        std::swap(cand, cand_other);
        std::swap(cnt, cnt_other);
        n_lists = groups;
    }
    SelectArgs sa{};
    sa.cand = cand;
    sa.cand_cnt = cnt;
    sa.bq = plan.bq;
    sa.B = B;
    sa.kprime = plan.kprime;
    sa.kcap = plan.kprime;   // the screen's final compaction is exact: a list holds at most K' keys
    sa.hnorm_max = hnorm_max;
    sa.key_rel_err = f16_screen ? 1.02f * 0.0009765625f : 0.f;   // two float16 roundings per product (2^-11 each), as in the small-batch path
    sa.emit = EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, diag_env("HWER_SELECT_SKIP") ? atoi(diag_env("HWER_SELECT_SKIP")) : 0, row_ids, id_base, k,
                       out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)};
    return launch_select_rerank(sa, d, score_mode, st);
}



int hwer_knn_search_opts_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                             const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k, int score_mode,
                             int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                             double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                             size_t workspace_bytes, uint32_t options, const void* X16, int x16_exp, void* stream) {
    return knn_search_stages(3, X, hnorm, n_items, d, ldx, Q32, Q64, A32, lda, a_idx, B, k, score_mode, sort_by_score, row_ids, id_base, hnorm_max,
                             out_idx, out_score, out_dist2, out_flags, workspace, workspace_bytes, options, X16, x16_exp, stream);
}

int hwer_knn_screen_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32, int64_t B, int k,
                        void* workspace, size_t workspace_bytes, uint32_t options, const void* X16, int x16_exp, void* stream) {
    return knn_search_stages(1, X, hnorm, n_items, d, ldx, Q32, nullptr, nullptr, 0, nullptr, B, k, 0, 1, nullptr, 0, 0.f, nullptr, nullptr, nullptr,
                             nullptr, workspace, workspace_bytes, options, X16, x16_exp, stream);
}

int hwer_knn_select_f32(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32, int64_t lda,
                        const int32_t* a_idx, int64_t B, int k, int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base,
                        float hnorm_max, int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                        size_t workspace_bytes, uint32_t options, const void* X16, void* stream) {
    return knn_search_stages(2, X, nullptr, n_items, d, ldx, nullptr, Q64, A32, lda, a_idx, B, k, score_mode, sort_by_score, row_ids, id_base,
                             hnorm_max, out_idx, out_score, out_dist2, out_flags, workspace, workspace_bytes, options, X16, 0, stream);
}

// Stage-A chunk of the on-device exhaustive pass: about 64 blocks per query on small catalogues (more parallelism, shorter
// sorts), EX_CHUNK items on large ones; always a power of two >= k so that a block can hand over k entries.
static int exact_device_chunk(int64_t n_items, int k) {
    int c = 1024;
    while (c < EX_CHUNK && int64_t(c) * 64 < n_items) c <<= 1;
    while (c < k) c <<= 1;
    return c;
}

// chunks per first-level group: as many as one 4096-entry sort takes, more when that would give over 64 groups
static int exact_device_group(int64_t n_chunks, int k) {
    const int64_t by_sort = std::max<int64_t>(1, EX_CHUNK / k);
    return int(std::max<int64_t>(by_sort, cdiv64(n_chunks, EXACT_DEVICE_GROUPS)));
}

size_t hwer_knn_exact_device_workspace_bytes(int64_t n_items, int k) {
    if (n_items < 1 || k < 1 || k > K_MAX) return 0;
    const int64_t n_chunks = cdiv64(n_items, exact_device_chunk(n_items, k));
    const int64_t n_groups = cdiv64(n_chunks, exact_device_group(n_chunks, k));
    return align256(size_t(EXACT_DEVICE_MAXQ) * size_t(n_chunks) * size_t(k) * sizeof(Ent)) +
           align256(size_t(EXACT_DEVICE_MAXQ) * size_t(n_groups) * size_t(k) * sizeof(Ent));
}

int hwer_knn_exact_device(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                          int64_t lda, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                          const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                          double* out_dist2, int32_t* out_flags, void* knn_workspace, int32_t* missed_accum,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !Q64 || !out_idx || !out_score || !out_flags || !knn_workspace || !workspace)
        return fail(HWER_EINVAL, "knn_exact_device: null pointer");
    if (d < 1 || d > 256 || ldx < ((d + 15) / 16) * 16)
        return fail(HWER_EINVAL, "knn_exact_device: need 1 <= d <= 256 and ldx >= roundup16(d)");
    if (k < 1 || k > K_MAX) return fail(HWER_EINVAL, "knn_exact_device: k=%d outside 1..%d", k, K_MAX);
    if (k > n_items) return fail(HWER_EINVAL, "knn_exact_device: k must be less than or equal to the number of training points");
    if (n_items >= (int64_t(1) << 32)) return fail(HWER_EINVAL, "knn_exact_device: n_items must be < 2^32");
    if (score_mode < 0 || score_mode > 2 || (score_mode != 2 && !A32)) return fail(HWER_EINVAL, "knn_exact_device: score_mode / A32");
    const size_t need = hwer_knn_exact_device_workspace_bytes(n_items, k);
    if (workspace_bytes < need) return fail(HWER_EWORKSPACE, "knn_exact_device: workspace %zu < %zu", workspace_bytes, need);
    ExactDevArgs ea{};
    ea.X = X;
    ea.n_items = n_items;
    ea.ldx = ldx;
    ea.Q64 = Q64;
    ea.hdr = reinterpret_cast<WsHeader*>(knn_workspace);
    ea.cand = reinterpret_cast<Ent*>(workspace);
    ea.chunk = exact_device_chunk(n_items, k);
    ea.n_chunks = int(cdiv64(n_items, ea.chunk));
    ea.chunks_per_group = exact_device_group(ea.n_chunks, k);
    ea.n_groups = int(cdiv64(ea.n_chunks, ea.chunks_per_group));
    ea.cand2 = reinterpret_cast<Ent*>(static_cast<unsigned char*>(workspace) + align256(size_t(EXACT_DEVICE_MAXQ) * size_t(ea.n_chunks) * size_t(k) * sizeof(Ent)));
    ea.missed_accum = missed_accum;
    ea.emit = EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, 0, row_ids, id_base, k,
                       out_idx, out_score, out_dist2, out_flags, nullptr, nullptr};
    // sort area | max(re-rank area of the emitting wave, four waves' staging areas of stage A)
    const size_t lds = size_t(EX_CHUNK) * sizeof(Ent) + std::max(wave_lds_bytes(d, ((k + 63) / 64) * 64), size_t(4) * 64 * RR_RS * 4);
    int rc;
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&exact_device_kernel), lds, "knn_exact_device"))) return rc;
    hipLaunchKernelGGL(exact_device_kernel, dim3(unsigned(ea.n_chunks), EXACT_DEVICE_MAXQ), dim3(256), lds, as_stream(stream), ea);
    return check_launch("knn_exact_device");
}

size_t hwer_knn_exact_workspace_bytes(int64_t n_items, int k) {
    const size_t dev = hwer_knn_exact_device_workspace_bytes(n_items, k);
    return dev ? WS_HEADER_BYTES + dev : 0;   // its own list header + the device pass's buffers
}

// Host-driven exhaustive pass: the listed queries go through the same kernel as hwer_knn_exact_device, 8 per launch.
int hwer_knn_exact_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                       int64_t lda, const int32_t* a_idx, const int32_t* qsel_host, int64_t nq, int k, int score_mode, int sort_by_score,
                       const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                       double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (nq == 0) return HWER_OK;
    if (!qsel_host || !workspace) return fail(HWER_EINVAL, "knn_exact: null pointer");
    const size_t need = hwer_knn_exact_workspace_bytes(n_items, k);
    if (need == 0) return fail(HWER_EINVAL, "knn_exact: k=%d outside 1..%d or empty catalogue", k, K_MAX);
    if (workspace_bytes < need) return fail(HWER_EWORKSPACE, "knn_exact: workspace %zu < %zu", workspace_bytes, need);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(HWER_EINVAL, "knn_exact: workspace must be 256-byte aligned");
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    WsHeader* hdr = reinterpret_cast<WsHeader*>(ws);
    for (int64_t q0 = 0; q0 < nq; q0 += EXACT_DEVICE_MAXQ) {
        const int n = int(std::min<int64_t>(EXACT_DEVICE_MAXQ, nq - q0));
        ExactIds ids{};
        for (int i = 0; i < n; ++i) ids.id[i] = qsel_host[q0 + i];
        hipLaunchKernelGGL(exact_list_init_kernel, dim3(1), dim3(256), 0, as_stream(stream), hdr, n, ids);
        int rc;
        if ((rc = check_launch("knn_exact(list)"))) return rc;
        if ((rc = hwer_knn_exact_device(X, n_items, d, ldx, Q64, A32, lda, a_idx, k, score_mode, sort_by_score, row_ids, id_base, out_idx,
                                        out_score, out_dist2, out_flags, ws, nullptr, ws + WS_HEADER_BYTES, workspace_bytes - WS_HEADER_BYTES,
                                        stream)))
            return rc;
    }
    return HWER_OK;
}

int hwer_merge_topk(const int64_t* idx, const double* score, const double* dist2, int n_lists, int64_t B, int k,
                    int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2, void* stream) {
    return hwer_merge_topk_lists(idx, score, dist2, B * k, n_lists, B, k, k, sort_by_score, out_idx, out_score, out_dist2, nullptr, stream);
}

int hwer_merge_topk_strided(const int64_t* idx, const double* score, const double* dist2, int64_t list_stride, int n_lists,
                            int64_t B, int k, int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2,
                            void* stream) {
    return hwer_merge_topk_lists(idx, score, dist2, list_stride, n_lists, B, k, k, sort_by_score, out_idx, out_score, out_dist2, nullptr, stream);
}

int hwer_merge_topk_lists(const int64_t* idx, const double* score, const double* dist2, int64_t list_stride, int n_lists,
                          int64_t B, int k_in, int k, int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2,
                          int32_t* out_exhausted, void* stream) {
    if (B == 0) return HWER_OK;
    if (!idx || !score || !dist2 || !out_idx || !out_score || !out_dist2) return fail(HWER_EINVAL, "merge_topk: null pointer");
    if (n_lists < 1 || k < 1 || k > 512) return fail(HWER_EINVAL, "merge_topk: need n_lists >= 1 and 1 <= k <= 512");
    if (k_in < 1 || k_in > k) return fail(HWER_EINVAL, "merge_topk: need 1 <= k_in <= k (got k_in=%d, k=%d)", k_in, k);
    if (int64_t(n_lists) * k_in < k) return fail(HWER_EINVAL, "merge_topk: n_lists * k_in = %lld < k = %d", (long long)(int64_t(n_lists) * k_in), k);
    if (list_stride < B * k_in) return fail(HWER_EINVAL, "merge_topk: list_stride %lld < B * k_in", (long long)list_stride);
    if (k_in != k && n_lists > MERGE_RANK_LISTS) return fail(HWER_EINVAL, "merge_topk: shortened lists (k_in < k) need n_lists <= %d", MERGE_RANK_LISTS);
    const int64_t total = int64_t(n_lists) * k_in;
    if (total > 8192) return fail(HWER_EINVAL, "merge_topk: n_lists * k_in = %lld > 8192", (long long)total);
    const size_t lds = merge_lds_bytes(n_lists, k);   // (sized for full lists: covers k_in <= k)
    int rc;
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&merge_topk_kernel), lds, "merge_topk"))) return rc;
    if (out_exhausted && n_lists > MERGE_RANK_LISTS) {   // the sort path merges full lists only: nothing can be exhausted
        hipError_t e = hipMemsetAsync(out_exhausted, 0, size_t(B) * 4, as_stream(stream));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "merge_topk: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    MergeArgs ma{idx, score, dist2, n_lists, B, k, sort_by_score, out_idx, out_score, out_dist2, list_stride, k_in, out_exhausted};
    hipLaunchKernelGGL(merge_topk_kernel, dim3(unsigned(B)), dim3(256), lds, as_stream(stream), ma);
    return check_launch("merge_topk");
}

}  // extern "C"
