// Template kernels of stage 2 of the exact k-NN (select / re-rank / certificate / emit, on-device exhaustive pass), shared by the
// translation units that instantiate them: knn_select.hip (float32 rows) and knn_select_f64.hip (genuine float64 rows, round 6).
#pragma once
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
    // F64 instantiations (genuine float64 rows, round 6; knn_select_f64.hip): every VALUE - distances, scores - comes from these rows, X above is
    // only the float32 rounding the screen ran on (never read by an F64 kernel).  Row of local item i: X64 + (x64_rows ? x64_rows[i] : i) * ld64;
    // ld64 is a multiple of 8 doubles with zero padding (whole 64-byte segments per 8-column chunk), the table 64-byte aligned.
    const double* X64;
    int64_t ld64;
    const int64_t* x64_rows;
    const double* A64;     // anchor rows for score mode 0: A64[(a_idx ? a_idx[b] : b) * lda64]
    int64_t lda64;
};
#ifdef HWER_DIAG_STAMPS
#define SEL_STAMP(slot) do { if (a_stamps && (threadIdx.x & 63) == 0) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); __builtin_amdgcn_sched_barrier(0); a_stamps[(slot)] = t_; \
    if ((slot) == 0) a_stamps[10] = __builtin_amdgcn_s_memrealtime(); if ((slot) == 6) a_stamps[11] = __builtin_amdgcn_s_memrealtime(); } } while (0)   /* 10 / 11: start / end in 10-ns ticks (one clock for the whole chip) */
#define CK_NOW(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
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
// Round 6: the staging area is filled by LDS-DMA (global_load_lds_dwordx4: no register destination, no ds_write_b128 - that store costs ~13 LDS
// cycles per wave-instruction against 4 for the ds_read_b128 that takes the transposed rows back out, MI355X_MICROARCH.md, LDS).  A DMA
// instruction writes its 64 x 16 bytes lane-linearly (1 KiB, no padding), so the conflict-free image comes from WHICH row a lane fetches:
// piece b (b = 0 .. 3, at byte b * RR_PIECE) holds 16 rows x 64 B; the candidate that reader lane l takes out sits in piece m >> 2 at row
// 4 g + (m & 3), where g is l's ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32) and m its position in the
// group: the 16 lanes of a group then read 16-byte slots (b + 4 (m & 3) + t) mod 16 - all different - of the 256-byte bank row.
#ifndef HWER_SELECT_GLDS
#define HWER_SELECT_GLDS 1
#endif
constexpr int RR_PIECE = 1024 + 16;  // byte distance of the four 1-KiB DMA pieces (one 16-byte slot of skew per piece)
constexpr int RR_BUF = 4160;         // one chunk's four pieces (3 * RR_PIECE + 1024 = 4144, rounded to 64 bytes)
static_assert(3 * RR_PIECE + 1024 <= RR_BUF && RR_BUF <= 64 * RR_RS * 4, "the four DMA pieces fit the staging area");
// Chunk buffers per wave of the COOP form (one query per workgroup, few workgroups: LDS is plentiful).  Two - the DMA two chunks ahead of the
// arithmetic - were built and measured: c1's chunk loop 6 463 -> 6 446 cycles per query, step 0.0481 -> 0.0490 ms; a chunk's time is not its
// gather (the in-loop stamps: 11 % waiting for the pieces, 79 % in the arithmetic part with its scalar operand loads; profiles/r6_ab_select_rounds.txt).
constexpr int RR_COOP_NBUF = 1;
constexpr int RR_COOP_STAGE = (HWER_SELECT_GLDS && RR_COOP_NBUF > 1) ? RR_COOP_NBUF * RR_BUF : 64 * RR_RS * 4;   // staging bytes per wave of the COOP form
// reader lane l -> (group g, member m)
__device__ __forceinline__ void rr_reader_gm(int l, int& g, int& m) {
    const int x = l & 31;
    const bool odd = (x >= 4 && x < 12) || (x >= 16 && x < 20) || x >= 28;
    g = 2 * (l >> 5) + (odd ? 1 : 0);
    m = odd ? (x < 12 ? x - 4 : (x < 20 ? x - 8 : x - 16)) : (x < 4 ? x : (x < 16 ? x - 8 : x - 12));
}
// (group g, member m) -> reader lane
__device__ __forceinline__ int rr_reader_lane(int g, int m) {
    const int base = 32 * (g >> 1);
    if (g & 1) return base + (m < 8 ? m + 4 : (m < 12 ? m + 8 : m + 16));
    return base + (m < 4 ? m : (m < 8 ? m + 8 : m + 12));
}
__host__ __device__ inline size_t wave_lds_bytes(int d, int kp) {
    (void)d;
    return size_t(kp) * (8 + 8 + 4) + size_t(64) * RR_RS * 4;
}
// COOP variant: ONE query per 256-thread workgroup - the four waves share the query's tables and split the re-rank rounds
// (one staging area per wave); for launches with so few queries that a wave per query leaves most of the chip idle
__host__ __device__ inline size_t coop_lds_bytes(int d, int kp) { (void)d; return size_t(kp) * (8 + 8 + 4) + size_t(4) * RR_COOP_STAGE; }

// F64ROWS: the keys were computed on the float32 ROUNDING of float64 rows, the values come from the float64 rows: rounding moves q.x by at most
// ||q|| 2^-24 ||x|| and 0.5 ||x||^2 by at most 2^-24 ||x||^2 (1 + 2^-25), i.e. the true key by <= 2^-24 (||q|| ||x||max + 2 hnorm_max) (1 + tiny)
// <= 2 u M - two more units of u M on top of the (2 d + 16) the float32 flavour uses.
template <bool F64ROWS = false>
__device__ __forceinline__ float e32_bound(int d, double Q2, float hnorm_max) {
    // |key32 - (q.x - 0.5||x||^2)| <= E32 for every item: fmaf chain of d+1 terms, float32 rounding of q and of the
    // half norm.  Generous constant factor; typical errors are ~100x smaller.
    // With M = ||q|| ||x||max + hnorm_max the first-order error of the plain chain is (d + 3) u M (d + 1 additions on partial sums <= M, the
    // roundings of q and of the half norm).  The ping-pong screen's folded form (knn_screen.hip: the accumulator runs tau' - key', the key is
    // recovered as tau' - accumulator) has partial sums <= |tau'| + M <= 2 M (tau' is a key of the same query) over d + 2 additions and one more
    // rounding in the recovery: (2 d + 7) u M - still inside the (2 d + 16) u M used here, whose second-order remainder (d u)^2 M is 1e-9 M.
    const double u = 5.9604644775390625e-08;  // 2^-24
    const double xn = sqrt(2.0 * double(hnorm_max)) * (1.0 + 1e-6);
    return float((2.0 * d + (F64ROWS ? 19.0 : 16.0)) * u * (sqrt(Q2) * xn + double(hnorm_max)) * (1.0 + 1e-6));
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
// F64 (round 6): the candidates' rows are GENUINE float64 rows (EmitArgs::X64) - what the reference's KD-tree and predict work on when `vectors` is
// float64 (hwer/recommendation_base.py:74,79,141-143; ContentRecommendation: hwer/content_recommender.py:96).  Same round structure with 8-column
// chunks (one 64-byte segment of a float64 row per 4 lanes, the same 80-byte LDS row stride); distances as above; the score's products are no longer
// exact, so numpy's `(a * b).sum(1)` is a rounded product and a rounded add per element - 5 float64 instructions per element and candidate where
// the float32 rows need 6 (two conversions).
template <int MODE, bool COOP = false, bool F64 = false>   // score_mode as a compile-time constant: the float64 and float32 score chains never share registers
__device__ __forceinline__ void wave_rerank_emit_t(const EmitArgs& a, int64_t b, unsigned char* lds, int kp, int n_r, int n_b, int flag) {
    static_assert(!(F64 && MODE == 1), "float64 rows are scored in float64 (mode 0) or from the distance (mode 2)");
    const int d = a.d, lane = threadIdx.x & 63;
    const int coop_wave = COOP ? int(threadIdx.x >> 6) : 0;
    unsigned long long* const a_stamps = (a.stamps && coop_wave == 0) ? a.stamps + b * 16 : nullptr;
    (void)a_stamps;
    SEL_STAMP(3);
    double* t_d2 = reinterpret_cast<double*>(lds);
    double* t_sc = t_d2 + kp;
    uint32_t* cand = reinterpret_cast<uint32_t*>(t_sc + kp);
    constexpr int NBUF = COOP ? RR_COOP_NBUF : 1;   // chunk buffers of this wave's staging area
    float* stage = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(cand + kp) + coop_wave * (COOP ? RR_COOP_STAGE : 64 * RR_RS * 4));

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
    const cdouble_p a64 = (cdouble_p)((F64 && a.A64) ? a.A64 + arow * a.lda64 : a.Q64);   // (F64, mode 0)
    // row of the float64 table that holds local item `item` (F64)
    // (row strides are < 2^32 bytes - ldx <= 256 floats, ld64 checked by the launchers - and item ids are 32-bit: one v_mad_u64_u32 per row address
    // where the int64 stride took two and two moves; round 6)
    const uint32_t ld64_bytes = uint32_t(a.ld64) * 8u, ldx_bytes = uint32_t(a.ldx) * 4u;
    auto xrow64 = [&](uint32_t item) -> const double* {
        return a.x64_rows ? a.X64 + a.x64_rows[item] * a.ld64
                          : reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.X64) + uint64_t(item) * uint64_t(ld64_bytes));
    };
    auto xrow32 = [&](uint32_t item) -> const float* { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.X) + uint64_t(item) * uint64_t(ldx_bytes)); };
    (void)a32; (void)a64; (void)xrow64; (void)xrow32;
    // numpy pairwise structure for this d (d <= 256: at most two leaves)
    const bool two = d > 128;
    int n2 = d / 2;
    n2 -= n2 & 7;
    const int gA = two ? n2 / 8 : -1;          // first 8-group of the second leaf
    const int lim = d - (d & 7);               // elements >= lim form the sequential tail of the last leaf
    const int n_groups8 = lim / 8;
    constexpr int mode = MODE;
    constexpr int CW = F64 ? 8 : RR_CW;   // columns per step: one 64-byte segment of a row
    const int n_chunks = (d + CW - 1) / CW;
    const bool lazy = a.out_dist2 == nullptr && a.sort_by_score && mode != 2 && n_b < n_r;
#if HWER_SELECT_GLDS
    // LDS-DMA staging maps (see RR_PIECE): what this lane fetches for the four pieces, and where it reads its own candidate's row
    const int fq = lane & 3;
    const int fl0 = rr_reader_lane(lane >> 4, 0 + ((lane >> 2) & 3)), fl1 = rr_reader_lane(lane >> 4, 4 + ((lane >> 2) & 3));
    const int fl2 = rr_reader_lane(lane >> 4, 8 + ((lane >> 2) & 3)), fl3 = rr_reader_lane(lane >> 4, 12 + ((lane >> 2) & 3));
    int rd_g, rd_m;
    rr_reader_gm(lane, rd_g, rd_m);
    const unsigned char* const rd = reinterpret_cast<const unsigned char*>(stage) + (rd_m >> 2) * RR_PIECE + 64 * (4 * rd_g + (rd_m & 3));
    const uint32_t stage_lds = uint32_t(__builtin_amdgcn_readfirstlane(int(lds_addr(stage))));
#endif

#ifdef HWER_DIAG_STAMPS
    unsigned long long ck_vm = 0, ck_lds = 0, ck_cmp = 0, ck_n = 0, ck_pro = 0;   // chunk-loop time by part (wave-per-query forms: stamp slots 12 - 15)
#endif
#pragma unroll 1
    for (int g0 = 64 * coop_wave; g0 < ((a.debug_skip & 1) ? 0 : n_r); g0 += COOP ? 256 : 64) {   // wave-uniform rounds of 64 candidates
#ifdef HWER_DIAG_STAMPS   // ablation of the round's ingredients (HWER_SELECT_SKIP bits 8 no distances, 16 no score arithmetic, 32 no row gathers, 64 no LDS staging)
        const bool need_d2 = (!lazy || g0 < n_b) && !(a.debug_skip & 8);
        const bool abl_score = a.debug_skip & 16, abl_gather = a.debug_skip & 32, abl_stage = a.debug_skip & 64;
#else
        const bool need_d2 = !lazy || g0 < n_b;   // wave-uniform
        constexpr bool abl_score = false, abl_gather = false, abl_stage = false;
#endif
#ifdef HWER_DIAG_STAMPS
        unsigned long long ckr0, ckr1;
        CK_NOW(ckr0);
#endif
        const int ci = g0 + lane;
        const bool live = ci < n_r;
        const uint32_t it = min(cand[live ? ci : g0], a.item_max);
        double acc = 0.0;
        double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0, sA = 0.0;
        float f0 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, f5 = 0, f6 = 0, f7 = 0, fA = 0.f;
        if constexpr (F64) {
        // the same staging map for 8-byte elements: 4 lanes x 16 B (two doubles) per row = the 64-byte segment of an 8-column chunk, 16 rows per
        // instruction, 4 instructions per chunk, transposed through the same wave-private LDS area (row stride 80 B = 10 doubles)
#if HWER_SELECT_GLDS
        const double* src0 = xrow64(min(cand[min(g0 + fl0, n_r - 1)], a.item_max)) + 2 * fq;
        const double* src1 = xrow64(min(cand[min(g0 + fl1, n_r - 1)], a.item_max)) + 2 * fq;
        const double* src2 = xrow64(min(cand[min(g0 + fl2, n_r - 1)], a.item_max)) + 2 * fq;
        const double* src3 = xrow64(min(cand[min(g0 + fl3, n_r - 1)], a.item_max)) + 2 * fq;
        lds_dma16_v4<RR_PIECE>(src0, src1, src2, src3, stage_lds);
        if (NBUF == 2 && n_chunks > 1) lds_dma16_v4<RR_PIECE>(src0 + CW, src1 + CW, src2 + CW, src3 + CW, stage_lds + RR_BUF);
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            // the chunk's four pieces have landed (the DMA is invisible to hipcc's counters; loads retire in order: 4 = the next chunk's pieces)
            if (NBUF == 2 && c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned char* const rdc = rd + (NBUF == 2 ? (c & 1) * RR_BUF : 0);
            double xv[CW];
#pragma unroll
            for (int t = 0; t < CW / 2; ++t) {
                const double2 v = *reinterpret_cast<const double2*>(rdc + 16 * t);
                xv[2 * t] = v.x; xv[2 * t + 1] = v.y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are in registers: the next pieces may overwrite them
            if (c + NBUF < n_chunks) {   // (ld64 is a multiple of 8: the last chunk's loads stay inside the row's zero padding)
                const int o = CW * (c + NBUF);
                lds_dma16_v4<RR_PIECE>(src0 + o, src1 + o, src2 + o, src3 + o, stage_lds + (NBUF == 2 ? (c & 1) * RR_BUF : 0));
            }
#else
        constexpr int RS64 = RR_RS / 2;
        const int srow = lane >> 2, scol = 2 * (lane & 3);
        const double* src0 = xrow64(min(cand[min(g0 + 0 + srow, n_r - 1)], a.item_max)) + scol;
        const double* src1 = xrow64(min(cand[min(g0 + 16 + srow, n_r - 1)], a.item_max)) + scol;
        const double* src2 = xrow64(min(cand[min(g0 + 32 + srow, n_r - 1)], a.item_max)) + scol;
        const double* src3 = xrow64(min(cand[min(g0 + 48 + srow, n_r - 1)], a.item_max)) + scol;
        double* const stage64 = reinterpret_cast<double*>(stage);
        double* dst = stage64 + srow * RS64 + scol;
        double2 pre0 = *reinterpret_cast<const double2*>(src0), pre1 = *reinterpret_cast<const double2*>(src1);
        double2 pre2 = *reinterpret_cast<const double2*>(src2), pre3 = *reinterpret_cast<const double2*>(src3);
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            *reinterpret_cast<double2*>(dst) = pre0;
            *reinterpret_cast<double2*>(dst + 16 * RS64) = pre1;
            *reinterpret_cast<double2*>(dst + 32 * RS64) = pre2;
            *reinterpret_cast<double2*>(dst + 48 * RS64) = pre3;
            if (c + 1 < n_chunks) {   // (ld64 is a multiple of 8: the last chunk's loads stay inside the row's zero padding)
                const int o = CW * (c + 1);
                pre0 = *reinterpret_cast<const double2*>(src0 + o);
                pre1 = *reinterpret_cast<const double2*>(src1 + o);
                pre2 = *reinterpret_cast<const double2*>(src2 + o);
                pre3 = *reinterpret_cast<const double2*>(src3 + o);
            }
            wave_lds_sync();
            double xv[CW];
#pragma unroll
            for (int t = 0; t < CW / 2; ++t) {
                const double2 v = *reinterpret_cast<const double2*>(stage64 + lane * RS64 + 2 * t);
                xv[2 * t] = v.x; xv[2 * t + 1] = v.y;
            }
            wave_lds_sync();
#endif
            const int j0 = CW * c;
            if (!need_d2) {
            } else if (j0 + CW <= d) {   // whole chunk: straight-line code
#pragma unroll
                for (int t = 0; t < CW; ++t) {
                    const double tmp = __dsub_rn(q64[j0 + t], xv[t]);
                    acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                }
            } else {
#pragma unroll
                for (int t = 0; t < CW; ++t) {
                    if (j0 + t < d) {   // uniform (never reads the query beyond its d elements)
                        const double tmp = __dsub_rn(q64[j0 + t], xv[t]);
                        acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                    }
                }
            }
            if constexpr (mode == 0) {
                const int g8 = c;   // one 8-group of numpy's pairwise leaf per chunk
                if (g8 < n_groups8) {   // uniform
                    const int jj = 8 * g8;
                    const double p0 = __dmul_rn(a64[jj + 0], xv[0]), p1 = __dmul_rn(a64[jj + 1], xv[1]), p2 = __dmul_rn(a64[jj + 2], xv[2]), p3 = __dmul_rn(a64[jj + 3], xv[3]);
                    const double p4 = __dmul_rn(a64[jj + 4], xv[4]), p5 = __dmul_rn(a64[jj + 5], xv[5]), p6 = __dmul_rn(a64[jj + 6], xv[6]), p7 = __dmul_rn(a64[jj + 7], xv[7]);
                    // (the chains start from zero: 0 + p = p up to the sign of a zero, which (res + 1) / 2 cannot see; the leaf hand-over is a real branch -
                    // see the float32-row block below)
                    r0 = __dadd_rn(r0, p0); r1 = __dadd_rn(r1, p1); r2 = __dadd_rn(r2, p2); r3 = __dadd_rn(r3, p3);
                    r4 = __dadd_rn(r4, p4); r5 = __dadd_rn(r5, p5); r6 = __dadd_rn(r6, p6); r7 = __dadd_rn(r7, p7);
                    if (g8 == gA - 1) {
                        asm volatile("" ::: "memory");
                        sA = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                        r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = 0.0;
                    }
                }
            }
        }
        } else {
        // staging map: 4 lanes x 16 B per row (whole 64-byte segments: the L1 serves 64 B per clock, per-lane row walks
        // would use a quarter of each access), 16 rows per instruction, 4 instructions per 16-column chunk; the rows are
        // transposed through wave-private LDS (row stride 80 B: conflict-free ds_read_b128).
        // (named scalars, not arrays: hipcc leaves small arrays touched inside a rolled loop in scratch)
#if HWER_SELECT_GLDS
        // (diagnostic mask 32: the rows of the candidate SLOTS instead of the candidates' - 256 rows that stay in L2, distinct per slot so that no
        // ties appear: what the select costs when its gathers never leave the L2)
        const float* src0 = xrow32(min(abl_gather ? uint32_t(g0 + fl0) : cand[min(g0 + fl0, n_r - 1)], a.item_max)) + 4 * fq;
        const float* src1 = xrow32(min(abl_gather ? uint32_t(g0 + fl1) : cand[min(g0 + fl1, n_r - 1)], a.item_max)) + 4 * fq;
        const float* src2 = xrow32(min(abl_gather ? uint32_t(g0 + fl2) : cand[min(g0 + fl2, n_r - 1)], a.item_max)) + 4 * fq;
        const float* src3 = xrow32(min(abl_gather ? uint32_t(g0 + fl3) : cand[min(g0 + fl3, n_r - 1)], a.item_max)) + 4 * fq;
        // software pipeline: the DMA of chunk c + 1 is in flight while chunk c is reduced in float64
        lds_dma16_v4<RR_PIECE>(src0, src1, src2, src3, stage_lds);
        if (NBUF == 2 && n_chunks > 1) lds_dma16_v4<RR_PIECE>(src0 + RR_CW, src1 + RR_CW, src2 + RR_CW, src3 + RR_CW, stage_lds + RR_BUF);
        if (COOP && g0 == 0) SEL_STAMP(7);
#ifdef HWER_DIAG_STAMPS
        CK_NOW(ckr1);
        ck_pro += ckr1 - ckr0;
#endif
#pragma unroll 1
        for (int c = 0; c < n_chunks; ++c) {
            // the chunk's four pieces have landed (the DMA is invisible to hipcc's counters; loads retire in order: 4 = the next chunk's pieces)
#ifdef HWER_DIAG_STAMPS
            unsigned long long ck0, ck1, ck2, ck3;
            CK_NOW(ck0);
#endif
            if (NBUF == 2 && c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef HWER_DIAG_STAMPS
            CK_NOW(ck1);
#endif
            if (COOP && g0 == 0 && c == 0) SEL_STAMP(8);
            const unsigned char* const rdc = rd + (NBUF == 2 ? (c & 1) * RR_BUF : 0);
            float xv[RR_CW];
#pragma unroll
            for (int t = 0; t < RR_CW / 4; ++t) {
                const float4 v = *reinterpret_cast<const float4*>(rdc + 16 * t);
                xv[4 * t] = v.x; xv[4 * t + 1] = v.y; xv[4 * t + 2] = v.z; xv[4 * t + 3] = v.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are in registers: the next pieces may overwrite them
#ifdef HWER_DIAG_STAMPS
            CK_NOW(ck2);
#endif
            if (c + NBUF < n_chunks) {
                const int o = RR_CW * (c + NBUF);
                lds_dma16_v4<RR_PIECE>(src0 + o, src1 + o, src2 + o, src3 + o, stage_lds + (NBUF == 2 ? (c & 1) * RR_BUF : 0));
            }
#else
        const int srow = lane >> 2, scol = 4 * (lane & 3);
        const float* src0 = xrow32(min(cand[min(g0 + 0 + srow, n_r - 1)], a.item_max)) + scol;
        const float* src1 = xrow32(min(cand[min(g0 + 16 + srow, n_r - 1)], a.item_max)) + scol;
        const float* src2 = xrow32(min(cand[min(g0 + 32 + srow, n_r - 1)], a.item_max)) + scol;
        const float* src3 = xrow32(min(cand[min(g0 + 48 + srow, n_r - 1)], a.item_max)) + scol;
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
#endif
            // Per 8-column group (one 8-group of numpy's pairwise leaf): the candidate's elements are widened ONCE and serve both the distance
            // chain and the score chains (round 6: the distance and the score blocks each converted them - 16 v_cvt per chunk of a boundary round).
            // The score chains start from zero and always accumulate: fma(a, x, 0) is the exact product (the only difference from numpy's
            // r = a * b is the sign of a zero, which (res + 1) / 2 cannot see), so a group has no `first` branch, and the leaf hand-over
            // (d > 128: two leaves) is a REAL branch - as a select it cost the 7-add reduction tree + 2 v_cndmask per 8-group, 18 of the 70 vector
            // instructions of a chunk, for a value that is only wanted once per round (round 6, found in the ISA).
#pragma unroll
            for (int gg = 0; gg < RR_CW / 8; ++gg) {
                const int g8 = (RR_CW / 8) * c + gg;
                const int jj = 8 * g8;
                double xd[8];
                if constexpr (mode == 0) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) xd[t] = double(xv[8 * gg + t]);
                }
                if (need_d2) {   // wave-uniform
                    if (jj + 8 <= d) {   // whole group (always, when d is a multiple of 8): straight-line code
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            double xt;
                            if constexpr (mode == 0) xt = xd[t]; else xt = double(xv[8 * gg + t]);
                            const double tmp = __dsub_rn(q64[jj + t], xt);
                            acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            if (jj + t < d) {   // uniform (never reads the query beyond its d elements)
                                const double tmp = __dsub_rn(q64[jj + t], double(xv[8 * gg + t]));
                                acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
                            }
                        }
                    }
                }
                if constexpr (mode != 2) {
                    if (g8 < n_groups8 && !abl_score) {   // uniform
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
#ifdef HWER_ABL_NO_A32   // timing experiment only (wrong scores): no scalar loads of the anchor row
                            const double a0 = xd[1], a1 = xd[2], a2 = xd[3], a3 = xd[4], a4 = xd[5], a5 = xd[6], a6 = xd[7], a7 = xd[0];
#else
                            // (round 6, measured and NOT kept: the anchor row staged behind the chunk buffer and read back as four broadcast ds_read_b128 beside
                            // the candidate rows - no scalar loads in the chunk: c3 select -2 %, c2 +2 %, c1 +7 %; profiles/r6_ab_select_rounds.txt)
                            const double a0 = double(a32[jj + 0]), a1 = double(a32[jj + 1]), a2 = double(a32[jj + 2]), a3 = double(a32[jj + 3]);
                            const double a4 = double(a32[jj + 4]), a5 = double(a32[jj + 5]), a6 = double(a32[jj + 6]), a7 = double(a32[jj + 7]);
#endif
                            r0 = __fma_rn(a0, xd[0], r0); r1 = __fma_rn(a1, xd[1], r1); r2 = __fma_rn(a2, xd[2], r2); r3 = __fma_rn(a3, xd[3], r3);
                            r4 = __fma_rn(a4, xd[4], r4); r5 = __fma_rn(a5, xd[5], r5); r6 = __fma_rn(a6, xd[6], r6); r7 = __fma_rn(a7, xd[7], r7);
                            if (g8 == gA - 1) {   // end of the first leaf (d > 128 only): its sum, and the chains start over
                                asm volatile("" ::: "memory");   // keeps this a branch: as a select the tree below runs after every group
                                sA = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
                                r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = 0.0;
                            }
                        } else {
                            const float p0 = __fmul_rn((a32[jj + 0]), xv[8 * gg + 0]), p1 = __fmul_rn((a32[jj + 1]), xv[8 * gg + 1]);
                            const float p2 = __fmul_rn((a32[jj + 2]), xv[8 * gg + 2]), p3 = __fmul_rn((a32[jj + 3]), xv[8 * gg + 3]);
                            const float p4 = __fmul_rn((a32[jj + 4]), xv[8 * gg + 4]), p5 = __fmul_rn((a32[jj + 5]), xv[8 * gg + 5]);
                            const float p6 = __fmul_rn((a32[jj + 6]), xv[8 * gg + 6]), p7 = __fmul_rn((a32[jj + 7]), xv[8 * gg + 7]);
                            // (0 + p = p: the chains start from zero, see above)
                            f0 = __fadd_rn(f0, p0); f1 = __fadd_rn(f1, p1); f2 = __fadd_rn(f2, p2); f3 = __fadd_rn(f3, p3);
                            f4 = __fadd_rn(f4, p4); f5 = __fadd_rn(f5, p5); f6 = __fadd_rn(f6, p6); f7 = __fadd_rn(f7, p7);
                            if (g8 == gA - 1) {
                                asm volatile("" ::: "memory");
                                fA = ((f0 + f1) + (f2 + f3)) + ((f4 + f5) + (f6 + f7));
                                f0 = f1 = f2 = f3 = f4 = f5 = f6 = f7 = 0.f;
                            }
                        }
                    }
                }
            }
#ifdef HWER_DIAG_STAMPS
            CK_NOW(ck3);
            ck_vm += ck1 - ck0; ck_lds += ck2 - ck1; ck_cmp += ck3 - ck2; ++ck_n;
#endif
        }
        }   // !F64
        if (COOP && g0 == 0) SEL_STAMP(9);
        double sco;
        const float* xr = xrow32(it);
        (void)xr;
        if constexpr (mode == 0 && F64) {
            double res = n_groups8 > 0 ? ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) : 0.0;
            const double* xr64 = xrow64(it);
            for (int j = lim; j < d; ++j) res = __dadd_rn(res, __dmul_rn(a64[j], xr64[j]));   // <= 7 tail elements
            if (two) res = sA + res;
            sco = (res + 1.0) / 2.0;
        } else if constexpr (mode == 0) {
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
#ifdef HWER_DIAG_STAMPS
    if (!COOP && a_stamps && lane == 0 && ck_n) {   // (slot 15's top bit tells these from the one-workgroup-per-query stamps)
        a_stamps[12] = ck_vm; a_stamps[13] = ck_lds; a_stamps[14] = ck_cmp; a_stamps[15] = (1ull << 62) | (ck_pro << 8) | ck_n;
    }
#endif
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
            d2 = 0.0;
            if constexpr (F64) {
                const double* xr = xrow64(min(cand[slot], a.item_max));
                for (int j = 0; j < d; ++j) {
                    const double tmp = __dsub_rn(q64[j], xr[j]);
                    d2 = __dadd_rn(d2, __dmul_rn(tmp, tmp));
                }
            } else {
                const float* xr = xrow32(min(cand[slot], a.item_max));
                for (int j = 0; j < d; ++j) {
                    const double tmp = __dsub_rn(q64[j], double(xr[j]));
                    d2 = __dadd_rn(d2, __dmul_rn(tmp, tmp));
                }
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

template <bool F64 = false>
__device__ __forceinline__ void wave_rerank_emit(const EmitArgs& a, int64_t b, unsigned char* lds, int kp, int n_r, int n_b, int flag) {
    if constexpr (F64) {
        if (a.score_mode == 0) wave_rerank_emit_t<0, false, true>(a, b, lds, kp, n_r, n_b, flag);   // wave-uniform
        else wave_rerank_emit_t<2, false, true>(a, b, lds, kp, n_r, n_b, flag);
    } else {
    switch (a.score_mode) {   // wave-uniform
        case 0: wave_rerank_emit_t<0>(a, b, lds, kp, n_r, n_b, flag); break;
        case 1: wave_rerank_emit_t<1>(a, b, lds, kp, n_r, n_b, flag); break;
        default: wave_rerank_emit_t<2>(a, b, lds, kp, n_r, n_b, flag); break;
    }
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

template <int MODE, bool F64 = false>
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
    const float e32 = e32_bound<F64>(d, Q2, a.hnorm_max) + a.key_rel_err * float(sqrt(Q2)) * sqrtf(2.0f * a.hnorm_max) * 1.0001f;
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
    wave_rerank_emit_t<MODE, false, F64>(a.emit, b, lds_emit, kp, n_r, n_b, flag);
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
template <int MAXV, int MODE, bool LIST, bool COOP = false, bool F64 = false>   // float4 slices of the key row per lane: rows up to 256 * MAXV keys; score mode
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
    const float e32 = e32_bound<F64>(d, Q2, a.hnorm_max) + a.key_rel_err * float(sqrt(Q2)) * sqrtf(2.0f * a.hnorm_max) * 1.0001f;
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
    wave_rerank_emit_t<MODE, COOP, F64>(a.emit, b, lds_emit, kp, n_r, n_b, flag);
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

template <bool F64 = false>   // F64: distances and scores from the genuine float64 rows (EmitArgs::X64), 8-column chunks
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
        for (int r0 = wave * 64; F64 && r0 < a.chunk; r0 += 256) {
            constexpr int RS64 = RR_RS / 2;
            const int64_t item = base + r0 + lane;
            const int scol2 = 2 * (lane & 3);
            auto xrow64 = [&](int64_t it_) -> const double* { return a.emit.X64 + (a.emit.x64_rows ? a.emit.x64_rows[it_] : it_) * a.emit.ld64; };
            const double* src0 = xrow64(min(base + r0 + 0 + srow, last)) + scol2;
            const double* src1 = xrow64(min(base + r0 + 16 + srow, last)) + scol2;
            const double* src2 = xrow64(min(base + r0 + 32 + srow, last)) + scol2;
            const double* src3 = xrow64(min(base + r0 + 48 + srow, last)) + scol2;
            double* const stage64 = reinterpret_cast<double*>(stage);
            double* dst = stage64 + srow * RS64 + scol2;
            double2 pre0 = *reinterpret_cast<const double2*>(src0), pre1 = *reinterpret_cast<const double2*>(src1);
            double2 pre2 = *reinterpret_cast<const double2*>(src2), pre3 = *reinterpret_cast<const double2*>(src3);
            double acc = 0.0;
            const int n_steps8 = (d + 7) / 8;
#pragma unroll 1
            for (int c = 0; c < n_steps8; ++c) {
                *reinterpret_cast<double2*>(dst) = pre0;
                *reinterpret_cast<double2*>(dst + 16 * RS64) = pre1;
                *reinterpret_cast<double2*>(dst + 32 * RS64) = pre2;
                *reinterpret_cast<double2*>(dst + 48 * RS64) = pre3;
                if (c + 1 < n_steps8) {
                    const int o = 8 * (c + 1);
                    pre0 = *reinterpret_cast<const double2*>(src0 + o);
                    pre1 = *reinterpret_cast<const double2*>(src1 + o);
                    pre2 = *reinterpret_cast<const double2*>(src2 + o);
                    pre3 = *reinterpret_cast<const double2*>(src3 + o);
                }
                wave_lds_sync();
                double xv[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double2 v = *reinterpret_cast<const double2*>(stage64 + lane * RS64 + 2 * t);
                    xv[2 * t] = v.x; xv[2 * t + 1] = v.y;
                }
                wave_lds_sync();
                const int j0 = 8 * c;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (j0 + t < d) {   // uniform
                        const double tmp = __dsub_rn(q64[j0 + t], xv[t]);
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
#pragma unroll 1
        for (int r0 = wave * 64; !F64 && r0 < a.chunk; r0 += 256) {
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
    if (tid < 64) wave_rerank_emit<F64>(a.emit, b, lds_emit, kp, k, k, 2);
}

// genuine float64 rows (round 6): what an F64 launch reads beside the float32 arguments (EmitArgs::X64 ...)
struct F64Rows {
    const double* X64;
    int64_t ld64;
    const int64_t* x64_rows;
    const double* A64;
    int64_t lda64;
};
// launchers of the float64-row instantiations (knn_select_f64.hip); score_mode 0 or 2
int launch_select_list_f64(const DenseSelectArgs& da, int d, int score_mode, bool coop, hipStream_t st);
int launch_select_dense_f64(const DenseSelectArgs& da, int d, int score_mode, bool coop, hipStream_t st);
int launch_select_rerank_f64(const SelectArgs& sa, int d, int score_mode, hipStream_t st);
int launch_exact_device_f64(const ExactDevArgs& ea, size_t lds, hipStream_t st);

static inline int set_max_lds(const void* fn, size_t bytes, const char* what) {
    if (bytes <= 64 * 1024) return HWER_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "%s: hipFuncSetAttribute(%zu): %s", what, bytes, hipGetErrorString(e));
    return HWER_OK;
}

}  // namespace hwer
