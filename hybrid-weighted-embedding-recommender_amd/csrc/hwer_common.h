// Shared helpers for the gfx950 kernels behind include/hwer_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "hwer_hip.h"

namespace hwer {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid cap for grid-stride memory-bound kernels: 256 CUs x 8 blocks
constexpr int kMaxStreamBlocks = 2048;

// ---- order-preserving float <-> uint mapping (larger float  <=>  larger uint) -------------------
__device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
// candidate key: high word = orderable score, low word = ~local item id, so that a DESCENDING u64 sort
// is (score desc, id asc).
__device__ __forceinline__ uint64_t pack_key(float score, uint32_t item) {
    return (uint64_t(f32_orderable(score)) << 32) | uint64_t(0xFFFFFFFFu - item);
}
__device__ __forceinline__ float key_score(uint64_t k) { return f32_from_orderable(uint32_t(k >> 32)); }
__device__ __forceinline__ uint32_t key_item(uint64_t k) { return 0xFFFFFFFFu - uint32_t(k); }

// ---- numpy's pairwise summation, executed by a group of 8 consecutive lanes ----------------------
// p: LDS array of n values (already the products / squares), k = lane & 7.  Returns the sum in all 8 lanes,
// bit-identical to numpy `add.reduce` over a contiguous axis (loops_utils.h.src @TYPE@_pairwise_sum):
//   n < 8        : sequential from 0
//   8 <= n <= 128: 8 strided chains, ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n%8 tail sequentially
//   n > 128      : split at n2 = n/2 - (n/2)%8, recurse
template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int m) {
    return __shfl_xor(v, m, kWave);
}
template <typename T>
__device__ __forceinline__ T np_pairwise_leaf(const T* p, int n, int k) {
    if (n < 8) {
        T r = T(0);
        for (int i = 0; i < n; ++i) r = r + p[i];
        return r;
    }
    const int lim = n - (n & 7);
    T r = p[k];
    for (int i = 8; i < lim; i += 8) r = r + p[i + k];
    r = r + shfl_xor_t(r, 1);
    r = r + shfl_xor_t(r, 2);
    r = r + shfl_xor_t(r, 4);
    for (int i = lim; i < n; ++i) r = r + p[i];
    return r;
}
template <typename T, int DEPTH>
__device__ __forceinline__ T np_pairwise(const T* p, int n, int k) {
    if constexpr (DEPTH == 0) {
        return np_pairwise_leaf(p, n, k);
    } else {
        if (n <= 128) return np_pairwise_leaf(p, n, k);
        int n2 = n / 2;
        n2 -= n2 & 7;
        T a = np_pairwise<T, DEPTH - 1>(p, n2, k);
        T b = np_pairwise<T, DEPTH - 1>(p + n2, n - n2, k);
        return a + b;
    }
}
constexpr int kPairwiseDepth = 3;  // d <= 1024

// ---- wave-level selection by float bisection ---------------------------------------------------------------------
// v[0..NV) per lane (entries that do not exist hold -inf).  Finds `lo` with rank <= count(v >= lo) <= rank + slack
// whenever the data allows it (stops early when the interval is exhausted by ties: then count(>= lo) may be larger and
// the caller trims the keys equal to lo).  Conflict-free: ballot counting (v_cmp + s_bcnt1) per register, no LDS.
// Precondition: at least `rank` finite entries.
template <int NV>
__device__ __forceinline__ int wave_count_ge(const float (&v)[NV], float t) {
    // one v_cmp (-> SGPR pair) + s_bcnt1 + s_add per register: scalar counting, no cross-lane traffic
    int c = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) c += __builtin_popcountll(__ballot(v[i] >= t));
    return c;
}
template <int NV>
__device__ __forceinline__ float wave_bisect_rank(const float (&v)[NV], int rank, int slack, int& count_ge_lo) {
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        vmax = fmaxf(vmax, v[i]);
        if (v[i] > -INFINITY) vmin = fminf(vmin, v[i]);
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, off, kWave));
        vmin = fminf(vmin, __shfl_xor(vmin, off, kWave));
    }
    float lo = vmin, hi = vmax;   // invariant: count(>= lo) >= rank
    int c_lo = wave_count_ge<NV>(v, lo);
    for (int iter = 0; iter < 64 && c_lo > rank + slack; ++iter) {
        const float mid = 0.5f * lo + 0.5f * hi;
        if (!(mid > lo) || !(mid < hi)) {   // hi is the next float above lo (or equal): try hi itself, then stop
            const int c = wave_count_ge<NV>(v, hi);
            if (c >= rank) { lo = hi; c_lo = c; }
            break;
        }
        const int c = wave_count_ge<NV>(v, mid);
        if (c >= rank) { lo = mid; c_lo = c; } else { hi = mid; }
    }
    count_ge_lo = c_lo;
    return lo;
}

// make LDS traffic of one wave visible to its own lanes (no cross-wave sync)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace hwer
