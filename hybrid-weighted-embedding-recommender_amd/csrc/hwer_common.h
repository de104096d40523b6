// Shared helpers for the gfx950 kernels behind include/hwer_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <math.h>
#include <type_traits>

#include "hwer_hip.h"

namespace hwer {

constexpr int kWave = 64;  // CDNA4 wavefront

void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);

// Tuning / ablation switches read from the environment exist only in diagnostic builds (HIPCC_EXTRA=-DHWER_DIAG_STAMPS or -DHWER_DIAG_ABLATE):
// the production library never calls getenv() on the search path, and no stray variable can change its results.
#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
static inline const char* diag_env(const char* name) { return getenv(name); }
#else
static inline const char* diag_env(const char*) { return nullptr; }
#endif

// hipFuncSetAttribute applies to the CURRENT device only: a "done" cache must be kept per device (bit = device id)
static inline bool attr_pending(uint64_t mask, int* dev_out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    *dev_out = dev & 63;
    return ((mask >> (dev & 63)) & 1ull) == 0ull;
}

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

// ---- lane exchanges without the LDS crossbar -----------------------------------------------------------------------
// lane_xor32<LM>(v): the value of lane (lane ^ LM).  __shfl_xor compiles to ds_bpermute_b32 (an LDS-pipe round trip, ~100
// cycles of latency per dependent step); these are one or two VALU moves: DPP quad permutes / row mirrors inside a row of 16
// lanes, v_permlane16_swap / v_permlane32_swap (gfx950) across rows.  Sort networks and wave reductions are chains of such
// exchanges, so their latency - what a wave-per-query kernel waits for - drops several-fold.
template <int LM>
__device__ __forceinline__ uint32_t lane_xor32(uint32_t v) {
    static_assert(LM == 1 || LM == 2 || LM == 4 || LM == 8 || LM == 16 || LM == 32, "power-of-two lane mask");
    const int x = int(v);
    if constexpr (LM == 1) {
        return uint32_t(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    } else if constexpr (LM == 2) {
        return uint32_t(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    } else if constexpr (LM == 4) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, true);      // row_half_mirror: i -> 7 - i
        return uint32_t(__builtin_amdgcn_update_dpp(t, t, 0x1B, 0xF, 0xF, true));    // quad_perm [3,2,1,0]: (7 - i) -> i ^ 4
    } else if constexpr (LM == 8) {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, true);      // row_mirror: i -> 15 - i
        return uint32_t(__builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, 0xF, true));   // row_half_mirror: (15 - i) -> i ^ 8
    } else if constexpr (LM == 16) {
        // [0]: odd rows replaced by the even rows' data, [1]: even rows replaced by the odd rows' data
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);          // [0]: both halves = lanes 0..31, [1]: = lanes 32..63
        return (threadIdx.x & 32) ? r[0] : r[1];
    }
}
template <int LM>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v) {
    return (uint64_t(lane_xor32<LM>(uint32_t(v >> 32))) << 32) | uint64_t(lane_xor32<LM>(uint32_t(v)));
}
// sum over the wave, result in every lane
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += int(lane_xor32<1>(uint32_t(v)));
    v += int(lane_xor32<2>(uint32_t(v)));
    v += int(lane_xor32<4>(uint32_t(v)));
    v += int(lane_xor32<8>(uint32_t(v)));
    v += int(lane_xor32<16>(uint32_t(v)));
    v += int(lane_xor32<32>(uint32_t(v)));
    return v;
}

// wave-wide max / min / float64 sum (result in every lane) on the VALU lane exchanges above: six dependent steps of one or two moves
// instead of six ds_bpermute round trips (~100 cycles each in a chain that has nothing else to issue); the butterfly order of __shfl_xor
// reductions, so float64 sums are bit-identical to them
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, __uint_as_float(lane_xor32<1>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(lane_xor32<2>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(lane_xor32<4>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(lane_xor32<8>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(lane_xor32<16>(__float_as_uint(v))));
    v = fmaxf(v, __uint_as_float(lane_xor32<32>(__float_as_uint(v))));
    return v;
}
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, __uint_as_float(lane_xor32<1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor32<2>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor32<4>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor32<8>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor32<16>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor32<32>(__float_as_uint(v))));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += __longlong_as_double((long long)lane_xor64<1>((uint64_t)__double_as_longlong(v)));
    v += __longlong_as_double((long long)lane_xor64<2>((uint64_t)__double_as_longlong(v)));
    v += __longlong_as_double((long long)lane_xor64<4>((uint64_t)__double_as_longlong(v)));
    v += __longlong_as_double((long long)lane_xor64<8>((uint64_t)__double_as_longlong(v)));
    v += __longlong_as_double((long long)lane_xor64<16>((uint64_t)__double_as_longlong(v)));
    v += __longlong_as_double((long long)lane_xor64<32>((uint64_t)__double_as_longlong(v)));
    return v;
}

// inclusive prefix sum over the 64 lanes (lane i <- v[0] + ... + v[i]) in six DPP steps, no LDS traffic: Hillis-Steele inside the 16-lane
// rows (row_shr never crosses a row: a lane whose source lies outside keeps `old` = 0), then lane 15 of rows 0 / 2 onto rows 1 / 3
// (row_bcast:15) and lane 31 onto rows 2 and 3 (row_bcast:31).  tools/ubench/scan_test.hip checks it against a host prefix sum.
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- wave-level selection on register-resident floats ------------------------------------------------------------
// v[0..NV) per lane (entries that do not exist hold -inf).  wave_find_rank returns `lo` with
//   rank <= count(v >= lo) <= rank + slack
// whenever the data allows it (it stops early when the interval is exhausted by ties: then count(>= lo) may be larger
// and the caller trims).  Counting is one v_cmp per key with the popcount / sum on the scalar unit;
// probes are placed by interpolating log(count) between the bracket ends (the keys of one query look like the tail of
// a bell curve, where log-count is close to linear), with a plain bisection step every third probe as a safeguard.
// Precondition: at least `rank` finite entries.
template <int NV>
__device__ __forceinline__ int wave_count_ge(const float (&v)[NV], float t) {
    int c = 0;   // ballot counting: one v_cmp per key, popcount and sum on the scalar unit
#pragma unroll
    for (int i = 0; i < NV; ++i) c += __builtin_popcountll(__ballot(v[i] >= t));
    return c;
}
// CountGe: callable float -> wave-uniform count of keys >= t.  `hi_out` (nullable): a key bound with count(>= hi) < rank,
// tightened until count(>= hi) >= rank - slack when `tighten_hi` (callers use it to tell provable members apart).
template <class CountGe>
__device__ __forceinline__ float wave_find_rank_fn(CountGe count_ge, float vmin, float vmax, int rank, int slack, int& count_ge_lo,
                                                  float* hi_out = nullptr, bool tighten_hi = false) {
    float lo = vmin, hi = vmax;   // invariants: count(>= lo) >= rank;  count(>= hi) = c_hi < rank (after the first test)
    int c_lo = count_ge(lo), c_hi = count_ge(hi);
    if (hi_out) *hi_out = INFINITY;
    if (c_hi >= rank) {           // the rank-th largest key IS the maximum (ties at the top)
        count_ge_lo = c_hi;
        return hi;
    }
    const float target = float(rank) + 0.5f * float(slack);
    for (int iter = 0; iter < 96 && (c_lo > rank + slack || (tighten_hi && c_hi < rank - slack)); ++iter) {
        float f = 0.5f;
        if (iter % 3 != 2) {
            // aim at the side that is still loose: just above the rank for lo, just below it for hi
            const float tg = c_lo > rank + slack ? target : float(rank) - 0.5f * float(slack);
            const float l0 = __logf(float(c_lo)), l1 = __logf(float(c_hi) + 0.5f);
            f = (l0 - __logf(tg)) / (l0 - l1);
            f = fminf(fmaxf(f, 0.03f), 0.97f);
        }
        float mid = lo + f * (hi - lo);
        if (!(mid > lo) || !(mid < hi)) {
            mid = 0.5f * lo + 0.5f * hi;
            if (!(mid > lo) || !(mid < hi)) break;   // hi is the next float above lo: the interval is exhausted
        }
        const int c = count_ge(mid);
        if (c >= rank) { lo = mid; c_lo = c; } else { hi = mid; c_hi = c; }
    }
    count_ge_lo = c_lo;
    if (hi_out) *hi_out = hi;
    return lo;
}
template <int NV>
__device__ __forceinline__ float wave_find_rank(const float (&v)[NV], int rank, int slack, int& count_ge_lo, float* hi_out = nullptr,
                                               bool tighten_hi = false) {
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        vmax = fmaxf(vmax, v[i]);
        if (v[i] > -INFINITY) vmin = fminf(vmin, v[i]);
    }
    vmax = wave_max_f32(vmax);
    vmin = wave_min_f32(vmin);
    return wave_find_rank_fn([&](float t) { return wave_count_ge<NV>(v, t); }, vmin, vmax, rank, slack, count_ge_lo, hi_out, tighten_hi);
}

// ---- wave-level sort of NE u64 keys per lane, ascending, in registers ---------------------------------------------
// Blocked layout: after the call, the key of sorted position p sits in lane p / NE, register p % NE.  Bitonic network:
// strides >= NE are one cross-lane exchange per key, smaller strides are register swaps.
template <int NE>
__device__ __forceinline__ void wave_sort_u64(uint64_t (&key)[NE]) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll 1
    for (int size = 2; size <= NE * kWave; size <<= 1) {
#pragma unroll 1
        for (int stride = size >> 1; stride >= NE; stride >>= 1) {
            const int lm = stride / NE;
            const bool keep_min = ((lane & lm) == 0) == (((NE * lane) & size) == 0);   // size >= 2 NE here
            // (ds_bpermute exchanges: the 2 NE independent permutes of a stage pipeline through the LDS crossbar; a DPP /
            // permlane-swap version of this network - lane_xor64 - measured SLOWER, 17.7 K against 12 K cycles for 256 keys with
            // one wave per SIMD: its VALU moves sit in the stage's dependent chain, profiles/r2_select_stamps.txt)
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const uint64_t o = __shfl_xor(key[j], lm, kWave);
                if ((o < key[j]) == keep_min) key[j] = o;   // equal keys are identical padding: either copy will do
            }
        }
#pragma unroll
        for (int stride = NE >> 1; stride > 0; stride >>= 1) {
            if (stride < size) {
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    if ((j & stride) == 0) {
                        const bool asc = ((NE * lane + j) & size) == 0;
                        const uint64_t x = key[j], y = key[j | stride];
                        if ((y < x) == asc) { key[j] = y; key[j | stride] = x; }
                    }
                }
            }
        }
    }
}
// ONE key per lane, ascending over the lanes (lane l ends up with the l-th smallest): the 21 compare-exchange stages of the 64-input bitonic
// network on DPP / permlane-swap exchanges.  With a single key per lane a stage has nothing to pipeline through the LDS crossbar, so the
// VALU exchanges win here (the multi-key sort above keeps ds_bpermute).  Keys must be distinct or identical copies.
template <int SIZE, int LM>
__device__ __forceinline__ void wave_sort1_stage(uint64_t& key, int lane) {
    const uint64_t o = lane_xor64<LM>(key);
    const bool keep_min = ((lane & LM) == 0) == ((lane & SIZE) == 0);   // SIZE = 64: ascending everywhere
    if ((o < key) == keep_min) key = o;
}
__device__ __forceinline__ void wave_sort1_u64(uint64_t& key) {
    const int lane = threadIdx.x & (kWave - 1);
    wave_sort1_stage<2, 1>(key, lane);
    wave_sort1_stage<4, 2>(key, lane); wave_sort1_stage<4, 1>(key, lane);
    wave_sort1_stage<8, 4>(key, lane); wave_sort1_stage<8, 2>(key, lane); wave_sort1_stage<8, 1>(key, lane);
    wave_sort1_stage<16, 8>(key, lane); wave_sort1_stage<16, 4>(key, lane); wave_sort1_stage<16, 2>(key, lane); wave_sort1_stage<16, 1>(key, lane);
    wave_sort1_stage<32, 16>(key, lane); wave_sort1_stage<32, 8>(key, lane); wave_sort1_stage<32, 4>(key, lane); wave_sort1_stage<32, 2>(key, lane);
    wave_sort1_stage<32, 1>(key, lane);
    wave_sort1_stage<64, 32>(key, lane); wave_sort1_stage<64, 16>(key, lane); wave_sort1_stage<64, 8>(key, lane); wave_sort1_stage<64, 4>(key, lane);
    wave_sort1_stage<64, 2>(key, lane); wave_sort1_stage<64, 1>(key, lane);
}
// number of entries of an ascending run of 64 distinct keys (LDS) that are smaller than `key`: seven dependent probes
__device__ __forceinline__ int run64_count_less(const uint64_t* run, uint64_t key) {
    int idx = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) idx += run[idx + s - 1] < key ? s : 0;
    return idx + ((idx == 63 && run[63] < key) ? 1 : 0);
}

// The keys sorted above are (approximation << 32 | payload) with a MONOTONE approximation of the exact sort key, so
// only runs of equal approximations can be out of order.  Detect such runs (rare) and finish them with odd-even
// transposition passes under the exact comparator `ex_less(key_x, key_y)`.  Approximation 0xFFFFFFFF = padding.
template <int NE, class ExLess>
__device__ __forceinline__ void wave_sort_fixup(uint64_t (&key)[NE], ExLess ex_less) {
    static_assert(NE >= 2 && NE % 2 == 0, "blocked layout with an even number of keys per lane");
    const int lane = threadIdx.x & (kWave - 1);
    auto same = [](uint64_t x, uint64_t y) { return uint32_t(x >> 32) == uint32_t(y >> 32) && uint32_t(x >> 32) != 0xFFFFFFFFu; };
    bool tie = false;
#pragma unroll
    for (int j = 0; j + 1 < NE; ++j) tie = tie || same(key[j], key[j + 1]);
    {
        const uint64_t dn = __shfl_down(key[0], 1, kWave);
        tie = tie || (lane < kWave - 1 && same(key[NE - 1], dn));
    }
    if (!__any(tie)) return;
#pragma unroll 1
    for (int pass = 0; pass < NE * kWave; ++pass) {
        bool swapped = false;
#pragma unroll
        for (int j = 0; j + 1 < NE; j += 2) {      // even positions
            if (same(key[j], key[j + 1]) && ex_less(key[j + 1], key[j])) {
                const uint64_t t = key[j]; key[j] = key[j + 1]; key[j + 1] = t;
                swapped = true;
            }
        }
#pragma unroll
        for (int j = 1; j + 1 < NE; j += 2) {      // odd positions inside the lane
            if (same(key[j], key[j + 1]) && ex_less(key[j + 1], key[j])) {
                const uint64_t t = key[j]; key[j] = key[j + 1]; key[j + 1] = t;
                swapped = true;
            }
        }
        {                                          // odd position across the lane boundary: (lane's last, next lane's first)
            const uint64_t up = __shfl_up(key[NE - 1], 1, kWave), dn = __shfl_down(key[0], 1, kWave);
            const bool sa = lane > 0 && same(up, key[0]) && ex_less(key[0], up);
            const bool sb = lane < kWave - 1 && same(key[NE - 1], dn) && ex_less(dn, key[NE - 1]);
            if (sa) key[0] = up;
            if (sb) key[NE - 1] = dn;
            swapped = swapped || sa || sb;
        }
        if (!__any(swapped)) break;
    }
}

// Order the LDS traffic of ONE wave (cross-lane hand-over through wave-private LDS).  The LDS pipeline executes the
// instructions of a wave in issue order, so a ds_read issued after a ds_write already sees every lane's data: all that
// is needed is to stop the COMPILER from reordering or caching the accesses.  (A workgroup-scope fence would also drain
// vmcnt(0), i.e. wait for every global load in flight - fatal for software-pipelined gathers.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace hwer
