// Stage 1 of the exact k-NN (K5): float32 MFMA screen with a fused streaming top-K' filter, gfx950.
//
//   key(q, i) = q . x_i - 0.5 ||x_i||^2        (ordering by key desc == ordering by ||q - x_i|| asc)
//
// Orientation: items are the MFMA A rows, queries the B columns, so with the 32x32 C layout
// (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) every lane holds 16 item keys of ONE query:
// the top-K reduction axis is lane-local and the per-query threshold lives in a register.
//
// Work split: a workgroup = 4 waves = `wq` waves along queries x `wi = 4 / wq` waves along the item subtiles.
// A wave owns 32*TQ queries for the whole launch (their MFMA B fragments stay in registers: NG*TQ float4) and a
// private pair of candidate streams per query (one per lane half), so the filter needs no atomics and no
// cross-lane traffic.  Item tiles (IB rows) stream HBM -> registers -> LDS (double buffered, one barrier per
// tile).  When a stream is about to overflow, the wave radix-selects the exact K'-th key of that query (ballot /
// LDS histogram), compacts both streams in place and raises the threshold.
//
// Bound: fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles / SIMD) once a workgroup keeps >= ~50 queries resident;
// HBM (streaming X once per query block) below that.
#include <algorithm>

#include "knn_internal.h"

namespace hwer {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct ScreenArgs {
    const float* X;
    const float* hnorm;
    int64_t n_items;
    int64_t ldx;
    const float* Q;
    int64_t B;
    uint64_t* cand;
    int32_t* cand_cnt;
    int n_qblocks;
    int wq;
    int64_t split_items;
    int64_t bq;
    int kprime;
};

__device__ __forceinline__ uint32_t ld_u32_l2(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: served by L2, never by a stale L1 line
}
__device__ __forceinline__ uint64_t ld_u64_l2(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exact K'-selection over the two streams (s0: c0 entries, s1: c1 entries) of one query, by the whole wave.
// Afterwards the streams hold exactly `kprime` entries in total (all keys > T, plus the first few == T), and
// T (orderable bits of the K'-th largest score) is the new strict threshold.
__device__ __noinline__ void compact_query(uint64_t* s0, uint64_t* s1, int c0, int c1, int kprime, uint32_t* hist,
                                           uint32_t& T_out, int& n0_out, int& n1_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's appends have reached L2
    uint32_t sc[2 * CAPH / 64];
#pragma unroll
    for (int c = 0; c < CAPH / 64; ++c) {
        const int e = 64 * c + lane;
        sc[c] = e < c0 ? ld_u32_l2(reinterpret_cast<const uint32_t*>(s0 + e) + 1) : 0u;
        sc[CAPH / 64 + c] = e < c1 ? ld_u32_l2(reinterpret_cast<const uint32_t*>(s1 + e) + 1) : 0u;
    }
    uint32_t prefix = 0, mask = 0;
    int want = kprime;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        *reinterpret_cast<uint4*>(hist + 4 * lane) = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 2 * CAPH / 64; ++i) {
            if (sc[i] != 0u && (sc[i] & mask) == prefix) atomicAdd(&hist[(sc[i] >> shift) & 255u], 1u);
        }
        wave_lds_sync();
        const uint4 h4 = *reinterpret_cast<const uint4*>(hist + 4 * lane);
        const int s = int(h4.x + h4.y + h4.z + h4.w);
        int S = s;  // inclusive suffix sum over lanes (bins descending)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_down(S, off, kWave);
            if (lane + off < 64) S += t;
        }
        const uint64_t bal = __ballot(S >= want);
        const int lstar = 63 - __builtin_clzll(bal);  // bal != 0: the streams hold >= kprime valid keys
        int bstar = 0, want_new = want;
        {
            int cum = S - s;
            const uint32_t hh[4] = {h4.x, h4.y, h4.z, h4.w};
            bool done = false;
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                if (!done) {
                    if (cum + int(hh[t]) >= want) {
                        bstar = 4 * lane + t;
                        want_new = want - cum;
                        done = true;
                    } else {
                        cum += int(hh[t]);
                    }
                }
            }
        }
        bstar = __shfl(bstar, lstar, kWave);
        want = __shfl(want_new, lstar, kWave);
        prefix |= uint32_t(bstar) << shift;
        mask |= 0xFFu << shift;
    }
    const uint32_t T = prefix;  // exact K'-th largest orderable score; keep `want` of the entries equal to it
    int eq_seen = 0;
    int n_out[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        uint64_t* s = st == 0 ? s0 : s1;
        const int cn = st == 0 ? c0 : c1;
        int out = 0;
        for (int c = 0; 64 * c < cn; ++c) {
            const int e = 64 * c + lane;
            const bool valid = e < cn;
            const uint64_t key = valid ? ld_u64_l2(s + e) : 0ull;
            const uint32_t sv = uint32_t(key >> 32);
            const bool gt = valid && sv > T;
            const bool eq = valid && sv == T;
            const uint64_t be = __ballot(eq);
            const int rank = eq_seen + __builtin_popcountll(be & lt_mask);
            const bool keep = gt || (eq && rank < want);
            const uint64_t bk = __ballot(keep);
            if (keep) s[out + __builtin_popcountll(bk & lt_mask)] = key;  // in place: out + .. <= e
            out += __builtin_popcountll(bk);
            eq_seen += __builtin_popcountll(be);
        }
        n_out[st] = out;
    }
    T_out = T;
    n0_out = n_out[0];
    n1_out = n_out[1];
}

template <int NG, int TQ, int IB>
__global__ __launch_bounds__(256, 2) void screen_kernel(const ScreenArgs a) {
    constexpr int RS = NG * 8 + 4;                 // LDS row stride in floats (one 16-B pad chunk: conflict-free b128)
    constexpr int CHUNKS = IB * NG * 2;            // float4 chunks per tile
    constexpr int NST = CHUNKS / 256;              // staging float4 per thread
    constexpr int S = IB / 32;                     // 32-row subtiles per tile
    static_assert(CHUNKS % 256 == 0, "tile must split evenly over 256 threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);               // [2][IB][RS]
    float* hn = xs + 2 * IB * RS;                                 // [2][IB]
    uint32_t* hist_all = reinterpret_cast<uint32_t*>(hn + 2 * IB);  // [4][256]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wq = a.wq, wi_n = 4 / wq;
    const int wqi = wave % wq, wii = wave / wq;
    const int qblock = blockIdx.x % a.n_qblocks, split = blockIdx.x / a.n_qblocks;
    const int list = split * wi_n + wii;
    const int64_t qw0 = int64_t(qblock) * (wq * 32 * TQ) + int64_t(wqi) * 32 * TQ;  // wave's first query
    uint32_t* hist = hist_all + wave * 256;

    // ---- resident query fragments (MFMA B operand): lane (r, h) holds Q[qw0 + 32 b + r][8 g + 4 h .. +3] -------
    float4 qf[TQ][NG];
    float tau[TQ];
    int cnt[TQ];
    uint64_t* sbuf[TQ];
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = qw0 + 32 * b + r;
        const bool live = q < a.B;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            qf[b][g] = live ? *reinterpret_cast<const float4*>(a.Q + q * a.ldx + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
        tau[b] = live ? -INFINITY : INFINITY;  // strict '>' filter: dead queries never append
        cnt[b] = 0;
        sbuf[b] = a.cand + ((int64_t(list) * a.bq + q) * 2 + h) * CAPH;
    }

    const int64_t i_begin = int64_t(split) * a.split_items;
    const int64_t i_end = std::min<int64_t>(i_begin + a.split_items, a.n_items);
    const int n_tiles = int((i_end - i_begin + IB - 1) / IB);

    float4 st[NST];
    float st_h = 0.f;
    auto tile_load = [&](int t) {
        const int64_t base = i_begin + int64_t(t) * IB;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + 256 * i;
            const int row = e / (2 * NG), c = e % (2 * NG);
            const int64_t gi = std::min<int64_t>(base + row, a.n_items - 1);
            st[i] = *reinterpret_cast<const float4*>(a.X + gi * a.ldx + 4 * c);
        }
        if (tid < IB) {
            const int64_t gi = base + tid;
            st_h = gi < i_end ? a.hnorm[gi] : INFINITY;  // +inf half-norm: key = -inf, never passes
        }
    };
    auto tile_store = [&](int buf) {
        float* xb = xs + buf * IB * RS;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + 256 * i;
            const int row = e / (2 * NG), c = e % (2 * NG);
            *reinterpret_cast<float4*>(xb + row * RS + 4 * c) = st[i];
        }
        if (tid < IB) hn[buf * IB + tid] = st_h;
    };

    if (n_tiles > 0) {
        tile_load(0);
        tile_store(0);
    }
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < n_tiles) tile_load(t + 1);
        const float* xb = xs + buf * IB * RS;
        const float* hb = hn + buf * IB;
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            if ((t * S + s) % wi_n != wii) continue;  // wave-uniform: item subtiles are dealt round-robin over wi waves
            const int64_t i0 = i_begin + int64_t(t) * IB + 32 * s;
            if (i0 >= i_end) continue;
            f32x16 acc[TQ];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + 32 * s + 8 * rg + 4 * h);
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    acc[b][4 * rg + 0] = -hv.x;
                    acc[b][4 * rg + 1] = -hv.y;
                    acc[b][4 * rg + 2] = -hv.z;
                    acc[b][4 * rg + 3] = -hv.w;
                }
            }
            const float* arow = xb + (32 * s + r) * RS + 4 * h;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(arow + 8 * g);
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, qf[b][g].x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, qf[b][g].y, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, qf[b][g].z, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, qf[b][g].w, acc[b], 0, 0, 0);
                }
            }
            // ---- streaming filter: append keys > tau to this lane's stream ------------------------------------
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                float m = acc[b][0];
#pragma unroll
                for (int q = 1; q < 16; ++q) m = fmaxf(m, acc[b][q]);
                if (__any(m > tau[b])) {
                    const uint32_t item0 = uint32_t(i0) + 4 * h;
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float v = acc[b][q];
                        if (v > tau[b]) {
                            sbuf[b][cnt[b]] = pack_key(v, item0 + (q & 3) + 8 * (q >> 2));
                            ++cnt[b];
                        }
                    }
                }
                // ---- keep every stream at least 16 slots from its end --------------------------------------
                uint64_t need = __ballot(cnt[b] > CAPH - 16);
                while (need != 0ull) {
                    const int jj = (__builtin_ctzll(need)) & 31;
                    const int c0 = __shfl(cnt[b], jj, kWave), c1 = __shfl(cnt[b], jj + 32, kWave);
                    uint64_t* s0 = a.cand + ((int64_t(list) * a.bq + (qw0 + 32 * b + jj)) * 2) * CAPH;
                    uint32_t T;
                    int n0, n1;
                    compact_query(s0, s0 + CAPH, c0, c1, a.kprime, hist, T, n0, n1);
                    if (r == jj) {
                        tau[b] = f32_from_orderable(T);
                        cnt[b] = h == 0 ? n0 : n1;
                    }
                    need &= ~((1ull << jj) | (1ull << (jj + 32)));
                }
            }
        }
        if (t + 1 < n_tiles) tile_store(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = qw0 + 32 * b + r;
        a.cand_cnt[(int64_t(list) * a.bq + q) * 2 + h] = cnt[b];
    }
}

// =====================================================================================================
// host side: plan + launch
// =====================================================================================================
ScreenPlan make_screen_plan(int64_t B, int64_t n_items, int ldx, int k) {
    ScreenPlan p{};
    p.ng = ldx / 8;
    p.kprime = ((k + SLACK + 63) / 64) * 64;
    // queries per wave: TQ = 2 subtiles while the B fragments fit the register budget (NG * TQ float4 <= 32 float4)
    p.tq = (p.ng <= 16 && B > 32) ? 2 : 1;
    const int qpw = 32 * p.tq;
    p.wq = B > 2 * qpw ? 4 : (B > qpw ? 2 : 1);
    p.qb = p.wq * qpw;
    p.n_qblocks = int(cdiv64(B, p.qb));
    p.bq = int64_t(p.n_qblocks) * p.qb;
    const int wi = 4 / p.wq;
    // item splits: aim at >= 2 workgroups per CU-slot (256 CUs x 2 resident), but keep >= 2048 items per list
    const int64_t target_wgs = 1024;
    int64_t splits = std::max<int64_t>(1, target_wgs / p.n_qblocks);
    const int64_t max_splits = std::max<int64_t>(1, n_items / (2048 * wi));
    splits = std::min(splits, max_splits);
    const int ib = p.ng < 16 ? 64 : 32;
    int64_t per = cdiv64(cdiv64(n_items, splits), ib) * ib;
    splits = cdiv64(n_items, per);
    p.n_splits = int(splits);
    p.split_items = per;
    p.n_lists = p.n_splits * wi;
    return p;
}

template <int NG, int TQ, int IB>
static int launch_screen_t(const ScreenPlan& plan, const ScreenArgs& a, hipStream_t stream) {
    constexpr int RS = NG * 8 + 4;
    const size_t lds = size_t(2) * IB * RS * 4 + size_t(2) * IB * 4 + 4 * 256 * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen_kernel<NG, TQ, IB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_screen: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int grid = plan.n_qblocks * plan.n_splits;
    hipLaunchKernelGGL((screen_kernel<NG, TQ, IB>), dim3(grid), dim3(256), lds, stream, a);
    return check_launch("knn_screen");
}

int launch_screen(const ScreenPlan& plan, const float* X, const float* hnorm, int64_t n_items, int64_t ldx,
                  const float* Q32, int64_t B, uint64_t* cand, int32_t* cand_cnt, hipStream_t stream) {
    ScreenArgs a{X, hnorm, n_items, ldx, Q32, B, cand, cand_cnt, plan.n_qblocks, plan.wq, plan.split_items, plan.bq,
                 plan.kprime};
    const int key = plan.ng * 10 + plan.tq;
    switch (key) {
        case 41: return launch_screen_t<4, 1, 64>(plan, a, stream);
        case 42: return launch_screen_t<4, 2, 64>(plan, a, stream);
        case 81: return launch_screen_t<8, 1, 64>(plan, a, stream);
        case 82: return launch_screen_t<8, 2, 64>(plan, a, stream);
        case 161: return launch_screen_t<16, 1, 32>(plan, a, stream);
        case 162: return launch_screen_t<16, 2, 32>(plan, a, stream);
        case 321: return launch_screen_t<32, 1, 32>(plan, a, stream);
        default: return fail(HWER_EINVAL, "knn_screen: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
    }
}

}  // namespace hwer
