// Small-batch path of the exact k-NN (K5): a handful of queries (B <= 256) against a catalogue too large for the
// dense select.  The problem is HBM-bound (the catalogue streams once per block of 32 * TQ queries), so:
//
//   1. small_screen_kernel   every WAVE streams its own item tiles HBM -> LDS by LDS-DMA (wave-private double buffer,
//                            no workgroup barrier anywhere), computes key = q.x - 0.5 ||x||^2 with the fp32 MFMA and
//                            stores the keys of the live queries densely: keys[q][item].
//   2. small_threshold_kernel per query: the r-th largest of a strided SAMPLE of its key row -> a threshold that leaves
//                            about 4.75 K' keys (K' <= count <= list capacity with overwhelming probability).
//   3. small_collect_kernel   all workgroups scan the key rows and append keys >= threshold to the query's list.
//   4. small_final_kernel     per query (one wave): the K' best of the list -> the candidate stream read by
//                            select_rerank_kernel (float64 re-rank, certificate, sort: knn_select.hip).
// A list that ends short of K' keys (or overflows) is flagged UNDERFULL: the select kernel then refuses to certify and
// the query goes to the exhaustive pass - the sampling can cost time, never exactness.
#include <algorithm>

#include "knn_internal.h"

namespace hwer {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct SmallArgs {
    const float* X;
    const float* hnorm;
    int64_t n_items;
    int64_t ldx;
    const float* Q;
    int64_t B;
    float* keys;      // (bq, ld)
    int64_t ld;
    int n_tiles;
    WsHeader* hdr;
};

// One wave = one stream of tiles.  NW waves per workgroup only share the CU (LDS budget: NW * 2 tiles).
template <int NG, int TQ, int NW>
__global__ __launch_bounds__(64 * NW, 1) void small_screen_kernel(const SmallArgs a) {
    constexpr int IB = 32;
    constexpr int CH = 2 * NG;                    // 16-byte chunks per row
    constexpr int TILE_BYTES = IB * CH * 16;
    constexpr int DMA_PER_TILE = IB * CH / 64;    // 1-KiB LDS-DMA pieces per tile (all issued by the owning wave)
    constexpr int PIECE_ROWS = 64 / CH > 0 ? 64 / CH : 1;
    static_assert(CH <= 64, "one LDS-DMA piece must hold at least one whole row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 2 + EXACT_DEVICE_MAXQ) reinterpret_cast<int32_t*>(a.hdr)[tid] = 0;   // empty uncertified list
    float* ring = reinterpret_cast<float*>(smem_raw + size_t(wave) * (2 * TILE_BYTES + 2 * IB * 4));   // [2][IB][CH*4] wave-private
    float* hn = ring + 2 * IB * CH * 4;                                                               // [2][IB] half norms
    const int64_t q0 = int64_t(blockIdx.y) * (32 * TQ);

    float4 qf[TQ][NG];
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = q0 + 32 * b + r;
        const bool live = q < a.B;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            qf[b][g] = live ? *reinterpret_cast<const float4*>(a.Q + q * a.ldx + 8 * g + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int n_waves = gridDim.x * NW;
    const uint32_t row_bytes = uint32_t(a.ldx) * 4u;
    const uint32_t lane_row = uint32_t(lane / CH), lane_chunk = uint32_t(lane % CH);
    auto tile_dma = [&](int t, int slot) {
        const char* tile_ptr = reinterpret_cast<const char*>(a.X + int64_t(t) * IB * a.ldx);   // wave-uniform
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr(ring + slot * (IB * CH * 4)));
        uint32_t lr = lane_row, lc = lane_chunk;
        asm volatile("" : "+v"(lr), "+v"(lc));
#pragma unroll
        for (int p = 0; p < DMA_PER_TILE; ++p) {
            const uint32_t row = uint32_t(p * PIECE_ROWS) + lr;
            const uint32_t off = __umul24(row, row_bytes) + ((lc ^ uint32_t(swz<CH>(int(row)))) << 4);
            lds_dma16(tile_ptr, off, base + uint32_t(p) * 1024u);
        }
        // half norms of the tile rows: one 4-byte piece by lanes 0..31 (through LDS like the tile itself: a plain global
        // load would make the compiler drain vmcnt(0), i.e. the prefetch, before its first use)
        if (lane < IB) lds_dma4(a.hnorm + int64_t(t) * IB, uint32_t(lane) * 4u, __builtin_amdgcn_readfirstlane(lds_addr(hn + slot * IB)));
    };

    int t = blockIdx.x * NW + wave;
    if (t < a.n_tiles) tile_dma(t, 0);
    int slot = 0;
    for (; t < a.n_tiles; t += n_waves, slot ^= 1) {
        const int64_t i0 = int64_t(t) * IB;
        const int t_next = t + n_waves;
        if (t_next < a.n_tiles) {
            tile_dma(t_next, slot ^ 1);
            // vmcnt retires in issue order: everything older than the DMA pieces just issued (this tile's DMA, the previous
            // tile's key stores) has landed once at most DMA_PER_TILE + 1 operations are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE + 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float* xb = ring + slot * (IB * CH * 4);
        const float* hb = hn + slot * IB;
        f32x16 acc[TQ];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 hv = *reinterpret_cast<const float4*>(hb + 8 * rg + 4 * h);
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                acc[b][4 * rg + 0] = -hv.x;
                acc[b][4 * rg + 1] = -hv.y;
                acc[b][4 * rg + 2] = -hv.z;
                acc[b][4 * rg + 3] = -hv.w;
            }
        }
        int sw = swz<CH>(r);
        asm volatile("" : "+v"(sw));
        const int swh = (sw >> 1) & (NG >= 8 ? 7 : (NG - 1));
        const float* abase = xb + r * (CH * 4) + 4 * (h ^ (sw & 1));
        auto a_ptr = [&](int g) { return abase + 8 * ((g & 7) ^ swh) + 64 * (g >> 3); };
        float4 av_next = *reinterpret_cast<const float4*>(a_ptr(0));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float4 av = av_next;
            if (g + 1 < NG) av_next = *reinterpret_cast<const float4*>(a_ptr(g + 1));
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, qf[b][g].x, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, qf[b][g].y, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, qf[b][g].z, acc[b], 0, 0, 0);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, qf[b][g].w, acc[b], 0, 0, 0);
            }
        }
        // lane (r, h): query q0 + 32 b + r, items i0 + 8 rg + 4 h + 0..3 -> one 16-byte store each.  The key matrix has
        // bq >= B rows (whole query blocks), so every lane stores: a fixed store count keeps the vmcnt arithmetic exact.
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            float* row = a.keys + (q0 + 32 * b + r) * a.ld + i0 + 4 * h;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *reinterpret_cast<float4*>(row + 8 * rg) = make_float4(acc[b][4 * rg], acc[b][4 * rg + 1], acc[b][4 * rg + 2], acc[b][4 * rg + 3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct SmallSelectArgs {
    const float* keys;
    int64_t ld;         // multiple of 32 (pad items hold -inf)
    int64_t n_items;
    int64_t B;
    int kprime;
    float* thr;         // (B)
    int32_t* count;     // (B) list lengths (may exceed the capacity: overflow)
    uint64_t* list;     // (B, SMALL_LIST_CAP) packed keys
    uint64_t* cand;     // streams for select_rerank_kernel: (bq, 2, CAPH)
    int32_t* cand_cnt;  // (bq, 2)
};

// threshold of one query from a strided sample of its key row: the r-th largest sample key, r chosen so that about
// 4.75 K' keys of the row clear it (never fewer than 16 sample keys: the row count then has a relative spread <= 25 %)
__global__ __launch_bounds__(256) void small_threshold_kernel(const SmallSelectArgs a) {
    constexpr int PER = SMALL_SAMPLE / 256;   // samples per thread (64)
    __shared__ int s_cnt[4];
    const int64_t b = blockIdx.x;
    const float* row = a.keys + b * a.ld;
    const int64_t n = a.n_items;
    const int64_t S = n < SMALL_SAMPLE ? n : SMALL_SAMPLE;
    const double step = double(n) / double(S);
    float v[PER];
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int64_t j = int64_t(threadIdx.x) + 256 * m;
        v[m] = j < S ? row[min(int64_t(double(j) * step), n - 1)] : -INFINITY;
    }
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        vmax = fmaxf(vmax, v[m]);
        if (v[m] > -INFINITY) vmin = fminf(vmin, v[m]);
    }
    __shared__ float s_mm[8];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, off, kWave));
        vmin = fminf(vmin, __shfl_xor(vmin, off, kWave));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[threadIdx.x >> 6] = vmax; s_mm[4 + (threadIdx.x >> 6)] = vmin; }
    __syncthreads();
    vmax = fmaxf(fmaxf(s_mm[0], s_mm[1]), fmaxf(s_mm[2], s_mm[3]));
    vmin = fminf(fminf(s_mm[4], s_mm[5]), fminf(s_mm[6], s_mm[7]));
    __syncthreads();
    auto count_ge = [&](float t) {   // block-uniform result
        int c = 0;
#pragma unroll
        for (int m = 0; m < PER; ++m) c += __builtin_popcountll(__ballot(v[m] >= t));
        if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = c;
        __syncthreads();
        const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
        return tot;
    };
    // sample rank: expected row count 4.75 K' (>= K' with overwhelming probability, <= the list capacity)
    const double want = 4.75 * double(a.kprime) * double(S) / double(n);
    int rank = int(want + 0.5);
    if (rank < 16) rank = 16;
    if (rank > S) rank = int(S);
    int c_lo;
    const float t = wave_find_rank_fn(count_ge, vmin, vmax, rank, rank / 8, c_lo);
    if (threadIdx.x == 0) {
        a.thr[b] = t;
        a.count[b] = 0;
    }
}

// every workgroup scans 256 * 64 keys of one query's row and appends the keys >= threshold to the query's list
__global__ __launch_bounds__(256) void small_collect_kernel(const SmallSelectArgs a) {
    const int64_t b = blockIdx.y;
    const float* row = a.keys + b * a.ld;
    const float thr = a.thr[b];
    const int lane = threadIdx.x & 63;
    const int64_t base4 = (int64_t(blockIdx.x) * 256 * 16) ;   // float4 index of this block's first slice
    const int64_t n4 = a.ld / 4;
    uint64_t* list = a.list + b * SMALL_LIST_CAP;
#pragma unroll 4
    for (int m = 0; m < 16; ++m) {
        const int64_t i4 = base4 + int64_t(m) * 256 + threadIdx.x;
        float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (i4 < n4) v = reinterpret_cast<const float4*>(row)[i4];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in = vv[c] >= thr;
            const uint64_t bal = __ballot(in);
            if (bal != 0ull) {   // wave-uniform, rare
                const int n_in = __builtin_popcountll(bal);
                int pos0 = 0;
                if (lane == 0) pos0 = atomicAdd(&a.count[b], n_in);
                pos0 = __shfl(pos0, 0, kWave);
                const int pos = pos0 + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (in && pos < SMALL_LIST_CAP) list[pos] = pack_key(vv[c], uint32_t(4 * i4 + c));
            }
        }
    }
}

// one wave per query: the K' best keys of its list -> stream 0 of the query (stream 1 stays empty)
__global__ __launch_bounds__(64) void small_final_kernel(const SmallSelectArgs a) {
    constexpr int PER = SMALL_LIST_CAP / 64;   // 64 keys per lane
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int cnt_raw = a.count[b];
    const int n = min(cnt_raw, SMALL_LIST_CAP);
    const uint64_t* list = a.list + b * SMALL_LIST_CAP;
    float sv[PER];
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int i = lane + 64 * m;
        sv[m] = i < n ? key_score(list[i]) : -INFINITY;
    }
    uint64_t* s0 = a.cand + (b * 2) * CAPH;
    // the list must hold at least K' keys (else the threshold was too high) and must not have overflowed
    const bool bad = cnt_raw < a.kprime || cnt_raw > SMALL_LIST_CAP;
    int c_ge = 0;
    float t_k = -INFINITY;
    if (n >= a.kprime) t_k = wave_find_rank<PER>(sv, a.kprime, 0, c_ge);
    int n_out = 0;
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const bool in = sv[m] >= t_k && sv[m] > -INFINITY;
        const uint64_t bal = __ballot(in);
        const int pos = n_out + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        if (in && pos < a.kprime) s0[pos] = list[lane + 64 * m];   // ties at the cut beyond K' keys are dropped (<= every kept key)
        n_out += __builtin_popcountll(bal);
    }
    if (lane == 0) {
        a.cand_cnt[b * 2] = min(n_out, a.kprime) | (bad ? CNT_UNDERFULL : 0);
        a.cand_cnt[b * 2 + 1] = 0;
    }
}

// =====================================================================================================
static size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

bool use_small_path(int64_t B, int64_t n_items) {
    if (getenv("HWER_NO_SMALL")) return false;
    return B <= SMALL_MAX_QUERIES && n_items > DENSE_MAX_ITEMS;
}

SmallPlan make_small_plan(int64_t B, int64_t n_items, int ldx, int k) {
    SmallPlan p{};
    p.ng = ldx / 8;
    p.kprime = ((k + SLACK + 63) / 64) * 64;
    p.tq = (p.ng <= 16 && B > 32) ? 2 : 1;
    p.n_qblocks = int(cdiv64(B, 32 * p.tq));
    p.bq = int64_t(p.n_qblocks) * 32 * p.tq;
    p.ld = cdiv64(n_items, 32) * 32;
    return p;
}

struct SmallLayout { size_t keys, thr, count, list, cand, cnt, total; };
static SmallLayout small_layout(const SmallPlan& p) {
    SmallLayout w{};
    size_t off = 0;
    w.keys = off;  off += al256(size_t(p.bq) * size_t(p.ld) * 4);
    w.thr = off;   off += al256(size_t(p.bq) * 4);
    w.count = off; off += al256(size_t(p.bq) * 4);
    w.list = off;  off += al256(size_t(p.bq) * SMALL_LIST_CAP * 8);
    w.cand = off;  off += al256(size_t(p.bq) * 2 * CAPH * 8);
    w.cnt = off;   off += al256(size_t(p.bq) * 2 * 4);
    w.total = off;
    return w;
}
size_t small_ws_bytes(const SmallPlan& p) { return small_layout(p).total; }

template <int NG, int TQ, int NW>
static int launch_small_screen(const SmallPlan& p, const SmallArgs& a, hipStream_t stream) {
    const size_t lds = size_t(NW) * (2 * 32 * NG * 32 + 2 * 32 * 4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_screen_kernel<NG, TQ, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_small: hipFuncSetAttribute: %s", hipGetErrorString(e));
    // one workgroup per CU (LDS-limited); query blocks share the CUs: grid.x * grid.y ~ 256
    const int gx = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(a.n_tiles, NW), std::max(1, 256 / p.n_qblocks))));
    hipLaunchKernelGGL((small_screen_kernel<NG, TQ, NW>), dim3(gx, p.n_qblocks), dim3(64 * NW), lds, stream, a);
    return check_launch("knn_small(screen)");
}

int launch_small(const SmallPlan& p, const float* X, const float* hnorm, int64_t n_items, int64_t ldx, const float* Q32, int64_t B,
                 unsigned char* ws, uint64_t** cand_out, int32_t** cnt_out, WsHeader* hdr, hipStream_t stream) {
    const SmallLayout w = small_layout(p);
    SmallArgs a{X, hnorm, n_items, ldx, Q32, B, reinterpret_cast<float*>(ws + w.keys), p.ld, int(p.ld / 32), hdr};
    int rc;
    switch (p.ng * 10 + p.tq) {
        case 41: rc = launch_small_screen<4, 1, 4>(p, a, stream); break;
        case 42: rc = launch_small_screen<4, 2, 4>(p, a, stream); break;
        case 81: rc = launch_small_screen<8, 1, 4>(p, a, stream); break;
        case 82: rc = launch_small_screen<8, 2, 4>(p, a, stream); break;
        case 161: rc = launch_small_screen<16, 1, 4>(p, a, stream); break;
        case 162: rc = launch_small_screen<16, 2, 4>(p, a, stream); break;
        case 321: rc = launch_small_screen<32, 1, 2>(p, a, stream); break;
        default: return fail(HWER_EINVAL, "knn_small: unsupported padded dim %d / tq %d", p.ng * 8, p.tq);
    }
    if (rc) return rc;
    SmallSelectArgs sa{reinterpret_cast<const float*>(ws + w.keys), p.ld, n_items, B, p.kprime, reinterpret_cast<float*>(ws + w.thr),
                       reinterpret_cast<int32_t*>(ws + w.count), reinterpret_cast<uint64_t*>(ws + w.list),
                       reinterpret_cast<uint64_t*>(ws + w.cand), reinterpret_cast<int32_t*>(ws + w.cnt)};
    hipLaunchKernelGGL(small_threshold_kernel, dim3(unsigned(B)), dim3(256), 0, stream, sa);
    if ((rc = check_launch("knn_small(threshold)"))) return rc;
    hipLaunchKernelGGL(small_collect_kernel, dim3(unsigned(cdiv64(p.ld / 4, 256 * 16)), unsigned(B)), dim3(256), 0, stream, sa);
    if ((rc = check_launch("knn_small(collect)"))) return rc;
    hipLaunchKernelGGL(small_final_kernel, dim3(unsigned(B)), dim3(64), 0, stream, sa);
    if ((rc = check_launch("knn_small(final)"))) return rc;
    *cand_out = sa.cand;
    *cnt_out = sa.cand_cnt;
    return HWER_OK;
}

}  // namespace hwer
