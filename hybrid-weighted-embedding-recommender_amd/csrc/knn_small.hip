// Small-batch path of the exact k-NN (K5): up to ~1000 queries against a catalogue too large for the
// dense select.  The problem is HBM-bound (the catalogue streams once per block of 32 * TQ queries), so:
//
//   1. small_screen_kernel   every WAVE streams its own item tiles HBM -> LDS by LDS-DMA (wave-private double buffer,
//                            no workgroup barrier anywhere), computes key = q.x - 0.5 ||x||^2 with the fp32 MFMA and
//                            stores the keys of the live queries densely: keys[q][item].
//   2. small_threshold_kernel per query: the r-th largest of a strided SAMPLE of its key row -> a threshold that leaves
//                            about 4.75 K' keys (K' <= count <= list capacity with overwhelming probability).
//   3. small_collect_kernel   all workgroups scan the key rows and append keys >= threshold to the query's list.
//   4. select_dense_kernel<.., LIST> (knn_select.hip) per query (one wave): k-th key of the list, float64 re-rank,
//                            certificate, sort.
// A list that ends short of K' keys (or overflows) is not certified: the query goes to the exhaustive pass - the
// sampling can cost time, never exactness.
#include <algorithm>

#include "knn_internal.h"

namespace hwer {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Seeded mode of the small-batch screens (FILTER = true): instead of storing every key, a wave compares its keys with the
// query's seeded threshold tau0 (knn_screen.hip: seed pre-pass) and hands the few that pass (~4 K' per query over the whole
// catalogue) to the query's candidate list.  Passing keys are first parked in a wave-private LDS buffer - the wave is the only
// writer, so its fill count is a wave-uniform scalar, no atomics - and flushed in batches: the flush needs a RETURNING global
// atomic per entry (the list position), whose wait drains the wave's DMA prefetch ring, so it must be rare.
constexpr int SMALL_PARK = 256;   // entries (u64 key + i32 query) a wave parks before it flushes
struct SmallSink {
    const float* tau0;   // (B) seeded thresholds, real units
    uint64_t* list;      // (B, SMALL_LIST_CAP)
    int32_t* count;      // (B) appended keys (may exceed the capacity: overflow -> the query is not certified)
};

__device__ __forceinline__ void small_park_flush(const SmallSink& sk, const uint64_t* pk, const int32_t* pq, int n) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < n; i += 64) {
        const int32_t q = pq[i];
        const int pos = atomicAdd(&sk.count[int64_t(q) * COUNT_STRIDE], 1);
        if (pos < SMALL_LIST_CAP) sk.list[int64_t(q) * SMALL_LIST_CAP + pos] = pk[i];
    }
}

// keys of one tile vs the lane's thresholds; `scaled` keys are brought back to real units with 2^-e (exact) when they are parked
template <int TQ, bool SCALED>
__device__ __forceinline__ void small_filter_tile(const SmallSink& sk, const __attribute__((ext_vector_type(16))) float (&acc)[TQ], const float (&tau)[TQ],
                                                  const int (&e_tot)[TQ], int64_t q0, int r, int h, int64_t i0, uint64_t* pk, int32_t* pq, int& n_park) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool pass = acc[b][q] > tau[b];     // strict, like the streaming filter; dead queries carry tau = +inf
            const uint64_t bal = __ballot(pass);
            if (bal != 0ull) {                          // wave-uniform, rare (~4 K' / n_items per key)
                if (n_park + 64 > SMALL_PARK) {         // wave-uniform
                    wave_lds_sync();
                    small_park_flush(sk, pk, pq, n_park);
                    wave_lds_sync();
                    n_park = 0;
                }
                if (pass) {
                    const int pos = n_park + __builtin_popcountll(bal & lt_mask);
                    const float kv = SCALED ? ldexpf(acc[b][q], -e_tot[b]) : acc[b][q];
                    pk[pos] = pack_key(kv, uint32_t(i0) + uint32_t((q & 3) + 8 * (q >> 2) + 4 * h));
                    pq[pos] = int32_t(q0 + 32 * b + r);
                }
                n_park += __builtin_popcountll(bal);
            }
        }
    }
}

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
    SmallSink sink;   // FILTER kernels
};

// One wave = one stream of tiles through a wave-private ring of NSLOT tiles (NSLOT - 1 tiles in flight while one is
// computed: with a single tile in flight the kernel is bound by HBM latency, 16 KB per wave per ~4 us).  NW waves per
// workgroup only share the CU (LDS budget: NW * NSLOT tiles <= 160 KB).
template <int NG, int TQ, int NW, int NSLOT, bool FILTER = false>
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
    if (blockIdx.x == 0 && blockIdx.y == 0) hdr_reset(a.hdr, tid, 64 * NW);   // empty uncertified list, idle tickets
    constexpr int WAVE_LDS = NSLOT * TILE_BYTES + NSLOT * IB * 4 + (FILTER ? SMALL_PARK * 12 : 0);
    float* ring = reinterpret_cast<float*>(smem_raw + size_t(wave) * WAVE_LDS);   // [NSLOT][IB][CH*4] wave-private
    float* hn = ring + NSLOT * IB * CH * 4;                                        // [NSLOT][IB] half norms
    uint64_t* pk = reinterpret_cast<uint64_t*>(hn + NSLOT * IB);                    // FILTER: parked keys | their queries
    int32_t* pq = reinterpret_cast<int32_t*>(pk + SMALL_PARK);
    int n_park = 0;
    const int64_t q0 = int64_t(blockIdx.y) * (32 * TQ);

    float4 qf[TQ][NG];
    float tau[TQ];
    int e_zero[TQ];
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = q0 + 32 * b + r;
        const bool live = q < a.B;
        tau[b] = (FILTER && live) ? a.sink.tau0[q] : INFINITY;
        e_zero[b] = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            qf[b][g] = reinterpret_cast<const float4*>(a.Q)[((q0 / 32 + b) * NG + g) * 64 + lane];   // fragment order (repack_queries)
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
#pragma unroll
    for (int j = 0; j < NSLOT - 1; ++j)   // fill the pipeline (DMA pieces past the last tile are simply not issued)
        if (t + j * n_waves < a.n_tiles) tile_dma(t + j * n_waves, j);
    int slot = 0;
    for (; t < a.n_tiles; t += n_waves, slot = (slot + 1 == NSLOT ? 0 : slot + 1)) {
        const int64_t i0 = int64_t(t) * IB;
        const int t_pre = t + (NSLOT - 1) * n_waves;
        const int slot_pre = slot == 0 ? NSLOT - 1 : slot - 1;   // the slot computed in the previous iteration
        if (t_pre < a.n_tiles) {
            tile_dma(t_pre, slot_pre);
            // Loads retire in issue order: if any piece of this tile's DMA were still pending, so would be all
            // (NSLOT - 1) (DMA_PER_TILE + 1) younger pieces.  (Key stores in flight only make the wait more conservative.)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * (DMA_PER_TILE + 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pipeline drain: the last NSLOT - 1 tiles
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
        if constexpr (FILTER) {
            small_filter_tile<TQ, false>(a.sink, acc, tau, e_zero, q0, r, h, i0, pk, pq, n_park);
        } else {
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
    if constexpr (FILTER) {
        wave_lds_sync();
        small_park_flush(a.sink, pk, pq, n_park);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// float16 storage variant of the screen: the same wave-private streaming, but the item rows come from a
// float16 copy of the table (half the HBM bytes, the bound of this path) and the tile product runs on
// v_mfma_f32_32x32x16_f16 (float32 accumulation).  Rows and queries are scaled by powers of two before the rounding
// (sx for the table, 2^qexp[b] per query) so that no element falls into float16's subnormal range; the accumulator is
// seeded with -hnorm * scale and rescaled at the end (exact).  The keys then differ from the float32 keys by at most
// ~2^-10 ||q|| ||x|| (two float16 roundings per product): the select stage widens its error bound accordingly, re-ranks
// from the float32 rows in float64 as always, and the results stay bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_32x32x16_f16: lane (r, h) holds k = 8 h + 0..7 of a 16-wide k-step

template <int NG>
__device__ __forceinline__ int swz16(int row) {   // 16-byte chunk swizzle of a row of NG chunks (2 NG bytes x 8)
    if constexpr (NG >= 16) return row & 15;
    else if constexpr (NG == 8) return (row >> 1) & 7;
    else return (row >> 2) & 3;
}

struct Small16Args {
    const _Float16* X16;   // (Np, ldx) float16, scaled by 2^x_exp
    const float* hnorm;
    int64_t ldx;
    const half8_t* Qf16;   // query fragments (k-steps of 16 columns), scaled per query by 2^qexp[q]
    const int32_t* qexp;   // (bq)
    int x_exp;
    int64_t B;
    float* keys;
    int64_t ld;
    int n_tiles;
    WsHeader* hdr;
    SmallSink sink;   // FILTER kernels
};

template <int NG, int TQ, int NW, int NSLOT, bool FILTER = false>
__global__ __launch_bounds__(64 * NW, 1) void small_screen16_kernel(const Small16Args a) {
    constexpr int IB = 32;
    constexpr int TILE_BYTES = IB * NG * 16;         // rows of NG 16-byte chunks
    constexpr int DMA_PER_TILE = IB * NG / 64;       // 1-KiB pieces
    constexpr int PIECE_ROWS = 64 / NG;              // NG <= 32: at least two rows per piece
    static_assert(NG >= 4 && NG <= 32, "rows of 64 .. 512 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    if (blockIdx.x == 0 && blockIdx.y == 0) hdr_reset(a.hdr, tid, 64 * NW);   // empty uncertified list, idle tickets
    constexpr int WAVE_LDS = NSLOT * TILE_BYTES + NSLOT * IB * 4 + (FILTER ? SMALL_PARK * 12 : 0);
    unsigned char* ring = smem_raw + size_t(wave) * WAVE_LDS;                              // [NSLOT][IB][NG * 16 B] wave-private
    float* hn = reinterpret_cast<float*>(ring + NSLOT * TILE_BYTES);                       // [NSLOT][IB] half norms
    uint64_t* pk = reinterpret_cast<uint64_t*>(hn + NSLOT * IB);                            // FILTER: parked keys | their queries
    int32_t* pq = reinterpret_cast<int32_t*>(pk + SMALL_PARK);
    int n_park = 0;
    const int64_t q0 = int64_t(blockIdx.y) * (32 * TQ);

    constexpr int KS = NG / 2;   // k-steps of 16 columns (one 16-byte chunk per lane half)
    half8_t qf[TQ][KS];
    int e_tot[TQ];
    float tau[TQ];
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
#pragma unroll
        for (int g = 0; g < KS; ++g) qf[b][g] = a.Qf16[((q0 / 32 + b) * KS + g) * 64 + lane];
        e_tot[b] = a.x_exp + a.qexp[q0 + 32 * b + r];
        const int64_t q = q0 + 32 * b + r;
        tau[b] = (FILTER && q < a.B) ? ldexpf(a.sink.tau0[q], e_tot[b]) : INFINITY;   // the filter compares in the query's scaled units
    }
    const int n_waves = gridDim.x * NW;
    const uint32_t row_bytes = uint32_t(a.ldx) * 2u;
    const uint32_t lane_row = uint32_t(lane / NG), lane_chunk = uint32_t(lane % NG);
    auto tile_dma = [&](int t, int slot) {
        const char* tile_ptr = reinterpret_cast<const char*>(a.X16 + int64_t(t) * IB * a.ldx);   // wave-uniform
        const uint32_t base = __builtin_amdgcn_readfirstlane(lds_addr(ring + slot * TILE_BYTES));
        uint32_t lr = lane_row, lc = lane_chunk;
        asm volatile("" : "+v"(lr), "+v"(lc));
#pragma unroll
        for (int p = 0; p < DMA_PER_TILE; ++p) {
            const uint32_t row = uint32_t(p * PIECE_ROWS) + lr;
            const uint32_t off = __umul24(row, row_bytes) + ((lc ^ uint32_t(swz16<NG>(int(row)))) << 4);
            lds_dma16(tile_ptr, off, base + uint32_t(p) * 1024u);
        }
        if (lane < IB) lds_dma4(a.hnorm + int64_t(t) * IB, uint32_t(lane) * 4u, __builtin_amdgcn_readfirstlane(lds_addr(hn + slot * IB)));
    };

    int t = blockIdx.x * NW + wave;
#pragma unroll
    for (int j = 0; j < NSLOT - 1; ++j)
        if (t + j * n_waves < a.n_tiles) tile_dma(t + j * n_waves, j);
    int slot = 0;
    for (; t < a.n_tiles; t += n_waves, slot = (slot + 1 == NSLOT ? 0 : slot + 1)) {
        const int64_t i0 = int64_t(t) * IB;
        const int t_pre = t + (NSLOT - 1) * n_waves;
        const int slot_pre = slot == 0 ? NSLOT - 1 : slot - 1;
        if (t_pre < a.n_tiles) {
            tile_dma(t_pre, slot_pre);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * (DMA_PER_TILE + 1)) : "memory");   // see small_screen_kernel
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char* xb = ring + slot * TILE_BYTES;
        const float* hb = hn + slot * IB;
        f32x16 acc[TQ];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const float4 hv = *reinterpret_cast<const float4*>(hb + 8 * rg + 4 * h);
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                acc[b][4 * rg + 0] = -ldexpf(hv.x, e_tot[b]);
                acc[b][4 * rg + 1] = -ldexpf(hv.y, e_tot[b]);
                acc[b][4 * rg + 2] = -ldexpf(hv.z, e_tot[b]);
                acc[b][4 * rg + 3] = -ldexpf(hv.w, e_tot[b]);
            }
        }
        // A fragment of k-step g (columns 16 g .. 16 g + 15): logical chunk 2 g + h of row r, stored at chunk (2 g + h) ^ swz16(r)
        int sw = swz16<NG>(r);
        asm volatile("" : "+v"(sw));
        const unsigned char* abase = xb + r * (NG * 16);
        auto a_ptr = [&](int g) { return abase + 16 * ((2 * g + h) ^ sw); };
        half8_t av_next = *reinterpret_cast<const half8_t*>(a_ptr(0));
#pragma unroll
        for (int g = 0; g < KS; ++g) {
            const half8_t av = av_next;
            if (g + 1 < KS) av_next = *reinterpret_cast<const half8_t*>(a_ptr(g + 1));
#pragma unroll
            for (int b = 0; b < TQ; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, qf[b][g], acc[b], 0, 0, 0);
        }
        if constexpr (FILTER) {
            small_filter_tile<TQ, true>(a.sink, acc, tau, e_tot, q0, r, h, i0, pk, pq, n_park);
        } else {
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            float* row = a.keys + (q0 + 32 * b + r) * a.ld + i0 + 4 * h;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *reinterpret_cast<float4*>(row + 8 * rg) = make_float4(ldexpf(acc[b][4 * rg], -e_tot[b]), ldexpf(acc[b][4 * rg + 1], -e_tot[b]),
                                                                      ldexpf(acc[b][4 * rg + 2], -e_tot[b]), ldexpf(acc[b][4 * rg + 3], -e_tot[b]));
        }
        }
    }
    if constexpr (FILTER) {
        wave_lds_sync();
        small_park_flush(a.sink, pk, pq, n_park);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Up to four queries (round 3): the filtering pass as a plain matrix-vector product.  With one query a 32 x 32 MFMA tile uses
// one column in 32 and the tile ring costs 9 LDS-DMA requests + a ring wait per 8 KB - the pass ran at 4.4 TB/s (73 us for the
// 320 MB of a 1.25 M x 128 float16 table).  Here every lane loads 16 bytes (8 float16 columns) of a row straight into registers
// - NG lanes per row, 64 / NG rows per load instruction, 32 rows = NG / 2 loads in flight per wave while the previous 32 are
// reduced (v_dot2_f32_f16 on the lane's 8 columns, a butterfly over the row's lanes), many waves per CU.  Keys are in the same
// scaled units as the MFMA kernels' (query scaled by 2^qexp, table by 2^x_exp, float32 accumulation) and are unscaled when parked.
// Same sink as small_screen16_kernel<.., FILTER = true>: keys above tau0[b] go to the query's list.
// ---------------------------------------------------------------------------------------------------------------------
// sum over the NG (8 / 16 / 32) consecutive lanes that hold one row, result in every one of them
template <int NG>
__device__ __forceinline__ float row_lanes_sum(float v) {
    v += __uint_as_float(lane_xor32<1>(__float_as_uint(v)));
    v += __uint_as_float(lane_xor32<2>(__float_as_uint(v)));
    v += __uint_as_float(lane_xor32<4>(__float_as_uint(v)));
    if constexpr (NG >= 16) v += __uint_as_float(lane_xor32<8>(__float_as_uint(v)));
    if constexpr (NG >= 32) v += __uint_as_float(lane_xor32<16>(__float_as_uint(v)));
    return v;
}
struct Gemv16Args {
    const _Float16* X16;   // (Np, ldx) float16, scaled by 2^x_exp; Np a multiple of 64 (pad rows: zeros, half norm +inf)
    const float* hnorm;
    int64_t ldx;
    const float* Q32;      // (B, ldx) float32 queries
    const int32_t* qexp;   // (>= B) per-query scale exponents (repack_queries16)
    int x_exp;
    int B;                 // 1 .. NB
    int64_t n_blocks;      // 32-row blocks of the padded table
    WsHeader* hdr;
    SmallSink sink;
};
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
#ifndef HWER_GEMV_DOUBLE_BUFFER
#define HWER_GEMV_DOUBLE_BUFFER 0
#endif
constexpr bool GEMV_DOUBLE_BUFFER = HWER_GEMV_DOUBLE_BUFFER != 0;
#ifndef HWER_GEMV_MAX_B
#define HWER_GEMV_MAX_B 2   // queries up to which the matrix-vector filtering pass is taken (1.25 M x 128, ms per search vs the MFMA ring: 1 query 0.121 -> 0.098, 2 queries 0.123 -> 0.110; 3 - 4 queries 0.123 - 0.125, the ring 0.112 - 0.115)
#endif
#ifndef HWER_GEMV_WAVES_PER_CU
#define HWER_GEMV_WAVES_PER_CU 48
#endif
template <int NG, int NB>
__global__ __launch_bounds__(256) void small_gemv16_kernel(const Gemv16Args a) {
    constexpr int NL = NG / 2;          // 1-KiB loads per 32-row block
    constexpr int RPL = 64 / NG;        // rows per load instruction
    static_assert(NG == 8 || NG == 16 || NG == 32, "rows of 128 / 256 / 512 bytes");
    __shared__ uint64_t s_pk[4][SMALL_PARK];
    __shared__ int32_t s_pq[4][SMALL_PARK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0) hdr_reset(a.hdr, tid, 256);   // empty uncertified list, idle tickets
    uint64_t* pk = s_pk[wave];
    int32_t* pq = s_pq[wave];
    int n_park = 0;
    const int c = lane % NG, rl = lane / NG;   // the lane's 16-byte chunk of a row, its row within a load
    // the lane's 8 columns of every query, scaled and rounded exactly as repack_queries16 does
    half2_t qh[NB][4];
    float tau[NB], nscale[NB];
    int e_tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bool live = b < a.B;
        const int e = live ? a.qexp[b] : 0;
        e_tot[b] = a.x_exp + e;
        nscale[b] = -ldexpf(1.0f, e_tot[b]);
        tau[b] = live ? ldexpf(a.sink.tau0[b], e_tot[b]) : INFINITY;   // the filter compares in the query's scaled units
        const float* qrow = a.Q32 + int64_t(live ? b : 0) * a.ldx + 8 * c;
        const float4 x = *reinterpret_cast<const float4*>(qrow), y = *reinterpret_cast<const float4*>(qrow + 4);
        qh[b][0] = half2_t{(_Float16)ldexpf(x.x, e), (_Float16)ldexpf(x.y, e)};
        qh[b][1] = half2_t{(_Float16)ldexpf(x.z, e), (_Float16)ldexpf(x.w, e)};
        qh[b][2] = half2_t{(_Float16)ldexpf(y.x, e), (_Float16)ldexpf(y.y, e)};
        qh[b][3] = half2_t{(_Float16)ldexpf(y.z, e), (_Float16)ldexpf(y.w, e)};
    }
    const int64_t n_waves = int64_t(gridDim.x) * 4;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t row_bytes = uint32_t(a.ldx) * 2u;
    auto load_block = [&](int64_t blk, uint4 (&v)[NL], float& hn) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.X16) + blk * 32 * int64_t(row_bytes);
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const uint4*>(base + uint32_t(i * RPL + rl) * row_bytes + uint32_t(c) * 16u);
        hn = a.hnorm[blk * 32 + (lane & 31)];
    };
    int64_t blk = int64_t(blockIdx.x) * 4 + wave;
    uint4 cur[NL], nxt[GEMV_DOUBLE_BUFFER ? NL : 1];
    float hn_cur = 0.f, hn_nxt = 0.f;
    if (GEMV_DOUBLE_BUFFER && blk < a.n_blocks) load_block(blk, cur, hn_cur);
    for (; blk < a.n_blocks; blk += n_waves) {
        const int64_t nb = blk + n_waves;
        if constexpr (GEMV_DOUBLE_BUFFER) {
            if (nb < a.n_blocks) load_block(nb, nxt, hn_nxt);
        } else {
            load_block(blk, cur, hn_cur);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = i * RPL + rl;                       // row of this lane within the block
            const float hn = __shfl(hn_cur, row, kWave);
            const half2_t* xv = reinterpret_cast<const half2_t*>(&cur[i]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float dot = __builtin_amdgcn_fdot2(xv[0], qh[b][0], 0.f, false);
                dot = __builtin_amdgcn_fdot2(xv[1], qh[b][1], dot, false);
                dot = __builtin_amdgcn_fdot2(xv[2], qh[b][2], dot, false);
                dot = __builtin_amdgcn_fdot2(xv[3], qh[b][3], dot, false);
                dot = row_lanes_sum<NG>(dot);   // the row's NG lanes (DPP / permlane exchanges: no LDS round trips)
                const float key_s = fmaf(hn, nscale[b], dot);
                const bool pass = c == 0 && key_s > tau[b];      // one lane per row decides (pad rows: half norm +inf -> -inf)
                const uint64_t bal = __ballot(pass);
                if (bal != 0ull) {                                // wave-uniform, rare
                    if (n_park + 64 > SMALL_PARK) {
                        wave_lds_sync();
                        small_park_flush(a.sink, pk, pq, n_park);
                        wave_lds_sync();
                        n_park = 0;
                    }
                    if (pass) {
                        const int pos = n_park + __builtin_popcountll(bal & lt_mask);
                        pk[pos] = pack_key(ldexpf(key_s, -e_tot[b]), uint32_t(blk * 32 + row));
                        pq[pos] = b;
                    }
                    n_park += __builtin_popcountll(bal);
                }
            }
        }
        if constexpr (GEMV_DOUBLE_BUFFER) {
#pragma unroll
            for (int i = 0; i < NL; ++i) cur[i] = nxt[i];
            hn_cur = hn_nxt;
        }
    }
    wave_lds_sync();
    small_park_flush(a.sink, pk, pq, n_park);
}

// ONE query, seed pre-pass (round 3): the sample keys by the same matrix-vector product, straight from Q32 - the kernel derives the query's
// float16 scale itself (and leaves it in qexp[0] for the filtering pass), so the repacking launch in front of the pre-pass is gone.
// keys[j] (float32, real units) = key of sample row j; sample tile p (32 rows) = catalogue tile p * tile_stride.
struct SeedGemvArgs {
    const _Float16* X16;
    const float* hnorm;
    int64_t ldx;
    const float* Q32;
    int x_exp;
    int64_t n_tiles;       // sample tiles
    int64_t tile_stride;
    float* keys;           // (waves x rows-per-load) candidates: every row slot of every wave hands over the largest sample key it saw (the
                           // seed pre-pass's top-M scheme with M = 1: the r-th largest candidate is the r-th largest sample key unless two of the r best
                           // share a slot - then it is slightly lower, the safe side)
    int32_t* qexp;         // [0] <- the query's scale exponent
    WsHeader* hdr;
};
template <int NG>
__global__ __launch_bounds__(256) void small_seedgemv_kernel(const SeedGemvArgs a) {
    constexpr int NL = NG / 2, RPL = 64 / NG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0) hdr_reset(a.hdr, tid, 256);
    const int c = lane % NG, rl = lane / NG;
    const float* qrow = a.Q32 + 8 * c;
    const float4 x = *reinterpret_cast<const float4*>(qrow), y = *reinterpret_cast<const float4*>(qrow + 4);
    float m = fmaxf(fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))), fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))));
#pragma unroll
    for (int off = 1; off < NG; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));   // the row's NG lanes: the whole (padded) query
    int e = 0;
    if (m > 0.f && m < INFINITY) {   // as repack_queries16
        int me;
        frexpf(m, &me);
        e = 14 - me;
    }
    if (blockIdx.x == 0 && tid == 0) a.qexp[0] = e;
    const int e_tot = a.x_exp + e;
    const float nscale = -ldexpf(1.0f, e_tot);
    half2_t qh[4];
    qh[0] = half2_t{(_Float16)ldexpf(x.x, e), (_Float16)ldexpf(x.y, e)};
    qh[1] = half2_t{(_Float16)ldexpf(x.z, e), (_Float16)ldexpf(x.w, e)};
    qh[2] = half2_t{(_Float16)ldexpf(y.x, e), (_Float16)ldexpf(y.y, e)};
    qh[3] = half2_t{(_Float16)ldexpf(y.z, e), (_Float16)ldexpf(y.w, e)};
    const uint32_t row_bytes = uint32_t(a.ldx) * 2u;
    const int64_t n_waves = int64_t(gridDim.x) * 4;
    float best = -INFINITY;
    for (int64_t p = int64_t(blockIdx.x) * 4 + wave; p < a.n_tiles; p += n_waves) {
        const int64_t row0 = p * a.tile_stride * 32;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.X16) + row0 * int64_t(row_bytes);
        uint4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const uint4*>(base + uint32_t(i * RPL + rl) * row_bytes + uint32_t(c) * 16u);
        const float hn_l = a.hnorm[row0 + (lane & 31)];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int row = i * RPL + rl;
            const float hn = __shfl(hn_l, row, kWave);
            const half2_t* xv = reinterpret_cast<const half2_t*>(&v[i]);
            float dot = __builtin_amdgcn_fdot2(xv[0], qh[0], 0.f, false);
            dot = __builtin_amdgcn_fdot2(xv[1], qh[1], dot, false);
            dot = __builtin_amdgcn_fdot2(xv[2], qh[2], dot, false);
            dot = __builtin_amdgcn_fdot2(xv[3], qh[3], dot, false);
            dot = row_lanes_sum<NG>(dot);
            best = fmaxf(best, fmaf(hn, nscale, dot));   // (pad rows: -inf; scaled units: one power of two for the whole query)
        }
    }
    if (c == 0) a.keys[(int64_t(blockIdx.x) * 4 + wave) * RPL + rl] = ldexpf(best, -e_tot);
}

// queries -> float16 MFMA fragments, each query scaled by the power of two that puts its largest element in
// [2^13, 2^14).  One 256-thread block per 32-query subtile: 8 threads scan a query row for its maximum.
__global__ __launch_bounds__(256) void repack_queries16_kernel(const float* __restrict__ Q, int64_t B, int ldx, int ng, half8_t* __restrict__ Qf16,
                                                              int32_t* __restrict__ qexp) {
    __shared__ int s_exp[32];
    const int64_t s = blockIdx.x;
    const int ql = threadIdx.x >> 3, part = threadIdx.x & 7;
    const int64_t q = s * 32 + ql;
    float m = 0.f;
    if (q < B)
        for (int j = part; j < ldx; j += 8) m = fmaxf(m, fabsf(Q[q * ldx + j]));
    m = fmaxf(m, __shfl_xor(m, 1, kWave));
    m = fmaxf(m, __shfl_xor(m, 2, kWave));
    m = fmaxf(m, __shfl_xor(m, 4, kWave));
    if (part == 0) {
        int e = 0;
        if (m > 0.f && m < INFINITY) {
            int me;
            frexpf(m, &me);        // m = f * 2^me, f in [0.5, 1)  ->  m * 2^(14 - me) in [2^13, 2^14)
            e = 14 - me;
        }
        s_exp[ql] = e;
        qexp[q] = e;               // qexp has bq >= 32 * gridDim.x entries
    }
    __syncthreads();
    // fragment order of v_mfma_f32_32x32x16_f16: Qf16[(s * KS + g) * 64 + lane] = Q[32 s + (lane & 31)][16 g + 8 (lane >> 5) .. + 7]
    const int ks = ng / 2;
    for (int o = threadIdx.x; o < ks * 64; o += 256) {
        const int g = o >> 6, lane = o & 63;
        const int64_t qq = s * 32 + (lane & 31);
        half8_t v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)0.f;
        if (qq < B) {
            const float* src = Q + qq * ldx + 16 * g + 8 * (lane >> 5);
            const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
            const int e = s_exp[lane & 31];
            v[0] = (_Float16)ldexpf(x.x, e); v[1] = (_Float16)ldexpf(x.y, e); v[2] = (_Float16)ldexpf(x.z, e); v[3] = (_Float16)ldexpf(x.w, e);
            v[4] = (_Float16)ldexpf(y.x, e); v[5] = (_Float16)ldexpf(y.y, e); v[6] = (_Float16)ldexpf(y.z, e); v[7] = (_Float16)ldexpf(y.w, e);
        }
        Qf16[(s * ks + g) * 64 + lane] = v;
    }
}

__global__ __launch_bounds__(256) void pack_rows16_kernel(const float4* __restrict__ X, int64_t n4, int x_exp, half4_t* __restrict__ X16) {
    for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n4; i += int64_t(gridDim.x) * 256) {
        const float4 x = X[i];
        half4_t v;
        v[0] = (_Float16)ldexpf(x.x, x_exp); v[1] = (_Float16)ldexpf(x.y, x_exp); v[2] = (_Float16)ldexpf(x.z, x_exp); v[3] = (_Float16)ldexpf(x.w, x_exp);
        X16[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
#ifndef HWER_SEED_SLACK_DIV
#define HWER_SEED_SLACK_DIV 6   // seed thresholds: the bisection stops with rank .. rank + rank / DIV sample keys above the threshold
                                // (measured 2 -> 6: one more probe, ~12 % fewer appended keys: c3 -3.7 %, c4 -1 %; 12 = 6)
#endif
struct SmallSelectArgs {
    const float* keys;  // (seed pre-pass: float16 keys behind this pointer, KT = _Float16 in the threshold kernels)
    int64_t ld;         // multiple of 32 (pad items hold -inf)
    int64_t n_items;
    int64_t B;
    int kprime;
    float* thr;         // (B)
    int32_t* count;     // (B) list lengths (may exceed the capacity: overflow)
    uint64_t* list;     // (B, SMALL_LIST_CAP) packed keys
    double want_frac;   // > 0: the threshold's rank in the sample = want_frac * sample size (seed pre-pass of the streaming screen);
                        // 0: about 4.75 K' keys of the ROW clear the threshold (small-batch path)
};

// threshold of one query from a strided sample of its key row: the r-th largest sample key, r chosen so that about
// 4.75 K' keys of the row clear it (never fewer than 16 sample keys: the row count then has a relative spread <= 25 %)
// A threshold taken from float16 keys is lowered by their rounding error (2^-11 relative): it must not rise above the float32 key
// it stands for, or the catalogue would hold fewer keys above it than the sample rank promises.
template <typename KT>
__device__ __forceinline__ float seed_margin(float t) {
    if constexpr (std::is_same<KT, float>::value) return t;
    else return t - fabsf(t) * 0x1p-10f - 1e-7f;
}
template <int PER, typename KT = float, int NT_ = 1024>   // samples per thread: 16 (16 384-key sample), 32 or 64 (65 536): a probe is PER compares + one block reduction
__global__ __launch_bounds__(NT_) void small_threshold_kernel(const SmallSelectArgs a) {   // (NT_ = 256: one query's ~10 K sample keys - four waves reduce faster than sixteen)
    constexpr int NT = NT_, NWV = NT / 64;
    __shared__ int s_cnt[NWV];
    __shared__ float s_mm[2 * NWV];
    const int64_t b = blockIdx.x;
    const KT* row = reinterpret_cast<const KT*>(a.keys) + b * a.ld;
    const int64_t n = a.n_items;
    const int64_t S = n < int64_t(NT) * PER ? n : int64_t(NT) * PER;
    const double step = double(n) / double(S);
    const int wave = threadIdx.x >> 6;
    float v[PER];
    if (S == n) {   // the whole row is the sample (seed pre-pass; short rows): 16-byte loads - the row is padded to 32 keys (pad keys -inf)
        constexpr int VEC = 16 / int(sizeof(KT));
        static_assert(PER % VEC == 0, "whole vectors per thread");
#pragma unroll
        for (int m = 0; m < PER / VEC; ++m) {
            const int64_t j0 = (int64_t(threadIdx.x) + NT * m) * VEC;
            if (j0 < a.ld) {
                if constexpr (std::is_same<KT, float>::value) {
                    const float4 x = *reinterpret_cast<const float4*>(row + j0);
                    v[VEC * m + 0] = x.x; v[VEC * m + 1] = x.y; v[VEC * m + 2] = x.z; v[VEC * m + 3] = x.w;
                } else {
                    const half8_t x = *reinterpret_cast<const half8_t*>(row + j0);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[VEC * m + j] = float(x[j]);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (j0 + j >= n) v[VEC * m + j] = -INFINITY;
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[VEC * m + j] = -INFINITY;
            }
        }
    } else if (!std::is_same<KT, float>::value) {
        // float16 rows longer than the sample (seed pre-pass of a multi-million-row catalogue): the sample is taken as 16-byte
        // segments (8 keys) at a regular stride - scalar 2-byte loads at a stride of 3 keys made this kernel 0.85 ms on the
        // 10 M-row catalogue (219 K sample keys per query), as long as the pre-pass that wrote them
        constexpr int VEC = 8;
        const int64_t n_seg = n / VEC;                      // whole segments in the row (n is a multiple of 32)
        const int64_t s_seg = S / VEC;
        const double step_seg = double(n_seg) / double(s_seg);
#pragma unroll
        for (int m = 0; m < PER / VEC; ++m) {
            const int64_t sidx = int64_t(threadIdx.x) + NT * m;
            const int64_t j0 = min(int64_t(double(sidx) * step_seg), n_seg - 1) * VEC;
            const half8_t x = *reinterpret_cast<const half8_t*>(row + j0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[VEC * m + j] = sidx < s_seg ? float(x[j]) : -INFINITY;
        }
    } else {
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int64_t j = int64_t(threadIdx.x) + NT * m;
        v[m] = j < S ? float(row[min(int64_t(double(j) * step), n - 1)]) : -INFINITY;
    }
    }
    float vmax = -INFINITY, vmin = INFINITY;
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        vmax = fmaxf(vmax, v[m]);
        if (v[m] > -INFINITY) vmin = fminf(vmin, v[m]);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        vmax = fmaxf(vmax, __shfl_xor(vmax, off, kWave));
        vmin = fminf(vmin, __shfl_xor(vmin, off, kWave));
    }
    if ((threadIdx.x & 63) == 0) { s_mm[wave] = vmax; s_mm[NWV + wave] = vmin; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        vmax = fmaxf(vmax, s_mm[w]);
        vmin = fminf(vmin, s_mm[NWV + w]);
    }
    __syncthreads();
    auto count_ge = [&](float t) {   // block-uniform result
        int c = 0;
#pragma unroll
        for (int m = 0; m < PER; ++m) c += __builtin_popcountll(__ballot(v[m] >= t));
        if ((threadIdx.x & 63) == 0) s_cnt[wave] = c;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += s_cnt[w];
        __syncthreads();
        return tot;
    };
    // sample rank: expected row count 4.75 K' (>= K' with overwhelming probability, <= the list capacity)
    const double want = a.want_frac > 0.0 ? a.want_frac * double(S) : 4.75 * double(a.kprime) * double(S) / double(n);
    int rank = int(want + 0.5);
    if (rank < 16) rank = 16;
    if (rank > S) rank = int(S);
    int c_lo;
    // (seed pre-pass: any threshold with at least `rank` sample keys above it will do - a wide stopping window saves probes)
    const float t = wave_find_rank_fn(count_ge, vmin, vmax, rank, a.want_frac > 0.0 ? rank / HWER_SEED_SLACK_DIV : rank / 8, c_lo);
    if (threadIdx.x == 0) {
        a.thr[b] = seed_margin<KT>(t);
        a.count[b * COUNT_STRIDE] = 0;
    }
}

// Same threshold by ONE wave per query (no workgroup barriers) from a sample of 64 * PER keys taken as 16-key segments
// (64 contiguous bytes each) spread evenly over the row: a strided single-key sample would touch nearly every cache line
// of a short row.  Enough whenever the sample rank stays >= 16.  Four queries per 256-thread block.
template <int PER, typename KT = float>   // keys per lane: 16 (1024-key sample) or 64 (4096-key sample)
__global__ __launch_bounds__(256) void small_threshold_wave_kernel(const SmallSelectArgs a) {
    constexpr int NSEG = 64 * PER / 16;
    const int lane = threadIdx.x & 63;
    const int64_t b = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b >= a.B) return;
    const KT* row = reinterpret_cast<const KT*>(a.keys) + b * a.ld;
    const int64_t n = a.n_items;      // rows are padded to a multiple of 32 keys (pad keys are -inf): segments stay in the row
    const double step = n > 16 ? double(n - 16) / double(NSEG - 1) : 0.0;
    float v[PER];
    int64_t n_sample = 0;
#pragma unroll
    for (int m = 0; m < PER / 16; ++m) {
        const int seg = lane + 64 * m;
        if constexpr (std::is_same<KT, float>::value) {
            const int64_t p0 = (int64_t(double(seg) * step) >> 2) << 2;   // 16-byte aligned start
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 x = *reinterpret_cast<const float4*>(row + p0 + 4 * c);
                v[16 * m + 4 * c + 0] = x.x; v[16 * m + 4 * c + 1] = x.y; v[16 * m + 4 * c + 2] = x.z; v[16 * m + 4 * c + 3] = x.w;
            }
        } else {
            const int64_t p0 = (int64_t(double(seg) * step) >> 3) << 3;   // 16-byte aligned start (8 float16 keys)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const half8_t x = *reinterpret_cast<const half8_t*>(row + p0 + 8 * c);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[16 * m + 8 * c + j] = float(x[j]);
            }
        }
        n_sample += 16;
    }
    const double S = double(64 * PER);   // segments may overlap on rows shorter than the sample: the rank only gets safer
    const double want = a.want_frac > 0.0 ? a.want_frac * S : 4.75 * double(a.kprime) * S / double(n > 64 * PER ? n : 64 * PER);
    int rank = int(want + 0.5);
    if (rank < 16) rank = 16;
    if (rank > 64 * PER) rank = 64 * PER;
    int c_lo;
    const float t = wave_find_rank<PER>(v, rank, a.want_frac > 0.0 ? rank / HWER_SEED_SLACK_DIV : rank / 8, c_lo);
    if (lane == 0) {
        a.thr[b] = seed_margin<KT>(t);
        a.count[b * COUNT_STRIDE] = 0;
    }
}

// every workgroup scans 256 * 64 keys of one query's row; the keys >= threshold are gathered in LDS and appended to the
// query's list with ONE reservation per workgroup (per-wave atomics on a handful of counters serialise in L2)
__global__ __launch_bounds__(256) void small_collect_kernel(const SmallSelectArgs a) {
    constexpr int LCAP = 1024;   // per-block staging capacity; beyond it the block falls back to direct appends
    __shared__ uint64_t s_keys[LCAP];
    __shared__ int s_n, s_base;
    const int64_t b = blockIdx.y;
    const float* row = a.keys + b * a.ld;
    const float thr = a.thr[b];
    const int lane = threadIdx.x & 63;
    const int64_t base4 = int64_t(blockIdx.x) * 256 * 16;   // float4 index of this block's first slice
    const int64_t n4 = a.ld / 4;
    uint64_t* list = a.list + b * SMALL_LIST_CAP;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
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
                if (lane == 0) pos0 = atomicAdd(&s_n, n_in);   // LDS atomic
                pos0 = __shfl(pos0, 0, kWave);
                const int pos = pos0 + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (in) {
                    const uint64_t key = pack_key(vv[c], uint32_t(4 * i4 + c));
                    if (pos < LCAP) {
                        s_keys[pos] = key;
                    } else {   // staging full (massive ties): append directly
                        const int g = atomicAdd(&a.count[b * COUNT_STRIDE], 1);
                        if (g < SMALL_LIST_CAP) list[g] = key;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int n = min(s_n, LCAP);
    if (threadIdx.x == 0) s_base = n > 0 ? atomicAdd(&a.count[b * COUNT_STRIDE], n) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if (s_base + i < SMALL_LIST_CAP) list[s_base + i] = s_keys[i];
}

// =====================================================================================================
static size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

bool use_small_path(int64_t B, int64_t n_items, uint32_t options) {
    if (options & HWER_KNN_NO_SMALL) return false;
    if (n_items <= DENSE_MAX_ITEMS || B * n_items * 4 > SMALL_MAX_KEY_BYTES) return false;
    if (const char* e = diag_env("HWER_SMALL_MAXQ")) return B <= atoi(e);   // tuning override
    if (B <= SMALL_MAX_QUERIES) return true;
    // With seeded thresholds a candidate list costs next to nothing, so the streaming screen splits the items as finely as it
    // needs to fill the machine and the dense-key detour below no longer pays (c3's last 7 421 users: 1.0 -> 0.5 ms).
    if (!(options & HWER_KNN_NO_SEED)) return false;
    // Beyond that the streaming screen wins once its launch can fill the machine: 512 workgroup slots, one workgroup per
    // 256 queries and item split, lists of at least 4096 items.  A launch that cannot (e.g. the last 7 421 of C3's users
    // against 26 744 items: 29 query blocks x 6 splits) is 1.8x faster through the dense-key path.
    const int64_t n_qb = cdiv64(B, 256);
    const int64_t splits = std::min<int64_t>(std::max<int64_t>(1, 512 / n_qb), std::max<int64_t>(1, n_items / 4096));
    return n_qb * splits < 384;
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

struct SmallLayout { size_t qf, qexp, keys, thr, count, list, total; };
// `seed` (nullable): the screen plan whose sample pre-pass seeds the thresholds; then `keys` holds the SAMPLE keys
// (seed->bq x seed->seed_ld) instead of every key, and the fragment areas cover the larger of the two query paddings
static SmallLayout small_layout(const SmallPlan& p, const ScreenPlan* seed) {
    SmallLayout w{};
    const size_t bq = size_t(seed ? std::max<int64_t>(p.bq, seed->bq) : p.bq);
    size_t off = 0;
    w.qf = off;    off += al256(bq * size_t(p.ng) * 8 * 4);   // queries in fragment order (bq is a multiple of 32); float16 variant: half of it
    w.qexp = off;  off += al256(bq * 4);                      // per-query scale exponents of the float16 variant
    w.keys = off;  off += al256(seed ? seed_keys_bytes(*seed) /* float16 sample keys | the lanes' best candidates */ : size_t(p.bq) * size_t(p.ld) * 4);
    w.thr = off;   off += al256(bq * 4);
    w.count = off; off += al256(bq * 4 * COUNT_STRIDE);   // one counter per 128 bytes
    w.list = off;  off += al256(size_t(p.bq) * SMALL_LIST_CAP * 8);
    w.total = off;
    return w;
}
size_t small_ws_bytes(const SmallPlan& p, const ScreenPlan* seed) { return small_layout(p, seed).total; }

template <int NG, int TQ, int NW, int NSLOT>
static int launch_small_screen(const SmallPlan& p, const SmallArgs& a, hipStream_t stream) {
    const bool filter = a.sink.tau0 != nullptr;
    const size_t lds = size_t(NW) * (NSLOT * (32 * NG * 32 + 32 * 4) + (filter ? SMALL_PARK * 12 : 0));
    static_assert(size_t(NW) * (NSLOT * (32 * NG * 32 + 32 * 4) + SMALL_PARK * 12) <= 160 * 1024, "ring must fit the CU's LDS");
    const void* fn = filter ? reinterpret_cast<const void*>(&small_screen_kernel<NG, TQ, NW, NSLOT, true>)
                            : reinterpret_cast<const void*>(&small_screen_kernel<NG, TQ, NW, NSLOT, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_small: hipFuncSetAttribute: %s", hipGetErrorString(e));
    // one workgroup per CU (LDS-limited); query blocks share the CUs: grid.x * grid.y ~ 256
    const int gx = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(a.n_tiles, NW), std::max(1, 256 / p.n_qblocks))));
    if (filter) hipLaunchKernelGGL((small_screen_kernel<NG, TQ, NW, NSLOT, true>), dim3(gx, p.n_qblocks), dim3(64 * NW), lds, stream, a);
    else hipLaunchKernelGGL((small_screen_kernel<NG, TQ, NW, NSLOT, false>), dim3(gx, p.n_qblocks), dim3(64 * NW), lds, stream, a);
    return check_launch("knn_small(screen)");
}

template <int NG, int TQ, int NW, int NSLOT>
static int launch_small_screen16(const SmallPlan& p, const Small16Args& a, hipStream_t stream) {
    const bool filter = a.sink.tau0 != nullptr;
    const size_t lds = size_t(NW) * (NSLOT * (32 * NG * 16 + 32 * 4) + (filter ? SMALL_PARK * 12 : 0));
    static_assert(size_t(NW) * (NSLOT * (32 * NG * 16 + 32 * 4) + SMALL_PARK * 12) <= 160 * 1024, "ring must fit the CU's LDS");
    const void* fn = filter ? reinterpret_cast<const void*>(&small_screen16_kernel<NG, TQ, NW, NSLOT, true>)
                            : reinterpret_cast<const void*>(&small_screen16_kernel<NG, TQ, NW, NSLOT, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_small16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const int gx = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(a.n_tiles, NW), std::max(1, 256 / p.n_qblocks))));
    if (filter) hipLaunchKernelGGL((small_screen16_kernel<NG, TQ, NW, NSLOT, true>), dim3(gx, p.n_qblocks), dim3(64 * NW), lds, stream, a);
    else hipLaunchKernelGGL((small_screen16_kernel<NG, TQ, NW, NSLOT, false>), dim3(gx, p.n_qblocks), dim3(64 * NW), lds, stream, a);
    return check_launch("knn_small(screen16)");
}

int repack_queries16(const float* Q32, int64_t B, int64_t bq, int ldx, void* Qf16, int32_t* qexp, hipStream_t stream) {
    hipLaunchKernelGGL(repack_queries16_kernel, dim3(unsigned(bq / 32)), dim3(256), 0, stream, Q32, B, ldx, ldx / 8,
                       reinterpret_cast<half8_t*>(Qf16), qexp);
    return check_launch("knn_repack_queries16");
}

int f16_scale_exp(float hnorm_max) {
    // largest power of two that keeps every element of a row (|x_i| <= ||x|| <= sqrt(2 hnorm_max)) below 2^14 after scaling
    const float xmax = sqrtf(2.0f * hnorm_max) * 1.0001f;
    if (!(xmax > 0.f) || !(xmax < INFINITY)) return 0;
    int me;
    frexpf(xmax, &me);   // xmax = f 2^me, f in [0.5, 1)
    return 14 - me;
}

int pack_rows16(const float* X, int64_t n_rows_padded, int64_t ldx, int x_exp, void* X16, hipStream_t stream) {
    const int64_t n4 = n_rows_padded * ldx / 4;
    if (n4 == 0) return HWER_OK;
    const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(n4, 256), 4096)));
    hipLaunchKernelGGL(pack_rows16_kernel, dim3(grid), dim3(256), 0, stream, reinterpret_cast<const float4*>(X), n4, x_exp,
                       reinterpret_cast<half4_t*>(X16));
    return check_launch("pack_rows16");
}

int launch_seed_threshold(const float* keys, int64_t ld, int64_t n, int64_t B, double want_frac, float* tau0, int32_t* count_scratch,
                          hipStream_t stream) {
    // the seed pre-pass stores its sample keys as float16 (launch_seed_screen)
    using H = _Float16;
    SmallSelectArgs sa{keys, ld, n, B, 0, tau0, count_scratch, nullptr, want_frac};
    if (n <= 1024) hipLaunchKernelGGL((small_threshold_wave_kernel<16, H>), dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);
    else if (n <= 2048) hipLaunchKernelGGL((small_threshold_wave_kernel<32, H>), dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);
    else if (n <= 3072) hipLaunchKernelGGL((small_threshold_wave_kernel<48, H>), dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);   // (c3: 2 464 sample keys per query)
    else if (n <= 4096) hipLaunchKernelGGL((small_threshold_wave_kernel<64, H>), dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);
    else if (n <= 16384) hipLaunchKernelGGL((small_threshold_kernel<16, H>), dim3(unsigned(B)), dim3(1024), 0, stream, sa);
    else if (n <= 32768) hipLaunchKernelGGL((small_threshold_kernel<32, H>), dim3(unsigned(B)), dim3(1024), 0, stream, sa);
    else hipLaunchKernelGGL((small_threshold_kernel<64, H>), dim3(unsigned(B)), dim3(1024), 0, stream, sa);
    return check_launch("knn_seed(threshold)");
}

// tau0 of one query = the `rank`-th largest of its n <= 64 * PER candidates (the lanes' top-M of the seed pre-pass), one wave per
// query, four queries per block; fewer finite candidates than `rank` (tiny samples): the smallest of them
// (round 5: the same pick over the GROUP MAXIMA of a query, stored transposed - candidate i of query b at cand[i * ld_i + b * ld]; the lane
// candidates above: ld_i = 1)
template <int PER>
__global__ __launch_bounds__(256) void seed_pick_kernel(const float* __restrict__ cand, int64_t ld, int n, int64_t B, int rank,
                                                        float* __restrict__ tau0, int32_t* __restrict__ count, int64_t ld_i) {
    const int lane = threadIdx.x & 63;
    const int64_t b = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = cand + b * ld;
    float v[PER];
    int fin = 0;
#pragma unroll
    for (int m = 0; m < PER; ++m) {
        const int i = lane + 64 * m;
        v[m] = i < n ? row[int64_t(i) * ld_i] : -INFINITY;
        fin += __builtin_popcountll(__ballot(v[m] > -INFINITY));
    }
    float t = -INFINITY;   // no finite candidate: open threshold
    if (fin > 0) {
        const int rk = min(rank, fin);
        int c_lo;
        t = wave_find_rank<PER>(v, rk, rk / HWER_SEED_SLACK_DIV, c_lo);
    }
    if (lane == 0) {
        tau0[b] = t;
        count[b * COUNT_STRIDE] = 0;
    }
}

// Group maxima (round 5): tau0 with ONE LANE PER QUERY.  The (n, bq) matrix is stored transposed, so lane q of a wave reads row i of its
// query with the 63 neighbouring queries in one 256-byte segment; the n <= NV values stay in registers and the lane bisects on the VALUE for a
// threshold with rank .. rank + slack values at or above it (what wave_find_rank returns for a wave-per-query pick): n compares + adds per
// probe, ~10 probes, no cross-lane traffic, no LDS - the wave-per-query forms of this pick (column-strided reads: 95 us; through an LDS tile:
// 101 us for 131 072 queries x 154 groups) spent their time in the dependent probes of one query after the other.
template <int NV>
__global__ __launch_bounds__(256) void seed_pick_gmax_lane_kernel(const float* __restrict__ gm, int64_t bq, int n, int64_t B, int rank,
                                                                  float* __restrict__ tau0, int32_t* __restrict__ count) {
    const int64_t q = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t qc = q < bq ? q : bq - 1;      // (whole rows of bq queries are in bounds; lanes past B compute and do not store)
    float v[NV];
    float vmax = -INFINITY, vmin = INFINITY;
    int fin = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = i < n ? gm[int64_t(i) * bq + qc] : -INFINITY;
        vmax = fmaxf(vmax, v[i]);
        if (v[i] > -INFINITY) { vmin = fminf(vmin, v[i]); ++fin; }
    }
    float t = -INFINITY;   // no finite value: open threshold
    if (fin > 0) {
        const int rk = min(rank, fin), slack = rk / HWER_SEED_SLACK_DIV;
        auto count_ge = [&](float x) {
            int c = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) c += v[i] >= x ? 1 : 0;
            return c;
        };
        float lo = vmin, hi = vmax;        // count(>= lo) >= rk always; count(>= hi) < rk after the first test
        int c_lo = fin;
        if (count_ge(hi) >= rk) {
            lo = hi;                        // the rk-th largest IS the maximum (ties at the top)
        } else {
            for (int it = 0; it < 40 && c_lo > rk + slack; ++it) {
                float mid = 0.5f * lo + 0.5f * hi;
                if (!(mid > lo) || !(mid < hi)) break;   // adjacent floats: the interval is exhausted by ties
                const int c = count_ge(mid);
                if (c >= rk) { lo = mid; c_lo = c; } else { hi = mid; }
            }
        }
        t = lo;
    }
    if (q < B) {
        tau0[q] = t;
        count[q * COUNT_STRIDE] = 0;
    }
}

void small_outputs(const SmallPlan& p, const ScreenPlan* seed, unsigned char* ws, uint64_t** list_out, int32_t** cnt_out) {
    const SmallLayout w = small_layout(p, seed);
    *list_out = reinterpret_cast<uint64_t*>(ws + w.list);
    *cnt_out = reinterpret_cast<int32_t*>(ws + w.count);
}

size_t seed_keys_bytes(const ScreenPlan& plan) {
    if (plan.seed_gmax) return size_t(plan.seed_groups) * size_t(plan.bq) * 4;   // group maxima, (seed_groups, bq) float32
    return plan.seed_top_m ? size_t(plan.bq) * size_t(plan.seed_cand_ld) * 4 : size_t(plan.bq) * size_t(plan.seed_ld) * 2;   // float32 candidates | float16 sample keys
}

int launch_seed_tau0(const ScreenPlan& plan, const float* seed_keys, int64_t B, float* tau0, int32_t* count_scratch, hipStream_t stream) {
    if (!plan.seed_top_m && !plan.seed_gmax) return launch_seed_threshold(seed_keys, plan.seed_ld, plan.seed_ld, B, plan.seed_frac, tau0, count_scratch, stream);
    // lane candidates: row b = candidates of query b; group maxima: column b of a (seed_groups, bq) matrix
    const int n = plan.seed_gmax ? plan.seed_groups : plan.seed_cand_n;
    const int64_t ld_b = plan.seed_gmax ? 1 : plan.seed_cand_ld, ld_i = plan.seed_gmax ? plan.bq : 1;
    if (plan.seed_gmax && n <= 224) {   // one lane per query, the values in registers
        const dim3 g(unsigned(cdiv64(B, 256)));
        if (n <= 64) hipLaunchKernelGGL(seed_pick_gmax_lane_kernel<64>, g, dim3(256), 0, stream, seed_keys, plan.bq, n, B, plan.seed_rank, tau0, count_scratch);
        else if (n <= 160) hipLaunchKernelGGL(seed_pick_gmax_lane_kernel<160>, g, dim3(256), 0, stream, seed_keys, plan.bq, n, B, plan.seed_rank, tau0, count_scratch);
        else hipLaunchKernelGGL(seed_pick_gmax_lane_kernel<224>, g, dim3(256), 0, stream, seed_keys, plan.bq, n, B, plan.seed_rank, tau0, count_scratch);
        return check_launch("knn_seed(pick gmax)");
    }
    const dim3 grid(unsigned(cdiv64(B, 4)));
    if (n <= 256) hipLaunchKernelGGL(seed_pick_kernel<4>, grid, dim3(256), 0, stream, seed_keys, ld_b, n, B, plan.seed_rank, tau0, count_scratch, ld_i);
    else if (n <= 1024) hipLaunchKernelGGL(seed_pick_kernel<16>, grid, dim3(256), 0, stream, seed_keys, ld_b, n, B, plan.seed_rank, tau0, count_scratch, ld_i);
    else hipLaunchKernelGGL(seed_pick_kernel<64>, grid, dim3(256), 0, stream, seed_keys, ld_b, n, B, plan.seed_rank, tau0, count_scratch, ld_i);
    return check_launch("knn_seed(pick)");
}

int launch_small(const SmallPlan& p, const float* X, const float* hnorm, int64_t n_items, int64_t ldx, const float* Q32, int64_t B,
                 unsigned char* ws, uint64_t** list_out, int32_t** cnt_out, WsHeader* hdr, const void* X16, int x_exp, hipStream_t stream,
                 const ScreenPlan* seed) {
    const SmallLayout w = small_layout(p, seed);
    const int64_t bq_all = seed ? std::max<int64_t>(p.bq, seed->bq) : p.bq;   // rows of the fragment areas (zero beyond B)
    float* thr = reinterpret_cast<float*>(ws + w.thr);
    int32_t* count = reinterpret_cast<int32_t*>(ws + w.count);
    uint64_t* list = reinterpret_cast<uint64_t*>(ws + w.list);
    SmallSink sink{nullptr, list, count};
    int rc;
    const bool f16 = X16 != nullptr;
    const bool seed16 = f16 && seed && screen_f16_supported(*seed);   // the seed pre-pass reads the float16 copy when its kernel can
    float* qf = reinterpret_cast<float*>(ws + w.qf);
    int32_t* qexp = reinterpret_cast<int32_t*>(ws + w.qexp);
    // one query (matrix-vector passes, round 3): the pre-pass reads Q32 itself and leaves the query's scale in qexp[0] - no repacking launch;
    // its sample keys are float32, one row of seed_ld keys, and the block threshold kernel picks tau0 from them
    const bool gemv = f16 && seed && B <= HWER_GEMV_MAX_B && B == 1 && (p.ng == 8 || p.ng == 16 || p.ng == 32) && seed_keys_bytes(*seed) >= 4096 && !diag_env("HWER_SMALL_NO_GEMV");
    if (gemv) {
        float* seed_keys = reinterpret_cast<float*>(ws + w.keys);   // (>= 32 rows x seed_ld float16 were provisioned: one float32 row fits)
        SeedGemvArgs sg{reinterpret_cast<const _Float16*>(X16), hnorm, ldx, Q32, x_exp, seed->seed_tiles, seed->seed_stride, seed_keys, qexp, hdr};
        // at most 1 024 candidates (one wave picks tau0 from 16 per lane): waves x (64 / NG) row slots
        const int rpl = 64 / p.ng;
        const int n_wv = int(std::max<int64_t>(4, std::min<int64_t>(seed->seed_tiles, 1024 / rpl)));
        const int grid = (n_wv + 3) / 4;
        const int n_cand = grid * 4 * rpl;
        if (p.ng == 8) hipLaunchKernelGGL(small_seedgemv_kernel<8>, dim3(grid), dim3(256), 0, stream, sg);
        else if (p.ng == 16) hipLaunchKernelGGL(small_seedgemv_kernel<16>, dim3(grid), dim3(256), 0, stream, sg);
        else hipLaunchKernelGGL(small_seedgemv_kernel<32>, dim3(grid), dim3(256), 0, stream, sg);
        if ((rc = check_launch("knn_small(seed gemv)"))) return rc;
        // tau0 = the rank-th largest candidate, rank = the sample rank of the plan (the threshold kernel of the full sample took 10.7 us for
        // 9 766 keys; this pick 6)
        const int rank = std::max(16, int(seed->seed_frac * double(seed->seed_ld) + 0.5));
        if (n_cand <= 256) hipLaunchKernelGGL(seed_pick_kernel<4>, dim3(1), dim3(256), 0, stream, seed_keys, int64_t(n_cand), n_cand, int64_t(1), rank, thr, count, int64_t(1));
        else hipLaunchKernelGGL(seed_pick_kernel<16>, dim3(1), dim3(256), 0, stream, seed_keys, int64_t(n_cand), n_cand, int64_t(1), rank, thr, count, int64_t(1));
        if ((rc = check_launch("knn_small(seed pick)"))) return rc;
        sink.tau0 = thr;
    } else if (f16) {
        hipLaunchKernelGGL(repack_queries16_kernel, dim3(unsigned(bq_all / 32)), dim3(256), 0, stream, Q32, B, int(ldx), p.ng,
                           reinterpret_cast<half8_t*>(qf), qexp);
        if ((rc = check_launch("knn_small(repack16)"))) return rc;
    } else if ((rc = repack_queries(Q32, B, bq_all, int(ldx), qf, stream))) {
        return rc;
    }
    if (seed && !gemv) {
        // Seeded mode: sample pre-pass (the dense mode of the streaming screen over every seed_stride-th tile) -> tau0 per query
        // -> ONE pass over the catalogue that keeps only the keys above tau0.  No key matrix, no collect pass.
        float* seed_keys = reinterpret_cast<float*>(ws + w.keys);
        if (f16 && !seed16) return fail(HWER_EINVAL, "knn_small: the seeded float16 path needs a padded dim >= 64 (caller drops X16 below that)");
        ScreenF16 sf{X16, qf, qexp, x_exp};
        if ((rc = launch_seed_screen(*seed, X, hnorm, n_items, ldx, qf, B, seed_keys, hdr, stream, seed16 ? &sf : nullptr))) return rc;
        // the threshold kernel also zeroes count[b] for the queries it handles (b < B)
        if ((rc = launch_seed_tau0(*seed, seed_keys, B, thr, count, stream))) return rc;
        sink.tau0 = thr;
    }
    if (gemv || (f16 && seed && B <= HWER_GEMV_MAX_B && (p.ng == 8 || p.ng == 16 || p.ng == 32) && !diag_env("HWER_SMALL_NO_GEMV"))) {
        // a handful of queries: the filtering pass as a matrix-vector product (small_gemv16_kernel)
        const int64_t n_pad = ((n_items + ROW_PAD - 1) / ROW_PAD) * ROW_PAD;
        Gemv16Args ga{reinterpret_cast<const _Float16*>(X16), hnorm, ldx, Q32, qexp, x_exp, int(B), n_pad / 32, hdr, sink};
        // 4 waves per block, ~16 waves per CU: every wave keeps NG / 2 KiB in flight
        const int grid = int(std::max<int64_t>(1, std::min<int64_t>(cdiv64(ga.n_blocks, 4), int64_t(HWER_GEMV_WAVES_PER_CU / 4) * 256)));
        auto go = [&](auto ng, auto nb) {
            hipLaunchKernelGGL((small_gemv16_kernel<decltype(ng)::value, decltype(nb)::value>), dim3(grid), dim3(256), 0, stream, ga);
        };
        auto by_b = [&](auto ng) {
            if (B == 1) go(ng, std::integral_constant<int, 1>{});
            else if (B == 2) go(ng, std::integral_constant<int, 2>{});
            else go(ng, std::integral_constant<int, 4>{});
        };
        if (p.ng == 8) by_b(std::integral_constant<int, 8>{});
        else if (p.ng == 16) by_b(std::integral_constant<int, 16>{});
        else by_b(std::integral_constant<int, 32>{});
        if ((rc = check_launch("knn_small(gemv16)"))) return rc;
    } else if (f16) {   // float16 storage: half the HBM bytes for the catalogue stream
        Small16Args a16{reinterpret_cast<const _Float16*>(X16), hnorm, ldx, reinterpret_cast<const half8_t*>(qf), qexp, x_exp, B,
                        reinterpret_cast<float*>(ws + w.keys), p.ld, int(p.ld / 32), hdr, sink};
        // waves x ring slots per CU.  A wave's tile is a serial chain (LDS-DMA requests ~150 cycles each, MFMAs, filter), so the
        // stream rate grows with the waves on the CU, not with the depth of a wave's ring: 8 waves x 2 slots beat 4 x 4 at d = 128
        // (one query on 1.25 M rows 0.137 -> 0.124 ms, 64 queries 0.262 -> 0.239, 512 queries 0.62 -> 0.45 ms)
        switch (p.ng * 10 + p.tq) {
            case 41: rc = launch_small_screen16<4, 1, 8, 8>(p, a16, stream); break;      // 2 KB tiles
            case 42: rc = launch_small_screen16<4, 2, 8, 8>(p, a16, stream); break;
            case 81: rc = launch_small_screen16<8, 1, 8, 4>(p, a16, stream); break;      // 4 KB tiles
            case 82: rc = launch_small_screen16<8, 2, 8, 4>(p, a16, stream); break;
            case 161: rc = launch_small_screen16<16, 1, 8, 2>(p, a16, stream); break;    // 8 KB tiles
            case 162: rc = launch_small_screen16<16, 2, 8, 2>(p, a16, stream); break;
            case 321: rc = launch_small_screen16<32, 1, 4, 2>(p, a16, stream); break;
            default: return fail(HWER_EINVAL, "knn_small16: unsupported padded dim %d / tq %d", p.ng * 8, p.tq);
        }
        if (rc) return rc;
    } else {
        SmallArgs a{X, hnorm, n_items, ldx, qf, B, reinterpret_cast<float*>(ws + w.keys), p.ld, int(p.ld / 32), hdr, sink};
        switch (p.ng * 10 + p.tq) {
            // waves x ring slots per CU: (NSLOT - 1) tiles in flight per wave, ~96 KB in flight per CU
            case 41: rc = launch_small_screen<4, 1, 8, 4>(p, a, stream); break;      // 4 KB tiles
            case 42: rc = launch_small_screen<4, 2, 8, 4>(p, a, stream); break;
            case 81: rc = launch_small_screen<8, 1, 8, 2>(p, a, stream); break;      // 8 KB tiles
            case 82: rc = launch_small_screen<8, 2, 8, 2>(p, a, stream); break;
            case 161: rc = launch_small_screen<16, 1, 3, 3>(p, a, stream); break;    // 16 KB tiles
            case 162: rc = launch_small_screen<16, 2, 3, 3>(p, a, stream); break;
            case 321: rc = launch_small_screen<32, 1, 2, 2>(p, a, stream); break;    // 32 KB tiles
            default: return fail(HWER_EINVAL, "knn_small: unsupported padded dim %d / tq %d", p.ng * 8, p.tq);
        }
        if (rc) return rc;
    }
    *list_out = list;
    *cnt_out = count;
    if (seed) return HWER_OK;
    SmallSelectArgs sa{reinterpret_cast<const float*>(ws + w.keys), p.ld, n_items, B, p.kprime, thr, count, list, 0.0};
    // sample rank = 4.75 K' S / N must stay >= 16 (kept >= 32 here): one wave per query with a 1024- or 4096-key sample
    if (double(n_items) * 32.0 <= 1024.0 * 4.75 * double(p.kprime))
        hipLaunchKernelGGL(small_threshold_wave_kernel<16>, dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);
    else if (double(n_items) * 32.0 <= 4096.0 * 4.75 * double(p.kprime))
        hipLaunchKernelGGL(small_threshold_wave_kernel<64>, dim3(unsigned(cdiv64(B, 4))), dim3(256), 0, stream, sa);
    else
        hipLaunchKernelGGL(small_threshold_kernel<SMALL_SAMPLE / 1024>, dim3(unsigned(B)), dim3(1024), 0, stream, sa);
    if ((rc = check_launch("knn_small(threshold)"))) return rc;
    hipLaunchKernelGGL(small_collect_kernel, dim3(unsigned(cdiv64(p.ld / 4, 256 * 16)), unsigned(B)), dim3(256), 0, stream, sa);
    return check_launch("knn_small(collect)");
}

}  // namespace hwer
