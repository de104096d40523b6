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
// cross-lane traffic.  Item tiles (IB rows) stream HBM -> LDS by LDS-DMA (double buffered, one barrier per
// tile).  When a stream is about to overflow, the wave brackets the K'-th key of that query with a few counting
// probes on register-resident keys, compacts both streams in place and raises the threshold.
//
// Bound: fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles / SIMD) once a workgroup keeps >= ~50 queries resident;
// HBM (streaming X once per query block) below that.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "knn_internal.h"

namespace hwer {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));      // operand pair of v_pk_mul_f32 / v_pk_add_f32
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_32x32x16_f16

struct ScreenArgs {
    const float* X;
    const float* hnorm;
    int64_t n_items;
    int64_t ldx;
    const float* Q;      // queries in fragment order (repack_queries)
    int64_t B;
    float* dense;        // DENSE mode: every key is written to dense[q * ld_dense + item] instead of being filtered
    int64_t ld_dense;
    uint64_t* cand;
    int32_t* cand_cnt;   // a count with bit 30 set marks an UNDERFULL list (scaled threshold was too high): recompute exactly
    int n_qblocks;
    int wq;
    int64_t split_items;
    int64_t bq;
    int kprime;
    int scaled_first;    // 1: first compaction selects the sample-scaled rank (verified afterwards), 0: always K'
    int debug_nofilter;  // ablation only (HWER_SCREEN_NOFILTER=1): thresholds start at +inf, nothing is appended
    float debug_tau0;    // ablation only (HWER_SCREEN_TAU0=x): thresholds start at x instead of -inf
    unsigned long long* debug_stamps;   // diagnostic build only (HWER_SCREEN_STAMPS=1): per-wave phase cycle sums
    WsHeader* hdr;       // workspace header: the uncertified-query list of this search starts empty
    // float16 screen (F16 kernels; HWER_KNN_F16_SCREEN): the table scaled by 2^x_exp and rounded to float16, the queries in
    // fragment order scaled per query by 2^qexp[q] (knn_small.hip: repack_queries16), see small_screen16_kernel
    const void* X16;
    const void* Qf16;
    const int32_t* qexp;
    int x_exp;
    const float* tau0;   // streaming mode: per-query initial threshold from the seed pre-pass (NULL: open thresholds)
    int tile_stride;     // 1; the seed pre-pass visits every tile_stride-th tile of the catalogue (its items are sample-local)
    int dense_f16;       // DENSE mode: `dense` holds float16 keys (the seed pre-pass: its keys only place a threshold - half the bytes)
    int sparse_hits;     // streaming float16 mode: fewer than ~0.6 passing keys expected per tile and wave -> the max-tree filter (FILTER_V2)
    int many_hits;       // ping-pong kernel: >= ~8 passing keys expected per 32-row subtile and wave -> up to four pops per round of the pop loop
    int seed_m;          // DENSE mode, seed pre-pass (round 3): 0 = store every key; M in {1, 2, 4} = every lane keeps the M largest keys it
                         // sees for its query and stores only those: dense[q * ld_dense + bin * M + j], bin = (split * wi + wave_i) * 2 + lane half;
                         // -1 (round 5): the maximum of the lane's 16 keys per sample tile, transposed (see knn_internal.h: seed_gmax)
    int seed_group;      // seed_m > 0 (round 5): the lane's top-M of its GROUP MAXIMA (one candidate per 16 keys: a v_max3 tree + ONE insertion
                         // per subtile instead of 16 insertions) - the pick then takes a rank in groups
};

// sc1 load: served by L2, never by a stale L1 line
__device__ __forceinline__ uint64_t ld_u64_l2(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Selection over the two streams (s0: c0 entries, s1: c1 entries) of one query, by the whole wave.  All keys are fetched
// in ONE batch of loads (16 x 8 B per lane, a single L2 round trip) and stay in registers; a key t with
//   rank <= count(score >= t) <= rank + slack          (more only when scores tie at t)
// is bracketed by a handful of counting probes (wave_find_rank: one v_cmp per key, interpolated probe positions - a
// third of the instructions of a byte-wise radix select with an LDS histogram), then both streams are rewritten in
// place with every key > t plus as many of the keys == t as fit `rank + slack`.  t is the new strict threshold: later
// keys equal to it are dropped, which is safe - they are not better than the smallest key kept.
__device__ __forceinline__ void compact_query(uint64_t* s0, uint64_t* s1, int c0, int c1, int rank, int slack,
                                           float& t_out, int& n0_out, int& n1_out) {
    constexpr int NCH = CAPH / 64;
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's appends have reached L2
    uint64_t key[2 * NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int e = 64 * c + lane;
        key[c] = e < c0 ? ld_u64_l2(s0 + e) : 0ull;
        key[NCH + c] = e < c1 ? ld_u64_l2(s1 + e) : 0ull;
    }
    float sv[2 * NCH];
#pragma unroll
    for (int i = 0; i < 2 * NCH; ++i) sv[i] = key[i] != 0ull ? key_score(key[i]) : -INFINITY;
    int c_ge;
    const float t = wave_find_rank<2 * NCH>(sv, rank, slack, c_ge);   // the streams hold >= rank valid keys
    int c_gt = 0;
#pragma unroll
    for (int i = 0; i < 2 * NCH; ++i) c_gt += __builtin_popcountll(__ballot(sv[i] > t));
    const int want_eq = min(c_ge - c_gt, rank + slack - c_gt);   // >= 1: count(> t) < rank + slack and count(>= t) >= rank
    int eq_seen = 0;
    int out0 = 0, out1 = 0;
#pragma unroll
    for (int i = 0; i < 2 * NCH; ++i) {
        const bool gt = sv[i] > t;
        const bool eq = sv[i] == t;
        const uint64_t be = __ballot(eq);
        const int erank = eq_seen + __builtin_popcountll(be & lt_mask);
        const bool keep = gt || (eq && erank < want_eq);
        const uint64_t bk = __ballot(keep);
        const int pos = __builtin_popcountll(bk & lt_mask);
        if (i < NCH) {
            if (keep) s0[out0 + pos] = key[i];  // every load of this stream happened before the first store
            out0 += __builtin_popcountll(bk);
        } else {
            if (keep) s1[out1 + pos] = key[i];
            out1 += __builtin_popcountll(bk);
        }
        eq_seen += __builtin_popcountll(be);
    }
    t_out = t;
    n0_out = out0;
    n1_out = out1;
}

// a step near 0.618 n that is coprime with n: t -> (t * step) mod n is a permutation that scatters consecutive t
__device__ __forceinline__ int coprime_step(int n) {
    if (n <= 2) return 1;
    int s = int(0.6180339887f * float(n)) | 1;
    for (;; s += 2) {
        int x = n, y = s % n;
        while (y) { const int tt = x % y; x = y; y = tt; }
        if (x == 1) return s % n == 0 ? 1 : s % n;
    }
}

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#ifdef HWER_DIAG_STAMPS
#define DIAG_STAMPS (a.debug_stamps != nullptr)
#else
#define DIAG_STAMPS false
#endif
#define STAMP(var) if (DIAG_STAMPS) { var = stamp(); }
#ifndef HWER_FILTER_V2
#define HWER_FILTER_V2 1   // float16 streaming screen: max-tree + per-register appends instead of pass masks + LDS park + pop loop (round 3)
#endif
constexpr bool FILTER_V2 = HWER_FILTER_V2 != 0;
#ifndef HWER_WIDE64
#define HWER_WIDE64 1   // d = 256 (NG = 32): 64-row tiles too - with the park-free filter (sparse_hits forced) the 2 x 32 KB ring leaves two workgroups per CU
#endif
constexpr bool WIDE64 = HWER_WIDE64 != 0;
#ifndef HWER_F16_PREFETCH
#define HWER_F16_PREFETCH 1    // float16 tile product: A fragments requested this many k-steps ahead of their MFMAs
#endif
#ifndef HWER_F16_MFMA_PRIO
#define HWER_F16_MFMA_PRIO 0   // wave priority during the float16 MFMA phase ...
#endif
#ifndef HWER_F16_REST_PRIO
#define HWER_F16_REST_PRIO 3   // ... and during everything else (measured: 0 / 3 beats 3 / 3 by 6 % at c4)
#endif
// A/B switches of diagnostic builds (-DHWER_DIAG_STAMPS or -DHWER_DIAG_ABLATE; HWER_SCREEN_NOFILTER=<bit mask>, float16 streaming
// loop: 1 no appends, 4 no pass masks, 8 no accumulator set-up, 16 no LDS-DMA requests after the first two tiles, 32 one fragment
// read per tile instead of NQF, 64 no tile barrier, 128 no MFMAs; any kernel: 256 no XCD-aware workgroup order).  Results are
// wrong under 4 .. 128 - tools/ablate_screen.sh only reads the kernel durations.
#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
#define ABLATE(bit) ((a.debug_nofilter & (bit)) != 0)
#else
#define ABLATE(bit) false
#endif

__device__ __forceinline__ void make_opaque(float4& f) { asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w)); }
__device__ __forceinline__ void make_opaque(half8_t& f) { asm volatile("" : "+v"(f)); }

// F16: the item tiles come from the float16 copy of the table and the tile product runs on v_mfma_f32_32x32x16_f16 (float32
// accumulation, a quarter of the MFMA cycles): keys are computed in per-query scaled units (thresholds too) and unscaled
// when appended; the select stage widens its error bound by the float16 rounding (key_rel_err), re-ranks from the float32
// rows in float64 as always, so the results are unchanged.  Default for the streaming mode above 8 192 rows; opt-in for the
// dense mode (its keys are unscaled when they are stored).
template <int NG, int TQ, bool DENSE, bool F16 = false>
__global__ __launch_bounds__(256, 2) void screen_kernel(const ScreenArgs a) {
    // item rows per tile: one 32-row MFMA subtile; the float16 streaming screen takes TWO per tile (round 3) - the tile barrier, the
    // LDS-DMA issue phase (~150 cycles per 1-KiB piece in the wave's chain), the capacity check and the loop control are ~1 000 of a
    // 32-row tile's ~2 100 cycles and amortise over twice the MFMAs
    // (d = 256: a 64-row ring is 64 KB and leaves ONE workgroup per CU - measured 3.09 -> 3.96 ms at c5: 32-row tiles there)
    constexpr int IB = (F16 && !DENSE && (NG <= 16 || WIDE64)) ? SCREEN16_IB : 32;
    static_assert(IB == 32 || IB == 64, "the half-norm request (one per lane of wave 0) and the capacity check cover at most 64 rows per tile");
    constexpr int NSUB = IB / 32;
    constexpr int CH = F16 ? NG : 2 * NG;          // 16-byte chunks per row
    constexpr int TILE_CHUNKS = IB * CH;
    constexpr int DMA_PER_WAVE = TILE_CHUNKS / 256 > 0 ? TILE_CHUNKS / 256 : 1;   // 1-KiB LDS-DMA pieces per wave and tile
    static_assert(TILE_CHUNKS % 64 == 0, "tile must be a whole number of 1-KiB pieces");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);                         // [2][IB][CH*4]
    float* hn = xs + 2 * IB * CH * 4;                                       // [2][IB]
    float* keypark_all = hn + 2 * IB;                                       // [4][64][16 * TQ]: a wave's keys per lane
    // streaming mode: append staging, per wave [8 slots][TQ][64 lanes] packed keys.  A lane's stream grows by 8-byte keys a few
    // tiles apart; written one by one, every key costs a 32-byte memory write (measured: 4.3 GB written for 1.1 GB of keys at
    // c3).  Keys wait here until a group of 8 is complete and leave as ONE 64-byte line (4 x 16-byte stores back to back).
    uint64_t* abuf_all = reinterpret_cast<uint64_t*>(keypark_all + 4 * 64 * 16 * TQ);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const unsigned long long rt_entry = DIAG_STAMPS ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (blockIdx.x == 0) hdr_reset(a.hdr, tid, 256);   // empty uncertified list, idle tickets
    const int wq = a.wq, wi_n = 4 / wq;
    const int wqi = wave % wq, wii = wave / wq;
    // Workgroups go to the 8 XCDs round-robin (workgroup i -> XCD i % 8), each with its own L2.  The logical id below hands
    // every XCD one contiguous eighth of the (split, query block) grid, so the query blocks that stream the SAME item split run on
    // the same XCD and share its L2: the catalogue is then pulled from HBM / Infinity Cache once, not once per XCD.
    const unsigned n_wg = gridDim.x;
    const unsigned wg = ((n_wg & 7u) == 0u && !ABLATE(256)) ? (blockIdx.x & 7u) * (n_wg >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int qblock = int(wg % unsigned(a.n_qblocks)), split = int(wg / unsigned(a.n_qblocks));
    const int list = split * wi_n + wii;
    const int64_t qw0 = int64_t(qblock) * (wq * 32 * TQ) + int64_t(wqi) * 32 * TQ;  // wave's first query
    float* keypark = keypark_all + wave * (64 * 16 * TQ);
    uint64_t* const abuf = abuf_all + wave * (8 * TQ * 64) + lane;   // slot s of subtile b: abuf[(s * TQ + b) * 64]
    int lo[TQ];          // stream positions [lo, cnt) of the current group of 8 live in abuf, everything below is in memory

    // ---- resident query fragments (MFMA B operand): lane (r, h) holds Q[qw0 + 32 b + r][8 g + 4 h .. +3] -------
    // float32: NG k-groups of 8 columns (float4 per lane half); float16: NG / 2 k-steps of 16 columns (half8 per lane half)
    using QF = typename std::conditional<F16, half8_t, float4>::type;
    constexpr int NQF = F16 ? NG / 2 : NG;
    QF qf[TQ][NQF];
    int e_tot[TQ];       // F16: key scale of this lane's query (2^e_tot)
    float2_t nscale[TQ]; // F16: (-2^e_tot, -2^e_tot)
    float tau[TQ];
    int cnt[TQ];
    bool first[TQ], scaled[TQ], seeded[TQ];
    uint32_t soff[TQ];   // this lane's stream = wg_cand[soff .. soff + CAPH)
    uint64_t* const wg_cand = a.cand + ((int64_t(list) * a.bq + qw0) * 2) * CAPH;   // wave-uniform base
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = qw0 + 32 * b + r;
        const bool live = q < a.B;
        e_tot[b] = F16 ? a.x_exp + a.qexp[q] : 0;                        // qexp has bq entries
        nscale[b] = float2_t{-ldexpf(1.0f, e_tot[b]), -ldexpf(1.0f, e_tot[b])};
        const float t0 = (a.tau0 && live) ? a.tau0[q] : a.debug_tau0;    // seeded threshold of this query (real units), or open
        seeded[b] = a.tau0 != nullptr && live && t0 > -INFINITY;
        tau[b] = (live && !(a.debug_nofilter & 1)) ? (F16 ? ldexpf(t0, e_tot[b]) : t0) : INFINITY;  // strict '>' filter: dead queries never append
        cnt[b] = 0;
        lo[b] = 0;
        first[b] = true;
        scaled[b] = false;
        soff[b] = (uint32_t(32 * b + r) * 2u + uint32_t(h)) * CAPH;
    }

    const int64_t i_begin = int64_t(split) * a.split_items;
    const int64_t i_end = std::min<int64_t>(i_begin + a.split_items, a.n_items);
    const int n_tiles = int((i_end - i_begin + IB - 1) / IB);

    // LDS-DMA fill of tile t into ring slot (t & 1): every wave issues DMA_PER_WAVE pieces of 64 x 16 B.
    // Tiles are visited in the order perm(t) = (t * tile_mul) mod n_tiles (tile_mul coprime with n_tiles), so the
    // first tiles of a stream sample the whole list instead of its head (needed by the scaled first threshold).
    // The permutation is advanced incrementally (add the step, subtract n_tiles on overflow): a modulo per tile would
    // be a ~150-instruction software division on this hardware.
    const int tile_mul = (!DENSE && a.scaled_first) ? coprime_step(n_tiles) : 1;
    auto perm_next = [&](int p) { p += tile_mul; return p >= n_tiles ? p - n_tiles : p; };
    const int wi_mask = wi_n - 1;   // wi_n is 1, 2 or 4
    // The table is allocated with its row count padded to a multiple of 32 (zero rows whose half norm is +inf, see
    // hwer_pack_rows_f32), so tile reads never leave the allocation and pad rows can never pass the filter.
    // Per-lane byte offset of this lane's chunk inside a tile (piece 0); piece i adds i * PIECE_ROWS rows.
    constexpr int PIECE_ROWS = 64 / CH > 0 ? 64 / CH : 1;        // rows covered by one 1-KiB piece (CH <= 64)
    static_assert(CH <= 64, "one LDS-DMA piece must hold at least one whole row");
    const uint32_t row_bytes = uint32_t(a.ldx) * (F16 ? 2u : 4u);
    // Piece p = wave + 4 i covers rows p * PIECE_ROWS + lane / CH.  For CH >= 16 the swizzle is row & 15 and
    // PIECE_ROWS * 4 is a multiple of ... handled generically: the offset of piece i is recomputed from two lane
    // constants with 32-bit integer ops only (v_mad_u32_u24-class: row < 32, row_bytes <= 1024).
    const uint32_t lane_row = uint32_t(lane / CH), lane_chunk = uint32_t(lane % CH);
    // wave-uniform loop invariants of the fill, as plain integers / byte pointers (no generic-pointer conversions, no
    // 64-bit multiplies per tile): LDS byte address of this wave's first piece in ring slot 0, byte size of a tile in
    // HBM, and the split's first row.
    static_assert(TILE_CHUNKS % 256 == 0, "every wave issues the same number of pieces");
    const uint32_t xs_addr0 = __builtin_amdgcn_readfirstlane(lds_addr(xs)) + uint32_t(wave) * 1024u;
    const uint32_t hn_addr0 = __builtin_amdgcn_readfirstlane(lds_addr(hn));
    // (seed pre-pass: sample tile p is catalogue tile p * tile_stride; everything else works in sample-local item numbers)
    const char* const x_split = F16 ? reinterpret_cast<const char*>(a.X16) + i_begin * a.ldx * 2 * a.tile_stride
                                    : reinterpret_cast<const char*>(a.X + i_begin * a.ldx * a.tile_stride);
    const float* const hn_split = a.hnorm + i_begin * a.tile_stride;
    const uint32_t tile_bytes = uint32_t(IB) * row_bytes * uint32_t(a.tile_stride);
    const uint32_t hn_tile = uint32_t(IB) * uint32_t(a.tile_stride);
    auto tile_dma = [&](int t, int ptile) {
        const uint32_t xb_addr = xs_addr0 + uint32_t(t & 1) * uint32_t(IB * CH * 16);
        const char* tile_ptr = x_split + uint64_t(uint32_t(ptile)) * tile_bytes;    // wave-uniform
        uint32_t lr = lane_row, lc = lane_chunk;
        // float32 kernels: opaque - keeps the per-piece offsets from being hoisted into (spilled) registers.  The float16 kernels
        // have the registers (229 of 256 at d = 128, TQ = 2): their 1 - 4 piece offsets are loop invariants (round 3: -8 VALU per tile)
        if constexpr (!F16 || DMA_PER_WAVE > 4) asm volatile("" : "+v"(lr), "+v"(lc));
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            // piece = wave + 4 i covers rows piece * PIECE_ROWS + lane / CH
            const uint32_t row = uint32_t(wave * PIECE_ROWS) + uint32_t(4 * i * PIECE_ROWS) + lr;
            const uint32_t off = __umul24(row, row_bytes) + ((lc ^ uint32_t(swz<CH>(int(row)))) << 4);
            lds_dma16(tile_ptr, off, xb_addr + uint32_t(i) * 4096u);
        }
        if (wave == 0 && lane < IB)   // half norms of the tile rows: one 4-byte piece, lanes 0..31
            lds_dma4(hn_split + int64_t(ptile) * hn_tile, uint32_t(lane) * 4u, hn_addr0 + uint32_t(t & 1) * uint32_t(IB * 4));
    };

    // staged appends (streaming mode).  stage(b, key): park the key; a completed group of 8 leaves as one 64-byte line when all
    // of it is staged (lo <= group start), else - once after every compaction - its staged part goes out key by key.
    auto flush_tail = [&](int b) {       // write the staged keys [lo, cnt) of subtile b (fewer than 8) key by key
        for (int p = lo[b]; p < cnt[b]; ++p) wg_cand[soff[b] + uint32_t(p)] = abuf[((p & 7) * TQ + b) * 64];
        lo[b] = cnt[b];
    };
    auto stage = [&](int b, uint64_t key) {
        if constexpr (!STAGE_APPENDS) {   // direct append: one 8-byte store per key
            wg_cand[soff[b] + uint32_t(cnt[b])] = key;
            ++cnt[b];
            lo[b] = cnt[b];
            return;
        }
        abuf[((cnt[b] & 7) * TQ + b) * 64] = key;
        ++cnt[b];
        if ((cnt[b] & 7) == 0) {
            if (lo[b] <= cnt[b] - 8) {
                uint4* dst = reinterpret_cast<uint4*>(wg_cand + soff[b] + uint32_t(cnt[b] - 8));   // 64-byte aligned
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const uint64_t k0 = abuf[((2 * s2) * TQ + b) * 64], k1 = abuf[((2 * s2 + 1) * TQ + b) * 64];
                    dst[s2] = make_uint4(uint32_t(k0), uint32_t(k0 >> 32), uint32_t(k1), uint32_t(k1 >> 32));
                }
                lo[b] = cnt[b];
            } else {
                flush_tail(b);
            }
        }
    };
    if constexpr (F16) __builtin_amdgcn_s_setprio(HWER_F16_REST_PRIO); else __builtin_amdgcn_s_setprio(3);
    int p_cur = 0;                       // perm(t)
    int p_dma = perm_next(0);            // perm(t + 1), then perm(t + 2) while tile t is computed
    if (n_tiles > 0) tile_dma(0, 0);
    // The query fragments are requested AFTER the first tile: vmcnt retires in issue order, so waiting for "all but the
    // TQ * NG fragment loads" means the tile has landed, and the first tile's MFMAs start as soon as their own fragments
    // arrive (the compiler's per-register waits) instead of after the whole 32-KiB-per-wave prologue.
    static_assert(TQ * NQF <= 63, "vmcnt immediate");
#pragma unroll
    for (int b = 0; b < TQ; ++b)
#pragma unroll
        for (int g = 0; g < NQF; ++g)
            qf[b][g] = reinterpret_cast<const QF*>(F16 ? a.Qf16 : static_cast<const void*>(a.Q))[((qw0 / 32 + b) * NQF + g) * 64 + lane];   // fragment order: 1 KiB per instruction
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TQ * NQF) : "memory");
    __syncthreads();   // tile 0 has landed for every wave
    // d = 256: the fragments are made opaque HERE, once - the compiler then places its wait for them in front of this statement
    // instead of keeping per-register waits (ending in vmcnt(0), i.e. a drain of the tile prefetch and the appends in flight)
    // inside the tile loop.  Measured against the kernel without it on the same box: c5 screen -2.7 %; at d = 128 (TQ = 2) +0.8 %,
    // so only the wide shape takes it.
    if constexpr (F16 && NG >= 32) {   // (the float32 kernel of this width has no register to spare: 23.6 -> 26.2 ms with it)
#pragma unroll
        for (int b = 0; b < TQ; ++b)
#pragma unroll
            for (int g = 0; g < NQF; ++g) make_opaque(qf[b][g]);
    }
    if (n_tiles > 1) tile_dma(1, p_dma);
    p_dma = perm_next(p_dma);

#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
    // phase experiment (HWER_SCREEN_NOFILTER = n << 16): the workgroup in the CU's odd threadgroup slot starts its tile loop n x ~64 cycles late
    if (!DENSE && (a.debug_nofilter >> 16) != 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid >> 16) & 1u)
            for (int i = 0; i < (a.debug_nofilter >> 16); ++i) __builtin_amdgcn_s_sleep(1);
    }
#endif
    float top[DENSE ? TQ : 1][4];   // seed pre-pass (seed_m): the lane's best keys per query subtile (scaled units when F16)
#pragma unroll
    for (int b = 0; b < (DENSE ? TQ : 1); ++b) top[b][0] = top[b][1] = top[b][2] = top[b][3] = -INFINITY;
    const bool getenv_calib = a.debug_nofilter == 2;
    unsigned long long clk0 = 0, rt0 = 0;
    if (DIAG_STAMPS) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    unsigned long long rt_m0 = 0, rt_n0 = 0, rt_m2 = 0;   // phase marks of tile 64 / 66 (10-ns ticks): are the two workgroups of a CU in step?
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, acc_pre = 0, acc_mfma = 0, acc_epi = 0, acc_wait = 0, acc_post = 0, acc_flush = 0, acc_dma = 0;
    for (int t = 0; t < n_tiles; ++t) {
        STAMP(t0);
        const float* xb_tile = xs + (t & 1) * IB * CH * 4;
        const float* hb_tile = hn + (t & 1) * IB;
        const int64_t i0_tile = i_begin + int64_t(p_cur) * IB;
        p_cur = perm_next(p_cur);
        if ((t & wi_mask) == wii) {   // wave-uniform: tiles are dealt round-robin over the wi waves that share these queries
#pragma unroll
          for (int sub = 0; sub < NSUB; ++sub) {   // the 32-row MFMA subtiles of this tile (same swizzle: (row + 32) & 15 == row & 15)
            const float* xb = xb_tile + sub * 32 * CH * 4;
            const float* hb = hb_tile + sub * 32;
            const int64_t i0 = i0_tile + 32 * sub;
            f32x16 acc[TQ];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + 8 * rg + 4 * h);
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    if constexpr (F16) {   // scaled units: exact (powers of two); two keys per v_pk_mul_f32
                        if (ABLATE(8)) { acc[b][4 * rg + 0] = acc[b][4 * rg + 1] = acc[b][4 * rg + 2] = acc[b][4 * rg + 3] = 0.f; continue; }
                        const float2_t lo = float2_t{hv.x, hv.y} * nscale[b], hi = float2_t{hv.z, hv.w} * nscale[b];
                        acc[b][4 * rg + 0] = lo.x;
                        acc[b][4 * rg + 1] = lo.y;
                        acc[b][4 * rg + 2] = hi.x;
                        acc[b][4 * rg + 3] = hi.y;
                    } else {
                    acc[b][4 * rg + 0] = -hv.x;
                    acc[b][4 * rg + 1] = -hv.y;
                    acc[b][4 * rg + 2] = -hv.z;
                    acc[b][4 * rg + 3] = -hv.w;
                    }
                }
            }
            if constexpr (F16) {
                // A fragment of k-step g (columns 16 g .. 16 g + 15): logical chunk 2 g + h of row r, stored at chunk (2 g + h) ^ swz(r):
                // one conflict-free ds_read_b128 per v_mfma_f32_32x32x16_f16 (gfx950: twice the K of the 32x32x8 form per issue)
                int sw = swz<CH>(r);
                asm volatile("" : "+v"(sw));
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(xb) + r * (NG * 16);
                auto a_ptr = [&](int g) { return abase + 16 * ((2 * g + h) ^ sw); };
                // fragment look-ahead: PF ds_read_b128 in flight ahead of the MFMAs that consume them (two MFMAs = 64 matrix-pipe cycles per
                // k-step cover less than one LDS round trip)
                constexpr int PF = HWER_F16_PREFETCH < NQF ? HWER_F16_PREFETCH : NQF;
                half8_t avq[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) avq[g] = *reinterpret_cast<const half8_t*>(a_ptr(g));
                __builtin_amdgcn_s_setprio(HWER_F16_MFMA_PRIO);
                STAMP(t1);
                if (DIAG_STAMPS && sub == 0 && (t == 64 || t == 66)) { if (t == 64) rt_m0 = __builtin_amdgcn_s_memrealtime(); else rt_m2 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
                for (int g = 0; g < NQF; ++g) {
                    const half8_t av = avq[g % PF];
                    if (g + PF < NQF && !ABLATE(32)) avq[g % PF] = *reinterpret_cast<const half8_t*>(a_ptr(g + PF));
                    if (ABLATE(128)) continue;
#pragma unroll
                    for (int b = 0; b < TQ; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, qf[b][g], acc[b], 0, 0, 0);
                }
            } else {
            // A fragment of k-group g: logical chunk 2g + h of row r, stored at chunk (2g + h) ^ swz(r).
            // (2g + h) ^ sw = 2 (g ^ (sw >> 1)) + (h ^ (sw & 1)), and XOR-ing g with a 3-bit value only touches g's low
            // three bits: the address is lane_base + 32 B * ((g & 7) ^ swh) + 256 B * (g >> 3).  `swh` is made opaque to
            // the optimiser once per tile so that the 8 XOR terms are recomputed (1 VALU each) instead of being hoisted
            // out of the tile loop into registers that spill.
            int sw = swz<CH>(r);
            asm volatile("" : "+v"(sw));
            const int swh = (sw >> 1) & (NG >= 8 ? 7 : (NG - 1));
            const float* abase = xb + r * (CH * 4) + 4 * (h ^ (sw & 1));
            auto a_ptr = [&](int g) { return abase + 8 * ((g & 7) ^ swh) + 64 * (g >> 3); };
            float4 av_next = *reinterpret_cast<const float4*>(a_ptr(0));
            // MFMA phase at low priority, everything else at high priority: the co-resident wave's back-to-back MFMAs
            // otherwise leave this wave about one issue slot per 64-cycle MFMA for its filter / bookkeeping code
            // (measured: 8 compare+branch pairs took ~1500 cycles).  An MFMA needs the issue port for a few cycles only,
            // so yielding to the other wave's scalar / vector work costs the MFMA stream nothing.
            __builtin_amdgcn_s_setprio(0);
            STAMP(t1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float4 av = av_next;
                if (g + 1 < NG) av_next = *reinterpret_cast<const float4*>(a_ptr(g + 1));   // prefetch under the MFMAs
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, qf[b][g].x, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, qf[b][g].y, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, qf[b][g].z, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, qf[b][g].w, acc[b], 0, 0, 0);
                }
            }
            }
            if constexpr (F16) __builtin_amdgcn_s_setprio(HWER_F16_REST_PRIO); else __builtin_amdgcn_s_setprio(3);
            if (DIAG_STAMPS) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[TQ - 1][15])); t2 = stamp(); if (t == 64 && sub == NSUB - 1) rt_n0 = __builtin_amdgcn_s_memrealtime(); }
            // Streaming mode: outstanding VMEM here = the DMA of tile t+1 (issued a whole MFMA phase ago) and the appends
            // of the previous tile (a whole tile ago): this wait is free, and it lets this tile's appends go straight to
            // HBM and stay in flight across the barrier below (no LDS staging of the appends, no flush).
            // (second subtile of a 64-row tile: no wait - it would only stall on the appends the first one has just issued)
            if constexpr (!DENSE) { if (sub == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if constexpr (DENSE) {
                if (a.seed_m < 0) {   // (wave-uniform) seed pre-pass, group maxima: ONE value per (sample tile, lane half, query) - the maximum of the
                    // lane's 16 keys (a v_max3 tree: 8 instructions per subtile where the top-4 insertion took 64 and the float16 key rows 8
                    // conversions + 4 stores), stored transposed: dense[(2 tile + h) * ld_dense + query], ld_dense = bq - the 32 queries of a
                    // lane half are one 128-byte segment.  Exactly TQ stores per tile: the counted wait at the tile boundary relies on it.
                    const int64_t slot = (i0 >> 5) * 2 + h;
#pragma unroll
                    for (int b = 0; b < TQ; ++b) {
                        const float m0 = fmaxf(fmaxf(acc[b][0], acc[b][1]), acc[b][2]), m1 = fmaxf(fmaxf(acc[b][3], acc[b][4]), acc[b][5]);
                        const float m2 = fmaxf(fmaxf(acc[b][6], acc[b][7]), acc[b][8]), m3 = fmaxf(fmaxf(acc[b][9], acc[b][10]), acc[b][11]);
                        const float m4 = fmaxf(fmaxf(acc[b][12], acc[b][13]), acc[b][14]);
                        const float mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), acc[b][15]));
                        a.dense[slot * a.ld_dense + (qw0 + 32 * b + r)] = F16 ? ldexpf(mx, -e_tot[b]) : mx;
                    }
                } else if (a.seed_m && a.seed_group) {   // (wave-uniform) top-4 of the lane's group maxima: 8 + 4 instructions per 16 keys
#pragma unroll
                    for (int b = 0; b < TQ; ++b) {
                        const float m0 = fmaxf(fmaxf(acc[b][0], acc[b][1]), acc[b][2]), m1 = fmaxf(fmaxf(acc[b][3], acc[b][4]), acc[b][5]);
                        const float m2 = fmaxf(fmaxf(acc[b][6], acc[b][7]), acc[b][8]), m3 = fmaxf(fmaxf(acc[b][9], acc[b][10]), acc[b][11]);
                        const float m4 = fmaxf(fmaxf(acc[b][12], acc[b][13]), acc[b][14]);
                        const float x = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), acc[b][15]));
                        const float n0 = fmaxf(top[b][0], x);
                        const float n1 = __builtin_amdgcn_fmed3f(top[b][0], top[b][1], x);
                        const float n2 = __builtin_amdgcn_fmed3f(top[b][1], top[b][2], x);
                        const float n3 = __builtin_amdgcn_fmed3f(top[b][2], top[b][3], x);
                        top[b][0] = n0; top[b][1] = n1; top[b][2] = n2; top[b][3] = n3;
                    }
                } else if (a.seed_m) {   // (wave-uniform) seed pre-pass: the lane's running top-4 per query subtile, sorted descending
#pragma unroll
                    for (int b = 0; b < TQ; ++b)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            // insertion into the sorted four as ONE max and THREE medians of the OLD values (t0 >= t1 >= t2 >= t3: the new t_j is the
                            // median of t_(j-1), t_j and x) - four independent instructions per key where the max / min chain took seven; the wave-level
                            // "any lane improves" test in front of it is gone: with 64 lanes it was true for nearly every key (round 4: the pre-pass
                            // spent ~290 vector instructions per 32-row tile on this against the MFMAs' 16)
                            const float x = acc[b][q];
                            const float n0 = fmaxf(top[b][0], x);
                            const float n1 = __builtin_amdgcn_fmed3f(top[b][0], top[b][1], x);
                            const float n2 = __builtin_amdgcn_fmed3f(top[b][1], top[b][2], x);
                            const float n3 = __builtin_amdgcn_fmed3f(top[b][2], top[b][3], x);
                            top[b][0] = n0; top[b][1] = n1; top[b][2] = n2; top[b][3] = n3;
                        }
                } else {
                // small catalogues: keep every key (lane (r, h): query r, items i0 + 8 rg + 4 h + 0..3 -> one 16-byte store)
#pragma unroll
                // The key matrix has bq >= B rows, so every lane stores unconditionally: exactly 4 * TQ store
                // instructions per wave and tile, which the counted wait at the tile boundary relies on.
                for (int b = 0; b < TQ; ++b) {
                    const int64_t q = qw0 + 32 * b + r;
                    if (a.dense_f16) {   // (wave-uniform) sample keys of the seed pre-pass, rounded to float16: still 4 stores per subtile
                        half4_t* row16 = reinterpret_cast<half4_t*>(reinterpret_cast<_Float16*>(a.dense) + q * a.ld_dense + i0 + 4 * h);
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            half4_t kv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) kv[j] = (_Float16)(F16 ? ldexpf(acc[b][4 * rg + j], -e_tot[b]) : acc[b][4 * rg + j]);
                            row16[2 * rg] = kv;
                        }
                        continue;
                    }
                    float* row = a.dense + q * a.ld_dense + i0 + 4 * h;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        if constexpr (F16)   // back to real units (exact)
                            *reinterpret_cast<float4*>(row + 8 * rg) = make_float4(ldexpf(acc[b][4 * rg], -e_tot[b]), ldexpf(acc[b][4 * rg + 1], -e_tot[b]),
                                                                                  ldexpf(acc[b][4 * rg + 2], -e_tot[b]), ldexpf(acc[b][4 * rg + 3], -e_tot[b]));
                        else
                            *reinterpret_cast<float4*>(row + 8 * rg) = make_float4(acc[b][4 * rg], acc[b][4 * rg + 1], acc[b][4 * rg + 2], acc[b][4 * rg + 3]);
                    }
                }
                }   // !seed_m
            } else {
            // ---- streaming filter ---------------------------------------------------------------------------------
            // Lane (r, h) holds 16 keys of query r.  Passing keys are rare (K'/n per key), so instead of 16 predicated
            // append blocks the lane builds a 16-bit pass mask, parks its 16 keys in LDS and pops set bits: the loop runs
            // max-popcount-over-the-wave times (2-3 in steady state, 16 only during warm-up).  (Measured against 16
            // independent predicated blocks: 7.0K vs 10.0K cycles per tile.)
            // pass masks, 2 VALU per key: tau - key is negative exactly when key > tau (strict), and a funnel shift
            // (v_alignbit) pushes that sign bit into the mask.  Bit (15 - q) of half-word b <-> accumulator register q of
            // query subtile b.  Both subtiles share ONE park and ONE pop loop: the loop runs max-over-lanes of the lane's
            // total pops, not the sum of the per-subtile maxima.
            constexpr bool ALWAYS_SPARSE = F16 && FILTER_V2 && NG > 16 && WIDE64;   // d = 256: the park-free filter only (no park in its LDS, launch_screen16_t)
            if (ALWAYS_SPARSE || (F16 && FILTER_V2 && a.sparse_hits)) {   // (wave-uniform)
                // Round 3, catalogues so large that most tiles of a wave hold NO passing key (seeded thresholds let c K' keys of a query
                // through whatever the catalogue size: 0.2 per 32 x 64 tile and wave on 10 M rows, 1.7 on 1.25 M).  What bounds this loop
                // is the wave's serial chain and instruction count, not a pipe (SQ counters): the per-key pass masks (48
                // VALU per tile) are replaced by
                //   * the lane's maximum of its 16 keys (a v_max3_f32 tree: 8 VALU) and ONE compare with the threshold per subtile;
                //   * only when some lane passes (wave-uniform branch): one compare per accumulator register, and the append code of a
                //     register runs under the lanes that pass (s_cbranch_execz skips it otherwise) - the key is still in its register:
                //     no park, no pop loop, no LDS traffic.  The 16 compare / branch pairs are a ~600-cycle chain, so this variant only
                //     pays while hit tiles are the exception: same-box A/B (rocprofv3 kernel durations) 10.62 -> 10.27 ms on the 10 M-row
                //     catalogue, but 1.65 -> 1.71 ms on the 1.25 M-row shard and 2.91 -> 2.95 at d = 256 - the plan switches it on
                //     (sparse_hits) below 0.6 expected passing keys per tile and wave.
                if (!ABLATE(4)) {
                    const uint32_t item0 = uint32_t(i0) + 4 * h;
#pragma unroll
                    for (int b = 0; b < TQ; ++b) {
                        const float m0 = fmaxf(fmaxf(acc[b][0], acc[b][1]), acc[b][2]), m1 = fmaxf(fmaxf(acc[b][3], acc[b][4]), acc[b][5]);
                        const float m2 = fmaxf(fmaxf(acc[b][6], acc[b][7]), acc[b][8]), m3 = fmaxf(fmaxf(acc[b][9], acc[b][10]), acc[b][11]);
                        const float m4 = fmaxf(fmaxf(acc[b][12], acc[b][13]), acc[b][14]);
                        const float mx = fmaxf(fmaxf(fmaxf(m0, m1), m2), fmaxf(fmaxf(m3, m4), acc[b][15]));
                        if (__any(mx > tau[b]) && !ABLATE(1)) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                if (acc[b][q] > tau[b])   // back to real units (exact), then straight to the lane's stream
                                    stage(b, pack_key(ldexpf(acc[b][q], -e_tot[b]), item0 + (q & 3) + 8 * (q >> 2)));
                            }
                        }
                    }
                } else {
                    asm volatile("" :: "v"(acc[0][0]), "v"(acc[TQ - 1][15]));
                }
            } else if constexpr (!ALWAYS_SPARSE) {
            uint32_t mask = 0;
            if (ABLATE(4)) asm volatile("" :: "v"(acc[0][0]), "v"(acc[TQ - 1][15]));
            else
#pragma unroll
            for (int b = TQ - 1; b >= 0; --b) {
                const float2_t tau2 = {tau[b], tau[b]};
#pragma unroll
                for (int q = 0; q < 16; q += 2) {   // (two differences per v_pk_add_f32)
                    const float2_t df = tau2 - float2_t{acc[b][q], acc[b][q + 1]};
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(df.x), 31);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(df.y), 31);
                }
            }
            if (__any(mask != 0u)) {
                const uint32_t item0 = uint32_t(i0) + 4 * h;
                // park layout [16 TQ key slots][64 lanes]: every access is 64 consecutive words, conflict-free whatever slot a
                // lane pops (a lane-major layout puts all lanes on the same few banks)
                float* park = keypark + lane;
#pragma unroll
                for (int b = 0; b < TQ; ++b)
#pragma unroll
                    for (int q = 0; q < 16; ++q) park[(16 * b + q) * 64] = acc[b][q];
                while (__any(mask != 0u)) {
                    if (mask != 0u) {
                        const int bit = __builtin_ctz(mask);
                        mask &= mask - 1u;
                        const int q = 15 - (bit & 15);
                        float kval = park[((bit & 16) + q) * 64];
                        if constexpr (F16) kval = ldexpf(kval, -((TQ == 2 && (bit & 16)) ? e_tot[TQ - 1] : e_tot[0]));   // back to real units (exact)
                        const uint64_t key = pack_key(kval, item0 + (q & 3) + 8 * (q >> 2));
                        if constexpr (TQ == 1) {
                            stage(0, key);
                        } else {
                            if ((bit & 16) != 0) stage(1, key);   // half-word 1 = subtile 1
                            else stage(0, key);
                        }
                    }
                }
            }
            }   // !(F16 && FILTER_V2)
            }   // !DENSE
          }   // sub
        }
        STAMP(t3);
        // ---- tile boundary ---------------------------------------------------------------------------------------------
        if constexpr (DENSE) {
            // the 4 * TQ key stores of this tile are YOUNGER than the DMA of tile t+1 (vmcnt retires in issue order):
            // wait for everything but them, so the DMA has landed while the stores stay in flight
            // (the seed pre-pass with seed_m stores nothing in the loop: plain wait)
            if ((t & wi_mask) == wii && !a.seed_m) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * TQ) : "memory");
            else if ((t & wi_mask) == wii && a.seed_m < 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TQ) : "memory");   // group maxima: TQ stores per tile
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // waves that computed this tile waited for the DMA of tile t+1 before their appends; the others do it here
            if ((t & wi_mask) != wii) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!ABLATE(64)) __syncthreads();
        STAMP(t4);
        if constexpr (DENSE) {
            if (t + 2 < n_tiles) tile_dma(t + 2, p_dma);
            p_dma = perm_next(p_dma);
            if (DIAG_STAMPS) {
                t5 = stamp();
                acc_pre += t1 - t0; acc_mfma += t2 - t1; acc_epi += t3 - t2; acc_wait += t4 - t3; acc_post += t5 - t4;
            }
            continue;
        }
        unsigned long long t4a = 0, t4b = 0;
        STAMP(t4a);
        if (DIAG_STAMPS && getenv_calib) { t4 = stamp(); t4a = stamp(); }   // calibration: flush := cost of one stamp
        if (t + 2 < n_tiles && !ABLATE(16)) tile_dma(t + 2, p_dma);      // refill the ring slot every wave has just finished reading
        p_dma = perm_next(p_dma);
        STAMP(t4b);
        // ---- stream capacity: a lane appends at most 16 keys per 32-row subtile, so checking every 64 rows (every SECOND 32-row tile,
        // every 64-row tile) against CAPH - 32 keeps every stream within its CAPH slots; a compaction leaves at most K' <= 448 <=
        // CAPH - 32 keys in a stream.
        if ((t & 1) || NSUB == 2) {
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            uint64_t need = __ballot(cnt[b] > CAPH - 32);
            while (need != 0ull) {
                const int jj = (__builtin_ctzll(need)) & 31;
                const int c0 = __shfl(cnt[b], jj, kWave), c1 = __shfl(cnt[b], jj + 32, kWave);
                const int was_first = __shfl(int(first[b]), jj, kWave);
                uint64_t* s0 = wg_cand + (uint32_t(32 * b + jj) * 2u) * CAPH;
                // First compaction of a query: its streams hold (nearly) every key of the n items seen so far, a
                // pseudo-random sample of the list (tiles are visited in a permuted order).  Of the list's K' best,
                // mu = K' n / N are expected in the sample; selecting rank mu + 4 sqrt(mu) + 4 instead of K' puts the
                // threshold just below the list's K'-th best key with overwhelming probability, and the end-of-stream
                // count check (>= K' keys above the threshold, else UNDERFULL) makes the shortcut exact.
                int rank = a.kprime;
                if (was_first && a.scaled_first) {
                    const float mu = float(a.kprime) * float(c0 + c1) * float(wi_n) / float(i_end - i_begin);   // this list sees 1/wi_n of the tiles
                    const int mrank = int(mu + 4.0f * sqrtf(mu) + 5.0f);
                    if (2 * mrank < a.kprime) rank = mrank;
                }
                float T;
                int n0, n1;
                if (r == jj) flush_tail(b);   // the compaction reads the streams from memory: the owner lanes' staged keys first
                // in-loop compactions may keep up to 32 keys more than asked (a stream then holds <= K' + 32 <= CAPH - 32)
                compact_query(s0, s0 + CAPH, c0, c1, rank, rank == a.kprime ? COMPACT_SLACK : min(COMPACT_SLACK, rank / 4 + 1), T, n0, n1);
                if (r == jj) {
                    tau[b] = F16 ? ldexpf(T, e_tot[b]) : T;   // the filter compares in the query's scaled units
                    cnt[b] = h == 0 ? n0 : n1;
                    lo[b] = cnt[b];                            // the rewritten stream is in memory; nothing is staged
                    scaled[b] = scaled[b] || (rank != a.kprime);
                    first[b] = false;
                }
                need &= ~((1ull << jj) | (1ull << (jj + 32)));
            }
        }
        }   // capacity check on odd tiles
        if (DIAG_STAMPS) {
            t5 = stamp();
            acc_pre += t1 - t0; acc_mfma += t2 - t1; acc_epi += t3 - t2; acc_wait += t4 - t3; acc_post += t5 - t4;
            acc_flush += t4a - t4; acc_dma += t4b - t4a;
        }
    }
    if (DIAG_STAMPS && lane == 0) {
        unsigned long long* o = a.debug_stamps + (size_t(blockIdx.x) * 4 + wave) * 8;
        {
            uint32_t hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* o2 = a.debug_stamps + size_t(gridDim.x) * 32 + (size_t(blockIdx.x) * 4 + wave) * 4;
            o2[0] = (static_cast<unsigned long long>(xcc) << 32) | hwid; o2[1] = rt_m0; o2[2] = rt_n0; o2[3] = rt_m2;
        }
        o[0] = rt0 - rt_entry; o[1] = acc_mfma; o[2] = acc_epi; o[3] = acc_wait; o[4] = acc_post; o[5] = n_tiles;   // o[0]: prologue in 10-ns ticks
        o[6] = __builtin_amdgcn_s_memtime() - clk0; o[7] = __builtin_amdgcn_s_memrealtime() - rt0;   // in-kernel clock = o[6] / o[7] x 100 MHz
    }
    if constexpr (DENSE) {
        if (a.seed_m > 0) {   // hand over the lane's top-M (real units; -inf where the lane saw fewer keys)
            const int bin = ((split * wi_n + wii) * 2 + h) * a.seed_m;
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                float* dst = a.dense + (qw0 + 32 * b + r) * a.ld_dense + bin;   // bq rows: every lane stores
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < a.seed_m) dst[j] = F16 ? ldexpf(top[b][j], -e_tot[b]) : top[b][j];
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < TQ; ++b) flush_tail(b);   // every staged key reaches its stream before the hand-over
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        // final compaction: queries holding more than K' keys keep exactly their K' best
        const int partner = __shfl_xor(cnt[b], 32, kWave);
        const int total = cnt[b] + partner;
        const bool underfull = scaled[b] && total < a.kprime;   // the scaled threshold dropped keys it should not have
        // seeded threshold, never compacted, fewer than K' keys above it: complete only together with the query's other lists
        const bool seeded_short = seeded[b] && first[b] && total < a.kprime;
        uint64_t need = __ballot(total > a.kprime) & 0xFFFFFFFFull;
        while (need != 0ull) {
            const int jj = __builtin_ctzll(need);
            const int c0 = __shfl(cnt[b], jj, kWave), c1 = __shfl(cnt[b], jj + 32, kWave);
            uint64_t* s0 = wg_cand + (uint32_t(32 * b + jj) * 2u) * CAPH;
            float T;
            int n0, n1;
            compact_query(s0, s0 + CAPH, c0, c1, a.kprime, 0, T, n0, n1);   // exact: a list hands over <= K' keys
            if (r == jj) cnt[b] = h == 0 ? n0 : n1;
            need &= ~(1ull << jj);
        }
        const int64_t q = qw0 + 32 * b + r;
        a.cand_cnt[(int64_t(list) * a.bq + q) * 2 + h] = cnt[b] | (underfull ? CNT_UNDERFULL : 0) | (seeded_short ? CNT_SEEDED_SHORT : 0);
    }
}

// =====================================================================================================
// Round 4: the float16 streaming screen as a PING-PONG of SIMD partners (screen16_pp_kernel).
// =====================================================================================================
// screen_kernel above runs two independent 256-thread workgroups per CU: the two waves that share a SIMD belong to DIFFERENT workgroups,
// nothing couples their phases, and their MFMA phases overlap most of the time - matrix beside matrix on one pipe, filter beside filter
// on one issue port (round-3 stamps: an MFMA phase of 512 pipe-cycles takes 1 280 - 1 330 cycles, the pipe is busy 0.33 - 0.42 of the
// launch; a start-phase stagger changed nothing because nothing holds a phase).  Here ONE 512-thread workgroup owns the CU: waves w and
// w + 4 sit on the same SIMD (cdna guide, "Two waves per SIMD") and take turns on the matrix pipe through a TOKEN in LDS (one word per
// SIMD, strict alternation): a wave runs the 8 NG / 16 TQ MFMAs of a 32-row subtile only while it holds the token and hands it over
// right behind its last MFMA, so a wave's MFMA burst always runs beside its partner's filter / LDS-DMA / bookkeeping code - matrix beside
// VALU and memory on every SIMD, by construction - and the burst takes ~880 instead of ~1 300 cycles (in-kernel stamps).
//   * 8 waves x 32 TQ queries per workgroup (512 at TQ = 2); every wave streams every item tile of the workgroup's split;
//   * item tiles of 64 rows in a 3-slot LDS ring, ONE workgroup barrier per tile (ring safety only), STAGGERED: waves 0-3 take it behind the
//     tile's last filter segment, waves 4-7 between the tile's last burst and that filter segment, so SIMD partners leave every barrier half
//     a step apart; every wave requests its 1/8 of tile t + 2 behind the MFMA burst of (t, subtile 0) - the slot held tile t - 1, which
//     nobody reads after the barrier that closed tile t - 1 - and waits for those pieces one tile later at the same place;
//   * everything a wave does to its keys - thresholds in the query's scaled units, pass masks / max-tree filter, appends, capacity check,
//     compaction, hand-over - is screen_kernel's float16 streaming code, unchanged: the candidate lists are identical in content.
// First version (measured, not kept): rigid lockstep - two workgroup barriers per subtile, waves 4-7 one barrier behind - ran 10 %
// SLOWER than screen_kernel: every half step then costs the slowest of eight waves plus the barrier (stamps: segments of 877 / 776
// cycles, 970 cycles per subtile spent waiting at the two barriers).
#ifndef HWER_PP_PK_MASKS
#define HWER_PP_PK_MASKS 1    // pass masks of the ping-pong kernel from v_pk_add_f32 differences (0: one v_sub_f32 per key)
#endif
#ifndef HWER_PP_SPARSE_MAX
#define HWER_PP_SPARSE_MAX 0.6   // expected passing keys per 32-row subtile and wave below which the ping-pong kernel takes the max-tree filter (4.0 measured: c4 +-0)
#endif
#ifndef HWER_PP_WIDE_PARK
#ifndef HWER_PP_MOVREL_POP
#define HWER_PP_MOVREL_POP 0   // round 6, MEASURED AND NOT KEPT (1 = on): few passing keys per subtile (shard regime) - a passing key read out of its ACCUMULATOR
#endif                         // REGISTER by a wave-uniform register index (the compiler emits s_set_gpr_idx_on / off around a v_mov) instead of parking the subtile's
                               // 16 TQ accumulators in LDS and reading one back.  No scratch, 227 VGPRs - and the shard's screen runs 8 % SLOWER (c4, same box,
                               // interleaved: 1.444 - 1.466 -> 1.573 - 1.576 ms, step 2.07 - 2.10 -> 2.19 - 2.24; profiles/r6_ab_movrel_pop.txt): the index-mode
                               // switches and the ballot -> readlane -> index chain of a round are dependent scalar / vector hand-overs that nothing hides, where the
                               // park's ds_writes issue back to back and the pops' LDS reads overlap with the partner wave's burst.
#define HWER_PP_WIDE_PARK 0   // 1: d = 256 under the ping-pong kernel filters by the expected-hits rule (its LDS has room for the key park)
#endif
#ifndef HWER_PP_TOKEN
#define HWER_PP_TOKEN 0      // 0: no token - SIMD partners are only staggered at the tile barrier (A/B switch)
#endif
#ifndef HWER_PP_PREFETCH
#define HWER_PP_PREFETCH 2   // A fragments requested this many k-steps ahead of their MFMAs (the MFMA segment has the pipe to itself)
#endif
template <int NG, int TQ, bool FOLD>   // FOLD: the threshold folded into the accumulators (pass-mask launches); max-tree launches (a.sparse_hits) run <.., false>
__global__ __launch_bounds__(512, 2) void screen16_pp_kernel(const ScreenArgs a) {
    constexpr int IB = 64, NSUB = 2, NSLOT = 3;
    constexpr int CH = NG;                          // 16-byte chunks per float16 row
    constexpr int TILE_CHUNKS = IB * CH;
    constexpr int DMA_PER_WAVE = TILE_CHUNKS / 512 > 0 ? TILE_CHUNKS / 512 : 1;   // 1-KiB pieces per wave and tile
    static_assert(TILE_CHUNKS % 512 == 0, "every wave issues the same number of pieces");
    constexpr int NQF = NG / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* xs = reinterpret_cast<float*>(smem_raw);                         // [NSLOT][IB][CH * 4]
    float* hn = xs + NSLOT * IB * CH * 4;                                   // [NSLOT][IB]
    float* keypark_all = hn + NSLOT * IB;                                   // [8][64][16 * TQ]
    int* tok = reinterpret_cast<int*>(keypark_all + 8 * 64 * 16 * TQ);      // [4]: whose turn it is on the matrix pipe of SIMD (wave & 3)
    float* hpriv_all = reinterpret_cast<float*>(tok + 4);                   // [8][TQ][64]: UNI - the tile's half norms in the wave's key units, one copy per query subtile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 2;       // 0: waves 0-3, 1: their SIMD partners 4-7 (one barrier behind)
    const int r = lane & 31, h = lane >> 5;
    if (blockIdx.x == 0) hdr_reset(a.hdr, tid, 512);
    if (tid < 4) tok[tid] = 0;        // waves 0-3 first
    const unsigned n_wg = gridDim.x;
    const unsigned wg = (n_wg & 7u) == 0u ? (blockIdx.x & 7u) * (n_wg >> 3) + (blockIdx.x >> 3) : blockIdx.x;   // XCD-aware order (screen_kernel)
    const int n_qb2 = a.n_qblocks >> 1;                                      // workgroups along the queries: two 4-wave query blocks each
    const int qblock = int(wg % unsigned(n_qb2)), split = int(wg / unsigned(n_qb2));
    const int list = split;
    const int64_t qw0 = int64_t(qblock) * (8 * 32 * TQ) + int64_t(wave) * 32 * TQ;
    float* keypark = keypark_all + wave * (64 * 16 * TQ);

    half8_t qf[TQ][NQF];
    int e_tot[TQ];
    float2_t nscale[TQ];
    float tau[TQ];
    int cnt[TQ];
    bool first[TQ], scaled[TQ], seeded[TQ];
    uint32_t soff[TQ];
    uint64_t* const wg_cand = a.cand + ((int64_t(list) * a.bq + qw0) * 2) * CAPH;
    // UNI (launches that do not fold the threshold; round 4): ONE key exponent per wave.  The float16 fragments of a query are scaled by the
    // query's own power of two (its largest element in [2^13, 2^14)); scaled down further to the smallest exponent of the wave's queries they keep
    // every bit (float16 is floating point; the shift is at most 10 binades, so an element would have to be 2^-17 of its row's largest to leave the
    // normal range) - and then the accumulators' initial value, half norm x 2^e, is the same for every lane: each wave scales a tile's 64 half norms
    // ONCE (one multiply per lane and tile) into a private LDS row, and a subtile's set-up is eight ds_read_b128 straight into the accumulators
    // instead of four reads + 16 v_pk_mul_f32.  A wave whose queries' exponents differ by more than 10 keeps the per-lane scaling.
#ifndef HWER_PP_UNIFORM_EXP
#define HWER_PP_UNIFORM_EXP 1
#endif

    // (one query subtile per wave only: with two, eight LDS reads - or four reads and sixteen register copies - in front of every burst measured
    // 3 - 6 % SLOWER than four reads + 16 v_pk_mul_f32 on the 10 M-row launch; with one - d = 256 - the screen stage gains 4.6 %)
    constexpr bool UNI = HWER_PP_UNIFORM_EXP && !FOLD && TQ == 1;
    bool uni = false;      // (wave-uniform)
    int delta[TQ];
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        e_tot[b] = a.x_exp + a.qexp[qw0 + 32 * b + r];
        delta[b] = 0;
    }
    if constexpr (UNI) {
        int emin = 0x7FFFFFFF, emax = -0x7FFFFFFF;
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            if (qw0 + 32 * b + r < a.B) { emin = min(emin, e_tot[b]); emax = max(emax, e_tot[b]); }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            emin = min(emin, __shfl_xor(emin, off, kWave));
            emax = max(emax, __shfl_xor(emax, off, kWave));
        }
        uni = emin != 0x7FFFFFFF && emax - emin <= 10;
        if (uni) {
#pragma unroll
            for (int b = 0; b < TQ; ++b) {
                delta[b] = (qw0 + 32 * b + r < a.B) ? e_tot[b] - emin : 0;
                e_tot[b] = emin;
            }
        }
    }
    float* const hpriv = hpriv_all + wave * (TQ * 64);
    const float hscale_w = ldexpf(1.0f, e_tot[0]);      // (UNI: the wave's key units)
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int64_t q = qw0 + 32 * b + r;
        const bool live = q < a.B;
        nscale[b] = float2_t{ldexpf(1.0f, e_tot[b]), ldexpf(1.0f, e_tot[b])};      // (+: the accumulators hold tau - key, see below)
        const float t0 = (a.tau0 && live) ? a.tau0[q] : a.debug_tau0;
        seeded[b] = a.tau0 != nullptr && live && t0 > -INFINITY;
        tau[b] = live ? ldexpf(t0, e_tot[b]) : INFINITY;
        cnt[b] = 0;
        first[b] = true;
        scaled[b] = false;
        soff[b] = (uint32_t(32 * b + r) * 2u + uint32_t(h)) * CAPH;
    }
    // THE THRESHOLD LIVES IN THE ACCUMULATOR (round 4).  The query fragments are negated once, here, and a subtile's accumulators start at
    // (half norm + tau) 2^e, so what the MFMAs leave is  tau' - key'  (scaled units): a key passes exactly when its accumulator is negative, the
    // pass mask is the 32 sign bits (one v_alignbit each), and the 16 v_pk_add_f32 of "tau - key" per subtile are gone (packed float32 adds cost
    // two issue slots each beside MFMAs; this loop's time is what its waves issue).  The key of a passing item is recovered as tau' - accumulator
    // when it is appended (one subtraction per APPENDED key).  tau' is a key of the same query or absent: a lane whose threshold is still open
    // (-inf: everything passes) folds 0 and forces its 16 mask bits (tau_or); a lane without a query (+inf: nothing passes) folds 3e38.
    // Rounding: the partial sums are bounded by |tau'| + |key'| <= 2 (||q|| ||x||max + hnorm_max) instead of 1 x that, and the recovery rounds once
    // more: (2 d + 7) u M to first order, inside the (2 d + 16) u M of the select stage's bound (knn_select.hip: e32_bound); keys that do NOT pass
    // have tau' - acc <= tau' in this same arithmetic, which is all the certificate's "every item outside the list is below the list" needs.
    // The max-tree filter of sparse launches (a.sparse_hits: one compare per 16 keys) has no subtraction to lose and keeps plain -key' accumulators:
    // FOLD = false (the launcher picks the instantiation).
    constexpr bool fold = FOLD;
    float tau_f[TQ];
    uint32_t tau_or = 0u;        // 16 forced mask bits per query subtile whose threshold is open
    auto fold_tau = [&]() {
        tau_or = 0u;
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            const bool open_b = tau[b] == -INFINITY, dead_b = tau[b] == INFINITY;
            // no query (+inf: nothing passes): a huge finite threshold - the lane's fragments are zero, its accumulators stay positive
            tau_f[b] = open_b ? 0.f : (dead_b ? 3.0e38f : tau[b]);
            if (open_b) tau_or |= 0xFFFFu << (16 * b);
        }
    };
    fold_tau();
    const int64_t i_begin = int64_t(split) * a.split_items;
    const int64_t i_end = std::min<int64_t>(i_begin + a.split_items, a.n_items);
    const int n_tiles = int((i_end - i_begin + IB - 1) / IB);
    const int tile_mul = a.scaled_first ? coprime_step(n_tiles) : 1;
    auto perm_next = [&](int p) { p += tile_mul; return p >= n_tiles ? p - n_tiles : p; };
    constexpr int PIECE_ROWS = 64 / CH > 0 ? 64 / CH : 1;
    const uint32_t row_bytes = uint32_t(a.ldx) * 2u;
    const uint32_t lane_row = uint32_t(lane / CH), lane_chunk = uint32_t(lane % CH);
    const uint32_t xs_addr0 = __builtin_amdgcn_readfirstlane(lds_addr(xs)) + uint32_t(wave) * 1024u;
    const uint32_t hn_addr0 = __builtin_amdgcn_readfirstlane(lds_addr(hn));
    const char* const x_split = reinterpret_cast<const char*>(a.X16) + i_begin * a.ldx * 2;
    const float* const hn_split = a.hnorm + i_begin;
    const uint32_t tile_bytes = uint32_t(IB) * row_bytes;
    auto tile_dma = [&](int slot, int ptile) {     // this wave's pieces of catalogue tile `ptile` -> ring slot `slot`
        const uint32_t xb_addr = xs_addr0 + uint32_t(slot) * uint32_t(IB * CH * 16);
        const char* tile_ptr = x_split + uint64_t(uint32_t(ptile)) * tile_bytes;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {   // piece = wave + 8 i covers rows piece * PIECE_ROWS + lane / CH
            const uint32_t row = uint32_t(wave * PIECE_ROWS) + uint32_t(8 * i * PIECE_ROWS) + lane_row;
            const uint32_t off = __umul24(row, row_bytes) + ((lane_chunk ^ uint32_t(swz<CH>(int(row)))) << 4);
            lds_dma16(tile_ptr, off, xb_addr + uint32_t(i) * 8192u);
        }
        if (wave == 0) lds_dma4(hn_split + int64_t(ptile) * IB, uint32_t(lane) * 4u, hn_addr0 + uint32_t(slot) * uint32_t(IB * 4));
    };
    auto append = [&](int b, uint64_t key) {
        wg_cand[soff[b] + uint32_t(cnt[b])] = key;
        ++cnt[b];
    };
    __builtin_amdgcn_s_setprio(HWER_F16_REST_PRIO);
    // prologue: tiles 0 and 1 requested, then the query fragments
    int p_cur = 0;        // perm(t)
    int p_dma = 0;        // perm(next tile to request)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s < n_tiles) tile_dma(s, p_dma);
        p_dma = perm_next(p_dma);
    }
#pragma unroll
    for (int b = 0; b < TQ; ++b)
#pragma unroll
        for (int g = 0; g < NQF; ++g)
            qf[b][g] = reinterpret_cast<const half8_t*>(a.Qf16)[((qw0 / 32 + b) * NQF + g) * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < TQ; ++b)
#pragma unroll
        for (int g = 0; g < NQF; ++g) qf[b][g] = -qf[b][g];      // accumulators = tau' - key' (above)
    if (UNI && uni) {
#pragma unroll
        for (int b = 0; b < TQ; ++b) {
            const _Float16 down = (_Float16)ldexpf(1.0f, -delta[b]);      // (2^0 ... 2^-10: exact)
#pragma unroll
            for (int g = 0; g < NQF; ++g) qf[b][g] = qf[b][g] * down;
        }
    }
    // the fragments are made opaque HERE: the compiler then waits for their loads in front of this statement, not - per register, ending
    // in vmcnt(0), a drain of the pieces and appends in flight - inside the first MFMA segment of every tile
#pragma unroll
    for (int b = 0; b < TQ; ++b)
#pragma unroll
        for (int g = 0; g < NQF; ++g) make_opaque(qf[b][g]);
    __syncthreads();      // tiles 0 and 1 have landed for every wave, the tokens are set (the only fenced barrier of the kernel)
    int* const my_tok = tok + (wave & 3);
    auto tok_acquire = [&]() {
        if constexpr (HWER_PP_TOKEN) {
            while (__hip_atomic_load(my_tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != role) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto tok_release = [&]() {
        if constexpr (HWER_PP_TOKEN) {
            __builtin_amdgcn_sched_barrier(0);        // behind the last MFMA of the burst (issue order), not before
            __hip_atomic_store(my_tok, 1 - role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, pt4 = 0, pa_w1 = 0, pa_mfma = 0, pa_w2 = 0, pa_filt = 0, pa_drain = 0, pclk0 = 0, prt0 = 0;
    if (DIAG_STAMPS) { pclk0 = __builtin_amdgcn_s_memtime(); prt0 = __builtin_amdgcn_s_memrealtime(); }
    for (int t = 0; t < n_tiles; ++t) {
        const int slot = t % NSLOT;
        const float* xb_tile = xs + slot * IB * CH * 4;
        const float* hb_tile = hn + slot * IB;
        const int64_t i0_tile = i_begin + int64_t(p_cur) * IB;
        p_cur = perm_next(p_cur);
        if (UNI && uni) {      // (tile t is complete in LDS: this wave has passed tile t - 1's barrier)
            const float hs = hb_tile[lane] * hscale_w;
#pragma unroll
            for (int b = 0; b < TQ; ++b) hpriv[64 * b + lane] = hs;
            wave_lds_sync();
        }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const float* xb = xb_tile + sub * 32 * CH * 4;
            const float* hb = hb_tile + sub * 32;
            const int64_t i0 = i0_tile + 32 * sub;
            // ---- accumulator set-up and the first fragments: before the token ----------------------------------------------
            f32x16 acc[TQ];
            if (UNI && uni) {      // (wave-uniform) the scaled half norms straight into the accumulators
#pragma unroll
                for (int b = 0; b < TQ; ++b)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const float4 hv = *reinterpret_cast<const float4*>(hpriv + 64 * b + 32 * sub + 8 * rg + 4 * h);
                        acc[b][4 * rg + 0] = hv.x;
                        acc[b][4 * rg + 1] = hv.y;
                        acc[b][4 * rg + 2] = hv.z;
                        acc[b][4 * rg + 3] = hv.w;
                    }
            } else
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 hv = *reinterpret_cast<const float4*>(hb + 8 * rg + 4 * h);
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    if constexpr (fold) {   // tau' - key': two v_pk_fma_f32 (four plain v_fma_f32 held apart by a compiler barrier measured 3 - 4 % slower)
                        const float2_t tf2 = {tau_f[b], tau_f[b]};
                        const float2_t lo2 = __builtin_elementwise_fma(float2_t{hv.x, hv.y}, nscale[b], tf2), hi2 = __builtin_elementwise_fma(float2_t{hv.z, hv.w}, nscale[b], tf2);
                        acc[b][4 * rg + 0] = lo2.x;
                        acc[b][4 * rg + 1] = lo2.y;
                        acc[b][4 * rg + 2] = hi2.x;
                        acc[b][4 * rg + 3] = hi2.y;
                    } else {                // -key': exact scaling, two keys per v_pk_mul_f32
                        const float2_t lo2 = float2_t{hv.x, hv.y} * nscale[b], hi2 = float2_t{hv.z, hv.w} * nscale[b];
                        acc[b][4 * rg + 0] = lo2.x;
                        acc[b][4 * rg + 1] = lo2.y;
                        acc[b][4 * rg + 2] = hi2.x;
                        acc[b][4 * rg + 3] = hi2.y;
                    }
                }
            }
            {
                int sw = swz<CH>(r);
                asm volatile("" : "+v"(sw));
                const unsigned char* abase = reinterpret_cast<const unsigned char*>(xb) + r * (NG * 16);
                auto a_ptr = [&](int g) { return abase + 16 * ((2 * g + h) ^ sw); };
                constexpr int PF = HWER_PP_PREFETCH < NQF ? HWER_PP_PREFETCH : NQF;
                half8_t avq[PF];
#pragma unroll
                for (int g = 0; g < PF; ++g) avq[g] = *reinterpret_cast<const half8_t*>(a_ptr(g));
                // ---- MFMA burst: only while this wave holds its SIMD's token (the partner filters meanwhile) ---------------
                STAMP(pt0);
                tok_acquire();
                STAMP(pt1);
                __builtin_amdgcn_s_setprio(HWER_F16_MFMA_PRIO);
#pragma unroll
                for (int g = 0; g < NQF; ++g) {
                    const half8_t av = avq[g % PF];
                    if (g + PF < NQF) avq[g % PF] = *reinterpret_cast<const half8_t*>(a_ptr(g + PF));
#pragma unroll
                    for (int b = 0; b < TQ; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, qf[b][g], acc[b], 0, 0, 0);
                }
                tok_release();
                __builtin_amdgcn_s_setprio(HWER_F16_REST_PRIO);
            }
            if (DIAG_STAMPS) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[TQ - 1][15])); pt2 = stamp(); pt3 = pt2; }
            // ---- the tile barrier (ring safety: every wave is done READING tile t and has waited for its pieces of tile t + 1), STAGGERED:
            // waves 4-7 take it here, between the last MFMA burst of the tile and its filter segment, waves 0-3 behind that filter segment
            // (below).  After the barrier waves 0-3 go straight into the next tile's first burst while their partners still filter - the
            // two waves of a SIMD leave every barrier half a step apart, and from there bursts and filter segments interleave by themselves.
            if (sub == NSUB - 1 && role == 1) {
                __builtin_amdgcn_s_barrier();
                if (DIAG_STAMPS) { pt3 = stamp(); pa_w2 += pt3 - pt2; }
            }
            // ---- filter segment (the partner runs its MFMAs) --------------------------------------------------------------
            if (sub == 0) {
                // ONE drain per tile, in front of this segment's appends: outstanding are this wave's pieces of tile t + 1 (requested a
                // tile ago) and the appends of the last tile's filter segments (a whole MFMA burst old or more) - the wait is free.
                // Then the pieces of tile t + 2 into the slot tile t - 1 held: nobody reads that one after this tile's barrier.
                unsigned long long pd0 = 0;
                STAMP(pd0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (DIAG_STAMPS) pa_drain += stamp() - pd0;
                if (t + 2 < n_tiles) tile_dma((t + 2) % NSLOT, p_dma);
                p_dma = perm_next(p_dma);
            }
            if (a.sparse_hits) {   // (wave-uniform) max-tree variant: see screen_kernel
                const uint32_t item0 = uint32_t(i0) + 4 * h;
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    const float m0 = fminf(fminf(acc[b][0], acc[b][1]), acc[b][2]), m1 = fminf(fminf(acc[b][3], acc[b][4]), acc[b][5]);
                    const float m2 = fminf(fminf(acc[b][6], acc[b][7]), acc[b][8]), m3 = fminf(fminf(acc[b][9], acc[b][10]), acc[b][11]);
                    const float m4 = fminf(fminf(acc[b][12], acc[b][13]), acc[b][14]);
                    const float mn = fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), acc[b][15]));
                    // (not folded: the accumulators hold -key', compared with -tau': +inf where the threshold is open, -inf where there is no query)
                    // (folded accumulators - a diagnostic switch forcing the max tree onto a pass-mask launch - compare with 0; open thresholds
                    // do not occur there: the switch needs seeded launches)
                    const float ntau = fold ? 0.f : -tau[b];
                    if (__any(mn < ntau)) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            if (acc[b][q] < ntau) append(b, pack_key(ldexpf((fold ? tau_f[b] : 0.f) - acc[b][q], -e_tot[b]), item0 + (q & 3) + 8 * (q >> 2)));
                        }
                    }
                }
            } else {
                uint32_t mask = 0;
#pragma unroll
                for (int b = TQ - 1; b >= 0; --b) {
                    if constexpr (fold) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(acc[b][q]), 31);   // sign of tau' - key'
                    } else {   // not folded: every pass-mask launch with one query subtile per wave (d = 256; launch_screen16_pp_t picks FOLD = false for TQ == 1)
                        const float2_t tau2 = {tau[b], tau[b]};
#pragma unroll
                        for (int q = 0; q < 16; q += 2) {
                            const float2_t df = tau2 + float2_t{acc[b][q], acc[b][q + 1]};   // tau' - key' with -key' accumulators
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(df.x), 31);
                            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(df.y), 31);
                        }
                    }
                }
                if constexpr (fold) mask |= tau_or;      // lanes with an open threshold
                if (ABLATE(2)) mask = 0u;      // (diagnostic: masks computed, nothing popped)
                if (__any(mask != 0u)) {
                    const uint32_t item0 = uint32_t(i0) + 4 * h;
                    float* park = keypark + lane;
                    // (only the query subtiles that have a passing key are parked: on a 1.25 M-row shard a subtile of a wave has a hit 57 % of the time -
                    // c4 screen -1.5 %.  Measured with it and left out: the second subtile's accumulator set-up inside the first subtile's burst (second
                    // accumulator set) +-0; the fragments' swizzled offsets kept in registers: +-0, c4full +0.7 % - both sit in the MFMAs' shadow already)
                    const bool use_park = !HWER_PP_MOVREL_POP || a.many_hits;      // (wave-uniform)
#pragma unroll
                    for (int b = 0; b < TQ; ++b) {
                        if (use_park && (TQ == 1 || a.many_hits || __any(((mask >> (16 * b)) & 0xFFFFu) != 0u))) {
#pragma unroll
                            for (int q = 0; q < 16; ++q) park[(16 * b + q) * 64] = acc[b][q];
                        }
                    }
                    if (ABLATE(8)) mask = 0u;  // (diagnostic: masks + park, no pops)
                    if (a.many_hits) {
                        // (wave-uniform) many passing keys per subtile (single-list plans: c3 lets 2.25 K' keys of a 26 744-row catalogue through -
                        // 0.7 per lane and subtile): the pop loop below is a chain of one LDS round trip per pop and runs max-over-lanes times
                        // (3 - 4).  Here a round takes up to FOUR set bits per lane, issues their LDS reads together and appends branch-free (the
                        // stream of a key is selected arithmetically, not by a divergent branch): one LDS latency per round instead of four.
                        while (__any(mask != 0u)) {
                            int bits[4];
                            float kv4[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                bits[j] = mask != 0u ? __builtin_ctz(mask) : -1;
                                mask &= mask - 1u;      // (0 stays 0)
                                const int bb = bits[j] < 0 ? 0 : bits[j];
                                kv4[j] = park[((bb & 16) + 15 - (bb & 15)) * 64];
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                if (bits[j] >= 0) {
                                    const int q = 15 - (bits[j] & 15);
                                    const bool second = TQ == 2 && (bits[j] & 16) != 0;
                                    const float kval = ldexpf((fold ? (second ? tau_f[TQ - 1] : tau_f[0]) : 0.f) - kv4[j], -(second ? e_tot[TQ - 1] : e_tot[0]));
                                    const uint64_t key = pack_key(kval, item0 + (q & 3) + 8 * (q >> 2));
                                    const uint32_t pos = second ? soff[TQ - 1] + uint32_t(cnt[TQ - 1]) : soff[0] + uint32_t(cnt[0]);
                                    if (!ABLATE(1)) wg_cand[pos] = key;
                                    else asm volatile("" :: "v"(key), "v"(pos));
                                    if constexpr (TQ == 2) { cnt[0] += second ? 0 : 1; cnt[1] += second ? 1 : 0; }
                                    else cnt[0] += 1;
                                }
                            }
                        }
                    } else if (HWER_PP_MOVREL_POP) {
                        // One round per DISTINCT set bit of the wave (1.7 passing keys per subtile and wave on a 1.25 M-row shard: one or two rounds): the bit of the
                        // first lane that still has one is broadcast (v_readlane), the accumulator it names is selected by a wave-uniform register index, and
                        // every lane that has this bit set appends - no LDS park (16 - 32 ds_write per subtile with a hit), no LDS round trip per pop.
                        uint64_t act = __ballot(mask != 0u);
                        while (act != 0ull) {
                            const int src = __builtin_ctzll(act);
                            const int bit = __builtin_amdgcn_readlane(__builtin_ctz(mask | 0x80000000u), src);      // (wave-uniform)
                            const int q = 15 - (bit & 15);
                            const bool second = TQ == 2 && (bit & 16) != 0;
                            const float accv = second ? acc[TQ - 1][q] : acc[0][q];
                            const bool mine = ((mask >> bit) & 1u) != 0u;
                            mask &= ~(1u << bit);
                            if (mine) {
                                const float kval = ldexpf((fold ? (second ? tau_f[TQ - 1] : tau_f[0]) : 0.f) - accv, -(second ? e_tot[TQ - 1] : e_tot[0]));
                                const uint64_t key = pack_key(kval, item0 + (q & 3) + 8 * (q >> 2));
                                if (second) append(TQ - 1, key);
                                else append(0, key);
                            }
                            act = __ballot(mask != 0u);
                        }
                    } else
                    while (__any(mask != 0u)) {
                        if (mask != 0u) {
                            const int bit = __builtin_ctz(mask);
                            mask &= mask - 1u;
                            const int q = 15 - (bit & 15);
                            float kval = park[((bit & 16) + q) * 64];
                            kval = ldexpf((fold ? ((TQ == 2 && (bit & 16)) ? tau_f[TQ - 1] : tau_f[0]) : 0.f) - kval, -((TQ == 2 && (bit & 16)) ? e_tot[TQ - 1] : e_tot[0]));
                            const uint64_t key = pack_key(kval, item0 + (q & 3) + 8 * (q >> 2));
                            if constexpr (TQ == 1) {
                                append(0, key);
                            } else {
                                if ((bit & 16) != 0) append(1, key);
                                else append(0, key);
                            }
                        }
                    }
                }
            }
            if (sub == NSUB - 1) {
                // ---- stream capacity (every 64 rows) ----------------------------------------------------------------------
#pragma unroll
                for (int b = 0; b < TQ; ++b) {
                    uint64_t need = __ballot(cnt[b] > CAPH - 32);
                    const bool refold = need != 0ull;      // (wave-uniform) a compaction sets a lane's threshold: fold it again
                    while (need != 0ull) {
                        const int jj = (__builtin_ctzll(need)) & 31;
                        const int c0 = __shfl(cnt[b], jj, kWave), c1 = __shfl(cnt[b], jj + 32, kWave);
                        const int was_first = __shfl(int(first[b]), jj, kWave);
                        uint64_t* s0 = wg_cand + (uint32_t(32 * b + jj) * 2u) * CAPH;
                        int rank = a.kprime;
                        if (was_first && a.scaled_first) {
                            const float mu = float(a.kprime) * float(c0 + c1) / float(i_end - i_begin);
                            const int mrank = int(mu + 4.0f * sqrtf(mu) + 5.0f);
                            if (2 * mrank < a.kprime) rank = mrank;
                        }
                        float T;
                        int n0, n1;
                        compact_query(s0, s0 + CAPH, c0, c1, rank, rank == a.kprime ? COMPACT_SLACK : min(COMPACT_SLACK, rank / 4 + 1), T, n0, n1);
                        if (r == jj) {
                            tau[b] = ldexpf(T, e_tot[b]);
                            cnt[b] = h == 0 ? n0 : n1;
                            scaled[b] = scaled[b] || (rank != a.kprime);
                            first[b] = false;
                        }
                        need &= ~((1ull << jj) | (1ull << (jj + 32)));
                    }
                    if (refold) fold_tau();
                }
            }
            if (DIAG_STAMPS) { pt4 = stamp(); pa_w1 += pt1 - pt0; pa_mfma += pt2 - pt1; pa_filt += pt4 - pt3; }
            if (sub == NSUB - 1 && role == 0) {
                __builtin_amdgcn_s_barrier();
                if (DIAG_STAMPS) pa_w2 += stamp() - pt4;
            }
        }
    }
    if (DIAG_STAMPS && lane == 0) {   // same 8-word record per wave as screen_kernel: [1] MFMA segments, [2] filter segments, [3] token waits, [4] tile barriers
        unsigned long long* o = a.debug_stamps + (size_t(blockIdx.x) * 8 + wave) * 8;
        o[0] = 0; o[1] = pa_mfma; o[2] = pa_filt; o[3] = pa_drain /* (diagnosis: the once-per-tile vmcnt drain, part of the filter segments) */; o[4] = pa_w2; o[5] = 2ull * n_tiles;
        o[6] = __builtin_amdgcn_s_memtime() - pclk0; o[7] = __builtin_amdgcn_s_memrealtime() - prt0;
    }
    // ---- hand-over: screen_kernel's ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < TQ; ++b) {
        const int partner = __shfl_xor(cnt[b], 32, kWave);
        const int total = cnt[b] + partner;
        const bool underfull = scaled[b] && total < a.kprime;
        const bool seeded_short = seeded[b] && first[b] && total < a.kprime;
        uint64_t need = __ballot(total > a.kprime) & 0xFFFFFFFFull;
        while (need != 0ull) {
            const int jj = __builtin_ctzll(need);
            const int c0 = __shfl(cnt[b], jj, kWave), c1 = __shfl(cnt[b], jj + 32, kWave);
            uint64_t* s0 = wg_cand + (uint32_t(32 * b + jj) * 2u) * CAPH;
            float T;
            int n0, n1;
            compact_query(s0, s0 + CAPH, c0, c1, a.kprime, 0, T, n0, n1);
            if (r == jj) cnt[b] = h == 0 ? n0 : n1;
            need &= ~(1ull << jj);
        }
        const int64_t q = qw0 + 32 * b + r;
        a.cand_cnt[(int64_t(list) * a.bq + q) * 2 + h] = cnt[b] | (underfull ? CNT_UNDERFULL : 0) | (seeded_short ? CNT_SEEDED_SHORT : 0);
    }
}

__global__ __launch_bounds__(256) void repack_queries_kernel(const float* __restrict__ Q, int64_t B, int ldx, int ng, float4* __restrict__ Qf, int64_t n_out) {
    const int64_t o = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (o >= n_out) return;
    const int lane = int(o & 63);
    const int64_t sg = o >> 6;
    const int g = int(sg % ng);
    const int64_t q = (sg / ng) * 32 + (lane & 31);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < B) v = *reinterpret_cast<const float4*>(Q + q * ldx + 8 * g + 4 * (lane >> 5));
    Qf[o] = v;
}

size_t repacked_queries_bytes(int64_t B, int ldx) { return size_t(cdiv64(B, 32)) * 32 * size_t(ldx) * 4; }

int repack_queries(const float* Q32, int64_t B, int64_t B_padded, int ldx, float* Qf, hipStream_t stream) {
    const int ng = ldx / 8;
    const int64_t n_out = cdiv64(B_padded, 32) * ng * 64;
    hipLaunchKernelGGL(repack_queries_kernel, dim3(unsigned(cdiv64(n_out, 256))), dim3(256), 0, stream, Q32, B, ldx, ng,
                       reinterpret_cast<float4*>(Qf), n_out);
    return check_launch("knn_repack_queries");
}

// =====================================================================================================
// host side: plan + launch
// =====================================================================================================
#ifndef HWER_SEED_MAX_TILES
#define HWER_SEED_MAX_TILES 8192
#endif
constexpr int64_t SEED_MAX_TILES = HWER_SEED_MAX_TILES;   // largest sample: 262 144 rows
// K' = candidate keys kept per query.  The certificate needs some kept key below t_k - 2 E (E = the screen's key error bound, t_k ~ the k-th
// largest key, bracketed to within 12 ranks): the slack K' - k has to cover the ~12 keys of the bracket plus the keys that lie within 2 E of the
// k-th.  Under a float32 screen E is ~1e-5 of the key range and SLACK = 56 is ample; under the FLOAT16 screen E ~ 2^-10 ||q|| ||x|| is a fixed
// absolute width while the key spacing at the k-th key shrinks like 1 / (N pdf): on a 10 M x 256 catalogue ~28 keys sit inside 2 E and 0.3 % of
// the queries came back uncertified (round 6: c5full on ONE GPU, 12 of 4 096 - more than the on-device exhaustive pass takes).  Rule: with
// m = the expected number of keys within 2 E of the k-th key - unit rows with iid components, cosines ~ N(0, 1 / d): m = N phi(z_k) sqrt(d) 2 E,
// z_k = the upper k / N quantile - K' grows by 64 until K' - k - 12 - m >= 3.7 sqrt(m) (~1e-4 uncertified queries per query).  A PERFORMANCE
// heuristic only: a catalogue that does not look like the model gets more (or fewer) exhaustive passes, never a wrong answer.
int kprime_for(int k, int64_t n_items, int ldx, bool f16_screen) {
    int kp = ((k + SLACK + 63) / 64) * 64;
    if (!f16_screen || n_items <= DENSE_MAX_ITEMS || k >= n_items) return kp;
    const double p_k = double(k) / double(n_items);
    double lo = 0.0, hi = 12.0;
    for (int it = 0; it < 60; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (0.5 * erfc(mid * 0.7071067811865476) > p_k) lo = mid; else hi = mid;
    }
    const double z = 0.5 * (lo + hi);
    const double m = double(n_items) * exp(-0.5 * z * z) * 0.3989422804014327 * sqrt(double(ldx)) * 2.0 * 1.02 * 0.0009765625;
    while (kp + 64 <= KPRIME_MAX && double(kp - k) - 12.0 - m < 3.7 * sqrt(m)) kp += 64;
    return kp;
}

static ScreenPlan make_screen_plan_uncached(int64_t B, int64_t n_items, int ldx, int k, uint32_t options);
// The plan is a pure function of its arguments (production builds: no environment switches) and a search asks for it several times (workspace
// size, query padding, screen, select) - its seed-rank loops evaluate lgamma / exp tails, up to O(groups^2) terms: the last plan is kept per thread
// (advisor finding, round 5).
ScreenPlan make_screen_plan(int64_t B, int64_t n_items, int ldx, int k, uint32_t options) {
#if defined(HWER_DIAG_STAMPS) || defined(HWER_DIAG_ABLATE)
    return make_screen_plan_uncached(B, n_items, ldx, k, options);      // (diagnostic builds read tuning overrides from the environment on every call)
#else
    struct Key { int64_t B, n; int ldx, k; uint32_t opt; };
    thread_local Key last{-1, -1, 0, 0, 0};
    thread_local ScreenPlan plan{};
    if (last.B != B || last.n != n_items || last.ldx != ldx || last.k != k || last.opt != options) {
        plan = make_screen_plan_uncached(B, n_items, ldx, k, options);
        last = Key{B, n_items, ldx, k, options};
    }
    return plan;
#endif
}

static ScreenPlan make_screen_plan_uncached(int64_t B, int64_t n_items, int ldx, int k, uint32_t options) {
    ScreenPlan p{};
    p.ng = ldx / 8;
    p.kprime = kprime_for(k, n_items, ldx, (options & HWER_KNN_F16_SCREEN) != 0 && ldx >= 64);
    // queries per wave: TQ = 2 subtiles (64 queries, 256 per workgroup) while the B fragments fit the register budget
    // (NG * TQ float4 <= 32 float4) AND the query blocks alone fill the 512 workgroup slots.  Below that (B <= 65 536) the
    // launch is split over the items anyway, and TQ = 1 is the better trade: twice the query blocks -> half the item splits
    // -> half the candidate lists the select stage reduces, and shorter per-wave filter / DMA phases between the MFMA
    // phases (measured, C4 4 096 x 1.25 M: screen 12.5 -> 11.9 ms, select 1.0 -> 0.56 ms; a third resident workgroup per
    // CU - 168 VGPRs - was measured too and does not help).
    p.tq = (p.ng <= 16 && B > 65536) ? 2 : 1;
    // With seeded thresholds (below) a candidate list is nearly free, so the item splits that TQ = 2 doubles no longer cost a
    // warm-up each, while the per-tile barrier / DMA issue / accumulator set-up is shared by twice the queries
    // (c4, 4 096 x 1.25 M: screen 3.20 -> 2.53 ms, select + list reduction 0.41 -> 0.55 ms).
    // (long lists only: on a 26 744-row catalogue - c3's last 7 421 users - the extra lists cost more in the list reduction)
    if (p.ng <= 16 && B >= 2048 && n_items >= 65536 && !(options & (HWER_KNN_NO_SEED | HWER_KNN_EXACT_THRESHOLDS | HWER_KNN_NO_DENSE))) p.tq = 2;
    if (const char* e = diag_env("HWER_SCREEN_TQ")) p.tq = atoi(e) == 1 ? 1 : ((atoi(e) == 2 && p.ng <= 16 && B > 32) ? 2 : p.tq);   // tuning override
    const int qpw = 32 * p.tq;
    p.wq = B > 2 * qpw ? 4 : (B > qpw ? 2 : 1);
    p.qb = p.wq * qpw;
    p.n_qblocks = int(cdiv64(B, p.qb));
    // Round 4: the float16 streaming screen runs as 512-thread ping-pong workgroups (screen16_pp_kernel) that cover TWO query blocks
    // each: whole pairs of blocks (the padding queries are dead lanes, as in any last block)
    p.pp = ((options & HWER_KNN_F16_SCREEN) && !(options & HWER_KNN_NO_PINGPONG) && p.ng >= 8 && p.wq == 4 &&
            !(n_items <= DENSE_MAX_ITEMS && !(options & HWER_KNN_NO_DENSE))) ? 1 : 0;
    // (the ping-pong kernel streams 64-row tiles whatever the A/B switches say: a build with 32-row tiles - HWER_SCREEN16_IB=32, or HWER_WIDE64=0 at
    // d = 256 - aligns the item splits to 32 rows and keeps the two-workgroup kernel, or a split's last tile would run into the next split's rows)
    if (p.pp && !(SCREEN16_IB == 64 && (p.ng <= 16 || WIDE64))) p.pp = 0;
    if (p.pp) p.n_qblocks = (p.n_qblocks + 1) & ~1;
    p.bq = int64_t(p.n_qblocks) * p.qb;
    const int wi = 4 / p.wq;
    // item splits: every split re-pays the selection warm-up (~4 K' appends + a compaction per query), so split
    // only when the query blocks alone cannot fill the 512 workgroup slots (256 CUs x 2 resident), and keep
    // >= 4096 items per list (shorter lists were measured slower: more list-reduce work than the extra parallelism buys).  Callers chunk B to multiples of 512 * qb queries (DeviceIndex.QUERY_CHUNK).
    int64_t slots = 512;
    if (const char* e = diag_env("HWER_SCREEN_SLOTS")) slots = std::max(1, atoi(e));   // tuning override
    int64_t splits = std::max<int64_t>(1, slots / p.n_qblocks);
    // (seeded thresholds - decided below from the same quantities - make a list nearly free: then lists may be 4x shorter)
    const bool seedable = !(options & (HWER_KNN_NO_SEED | HWER_KNN_EXACT_THRESHOLDS)) && n_items > DENSE_MAX_ITEMS;
    int64_t min_list = seedable ? 1024 : 4096;
    if (const char* e = diag_env("HWER_SCREEN_SPLITS")) splits = std::max(1, atoi(e));  // tuning / ablation overrides
    if (const char* e = diag_env("HWER_SCREEN_MINLIST")) min_list = std::max(32, atoi(e));
    const int64_t max_splits = std::max<int64_t>(1, n_items / (min_list * wi));
    splits = std::min(splits, max_splits);
    const int ib = ((options & HWER_KNN_F16_SCREEN) && (p.ng <= 16 || WIDE64)) ? SCREEN16_IB : 32;   // item splits start on tile boundaries of the kernel that will run
    int64_t per = cdiv64(cdiv64(n_items, splits), ib) * ib;
    splits = cdiv64(n_items, per);
    p.n_splits = int(splits);
    p.split_items = per;
    p.n_lists = p.n_splits * wi;
    p.scaled_first = 1;
    p.seed = 0;
    // Small catalogues: materialise every key (8 B of HBM traffic per pair) instead of streaming selection, whose
    // warm-up (~4 K' appends + a compaction per query and list) dominates when N is only a few K'.
    p.dense = n_items <= DENSE_MAX_ITEMS && !(options & HWER_KNN_NO_DENSE);
    if (p.dense) {
        const int64_t n_tiles = cdiv64(n_items, 32);
        // one full round of workgroups: 256 CUs x 2 resident = 512 slots
        int64_t sp = std::min<int64_t>(n_tiles, std::max<int64_t>(1, 512 / p.n_qblocks));
        if (const char* e = diag_env("HWER_DENSE_SPLITS")) sp = std::min<int64_t>(n_tiles, std::max(1, atoi(e)));   // tuning override
        int64_t per_d = cdiv64(n_tiles, sp) * 32;
        p.n_splits = int(cdiv64(n_items, per_d));
        p.split_items = per_d;
        p.n_lists = 0;
        p.ld_dense = n_tiles * 32;
    }
    // Seeded thresholds (streaming mode).  The sample: 0.6 - 9 % of the catalogue's 32-row tiles at a regular stride (1 024 to
    // 65 536 items); tau0 = the sample key of rank r, which stands for c K' keys of the whole catalogue.  The fraction of the
    // catalogue above the r-th of S sample keys is Beta(r, S - r + 1) ~ Gamma(r) / S, so the catalogue holds fewer than K' keys
    // above tau0 - the only way a seed can hurt: the query is flagged CNT_SEEDED_SHORT -> uncertified -> exhaustive pass, never
    // wrong - with probability P(Gamma(r) < r / c) = P(Poisson(r / c) >= r): 4e-10 at (c, r) = (8, 16), 1.4e-9 at (4, 28),
    // 9e-8 at (2.25, 52).  (The normal approximation used earlier - (1 - 1/c) sqrt(r) sigmas - overstates that tail by orders of
    // magnitude and had the samples 1.6 - 2.7x larger than they need to be.)
    if (!p.dense && !(options & (HWER_KNN_NO_SEED | HWER_KNN_EXACT_THRESHOLDS)) && !diag_env("HWER_SCREEN_NOSEED")) {
        const int64_t n_tiles = cdiv64(n_items, 32);
        const bool small = use_small_path(B, n_items, options);
        // How many keys tau0 lets through, c K' per query (c_min; more when the sample is capped):
        //   4    by default;
        //   8    where the kept keys are cheap and the pre-pass is not: a handful of queries on the small-batch path (latency: one
        //        query on 1.25 M rows 0.161 -> 0.147 ms; from 32 queries on the parked keys dominate and 4 is faster), and large
        //        catalogues with item splits (c4: 32 short lists per query; pre-pass + threshold kernel 0.60 -> 0.30 ms);
        //   2.25 when a query has ONE list (c3): 4 K' keys overflow the list's two 480-key streams, and every in-loop compaction
        //        (~5 K cycles of one wave) stalls the three other waves of its workgroup at the tile barrier - measured 24 % of
        //        the launch; 2.25 K' stay below the trigger for all but ~2e-4 of the queries (the pre-pass is 9 % of the catalogue;
        //        measured against 2.75: c3's main search 5.27 -> 5.03 ms).
        const bool single_list = p.n_lists == 1 && !small;
        // (round 3: with the candidate pick the pre-pass no longer pays per sample key, and c = 4 beats 8 on the large catalogues
        // too - same-box sweep, diagnostic build: c4 2.27 -> 2.20 ms per search, c5 4.34 -> 4.30, c4full 13.7 -> 13.5)
        double c_min = single_list ? 2.25 : ((B <= 2 && small) ? 8.0 : 4.0);   // (3 - 4 queries: 0.127 ms at c = 8, 0.112 - 0.115 at c = 4 on 1.25 M rows)
        if (const char* e = diag_env("HWER_SEED_C")) c_min = std::max(1.5, atof(e));   // tuning override (diagnostic builds)
        // sample rank r by c (probabilities above; the threshold kernels never take a rank below 16)
        const double r_min = single_list ? 52.0 : (c_min >= 8.0 ? 16.0 : 28.0);
        // sample size for that rank at c_min (+ 2 %), in 32-row tiles: 1 024 ... 262 144 items (the 10 M-row catalogue: 80 K rows
        // for c = 8, r = 16; the threshold kernel reads up to 65 536 keys of a row, at a regular stride beyond that)
        int64_t ns = std::min<int64_t>(SEED_MAX_TILES, std::max<int64_t>(32, cdiv64(int64_t(1.02 * r_min * double(n_items) / (c_min * double(p.kprime))), 32)));
        ns = std::min(ns, n_tiles);
        // the sample keys of all queries live in the workspace (float16): at most 2 GiB of them (131 072 queries: an 8 192-item sample)
        ns = std::min<int64_t>(ns, std::max<int64_t>(32, (int64_t(1) << 30) / (p.bq * 32)));   // (float16 keys: 2 GiB)
        const double S = double(ns) * 32.0;
        const double c = std::max(c_min, r_min * double(n_items) / (double(p.kprime) * S));
        const double frac = c * double(p.kprime) / double(n_items);
        // (not worth it: a catalogue of a few K' rows forced onto this path, or a sample capped so hard that tau0 would let
        // more than 64 K' keys per query through)
        if (frac <= 0.25 && c <= 64.0) {
            p.seed = 1;
            p.seed_tiles = int(ns);
            p.seed_stride = int(n_tiles / ns);
            p.seed_ld = ns * 32;
            p.seed_frac = frac;
            p.seed_splits = int(std::min<int64_t>(ns, std::max<int64_t>(1, 512 / p.n_qblocks)));
            p.scaled_first = 0;   // every in-loop compaction selects the exact K'-th key: lists are either exact or seeded
            // Round 3: the pre-pass does not store its keys - tau0 is the r-th largest of the sample, r is small (16 ... 52), and the
            // sample is split over `bins` = seed workgroups x item waves x 2 lane halves per query: every lane keeps the M largest keys
            // it sees (registers) and hands over only those; the r-th largest of the bins x M candidates IS the r-th largest sample
            // key unless some bin holds more than M of the r best - then it is slightly LOWER (the safe side: a few more keys pass
            // the filter).  With lambda = r / bins <= M / 4 keys of the top r expected per bin that is rare and costs ~1 / r more keys.
            // The threshold kernel then reads <= 4 096 candidates per query instead of the whole sample row (c4: 80 -> ~8 us and no
            // 82 MB key matrix; the one-wave pick replaces a 1 024-thread workgroup per query).
            {
                // (a handful of queries: hundreds of pre-pass workgroups per query block would hand the one-wave pick thousands of
                // candidates - 2 488 for one query on 1.25 M rows: 19 us of pick; at most 1 024 of them: 128 workgroups of ~3 tiles)
                if (int64_t(p.seed_splits) * wi * 2 > 1024) p.seed_splits = std::max(1, 1024 / (wi * 2));
                const int64_t per_t = cdiv64(ns, p.seed_splits);
                const int n_sp = int(cdiv64(ns, per_t));               // workgroups per query block the pre-pass really launches
                const int bins = n_sp * wi * 2;
                const int rank = std::max(16, int(frac * S + 0.5));
                // M: as many as keep the pick within 1 024 candidates (16 keys per lane of its wave), at most 4
                const int m = bins * 4 <= 1024 ? 4 : (bins * 2 <= 1024 ? 2 : (bins <= 4096 ? 1 : 0));
                if (m > 0 && int64_t(bins) * m >= int64_t(4) * rank && !diag_env("HWER_SEED_NO_TOPM")) {
                    p.seed_top_m = m;
                    p.seed_rank = rank;
                    p.seed_cand_ld = ((int64_t(bins) * m + 63) / 64) * 64;
                    p.seed_cand_n = bins * m;
                    // Round 5: the lanes' candidates are GROUP MAXIMA (the largest of a lane's 16 keys per sample tile) - one insertion per 16
                    // keys.  With 16 f << 1 a group above the threshold holds one key above it, so the r-th largest group maximum is the r-th
                    // largest key but for the rare groups with two: rank in groups r' = g(f) S / 16 <= r, g(f) = 1 - (1 - f)^16, raised until
                    // P(Binomial(S / 16, g(K' / n)) >= r') <= 1e-7 (the statement that the catalogue holds K' keys above tau0).
                    if (16.0 * frac <= 0.25 && !diag_env("HWER_SEED_NO_GROUPS")) {
                        const int64_t n_g = int64_t(S) / 16;
                        auto g_of = [](double f) { return 1.0 - pow(1.0 - f, 16.0); };
                        int r2 = std::max(8, int(g_of(frac) * double(n_g) + 0.5));
                        const double g0 = g_of(std::min(1.0, double(p.kprime) / double(n_items)));
                        const double mu = double(n_g) * g0;          // Poisson tail (n_g is large, g0 tiny): P(X >= r2), summed upwards from r2
                        auto tail = [&](int r) {
                            double term = exp(-mu + r * log(mu) - lgamma(r + 1.0)), sum = 0.0;
                            for (int j = r; j < r + 200 && term > 1e-30; ++j) { sum += term; term *= mu / double(j + 1); }
                            return sum;
                        };
                        while (r2 < 4096 && tail(r2) > 1e-7) ++r2;
                        if (int64_t(bins) * m >= int64_t(4) * r2) {
                            p.seed_group = 1;
                            p.seed_rank = r2;
                        }
                    }
                }
            }
            // group maxima (knn_internal.h): streaming plans that would otherwise store every sample key; 16 f <= 0.8 keeps the statistic
            // far from saturation (g <= 0.55), <= 4 096 groups keep the pick a one-wave job
            if (!p.seed_top_m && !small && 16.0 * frac <= 0.8 && ns * 2 <= 4096 && ns * 2 >= 32 && !diag_env("HWER_SEED_NO_GMAX")) {
                const int n_g = int(ns) * 2;
                auto g_of = [](double f) { return 1.0 - pow(1.0 - f, 16.0); };
                int rank = std::max(8, int(g_of(frac) * n_g + 0.5));
                // raise the rank until P(Binomial(n_g, g(K' / n)) >= rank) <= 1e-7 (log-space terms, summed from the tail end)
                const double g0 = g_of(std::min(1.0, double(p.kprime) / double(n_items)));
                auto tail = [&](int r) {
                    double sum = 0.0;
                    for (int j = n_g; j >= r; --j)
                        sum += exp(lgamma(n_g + 1.0) - lgamma(j + 1.0) - lgamma(n_g - j + 1.0) + j * log(g0) + (n_g - j) * log1p(-g0));
                    return sum;
                };
                while (rank < n_g && tail(rank) > 1e-7) ++rank;
                if (rank <= n_g / 2) {   // (a sample too small for that: keep the sample-key path)
                    p.seed_gmax = 1;
                    p.seed_groups = n_g;
                    p.seed_rank = rank;
                }
            }
        }
    }
    return p;
}

bool screen_f16_supported(const ScreenPlan& plan) { return plan.ng >= 8; }

template <int NG, int TQ, bool DENSE>
static int launch_screen_t2(const ScreenPlan& plan, const ScreenArgs& a, hipStream_t stream) {
    // tile ring 2 x 32 rows x (NG*32 B) | half norms 2 x 32 | key park 4 x 64 x 16 TQ
    size_t lds = size_t(2) * 32 * NG * 32 + size_t(2) * 32 * 4 + size_t(4) * 64 * 16 * TQ * 4 + ((DENSE || !STAGE_APPENDS) ? 0 : size_t(4) * 8 * TQ * 64 * 8);   // + append staging
    if (const char* e = diag_env("HWER_SCREEN_LDS_PAD_KB")) lds += size_t(atoi(e)) * 1024;   // occupancy experiments
    static uint64_t attr_done = 0;   // per device
    int dev;
    if (attr_pending(attr_done, &dev) || diag_env("HWER_SCREEN_LDS_PAD_KB")) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen_kernel<NG, TQ, DENSE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_screen: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done |= 1ull << dev;
    }
    const int grid = plan.n_qblocks * plan.n_splits;
    hipLaunchKernelGGL((screen_kernel<NG, TQ, DENSE>), dim3(grid), dim3(256), lds, stream, a);
    return check_launch("knn_screen");
}
template <int NG, int TQ>
static int launch_screen_t(const ScreenPlan& plan, const ScreenArgs& a, hipStream_t stream) {
    return plan.dense ? launch_screen_t2<NG, TQ, true>(plan, a, stream) : launch_screen_t2<NG, TQ, false>(plan, a, stream);
}
// float16 screen (streaming mode): tiles of 32 rows x NG 16-byte chunks
template <int NG, int TQ>
static int launch_screen16_t(const ScreenPlan& plan, const ScreenArgs& a, hipStream_t stream) {
    const size_t lds_dense = size_t(2) * 32 * NG * 16 + size_t(2) * 32 * 4 + size_t(4) * 64 * 16 * TQ * 4;
    constexpr int IBS = (NG <= 16 || WIDE64) ? SCREEN16_IB : 32;
    // 64-row tiles (+ append staging).  d = 256: no key park - that launch always filters with the max-tree variant (launch_screen sets
    // sparse_hits), and 2 x 32 KB of ring + 16 KB of park would leave one workgroup per CU
    constexpr bool NO_PARK = NG > 16 && WIDE64;
    size_t lds_stream = size_t(2) * IBS * NG * 16 + size_t(2) * IBS * 4 + (NO_PARK ? 0 : size_t(4) * 64 * 16 * TQ * 4) +
                        (STAGE_APPENDS ? size_t(4) * 8 * TQ * 64 * 8 : 0);
    if (const char* e = diag_env("HWER_SCREEN_LDS_PAD_KB")) lds_stream += size_t(atoi(e)) * 1024;   // occupancy experiments (diagnostic builds)
    static uint64_t attr_done = 0, attr_done_d = 0;   // per device
    int dev;
    const int grid = plan.n_qblocks * plan.n_splits;
    if (plan.dense) {
        if (attr_pending(attr_done_d, &dev)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen_kernel<NG, TQ, true, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_dense));
            if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_screen16(dense): hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_done_d |= 1ull << dev;
        }
        hipLaunchKernelGGL((screen_kernel<NG, TQ, true, true>), dim3(grid), dim3(256), lds_dense, stream, a);
        return check_launch("knn_screen16(dense)");
    }
    if (attr_pending(attr_done, &dev) || diag_env("HWER_SCREEN_LDS_PAD_KB")) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen_kernel<NG, TQ, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_stream));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_screen16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done |= 1ull << dev;
    }
    hipLaunchKernelGGL((screen_kernel<NG, TQ, false, true>), dim3(grid), dim3(256), lds_stream, stream, a);
    return check_launch("knn_screen16");
}

template <int NG, int TQ>
static int launch_screen16_pp_t(const ScreenPlan& plan, const ScreenArgs& a, hipStream_t stream) {
    // ring 3 x 64 rows x (NG * 16 B) | half norms 3 x 64 | key park 8 x 64 x 16 TQ | 4 tokens | the waves' scaled half norms
    const size_t lds = size_t(3) * 64 * NG * 16 + size_t(3) * 64 * 4 + size_t(8) * 64 * 16 * TQ * 4 + 16 + size_t(8) * TQ * 64 * 4;   // ... | 8 x TQ x 64 scaled half norms
    static uint64_t attr_done = 0;   // per device
    int dev;
    if (attr_pending(attr_done, &dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen16_pp_kernel<NG, TQ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&screen16_pp_kernel<NG, TQ, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(HWER_ELAUNCH, "knn_screen16_pp: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done |= 1ull << dev;
    }
    const int grid = (plan.n_qblocks / 2) * plan.n_splits;
    // pass-mask launches fold the threshold into the accumulators; max-tree launches (sparse_hits) have nothing to gain from it
#ifndef HWER_PP_FOLD_TQ1
#define HWER_PP_FOLD_TQ1 0      // one query subtile per wave (d = 256): 8 v_pk_add_f32 saved against 8 v_pk_mul_f32 -> v_pk_fma_f32: measured 3 - 6 % slower
#endif
    if (a.sparse_hits || (TQ == 1 && !HWER_PP_FOLD_TQ1)) hipLaunchKernelGGL((screen16_pp_kernel<NG, TQ, false>), dim3(grid), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((screen16_pp_kernel<NG, TQ, true>), dim3(grid), dim3(512), lds, stream, a);
    return check_launch("knn_screen16_pp");
}

int launch_seed_screen(const ScreenPlan& plan, const float* X, const float* hnorm, int64_t n_items, int64_t ldx, const float* Qf, int64_t B,
                       float* seed_keys, WsHeader* hdr, hipStream_t stream, const ScreenF16* f16) {
    (void)n_items;
    // the DENSE mode of the screen kernel over the sample: same query blocks / fragments as the main pass, the sample's tiles
    // split over seed_splits workgroups per query block, every key stored to seed_keys[q][sample-local item]
    ScreenPlan sp = plan;
    sp.dense = 1;
    sp.ld_dense = plan.seed_ld;
    const int64_t per = cdiv64(plan.seed_tiles, plan.seed_splits) * 32;
    sp.split_items = per;
    sp.n_splits = int(cdiv64(plan.seed_ld, per));
    sp.n_lists = 0;
    // seed_top_m: the lanes keep their M best keys and store only those, (bq, seed_cand_ld) float32; else every key as float16
    // ... or (round 5) the group maxima: (seed_groups, bq) float32, seed_m = -1
    ScreenArgs a{X, hnorm, plan.seed_ld, ldx, Qf, B, seed_keys, plan.seed_gmax ? plan.bq : (plan.seed_top_m ? plan.seed_cand_ld : plan.seed_ld), nullptr, nullptr, plan.n_qblocks, plan.wq, per, plan.bq,
                 plan.kprime, 0, 0, -INFINITY, nullptr, hdr, f16 ? f16->X16 : nullptr, f16 ? f16->Qf16 : nullptr, f16 ? f16->qexp : nullptr,
                 f16 ? f16->x_exp : 0, nullptr, plan.seed_stride, (plan.seed_top_m || plan.seed_gmax) ? 0 : 1, 0, 0, plan.seed_gmax ? -1 : plan.seed_top_m,
                 plan.seed_gmax ? 0 : plan.seed_group};
    const int key = plan.ng * 10 + plan.tq;
    if (f16) {
        switch (key) {
            case 81: return launch_screen16_t<8, 1>(sp, a, stream);
            case 82: return launch_screen16_t<8, 2>(sp, a, stream);
            case 161: return launch_screen16_t<16, 1>(sp, a, stream);
            case 162: return launch_screen16_t<16, 2>(sp, a, stream);
            case 321: return launch_screen16_t<32, 1>(sp, a, stream);
            default: return fail(HWER_EINVAL, "knn_seed16: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
        }
    }
    switch (key) {
        case 41: return launch_screen_t<4, 1>(sp, a, stream);
        case 42: return launch_screen_t<4, 2>(sp, a, stream);
        case 81: return launch_screen_t<8, 1>(sp, a, stream);
        case 82: return launch_screen_t<8, 2>(sp, a, stream);
        case 161: return launch_screen_t<16, 1>(sp, a, stream);
        case 162: return launch_screen_t<16, 2>(sp, a, stream);
        case 321: return launch_screen_t<32, 1>(sp, a, stream);
        default: return fail(HWER_EINVAL, "knn_seed: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
    }
}

int launch_screen(const ScreenPlan& plan, const float* X, const float* hnorm, int64_t n_items, int64_t ldx,
                  const float* Q32, int64_t B, uint64_t* cand, int32_t* cand_cnt, float* dense, WsHeader* hdr, hipStream_t stream,
                  const ScreenF16* f16, const float* tau0) {
    static unsigned long long* dbg = nullptr;
    static int dbg_grid = 0;
#ifndef HWER_DIAG_STAMPS
    if (diag_env("HWER_SCREEN_STAMPS")) fprintf(stderr, "[stamps] this library was built without -DHWER_DIAG_STAMPS (HIPCC_EXTRA=-DHWER_DIAG_STAMPS python csrc/build.py --force)\n");
#endif
    if (diag_env("HWER_SCREEN_STAMPS")) {
        if (dbg) {   // print the previous launch's stamps
            std::vector<unsigned long long> h(size_t(dbg_grid) * 48);
            (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
            if (const char* raw = diag_env("HWER_SCREEN_STAMPS_RAW")) {   // (hw id, phase marks) per wave behind the phase sums: tools/phase_marks.py
                if (FILE* f = fopen(raw, "wb")) { int g = dbg_grid; fwrite(&g, 4, 1, f); fwrite(h.data(), 8, h.size(), f); fclose(f); }
            }
            double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}; double tiles = 0; int n = 0;
            for (int w = 0; w < dbg_grid * 4; ++w) { if (h[w * 8 + 5] == 0) continue; for (int c = 0; c < 8; ++c) sum[c] += double(h[w * 8 + c]); tiles += double(h[w * 8 + 5]); ++n; }
            fprintf(stderr, "[stamps] waves=%d tiles/wave=%.0f prologue=%.1f us; cycles/tile: mfma=%.0f epilogue=%.0f wait+barrier=%.0f post=%.0f; loop %.0f cycles in %.1f us -> in-kernel clock %.2f GHz\n", n, tiles / n,
                    sum[0] / n / 100.0, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[6] / n, sum[7] / n / 100.0, sum[7] > 0 ? sum[6] / sum[7] * 0.1 : 0.0);
            (void)hipFree(dbg);
        }
        dbg_grid = plan.n_qblocks * plan.n_splits;
        if (hipMalloc(&dbg, size_t(dbg_grid) * 48 * 8) != hipSuccess || hipMemset(dbg, 0, size_t(dbg_grid) * 48 * 8) != hipSuccess) dbg = nullptr;
    }
    ScreenArgs a{X, hnorm, n_items, ldx, Q32, B, dense, plan.ld_dense, cand, cand_cnt, plan.n_qblocks, plan.wq, plan.split_items, plan.bq,
                 plan.kprime, (plan.scaled_first && !diag_env("HWER_SCREEN_EXACT")) ? 1 : 0, diag_env("HWER_SCREEN_NOFILTER") ? atoi(diag_env("HWER_SCREEN_NOFILTER")) : 0,
                 diag_env("HWER_SCREEN_TAU0") ? float(atof(diag_env("HWER_SCREEN_TAU0"))) : -INFINITY, diag_env("HWER_SCREEN_STAMPS") ? dbg : nullptr, hdr,
                 f16 ? f16->X16 : nullptr, f16 ? f16->Qf16 : nullptr, f16 ? f16->qexp : nullptr, f16 ? f16->x_exp : 0, tau0, 1, 0,
                 // expected passing keys per 32-row tile and wave (32 tq queries) under the seeded thresholds
                 // (ping-pong kernel: the park-free variant up to ~4 expected hits was 2 % faster in the diagnostic build at c4 - 1.7 hits per subtile and
                 //  wave - and +-0 in the production build: the 0.6 rule stays for both kernels)
                 (tau0 && plan.seed && plan.seed_frac * 32.0 * 32.0 * plan.tq < ((f16 && plan.pp) ? HWER_PP_SPARSE_MAX : 0.6) && !diag_env("HWER_SCREEN_DENSE_HITS")) || diag_env("HWER_SCREEN_SPARSE_HITS") ||
                 (f16 && WIDE64 && plan.ng > 16 && !plan.dense && !(plan.pp && HWER_PP_WIDE_PARK)) ? 1 : 0,
                 // many_hits: >= 8 expected passing keys per subtile and wave (single-list plans; unseeded launches): four pops per round
                 diag_env("HWER_SCREEN_MANY_HITS") ? atoi(diag_env("HWER_SCREEN_MANY_HITS")) : ((!tau0 || !plan.seed || plan.seed_frac * 32.0 * 32.0 * plan.tq >= 8.0) ? 1 : 0), 0};
    const int key = plan.ng * 10 + plan.tq;
    if (f16 && plan.pp && !plan.dense && (plan.n_qblocks & 1) == 0) {
        switch (key) {
            case 81: return launch_screen16_pp_t<8, 1>(plan, a, stream);
            case 82: return launch_screen16_pp_t<8, 2>(plan, a, stream);
            case 161: return launch_screen16_pp_t<16, 1>(plan, a, stream);
            case 162: return launch_screen16_pp_t<16, 2>(plan, a, stream);
            case 321: return launch_screen16_pp_t<32, 1>(plan, a, stream);
            default: return fail(HWER_EINVAL, "knn_screen16_pp: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
        }
    }
    if (f16) {
        switch (key) {
            case 81: return launch_screen16_t<8, 1>(plan, a, stream);
            case 82: return launch_screen16_t<8, 2>(plan, a, stream);
            case 161: return launch_screen16_t<16, 1>(plan, a, stream);
            case 162: return launch_screen16_t<16, 2>(plan, a, stream);
            case 321: return launch_screen16_t<32, 1>(plan, a, stream);
            default: return fail(HWER_EINVAL, "knn_screen16: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
        }
    }
    switch (key) {
        case 41: return launch_screen_t<4, 1>(plan, a, stream);
        case 42: return launch_screen_t<4, 2>(plan, a, stream);
        case 81: return launch_screen_t<8, 1>(plan, a, stream);
        case 82: return launch_screen_t<8, 2>(plan, a, stream);
        case 161: return launch_screen_t<16, 1>(plan, a, stream);
        case 162: return launch_screen_t<16, 2>(plan, a, stream);
        case 321: return launch_screen_t<32, 1>(plan, a, stream);
        default: return fail(HWER_EINVAL, "knn_screen: unsupported padded dim %d / tq %d", plan.ng * 8, plan.tq);
    }
}

}  // namespace hwer
