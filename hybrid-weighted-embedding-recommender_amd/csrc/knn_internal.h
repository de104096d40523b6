// Internal layout shared by the k-NN screen (knn_screen.hip) and select / re-rank (knn_select.hip) stages.
#pragma once
#include "hwer_common.h"

namespace hwer {

// Candidate streams.  Every (list, query) owns two append-only streams (one per MFMA lane half) of CAPH packed
// keys each, living in the caller's workspace:  cand[((list * Bq + q) * 2 + half) * CAPH + pos].
constexpr int CAPH = 512;            // entries per stream
constexpr int KPRIME_MAX = 448;      // a compaction retains K' entries, possibly all in ONE stream, and must leave
                                     // that stream below the trigger (CAPH - 32, checked every second tile): rounded to 64
constexpr int SLACK = 56;            // K' = roundup64(k + SLACK)
constexpr int COMPACT_SLACK = 32;    // a compaction keeps between K' and K' + 32 keys (bisection stops early)
constexpr int KEEP_MAX = KPRIME_MAX + COMPACT_SLACK;   // most keys one list hands to the select stage (480)
constexpr int K_MAX = KPRIME_MAX - SLACK;
// Streaming screen: stage a lane's appended keys in LDS and write them as whole 64-byte lines (true), or append every key with
// its own 8-byte store (false).  Measured at c3 (profiles/r2_*): staging cuts the bytes written 4x but its flush sits in the pop
// loop of the filter epilogue - the step got SLOWER; kept as a compile-time switch.
#ifndef HWER_STAGE_APPENDS
#define HWER_STAGE_APPENDS 0
#endif
constexpr bool STAGE_APPENDS = HWER_STAGE_APPENDS != 0;
constexpr int SCREEN_IB = 32;        // item rows per LDS tile (one 32-row MFMA subtile)
#ifndef HWER_SCREEN16_IB
#define HWER_SCREEN16_IB 64
#endif
constexpr int SCREEN16_IB = HWER_SCREEN16_IB;   // ... of the float16 streaming screen: two subtiles per tile barrier / LDS-DMA phase (32: round 2's tiles)
constexpr int ROW_PAD = 64;          // tables are allocated and padded (zero rows, half norm +inf) to whole ROW_PAD-row tiles
constexpr int DENSE_MAX_ITEMS = 8192; // catalogues up to this size take the dense path
constexpr int MAX_LISTS_DIRECT = 8;  // lists the select kernel can take in one pass (8 * 1024 keys = 64 KB LDS)

struct ScreenPlan {
    int ng;        // k-groups of 8 columns = ldx / 8
    int tq;        // query subtiles (of 32) per wave
    int wq;        // waves along the query dimension (1, 2, 4); wi = 4 / wq waves split the item subtiles
    int qb;        // queries per workgroup = wq * 32 * tq
    int n_qblocks;
    int n_splits;  // item ranges
    int64_t split_items;  // items per split (multiple of SCREEN_IB)
    int n_lists;   // n_splits * wi
    int64_t bq;    // B rounded up to qb (stream addressing)
    int pp;        // float16 streaming screen as 512-thread ping-pong workgroups (round 4): n_qblocks is even, a workgroup takes two blocks
    int kprime;
    int scaled_first;  // use the sample-scaled first threshold (verified; UNDERFULL lists are flagged)
    int dense;         // small catalogue: write every key to a (B, ld_dense) float matrix, select from it
    int64_t ld_dense;
    // Seeded thresholds (streaming mode): a pre-pass computes the keys of `seed_tiles` sample tiles (every seed_stride-th
    // 32-row tile of the catalogue) for every query into a dense (bq, seed_ld) matrix, the seed_rank-th largest sample key of
    // a query becomes its initial threshold tau0 - about seed_c K' of the catalogue's keys clear it - and every list of the
    // query starts filtering with it instead of collecting its first ~1 000 keys unconditionally.
    int seed;
    int seed_tiles, seed_stride, seed_splits;
    int64_t seed_ld;       // = seed_tiles * 32
    double seed_frac;      // expected fraction of the catalogue's keys above tau0 (= seed_c * K' / n_items)
    int seed_top_m;        // > 0: the pre-pass keeps M candidates per (workgroup, wave, lane half) instead of every sample key
    int seed_rank;         // ... and tau0 = the seed_rank-th largest candidate (= sample rank seed_frac * sample size)
    int64_t seed_cand_ld;  // ... row stride of the candidate matrix (bins * M rounded up to 64; the tail is never read)
    int seed_cand_n;       // ... candidates per query = bins * M
    // Round 5, plans whose pre-pass cannot hand over lane candidates (one workgroup per query block: c3's 131 072 queries over 26 744 rows
    // gave every query a 2 464-key float16 sample row, 646 MB written and read back, and a 1 700-instruction rank search per query): the
    // pre-pass keeps, per (sample tile, lane half), only the MAXIMUM of the lane's 16 keys - seed_groups = 2 seed_tiles values per query,
    // stored transposed ((seed_groups, bq) float32: a wave's 32 queries are one 128-byte segment) - and tau0 = the seed_rank-th largest
    // GROUP MAXIMUM.  A group clears a threshold that a fraction f of the keys clear with probability g(f) = 1 - (1 - f)^16, so the rank
    // r' = g(seed_frac) seed_groups aims at the same threshold as the sample rank seed_frac S did; with 16 f well below 1 nearly every
    // clearing group holds ONE clearing key and the estimate loses little (c3: relative spread 14.8 % against 13.7 % from the whole sample).
    // Correlated neighbours (a catalogue sorted by cluster) put several good keys into one group: the threshold then comes out LOWER - more
    // keys pass, never fewer.  The safety statement is the same as for the sample rank: the catalogue holds fewer than K' keys above tau0
    // with probability P(Binomial(seed_groups, g(K' / n_items)) >= r') - r' is raised until that is below 1e-7 - and such a query is
    // flagged and recomputed exhaustively, never wrong.
    int seed_gmax;         // 1: group maxima instead of sample keys / lane candidates
    int seed_groups;       // ... values per query
    int seed_group;        // seed_top_m plans: the lanes' top-M candidates are group maxima too (seed_rank is then a rank in groups)
};

ScreenPlan make_screen_plan(int64_t B, int64_t n_items, int ldx, int k, uint32_t options = 0);
int kprime_for(int k, int64_t n_items, int ldx, bool f16_screen);   // candidate keys kept per query (knn_screen.hip)

// Header at the start of the k-NN workspace: the queries the last search could not certify (flag 0), appended by the
// select kernels, consumed by the on-device exhaustive pass (hwer_knn_exact_device) without a host round trip.
constexpr int WS_HEADER_BYTES = 4096;
constexpr int UNCERT_LIST_MAX = 256;    // ids kept in the header (the count keeps counting past it)
constexpr int EXACT_DEVICE_MAXQ = 8;    // queries one hwer_knn_exact_device call recomputes
constexpr int EXACT_DEVICE_GROUPS = 64; // first-level merge groups of the exhaustive pass (per query)
struct WsHeader {
    int32_t uncert_count;               // } zeroed by the screen kernel of every search (hdr_reset), with all tickets
    int32_t pad0;
    int32_t tickets[EXACT_DEVICE_MAXQ];                            // second-level arrival counters of the exhaustive pass
    int32_t group_tickets[EXACT_DEVICE_MAXQ][EXACT_DEVICE_GROUPS]; // first-level arrival counters
    int32_t uncert_list[UNCERT_LIST_MAX];
    // unit-norm audit of hwer_blend_prepare_search_f32 (rowops.hip): block partials are added here, the last block to arrive moves
    // the totals to the caller's outputs and leaves these fields zero again.  NOT touched by hdr_reset: zero once after the
    // workspace is allocated (hwer_knn_workspace_init).
    unsigned long long audit_acc[3];
    double audit_dev;
    unsigned int audit_ticket;
    unsigned int init_magic;            // WS_INIT_MAGIC once hwer_knn_workspace_init has run on this workspace
};
constexpr unsigned int WS_INIT_MAGIC = 0x48574552u;   // "HWER"
constexpr int HDR_RESET_INTS = 2 + EXACT_DEVICE_MAXQ + EXACT_DEVICE_MAXQ * EXACT_DEVICE_GROUPS;
// called by block 0 of the first kernel of every search: empty uncertified list, idle tickets
__device__ __forceinline__ void hdr_reset(WsHeader* hdr, int tid, int n_threads) {
    for (int i = tid; i < HDR_RESET_INTS; i += n_threads) reinterpret_cast<int32_t*>(hdr)[i] = 0;
}
static_assert(sizeof(WsHeader) <= WS_HEADER_BYTES, "header must fit its slot");
constexpr int CNT_UNDERFULL = 1 << 30;     // flag bit in cand_cnt: the list's (sample-scaled) threshold dropped keys it should not have
constexpr int CNT_SEEDED_SHORT = 1 << 29;  // flag bit: the list kept only the keys above the query's SEEDED threshold and they are fewer than K';
                                           // fine as long as the UNION of the query's lists holds >= K' keys (all of them > tau0 >= anything dropped)
constexpr int CNT_FLAGS = CNT_UNDERFULL | CNT_SEEDED_SHORT;

// float16 screen inputs (HWER_KNN_F16_SCREEN): the float16 table, the float16 query fragments and their per-query exponents
struct ScreenF16 {
    const void* X16;
    const void* Qf16;
    const int32_t* qexp;
    int x_exp;
};
bool screen_f16_supported(const ScreenPlan& plan);   // padded dim >= 64
int launch_screen(const ScreenPlan& plan, const float* X, const float* hnorm, int64_t n_items, int64_t ldx,
                  const float* Q32, int64_t B, uint64_t* cand, int32_t* cand_cnt, float* dense, WsHeader* hdr, hipStream_t stream,
                  const ScreenF16* f16 = nullptr, const float* tau0 = nullptr);
// seed pre-pass: keys of the plan's sample tiles for every query -> seed_keys (bq, plan.seed_ld); same query fragments as the main pass
int launch_seed_screen(const ScreenPlan& plan, const float* X, const float* hnorm, int64_t n_items, int64_t ldx, const float* Qf, int64_t B,
                       float* seed_keys, WsHeader* hdr, hipStream_t stream, const ScreenF16* f16 = nullptr);
// per query: tau0[b] = the (want_frac * sample size)-th largest key of its row of `keys` (B, ld), n valid keys per row (knn_small.hip)
int launch_seed_threshold(const float* keys, int64_t ld, int64_t n, int64_t B, double want_frac, float* tau0, int32_t* count_scratch,
                          hipStream_t stream);
// the same from the plan: dispatches to the candidate pick (seed_top_m) or to the sample-row kernels above
int launch_seed_tau0(const ScreenPlan& plan, const float* seed_keys, int64_t B, float* tau0, int32_t* count_scratch, hipStream_t stream);
size_t seed_keys_bytes(const ScreenPlan& plan);   // workspace bytes of the pre-pass output (bq rows)
// queries -> float16 MFMA fragments + per-query scale exponents (knn_small.hip); Qf16: bq * ng * 16 bytes, qexp: bq ints
int repack_queries16(const float* Q32, int64_t B, int64_t bq, int ldx, void* Qf16, int32_t* qexp, hipStream_t stream);

// LDS-DMA (global -> LDS without a register destination) issued through inline asm: hipcc counts a builtin
// global_load_lds as a pending LDS write and drains vmcnt(0) before the NEXT ds_read, which would serialise the
// prefetch with the tile that is being computed.  An asm statement is invisible to that bookkeeping; the kernel waits
// for it explicitly (s_waitcnt vmcnt(0) right before the tile barrier).  M0 carries the wave-uniform LDS byte
// address and is saved / restored inside the same statement (it is compiler-reserved).
// scalar 64-bit base (wave-uniform) + 32-bit per-lane byte offset: keeps the per-lane address state to ONE VGPR
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// the same with a 64-bit per-lane source address (row gathers from tables beyond 4 GB)
__device__ __forceinline__ void lds_dma16_v(const void* vptr, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vptr), "s"(lds_byte_addr) : "memory");
}
// four of them in one statement: pieces p0 .. p3 land at lds_byte_addr + {0, 1, 2, 3} * STRIDE (one save / restore of M0 for the four)
template <int STRIDE>
__device__ __forceinline__ void lds_dma16_v4(const void* p0, const void* p1, const void* p2, const void* p3, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                 "s_add_u32 m0, m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(lds_byte_addr), "i"(STRIDE) : "memory", "scc");
}
__device__ __forceinline__ void lds_dma4(const void* sbase, uint32_t voff, uint32_t lds_byte_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return uint32_t(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)(p)));
}

// LDS image of an item tile: IB rows of CH = 2*NG 16-byte chunks, UNPADDED (the LDS-DMA writes 1 KiB per
// wave-instruction linearly), XOR-swizzled so that the ds_read_b128 of the MFMA A fragments (16 lanes = 16
// different rows, same logical chunk) hits 16 different 16-byte slots of the 256-byte bank row:
//   physical chunk = logical chunk ^ swz(row)
template <int CH>
__device__ __forceinline__ int swz(int row) {
    if constexpr (CH >= 16) return row & 15;
    else return (row >> 1) & (CH - 1);   // CH == 8: two 128-byte rows share a bank row
}

// Queries in MFMA-fragment order (built once per search by repack_queries): Qf[(s * NG + g) * 64 + lane] (float4) =
// Q[32 s + (lane & 31)][8 g + 4 (lane >> 5) .. + 3], zero for queries >= B.  A wave then loads the B fragment of query
// subtile s and k-group g with ONE fully coalesced 1-KiB instruction; reading the row-major Q32 directly costs 32
// quarter-line requests per instruction, repeated by every workgroup that shares the query block.
size_t repacked_queries_bytes(int64_t B, int ldx);
int repack_queries(const float* Q32, int64_t B, int64_t B_padded, int ldx, float* Qf, hipStream_t stream);   // B_padded: multiple of 32, rows >= B are zero

// small-batch path (knn_small.hip): B <= SMALL_MAX_QUERIES queries over a catalogue too large for the dense select
constexpr int SMALL_MAX_QUERIES = 1024;   // always below this many queries (see use_small_path); key matrix (B x items x 4 B) <= SMALL_MAX_KEY_BYTES
constexpr int64_t SMALL_MAX_KEY_BYTES = int64_t(8) << 30;
constexpr int SMALL_LIST_CAP = 4096;      // candidate keys one query may collect between its sampled threshold and the select
constexpr int COUNT_STRIDE = 32;          // ints between two queries' list counters (128 B): every wave appends with returning atomics, and with the
                                          // counters of 32 queries in ONE cache line those atomics serialised - +45 us at 32 queries, +85 us at 64 (round 3)
constexpr int SMALL_SAMPLE = 16384;       // keys sampled per query for the threshold
struct SmallPlan {
    int ng, tq, n_qblocks;   // query blocks of 32 * tq
    int64_t bq;              // B rounded up to whole query blocks (rows of the key matrix)
    int64_t ld;              // key matrix row stride = item rows padded to 32
    int kprime;
};
SmallPlan make_small_plan(int64_t B, int64_t n_items, int ldx, int k);
bool use_small_path(int64_t B, int64_t n_items, uint32_t options = 0);
size_t small_ws_bytes(const SmallPlan& p, const ScreenPlan* seed = nullptr);   // without the header; seed: see launch_small
// keys -> per-query candidate lists: list_out (B, SMALL_LIST_CAP) packed keys, cnt_out (B) appended counts
// seed (nullable): a screen plan with seed = 1 for the same search -> seeded mode: sample pre-pass, per-query tau0, ONE filtering pass
// over the catalogue (no key matrix); NULL: every key is stored, thresholds from a sample of the key rows, collect pass
int launch_small(const SmallPlan& p, const float* X, const float* hnorm, int64_t n_items, int64_t ldx, const float* Q32, int64_t B,
                 unsigned char* ws_after_header, uint64_t** list_out, int32_t** cnt_out, WsHeader* hdr, const void* X16 /*nullable*/,
                 int x_exp, hipStream_t stream, const ScreenPlan* seed = nullptr);
// where launch_small leaves its candidate lists (for the select stage of a split search)
void small_outputs(const SmallPlan& p, const ScreenPlan* seed, unsigned char* ws_after_header, uint64_t** list_out, int32_t** cnt_out);
// float16 storage variant of the small-batch screen (§8f-4): the table scaled by 2^x_exp and rounded to float16
int f16_scale_exp(float hnorm_max);
int pack_rows16(const float* X, int64_t n_rows_padded, int64_t ldx, int x_exp, void* X16, hipStream_t stream);

}  // namespace hwer
