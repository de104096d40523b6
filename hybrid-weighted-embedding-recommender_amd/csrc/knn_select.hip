// Stage 2 of the exact k-NN, host side + the kernels that are not templates: list reduction, multi-GPU merge (X1), all-distances pass;
// the select / re-rank / exhaustive-pass templates live in knn_select_kernels.h (instantiated here for float32 rows).
#include "knn_select_kernels.h"

namespace hwer {

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
                if (!(t[12] && t[13] && t[14] && t[15]) || t[6] <= t[0] || (t[15] >> 62)) continue;
                if (t[7] && t[8] && t[9]) { rr[0] += double(t[7] - t[3]); rr[1] += double(t[8] - t[7]); rr[2] += double(t[9] - t[8]); rr[3] += double(t[12] - t[9]); }
                r[0] += double(t[12] - t[3]); r[1] += double(t[13] - t[12]); r[2] += double(t[14] - t[13]); r[3] += double(t[15] - t[14]); r[4] += double(t[4] - t[15]);
                ++nr;
            }
            if (nr) fprintf(stderr, "[select stamps]   inside rerank-rounds (%lld queries): wave 0's rounds=%.0f wait for the other waves=%.0f boundary rank=%.0f counting order=%.0f rest=%.0f\n",
                            (long long)nr, r[0] / nr, r[1] / nr, r[2] / nr, r[3] / nr, r[4] / nr);
            if (nr && rr[1] > 0) fprintf(stderr, "[select stamps]   inside wave 0's rounds: set-up + first gathers issued=%.0f until the first chunk is staged=%.0f rest of the chunk loop=%.0f score + table writes=%.0f\n",
                                         rr[0] / nr, rr[1] / nr, rr[2] / nr, rr[3] / nr);
        }
        {   // wave-per-query forms: the chunk loop of the re-rank rounds by part (accumulated per query)
            double ck[5] = {0, 0, 0, 0, 0};
            int64_t nq = 0;
            for (int64_t q = 0; q < buf_B; ++q) {
                const unsigned long long* t = &h[size_t(q) * 16];
                if (!(t[15] >> 62)) continue;
                ck[0] += double(t[12]); ck[1] += double(t[13]); ck[2] += double(t[14]); ck[3] += double(t[15] & 0xFF); ck[4] += double((t[15] & ~(3ull << 62)) >> 8);
                ++nq;
            }
            if (nq) fprintf(stderr, "[select stamps]   chunk loop of %lld queries (%.1f chunks each), cycles per query: wait for the chunk's pieces=%.0f LDS reads=%.0f widen + chains (scalar operand loads included)=%.0f round set-up=%.0f\n",
                            (long long)nq, ck[3] / nq, ck[0] / nq, ck[1] / nq, ck[2] / nq, ck[4] / nq);
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
                             size_t workspace_bytes, uint32_t options, const void* X16, int x16_exp, void* stream, const F64Rows* f64 = nullptr) {
    if (B == 0) return HWER_OK;
    const bool do_screen = (stages & 1) != 0, do_select = (stages & 2) != 0;
    if (f64 && do_select) {   // genuine float64 rows: values from X64 / A64 (knn_select_f64.hip), the screen on their float32 rounding X
        if (!f64->X64 || f64->ld64 % 8 != 0 || f64->ld64 < d || f64->ld64 > (int64_t(1) << 24) || (reinterpret_cast<uintptr_t>(f64->X64) & 63))
            return fail(HWER_EINVAL, "knn_search_f64rows: X64 must be 64-byte aligned with a row stride that is a multiple of 8, >= d (zero padded) and <= 2^24");
        if (score_mode != 0 && score_mode != 2) return fail(HWER_EINVAL, "knn_search_f64rows: score_mode must be 0 (float64 scores) or 2 (from the distance)");
        if (score_mode == 0 && (!f64->A64 || f64->lda64 < d)) return fail(HWER_EINVAL, "knn_search_f64rows: A64 (row stride >= d) required for score_mode 0");
    }
    auto with_f64 = [&](EmitArgs e) {
        if (f64) { e.X64 = f64->X64; e.ld64 = f64->ld64; e.x64_rows = f64->x64_rows; e.A64 = f64->A64; e.lda64 = f64->lda64; }
        return e;
    };
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
    if (do_select && score_mode != 2 && !A32 && !f64) return fail(HWER_EINVAL, "knn_search: A32 required for score_mode 0/1");
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
        da.emit = with_f64(EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, 0, row_ids, id_base, k,
                                    out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)});
        da.list = list;
        da.list_count = cnt;
        da.list_cap = SMALL_LIST_CAP;
        da.list_min = sp.kprime;
        // float16 screen: two float16 roundings per product (2^-11 each) on top of the float32 accumulation error
        da.key_rel_err = X16 ? 1.02f * 0.0009765625f : 0.f;
        if (f64) return launch_select_list_f64(da, d, score_mode, da.B <= coop_max_queries(), st);
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
        da.emit = with_f64(EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, diag_env("HWER_SELECT_SKIP") ? atoi(diag_env("HWER_SELECT_SKIP")) : 0, row_ids, id_base, k,
                                    out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)});
        const bool coop = B <= coop_max_queries();
        if (f64) return launch_select_dense_f64(da, d, score_mode, coop, st);
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
    sa.emit = with_f64(EmitArgs{X, uint32_t(n_items - 1), ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, diag_env("HWER_SELECT_SKIP") ? atoi(diag_env("HWER_SELECT_SKIP")) : 0, row_ids, id_base, k,
                                out_idx, out_score, out_dist2, out_flags, hdr, sel_stamps_buffer(B)});
    if (f64) return launch_select_rerank_f64(sa, d, score_mode, st);
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

int hwer_knn_search_f64rows(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32, const double* Q64,
                            const double* X64, int64_t ld64, const int64_t* x64_rows, const double* A64, int64_t lda64, const int32_t* a_idx,
                            int64_t B, int k, int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max,
                            int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                            uint32_t options, const void* X16, int x16_exp, void* stream) {
    const F64Rows f{X64, ld64, x64_rows, A64, lda64};
    return knn_search_stages(3, X, hnorm, n_items, d, ldx, Q32, Q64, nullptr, 0, a_idx, B, k, score_mode, sort_by_score, row_ids, id_base, hnorm_max,
                             out_idx, out_score, out_dist2, out_flags, workspace, workspace_bytes, options, X16, x16_exp, stream, &f);
}

int hwer_knn_select_f64rows(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const double* X64, int64_t ld64,
                            const int64_t* x64_rows, const double* A64, int64_t lda64, const int32_t* a_idx, int64_t B, int k, int score_mode,
                            int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx, double* out_score,
                            double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes, uint32_t options, const void* X16,
                            void* stream) {
    const F64Rows f{X64, ld64, x64_rows, A64, lda64};
    return knn_search_stages(2, X, nullptr, n_items, d, ldx, nullptr, Q64, nullptr, 0, a_idx, B, k, score_mode, sort_by_score, row_ids, id_base,
                             hnorm_max, out_idx, out_score, out_dist2, out_flags, workspace, workspace_bytes, options, X16, 0, stream, &f);
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

static int exact_device_impl(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                             int64_t lda, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                             const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                             double* out_dist2, int32_t* out_flags, void* knn_workspace, int32_t* missed_accum,
                             void* workspace, size_t workspace_bytes, void* stream, const F64Rows* f64) {
    if ((!X && !f64) || !Q64 || !out_idx || !out_score || !out_flags || !knn_workspace || !workspace)
        return fail(HWER_EINVAL, "knn_exact_device: null pointer");
    if (f64) {
        if (!f64->X64 || f64->ld64 % 8 != 0 || f64->ld64 < d || (reinterpret_cast<uintptr_t>(f64->X64) & 63))
            return fail(HWER_EINVAL, "knn_exact_device_f64rows: X64 must be 64-byte aligned with a row stride that is a multiple of 8 and >= d");
        if (score_mode != 0 && score_mode != 2) return fail(HWER_EINVAL, "knn_exact_device_f64rows: score_mode must be 0 or 2");
        if (score_mode == 0 && (!f64->A64 || f64->lda64 < d)) return fail(HWER_EINVAL, "knn_exact_device_f64rows: A64 required for score_mode 0");
        if (d < 1 || d > 256) return fail(HWER_EINVAL, "knn_exact_device_f64rows: need 1 <= d <= 256");
    } else
    if (d < 1 || d > 256 || ldx < ((d + 15) / 16) * 16)
        return fail(HWER_EINVAL, "knn_exact_device: need 1 <= d <= 256 and ldx >= roundup16(d)");
    if (k < 1 || k > K_MAX) return fail(HWER_EINVAL, "knn_exact_device: k=%d outside 1..%d", k, K_MAX);
    if (k > n_items) return fail(HWER_EINVAL, "knn_exact_device: k must be less than or equal to the number of training points");
    if (n_items >= (int64_t(1) << 32)) return fail(HWER_EINVAL, "knn_exact_device: n_items must be < 2^32");
    if (score_mode < 0 || score_mode > 2 || (score_mode != 2 && !A32 && !f64)) return fail(HWER_EINVAL, "knn_exact_device: score_mode / A32");
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
    if (f64) { ea.emit.X64 = f64->X64; ea.emit.ld64 = f64->ld64; ea.emit.x64_rows = f64->x64_rows; ea.emit.A64 = f64->A64; ea.emit.lda64 = f64->lda64; }
    // sort area | max(re-rank area of the emitting wave, four waves' staging areas of stage A)
    const size_t lds = size_t(EX_CHUNK) * sizeof(Ent) + std::max(wave_lds_bytes(d, ((k + 63) / 64) * 64), size_t(4) * 64 * RR_RS * 4);
    int rc;
    if (f64) return launch_exact_device_f64(ea, lds, as_stream(stream));
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&exact_device_kernel<false>), lds, "knn_exact_device"))) return rc;
    hipLaunchKernelGGL(exact_device_kernel<false>, dim3(unsigned(ea.n_chunks), EXACT_DEVICE_MAXQ), dim3(256), lds, as_stream(stream), ea);
    return check_launch("knn_exact_device");
}

int hwer_knn_exact_device(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                          int64_t lda, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                          const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                          double* out_dist2, int32_t* out_flags, void* knn_workspace, int32_t* missed_accum,
                          void* workspace, size_t workspace_bytes, void* stream) {
    return exact_device_impl(X, n_items, d, ldx, Q64, A32, lda, a_idx, k, score_mode, sort_by_score, row_ids, id_base, out_idx, out_score, out_dist2,
                             out_flags, knn_workspace, missed_accum, workspace, workspace_bytes, stream, nullptr);
}

int hwer_knn_exact_device_f64rows(const double* X64, int64_t ld64, const int64_t* x64_rows, int64_t n_items, int d, const double* Q64,
                                  const double* A64, int64_t lda64, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                                  const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score, double* out_dist2,
                                  int32_t* out_flags, void* knn_workspace, int32_t* missed_accum, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    const F64Rows f{X64, ld64, x64_rows, A64, lda64};
    return exact_device_impl(nullptr, n_items, d, 0, Q64, nullptr, 0, a_idx, k, score_mode, sort_by_score, row_ids, id_base, out_idx, out_score, out_dist2,
                             out_flags, knn_workspace, missed_accum, workspace, workspace_bytes, stream, &f);
}

size_t hwer_knn_exact_workspace_bytes(int64_t n_items, int k) {
    const size_t dev = hwer_knn_exact_device_workspace_bytes(n_items, k);
    return dev ? WS_HEADER_BYTES + dev : 0;   // its own list header + the device pass's buffers
}

// Host-driven exhaustive pass: the listed queries go through the same kernel as hwer_knn_exact_device, 8 per launch.
static int exact_host_impl(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                           int64_t lda, const int32_t* a_idx, const int32_t* qsel_host, int64_t nq, int k, int score_mode, int sort_by_score,
                           const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                           double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                           void* stream, const F64Rows* f64) {
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
        if ((rc = exact_device_impl(X, n_items, d, ldx, Q64, A32, lda, a_idx, k, score_mode, sort_by_score, row_ids, id_base, out_idx,
                                    out_score, out_dist2, out_flags, ws, nullptr, ws + WS_HEADER_BYTES, workspace_bytes - WS_HEADER_BYTES,
                                    stream, f64)))
            return rc;
    }
    return HWER_OK;
}

int hwer_knn_exact_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                       int64_t lda, const int32_t* a_idx, const int32_t* qsel_host, int64_t nq, int k, int score_mode, int sort_by_score,
                       const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                       double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                       void* stream) {
    return exact_host_impl(X, n_items, d, ldx, Q64, A32, lda, a_idx, qsel_host, nq, k, score_mode, sort_by_score, row_ids, id_base, out_idx, out_score,
                           out_dist2, out_flags, workspace, workspace_bytes, stream, nullptr);
}

int hwer_knn_exact_f64rows(const double* X64, int64_t ld64, const int64_t* x64_rows, int64_t n_items, int d, const double* Q64, const double* A64,
                           int64_t lda64, const int32_t* a_idx, const int32_t* qsel_host, int64_t nq, int k, int score_mode, int sort_by_score,
                           const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags,
                           void* workspace, size_t workspace_bytes, void* stream) {
    const F64Rows f{X64, ld64, x64_rows, A64, lda64};
    return exact_host_impl(nullptr, n_items, d, 0, Q64, nullptr, 0, a_idx, qsel_host, nq, k, score_mode, sort_by_score, row_ids, id_base, out_idx, out_score,
                           out_dist2, out_flags, workspace, workspace_bytes, stream, &f);
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
