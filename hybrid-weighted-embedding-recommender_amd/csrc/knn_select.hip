// Stage 2 of the exact k-NN (K5-K7): candidate selection, float64 re-rank in the reference's exact summation
// orders, exactness certificate, re-score + final sort; plus the exhaustive float64 fallback and the multi-GPU
// merge (X1).  One 256-thread workgroup per query.
//
// Orders reproduced bit for bit (verified against numpy 2.2.6 / scikit-learn 1.7.2, see oracle/hwer_oracle.py):
//   dist2 : scikit-learn `euclidean_rdist64`  -> sequential  d += (q_j - x_j)^2, separate mul and add roundings
//   score : numpy `(a * b).sum(1)`             -> pairwise summation of the rounded products
#include <algorithm>

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
                const int lo = ((t / stride) * (stride << 1)) + (t % stride);
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
                const int lo = ((t / stride) * (stride << 1)) + (t % stride);
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
// per-thread restatements of the reference's summation orders
// ---------------------------------------------------------------------------------------------------
// numpy pairwise sum of p_j = A_j * X_j, j in [0, n), evaluated by ONE thread (8 chains in registers).
template <typename T, typename FA, typename FX>
__device__ __forceinline__ T np_dot_leaf(FA a_at, FX x_at, int j0, int n) {
    if (n < 8) {
        T r = T(0);
        for (int i = 0; i < n; ++i) r = r + T(a_at(j0 + i)) * T(x_at(j0 + i));
        return r;
    }
    T r0, r1, r2, r3, r4, r5, r6, r7;
    {
        r0 = T(a_at(j0 + 0)) * T(x_at(j0 + 0)); r1 = T(a_at(j0 + 1)) * T(x_at(j0 + 1));
        r2 = T(a_at(j0 + 2)) * T(x_at(j0 + 2)); r3 = T(a_at(j0 + 3)) * T(x_at(j0 + 3));
        r4 = T(a_at(j0 + 4)) * T(x_at(j0 + 4)); r5 = T(a_at(j0 + 5)) * T(x_at(j0 + 5));
        r6 = T(a_at(j0 + 6)) * T(x_at(j0 + 6)); r7 = T(a_at(j0 + 7)) * T(x_at(j0 + 7));
    }
    const int lim = n - (n & 7);
    for (int i = 8; i < lim; i += 8) {
        // products are rounded separately from the adds (the reference materialises a*b first): no FMA
        const T p0 = T(a_at(j0 + i + 0)) * T(x_at(j0 + i + 0)), p1 = T(a_at(j0 + i + 1)) * T(x_at(j0 + i + 1));
        const T p2 = T(a_at(j0 + i + 2)) * T(x_at(j0 + i + 2)), p3 = T(a_at(j0 + i + 3)) * T(x_at(j0 + i + 3));
        const T p4 = T(a_at(j0 + i + 4)) * T(x_at(j0 + i + 4)), p5 = T(a_at(j0 + i + 5)) * T(x_at(j0 + i + 5));
        const T p6 = T(a_at(j0 + i + 6)) * T(x_at(j0 + i + 6)), p7 = T(a_at(j0 + i + 7)) * T(x_at(j0 + i + 7));
        r0 = r0 + p0; r1 = r1 + p1; r2 = r2 + p2; r3 = r3 + p3;
        r4 = r4 + p4; r5 = r5 + p5; r6 = r6 + p6; r7 = r7 + p7;
    }
    T res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (int i = lim; i < n; ++i) res = res + T(a_at(j0 + i)) * T(x_at(j0 + i));
    return res;
}
template <typename T, int DEPTH, typename FA, typename FX>
__device__ __forceinline__ T np_dot(FA a_at, FX x_at, int j0, int n) {
    if constexpr (DEPTH == 0) {
        return np_dot_leaf<T>(a_at, x_at, j0, n);
    } else {
        if (n <= 128) return np_dot_leaf<T>(a_at, x_at, j0, n);
        int n2 = n / 2;
        n2 -= n2 & 7;
        const T s0 = np_dot<T, DEPTH - 1>(a_at, x_at, j0, n2);
        const T s1 = np_dot<T, DEPTH - 1>(a_at, x_at, j0 + n2, n - n2);
        return s0 + s1;
    }
}

// ---------------------------------------------------------------------------------------------------
// shared tail: given `n_c` candidate items in LDS (c_item), compute exact dist2 / score, sort, certify, emit
// ---------------------------------------------------------------------------------------------------
struct EmitArgs {
    const float* X;
    int64_t ldx;
    int d;
    const double* Q64;
    const float* A32;
    int64_t lda;
    const int32_t* a_idx;  // nullable: anchor row of query b is A32[a_idx[b]] (else A32[b])
    int score_mode;
    int sort_by_score;
    const int64_t* row_ids;
    int64_t id_base;
    int k;
    int64_t* out_idx;
    double* out_score;
    double* out_dist2;
    int32_t* out_flags;
};

// LDS carve used by rerank_emit (bytes): q64[d] | a32 as double[d] | item[NC] | d2[NC] | sc[NC] | ent[NC2]
__host__ __device__ inline size_t emit_lds_bytes(int d, int nc2) {
    return size_t(d) * 8 * 2 + size_t(nc2) * (4 + 4 + 8 + 8 + 16);
}

// b: query row; c_item[0..n_c): local item ids (any order); cut_key: every item NOT in c_item has screen key <= cut_key
// (-inf when c_item is exhaustive); e32_scale: factor of the float32 error bound.  flag_value: what to store
// when certified (1 = screened + certified, 2 = exhaustive).
__device__ void rerank_emit(const EmitArgs& a, int64_t b, unsigned char* lds, const uint32_t* c_item_in, int n_c, int nc2,
                            bool exhaustive, float cut_key, float hnorm_max, int flag_value) {
    const int d = a.d;
    double* q64 = reinterpret_cast<double*>(lds);
    double* a64 = q64 + d;
    uint32_t* item = reinterpret_cast<uint32_t*>(a64 + d);
    uint32_t* pad_ = item + nc2;  // keeps the doubles 8-byte aligned
    double* d2 = reinterpret_cast<double*>(pad_ + nc2);
    double* sc = d2 + nc2;
    Ent* ent = reinterpret_cast<Ent*>(sc + nc2);

    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        q64[j] = a.Q64[b * d + j];
        a64[j] = a.A32 ? double(a.A32[(a.a_idx ? int64_t(a.a_idx[b]) : b) * a.lda + j]) : 0.0;
    }
    for (int i = threadIdx.x; i < n_c; i += blockDim.x) item[i] = c_item_in[i];
    __syncthreads();

    for (int i = threadIdx.x; i < n_c; i += blockDim.x) {
        const float* xr = a.X + int64_t(item[i]) * a.ldx;
        // scikit-learn euclidean_rdist64: sequential, (q - x) in float64, mul and add rounded separately
        double acc = 0.0;
        for (int j = 0; j < d; ++j) {
            const double tmp = __dsub_rn(q64[j], double(xr[j]));
            acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
        }
        d2[i] = acc;
        double s = 0.0;
        if (a.score_mode == 0) {
            const double dot = np_dot<double, 2>([&](int j) { return a64[j]; }, [&](int j) { return double(xr[j]); }, 0, d);
            s = (dot + 1.0) / 2.0;
        } else if (a.score_mode == 1) {
            const float dot = np_dot<float, 2>([&](int j) { return float(a64[j]); }, [&](int j) { return xr[j]; }, 0, d);
            s = double((dot + 1.0f) / 2.0f);
        } else {
            s = (-1.0 * sqrt(acc) + 2.0) / 2.0;  // hwer/gcn_ncf.py:381
        }
        sc[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc2; i += blockDim.x) {
        Ent e;
        if (i < n_c) {
            e.hi = f64_orderable(d2[i]);
            e.lo = (uint64_t(item[i]) << 32) | uint64_t(uint32_t(i));
        } else {
            e.hi = ~0ull;
            e.lo = ~0ull;
        }
        ent[i] = e;
    }
    bitonic_asc_ent(ent, nc2);  // (dist2 asc, item asc)

    const int k = a.k;  // k <= n_c guaranteed by the host
    // ---- certificate -------------------------------------------------------------------------------------
    int flag = flag_value;
    if (!exhaustive) {
        double Q2 = 0.0;
        for (int j = 0; j < d; ++j) Q2 += q64[j] * q64[j];
        const double Dk = d2[uint32_t(ent[k - 1].lo)];
        const double u = 5.9604644775390625e-08;  // 2^-24
        const double xn = sqrt(2.0 * double(hnorm_max)) * (1.0 + 1e-6);
        const double E32 = (2.0 * d + 16.0) * u * (sqrt(Q2) * xn + double(hnorm_max));
        const double bound = Q2 - 2.0 * double(cut_key) - 2.0 * E32 - 1e-12 * (1.0 + Q2);
        flag = (Dk < bound) ? flag_value : 0;
    }
    // ---- final order ---------------------------------------------------------------------------------------
    const int k2 = pow2ceil_i(k);
    __syncthreads();
    if (a.sort_by_score) {
        // stable sort by score desc == ascending (~orderable(score), rank in the dist order)
        auto make = [&](int i) {
            Ent e;
            if (i < k) {
                const uint32_t slot = uint32_t(ent[i].lo);
                e.hi = ~f64_orderable(sc[slot]);
                e.lo = (uint64_t(uint32_t(i)) << 32) | uint64_t(slot);
            } else {
                e.hi = ~0ull;
                e.lo = ~0ull;
            }
            return e;
        };
        const int i0 = threadIdx.x, i1 = threadIdx.x + 256;  // k2 <= 512, 256 threads
        Ent e0{}, e1{};
        if (i0 < k2) e0 = make(i0);
        if (i1 < k2) e1 = make(i1);
        __syncthreads();
        if (i0 < k2) ent[i0] = e0;
        if (i1 < k2) ent[i1] = e1;
        bitonic_asc_ent(ent, k2);
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const uint32_t slot = uint32_t(ent[i].lo);
        const uint32_t it = item[slot];
        a.out_idx[b * k + i] = a.row_ids ? a.row_ids[it] : a.id_base + int64_t(it);
        a.out_score[b * k + i] = sc[slot];
        a.out_dist2[b * k + i] = d2[slot];
    }
    if (threadIdx.x == 0) a.out_flags[b] = flag;
}

// ---------------------------------------------------------------------------------------------------
// select: gather the candidate streams of one query, keep the K' best screen keys, re-rank
// ---------------------------------------------------------------------------------------------------
struct SelectArgs {
    const uint64_t* cand;
    const int32_t* cand_cnt;
    int n_lists;
    int64_t bq;
    int kprime;
    int m2max;  // power of two >= n_lists * 2 * CAPH
    float hnorm_max;
    EmitArgs emit;
};

__global__ __launch_bounds__(256) void select_rerank_kernel(const SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_off = reinterpret_cast<int*>(smem_raw);  // [2 * MAX_LISTS_DIRECT + 1] in the first 128 bytes
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);
    const int64_t b = blockIdx.x;
    const int n_streams = a.n_lists * 2;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int s = 0; s < n_streams; ++s) {
            const int list = s >> 1, half = s & 1;
            s_off[s] = off;
            off += a.cand_cnt[(int64_t(list) * a.bq + b) * 2 + half];
        }
        s_off[n_streams] = off;
    }
    __syncthreads();
    const int M = s_off[n_streams];
    const int m2 = pow2ceil_i(max(M, 2));
    for (int s = 0; s < n_streams; ++s) {
        const int list = s >> 1, half = s & 1;
        const uint64_t* src = a.cand + ((int64_t(list) * a.bq + b) * 2 + half) * CAPH;
        const int n = s_off[s + 1] - s_off[s];
        for (int i = threadIdx.x; i < n; i += blockDim.x) keys[s_off[s] + i] = src[i];
    }
    for (int i = M + threadIdx.x; i < m2; i += blockDim.x) keys[i] = 0ull;
    bitonic_desc_u64(keys, m2);  // (screen key desc, item asc)

    const int n_c = min(M, a.kprime);
    const bool exhaustive = M < a.kprime;  // no stream ever compacted -> nothing was dropped
    const float cut = exhaustive ? -INFINITY : key_score(keys[n_c - 1]);
    // candidate ids move behind the key array's live prefix
    const int nc2 = pow2ceil_i(max(n_c, 2));
    unsigned char* tail = smem_raw + 128 + size_t(a.m2max) * 8;
    uint32_t* c_item = reinterpret_cast<uint32_t*>(tail);
    for (int i = threadIdx.x; i < n_c; i += blockDim.x) c_item[i] = key_item(keys[i]);
    __syncthreads();
    rerank_emit(a.emit, b, tail + size_t(KPRIME_MAX) * 4, c_item, n_c, nc2, exhaustive, cut, a.hnorm_max, 1);
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
};

__global__ __launch_bounds__(256) void reduce_lists_kernel(const ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_off = reinterpret_cast<int*>(smem_raw);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);
    const int64_t b = blockIdx.x;
    const int group = blockIdx.y;
    const int l0 = group * MAX_LISTS_DIRECT;
    const int nl = min(MAX_LISTS_DIRECT, a.n_lists_in - l0);
    const int n_streams = nl * 2;
    if (threadIdx.x == 0) {
        int off = 0;
        for (int s = 0; s < n_streams; ++s) {
            s_off[s] = off;
            off += a.cnt_in[(int64_t(l0 + (s >> 1)) * a.bq + b) * 2 + (s & 1)];
        }
        s_off[n_streams] = off;
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
        a.cnt_out[(int64_t(group) * a.bq + b) * 2 + 0] = min(n_c, CAPH);
        a.cnt_out[(int64_t(group) * a.bq + b) * 2 + 1] = max(n_c - CAPH, 0);
    }
}

// ---------------------------------------------------------------------------------------------------
// exhaustive float64 fallback: (1) dist2 of every item, (2) chunk-wise sort keeps k per chunk, repeat, (3) emit
// ---------------------------------------------------------------------------------------------------
constexpr int EX_CHUNK = 4096;

__global__ __launch_bounds__(256) void exact_dist2_kernel(const float* __restrict__ X, int64_t n_items, int64_t ldx,
                                                          int d, const double* __restrict__ q64, Ent* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* q = reinterpret_cast<double*>(smem_raw);
    for (int j = threadIdx.x; j < d; j += blockDim.x) q[j] = q64[j];
    __syncthreads();
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_items; i += int64_t(gridDim.x) * blockDim.x) {
        const float* xr = X + i * ldx;
        double acc = 0.0;
        for (int j = 0; j < d; ++j) {
            const double tmp = __dsub_rn(q[j], double(xr[j]));
            acc = __dadd_rn(acc, __dmul_rn(tmp, tmp));
        }
        Ent e;
        e.hi = f64_orderable(acc);
        e.lo = uint64_t(i);
        out[i] = e;
    }
}

// each block sorts one chunk of EX_CHUNK entries and writes its `keep` smallest to out[block * keep ..]
__global__ __launch_bounds__(256) void exact_chunk_select_kernel(const Ent* __restrict__ in, int64_t n, int keep,
                                                                 Ent* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Ent* ent = reinterpret_cast<Ent*>(smem_raw);
    const int64_t base = int64_t(blockIdx.x) * EX_CHUNK;
    for (int i = threadIdx.x; i < EX_CHUNK; i += blockDim.x) {
        Ent e;
        if (base + i < n) {
            e = in[base + i];
        } else {
            e.hi = ~0ull;
            e.lo = ~0ull;
        }
        ent[i] = e;
    }
    bitonic_asc_ent(ent, EX_CHUNK);
    for (int i = threadIdx.x; i < keep; i += blockDim.x) out[int64_t(blockIdx.x) * keep + i] = ent[i];
}

__global__ __launch_bounds__(256) void exact_emit_kernel(const Ent* __restrict__ sel, int n_sel, int64_t b,
                                                         const EmitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t* c_item = reinterpret_cast<uint32_t*>(smem_raw);
    int n_c = 0;
    for (int i = 0; i < n_sel; ++i) n_c += (sel[i].lo != ~0ull);  // padding entries sort last; count is tiny work
    n_c = min(n_c, a.k);
    for (int i = threadIdx.x; i < n_c; i += blockDim.x) c_item[i] = uint32_t(sel[i].lo);
    __syncthreads();
    const int nc2 = pow2ceil_i(max(n_c, 2));
    rerank_emit(a, b, smem_raw + size_t(K_MAX + 8) * 4, c_item, n_c, nc2, true, -INFINITY, 0.f, 2);
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
};

// entries: hi = orderable(dist2), lo = (position in the gathered array); ties broken by global id through a
// second compare on equal hi (ids are fetched from `idx`).
__global__ __launch_bounds__(256) void merge_topk_kernel(const MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Ent* ent = reinterpret_cast<Ent*>(smem_raw);
    const int64_t b = blockIdx.x;
    const int total = a.n_lists * a.k;
    const int n2 = pow2ceil_i(max(total, 2));
    // Two-level key: (dist2, global id).  Global ids are < 2^40 in practice; we need them in the key for the
    // tie rule, so sort by (hi = dist2, lo = id) and carry the source position in a parallel pass afterwards.
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        Ent e;
        if (i < total) {
            const int l = i / a.k, j = i % a.k;
            const int64_t src = (int64_t(l) * a.B + b) * a.k + j;
            e.hi = f64_orderable(a.dist2[src]);
            e.lo = (uint64_t(a.idx[src]) << 20) | uint64_t(uint32_t(i));  // id < 2^44, position < 2^20
        } else {
            e.hi = ~0ull;
            e.lo = ~0ull;
        }
        ent[i] = e;
    }
    bitonic_asc_ent(ent, n2);
    const int k = a.k;
    const int k2 = pow2ceil_i(k);
    auto src_of = [&](const Ent& e) {
        const int i = int(e.lo & 0xFFFFFu);
        return (int64_t(i / k) * a.B + b) * k + (i % k);
    };
    if (a.sort_by_score) {
        auto make = [&](int i) {
            Ent e;
            if (i < k) {
                e.hi = ~f64_orderable(a.score[src_of(ent[i])]);
                e.lo = (uint64_t(uint32_t(i)) << 32) | (ent[i].lo & 0xFFFFFu);
            } else {
                e.hi = ~0ull;
                e.lo = ~0ull;
            }
            return e;
        };
        const int i0 = threadIdx.x, i1 = threadIdx.x + 256;  // k2 <= 512, 256 threads
        Ent e0{}, e1{};
        if (i0 < k2) e0 = make(i0);
        if (i1 < k2) e1 = make(i1);
        __syncthreads();
        if (i0 < k2) ent[i0] = e0;
        if (i1 < k2) ent[i1] = e1;
        bitonic_asc_ent(ent, k2);
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const int64_t src = src_of(ent[i]);
        a.out_idx[b * k + i] = a.idx[src];
        a.out_score[b * k + i] = a.score[src];
        a.out_dist2[b * k + i] = a.dist2[src];
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
bool g_prof_on = false;
ProfSlot g_prof[2];  // 0: knn_screen, 1: knn_select (+ reduce)

struct ProfScope {
    int which;
    hipStream_t st;
    bool on;
    ProfScope(int w, hipStream_t s) : which(w), st(s), on(g_prof_on) {
        if (!on) return;
        ProfSlot& p = g_prof[which];
        if (p.used == p.start.size()) {
            hipEvent_t a, b;
            hipEventCreate(&a);
            hipEventCreate(&b);
            p.start.push_back(a);
            p.stop.push_back(b);
        }
        hipEventRecord(p.start[p.used], st);
    }
    ~ProfScope() {
        if (!on) return;
        ProfSlot& p = g_prof[which];
        hipEventRecord(p.stop[p.used], st);
        ++p.used;
    }
};
}  // namespace

extern "C" int hwer_profile_enable(int on) {
    g_prof_on = on != 0;
    return HWER_OK;
}
// Sum of the event-bracketed durations since the last read.  Synchronises on the recorded events (call it outside
// the timed region).
extern "C" int hwer_profile_read(int which, double* total_ms, int64_t* launches) {
    if (which < 0 || which > 1 || !total_ms || !launches) return fail(HWER_EINVAL, "profile_read: bad argument");
    ProfSlot& p = g_prof[which];
    double sum = 0.0;
    for (size_t i = 0; i < p.used; ++i) {
        hipEventSynchronize(p.stop[i]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, p.start[i], p.stop[i]);
        sum += ms;
    }
    *total_ms = sum;
    *launches = int64_t(p.used);
    p.used = 0;
    return HWER_OK;
}

// workspace layout: [cand A][cnt A][cand B][cnt B]   (B only when lists must be pre-reduced)
static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

struct WsLayout {
    size_t cand_a, cnt_a, cand_b, cnt_b, total;
};
static WsLayout ws_layout(const ScreenPlan& p) {
    WsLayout w{};
    size_t off = 0;
    w.cand_a = off;
    off += align256(size_t(p.n_lists) * p.bq * 2 * CAPH * 8);
    w.cnt_a = off;
    off += align256(size_t(p.n_lists) * p.bq * 2 * 4);
    int groups = p.n_lists > MAX_LISTS_DIRECT ? int(cdiv64(p.n_lists, MAX_LISTS_DIRECT)) : 0;
    w.cand_b = off;
    off += align256(size_t(groups) * p.bq * 2 * CAPH * 8);
    w.cnt_b = off;
    off += align256(size_t(groups) * p.bq * 2 * 4);
    w.total = off;
    return w;
}

extern "C" {

int hwer_knn_max_k(void) { return K_MAX; }

size_t hwer_knn_workspace_bytes(int64_t B, int64_t n_items, int d, int k) {
    const int ldx = hwer_knn_padded_dim(d);
    if (ldx < 0 || B < 1 || n_items < 1 || k < 1 || k > K_MAX) return 0;
    return ws_layout(make_screen_plan(B, n_items, ldx, k)).total;
}

int hwer_knn_search_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                        const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k, int score_mode,
                        int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                        double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (B == 0) return HWER_OK;
    if (!X || !hnorm || !Q32 || !Q64 || !out_idx || !out_score || !out_dist2 || !out_flags || !workspace)
        return fail(HWER_EINVAL, "knn_search: null pointer");
    const int dp = hwer_knn_padded_dim(d);
    if (dp < 0) return fail(HWER_EINVAL, "knn_search: d=%d unsupported (1..256)", d);
    if (ldx != dp) return fail(HWER_EINVAL, "knn_search: ldx must be hwer_knn_padded_dim(d)=%d (got %lld)", dp, (long long)ldx);
    if (k < 1 || k > K_MAX) return fail(HWER_EINVAL, "knn_search: k=%d outside 1..%d", k, K_MAX);
    if (k > n_items) return fail(HWER_EINVAL, "knn_search: k must be less than or equal to the number of training points");
    if (n_items >= (int64_t(1) << 32)) return fail(HWER_EINVAL, "knn_search: n_items must be < 2^32 per shard");
    if (score_mode < 0 || score_mode > 2) return fail(HWER_EINVAL, "knn_search: score_mode must be 0, 1 or 2");
    if (score_mode != 2 && !A32) return fail(HWER_EINVAL, "knn_search: A32 required for score_mode 0/1");
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(Q32) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(HWER_EINVAL, "knn_search: X/Q32 must be 16-byte and workspace 256-byte aligned");
    const ScreenPlan plan = make_screen_plan(B, n_items, int(ldx), k);
    const WsLayout w = ws_layout(plan);
    if (workspace_bytes < w.total) return fail(HWER_EWORKSPACE, "knn_search: workspace %zu < %zu", workspace_bytes, w.total);
    hipStream_t st = as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint64_t* cand = reinterpret_cast<uint64_t*>(ws + w.cand_a);
    int32_t* cnt = reinterpret_cast<int32_t*>(ws + w.cnt_a);
    int rc;
    {
        ProfScope ps(0, st);
        rc = launch_screen(plan, X, hnorm, n_items, ldx, Q32, B, cand, cnt, st);
    }
    if (rc) return rc;
    ProfScope ps_select(1, st);

    int n_lists = plan.n_lists;
    uint64_t* cand_other = reinterpret_cast<uint64_t*>(ws + w.cand_b);
    int32_t* cnt_other = reinterpret_cast<int32_t*>(ws + w.cnt_b);
    while (n_lists > MAX_LISTS_DIRECT) {
        const int groups = int(cdiv64(n_lists, MAX_LISTS_DIRECT));
        ReduceArgs ra{cand, cnt, n_lists, cand_other, cnt_other, plan.bq, plan.kprime};
        const size_t lds = 128 + size_t(MAX_LISTS_DIRECT) * 2 * CAPH * 8;
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
    sa.n_lists = n_lists;
    sa.bq = plan.bq;
    sa.kprime = plan.kprime;
    int m2 = 2;
    while (m2 < n_lists * 2 * CAPH) m2 <<= 1;
    sa.m2max = m2;
    sa.hnorm_max = hnorm_max;
    sa.emit = EmitArgs{X, ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, row_ids, id_base, k,
                       out_idx, out_score, out_dist2, out_flags};
    const size_t lds = 128 + size_t(m2) * 8 + size_t(KPRIME_MAX) * 4 + emit_lds_bytes(d, 512);
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&select_rerank_kernel), lds, "knn_select"))) return rc;
    hipLaunchKernelGGL(select_rerank_kernel, dim3(unsigned(B)), dim3(256), lds, st, sa);
    return check_launch("knn_select");
}

size_t hwer_knn_exact_workspace_bytes(int64_t n_items, int k) {
    if (n_items < 1 || k < 1 || k > K_MAX) return 0;
    // level 0: n_items entries; level 1: ceil(n/CHUNK) * keep entries (ping-pong with level 0)
    const size_t l0 = align256(size_t(n_items) * sizeof(Ent));
    const size_t l1 = align256(size_t(cdiv64(n_items, EX_CHUNK)) * size_t(k) * sizeof(Ent) + sizeof(Ent) * EX_CHUNK);
    return l0 + l1;
}

int hwer_knn_exact_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                       int64_t lda, const int32_t* a_idx, const int32_t* qsel_host, int64_t nq, int k, int score_mode, int sort_by_score,
                       const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                       double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (nq == 0) return HWER_OK;
    if (!X || !Q64 || !qsel_host || !out_idx || !out_score || !out_dist2 || !out_flags || !workspace)
        return fail(HWER_EINVAL, "knn_exact: null pointer");
    if (d < 1 || d > 1024 || ldx < d) return fail(HWER_EINVAL, "knn_exact: bad d / ldx");
    if (k < 1 || k > K_MAX) return fail(HWER_EINVAL, "knn_exact: k=%d outside 1..%d", k, K_MAX);
    if (k > n_items) return fail(HWER_EINVAL, "knn_exact: k must be less than or equal to the number of training points");
    if (n_items >= (int64_t(1) << 32)) return fail(HWER_EINVAL, "knn_exact: n_items must be < 2^32");
    if (score_mode < 0 || score_mode > 2 || (score_mode != 2 && !A32)) return fail(HWER_EINVAL, "knn_exact: score_mode / A32");
    const size_t need = hwer_knn_exact_workspace_bytes(n_items, k);
    if (workspace_bytes < need) return fail(HWER_EWORKSPACE, "knn_exact: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    Ent* bufA = reinterpret_cast<Ent*>(ws);
    Ent* bufB = reinterpret_cast<Ent*>(ws + align256(size_t(n_items) * sizeof(Ent)));
    int rc;
    const size_t lds_chunk = size_t(EX_CHUNK) * sizeof(Ent);
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&exact_chunk_select_kernel), lds_chunk, "knn_exact"))) return rc;
    for (int64_t qi = 0; qi < nq; ++qi) {
        const int64_t b = qsel_host[qi];
        const int grid = int(std::min<int64_t>(cdiv64(n_items, 256), 4096));
        hipLaunchKernelGGL(exact_dist2_kernel, dim3(grid), dim3(256), size_t(d) * 8, st, X, n_items, ldx, d,
                           Q64 + b * d, bufA);
        if ((rc = check_launch("knn_exact(dist2)"))) return rc;
        Ent* in = bufA;
        Ent* out = bufB;
        int64_t n = n_items;
        const int keep = k;
        do {
            const int64_t chunks = cdiv64(n, EX_CHUNK);
            hipLaunchKernelGGL(exact_chunk_select_kernel, dim3(unsigned(chunks)), dim3(256), lds_chunk, st, in, n, keep, out);
            if ((rc = check_launch("knn_exact(select)"))) return rc;
            n = chunks * keep;
            std::swap(in, out);
        } while (n > keep);
        EmitArgs ea{X, ldx, d, Q64, A32, lda, a_idx, score_mode, sort_by_score, row_ids, id_base, k,
                    out_idx, out_score, out_dist2, out_flags};
        const size_t lds = size_t(K_MAX + 8) * 4 + emit_lds_bytes(d, 512);
        hipLaunchKernelGGL(exact_emit_kernel, dim3(1), dim3(256), lds, st, in, int(n), b, ea);
        if ((rc = check_launch("knn_exact(emit)"))) return rc;
    }
    return HWER_OK;
}

int hwer_merge_topk(const int64_t* idx, const double* score, const double* dist2, int n_lists, int64_t B, int k,
                    int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2, void* stream) {
    if (B == 0) return HWER_OK;
    if (!idx || !score || !dist2 || !out_idx || !out_score || !out_dist2) return fail(HWER_EINVAL, "merge_topk: null pointer");
    if (n_lists < 1 || k < 1 || k > 512) return fail(HWER_EINVAL, "merge_topk: need n_lists >= 1 and 1 <= k <= 512");
    const int64_t total = int64_t(n_lists) * k;
    if (total > 8192) return fail(HWER_EINVAL, "merge_topk: n_lists * k = %lld > 8192", (long long)total);
    int n2 = 2;
    while (n2 < total) n2 <<= 1;
    const size_t lds = size_t(std::max(n2, 512)) * sizeof(Ent);
    int rc;
    if ((rc = set_max_lds(reinterpret_cast<const void*>(&merge_topk_kernel), lds, "merge_topk"))) return rc;
    MergeArgs ma{idx, score, dist2, n_lists, B, k, sort_by_score, out_idx, out_score, out_dist2};
    hipLaunchKernelGGL(merge_topk_kernel, dim3(unsigned(B)), dim3(256), lds, as_stream(stream), ma);
    return check_launch("merge_topk");
}

}  // extern "C"
