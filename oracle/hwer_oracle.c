/* ORACLE (C part) — CPU restatement of the arithmetic of the hwer retrieval hot path.  TEST INFRASTRUCTURE ONLY:
 * linked / called only by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
 * Parity status: PINNED — tests/test_oracle_golden.py checks it against the numpy oracle and, through it, against the
 * golden vectors captured from the unmodified reference (tools/capture_goldens.py).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fopenmp (oracle/Makefile).  No -ffast-math: the summation orders
 * below are the reference's and must not be re-associated.  Citations are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* scikit-learn euclidean_rdist64 (sklearn/metrics/_dist_metrics.pyx.tp), call site hwer/recommendation_base.py:74,79:
 * float64, sequential `d += tmp * tmp`. */
static double rdist_seq(const double* q, const float* x, int d) {
    double acc = 0.0;
    for (int j = 0; j < d; ++j) {
        double tmp = q[j] - (double)x[j];
        acc += tmp * tmp;
    }
    return acc;
}

/* numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src), the order behind `(a*b).sum(1)` at
 * hwer/recommendation_base.py:142 */
static double pairwise_dot_f64(const double* a, const float* x, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i] * (double)x[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k] * (double)x[k];
        int lim = n - (n % 8), i;
        for (i = 8; i < lim; i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k] * (double)x[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i] * (double)x[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_dot_f64(a, x, n2) + pairwise_dot_f64(a + n2, x + n2, n - n2);
}

typedef struct {
    double d2;
    int64_t idx;
} cand_t;

static int cand_less(const cand_t* a, const cand_t* b) { return a->d2 < b->d2 || (a->d2 == b->d2 && a->idx < b->idx); }

/* max-heap on (d2, idx) holding the k best so far */
static void sift_down(cand_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && cand_less(&h[m], &h[l])) m = l;
        if (r < n && cand_less(&h[m], &h[r])) m = r;
        if (m == i) return;
        cand_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}
static int cmp_cand(const void* a, const void* b) {
    const cand_t *x = a, *y = b;
    return cand_less(x, y) ? -1 : (cand_less(y, x) ? 1 : 0);
}

/* Exact k nearest rows of X (N,d) float32 (rows upcast to float64) for B float64 queries: the reference's
 * MultiKNN.query (hwer/recommendation_base.py:78-83) as brute force; ties -> ascending row id.
 * Then the re-score of find_closest_neighbours (:172-173): score = (a . x + 1)/2 in float64 pairwise order against the
 * anchor rows A (B,d) float64, stable sort by score descending.  out_*: (B,k).  threads <= 0 -> all cores.
 * One query at a time, one row at a time: the plain restatement; orc_find_closest_f64 below computes the same values faster. */
int orc_find_closest_f64_scalar(const float* X, int64_t N, int d, const double* Q, const double* A, int64_t B, int k,
                         int sort_by_score, int threads, int64_t* out_idx, double* out_score, double* out_dist2) {
    if (k > N || k < 1) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
        double* sc = (double*)malloc(sizeof(double) * (size_t)k);
        int* ord = (int*)malloc(sizeof(int) * (size_t)k);
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < B; ++b) {
            const double* q = Q + b * d;
            int n = 0;
            for (int64_t i = 0; i < N; ++i) {
                cand_t c = {rdist_seq(q, X + i * d, d), i};
                if (n < k) {
                    heap[n++] = c;
                    if (n == k)
                        for (int j = k / 2 - 1; j >= 0; --j) sift_down(heap, k, j);
                } else if (cand_less(&c, &heap[0])) {
                    heap[0] = c;
                    sift_down(heap, k, 0);
                }
            }
            qsort(heap, (size_t)k, sizeof(cand_t), cmp_cand);
            for (int j = 0; j < k; ++j) {
                sc[j] = A ? (pairwise_dot_f64(A + b * d, X + heap[j].idx * d, d) + 1.0) / 2.0 : (-1.0 * sqrt(heap[j].d2) + 2.0) / 2.0;
                ord[j] = j;
            }
            if (sort_by_score) { /* stable insertion sort by score desc */
                for (int j = 1; j < k; ++j) {
                    int o = ord[j], p = j - 1;
                    while (p >= 0 && sc[ord[p]] < sc[o]) {
                        ord[p + 1] = ord[p];
                        --p;
                    }
                    ord[p + 1] = o;
                }
            }
            for (int j = 0; j < k; ++j) {
                out_idx[b * k + j] = heap[ord[j]].idx;
                out_score[b * k + j] = sc[ord[j]];
                out_dist2[b * k + j] = heap[ord[j]].d2;
            }
        }
        free(heap);
        free(sc);
        free(ord);
    }
    return 0;
}


/* The same function, laid out for speed; every (query, row) pair still runs rdist_seq's arithmetic - acc = 0, then for j = 0..d-1 in
 * order: tmp = q[j] - (double)x[j]; acc += tmp * tmp, products and sums rounded separately - so every distance is bit-identical to the
 * scalar version (tests/test_oracle_golden.py holds the two against each other and against the goldens).  What changes is only WHICH
 * pairs are in flight together: RB = 32 rows are transposed to a [j][row] float64 tile once per (row block, query group) and their 32
 * independent chains advance side by side in SIMD lanes (gcc vector extensions: lane-wise IEEE operations); a group of QG queries
 * shares every tile, so the table streams from memory once per group instead of once per query.  Heaps, tie rule ((d2, row) ascending),
 * re-score and the stable sort by score are the scalar version's. */
#define ORC_RB 32
#define ORC_QG 32
typedef double orc_v4d __attribute__((vector_size(32)));

static void emit_query(const float* X, int d, const double* A, int64_t b, int k, int sort_by_score, cand_t* heap, int n, double* sc, int* ord,
                       int64_t* out_idx, double* out_score, double* out_dist2) {
    (void)n;
    qsort(heap, (size_t)k, sizeof(cand_t), cmp_cand);
    for (int j = 0; j < k; ++j) {
        sc[j] = A ? (pairwise_dot_f64(A + b * d, X + heap[j].idx * d, d) + 1.0) / 2.0 : (-1.0 * sqrt(heap[j].d2) + 2.0) / 2.0;
        ord[j] = j;
    }
    if (sort_by_score) { /* stable insertion sort by score desc */
        for (int j = 1; j < k; ++j) {
            int o = ord[j], p = j - 1;
            while (p >= 0 && sc[ord[p]] < sc[o]) {
                ord[p + 1] = ord[p];
                --p;
            }
            ord[p + 1] = o;
        }
    }
    for (int j = 0; j < k; ++j) {
        out_idx[b * k + j] = heap[ord[j]].idx;
        out_score[b * k + j] = sc[ord[j]];
        out_dist2[b * k + j] = heap[ord[j]].d2;
    }
}

int orc_find_closest_f64(const float* X, int64_t N, int d, const double* Q, const double* A, int64_t B, int k,
                         int sort_by_score, int threads, int64_t* out_idx, double* out_score, double* out_dist2) {
    if (k > N || k < 1) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const int64_t n_groups = (B + ORC_QG - 1) / ORC_QG;
#pragma omp parallel
    {
        cand_t* heaps = (cand_t*)malloc(sizeof(cand_t) * (size_t)k * ORC_QG);
        int* fill = (int*)malloc(sizeof(int) * ORC_QG);
        double* sc = (double*)malloc(sizeof(double) * (size_t)k);
        int* ord = (int*)malloc(sizeof(int) * (size_t)k);
        double* xt = (double*)aligned_alloc(64, sizeof(double) * (size_t)d * ORC_RB); /* [j][row] tile */
#pragma omp for schedule(dynamic, 1)
        for (int64_t g = 0; g < n_groups; ++g) {
            const int64_t b0 = g * ORC_QG;
            const int nq = (int)((B - b0) < ORC_QG ? (B - b0) : ORC_QG);
            for (int q = 0; q < nq; ++q) fill[q] = 0;
            for (int64_t i0 = 0; i0 < N; i0 += ORC_RB) {
                const int nr = (int)((N - i0) < ORC_RB ? (N - i0) : ORC_RB);
                for (int r = 0; r < nr; ++r) {
                    const float* x = X + (i0 + r) * d;
                    for (int j = 0; j < d; ++j) xt[(size_t)j * ORC_RB + r] = (double)x[j];
                }
                for (int r = nr; r < ORC_RB; ++r)
                    for (int j = 0; j < d; ++j) xt[(size_t)j * ORC_RB + r] = 0.0; /* lanes past the table: computed, never read */
                for (int q = 0; q < nq; ++q) {
                    const double* qv = Q + (b0 + q) * d;
                    orc_v4d acc[ORC_RB / 4];
                    for (int v = 0; v < ORC_RB / 4; ++v) acc[v] = (orc_v4d){0.0, 0.0, 0.0, 0.0};
                    for (int j = 0; j < d; ++j) {
                        const orc_v4d qj = {qv[j], qv[j], qv[j], qv[j]};
                        const orc_v4d* row = (const orc_v4d*)(xt + (size_t)j * ORC_RB);
                        for (int v = 0; v < ORC_RB / 4; ++v) {
                            const orc_v4d tmp = qj - row[v];
                            acc[v] += tmp * tmp; /* -ffp-contract=off: product and sum rounded separately, as rdist_seq */
                        }
                    }
                    cand_t* heap = heaps + (size_t)q * k;
                    int n = fill[q];
                    for (int r = 0; r < nr; ++r) {
                        cand_t c = {acc[r >> 2][r & 3], i0 + r};
                        if (n < k) {
                            heap[n++] = c;
                            if (n == k)
                                for (int j = k / 2 - 1; j >= 0; --j) sift_down(heap, k, j);
                        } else if (cand_less(&c, &heap[0])) {
                            heap[0] = c;
                            sift_down(heap, k, 0);
                        }
                    }
                    fill[q] = n;
                }
            }
            for (int q = 0; q < nq; ++q)
                emit_query(X, d, A, b0 + q, k, sort_by_score, heaps + (size_t)q * k, fill[q], sc, ord, out_idx, out_score, out_dist2);
        }
        free(heaps);
        free(fill);
        free(sc);
        free(ord);
        free(xt);
    }
    return 0;
}

/* numpy pairwise summation of an explicit array (same recursion as pairwise_dot_f64) */
static double pairwise_sum_f64(const double* p, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += p[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = p[k];
        int lim = n - (n % 8), i;
        for (i = 8; i < lim; i += 8)
            for (int k = 0; k < 8; ++k) r[k] += p[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += p[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum_f64(p, n2) + pairwise_sum_f64(p + n2, n - n2);
}

static double clip_unknown(double v) { return v < 1e-6 ? 1e-6 : (v > 1e-5 ? 1e-5 : v); }

/* RecommendationBase.predict on resolved indices (hwer/recommendation_base.py:141-150), float64 flavour:
 * idx == -1 -> row 0 clipped to [1e-6, 1e-5]; out[p] = ((E_src * E_dst).sum() + 1) / 2. */
int orc_pair_score_f64(const float* V, int64_t N, int d, const int32_t* src, const int32_t* dst, int64_t P, int threads,
                       double* out) {
    (void)N;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        double* prod = (double*)malloc(sizeof(double) * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t p = 0; p < P; ++p) {
            const int us = src[p] < 0, ut = dst[p] < 0;
            const float* rs = V + (int64_t)(us ? 0 : src[p]) * d;
            const float* rt = V + (int64_t)(ut ? 0 : dst[p]) * d;
            for (int j = 0; j < d; ++j) {
                double a = (double)rs[j], b = (double)rt[j];
                if (us) a = clip_unknown(a);
                if (ut) b = clip_unknown(b);
                prod[j] = a * b;
            }
            out[p] = (pairwise_sum_f64(prod, d) + 1.0) / 2.0;
        }
        free(prod);
    }
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
