/*
 * hwer_hip.h — C ABI of the MI355X (gfx950) retrieval hot path of `hwer`.
 *
 * Drop-in boundary.  The reference is pure Python (numpy / scikit-learn) and has no FFI of
 * its own; these are the entry points a maintainer binds with `ctypes` (INTEGRATION.md shows the stub) to
 * replace the numpy / KDTree calls cited per function below (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) unless it says "host";
 *   - the library never allocates or frees device memory and never synchronises: callers pass output and
 *     workspace buffers and a `hipStream_t` (as void*; NULL = default stream);
 *   - tables are row-major float32; `ld*` are row strides in ELEMENTS;
 *   - return value: HWER_OK (0) or a negative HWER_E* code; `hwer_last_error()` gives the text (thread local).
 *   - index semantics: `-1` in a node-index array means "unknown node": row 0 clipped element-wise to
 *     [1e-6, 1e-5] (hwer/recommendation_base.py:147-150).
 */
#ifndef HWER_HIP_H
#define HWER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HWER_OK 0
#define HWER_EINVAL (-1)     /* bad argument (shape, alignment, unsupported d / k) */
#define HWER_ELAUNCH (-2)    /* HIP launch / runtime error */
#define HWER_EWORKSPACE (-3) /* workspace too small */

#define HWER_ABI_VERSION 1

/* library identity / diagnostics */
int hwer_abi_version(void);
const char* hwer_last_error(void);
/* Blocks the calling host thread until `stream` has drained (hipStreamSynchronize).  The ONLY entry point that synchronises: the host
 * mirror's per-anchor call (one cached hipGraph replay per find_closest_neighbours, hwer/recommendation_base.py:157-174) waits for its
 * results with it instead of going through a framework stream object. */
int hwer_stream_synchronize(void* stream);
/* Optional kernel timing: hwer_profile_enable(n), n >= 1, makes every n-th hwer_knn_search_f32 call bracket its screen
 * kernel (which = 0) and its select / re-rank kernels (which = 1) with HIP events recorded on the launch stream
 * (n = 0: off).  hwer_profile_read sums the bracketed durations, reports how many there were, and clears. */
int hwer_profile_enable(int on);
int hwer_profile_read(int which, double* total_ms, int64_t* launches);
/* Self-test of the wave-level primitives the select kernels are built from since round 4 (csrc/hwer_common.h: the DPP prefix scan, the
 * one-key-per-lane register sort, the lower bound in a sorted run, the DPP max / min / float64 sum): `n_waves` waves of seeded pseudo-random
 * lane values on the device, every result checked on the host against a plain loop.  Returns HWER_OK, or HWER_EINVAL with the first
 * mismatch in hwer_last_error().  Test infrastructure (the driver-run GPU suite calls it); synchronises the device. */
int hwer_selftest_wave_ops(int n_waves, unsigned seed);

/* ---------------------------------------------------------------------------------------------------
 * K1+K2  pair scoring — replaces RecommendationBase.predict, hwer/recommendation_base.py:135-144
 *        (same body hwer/gcn_ncf.py:330-334): out[p] = (sum_j E_src[p,j]*E_dst[p,j] + 1)/2.
 *   _f32: products and sums in float32 in numpy's pairwise order  -> bit-identical to the reference when
 *         its `vectors` are float32 (GcnNCF flavour).
 *   _f64: rows upcast to float64, same order -> bit-identical to the reference run on the float64 upcast.
 * ------------------------------------------------------------------------------------------------- */
int hwer_pair_score_f32(const float* V, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst,
                        int64_t P, float* out, void* stream);
int hwer_pair_score_f64(const float* V, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst,
                        int64_t P, double* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K8     neighbour mean — replaces the list-comprehension at hwer/content_recommender.py:83:
 *        out[n] = mean_{m in indices[indptr[n]:indptr[n+1]]} E[m]  (sequential adds in CSR order, then / deg).
 * ------------------------------------------------------------------------------------------------- */
int hwer_segment_mean_f32(const float* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N,
                          int d, float* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K9+K10 blend + L2-normalise — generalises hwer/content_recommender.py:85 and :96 (utils.py:43-44):
 *        out[r] = unit( sum_t alpha[r or 0][t] * tables[t][ row_idx[t] ? row_idx[t][r] : r ] ), no epsilon.
 *   tables  : device array of T device pointers, each (rows_t, d) with row stride ld_tables (same for all)
 *   row_idx : device array of T device pointers (entries may be NULL) or NULL
 *   alpha   : (T) if alpha_per_row == 0 else (R, T)
 *   out_norm: optional (R) pre-normalisation L2 norms.
 *   T = 2, tables = (E, A), alpha = (.5, .5) is exactly the reference's (E + A)/2 then unit_length.
 * ------------------------------------------------------------------------------------------------- */
int hwer_blend_normalize_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx,
                             const float* alpha, int alpha_per_row, int T, int64_t R, int d, float* out,
                             int64_t ldo, float* out_norm, void* stream);

/* Same, with the K11 unit-norm audit of the OUTPUT rows fused into the epilogue (one pass instead of three kernels):
 * audit_counts[0..2] = rows with norm > 1+eps / < 1-eps / NaN (4 x uint64 buffer), audit_absdev_sum = sum |norm - 1|. */
int hwer_blend_normalize_audit_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx,
                                   const float* alpha, int alpha_per_row, int T, int64_t R, int d, float* out,
                                   int64_t ldo, float eps, uint64_t* audit_counts, double* audit_absdev_sum, void* stream);

/* Fused content tail — hwer/content_recommender.py:77-86 + :96 in one pass:
 *        out[n] = unit( (E[n] + mean_{m in adj(n)} E[m]) / 2 ).  CSR as hwer_segment_mean_f32. */
int hwer_content_tail_f32(const float* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N,
                          int d, float* out, int64_t ldo, void* stream);

/* The same three stages in FLOAT64 - the arithmetic the reference itself runs them in (round 5).  ContentRecommendation keeps its table
 * in float64 from the PCA on: `all_embeddings[adj].mean(0)` (hwer/content_recommender.py:83), `(all_embeddings + adjacent_vectors)/2.0`
 * (:85) and `unit_length(embeddings, axis=1)` (:96 -> hwer/utils.py:43-44) are float64 numpy expressions.  On the same float64 table
 * these entry points return the reference's BITS (tests/golden g6 / g15: `final` array_equal), where the _f32 flavours are within 1e-6:
 *   neighbour mean : rows added sequentially in CSR (= adjacency append) order, one IEEE division by the count   (numpy mean(0))
 *   blend          : content tail (E + A) / 2.0;  general form acc = 0, acc = acc + alpha_t * T_t[row] for t = 0 .. T-1, product and sum
 *                    rounded separately (numpy evaluates `acc + a * rows` as two ufunc passes)
 *   unit rows      : x / sqrt(add.reduce(x * x)) with numpy's pairwise summation order, one IEEE division per element, no epsilon
 * Tables: row-major float64, row stride EVEN (ld >= roundup2(d)), 16-byte aligned, pad column (odd d) readable.  1 <= d <= 1024.
 * alpha is float64 here: (T) or (R, T).  out_norm (nullable): the pre-normalisation norms.  HBM-bound: 8 d (deg + 2) bytes per row
 * (tail), 8 d (T + 1) (blend). */
int hwer_segment_mean_f64(const double* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N, int d, double* out,
                          int64_t ldo, void* stream);
int hwer_content_tail_f64(const double* E, int64_t lde, const int64_t* indptr, const int32_t* indices, int64_t N, int d, double* out,
                          int64_t ldo, void* stream);
int hwer_blend_normalize_f64(const double* const* tables, int64_t ld_tables, const int32_t* const* row_idx, const double* alpha,
                             int alpha_per_row, int T, int64_t R, int d, double* out, int64_t ldo, double* out_norm, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * F1+F2  fit-side table construction — the data passes of
 *          IncrementalPCA(n_components, batch_size=2**16).fit_transform(table)   hwer/content_recommender.py:68-69
 *          unit_length(., axis=1) and the one-hot node-type columns               hwer/content_recommender.py:70-74
 *        over the block-diagonal feature table of hwer/content_recommender.py:50-66, which is never materialised: a row
 *        of node type t is non-zero only in block t's columns and in one-hot column t.  The host mirror (hwer/fit_pca.py)
 *        runs IncrementalPCA's recurrence on the per-(batch, type) Gram matrices (eigen-decompositions of W x W matrices).
 *
 *   hwer_segment_gram_f64: for every segment s (rows seg_ptr[s] .. seg_ptr[s+1] of the float64 block B, row stride ldb),
 *        G[s] = [B | 1]^T [B | 1], a dense (D+1) x (D+1) row-major matrix: G[i][j] = sum_r B[r][i] B[r][j],
 *        G[i][D] = G[D][i] = column sums, G[D][D] = row count.  float64 MFMA; partial sums over 8 192-row chunks are reduced
 *        in a fixed order (deterministic).  seg_ptr: device (n_seg + 1); max_seg_rows >= the longest segment.
 *   hwer_pca_project_unit_f64: for every row l of B: y = B[l] . P + bias (n components), u = y / ||y||_2 (no epsilon),
 *        out[out_rows ? out_rows[l] : l] = [u | one-hot(onehot_col) over n_onehot columns | zeros up to the row stride].
 *        P: (roundup4(D), ldp) float64 row-major, rows >= D and columns >= n zero; ldp >= hwer_pca_padded_components(n);
 *        bias: hwer_pca_padded_components(n) values, zero padded.  out32 (float32) required, out64 (float64) optional.
 * ------------------------------------------------------------------------------------------------- */
int hwer_gram_padded_dim(int D);
size_t hwer_segment_gram_workspace_bytes(int D, int n_seg, int64_t max_seg_rows);
int hwer_segment_gram_f64(const double* B, int64_t ldb, int D, const int64_t* seg_ptr, int n_seg, int64_t max_seg_rows,
                          double* G, void* workspace, size_t workspace_bytes, void* stream);
int hwer_pca_padded_components(int n); /* 16, 32, 64, 128 or 256; -1 if n is unsupported (n < 1 or n > 256) */
int hwer_pca_project_unit_f64(const double* B, int64_t ldb, int64_t n_rows, int D, const double* P, int64_t ldp,
                              const double* bias, int n, const int64_t* out_rows /*nullable*/, int onehot_col, int n_onehot,
                              float* out32, int64_t ldo32, double* out64 /*nullable*/, int64_t ldo64, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K11    unit-norm audit — replaces utils.unit_length_violations (hwer/utils.py:51-57) as used by
 *        RecommendationBase.__build_knn__ (hwer/recommendation_base.py:106-107).
 *   out_counts[0..2] = rows with norm > 1+eps, rows with norm < 1-eps, rows with NaN norm
 *   out_absdev_sum   = sum_r |norm_r - 1|  (float64; NaN rows propagate like numpy's mean would)
 *   row_norm_ws      : float workspace of hwer_unit_audit_ws_floats(N) elements (receives every row norm, then partials).
 * ------------------------------------------------------------------------------------------------- */
int64_t hwer_unit_audit_ws_floats(int64_t N);
int hwer_unit_audit_f32(const float* V, int64_t N, int d, int64_t ldv, float eps, float* row_norm_ws,
                        uint64_t* out_counts, double* out_absdev_sum, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K6/A4  per-type table — replaces MultiKNN.__init__ grouping + `vectors[v]` copy
 *        (hwer/recommendation_base.py:64-76): X[i] = V[row_ids[i]] zero-padded to ldx columns,
 *        hnorm[i] = 0.5*||X[i]||^2.  ldx = hwer_knn_padded_dim(d); X / hnorm hold hwer_knn_padded_rows(n_rows) rows.
 * ------------------------------------------------------------------------------------------------- */
int hwer_knn_padded_dim(int d); /* 32, 64, 128 or 256; -1 if d is unsupported (d < 1 or d > 256) */
/* Rows the caller must ALLOCATE for X and hnorm: n_rows rounded up to a multiple of 32.  hwer_pack_rows_f32 fills
 * the pad rows with zeros and a +inf half norm, so that the screen kernel reads whole 32-row tiles without bounds
 * checks and pad rows can never be selected. */
int64_t hwer_knn_padded_rows(int64_t n_rows);
int hwer_pack_rows_f32(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids /*nullable*/,
                       int64_t n_rows, float* X, int64_t ldx, float* hnorm, void* stream);
/* Same, writing the float16 copy of the table (hwer_pack_rows_f16: X scaled by 2^x_exp, rounded to float16) in the SAME pass over
 * the rows - for callers that know a bound on the half norms beforehand (rows that passed the unit-norm audit: 0.5 (1 + eps)^2),
 * x_exp = hwer_knn_f16_scale_exp(bound).  X16 (nullable): hwer_knn_padded_rows(n_rows) * ldx float16 values. */
int hwer_pack_rows_f32_f16(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids /*nullable*/,
                           int64_t n_rows, float* X, int64_t ldx, float* hnorm, void* X16 /*nullable*/, int x_exp, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K3+K4  query construction — replaces get_average_embeddings + the list average at
 *        hwer/recommendation_base.py:153-155,164-170, batched over B anchors with ragged pos / neg lists:
 *        q = mean([unit(V[anchor]), unit(mean V[pos])?, -unit(mean V[neg])?])  (NOT re-normalised),
 *        numpy's summation orders, in float64 or float32 arithmetic (`arith`).  Writes Q64 (B, d) and its float32 rounding Q32 (B, ldq32) zero-padded.
 *   pos_ptr / neg_ptr : (B+1) CSR offsets or NULL (no lists); pos_idx / neg_idx may contain -1 (unknown).
 * ------------------------------------------------------------------------------------------------- */
int hwer_query_mix(const float* V, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                   const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B,
                   int arith /* 0: float64 arithmetic (reference with float64 vectors), 1: float32 */,
                   double* Q64, float* Q32, int64_t ldq32, void* stream);

/* Fused search prologue for small, latency-bound searches: hwer_pack_rows_f32_f16 and hwer_query_mix in ONE launch (both read only
 * the vector table), and - when `knn_workspace` is given - the float32 query fragments of the search that follows, written straight
 * into that workspace: pass the same workspace to hwer_knn_search_opts_f32 with HWER_KNN_QUERIES_READY and it skips its own repacking
 * launch.  bq = hwer_knn_query_padding(B, n_items, d, k, options) of that search.  Same arguments as the two functions; d <= 256.
 * Round 3: bq < 0 asks for the FLOAT16 fragments of a search that will run with HWER_KNN_F16_SCREEN (|bq| rows; per-query
 * power-of-two scales behind them - exactly what the search's own repacking launch would write; ldq32 must be a multiple of 16). */
int hwer_prepare_search_f32(const float* V, int64_t N, int d, int64_t ldv, const int64_t* row_ids /*nullable*/, int64_t n_rows, float* X,
                            int64_t ldx, float* hnorm, void* X16 /*nullable*/, int x_exp, const int32_t* anchor, const int64_t* pos_ptr,
                            const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, int arith, double* Q64,
                            float* Q32, int64_t ldq32, void* knn_workspace /*nullable*/, int64_t bq, void* stream);

/* (reference lines covered: the weighted blend + unit_length of hwer/content_recommender.py:77-96 / hwer/utils.py:43-44, the unit
 * audit of hwer/recommendation_base.py:106-107, the per-type table of MultiKNN.__init__ :64-76 and get_average_embeddings of the
 * anchor :153-155,164-170 - the same lines as the three entry points it fuses.)
 * The whole prologue of a blended-table search step in ONE launch: hwer_blend_normalize_audit_f32 over the rows listed below,
 * hwer_pack_rows_f32_f16 of the `n_rows` rows of the searched type and anchor-only hwer_query_mix (no positive / negative lists)
 * - for small, latency-bound steps (MovieLens-100K / 1M shapes: three launch-floor kernels and a memset node become one kernel).
 * Outputs are bit-identical to the separate calls.
 *   rows of V to blend : the searched type's rows  pack_ids[j] (or pack_row0 + j), j < n_rows   -> V, X, hnorm, X16
 *                        and the other rows        other_ids[i] (or other_row0 + i), i < n_other -> V
 *   queries            : q_b = unit(V[anchor[b]]) in `arith` arithmetic (anchor -1: the reference's clipped row 0); the anchor row
 *                        is re-blended inside the kernel, so no block waits for another
 *   knn_workspace      : REQUIRED - the search's workspace, zeroed once by hwer_knn_workspace_init after allocation (the audit totals
 *                        are accumulated in its header and handed to audit_counts / audit_absdev_sum by the last block; on a
 *                        workspace that was never initialised the audit counts come back as all-ones: the audit fails loudly);
 *                        bq = hwer_knn_query_padding(...) to also write the float32 query fragments (then search with
 *                        HWER_KNN_QUERIES_READY), -bq for the float16 fragments of an HWER_KNN_F16_SCREEN search, 0 to skip them.
 * d <= 256. */
int hwer_knn_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
int hwer_blend_prepare_search_f32(const float* const* tables, int64_t ld_tables, const int32_t* const* row_idx /*nullable*/, const float* alpha,
                                  int alpha_per_row, int T, int d, float* V, int64_t ldv, float eps, uint64_t* audit_counts,
                                  double* audit_absdev_sum, const int64_t* pack_ids /*nullable*/, int64_t pack_row0, int64_t n_rows,
                                  const int64_t* other_ids /*nullable*/, int64_t other_row0, int64_t n_other, float* X, int64_t ldx,
                                  float* hnorm, void* X16 /*nullable*/, int x_exp, const int32_t* anchor, int64_t B, int arith,
                                  double* Q64, float* Q32, int64_t ldq32, void* knn_workspace, int64_t bq, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K5-K7  exact k-NN + re-score + sort — replaces MultiKNN.query (KDTree, hwer/recommendation_base.py:78-83)
 *        and the re-score / sort at :171-173 (variant hwer/gcn_ncf.py:377-386).
 *
 *   Stage 1 (screen) : float32 MFMA  key = q.x - 0.5||x||^2 for all (query, item) pairs, streaming
 *                      threshold filter keeps >= K' = k + slack candidates per query.
 *   Stage 2 (re-rank): float64 squared distance of each candidate in scikit-learn's exact summation order,
 *                      sort by (dist2 asc, id asc), keep k; flags[q] = 1 when the float32 screen provably could
 *                      not have dropped a true top-k item (certificate), else 0 -> call hwer_knn_exact_f64.
 *   Scores           : score_mode 0: ((a.x)+1)/2 in float64, numpy pairwise order (reference with float64 vectors)
 *                      score_mode 1: same in float32 (reference with float32 vectors), widened to double
 *                      score_mode 2: (2 - dist)/2 (GcnNCF without NCF, hwer/gcn_ncf.py:379-383)
 *   Order            : sort_by_score != 0: stable sort by score desc (the reference's final order);
 *                      else ascending (dist2, id) (MultiKNN.query order; what the multi-GPU merge consumes).
 *
 *   X, hnorm : from hwer_pack_rows_f32, allocated with hwer_knn_padded_rows(n_items) rows
 *   Q32      : (B, ldx) zero-padded float32 queries;  Q64 : (B, d) exact queries
 *   A32, a_idx: anchor rows for score_mode 0/1 (NULL for mode 2): row of query b = A32[a_idx ? a_idx[b] : b], stride lda
 *              (pass the whole vector table as A32 and the anchors' global ids as a_idx)
 *   row_ids  : (n_items) local->global id map or NULL (then id = id_base + local)
 *   hnorm_max: max(hnorm) over the table (enters the float32 error bound of the certificate)
 *   outputs  : out_idx (B,k) int64, out_score (B,k) double, out_dist2 (B,k) double, out_flags (B) int32.
 *              out_dist2 may be NULL with score_mode 0/1 and sort_by_score (find_closest_neighbours only returns scores):
 *              then only the candidates whose membership the float32 screen leaves undecided get an exact distance
 *              (results are identical; the re-rank does ~40 % of the float64 work).
 * ------------------------------------------------------------------------------------------------- */
size_t hwer_knn_workspace_bytes(int64_t B, int64_t n_items, int d, int k);
int hwer_knn_max_k(void);
int hwer_knn_max_dim(void); /* widest d the MFMA screen takes (256); wider tables (d <= 1024) go through hwer_knn_all_dist2_f64 */
int hwer_knn_search_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                        const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k,
                        int score_mode,
                        int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                        double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Same search with options.  HWER_KNN_EXACT_THRESHOLDS: no sampled shortcuts (the streaming filter raises its thresholds
 * only to exact K'-th keys, the sampled small-batch path is not taken).  Slower, but a query can then only stay
 * uncertified on genuine near-ties: the retry tier for catalogues whose row order defeats the sampling (e.g. sorted by
 * cluster), before the exhaustive pass.  Workspace size: hwer_knn_workspace_bytes_opts with the same options. */
#define HWER_KNN_EXACT_THRESHOLDS 1u
/* HWER_KNN_F16_SCREEN (needs X16): the streaming screen (large catalogues, more than 1 024 queries) also runs on the float16
 * copy of the table - v_mfma_f32_32x32x8_f16, a quarter of the MFMA cycles - instead of only the small-batch path.  The select
 * stage widens its error bound by the float16 rounding and re-ranks from the float32 rows, so the results are identical;
 * a few more queries may come back uncertified (flag 0) on near-ties and take the exhaustive pass.  Off by default. */
#define HWER_KNN_F16_SCREEN 2u
/* Path selection (results are identical on every path; tests use these to hold each path to the same goldens):
 * HWER_KNN_NO_DENSE: catalogues of up to 8 192 rows also take the streaming screen;  HWER_KNN_NO_SMALL: small query batches
 * over large catalogues take the streaming screen instead of the HBM-bound small-batch path. */
#define HWER_KNN_NO_DENSE 4u
#define HWER_KNN_NO_SMALL 8u
/* HWER_KNN_NO_SEED: the streaming screen starts every candidate list with an open threshold (collect, then select) instead of the
 * per-query threshold seeded from a sample pre-pass.  Same results; kept for A/B measurements and path coverage in the tests. */
#define HWER_KNN_NO_SEED 16u
/* HWER_KNN_QUERIES_READY: the query fragments (float32, or float16 + scales with HWER_KNN_F16_SCREEN: prologue called with bq < 0)
 * are already in the workspace (hwer_prepare_search_f32 wrote them for
 * hwer_knn_query_padding(...) rows); honoured by the dense / streaming screens (float32 and float16), ignored (the search repacks)
 * on the small-batch path. */
#define HWER_KNN_QUERIES_READY 32u
/* HWER_KNN_NO_PINGPONG: the float16 streaming screen as two independent 256-thread workgroups per CU (rounds 1-3) instead of one
 * 512-thread workgroup whose SIMD partners alternate MFMA and filter segments (round 4).  Same candidate lists, same results; kept for
 * same-box A/B measurements and path coverage in the tests. */
#define HWER_KNN_NO_PINGPONG 64u
size_t hwer_knn_workspace_bytes_opts(int64_t B, int64_t n_items, int d, int k, uint32_t options);
int64_t hwer_knn_query_padding(int64_t B, int64_t n_items, int d, int k, uint32_t options); /* rows of query fragments the search reads (>= B) */
int hwer_knn_search_opts_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32,
                             const double* Q64, const float* A32, int64_t lda, const int32_t* a_idx, int64_t B, int k,
                             int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max,
                             int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                             size_t workspace_bytes, uint32_t options, const void* X16 /*nullable*/, int x16_exp, void* stream);

/* The same search as two stream-ordered calls (round 3), for callers that pipeline batches: the screen stage is MFMA / issue bound,
 * the select stage is bound by the gather of the candidates' rows - with the halves of a large batch staggered on two streams, half
 * A's select runs beside half B's screen (hwer/pipeline.py does this for more than 65 536 anchors).
 *   hwer_knn_screen_f32 : query fragments, seed pre-pass, MFMA screen  -> candidate keys in `workspace`
 *   hwer_knn_select_f32 : candidate keys -> float64 re-rank, certificate, re-score, sort -> outputs (+ uncertified list in the header)
 * Same arguments, options, X16 (NULL or not) and workspace for both; hwer_knn_screen_f32 + hwer_knn_select_f32 on one stream IS
 * hwer_knn_search_opts_f32.  Replaces the same reference lines (hwer/recommendation_base.py:78-83, :171-173). */
int hwer_knn_screen_f32(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32, int64_t B, int k,
                        void* workspace, size_t workspace_bytes, uint32_t options, const void* X16 /*nullable*/, int x16_exp, void* stream);
int hwer_knn_select_f32(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32, int64_t lda,
                        const int32_t* a_idx, int64_t B, int k, int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base,
                        float hnorm_max, int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                        size_t workspace_bytes, uint32_t options, const void* X16 /*nullable: only whether the screen had one*/, void* stream);

/* float16 STORAGE of the item table for the HBM-bound small-batch path.  X16 = the padded float32 table
 * X scaled by 2^x_exp (x_exp = hwer_knn_f16_scale_exp(hnorm_max): no element leaves float16's normal range) and rounded
 * to float16.  When passed to hwer_knn_search_opts_f32, small query batches screen against it with the float16 MFMA (half
 * the bytes streamed); the select stage widens its error bound by 2^-10 ||q|| ||x||max and the float64 re-rank still
 * reads the float32 rows, so the results are identical.  X16: n_rows_padded * ldx * 2 bytes. */
int hwer_knn_f16_scale_exp(float hnorm_max);
/* 1 when HWER_KNN_F16_SCREEN pays for B queries against n_items rows of width d (measured on MI355X): always on the streaming
 * path (catalogues above 8 192 rows: 1.8-2.5x), and on the dense path once the screen is 2 B n_items d >= 2e9 flops
 * (6 040 x 3 706 x 128: screen 66 -> 26 us, step 0.224 -> 0.196 ms; 943 x 1 682 x 64: the float16 copy costs more than it saves). */
int hwer_knn_f16_screen_pays(int64_t B, int64_t n_items, int d);
int hwer_pack_rows_f16(const float* X, int64_t n_rows_padded, int64_t ldx, int x_exp, void* X16, void* stream);

/* out_count[0] = number of queries with flags == 0 (device int32): one 4-byte read tells the host whether the
 * exhaustive pass is needed. */
int hwer_knn_count_uncertified(const int32_t* flags, int64_t B, int32_t* out_count, void* stream);

/* On-device exhaustive pass (no host round trip).  hwer_knn_search_f32 keeps, in the first 4 KiB of its workspace, the
 * list of queries it could not certify (flag 0).  This call, stream-ordered right after the search and given the SAME
 * `knn_workspace`, recomputes up to 8 of them exhaustively in float64 (rows overwritten, flags set to 2); with an empty
 * list its blocks exit at once.  Listed queries beyond the 8th stay flagged 0 and are counted into `missed_accum` (a
 * device int32 the caller zero-initialises once; NULL = not counted): those still need hwer_knn_exact_f64.
 * Same argument meaning as hwer_knn_search_f32 / hwer_knn_exact_f64. */
size_t hwer_knn_exact_device_workspace_bytes(int64_t n_items, int k);
int hwer_knn_exact_device(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                          int64_t lda, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                          const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                          double* out_dist2, int32_t* out_flags, void* knn_workspace, int32_t* missed_accum,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Exhaustive float64 k-NN for the (rare) queries whose certificate failed; same outputs / order as above.
 * `qsel` (nq) lists the query rows to recompute; their output rows are overwritten and flags set to 2. */
size_t hwer_knn_exact_workspace_bytes(int64_t n_items, int k);
int hwer_knn_exact_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const float* A32,
                       int64_t lda, const int32_t* a_idx, const int32_t* qsel_host /*host*/, int64_t nq, int k,
                       int score_mode, int sort_by_score,
                       const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                       double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes,
                       void* stream);

/* All float64 squared distances, out[q * n_items + i] = ||Q64[q] - X[i]||^2 in scikit-learn's summation order (the same
 * values the re-rank computes).  The host mirror uses it for k > hwer_knn_max_k() and for d > hwer_knn_max_dim(): the reference
 * accepts any k <= N_t and any width (hwer/recommendation_base.py:74,79), there the k nearest come from a stable device sort of
 * these rows.  1 <= d <= 1024, ldx >= roundup16(d) (X from hwer_pack_rows_f32 with that stride), nq <= 65535. */
int hwer_knn_all_dist2_f64(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, int64_t nq, double* out,
                           void* stream);

/* ---------------------------------------------------------------------------------------------------
 * F64    true float64 `vectors` (round 3).  The reference keeps ContentRecommendation's rows in float64
 *        (hwer/content_recommender.py:96, hwer/recommendation_base.py:109); its KD-tree (:74, :79) and predict (:135-144) work on
 *        them.  The MFMA screen runs on float32 / float16 roundings of the rows - a screen only has to keep a superset - and these
 *        entry points produce the reference's VALUES from the float64 table:
 *   hwer_query_mix_f64rows  : hwer_query_mix over a float64 table (float64 arithmetic; Q32 = the float32 rounding for the screen)
 *   hwer_pair_score_f64rows : hwer_pair_score_f64 over a float64 table
 *   hwer_knn_cand_dist2     : out[b][c] = ||Q64[b] - table[cand[b][c]]||^2 in scikit-learn's summation order for listed candidate
 *                             rows of a float32 (table_is_f64 = 0) or float64 table; cand < 0 -> +inf.  The exact re-rank behind a
 *                             candidate set: hwer_knn_keys_f32 + hwer_knn_topk_rows_f32 for tables wider than hwer_knn_max_dim() / k above
 *                             hwer_knn_max_k() (hwer/device_index.py; float64 rows inside those limits: hwer_knn_search_f64rows).  B <= 65535.
 * ------------------------------------------------------------------------------------------------- */
int hwer_query_mix_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                           const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                           int64_t ldq32, void* stream);
/* hwer_query_mix_f64rows + the MFMA query fragments of the search that follows, written straight into that search's workspace (what hwer_prepare_search_f32
 * does for float32 rows): bq = hwer_knn_query_padding(...) for the float32 fragments, -bq for the float16 fragments + per-query exponents of an
 * HWER_KNN_F16_SCREEN search; then search with HWER_KNN_QUERIES_READY (it skips its own repacking launch).  d <= 256, ldq32 = hwer_knn_padded_dim(d). */
int hwer_query_mix_frag_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* anchor, const int64_t* pos_ptr,
                                const int32_t* pos_idx, const int64_t* neg_ptr, const int32_t* neg_idx, int64_t B, double* Q64, float* Q32,
                                int64_t ldq32, void* knn_workspace, int64_t bq, void* stream);
int hwer_pair_score_f64rows(const double* V64, int64_t N, int d, int64_t ldv, const int32_t* src, const int32_t* dst, int64_t P,
                            double* out, void* stream);
int hwer_knn_cand_dist2(const void* table, int table_is_f64, int64_t n_rows, int d, int64_t ld, const int64_t* cand, int64_t ld_cand,
                        int64_t B, int64_t kc, const double* Q64, double* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * F64S   the FUSED search on genuine float64 rows (round 6) - replaces MultiKNN.query (KDTree over float64 `vectors`,
 *        hwer/recommendation_base.py:74,78-83) and the re-score / sort at :171-173 (predict on float64 rows, :141-143) for the rows
 *        ContentRecommendation.fit returns (hwer/content_recommender.py:96).  Same two stages as hwer_knn_search_opts_f32: the MFMA screen runs
 *        on the float32 (or float16) ROUNDING of the rows - X / hnorm / X16 from hwer_pack_rows_f64rows (or hwer_pack_rows_f32 over a rounded copy) -
 *        and the select stage reads the float64 rows themselves: candidate rows gathered in 64-byte segments (8 columns), squared distances in
 *        scikit-learn's sequential order, scores in numpy's pairwise order from float64 products, certificate widened by the rounding of the
 *        rows (|key32 - key| <= (2 d + 19) 2^-24 (||q|| ||x||max + hnorm_max) (+ the float16 term)), (distance, id) / score order, flags, the
 *        uncertified list in the workspace header - ONE select launch; rounds 3-5 composed ~12 launches (hwer_knn_cand_dist2,
 *        hwer_knn_order_rows_f64, hwer_pair_score_f64rows, ...) for the same result.  Results are bit-identical to the reference on float64 rows.
 *   X64      : (rows, ld64) float64, 64-byte aligned, ld64 % 8 == 0, ld64 >= d, columns >= d ZERO (whole 64-byte segments are loaded)
 *   x64_rows : (n_items) nullable: row of X64 that holds local item i (else i) - pass the whole `vectors` table as X64 and the type's row list
 *   A64      : anchor rows for score_mode 0 (row of query b = A64[(a_idx ? a_idx[b] : b) * lda64]; normally the same table); NULL with score_mode 2
 *   score_mode 0 (float64 scores, numpy order) or 2 (GcnNCF: (2 - dist) / 2); everything else as hwer_knn_search_opts_f32 (same workspace:
 *   hwer_knn_workspace_bytes_opts, same options, same flags / uncertified-list semantics).
 *   hwer_knn_select_f64rows            : the select stage alone, behind hwer_knn_screen_f32 on the same workspace (X only identifies the plan)
 *   hwer_knn_exact_device_f64rows      : hwer_knn_exact_device reading the float64 rows (on-device exhaustive pass of the listed queries)
 *   hwer_knn_exact_f64rows             : hwer_knn_exact_f64 reading the float64 rows (host-listed queries)
 *   hwer_pack_rows_f64rows             : X[i] = float32(V64[row_ids[i]]) zero-padded to ldx, hnorm[i] = 0.5 ||X[i]||^2 (pad rows: zeros, +inf), optional
 *                                        float16 copy scaled by 2^x_exp, and the unit-norm audit of the FLOAT64 rows (hwer/recommendation_base.py:
 *                                        106-107): audit_counts[0..2] += rows with norm > 1 + eps / < 1 - eps / NaN, audit_absdev_sum += |norm - 1|
 *                                        (both nullable, caller-zeroed accumulators).  X / hnorm / X16 hold hwer_knn_padded_rows(n_rows) rows.
 *                                        X = hnorm = X16 = NULL: only the audit of the listed rows (the rows of the other node types).
 *   hwer_audit_pack_rows_f64rows       : both in ONE launch when the searched type's rows are a range of the table: every row of V64 is audited, rows
 *                                        [pack_row0, pack_row0 + n_rows) are also rounded into X / hnorm / X16 (what hwer.pipeline.BlendSearchStepF64 runs).
 * ------------------------------------------------------------------------------------------------- */
int hwer_knn_search_f64rows(const float* X, const float* hnorm, int64_t n_items, int d, int64_t ldx, const float* Q32, const double* Q64,
                            const double* X64, int64_t ld64, const int64_t* x64_rows /*nullable*/, const double* A64, int64_t lda64,
                            const int32_t* a_idx, int64_t B, int k, int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base,
                            float hnorm_max, int64_t* out_idx, double* out_score, double* out_dist2, int32_t* out_flags, void* workspace,
                            size_t workspace_bytes, uint32_t options, const void* X16 /*nullable*/, int x16_exp, void* stream);
int hwer_knn_select_f64rows(const float* X, int64_t n_items, int d, int64_t ldx, const double* Q64, const double* X64, int64_t ld64,
                            const int64_t* x64_rows /*nullable*/, const double* A64, int64_t lda64, const int32_t* a_idx, int64_t B, int k,
                            int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base, float hnorm_max, int64_t* out_idx,
                            double* out_score, double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes, uint32_t options,
                            const void* X16 /*nullable: only whether the screen had one*/, void* stream);
int hwer_knn_exact_device_f64rows(const double* X64, int64_t ld64, const int64_t* x64_rows /*nullable*/, int64_t n_items, int d, const double* Q64,
                                  const double* A64, int64_t lda64, const int32_t* a_idx, int k, int score_mode, int sort_by_score,
                                  const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score, double* out_dist2,
                                  int32_t* out_flags, void* knn_workspace, int32_t* missed_accum, void* workspace, size_t workspace_bytes,
                                  void* stream);
int hwer_knn_exact_f64rows(const double* X64, int64_t ld64, const int64_t* x64_rows /*nullable*/, int64_t n_items, int d, const double* Q64,
                           const double* A64, int64_t lda64, const int32_t* a_idx, const int32_t* qsel_host /*host*/, int64_t nq, int k,
                           int score_mode, int sort_by_score, const int64_t* row_ids, int64_t id_base, int64_t* out_idx, double* out_score,
                           double* out_dist2, int32_t* out_flags, void* workspace, size_t workspace_bytes, void* stream);
int hwer_pack_rows_f64rows(const double* V64, int64_t N, int d, int64_t ld64, const int64_t* row_ids /*nullable*/, int64_t n_rows, float* X,
                           int64_t ldx, float* hnorm, void* X16 /*nullable*/, int x_exp, float eps, uint64_t* audit_counts /*nullable*/,
                           double* audit_absdev_sum /*nullable*/, void* stream);
int hwer_audit_pack_rows_f64rows(const double* V64, int64_t N, int d, int64_t ld64, int64_t pack_row0, int64_t n_rows, float* X, int64_t ldx,
                                 float* hnorm, void* X16 /*nullable*/, int x_exp, float eps, uint64_t* audit_counts, double* audit_absdev_sum /*nullable*/,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------------
 * W1-W4  exact k-NN for the shapes the fused search does not take (round 4) - tables wider than hwer_knn_max_dim(), k above
 *        hwer_knn_max_k(), genuine float64 rows.  The reference's KD-tree takes any width and any k <= N_t
 *        (hwer/recommendation_base.py:74,79; hwer/content_recommender.py:67 emits n_dims-wide rows); its final order is a stable sort
 *        by score (:172-173).  The host mirror (hwer/device_index.py: _search_candidates / _search_exhaustive) composes these into
 *        "screen a superset, re-rank it exactly (hwer_knn_cand_dist2), certify"; rounds 1-3 used a library GEMM / top-k / sort here.
 *
 *   hwer_knn_keys_f32      keys[q * ld_keys + i] = Q32[q] . X[i] - hnorm[i] for q < B, i < n_items: float32 MFMA
 *                          (v_mfma_f32_32x32x2_f32 = an fmaf chain), the columns consumed in 16-column slabs through LDS into the
 *                          same accumulators.  X: (n_rows_padded, ldx) with ldx % 16 == 0 and zero padding (hwer_pack_rows_f32 with
 *                          ldx = roundup16(d) for wide tables), Q32: (B, ldq >= ldx) zero-padded, ld_keys % 4 == 0, >= roundup4(n_items).
 *                          |key - exact| <= (2 d + 32) 2^-24 (||q|| ||x|| + 0.5 ||x||^2) whatever the summation order.
 *   hwer_knn_topk_rows_f32 per query the kc LARGEST of its n keys: out_idx[q][0..kc) = their positions ASCENDING (ties at the cut: the
 *                          lowest positions), out_kth[q] (nullable) = the kc-th largest key.  Exact (byte radix select + ordered compaction).
 *   hwer_knn_topk_rows_f64 the same for the k SMALLEST of n float64 values per row (squared distances of the all-rows fallback).
 *   hwer_knn_order_rows_f64 per query: entries c < n_in with value values[q][c] (or values[q][gather[q][c]]) and id ids[q][c] (default: the
 *                          position c; with gather: gather[q][c]) are put in (value ascending | descending, id ascending) order - what a
 *                          STABLE sort of id-ordered input returns - and the first n_out go to out_values / out_ids (= id_map[id] when id_map
 *                          is given, else id + id_add); payload arrays (B, n_in) are permuted alongside (position ids only);
 *                          out_flags[q] (nullable) = 1 when every_row or bound == NULL or out_values[q][n_out - 1] < bound[q] (the
 *                          certificate of the candidate scheme), else 0.  n_in <= 8 192: one bitonic network in LDS per query; longer rows
 *                          need hwer_knn_order_rows_workspace_bytes(B, n_in) bytes of workspace (rank sort).  B <= 65 535.
 * ------------------------------------------------------------------------------------------------- */
int hwer_knn_keys_f32(const float* X, const float* hnorm, int64_t n_rows_padded, int64_t n_items, int64_t ldx, const float* Q32, int64_t ldq,
                      int64_t B, float* keys, int64_t ld_keys, void* stream);
int hwer_knn_topk_rows_f32(const float* keys, int64_t ld_keys, int64_t n, int64_t B, int64_t kc, int64_t* out_idx, int64_t ld_out,
                           float* out_kth /*nullable*/, void* stream);
int hwer_knn_topk_rows_f64(const double* dist2, int64_t ld, int64_t n, int64_t B, int64_t k, int64_t* out_idx, int64_t ld_out, void* stream);
size_t hwer_knn_order_rows_workspace_bytes(int64_t B, int64_t n_in);
int hwer_knn_order_rows_f64(const double* values, int64_t ld_values, const int64_t* gather /*nullable*/, int64_t ld_gather,
                            const int64_t* ids /*nullable*/, int64_t ld_ids, int64_t B, int64_t n_in, int64_t n_out, int descending,
                            double* out_values, int64_t* out_ids /*nullable*/, int64_t id_add, const int64_t* id_map /*nullable*/,
                            const int64_t* payload_i64 /*nullable*/, int64_t* out_payload_i64, const double* payload_f64 /*nullable*/,
                            double* out_payload_f64, const double* bound /*nullable*/, int32_t* out_flags /*nullable*/, int every_row,
                            void* workspace /*nullable*/, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * K12    seen-item filter + cuts of the Recall@K evaluation — replaces the per-user list comprehensions of
 *        validation.extraction_efficiency (hwer/validation.py:133-141):
 *          unseen = [i for i in ranked[b] if i not in seen(b)][:cut_max]
 *   ranked   : (B, k) int64 ids in score order (out_idx of hwer_knn_search_f32); negative ids are skipped
 *   seen_*   : CSR of each user's training items, every row sorted ascending (NULL = nothing seen)
 *   truth_*  : CSR of each user's validation truth, rows sorted ascending (NULL = no hit marking)
 *   outputs  : out_unseen (B, cut_max) ids padded with -1, out_len (B), out_hits (B, ceil(cut_max/64)) bit p = the p-th
 *              survivor is a truth item; cover_bits (nullable, caller-zeroed, ceil(cover_n/32) words): bit (id - cover_base)
 *              set for every survivor ("diversity", hwer/validation.py:177).
 *   The float metrics (utils.recall / ndcg, hwer/utils.py:101-121) follow from these integers: hwer_rank_metrics below, or the caller.
 * ------------------------------------------------------------------------------------------------- */
int hwer_rank_filter_seen(const int64_t* ranked, int64_t B, int k, const int64_t* seen_ptr, const int64_t* seen_idx,
                          const int64_t* truth_ptr, const int64_t* truth_idx, int cut_max, int64_t* out_unseen,
                          int32_t* out_len, uint64_t* out_hits, uint32_t* cover_bits, int64_t cover_base, int64_t cover_n,
                          void* stream);

/* K12b   the float metrics of the same evaluation, on the device (round 4) - per validation user v (row vrow[v] of the arrays above) and cut c:
 *          recall = hits / max(min(min(n_pred, c), n_true), 1)                                  utils.recall,      hwer/utils.py:118-121
 *          ndcg_b = sum_p hit_p / log2(p + 2)  /  (idcg_table[min(n_true, min(n_pred, c))] + 1e-8)   utils.binary_ndcg, hwer/utils.py:101-111
 *        and out_means[ci] = mean_v recall, out_means[n_cuts + ci] = mean_v ndcg_b (hwer/validation.py:167-172) in numpy's arithmetic: the
 *        per-user sums in np.sum's order over <= 128 terms, the means as np.add.reduce's pairwise recursion / V - bit-identical to the
 *        reference's numpy expressions.  The discounts come from the host as numpy computes them: log2_table[L * ld_log2 + p] = np.log2(np.arange(L) + 2)[p]
 *        for every list length L <= cut_max (numpy's vectorised log2 and its scalar remainder loop may differ in the last place, so the value of
 *        position p depends on the length of the array it was computed in - the reference evaluates np.log2 on an array of the user's own list
 *        length); idcg_table[m] = np.sum(1 / np.log2(np.arange(m) + 2)).  cuts_host: HOST array of n_cuts <= 8 cuts <= min(128, 64 hit_words, ld_log2);
 *        vals_ws: 2 * n_cuts * V doubles (the per-user values stay there: recall rows first, then ndcg rows). */
int hwer_rank_metrics(const uint64_t* hits, int hit_words, const int32_t* n_pred, const int64_t* vrow, const int64_t* n_true, int64_t V,
                      const int32_t* cuts_host /*host*/, int n_cuts, const double* log2_table, int ld_log2, const double* idcg_table, double* vals_ws,
                      double* out_means, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * X1     multi-GPU merge (no reference counterpart): n_lists per-shard results, each
 *        (B,k) ascending (dist2, id), gathered as (n_lists, B, k) -> global k nearest, then the same
 *        optional stable sort by score.
 * ------------------------------------------------------------------------------------------------- */
int hwer_merge_topk(const int64_t* idx, const double* score, const double* dist2, int n_lists, int64_t B, int k,
                    int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2, void* stream);
/* Same merge over lists that are `list_stride` elements apart in each array (>= B * k): lets the receive buffer of the
 * user-partitioned exchange, laid out (rank, array, B, k), be merged in place. */
int hwer_merge_topk_strided(const int64_t* idx, const double* score, const double* dist2, int64_t list_stride, int n_lists,
                            int64_t B, int k, int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2,
                            void* stream);

/* Shortened local lists (round 3): every shard sends only its k_in <= k nearest (n_lists * k_in >= k; n_lists <= 16 when k_in < k),
 * the merge still returns k.  With G shards a shard holds ~k / G of a query's k nearest, so k_in ~ k / G + 8 sqrt(k / G) + 8 is enough
 * for all but skewed catalogues - and the merge PROVES it per query: out_exhausted[b] (nullable, (B) int32) = 1 when some list gave
 * ALL its k_in real entries (id >= 0) to the k winners, i.e. its shard may hold further members; the caller then repeats that query
 * with k_in = k (hwer/distributed.py: ShardedSearch.search).  out_exhausted[b] = 0 means the k winners are exactly what full lists
 * would have produced: a list with an entry left over contributes nothing that sorts behind that entry.
 * A shard's local search costs less with a smaller k (K' = roundup64(k_in + 56) keys kept per query instead of 256 at k = 200;
 * 1.25 M x 128 shard, 4 096 queries: 1.88 -> 1.75 ms) and the exchange moves k_in / k of the bytes. */
int hwer_merge_topk_lists(const int64_t* idx, const double* score, const double* dist2, int64_t list_stride, int n_lists,
                          int64_t B, int k_in, int k, int sort_by_score, int64_t* out_idx, double* out_score, double* out_dist2,
                          int32_t* out_exhausted, void* stream);

/* Send buffer of the user-partitioned exchange (hwer/distributed.py: exchange_by_query): with Bs = ceil(B / n_ranks),
 * send[g][c][j][0..k) = array c (0 idx, 1 score, 2 dist2, as 8-byte patterns) of query g * Bs + j; rows of queries >= B are
 * padding (id -1, score 0, dist2 +inf).  send: n_ranks * 3 * Bs * k int64.  One all_to_all_single with equal splits then
 * hands rank g its slice from every rank, and hwer_merge_topk_strided (list_stride = 3 * Bs * k) merges it in place. */
int hwer_pack_exchange(const int64_t* idx, const double* score, const double* dist2, int64_t B, int k, int n_ranks,
                       int64_t* send, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HWER_HIP_H */
