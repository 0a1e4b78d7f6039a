/*
 * kiez_amd.h — C ABI of the MI355X (gfx950) exact-kNN + hubness-reduction hot path.
 *
 * This is the drop-in boundary for the path
 *     kiez.Kiez.fit / kiez.Kiez.kneighbors                      (reference: kiez/kiez.py:160-223)
 *       -> HubnessReduction.fit / kneighbors / _sort            (kiez/hubness_reduction/base.py:33-105)
 *       -> SklearnNN._fit / _kneighbors                         (kiez/neighbors/exact/sklearn_nearest_neighbors.py:83-101)
 *       -> CSLS / MutualProximity / LocalScaling / DisSimLocal  (kiez/hubness_reduction/{csls,mutual_proximity,local_scaling,dis_sim}.py)
 * The reference has no native code (it is pure Python over scikit-learn), so there is no existing FFI to
 * mirror; the entry points below are what a `ctypes` binding inside the reference's plugin classes
 * (`NNAlgorithm._fit/_kneighbors`, `HubnessReduction._fit/transform`) binds.  INTEGRATION.md shows that stub.
 *
 * Conventions
 *   - plain C, `extern "C"`, no C++/torch types; every function returns 0 on success, non-zero on error and
 *     never throws across the boundary; `kz_last_error()` returns a thread-local message for the last failure.
 *   - all `d_*` pointers are DEVICE pointers (HBM) on the context's GPU; `h_*` are host pointers.
 *   - distances are float64, neighbour indices int64 (what the reference returns for the euclidean family,
 *     SURVEY.md §3.4); matrices are row-major [n, d], float32 or float64.
 *   - one kz_ctx per GPU; calls on one context are serialised by the caller (the reference is single-threaded).
 */
#ifndef KIEZ_AMD_H
#define KIEZ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZ_ABI_VERSION 2

/* status codes */
enum { KZ_OK = 0, KZ_ERR_INVALID = 1, KZ_ERR_HIP = 2, KZ_ERR_UNSUPPORTED = 3, KZ_ERR_NOMEM = 4, KZ_ERR_NONFINITE = 5 };
/* element types of an embedding matrix */
enum { KZ_F32 = 0, KZ_F64 = 1 };
/* metrics of the exact backend; minkowski(p=2) == euclidean (sklearn_nearest_neighbors.py:51-65) */
enum { KZ_EUCLIDEAN = 0, KZ_SQEUCLIDEAN = 1, KZ_COSINE = 2 };

typedef struct kz_ctx kz_ctx;       /* one GPU + one HIP stream + scratch                                   */
typedef struct kz_matrix kz_matrix; /* an embedding matrix resident in HBM: raw rows, MFMA-packed tiles, norms */

/* Statistics of one kz_knn call (all optional diagnostics; used by bench.py for the roofline line). */
typedef struct kz_knn_stats {
    double main_kernel_ms;   /* HIP-event time of the fused distance+top-k kernel (the dominant kernel)   */
    double finalize_ms;      /* merge + certify + float64 re-rank kernel                                   */
    double fallback_ms;      /* exact float64 brute-force for uncertified rows (0 if none)                 */
    int64_t n_fallback_rows; /* query rows whose candidate set could not be certified                      */
    int32_t list_len;        /* K' = per-list candidate count kept by the fused kernel                     */
    int32_t n_splits;        /* index range splits (grid.y)                                                */
    int32_t n_blocks;        /* workgroups launched                                                        */
    int32_t first_pass;      /* operand precision of the fused kernel that produced the result of the last
                                chunk: 0 = float32 MFMA, 1 = split-bf16 (bf16x2) MFMA                        */
    int64_t n_escalated_rows; /* query rows first tried with the split-bf16 pass and re-done with float32
                                operands because they (or too many rows of their chunk) failed its certification */
    double max_err_ratio;    /* self-check: max over all re-ranked candidates of |approximate key - exact key| / eps,
                                eps = the rounding bound the certification uses; must stay below 1          */
} kz_knn_stats;

/* ---- library / context -------------------------------------------------------------------------------- */
int kz_abi_version(void);
const char* kz_last_error(void);
int kz_device_count(int* n);
/* stream: an existing hipStream_t to run on (e.g. torch's current stream), or NULL for a private stream. */
int kz_ctx_create(int device, void* stream, kz_ctx** out);
int kz_ctx_destroy(kz_ctx* ctx);
int kz_ctx_sync(kz_ctx* ctx);
/* Options.  "precision": 0 (default) = split-bf16 first pass where the query tile fits in registers (d <= 128),
 * escalating to float32 operands / exact float64 per chunk as certification demands; 1 = float32 operands only.
 * The neighbour order is the float64 one either way.  Test/diagnostic knobs: "eps_scale" multiplies the
 * certification bound (huge value => every row takes the exact fallback); "force_splits" fixes the index split
 * count (0 = automatic); "kernel_variant" selects experimental float32 kernels. */
int kz_ctx_set_option(kz_ctx* ctx, const char* name, double value);

int kz_malloc(kz_ctx* ctx, size_t bytes, void** d_ptr);
int kz_free(kz_ctx* ctx, void* d_ptr);
int kz_memcpy_h2d(kz_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int kz_memcpy_d2h(kz_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int kz_memcpy_d2d(kz_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* ---- index construction: replaces SklearnNN._fit (sklearn_nearest_neighbors.py:83-94) ------------------ */
/* Copies rows [n, d] (host or device memory) into HBM, computes float64 row norms, and writes the
 * MFMA-packed float32 tile image used by the distance kernel.  Fails with KZ_ERR_NONFINITE on NaN/inf
 * (scikit-learn rejects those inputs too). */
int kz_matrix_create(kz_ctx* ctx, const void* rows, int rows_on_device, int64_t n, int64_t d, int dtype,
                     int metric, kz_matrix** out);
int kz_matrix_destroy(kz_matrix* m);
int kz_matrix_shape(const kz_matrix* m, int64_t* n, int64_t* d, int* dtype, int* metric);

/* ---- exact kNN: replaces SklearnNN._kneighbors (sklearn_nearest_neighbors.py:96-101) -------------------- */
/* For query rows [q_begin, q_begin+q_count) of `query` find the k nearest rows of `index`, ascending.
 * exclude_self != 0: query row r is index row r and is removed the way sklearn does for X=None
 * (sklearn/neighbors/_base.py:828-834, 937-965).  d_dist: [q_count, k] float64, d_ind: [q_count, k] int64.
 * Indices follow the float64 distance order (ties by smaller index); distances follow sklearn's dtype rule
 * (float32 inputs + euclidean: (double)sqrtf((float)d2)). */
int kz_knn(kz_ctx* ctx, const kz_matrix* query, int64_t q_begin, int64_t q_count, const kz_matrix* index,
           int k, int exclude_self, double* d_dist, int64_t* d_ind, kz_knn_stats* stats);

/* Host-only (no GPU needed): the work schedule kz_knn builds for a launch with `slots` resident workgroups
 * (DESIGN.md section 3.1 "greedy rounds").  Round r covers round_qtiles[r] query tiles of 128 rows, each swept as
 * round_pieces[r] index ranges of round_piece_tiles[r] tiles of 128 rows (the last range may be shorter).  Arrays
 * hold up to 8 rounds.  k_eff = neighbours kept per query (k + 1 in single-source mode). */
int kz_knn_plan(int64_t n_query_rows, int64_t n_index_rows, int k_eff, int slots, int force_splits, int min_splits,
                int* n_rounds, int* round_qtiles, int* round_pieces, int* round_piece_tiles);

/* ---- per-row statistics of a [n, K] distance array (numpy summation order) ----------------------------- */
/* mean: ndarray.mean(axis=1); std: np.nanstd(axis=1) (ddof=0); last: column K-1.  Any output may be NULL.
 * Used for the fit state of CSLS (csls.py:90), NICDM (local_scaling.py:143), LS (:136), MP normal
 * (mutual_proximity.py:102-103). */
int kz_row_stats(kz_ctx* ctx, const double* d_dist, int64_t n, int K, double* d_mean, double* d_std, double* d_last);

/* ---- hubness rescaling: replace HubnessReduction.transform of each method ------------------------------ */
/* All take the forward candidates d_dist/d_ind [n, K] and write the UNSORTED rescaled distances d_out [n, K]. */
/* CSLS.transform, csls.py:85-96.  r_train[j] = mean_K dist_t2s[j,:] (length n_t). */
int kz_csls(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_r_train,
            double* d_out);
/* LocalScaling.transform, local_scaling.py:129-151.  nicdm == 0: r_t = K-th reverse distance (:135-140);
 * nicdm != 0: r_t = mean reverse distance (:142-147). */
int kz_local_scaling(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_r_t,
                     int nicdm, double* d_out);
/* MutualProximity 'normal', mutual_proximity.py:166-183. */
int kz_mp_normal(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_mu_t,
                 const double* d_sd_t, double* d_out);
/* MutualProximity 'empiric', mutual_proximity.py:185-212 (target ids are looked up in the reverse lists of
 * source ids, as the reference does). */
int kz_mp_empiric(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K,
                  const double* d_dist_t2s, const int64_t* d_ind_t2s, int64_t n_t, int Kt, double* d_out);
/* DisSimLocal._fit, dis_sim.py:96-107: t2c[j] = |target[t_begin+j] - mean_K source[ind_t2s[j,:]]|^2 for
 * n_rows reverse-list rows. */
int kz_dsl_fit(kz_ctx* ctx, const int64_t* d_ind_t2s, int64_t n_rows, int Kt, const kz_matrix* source,
               const kz_matrix* target, int64_t t_begin, double* d_t2c);
/* DisSimLocal.transform, dis_sim.py:139-166 (before the global shift): writes d_out and the running
 * minimum into *d_min (caller initialises *d_min to +inf; multi-GPU callers all-reduce it with MIN). */
int kz_dsl_transform(kz_ctx* ctx, const int64_t* d_ind, int64_t n, int K, const kz_matrix* query, int64_t q_begin,
                     const kz_matrix* target, const double* d_t2c, double* d_out, double* d_min);
/* dis_sim.py:171-177: shift by -min if min < 0, then sqrt unless squared. */
int kz_dsl_finalize(kz_ctx* ctx, double* d_out, int64_t count, double min_value, int squared);

/* ---- final candidate sort: replaces HubnessReduction._sort (base.py:72-87, numpy branch) ---------------- */
/* Selection sort with swaps of the first k positions == np.argpartition(kth=arange(k)) + take_along_axis
 * (SURVEY.md §8 a-6).  d_odist/d_oind: [n, k]. */
int kz_select_topk(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, int k,
                   double* d_odist, int64_t* d_oind);

/* ---- "next" row (f-1): kiez.analysis.hubness_score on the neighbour index matrix (kiez/analysis/estimation.py:197-351) */
/* min / max of an int64 device array (validation + bincount length). */
int kz_minmax_i64(kz_ctx* ctx, const int64_t* d_in, int64_t count, int64_t* h_min, int64_t* h_max);
/* the same over the first k columns of a row-major [rows, cols] matrix: the reference slices nn_ind[:, :k] BEFORE
 * np.bincount(minlength=n_train) (estimation.py:276-292), so ids in the dropped columns must not size the histogram. */
int kz_minmax_i64_2d(kz_ctx* ctx, const int64_t* d_in, int64_t rows, int cols, int k, int64_t* h_min, int64_t* h_max);
/* k-occurrence: np.bincount(nn_ind[:, :k].ravel(), minlength=n_bins) with negative ids dropped (estimation.py:283-292).
 * d_ind: [n_rows, cols] int64; d_kocc: [n_bins] int64. */
int kz_k_occurrence(kz_ctx* ctx, const int64_t* d_ind, int64_t n_rows, int cols, int k, int64_t n_bins, int64_t* d_kocc);
/* Reductions of the k-occurrence vector for skewness, Robin Hood, Atkinson, hub/antihub statistics and (optionally, O(n^2))
 * the Gini numerator.  h_out[10] (host): sum, sum|x-mean|, sum(x-mean)^2, sum(x-mean)^3, sum sqrt(x), max, #zeros,
 * sum over hubs (x >= hub_threshold), #hubs, gini numerator. */
int kz_kocc_stats(kz_ctx* ctx, const int64_t* d_kocc, int64_t n, double hub_threshold, int with_gini, double* h_out);
/* np.argwhere(kocc == 0) (mode 0) / np.argwhere(kocc >= thr) (mode 1), ascending; d_out must hold n entries. */
int kz_kocc_select(kz_ctx* ctx, const int64_t* d_kocc, int64_t n, int mode, double thr, int64_t* d_out, int64_t* h_count);

/* "next" row (f-2): kiez.evaluate.hits (kiez/evaluate/eval_metrics.py:23-61).  d_gold[i] = gold target of source row i or
 * INT64_MIN when the row has no gold entry; d_hist[c] (c < cols) = rows whose gold id sits at position c, d_hist[cols] =
 * rows whose gold id is absent.  hits@k = sum(d_hist[:k]) / len(gold). */
int kz_hit_positions(kz_ctx* ctx, const int64_t* d_ind, const int64_t* d_gold, int64_t n, int cols, int64_t* d_hist);

/* float64 -> float32 cast of an [count] array (cosine + float32 inputs keep the reference's output dtype). */
int kz_cast_f64_f32(kz_ctx* ctx, const double* d_in, float* d_out, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* KIEZ_AMD_H */
