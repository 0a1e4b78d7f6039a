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

#define KZ_ABI_VERSION 7
/* candidates per query the rescaling / sort kernels take (kz_knn itself returns up to 4096 neighbours) */
#define KZ_MAX_CANDIDATES 4096
/* entries per row kz_merge_topk merges (segments x segment length) */
#define KZ_MERGE_MAX_ENTRIES 8192

/* status codes */
enum { KZ_OK = 0, KZ_ERR_INVALID = 1, KZ_ERR_HIP = 2, KZ_ERR_UNSUPPORTED = 3, KZ_ERR_NOMEM = 4, KZ_ERR_NONFINITE = 5 };
/* element types of an embedding matrix */
enum { KZ_F32 = 0, KZ_F64 = 1 };
/* metrics of the exact backend; minkowski(p=2) == euclidean (sklearn_nearest_neighbors.py:51-65).  0 .. 2 run the fused MFMA
 * kernels; 3 .. 5 (the rest of the Minkowski family: manhattan = cityblock = l1, chebyshev, minkowski with any p >= 1 set by
 * kz_matrix_set_minkowski_p) have no inner-product form: a register-tiled VALU kernel computes scikit-learn's generic
 * DistanceMetric expression (sklearn/metrics/_dist_metrics.pyx.tp: |x_j - y_j| in the input dtype, float64 accumulation in
 * feature order, result rounded to the input dtype), the exact float64 selection kernels pick the neighbours */
enum { KZ_EUCLIDEAN = 0, KZ_SQEUCLIDEAN = 1, KZ_COSINE = 2, KZ_MANHATTAN = 3, KZ_CHEBYSHEV = 4, KZ_MINKOWSKI = 5 };

typedef struct kz_ctx kz_ctx;       /* one GPU + one HIP stream + scratch                                   */
typedef struct kz_matrix kz_matrix; /* an embedding matrix resident in HBM: raw rows, MFMA-packed tiles, norms */

/* Statistics of one kz_knn call (all optional diagnostics; used by bench.py for the roofline line). */
typedef struct kz_knn_stats {
    double main_kernel_ms;   /* HIP-event time of the fused distance+top-k kernel (the dominant kernel)   */
    double finalize_ms;      /* merge + certify + float64 re-rank kernel                                   */
    double fallback_ms;      /* exact float64 brute-force for uncertified rows (0 if none)                 */
    int64_t n_fallback_rows; /* query rows answered by the exact float64 kernels: no approximate tier could certify their
                                candidate set (or, a handful left by a pass, sent there directly: n_spec_rows)         */
    int32_t list_len;        /* K' = per-list candidate count kept by the fused kernel                     */
    int32_t n_splits;        /* index range splits (grid.y)                                                */
    int32_t n_blocks;        /* workgroups launched                                                        */
    int32_t first_pass;      /* operand precision of the fused kernel the call started with: 0 = float32 MFMA,
                                1 = split-bf16 (bf16x2) MFMA, 2 = fp16 MFMA on centred operands               */
    int64_t n_escalated_rows; /* query rows whose candidate set the first pass could not certify and that were
                                searched again one tier down (longer lists -> float32 operands -> exact float64) */
    double max_err_ratio;    /* self-check: max over all re-ranked candidates of |approximate key - exact key| / eps,
                                eps = the rounding bound the certification uses; must stay below 1          */
    int32_t dual;            /* kz_knn_dual: 1 = this direction came out of the shared sweep, 0 = ordinary search */
    int32_t n_first_pass_fail; /* query rows the call's FIRST pass left uncertified, each counted once (n_escalated_rows is
                                cumulative over the levels below: a row that went two levels down counts twice there)  */
    int64_t n_events;        /* kz_knn_dual, reverse direction: events filed for the rows of b                */
    int64_t n_overflow_rows; /* kz_knn_dual, reverse direction: rows of b whose event buffer overflowed (searched again) */
    int64_t n_logged_groups; /* kz_knn_dual, reverse direction: groups of four keys the sweep logged (>= n_events / 4)   */
    int32_t wide_lists;      /* > 0: the call ran the fp16 tier's WIDE route -- that many lists of 16 per query (data whose keys
                                are dense around the k-th neighbour: margin in ranks instead of better operands)           */
    int32_t n_spec_rows;     /* of n_fallback_rows: a handful of rows the first pass left uncertified, answered by the exact kernels
                                launched speculatively behind the finalize kernel (no re-search, no extra synchronisation)        */
    double probe_ms;         /* tier / floor probe of a large search (a strided sample of the query rows searched first), incl. the
                                ladder's second rung; not part of fallback_ms                                                 */
    int64_t n_range_rows;    /* of n_fallback_rows: rows answered by the RANGE re-search -- the exact kernels on the index rows whose
                                approximate key lies within the rounding bound of the row's k-th candidate, not on the whole index */
    int64_t n_range_pairs;   /* (query row, index row) pairs the range re-search evaluated in float64                            */
    int64_t n_range_group_rows; /* of n_range_rows: rows answered as part of a GROUP -- rows of one tight cluster share the range of a
                                representative row: a dense block of pairs instead of one range per row                        */
} kz_knn_stats;

/* ---- library / context -------------------------------------------------------------------------------- */
int kz_abi_version(void);
const char* kz_last_error(void);
int kz_device_count(int* n);
/* stream: an existing hipStream_t to run on (e.g. torch's current stream), or NULL for a private stream. */
int kz_ctx_create(int device, void* stream, kz_ctx** out);
int kz_ctx_destroy(kz_ctx* ctx);
int kz_ctx_sync(kz_ctx* ctx);
/* Hands the context's cached device buffers back to the driver (buffers released by kz_free / kz_matrix_destroy / the
 * searches are kept for reuse -- up to 48 GiB -- so that a repeated fit() does not pay hipMalloc + hipFree; a process that
 * shares the GPU with another allocator, e.g. torch's, calls this after a large search).  Waits for the stream. */
int kz_ctx_trim(kz_ctx* ctx);
/* Options (the public contract -- four names):
 *   "precision"    0 (default) = fp16 first pass on centred operands (16 <= d_pad <= 384), rows it cannot certify go down the
 *                  tiers (more / longer lists -> split-bf16 operands -> float32 operands -> exact float64); 2 = split-bf16 first
 *                  pass; 1 = float32 operands only.  The neighbour order is the float64 one either way.
 *   "dual_stride"  kz_knn_dual samples every n-th tile of a for its thresholds: 1 (default) = chosen from the shapes, 0 = always
 *                  two ordinary searches, 2 .. 64 = that stride.
 *   "dual_max_gb"  transient footprint kz_knn_dual may claim, GiB (0 = 32): beyond it, or beyond what is free, it searches twice.
 *   "eps_scale"    multiplies the certification bound (test knob: a huge value sends every row to the exact float64 kernels).
 * Every other name the call accepts is an internal tuning / diagnostic knob of this build (one table: kiez_amd/csrc/kz_options.h),
 * outside this contract.  Options decide how fast a result arrives, never what it is.  Unknown name or value: KZ_ERR_INVALID. */
int kz_ctx_set_option(kz_ctx* ctx, const char* name, double value);

int kz_malloc(kz_ctx* ctx, size_t bytes, void** d_ptr);
int kz_free(kz_ctx* ctx, void* d_ptr);
int kz_memcpy_h2d(kz_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int kz_memcpy_d2h(kz_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int kz_memcpy_d2d(kz_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);

/* ---- index construction: replaces SklearnNN._fit (sklearn_nearest_neighbors.py:83-94) ------------------ */
/* rows [n, d]: rows_on_device = 0 host memory (copied into HBM), 1 device memory (copied), 2 device memory BORROWED
 * (read in place; the caller keeps the buffer alive and unchanged until kz_matrix_destroy), 3 device memory borrowed as a ROW
 * SOURCE only (no norms, no operand images: accepted by kz_dsl_fit as `source`, refused by every search entry point).  Computes the float64 row
 * norms; the MFMA operand images (fp16 / split-bf16 / float32 tiles) are built by the first kz_knn that needs them.
 * NaN/inf input is an error (KZ_ERR_NONFINITE; scikit-learn rejects those inputs too): reported here for host rows, by
 * the first kz_knn / kz_knn_dual that searches the matrix for device rows (this call then waits for nothing). */
int kz_matrix_create(kz_ctx* ctx, const void* rows, int rows_on_device, int64_t n, int64_t d, int dtype,
                     int metric, kz_matrix** out);
/* metric KZ_MINKOWSKI: the exponent p >= 1 (default 2; both matrices of a search must agree). */
int kz_matrix_set_minkowski_p(kz_matrix* m, double p);
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

/* Both directions between two matrices from ONE sweep of the distance matrix: what HubnessReduction.fit (target ->
 * source neighbours, kiez/hubness_reduction/base.py:48-58) and .kneighbors (source -> target, base.py:95-112) compute
 * with two brute-force searches.  d_dist_ab / d_ind_ab: [a.n, k] = for every row of a its k nearest rows of b;
 * d_dist_ba / d_ind_ba: [b.n, k] the reverse.  Results are identical to kz_knn(a, b) and kz_knn(b, a) (same float64
 * order, same tie rule); pass the LARGER matrix as a (fewer rows get event buffers).  Falls back to two ordinary
 * searches where the shared sweep does not apply (precision != 0, tiny inputs, k > 110).  stats_* may be NULL. */
int kz_knn_dual(kz_ctx* ctx, const kz_matrix* a, const kz_matrix* b, int k, double* d_dist_ab, int64_t* d_ind_ab,
                double* d_dist_ba, int64_t* d_ind_ba, kz_knn_stats* stats_ab, kz_knn_stats* stats_ba);

/* Host-only (no GPU needed): the work schedule kz_knn builds for a launch with `slots` resident workgroups
 * (DESIGN.md section 3.0 "host schedule": greedy rounds).  Round r covers round_qtiles[r] query tiles of 128 rows, each swept as
 * round_pieces[r] index ranges of round_piece_tiles[r] tiles of 128 rows (the last range may be shorter).  Arrays
 * hold up to 8 rounds.  k_eff = neighbours kept per query (k + 1 in single-source mode).
 * Diagnostic ABI: it plans the CLASSIC route -- one candidate list of K' in {16, 32, 64, 128} per query and index range, one
 * query tile per workgroup, k_eff <= 110 -- through the same kz_plan_rounds the product calls.  kz_knn itself also takes
 * routes this function does not describe: several lists of 16 over forced index ranges (13 <= k on a large index, the shared
 * sweep's forward lists, re-searches), the long-k route (111 .. ~540 neighbours) and work items of two or three query tiles
 * (wide and 64-query builds); their plans come out of the same kz_plan_pass with other arguments (tests/host/plan_sanitize.cpp
 * runs those argument ranges under the sanitizers). */
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

/* ---- multi-GPU exchange step (source row-sharded, SURVEY.md section 8e "alternative": per-shard partial top-K + merge) ----- */
/* The reverse search of HubnessReduction.fit (every target row against ALL source rows, base.py:37-42) over a row-sharded
 * source is the merge of the per-shard searches.  kz_knn orders neighbours by the exact float64 value (squared euclidean /
 * cosine distance), ties by smaller row, but RETURNS rounded distances (float32 + euclidean: (double)sqrtf((float)d2)), so
 * per-shard lists are merged by the exact values:
 * kz_pair_values: d_val[r, c] = the value kz_knn ranks index row d_ind[r, c] by for query row q_begin + r -- bit-identical to
 * what the search computed (one canonical dot product); entries with d_ind outside [0, index.n) get +inf. */
int kz_pair_values(kz_ctx* ctx, const kz_matrix* query, int64_t q_begin, int64_t q_count, const kz_matrix* index,
                   const int64_t* d_ind, int k, double* d_val);
/* kz_merge_topk: every row of d_key / d_ind / d_dist [n, segs * seg_len] holds `segs` segments (columns [s seg_len,
 * (s + 1) seg_len)), each sorted ascending by (key, ind) -- one per shard, ind = GLOBAL row ids.  Writes the k smallest
 * entries by (key, ind) in that order: d_odist[n, k] = their d_dist (or their key when d_dist is NULL), d_oind[n, k] = their ind
 * (0 when d_ind is NULL: kinds that only need the merged distances).  segs * seg_len <= KZ_MERGE_MAX_ENTRIES. */
int kz_merge_topk(kz_ctx* ctx, const double* d_key, const int64_t* d_ind, const double* d_dist, int64_t n, int segs, int seg_len,
                  int k, double* d_odist, int64_t* d_oind);

/* ---- multi-GPU: the collectives of the sharded path (kz_comm.hip), over RCCL -- ABI v6 ------------------------------------------
 * One process per GPU.  What the reference's only multi-device call hands to a library (kiez/neighbors/approximate/faiss.py:138,
 * faiss.index_cpu_to_all_gpus) a host in ANY language binds here, without torch: rank 0 draws a 128-byte id (kz_comm_unique_id)
 * and distributes it by whatever the host has (MPI, a file, a socket); every rank calls kz_comm_create with its context.  The
 * collectives are enqueued on the context's stream -- ordered with its kernels, no host synchronisation.  librccl.so is loaded by
 * the first of these calls (dlopen; a copy the process already holds is taken first) and is not a link-time dependency: without
 * RCCL they return KZ_ERR_UNSUPPORTED.  Buffers are device pointers, sizes are bytes.
 *   kz_comm_broadcast           d_buf of rank `root` to every rank, in place                  (the replicated target)
 *   kz_comm_all_gather          d_recv [world][bytes_per_rank] = every rank's d_send          (fit state, shard sizes, source shards)
 *   kz_comm_all_to_all          block r of d_send (send_offset[r], send_bytes[r]) to rank r; d_recv [world][recv_bytes] = the blocks
 *                               received, in rank order                                       (per-shard reverse lists -> kz_merge_topk)
 *   kz_comm_all_reduce_min_f64  element-wise minimum over the ranks, in place                 (DisSimLocal's global shift) */
#define KZ_COMM_ID_BYTES 128
typedef struct kz_comm kz_comm;
int kz_comm_unique_id(void* id128);
int kz_comm_create(kz_ctx* ctx, const void* id128, int rank, int world, kz_comm** out);
int kz_comm_destroy(kz_comm* comm);
int kz_comm_rank(const kz_comm* comm, int* rank, int* world);
int kz_comm_broadcast(kz_comm* comm, void* d_buf, size_t bytes, int root);
int kz_comm_all_gather(kz_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank);
int kz_comm_all_to_all(kz_comm* comm, const void* d_send, const size_t* send_offset, const size_t* send_bytes, void* d_recv,
                       size_t recv_bytes);
int kz_comm_all_reduce_min_f64(kz_comm* comm, double* d_buf, size_t count);

/* Single-source mode (fit(source) only): d_dist / d_ind = [n, K1] result of ONE kz_knn of the matrix against itself for
 * K1 = K + 1 neighbours WITHOUT exclude_self, rows row0 .. row0 + n.  Writes both views the reference computes with two
 * searches: rev = the first K columns (HubnessReduction.fit's reverse pass keeps each row as its own first neighbour,
 * base.py:37-42) and fwd = the row minus itself, removed as sklearn does (neighbors/_base.py:937-965) = kz_knn(..., K,
 * exclude_self = 1).  All four outputs are [n, K]. */
int kz_split_self(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K1, int64_t row0,
                  double* d_rev_dist, int64_t* d_rev_ind, double* d_fwd_dist, int64_t* d_fwd_ind);

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

/* Self-test hook (tests/test_gpu_fast_div.py): the shared-reciprocal division of the cosine re-rank (kz_div_shared, kz_common.h)
 * against the plain float64 division on `count` pseudo-random pairs (numerator a float32 value, divisor a norm >= |a|; `mode`
 * 1: numerators and divisors with all-ones / single-bit significands mixed in); *h_mismatch = pairs whose bits differ. */
int kz_selftest_div(kz_ctx* ctx, int64_t count, uint64_t seed, int mode, int64_t* h_mismatch);

#ifdef __cplusplus
}
#endif
#endif /* KIEZ_AMD_H */
