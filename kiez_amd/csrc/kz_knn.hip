// Exact k-nearest-neighbour search on MI355X (gfx950):  replaces SklearnNN._kneighbors
// (kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101 -> sklearn brute-force ArgKmin,
//  sklearn/metrics/_pairwise_distances_reduction/_argkmin.pyx.tp:311-510).
//
// Three stages (DESIGN.md "Kernels"):
//   1. kz_knn_cand_kernel   fused  X.Y^T (float32 MFMA 32x32x2)  +  per-query top-K' candidate lists.
//                           The n_q x n_i similarity matrix never leaves registers.
//   2. kz_knn_finalize      merge the lists, CERTIFY that the true top-k is inside the candidate set using a
//                           rigorous float32 rounding bound, re-rank the K' candidates with exact float64
//                           distances, sort, strip the query itself (single-source mode), write [q, k].
//   3. kz_exact_*           exact float64 brute force for the (rare) rows that could not be certified.
// Result: neighbour order == order of the float64 distances the reference computes; no approximation.
#include <vector>

#include <algorithm>

#include "kz_common.h"
#include "kz_floor.h"

#include "kz_knn_device.h"

// The shipped fused kernel (kernel_variant 0).
// NRES = number of leading 16-k slices whose QUERY fragments stay resident in registers for the whole sweep
// (8 slices = d 128 = 64 VGPRs); slices beyond NRES are streamed from L2 with non-temporal loads.  Residency is opt-in
// (force_nres): at the register budget of three waves per SIMD it spills and measured slower than streaming
// (HISTORY.md section 7); NRES = 0 is what runs by default.
template <int KP, int NRES>
__global__ __launch_bounds__(256, 3) void kz_knn_cand_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);  // 2 x 2048 floats (+ 2 x 128 bias floats behind them)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int h = lane >> 5;

    // The host schedules the work (kz_knn): large items first, small items to fill the tail, and an XCD-aware
    // order (blocks b and b+8 share an XCD and its L2: co-resident blocks stream the SAME index range).
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x;
    const int t_begin = wd.y;
    const int t_end = wd.z;
    const int s = wd.w;
    const int NS = p.kg >> 2;  // >= NRES (host picks NRES)
    const int total = (t_end - t_begin) * NS;

    // Candidate state of this lane = one (query, lane-half) pair:
    //   list  K' best (key, row) so far, UNSORTED, living directly in the output arrays (global memory, L2-resident);
    //         touched only at merges;
    //   log   up to KZ_LOG_CAP keys that beat the pruning threshold since the last merge (LDS, append-only).
    // The threshold is only refreshed at merges, which follow a geometric schedule in the number of tiles seen
    // (identical for every lane, so merges run with all 64 lanes busy); a full log forces an early merge.
    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * wave + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_CAND_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_CAND_LDS_BASE + KZ_LOG_CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;

    if (total > 0) {
        const float4* ysrc = reinterpret_cast<const float4*>(p.ypack + ((int64_t)t_begin * NS) * 2048);
        const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * p.kg) * 512 + (32 * wave + j) * 4;
        float* bbuf = ybuf + 4096;  // 2 x 128 floats: accumulator-init (bias) rows of the current / next tile
        // prologue: slice 0 and the bias rows of the first tile
        {
            float4* nb = reinterpret_cast<float4*>(ybuf);
            nb[tid] = ysrc[tid];
            nb[tid + 256] = ysrc[256 + tid];
            bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
        }
        // resident query fragments: lane (j, h) needs k-groups 4*sl + 2*t + h of its query row
        float4 qres[NRES > 0 ? NRES : 1][2];
#pragma unroll
        for (int u = 0; u < NRES; ++u) {
            qres[u][0] = *reinterpret_cast<const float4*>(qbase + (4 * u + h) * 512);
            qres[u][1] = *reinterpret_cast<const float4*>(qbase + (4 * u + 2 + h) * 512);
        }
        // streamed query fragments (slices >= NRES), current slice in qb
        float4 qb0 = make_float4(0.f, 0.f, 0.f, 0.f), qb1 = qb0;
        if (NRES == 0 || NS > NRES) {
            qb0 = *reinterpret_cast<const float4*>(qbase + (4 * NRES + h) * 512);
            qb1 = *reinterpret_cast<const float4*>(qbase + (4 * NRES + 2 + h) * 512);
        }
        __syncthreads();

        int g = 0;
        f32x16 acc[4];
        const float* bias_n = p.ybias + (tid & 127);
        int tile = t_begin;

        // One 16-k slice: prefetch the next index slice (HBM/L2 -> registers) and the next tile's bias rows, run the
        // 32 MFMAs of the current slice out of LDS, then refill the other LDS buffer.  All loads are UNCONDITIONAL so
        // that hipcc's s_waitcnt placement keeps them in flight behind the MFMAs (a clamp re-reads the last slice at
        // the very end, harmless); sched_barriers pin loads above and the LDS refill below the MFMA block.
        // Lane (j, h) feeds k = 4*(2t+h)+jj for jj = 0..3: the k order inside a slice is permuted identically for
        // A and B, which leaves the dot product unchanged.
        auto slice_step = [&](const float4& bq0, const float4& bq1, const bool stream_q, const int sl_next) {
            const int gn = min(g + 1, total - 1);
            const float4* src = ysrc + (int64_t)gn * 512;
            const float4 ya0 = src[tid];
            const float4 ya1 = src[256 + tid];
            const int tile_n = min(tile + 1, p.n_ytiles - 1);
            const float bn = bias_n[(int64_t)tile_n * KZ_TILE];
            float4 qn0, qn1;
            if (stream_q) {
                // streaming policy for the query fragments: +2.6 % on C1 (they are never re-used from L1/L2 soon)
                qn0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sl_next + h) * 512));
                qn1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sl_next + 2 + h) * 512));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* buf = ybuf + (g & 1) * 2048;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    a[mt] = *reinterpret_cast<const float4*>(buf + ((2 * t + h) * KZ_TILE + 32 * mt + j) * 4);
                const float4 bq = t ? bq1 : bq0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float4* nb = reinterpret_cast<float4*>(ybuf + ((g + 1) & 1) * 2048);
                nb[tid] = ya0;
                nb[tid + 256] = ya1;
                bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
                if (stream_q) {
                    qb0 = qn0;
                    qb1 = qn1;
                }
            }
            // Raw barrier: __syncthreads() is fence + s_barrier and the fence drains vmcnt(0), i.e. it would wait here for
            // the query-fragment loads that are only needed at the top of the next slice.  LDS visibility needs lgkmcnt only.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            ++g;
        };

        for (; tile < t_end; ++tile) {
            __builtin_amdgcn_sched_barrier(0);  // do not hoist the next tile's init above the epilogue (64 VGPRs)
            {
                const float* bp = bbuf + (tile & 1) * 128 + 4 * h;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 v = *reinterpret_cast<const float4*>(bp + 32 * mt + 8 * g4);
                        acc[mt][4 * g4 + 0] = v.x;
                        acc[mt][4 * g4 + 1] = v.y;
                        acc[mt][4 * g4 + 2] = v.z;
                        acc[mt][4 * g4 + 3] = v.w;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // resident slices: fully unrolled, fragments by static register index
#pragma unroll
            for (int u = 0; u < NRES; ++u) {
                // the last resident slice prefetches the first streamed one (if any)
                slice_step(qres[u][0], qres[u][1], (u == NRES - 1) && (NS > NRES), NRES);
            }
            // streamed slices
            if (NRES == 0 || NS > NRES) {
                int sl = NRES;
                do {  // the do-while spares the compiler a zero-trip path
                    const int sln = (sl + 1 == NS) ? NRES : sl + 1;
                    const float4 c0 = qb0, c1 = qb1;
                    slice_step(c0, c1, true, sln);
                } while (++sl < NS);
            }
            __builtin_amdgcn_sched_barrier(0);
            kz_tile_epilogue<KP>(acc, st, tile, tile == t_end - 1, h);
        }
    }
}

#include "kz_knn_bf16.h"

// ---------------------------------------------------------------------------------------------------
// Stage 2: merge + certify + float64 re-rank
// ---------------------------------------------------------------------------------------------------

struct KnnFinParams {
    const float* in_key;  // [rows][M]
    const int* in_idx;
    KzListLayout lay;     // list layout (kz_list_base)
    int max_m;            // largest entry count of a query in this launch (sizes the dynamic LDS)
    int fast_div;         // cosine re-rank: y_k / |y| as kz_div_shared (one reciprocal per candidate row; same bits as the division)
    int64_t q_first, q_last;  // local query range [q_first, q_last) handled by this launch
    int KP;               // entries per list (per query and index range)
    int KSEL;             // candidates the finalize kernel selects from a query's lists and re-ranks (0: = KP).  Larger than KP on the
                          // long-k route (more than 110 neighbours: lists of 128 over many index ranges, kz_knn_impl)
    int64_t list_row0;    // list row of local query 0  (= q_begin - qt0*128)
    int64_t q_begin;      // global query row of local query 0
    int64_t q_count;
    const void* qraw;     // raw query rows (global row indexing)
    const void* yraw;     // raw index rows
    const double* ynorm64; // cosine, float32 rows: the index rows normalised in float64 (kz_matrix_norm64), or NULL
    const double* qsqn;
    const double* ysqn;
    int64_t n_i;
    int d;
    int metric;
    int k;                // neighbours to return
    int exclude_self;
    const int64_t* self_ids;  // optional: index row to strip per local query (escalated subsets); NULL = q_begin + q
    double gamma;         // rounding-bound factor (already multiplied by eps_scale)
    const double* ystats; // index matrix: [0] max row norm (device)
    // fp16 first pass (kz_knn_h16.h): keys are in centred, scaled units; the bound uses the measured operand residuals
    int tier_h;
    double eps_mult;      // eps_scale (test knob)
    double gamma_acc;     // float32 accumulation part of the bound
    const double* q_rowq; // query image: [n][3] = |x_c|^2, |x_h|, |x_c - x_h|
    const double* y_hmax; // index image: max |y_h|, max |y_c - y_h|, max |y_c|^2
    const double* hscale; // {S, 1 / S^2}
    // dual pass, reverse direction (kz_knn_dual.h): the list holds the K' best EVENTS of the row; rows outside the events
    // have an approximate key below excl_floor[q] (+inf: the row's events are incomplete, it must fail)
    const float* excl_floor;
    // seeded lists (KnnCandParams::qfloor): [q_begin + q] the key the query's lists started from -- rows that never entered a list
    // have an approximate key at or below it
    const float* list_floor;
    int dual_col;
    const int* idx_map;   // dual pass, forward direction: list entry r stands for index row idx_map[r] (NULL: identity)
    const int* row_map;   // dual pass, forward direction: the query image is permuted too -- image row r is matrix row row_map[r];
                          // raw row, norms, residuals, the output position and the fail-list entry all go by the MATRIX row
    double* out_dist;     // [q_count][k]
    int64_t* out_ind;
    int* fail_count;
    int* fail_list;
    double* fail_tau;     // optional, beside fail_list: the exact value of the row's k-th best CANDIDATE (+inf: fewer than k candidates) -- an
                          // upper bound of its k-th neighbour's value whatever the tier: what the range re-search starts from (kz_range.h)
    unsigned long long* err_ratio_bits;  // max over certified candidates of |key~ - key| / eps (bits of a non-negative double)
};

// (kz_exact_value: kz_common.h -- shared with kz_pair_values, which must reproduce the re-rank's values bit for bit)
template <typename T>
__device__ __forceinline__ double kz_output_distance(double v, int metric, double p = 2.0) {
    // (Minkowski family: the ranking value is the reduced distance; scikit-learn converts at the end,
    //  MinkowskiDistance._rdist_to_dist: rdist ** (1 / p), rounded to the input dtype -- measured on scikit-learn 1.7.2)
    if (metric == KZ_MINKOWSKI) return sizeof(T) == 4 ? (double)(float)pow(v, 1.0 / p) : pow(v, 1.0 / p);
    if (metric == KZ_EUCLIDEAN) {
        // ArgKmin32 converts the surrogate with the float32 metric object: (double)sqrtf((float)d2)
        // (_argkmin.pyx.tp:285-295 with INPUT_DTYPE_t = float32); ArgKmin64 uses sqrt in float64.
        // exactly what sklearn executes: float32 argument, double sqrt, result rounded back to float32
        if (sizeof(T) == 4) return (double)(float)sqrt((double)(float)v);
        return sqrt(v);
    }
    return v;
}

// Writes the final k entries of one query from its (value, idx)-sorted prefix.  sorted arrays live in LDS.
// sklearn self removal (neighbors/_base.py:947-965): among the first k+1, drop the entry whose index is the
// query row; if it is absent drop the first one.
template <typename T>
__device__ __forceinline__ void kz_emit_sorted(const double* sval, const int* sidx, int n_sorted, int k, int exclude_self,
                                               int64_t self_row, int metric, double* od, int64_t* oi, int lane, double p = 2.0) {
    int self_rank = -1;
    if (exclude_self) {
        self_rank = 0;
        const int lim = min(n_sorted, k + 1);
        for (int c = 0; c < lim; ++c)
            if ((int64_t)sidx[c] == self_row) {
                self_rank = c;
                break;
            }
    }
    for (int c = lane; c < n_sorted; c += 64) {
        if (c == self_rank) continue;
        const int o = (self_rank >= 0 && c > self_rank) ? c - 1 : c;
        if (o < k) {
            od[o] = kz_output_distance<T>(sval[c], metric, p);
            oi[o] = (int64_t)sidx[c];
        }
    }
}

// Gather parallelism of the finalize kernel: candidate rows per group (KZ_FIN_ROWS), groups in flight per wave (KZ_FIN_DEPTH:
// 2 = one group ahead, 3 = two) and the occupancy the kernel is compiled for (KZ_FIN_WAVES).  Round 3, same box, average
// launch on ns / C3 (tools/job_fin.sh): rows 4 depth 2 at 4 waves per SIMD (round 2's build, 112 VGPRs) 4.06 / 8.47 ms; rows 2
// depth 3 at 5 waves (4 spilled) 3.57 / 7.60; rows 1 depth 2 at 7 waves (70 VGPRs, no spill) 3.03 / 7.37; rows 1 depth 3 at 7
// (6 spilled) 3.14 / 7.25; rows 4 depth 3 at 3 waves 4.96 / 9.59.  Waves in flight beat rows in flight per wave: the phases
// around the gather loop (list load, rank select, rank sort) of one query hide under the gathers of the other waves' queries.
// Round 4 (the loads of the loop issued without branches, so that the prefetch overlaps at all; the per-query values in scalar
// registers: 71 -> 59 VGPRs), finalize time over 4 steps of C3 + 4 of ns, reverse chain not overlapped: rows 1 depth 2 at 8 waves
// 69.2 ms; rows 2 depth 2 at 7 (70 VGPRs) 69.1; rows 1 depth 3 at 7 69.5; **rows 1 depth 3 at 8 (64 VGPRs, no spill) 67.4**; rows 2
// depth 3 at 6 72.9.  Round 5: the selection paths added since (unsorted path, radix selections) brought the 8-wave build to 8
// spilled VGPRs; 7 waves (72 VGPRs, none spilled), same box, two runs each: ns 98.14 / 98.04 -> 97.56 / 97.16 ms per step, C3
// 122.43 / 122.34 -> 122.19 / 121.87.
#ifndef KZ_FIN_ROWS_N
#define KZ_FIN_ROWS_N 1
#endif
#ifndef KZ_FIN_DEPTH
#define KZ_FIN_DEPTH 3
#endif
constexpr int KZ_FIN_ROWS = KZ_FIN_ROWS_N;
constexpr int KZ_FIN_MAXM = 4096;  // list entries per query: 4 waves x (4096*8 + 128*28) B = 142 KiB of LDS at most
// Rows a K' = 16 pass could not certify: few (the usual handful) -> more lists of 16, it is all latency; many (hard data) -> lists
// of 64, which certify more of them in one go (400k x 400k, k = 10, clusters of very different density: 140 against 112 ms)
constexpr int KZ_ESC_SHORT_MAX_ROWS = 2048;
constexpr int KZ_MAX_PIECES = 128;  // index ranges per query tile (each range keeps its own K'-entry list per query).  Round 4: 64 -> 128
                                    // (a search of 8 .. 700 rows against 1 M: 0.84 .. 0.92 -> 0.61 .. 0.74 ms; 256: the finalize kernel's selection eats the gain)
static int kz_max_pieces(int KP, int halves) {
    const int m = KZ_FIN_MAXM / (halves * KP);
    return m < KZ_MAX_PIECES ? m : KZ_MAX_PIECES;
}

// Per-wave LDS of the finalize kernel for a launch whose queries hold at most max_m list entries.
__host__ __device__ __forceinline__ int kz_fin_wave_bytes(int max_m, int KP) {
    return ((max_m * 8 + KP * 28) + 15) & ~15;
}
// (the finalize kernel for many candidates shares bytes between arrays that are never live together: kz_knn_fin_wide.h)
__host__ __device__ __forceinline__ int kz_fin_wide_wave_bytes(int max_m, int KS) {
    return ((max_m * 8 + KS * 20) + 15) & ~15;
}

// k-th largest (rank = 1: the largest) of n float keys held as SORTABLE unsigned patterns in LDS; returns the pattern.
// Wave-cooperative: 32 counting passes at most, fewer below the common prefix of the patterns.  The entries are read ONCE into
// registers (E per lane, n <= 64 E): a counting pass is then E compares and E ballots, no LDS round trip in the dependent chain
// bit -> count -> next bit (round 5: the finalize kernel for many candidates runs three such selections per query at three waves
// per SIMD -- the chains, not the instruction count, were what it waited for).
// Largest "smallest key of a FULL list" over a query's lists of KP = 16 or 32 entries, the entries held E per lane (entry e = lane +
// 64 i): a list's entries sit in KP consecutive lanes of one i, so every list is reduced inside its lane group -- all lists of an
// i at once, no loop over the lists (32 lists: 8 x 4 shuffle steps instead of 32 dependent rounds of 5).  -inf: no full list.
template <int E>
__device__ __forceinline__ float kz_full_lists_bound(const float (&key)[E], const bool (&ok)[E], int M, int KP, int lane) {
    float bound = -INFINITY;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        if (64 * i >= M) break;   // (uniform)
        int c = ok[i] ? 1 : 0;
        float mn = ok[i] ? key[i] : INFINITY;
        for (int off = KP >> 1; off >= 1; off >>= 1) {   // (uniform trip count: 4 or 5)
            c += __shfl_xor(c, off, 64);
            mn = fminf(mn, __shfl_xor(mn, off, 64));
        }
        if (c == KP && lane + 64 * i < M) bound = fmaxf(bound, mn);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) bound = fmaxf(bound, __shfl_xor(bound, off, 64));
    return bound;
}

// (core: the lane's E patterns in registers; pattern 0 = no entry)
template <int E>
__device__ __forceinline__ unsigned kz_radix_kth_u32_regs(const unsigned (&x)[E], unsigned all_or, unsigned all_and, int rank) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const unsigned differ = all_or ^ all_and;
    const int top = differ ? 31 - __clz(differ) : -1;
    unsigned thr = top >= 31 ? 0u : (top < 0 ? all_and : (all_and & ~((2u << top) - 1u)));
    for (int bit = top; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < E; ++i) c += (int)__popcll(__ballot(x[i] >= cand));
        if (c >= rank) thr = cand;
    }
    return thr;
}
template <int E>
__device__ __forceinline__ unsigned kz_radix_kth_u32_e(const unsigned* u, int n, int rank, int lane) {
    unsigned x[E];
    unsigned all_or = 0u, all_and = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = lane + 64 * i;
        const bool in = e < n;
        x[i] = in ? u[e] : 0u;   // (pattern 0 is below every candidate threshold, which has at least one bit set)
        all_or |= x[i];
        all_and &= in ? x[i] : 0xffffffffu;
    }
    return kz_radix_kth_u32_regs<E>(x, all_or, all_and, rank);
}
// (the generic finalize kernel is compiled for 64 VGPRs: it keeps the LDS loops)
template <bool REGS = false>
__device__ __forceinline__ unsigned kz_radix_kth_u32(const unsigned* u, int n, int rank, int lane) {
    if (REGS && n <= 256) return kz_radix_kth_u32_e<4>(u, n, rank, lane);
    if (REGS && n <= 512) return kz_radix_kth_u32_e<8>(u, n, rank, lane);
    unsigned all_or = 0u, all_and = 0xffffffffu;
    for (int e = lane; e < n; e += 64) {
        all_or |= u[e];
        all_and &= u[e];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const unsigned differ = all_or ^ all_and;
    const int top = differ ? 31 - __clz(differ) : -1;
    unsigned thr = top >= 31 ? 0u : (top < 0 ? all_and : (all_and & ~((2u << top) - 1u)));
    for (int bit = top; bit >= 0; --bit) {
        const unsigned cand = thr | (1u << bit);
        int c = 0;
        for (int e0 = 0; e0 < n; e0 += 64) c += (int)__popcll(__ballot(e0 + lane < n && u[e0 + lane] >= cand));
        if (c >= rank) thr = cand;
    }
    return thr;
}
// rank-th SMALLEST (rank = 1: the smallest) of n non-negative doubles in LDS (their bit patterns order like the values).
template <int E>
__device__ __forceinline__ unsigned long long kz_radix_kth_small_f64_e(const double* v, int n, int rank, int lane) {
    unsigned long long x[E];
    unsigned long long all_or = 0ull, all_and = ~0ull;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = lane + 64 * i;
        const bool in = e < n;
        x[i] = in ? (unsigned long long)__double_as_longlong(v[e]) : ~0ull;   // (the largest pattern: never BELOW a candidate)
        all_or |= in ? x[i] : 0ull;
        all_and &= x[i];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const unsigned long long differ = all_or ^ all_and;
    const int top = differ ? 63 - __clzll(differ) : -1;
    unsigned long long thr = top >= 63 ? 0ull : (top < 0 ? all_and : (all_and & ~((2ull << top) - 1ull)));
    for (int bit = top; bit >= 0; --bit) {
        const unsigned long long cand = thr | (1ull << bit);
        int c = 0;   // entries below cand
#pragma unroll
        for (int i = 0; i < E; ++i) c += (int)__popcll(__ballot(x[i] < cand));
        if (c < rank) thr = cand;
    }
    return thr;
}
template <bool REGS = false>
__device__ __forceinline__ unsigned long long kz_radix_kth_small_f64(const double* v, int n, int rank, int lane) {
    if (REGS && n <= 256) return kz_radix_kth_small_f64_e<4>(v, n, rank, lane);
    unsigned long long all_or = 0ull, all_and = ~0ull;
    for (int e = lane; e < n; e += 64) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v[e]);
        all_or |= b;
        all_and &= b;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    const unsigned long long differ = all_or ^ all_and;
    const int top = differ ? 63 - __clzll(differ) : -1;
    // thr = the smallest value with at least `rank` entries <= it: build the largest prefix p such that fewer than `rank` entries are
    // BELOW p, bit by bit from the top
    unsigned long long thr = top >= 63 ? 0ull : (top < 0 ? all_and : (all_and & ~((2ull << top) - 1ull)));
    for (int bit = top; bit >= 0; --bit) {
        const unsigned long long cand = thr | (1ull << bit);
        int c = 0;   // entries below cand
        for (int e0 = 0; e0 < n; e0 += 64)
            c += (int)__popcll(__ballot(e0 + lane < n && (unsigned long long)__double_as_longlong(v[e0 + lane]) < cand));
        if (c < rank) thr = cand;
    }
    return thr;
}

// Rank-based selection of the KP best of M <= 64*E list entries (key descending, row ascending; entries with row < 0 are
// empty).  Lane l holds entries l, l+64, ...; returns the number of entries written to ck/ci (ordered by rank).
template <int E>
__device__ __forceinline__ int kz_rank_select(const float* ekey, const int* eidx, int M, int KP, float* ck, int* ci, int lane) {
    float x[E];
    int xi[E], rank[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
        const int e = lane + 64 * u;
        x[u] = e < M ? ekey[e] : -INFINITY;
        xi[u] = e < M ? eidx[e] : -1;
        rank[u] = 0;
    }
    int n_valid = 0;
#pragma unroll
    for (int v = 0; v < E; ++v) {
        n_valid += __popcll(__ballot(xi[v] >= 0));
        const int lim = min(64, M - 64 * v);
        for (int jj = 0; jj < lim; ++jj) {  // jj is wave-uniform: the broadcasts are v_readlane (spelled out: __shfl compiled to ds_bpermute)
            const int oi = __builtin_amdgcn_readlane(xi[v], jj);
            if (oi < 0) continue;   // (uniform)
            const float ox = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x[v]), jj));
#pragma unroll
            for (int u = 0; u < E; ++u)
                if (64 * u < M) rank[u] += (ox > x[u] || (ox == x[u] && oi < xi[u])) ? 1 : 0;
        }
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
        if (xi[u] >= 0 && rank[u] < KP) {
            ck[rank[u]] = x[u];
            ci[rank[u]] = xi[u];
        }
    }
    return n_valid < KP ? n_valid : KP;
}

// One query, one wave (only wave-level synchronisation inside).
// NV (round 6): 16-byte loads per lane and candidate row in the pipelined re-rank -- 1: float32 rows of up to 256 elements, 2: up to
// 512 (the second 256-element chunk's four fma continue the first chunk's chain: kz_wave_dot's order).  d = 300 -- the dimension
// of the entity-alignment embeddings kiez is used on, and of BASELINE configuration 4 -- used to take the generic loop below: no
// load in flight under the sums, the query row re-read per candidate (250 k x 1 M x 300: 7.3 ms per launch against 4.3 at d = 200).
template <typename T, int FROWS, int NV = 1>
__device__ __forceinline__ void kz_finalize_query(const KnnFinParams& p, const int64_t q, const int lane, char* wbase) {
    const int KS = p.KSEL > 0 ? p.KSEL : p.KP;   // candidates selected and re-ranked
    double* cv = reinterpret_cast<double*>(wbase);
    double* sv = cv + KS;
    float* ekey = reinterpret_cast<float*>(sv + KS);
    int* eidx = reinterpret_cast<int*>(ekey + p.max_m);
    float* ck = reinterpret_cast<float*>(eidx + p.max_m);
    int* ci = reinterpret_cast<int*>(ck + KS);
    int* si = ci + KS;
    const int KP = p.KP;
    const int k_eff = p.k + (p.exclude_self ? 1 : 0);
    // the query's own row and norm first: their latency passes under the list phase
    const int64_t qrow = p.row_map ? (int64_t)p.row_map[p.q_begin + q] : p.q_begin + q;
    const int64_t qout = p.row_map ? qrow : q;   // output row (row_map: out_dist / out_ind / fail_list are indexed by matrix rows)
    const T* qptr = reinterpret_cast<const T*>(p.qraw) + qrow * (int64_t)p.d;
    const double qs = p.qsqn[qrow];

    const int64_t lrow = p.list_row0 + q;
    const int n_pieces = p.lay.pieces[kz_list_region(lrow, p.lay)];
    const int halves = p.lay.halves;
    const int M = n_pieces * halves * KP;
    // entry e of this query: piece e / (halves KP), lane-half (e / KP) % halves, list entry e % KP  (kz_list_wave_base)
    if (p.lay.contig) {
        // fp16 kernel: the query's pieces x K' entries are one contiguous run
        const int64_t l0 = kz_list_contig_off(lrow, p.lay, KP, 0);
        for (int e = lane; e < M; e += 64) {
            ekey[e] = p.in_key[l0 + e];
            int r = p.in_idx[l0 + e];
            if (p.idx_map && r >= 0) r = p.idx_map[r];
            eidx[e] = r;
        }
    } else {
        const int64_t lwave = kz_list_wave_base(lrow, p.lay, KP, 0) + (lrow & 31);
        for (int e = lane; e < M; e += 64) {
            const int piece = e / (halves * KP);
            const int rem = e - piece * halves * KP;
            const int hh = rem / KP;
            const int ee = rem - hh * KP;
            const int64_t off = lwave + ((int64_t)piece * KP + ee) * KZ_LSTRIDE + hh * 32;
            ekey[e] = p.in_key[off];
            eidx[e] = p.in_idx[off];
        }
    }
    kz_wave_sync();

    // Long-k route (KS > KP): the union of the per-range lists holds the KS best approximate keys only if no range
    // contributes more than its list can hold.  A FULL list may have evicted rows: everything outside it has a key <= its
    // smallest entry -- the largest such value over the full lists joins the certification bound below.
    float piece_bound = p.list_floor ? p.list_floor[p.q_begin + q] : -INFINITY;
    if (KS > KP) {
        for (int l0 = 0; l0 < M; l0 += KP) {   // (uniform; KP is a multiple of 16, lists are at most 128 entries)
            float mn = INFINITY;
            int cnt = 0;
            for (int e = lane; e < KP; e += 64) {
                const bool ok = eidx[l0 + e] >= 0;
                cnt += ok ? 1 : 0;
                mn = ok ? fminf(mn, ekey[l0 + e]) : mn;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                cnt += __shfl_xor(cnt, off, 64);
                mn = fminf(mn, __shfl_xor(mn, off, 64));
            }
            if (cnt == KP) piece_bound = fmaxf(piece_bound, mn);
        }
    }
    // top-KS of the M entries by (key desc, idx asc).  Up to 64 entries: rank counting (ck / ci come out ordered).  More (round
    // 5): rank counting is O(M^2 / 64) per lane -- 160 entries (ten lists of 16): ~3 500 of a query's ~8 000 instructions -- and
    // NOTHING below needs the selected keys in order: the KS best by a radix select + compaction (unordered), further down the
    // k-th best of them by a second radix select and the candidates within 2 eps of it by compaction.
    int V = 0;
    bool unsorted = false;
    float sel_min = INFINITY, left_max = -INFINITY;   // unsorted path: smallest selected key; largest selected key NOT re-ranked
    if (M <= 64) {
        V = kz_rank_select<1>(ekey, eidx, M, KS, ck, ci, lane);
    } else {
        unsorted = true;
        unsigned* uk = reinterpret_cast<unsigned*>(ekey);   // (the keys are not needed as floats any more)
        auto key_of = [](unsigned u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); };
        int nv = 0;
        for (int e = lane; e < M; e += 64) {
            unsigned bts = __float_as_uint(ekey[e]);
            if (bts == 0x80000000u) bts = 0u;   // (-0 = +0)
            const bool valid = eidx[e] >= 0;
            uk[e] = valid ? (bts ^ ((bts >> 31) ? 0xffffffffu : 0x80000000u)) : 0u;
            nv += valid ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nv += __shfl_xor(nv, off, 64);
        kz_wave_sync();
        const bool all = nv <= KS;
        unsigned thr = 0u;
        if (!all) thr = kz_radix_kth_u32(uk, M, KS, lane);   // (invalid entries carry the smallest pattern: they never reach rank KS)
        for (int e0 = 0; e0 < M; e0 += 64) {   // entries above the threshold (all valid entries when there are at most KS)
            const int e = e0 + lane;
            const bool sel = e < M && eidx[e] >= 0 && (all || uk[e] > thr);
            const unsigned long long mask = __ballot(sel);
            if (sel) {
                const int pos = V + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                const float kf = key_of(uk[e]);
                ck[pos] = kf;
                ci[pos] = eidx[e];
                sel_min = fminf(sel_min, kf);
            }
            V += (int)__popcll(mask);
        }
        if (!all) {   // the remaining places go to the entries AT the threshold with the smallest rows
            int last = -1;
            while (V < KS) {
                int best = 0x7fffffff;
                for (int e = lane; e < M; e += 64) {
                    const int xi = eidx[e];
                    if (xi >= 0 && uk[e] == thr && xi > last && xi < best) best = xi;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
                if (best == 0x7fffffff) break;   // (cannot happen: at least KS entries are >= thr)
                if (lane == 0) {
                    ck[V] = key_of(thr);
                    ci[V] = best;
                }
                last = best;
                ++V;
                sel_min = fminf(sel_min, key_of(thr));
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sel_min = fminf(sel_min, __shfl_xor(sel_min, off, 64));
    }
    kz_wave_sync();

    // Rounding bound of this query's approximate keys and the exact key of a candidate from its exact value.
    //   float32 / split-bf16 operands: |key~ - key| <= gamma (|y|max^2 / 2 + |q| |y|max), key = (|q|^2 - d^2) / 2 (euclidean
    //   family) or 1 - dist (cosine);
    //   fp16 operands (centred vectors x_c = float32(x - mu), operands x_h, residuals r = x_c - x_h measured at pack time):
    //   q_h.y_h - q_c.y_c = -(r_q.y_h + q_h.r_y + r_q.r_y), so by Cauchy-Schwarz on the ACTUAL residual norms
    //     |key~ - key_c| <= |r_q| Yh + |q_h| Ry + |r_q| Ry              (operand rounding; Yh = max |y_h|, Ry = max |r_y|)
    //                      + gamma_acc (Yc2 / 2 + |q_h| Yh)               (float32 accumulation of exact products + bias)
    //                      + 2^-23 (|q_c| + sqrt(Yc2))^2 + 1e-12 (...) + 1e-14 (|q|^2 + |y|max^2)
    //                                                                     (float32 centring round-off, float64 re-rank)
    //   with key_c = (|q_c|^2 - d^2) / 2 and d^2 = the exact squared distance (cosine: 2 dist, rows are unit vectors).
    double eps_q, key_scale = 1.0, qref = qs;
    const bool cosine_plain = p.metric == KZ_COSINE && !p.tier_h;
    if (p.tier_h) {
        const double qc2 = p.q_rowq[qrow * 3 + 0], qh = p.q_rowq[qrow * 3 + 1], qr = p.q_rowq[qrow * 3 + 2];
        const double Yh = p.y_hmax[0], Ry = p.y_hmax[1], Yc2 = p.y_hmax[2];
        const double qc = sqrt(qc2), yc = sqrt(Yc2);
        // (the float64 re-rank evaluates |q|^2 + |y|^2 - 2 q.y on the UNCENTRED rows: its own round-off scales with those)
        const double ymax = p.ystats[0];
        const double raw2 = p.metric == KZ_COSINE ? 2.0 : qs + ymax * ymax;
        eps_q = p.eps_mult * (qr * Yh + qh * Ry + qr * Ry + p.gamma_acc * (0.5 * Yc2 + qh * Yh) +
                              1.1920928955078125e-07 * (qc + yc) * (qc + yc) + 1e-12 * (0.5 * Yc2 + qc2) + 1e-14 * raw2);
        // reverse direction of a dual pass: the key was accumulated on top of THIS row's bias (|q_c|^2 / 2 joins the
        // accumulation term) and went through one more float32 rounding when it was filed as key' = acc - bias(t) + bias(q)
        if (p.dual_col) eps_q += p.eps_mult * (p.gamma_acc * 0.5 * qc2 + 1.1920928955078125e-07 * (0.5 * Yc2 + 0.5 * qc2 + qh * Yh));
        key_scale = p.hscale[1];
        qref = qc2;
    } else if (p.metric == KZ_COSINE) {
        eps_q = p.gamma * 1.001;
    } else {
        const double ymax = p.ystats[0];
        const double scale = 0.5 * ymax * ymax + sqrt(qs) * ymax;
        eps_q = p.gamma * scale;
        // the relative bound assumes the products stay in the normal float32 range (data at the 1e-19 scale and below
        // underflows in the matrix pipe): such rows are left to the exact float64 kernels
        if (scale < 1e-30) eps_q = INFINITY;
    }
    auto exact_key = [&](double v) {
        if (cosine_plain) return 1.0 - v;
        return 0.5 * (qref - (p.metric == KZ_COSINE ? 2.0 * v : v));
    };

    // Which candidates need an exact distance?  Those that can still be among the exact top-k: a candidate c with
    // key~_c < key~_(k) - 2 eps has key_c <= key~_c + eps < key~_(k) - eps <= (k-th best exact key of the re-ranked ones),
    // so it is out.  The list is ordered by approximate key: the re-rank covers a prefix of Vr >= k_eff candidates (K' = 64,
    // k = 50: ~52 gathered rows instead of 64).  The certification below re-checks the first pruned candidate.
    int Vr = V;
    if (unsorted) {
        if (V > k_eff && eps_q < INFINITY) {
            // the k-th best selected key (radix select over the sortable patterns, scratch: sv is written after the re-rank), then
            // the candidates within 2 eps of it to the front of ekey / eidx (the list copy is spent): the re-rank's set, unordered
            unsigned* su = reinterpret_cast<unsigned*>(sv);
            for (int c = lane; c < V; c += 64) {
                unsigned b = __float_as_uint(ck[c]);
                if (b == 0x80000000u) b = 0u;
                su[c] = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
            }
            kz_wave_sync();
            const unsigned uk_k = kz_radix_kth_u32(su, V, k_eff, lane);
            const float key_k = __uint_as_float(uk_k ^ ((uk_k >> 31) ? 0x80000000u : 0xffffffffu));
            const double thr = (double)key_k * key_scale - 2.0 * eps_q;
            int cnt = 0;
            for (int c0 = 0; c0 < V; c0 += 64) {
                const int c = c0 + lane;
                const bool in = c < V && (double)ck[c] * key_scale >= thr;
                const unsigned long long mask = __ballot(in);
                if (in) {
                    const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                    ekey[pos] = ck[c];
                    eidx[pos] = ci[c];
                } else if (c < V) {
                    left_max = fmaxf(left_max, ck[c]);
                }
                cnt += (int)__popcll(mask);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) left_max = fmaxf(left_max, __shfl_xor(left_max, off, 64));
            Vr = cnt;       // (>= k_eff: the k_eff best keys are all >= key_k)
            ck = ekey;
            ci = eidx;
            kz_wave_sync();
        }
    } else if (V > k_eff && eps_q < INFINITY) {
        const double thr = (double)ck[k_eff - 1] * key_scale - 2.0 * eps_q;
        int cnt = 0;
        for (int c = lane; c < V; c += 64) cnt += ((double)ck[c] * key_scale >= thr) ? 1 : 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        Vr = cnt < k_eff ? k_eff : cnt;
    }

    // exact float64 re-rank of the Vr candidates, FROWS rows in flight per pass (independent gathers and butterfly sums
    // overlap); the per-candidate arithmetic is exactly kz_wave_dot / kz_wave_dot_normalized (kz_common.h)
    const T* yraw = reinterpret_cast<const T*>(p.yraw);
    const bool vec = kz_row_vec_ok(qptr, p.d) && kz_row_vec_ok(yraw, p.d);
#ifdef KZ_NO_FIN_PIPE
    if (false) {
#else
    if (sizeof(T) == 4 && vec && p.d <= 256 * NV && Vr > 0) {
#endif   // (Vr == 0: a reverse-direction row without a single event)
        // float32 rows of up to 256 NV elements (NV 16-byte loads per lane and row): the loads of the NEXT group of FROWS
        // candidates are issued before the current group's fma chains and butterfly sums -- same arithmetic in the same order
        // as the generic loop below (and as kz_wave_dot), only the memory latency of group g+1 hides under the sums of group g
        const int k0 = 4 * lane;
        bool act[NV];
        int k0r[NV];
        double qk[4 * NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            act[c] = k0 + 256 * c < p.d;
            k0r[c] = act[c] ? k0 + 256 * c : 0;
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            if (act[c]) {
                kz_row4(qptr, k0r[c], p.d, true, t);
                if (p.metric == KZ_COSINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] / qs;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) qk[4 * c + e] = t[e];
        }
        auto issue = [&](int c0, float4 (&buf)[FROWS][NV], double (&ysb)[FROWS]) {
#pragma unroll
            for (int u = 0; u < FROWS; ++u) {
                // (no branch around a load, and no load under a condition: with loads on some paths only the compiler waits for
                //  ALL outstanding loads -- s_waitcnt vmcnt(0), the group just issued included -- before the first use of the
                //  current group, and the prefetch hides nothing (the loop ran at latency + arithmetic per candidate).  Lanes
                //  past the end of the row read its first elements and never use them; the group past the last one is the last
                //  candidate again.)
                const int yi = ci[min(c0 + u, Vr - 1)];
                ysb[u] = p.ysqn[yi];
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    buf[u][c] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(yraw) + (int64_t)yi * p.d + k0r[c]);
            }
        };
        auto reduce = [&](int c0, const float4 (&buf)[FROWS][NV], const double (&ysb)[FROWS]) {
#pragma unroll
            for (int u = 0; u < FROWS; ++u) {
                double a = 0.0;
                bool done = false;
                if (p.metric == KZ_COSINE && p.fast_div) {
                    const double rcp = 1.0 / ysb[u];   // (wave-uniform: every lane holds the same row norm)
                    const int rcp_hi = __builtin_amdgcn_readfirstlane((int)((unsigned long long)__double_as_longlong(rcp) >> 32));
                    if ((rcp_hi & 0x7ff00000) != 0x7ff00000) {
#pragma unroll
                        for (int c = 0; c < NV; ++c) {
                            const double yk[4] = {(double)buf[u][c].x, (double)buf[u][c].y, (double)buf[u][c].z, (double)buf[u][c].w};
                            if (act[c]) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) a = fma(qk[4 * c + e], kz_div_shared(yk[e], ysb[u], rcp), a);
                            }
                        }
                        done = true;
                    }
                }
                if (!done) {
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const double yk[4] = {(double)buf[u][c].x, (double)buf[u][c].y, (double)buf[u][c].z, (double)buf[u][c].w};
                        if (act[c]) {
                            if (p.metric == KZ_COSINE) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) a = fma(qk[4 * c + e], yk[e] / ysb[u], a);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) a = fma(qk[4 * c + e], yk[e], a);
                            }
                        }
                    }
                }
                const double dot = kz_wave_sum(a);
                double v;
                if (p.metric == KZ_COSINE) {
                    v = fmin(fmax(1.0 - dot, 0.0), 2.0);
                } else {
                    v = fmax((qs + ysb[u]) - 2.0 * dot, 0.0);
                }
                if (lane == 0 && c0 + u < Vr) cv[c0 + u] = v;
            }
        };
#if KZ_FIN_DEPTH == 3
        // three groups in flight (a rotating set of three register buffers, the loop unrolled by three: no copies): the gathers
        // of groups g + 1 and g + 2 are under way while group g is reduced
        constexpr int R = FROWS;
        float4 b0[R][NV], b1[R][NV], b2[R][NV];
        double y0[R], y1[R], y2[R];
        issue(0, b0, y0);
        issue(R, b1, y1);
        for (int c0 = 0;;) {   // (all conditions wave-uniform)
            issue(c0 + 2 * R, b2, y2);
            reduce(c0, b0, y0);
            if ((c0 += R) >= Vr) break;
            issue(c0 + 2 * R, b0, y0);
            reduce(c0, b1, y1);
            if ((c0 += R) >= Vr) break;
            issue(c0 + 2 * R, b1, y1);
            reduce(c0, b2, y2);
            if ((c0 += R) >= Vr) break;
        }
#else
        float4 cur[FROWS][NV], nxt[FROWS][NV];
        double ys_c[FROWS], ys_n[FROWS];
        issue(0, cur, ys_c);
        for (int c0 = 0; c0 < Vr; c0 += 2 * FROWS) {   // (unrolled by two: the buffers swap roles, no copies)
            issue(c0 + FROWS, nxt, ys_n);
            reduce(c0, cur, ys_c);
            if (c0 + FROWS >= Vr) break;
            issue(c0 + 2 * FROWS, cur, ys_c);
            reduce(c0 + FROWS, nxt, ys_n);
        }
#endif
    } else
    for (int c0 = 0; c0 < Vr; c0 += FROWS) {
        const T* yp[FROWS];
        double ys[FROWS], acc[FROWS];
#pragma unroll
        for (int u = 0; u < FROWS; ++u) {
            const int yi = ci[min(c0 + u, Vr - 1)];
            yp[u] = yraw + (int64_t)yi * p.d;
            ys[u] = p.ysqn[yi];
            acc[u] = 0.0;
        }
        for (int k0 = 4 * lane; k0 < p.d; k0 += 256) {
            double qk[4];
            kz_row4(qptr, k0, p.d, vec, qk);
            if (p.metric == KZ_COSINE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) qk[e] = qk[e] / qs;
            }
#pragma unroll
            for (int u = 0; u < FROWS; ++u) {
                double yk[4];
                kz_row4(yp[u], k0, p.d, vec, yk);
                if (p.metric == KZ_COSINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u] = fma(qk[e], yk[e] / ys[u], acc[u]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u] = fma(qk[e], yk[e], acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FROWS; ++u) {
            const double dot = kz_wave_sum(acc[u]);
            double v;
            if (p.metric == KZ_COSINE) {
                v = fmin(fmax(1.0 - dot, 0.0), 2.0);  // sklearn cosine_distances: S *= -1; S += 1; clip(0, 2)
            } else {
                v = fmax((qs + ys[u]) - 2.0 * dot, 0.0);  // |x|^2 - 2 x.y + |y|^2, clamped (_argkmin.pyx.tp:494-502)
            }
            if (lane == 0 && c0 + u < Vr) cv[c0 + u] = v;
        }
    }
    kz_wave_sync();
    // rank by (value asc, idx asc) and scatter into sorted order
    for (int c = lane; c < Vr; c += 64) {
        const double v = cv[c];
        const int id = ci[c];
        int rank = 0;
        for (int o = 0; o < Vr; ++o) {
            const double ov = cv[o];
            const int oid = ci[o];
            rank += (ov < v || (ov == v && oid < id)) ? 1 : 0;
        }
        sv[rank] = v;
        si[rank] = id;
    }
    kz_wave_sync();

    // Self-check of the bound the certification rests on: for every candidate both the approximate key (ck, from the
    // fused kernel) and the exact key (from the float64 re-rank) are known here.
    bool bound_violated = false;
    if (eps_q > 0.0 && eps_q < INFINITY && p.err_ratio_bits) {
        double worst = 0.0;
        for (int c = lane; c < Vr; c += 64) {
            const double v = cv[c];
            if (v > 0.0)  // (a distance clamped at 0 no longer carries the exact key)
                worst = fmax(worst, fabs((double)ck[c] * key_scale - exact_key(v)) / eps_q);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off, 64));
        bound_violated = worst > 1.0;   // (wave-uniform after the butterfly) never expected: see the certification below
        if (lane == 0 && worst > 0.0) {
            // a million waves hit ONE address: read first -- an ordinary load (served by this XCD's L2; a stale value only costs
            // a redundant atomicMax), not an agent-scope atomic load that goes to the memory side every time -- and only the
            // (rare) new maxima pay for the atomic
            const unsigned long long bits = (unsigned long long)__double_as_longlong(worst);
            if (bits > *(const volatile unsigned long long*)p.err_ratio_bits) atomicMax(p.err_ratio_bits, bits);
        }
    }

    // Certification (DESIGN.md "Certified candidate sets").  |key~ - key| <= eps for every index row.  A row outside
    // the candidate set has key~ <= ck[KP-1] (the K'-th best approximate key), hence an exact key <= ck[KP-1] + eps.
    // The exact key of the k-th re-ranked candidate is known.  If it is strictly larger, no outside row can enter -- or
    // tie with -- the exact top-k.  V < KP means no list ever evicted anything: the set is complete.
    bool certified;
    if (p.excl_floor) {
        // dual pass: outside the list are events that lost the selection (key~ <= ck[KP-1], full lists only) and the rows
        // that never were events (key~ < floor)
        double bound = (double)p.excl_floor[qrow];
        if (V == KP) bound = fmax(bound, (double)(unsorted ? sel_min : ck[KP - 1]));
        certified = V >= k_eff && bound * key_scale + eps_q < exact_key(sv[k_eff - 1]);
    } else {
        // rows outside the selected set: behind the KS-th selected key (when the selection is full), or evicted from a full
        // list (long-k route: piece_bound; with KS = KP a full list implies a full selection whose KS-th key is at least as
        // large, so the first term alone is the round-1 rule).  Neither: no list ever evicted anything, the set is complete.
        float bound = piece_bound;
        if (V == KS) bound = fmaxf(bound, unsorted ? sel_min : ck[KS - 1]);
        if (bound == -INFINITY)
            certified = (V >= min((int64_t)k_eff, p.n_i));
        else
            certified = V >= k_eff && (double)bound * key_scale + eps_q < exact_key(sv[k_eff - 1]);
    }
    // ... and the candidates that were not re-ranked are out by the same argument (implied by how Vr was chosen; re-checked)
    if (Vr < V && !((double)(unsorted ? left_max : ck[Vr]) * key_scale + eps_q < exact_key(sv[k_eff - 1]))) certified = false;
    // An approximate key further than eps from its exact value contradicts the bound everything above rests on (a kernel
    // or hardware fault, not a property of the data): do not trust this row's candidate set, send it down a tier.
    if (bound_violated) certified = false;
    if (!certified) {
        if (lane == 0) {
            const int pos = atomicAdd(p.fail_count, 1);
            p.fail_list[pos] = (int)qout;
            if (p.fail_tau) p.fail_tau[pos] = Vr >= k_eff ? sv[k_eff - 1] : (double)INFINITY;
        }
        return;
    }
    kz_emit_sorted<T>(sv, si, Vr, p.k, p.exclude_self, p.self_ids ? p.self_ids[q] : qrow, p.metric,
                      p.out_dist + qout * (int64_t)p.k, p.out_ind + qout * (int64_t)p.k, lane);
}

// A workgroup finalizes KZ_FIN_QPB consecutive queries (wave w takes queries w, w+4, ...).  32 per workgroup (sharing the list
// cache lines of one wave-interleaved block) measured 2x SLOWER than 4: finalize is latency-bound and wants many workgroups.
constexpr int KZ_FIN_QPB = 4;
#ifndef KZ_FIN_WAVES_2
#define KZ_FIN_WAVES_2 5  // ... of the two-loads-per-row build (NV = 2: 24 more registers of gather buffers and query elements)
#endif
#ifndef KZ_FIN_WAVES
#define KZ_FIN_WAVES 7  // minimum waves per SIMD the finalize kernel is compiled for (see KZ_FIN_ROWS_N above)
#endif
// FROWS / MINW: candidate rows gathered per group and the occupancy compiled for.  <1, KZ_FIN_WAVES> is the kernel of every
// ordinary pass (a dozen to ~50 gathered rows per query: waves in flight beat rows in flight per wave); <8, 2> serves the long-k
// route (hundreds of gathered rows per query, one workgroup per CU for its LDS anyway: the gathers of a query were a chain of
// ~k / 2 round trips).
template <typename T, int FROWS, int MINW, int NV = 1>
__global__ __launch_bounds__(256, MINW) void kz_knn_finalize_kernel(KnnFinParams p) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63;
    // (wave number in a scalar register: the query number, its matrix row and everything loaded per query -- norm, image statistics
    //  -- are then scalar loads issued at the top of the query, not vector loads of one address by 64 lanes)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* wbase = fsm + (size_t)wave * kz_fin_wave_bytes(p.max_m, p.KSEL > 0 ? p.KSEL : p.KP);
    for (int rep = 0; rep < KZ_FIN_QPB / 4; ++rep) {
        const int64_t q = p.q_first + (int64_t)blockIdx.x * KZ_FIN_QPB + rep * 4 + wave;
        if (q >= p.q_last) break;  // whole wave leaves; only wave-level sync inside
        kz_finalize_query<T, FROWS, NV>(p, q, lane, wbase);
        kz_wave_sync();
    }
}

#include "kz_knn_fin_wide.h"

// ---------------------------------------------------------------------------------------------------
// Stage 3: exact float64 brute force for uncertified rows (rare; correctness backstop)
// ---------------------------------------------------------------------------------------------------
// SPECULATIVE launches of the exact kernels (kz_spec_rescue below): the grid is sized for `cap` rows BEFORE the host knows how
// many rows the finalize kernel left uncertified; the count is read from device memory, row b of the grid lives when
// b < count <= cap (count > cap: nothing runs here, the host takes the ordinary re-search).
__device__ __forceinline__ bool kz_spec_row_live(const int* __restrict__ dyn_n, int b, int cap) {
    const int n = *dyn_n;
    return n <= cap && b < n;
}

template <typename T>
__global__ __launch_bounds__(256) void kz_exact_dist_kernel(const int* __restrict__ fail_list, int batch0, int64_t q_begin,
                                                            const T* __restrict__ qraw, const T* __restrict__ yraw,
                                                            const double* __restrict__ qsqn, const double* __restrict__ ysqn,
                                                            int64_t n_i, int d, int metric, double p, double* __restrict__ vals,
                                                            const int* __restrict__ dyn_n = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    if (dyn_n && !kz_spec_row_live(dyn_n, b, (int)gridDim.y)) return;
    const int64_t qrow = q_begin + fail_list[batch0 + b];
    // (grid-stride over the index rows: the ordinary callers launch one wave per pair, a speculative launch a bounded grid --
    //  workgroups of a dead row cost their dispatch, and n_i / 4 x R of them would be milliseconds on a 1 M-row index)
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < n_i; i += (int64_t)gridDim.x * 4) {
        const double v = kz_exact_value<T>(qraw + qrow * (int64_t)d, yraw + i * (int64_t)d, qsqn[qrow], ysqn[i], d, metric, lane, p);
        if (lane == 0) vals[(int64_t)b * n_i + i] = v;
    }
}

// The same values for float32 rows of d <= 256 (d a multiple of 4), many pairs per wave step (round 5).  kz_exact_dist_kernel spends
// a wave on ONE pair -- at d = 64 a quarter of its lanes, re-reading the query row and, for cosine, dividing every element twice:
// 2.9 G pairs/s, 125 us per query row against 301 k index rows; on data with clusters three orders of magnitude tighter than the
// data's extent a tenth of the rows end there, and a 300 k x 300 k call took 8 s (tools/cliff_probe.py).  Here a wave keeps Q = 4
// query rows in registers and walks CONSECUTIVE index rows, G = 64 / LPR of them per step (a row needs LPR = d / 4 lanes rounded up
// to a power of two): one coalesced load serves G x Q pairs.  The arithmetic of a pair is kz_wave_dot's, operation for operation
// -- the lane's four fma in element order, then the butterfly inside the lane group (the steps of the full-wave butterfly that it
// skips add the exact zeros of lanes past the row) -- as in the finalize kernel for many candidates (kz_knn_fin_wide.h), so the
// values are bit for bit those of kz_exact_value, kz_pair_values and the re-rank.  Cosine: the index rows normalised once in
// float64 (kz_matrix_norm64) where that image exists, else the shared-reciprocal division.
template <int LPR, bool NORM, int NV = 1>
__global__ __launch_bounds__(256) void kz_exact_dist_rows_kernel(const int* __restrict__ fail_list, int batch0, int nb, int64_t q_begin,
                                                                 const float* __restrict__ qraw, const float* __restrict__ yraw,
                                                                 const double* __restrict__ ynorm64, const double* __restrict__ qsqn,
                                                                 const double* __restrict__ ysqn, int64_t n_i, int d, int metric,
                                                                 int rows_per_wave, double* __restrict__ vals,
                                                                 const int* __restrict__ dyn_n = nullptr) {
    // NV = 2 (round 6): rows of 260 .. 512 elements -- a lane owns elements 4 sl .. 4 sl + 3 of BOTH 256-element chunks of the row
    // (LPR = 64, one index row per wave step), the second chunk's four fma continue the first's chain: kz_wave_dot's order for d > 256.
    static_assert(NV == 1 || LPR == 64, "two chunks per lane: the whole wave owns one row");
    constexpr int G = 64 / LPR, Q = 4;
    if (dyn_n) {   // (speculative launch: nb was the grid's capacity)
        if (!kz_spec_row_live(dyn_n, blockIdx.y * Q, nb)) return;
        nb = *dyn_n;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPR, sl = lane & (LPR - 1);
    const int k0 = 4 * sl;
    bool act[NV];
    int k0r[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        act[c] = k0 + 256 * c < d;
        k0r[c] = act[c] ? k0 + 256 * c : 0;
    }
    const int b0 = blockIdx.y * Q;
    double qk[Q][4 * NV], qs[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int bq = b0 + j < nb ? b0 + j : nb - 1;
        const int64_t qrow = q_begin + fail_list[batch0 + bq];
        qs[j] = qsqn[qrow];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            if (act[c]) {
                kz_row4(qraw + qrow * (int64_t)d, k0r[c], d, true, t);
                if (metric == KZ_COSINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] / qs[j];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) qk[j][4 * c + e] = t[e];
        }
    }
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * rows_per_wave;
    const int64_t i1 = i0 + rows_per_wave < n_i ? i0 + rows_per_wave : n_i;
    if (i0 >= i1) return;
    struct Buf {
        float4 f[NV];
        double ys;
        double2 n0[NV], n1[NV];
    };
    auto issue = [&](int64_t i, Buf& b) {   // (rows past the end: the last row again, nothing is written for them)
        const int64_t yi = i + grp < i1 ? i + grp : i1 - 1;
        if (NORM) {
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                const double* row = ynorm64 + yi * (int64_t)d + k0r[c];
                b.n0[c] = *reinterpret_cast<const double2*>(row);
                b.n1[c] = *reinterpret_cast<const double2*>(row + 2);
            }
        } else {
            b.ys = ysqn[yi];
#pragma unroll
            for (int c = 0; c < NV; ++c) b.f[c] = *reinterpret_cast<const float4*>(yraw + yi * (int64_t)d + k0r[c]);
        }
    };
    auto reduce = [&](int64_t i, const Buf& b) {
        double yv[4 * NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[4 * c + e] = 0.0;
            if (act[c]) {
                if (NORM) {
                    yv[4 * c] = b.n0[c].x, yv[4 * c + 1] = b.n0[c].y, yv[4 * c + 2] = b.n1[c].x, yv[4 * c + 3] = b.n1[c].y;
                } else {
                    const double yk[4] = {(double)b.f[c].x, (double)b.f[c].y, (double)b.f[c].z, (double)b.f[c].w};
                    if (metric == KZ_COSINE) {
                        const double rcp = 1.0 / b.ys;
                        const bool fin = (((unsigned long long)__double_as_longlong(rcp) >> 52) & 0x7ff) != 0x7ff;
#pragma unroll
                        for (int e = 0; e < 4; ++e) yv[4 * c + e] = fin ? kz_div_shared(yk[e], b.ys, rcp) : yk[e] / b.ys;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) yv[4 * c + e] = yk[e];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            double a = 0.0;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                if (act[c]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fma(qk[j][4 * c + e], yv[4 * c + e], a);
                }
            }
#pragma unroll
            for (int off = LPR >> 1; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
            double v;
            if (metric == KZ_COSINE)
                v = fmin(fmax(1.0 - a, 0.0), 2.0);
            else
                v = fmax((qs[j] + b.ys) - 2.0 * a, 0.0);
            if (sl == 0 && i + grp < i1 && b0 + j < nb) vals[(int64_t)(b0 + j) * n_i + i + grp] = v;
        }
    };
    Buf ba, bb;
    issue(i0, ba);
    for (int64_t i = i0; i < i1;) {   // (two steps in flight; the conditions are wave-uniform)
        issue(i + G, bb);
        reduce(i, ba);
        i += G;
        if (i >= i1) break;
        issue(i + G, ba);
        reduce(i, bb);
        i += G;
    }
}
// -> true when the kernel above took the batch
static bool kz_launch_exact_rows(kz_ctx* ctx, const int* fl, int b0, int nb, int64_t cq_begin, const kz_matrix* query, const kz_matrix* index,
                                 int metric, double* vals, const int* dyn_n = nullptr, int rows_per_wave = 256) {
    const int d = (int)index->d;
    if (index->dtype != KZ_F32 || (d & 3) != 0 || d > 512 || metric > KZ_COSINE || (((uintptr_t)query->raw | (uintptr_t)index->raw) & 15u) != 0) return false;
    const bool norm = metric == KZ_COSINE && index->norm64 != nullptr;
    const dim3 grid((unsigned)((index->n + 4 * rows_per_wave - 1) / (4 * rows_per_wave)), (unsigned)((nb + 3) / 4));
    const int lanes = (d + 3) >> 2;
#define KZ_EXACT_ROWS(L)                                                                                                                \
    do {                                                                                                                                \
        if (norm)                                                                                                                       \
            hipLaunchKernelGGL((kz_exact_dist_rows_kernel<L, true>), grid, dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin,          \
                               (const float*)query->raw, (const float*)index->raw, index->norm64, query->sqn, index->sqn, index->n, d, \
                               metric, rows_per_wave, vals, dyn_n);                                                                     \
        else                                                                                                                            \
            hipLaunchKernelGGL((kz_exact_dist_rows_kernel<L, false>), grid, dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin,         \
                               (const float*)query->raw, (const float*)index->raw, (const double*)nullptr, query->sqn, index->sqn,      \
                               index->n, d, metric, rows_per_wave, vals, dyn_n);                                                        \
    } while (0)
    if (lanes <= 8)
        KZ_EXACT_ROWS(8);
    else if (lanes <= 16)
        KZ_EXACT_ROWS(16);
    else if (lanes <= 32)
        KZ_EXACT_ROWS(32);
    else if (lanes <= 64)
        KZ_EXACT_ROWS(64);
    else if (norm)   // (260 .. 512 elements: two chunks per lane)
        hipLaunchKernelGGL((kz_exact_dist_rows_kernel<64, true, 2>), grid, dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin, (const float*)query->raw,
                           (const float*)index->raw, index->norm64, query->sqn, index->sqn, index->n, d, metric, rows_per_wave, vals, dyn_n);
    else
        hipLaunchKernelGGL((kz_exact_dist_rows_kernel<64, false, 2>), grid, dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin, (const float*)query->raw,
                           (const float*)index->raw, (const double*)nullptr, query->sqn, index->sqn, index->n, d, metric, rows_per_wave, vals, dyn_n);
#undef KZ_EXACT_ROWS
    return true;
}

#include "kz_exact_lanes.h"
// -> true when the one-pair-per-lane kernel (kz_exact_lanes.h) took the batch: float32 rows of up to 512 elements (a multiple of 4,
// 16-byte aligned), the euclidean family on the raw rows, cosine on the normalised float64 rows where that image exists, and a
// batch of at least KZ_XL_MIN_ROWS query rows (a handful is the cooperative kernel's: it needs no staging and no pre-pass).
constexpr int KZ_XL_MIN_ROWS = 32;
static inline size_t kz_exact_lanes_qd_bytes(int nb, int d) {   // float64 operand rows + squared norms of whole blocks of query rows
    const size_t nb_pad = (size_t)(nb + 4 * KZ_XL_Q - 1) / (4 * KZ_XL_Q) * (4 * KZ_XL_Q);
    return (nb_pad * (size_t)d + nb_pad) * 8;
}
// dyn_n (speculative launch): nb is the capacity of the launch, the row count is read on the device
// groups != nullptr (kz_range.h, grouped ranges): ONE launch for n_groups dense blocks (KzXlGroup) -- fl [n_slots] then holds the
// query row of every operand row of every block (-1: padding), gather the blocks' lists of index rows; nb = n_slots, rows_max /
// q_max = the largest block's index rows / query rows; vals as the blocks' val_off say.
static int kz_launch_exact_lanes(kz_ctx* ctx, const int* fl, int b0, int nb, int64_t cq_begin, const kz_matrix* query, const kz_matrix* index,
                                 int metric, double* vals, bool* took, const int* dyn_n = nullptr, double* qd_buf = nullptr,
                                 const int* gather = nullptr, const KzXlGroup* groups = nullptr, int n_groups = 0, int rows_max = 0,
                                 int q_max = 0) {
    *took = false;
    const int d = (int)index->d;
    if (ctx->exact_rows < 2 || (nb < KZ_XL_MIN_ROWS && !dyn_n) || index->dtype != KZ_F32 || (d & 3) != 0 || d > 512 || metric > KZ_COSINE ||
        (((uintptr_t)query->raw | (uintptr_t)index->raw) & 15u) != 0)
        return KZ_OK;
    const bool cosine = metric == KZ_COSINE && index->norm64 != nullptr && d <= 256;   // (the normalised float64 rows, staged as they are)
    const bool cos_raw = metric == KZ_COSINE && !cosine;                               // (the raw rows, divided by the lane)
    const int d_pad = d;   // (a multiple of 4: whole leaves)
    const int nb_pad = groups ? nb : (nb + 4 * KZ_XL_Q - 1) / (4 * KZ_XL_Q) * (4 * KZ_XL_Q);   // (whole blocks of 4 waves x KZ_XL_Q rows; groups: the slots are padded per block)
    double* qd = qd_buf;   // (a caller that runs these launches on another stream than the pool's brings the buffer: kz_spec_alloc)
    if (!qd) {
        const int rc = kz_pool_alloc(ctx, kz_exact_lanes_qd_bytes(nb, d), (void**)&qd);
        if (rc != KZ_OK) return rc == KZ_ERR_NOMEM ? KZ_OK : rc;   // (no memory for the operand rows: the cooperative kernel)
    }
    double* qsq = qd + (size_t)nb_pad * d_pad;
    if (groups)
        hipLaunchKernelGGL(kz_exact_qprep_slots_kernel, dim3(nb_pad), dim3(256), 0, ctx->stream, fl, cq_begin, (const float*)query->raw, query->sqn, d,
                           d_pad, metric, qd, qsq);
    else
        hipLaunchKernelGGL(kz_exact_qprep_kernel, dim3(nb_pad), dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin, (const float*)query->raw, query->sqn, d,
                           d_pad, metric, qd, qsq, dyn_n);
    const int64_t n_rows = groups ? rows_max : index->n;
    // (groups: a workgroup takes q_chunk = 64 query rows of its block -- four rounds of its 16 -- for one tile of 64 index rows)
    const int q_chunk = 16 * KZ_XL_Q;
    const dim3 grid((unsigned)((n_rows + KZ_XL_ROWS - 1) / KZ_XL_ROWS), groups ? (unsigned)((q_max + q_chunk - 1) / q_chunk) : 1u,
                    groups ? (unsigned)n_groups : 1u);
    const size_t lds = (size_t)(d_pad / 4) * (KZ_XL_ROWS + 1) * (cosine ? 32 : 16);   // (<= 133 KiB: 512 float32 / 256 float64 elements)
    hipError_t e = hipSuccess;
#define KZ_XL_LAUNCH_G(NL, NVV, ELT, CR, GA, rows)                                                                                             \
    do {                                                                                                                                        \
        if (lds > 65536) e = hipFuncSetAttribute((const void*)kz_exact_dist_lanes_kernel<NL, NVV, ELT, CR, GA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e == hipSuccess)                                                                                                                     \
            hipLaunchKernelGGL((kz_exact_dist_lanes_kernel<NL, NVV, ELT, CR, GA>), grid, dim3(256), lds, ctx->stream, nb, (const double*)qd, (const double*)qsq, \
                               (const ELT*)(rows), index->sqn, n_rows, d, d_pad, metric, vals, dyn_n, gather, groups, q_chunk);                 \
    } while (0)
#define KZ_XL_LAUNCH(NL, NVV, ELT, rows)                          \
    do {                                                           \
        if (groups)                                                \
            KZ_XL_LAUNCH_G(NL, NVV, ELT, false, true, rows);       \
        else                                                       \
            KZ_XL_LAUNCH_G(NL, NVV, ELT, false, false, rows);      \
    } while (0)
    if (cos_raw) {
#define KZ_XL_LAUNCH_COS(NL, NVV)                                        \
    do {                                                                  \
        if (groups)                                                       \
            KZ_XL_LAUNCH_G(NL, NVV, float, true, true, index->raw);       \
        else                                                              \
            KZ_XL_LAUNCH_G(NL, NVV, float, true, false, index->raw);      \
    } while (0)
        if (d <= 64)
            KZ_XL_LAUNCH_COS(16, 1);
        else if (d <= 128)
            KZ_XL_LAUNCH_COS(32, 1);
        else if (d <= 256)
            KZ_XL_LAUNCH_COS(64, 1);
        else
            KZ_XL_LAUNCH_COS(64, 2);
#undef KZ_XL_LAUNCH_COS
    } else if (cosine) {
        if (d <= 64)
            KZ_XL_LAUNCH(16, 1, double, index->norm64);
        else if (d <= 128)
            KZ_XL_LAUNCH(32, 1, double, index->norm64);
        else
            KZ_XL_LAUNCH(64, 1, double, index->norm64);
    } else {
        if (d <= 64)
            KZ_XL_LAUNCH(16, 1, float, index->raw);
        else if (d <= 128)
            KZ_XL_LAUNCH(32, 1, float, index->raw);
        else if (d <= 256)
            KZ_XL_LAUNCH(64, 1, float, index->raw);
        else
            KZ_XL_LAUNCH(64, 2, float, index->raw);
    }
#undef KZ_XL_LAUNCH
#undef KZ_XL_LAUNCH_G
    if (e == hipSuccess) e = hipGetLastError();
    if (!qd_buf) kz_pool_free(ctx, qd, 0);   // (stream-ordered pool)
    if (e != hipSuccess) {
        kz_set_error("kz_knn: exact distance kernel (one pair per lane) failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    *took = true;
    return KZ_OK;
}

// The Minkowski family beyond p = 2 (KZ_MANHATTAN, KZ_CHEBYSHEV, KZ_MINKOWSKI): no inner-product form, hence no MFMA -- a
// register-tiled VALU kernel.  A workgroup of 256 threads owns 64 queries x 64 index rows, a thread 4 x 4 pairs; the rows are
// staged through LDS DK features at a time, transposed ([feature][row]: a thread reads its four query values and its four index
// values of a feature as one 16- / 32-byte LDS read each).  Per pair and feature: subtract in the input dtype, |.| into float64
// (the conversion carries the abs modifier), add -- three VALU operations; every thread adds the terms of its pairs in feature
// order (kz_common.h: kz_family_term / kz_family_add), which is scikit-learn's order.  VALU-bound: 15 k x 15 k x 300 float32,
// manhattan: see DESIGN section 9.  Output: the same [batch][n_i] float64 value matrix kz_exact_dist_kernel writes.
template <typename T, int METRIC, int DK, int CHAIN>
__global__ __launch_bounds__(256) void kz_family_dist_kernel(const int* __restrict__ fail_list, int batch0, int nb, int64_t q_begin,
                                                             const T* __restrict__ qraw, const T* __restrict__ yraw, int64_t n_i, int d,
                                                             double p, int p_int, double* __restrict__ vals) {
    __shared__ __attribute__((aligned(32))) T sQ[DK][64];
    __shared__ __attribute__((aligned(32))) T sY[DK][64];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int64_t y0 = (int64_t)blockIdx.x * 64;
    const int b0 = blockIdx.y * 64;
    // staging: thread t copies DK / 4 consecutive features of row (t & 63) of both tiles (rows past the end: the last row again)
    const int lrow = t & 63, lseg = (t >> 6) * (DK / 4);
    const int bq = b0 + lrow < nb ? b0 + lrow : nb - 1;
    const T* __restrict__ qp = qraw + (q_begin + fail_list[batch0 + bq]) * (int64_t)d;
    const T* __restrict__ yp = yraw + (y0 + lrow < n_i ? y0 + lrow : n_i - 1) * (int64_t)d;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
    for (int k0 = 0; k0 < d; k0 += DK) {
        T rq[DK / 4], ry[DK / 4];
#pragma unroll
        for (int u = 0; u < DK / 4; ++u) {
            const int k = k0 + lseg + u;
            rq[u] = k < d ? qp[k] : (T)0;
            ry[u] = k < d ? yp[k] : (T)0;   // (|0 - 0| = 0 changes no sum and no maximum)
        }
        __syncthreads();   // (the previous chunk has been read)
#pragma unroll
        for (int u = 0; u < DK / 4; ++u) {
            sQ[lseg + u][lrow] = rq[u];
            sY[lseg + u][lrow] = ry[u];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < DK; ++j) {
            T q4[4], y4[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                q4[a] = sQ[j][ty * 4 + a];
                y4[a] = sY[j][tx * 4 + a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = kz_family_add<METRIC>(acc[a][c], kz_family_term<T, METRIC, CHAIN>(q4[a], y4[c], p, p_int));
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int b = b0 + ty * 4 + a;
        if (b >= nb) continue;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t i = y0 + tx * 4 + c;
            if (i < n_i) vals[(int64_t)b * n_i + i] = sizeof(T) == 4 ? (double)(float)acc[a][c] : acc[a][c];
        }
    }
}
template <typename T>
static void kz_launch_family_dist(kz_ctx* ctx, const int* fl, int b0, int nb, int64_t cq_begin, const kz_matrix* query, const kz_matrix* index,
                                  double* vals) {
    constexpr int DK = sizeof(T) == 4 ? 32 : 16;
    const dim3 grid((unsigned)((index->n + 63) / 64), (unsigned)((nb + 63) / 64));
    const int p_int = kz_family_p_int(index->metric, index->mink_p, sizeof(T) == 4);
#define KZ_FAMILY_LAUNCH(M, C)                                                                                                          \
    hipLaunchKernelGGL((kz_family_dist_kernel<T, M, DK, C>), grid, dim3(256), 0, ctx->stream, fl, b0, nb, cq_begin, (const T*)query->raw, \
                       (const T*)index->raw, index->n, (int)index->d, index->mink_p, p_int, vals)
    if (index->metric == KZ_MANHATTAN)
        KZ_FAMILY_LAUNCH(KZ_MANHATTAN, -1);
    else if (index->metric == KZ_CHEBYSHEV)
        KZ_FAMILY_LAUNCH(KZ_CHEBYSHEV, -1);
    else if (p_int == 3)
        KZ_FAMILY_LAUNCH(KZ_MINKOWSKI, 3);     // (float32 inputs, p = 3 or 4: a product with one rounding, no pow() in the kernel)
    else if (p_int == 4)
        KZ_FAMILY_LAUNCH(KZ_MINKOWSKI, 4);
    else
        KZ_FAMILY_LAUNCH(KZ_MINKOWSKI, -1);
#undef KZ_FAMILY_LAUNCH
}

// First level of the exact selection on a long row: the k_eff smallest (value, index row) pairs of every CHUNK of KZ_EXACT_CHUNK
// values (the smallest k_eff of the row are among the smallest k_eff of their chunks); kz_exact_select_kernel then picks from
// n_chunks x k_eff survivors instead of passing k_eff times over the whole row with one workgroup (1 M index rows, k = 10: 2 ms
// per query row before, the distance kernel's time now).  One workgroup per (chunk, query row); a thread holds 16 values.
constexpr int KZ_EXACT_CHUNK = 4096;
__global__ __launch_bounds__(256) void kz_exact_chunk_kernel(const double* __restrict__ vals, int64_t n_i, int k_eff, int n_chunks,
                                                             double* __restrict__ cand_v, int* __restrict__ cand_i,
                                                             const int* __restrict__ dyn_n = nullptr) {
    __shared__ double s_v[4];
    __shared__ int s_i[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x, b = blockIdx.y;
    if (dyn_n && !kz_spec_row_live(dyn_n, b, (int)gridDim.y)) return;
    const double* v = vals + (int64_t)b * n_i;
    const int64_t i0 = (int64_t)c * KZ_EXACT_CHUNK;
    constexpr int PER = KZ_EXACT_CHUNK / 256;
    double x[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = i0 + tid + 256 * u;
        x[u] = i < n_i ? v[i] : INFINITY;
    }
    double* ov = cand_v + ((int64_t)b * n_chunks + c) * k_eff;
    int* oi = cand_i + ((int64_t)b * n_chunks + c) * k_eff;
    double pv = -1.0;  // values are >= 0
    int pi = -1;
    for (int r = 0; r < k_eff; ++r) {
        double bv = INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = i0 + tid + 256 * u;
            const int id = i < n_i ? (int)i : 0x7fffffff;   // (places past the end of the row: (+inf, INT_MAX), after every real entry)
            const bool after = (x[u] > pv) || (x[u] == pv && id > pi);
            if (after && (x[u] < bv || (x[u] == bv && id < bi))) {
                bv = x[u];
                bi = id;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o_v = __shfl_xor(bv, off, 64);
            const int o_i = __shfl_xor(bi, off, 64);
            if (o_v < bv || (o_v == bv && o_i < bi)) {
                bv = o_v;
                bi = o_i;
            }
        }
        if (lane == 0) {
            s_v[wave] = bv;
            s_i[wave] = bi;
        }
        __syncthreads();
        bv = s_v[0];
        bi = s_i[0];
        for (int ww = 1; ww < 4; ++ww) {
            if (s_v[ww] < bv || (s_v[ww] == bv && s_i[ww] < bi)) {
                bv = s_v[ww];
                bi = s_i[ww];
            }
        }
        if (tid == 0) {
            ov[r] = bv;
            oi[r] = bi;
        }
        pv = bv;
        pi = bi;
        __syncthreads();
    }
}

// The same first level for MANY neighbours (k_eff >= 24): the k_eff-th smallest value of the chunk by a workgroup-wide radix
// selection on the float64 bit patterns (non-negative doubles order like their patterns; a thread holds 16 of them, a counting pass
// is 16 compares, a wave sum and one exchange through LDS -- ~55 passes below the common prefix whatever k is, against k_eff rounds
// of a workgroup-wide arg-min: k = 50: 0.96 -> see r05_notes), then everything below it and, of the entries equal to it, those with
// the smallest index rows.  The survivors come out in no particular order: kz_exact_select_kernel orders by (value, index row).
__global__ __launch_bounds__(256) void kz_exact_chunk_radix_kernel(const double* __restrict__ vals, int64_t n_i, int k_eff, int n_chunks,
                                                                   double* __restrict__ cand_v, int* __restrict__ cand_i,
                                                                   const int* __restrict__ dyn_n = nullptr) {
    __shared__ unsigned long long s_or[4], s_and[4];
    __shared__ int s_cnt[4];
    __shared__ int s_pos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x, b = blockIdx.y;
    if (dyn_n && !kz_spec_row_live(dyn_n, b, (int)gridDim.y)) return;
    const double* v = vals + (int64_t)b * n_i;
    const int64_t i0 = (int64_t)c * KZ_EXACT_CHUNK;
    constexpr int PER = KZ_EXACT_CHUNK / 256;
    const int nvalid = (int)(n_i - i0 < KZ_EXACT_CHUNK ? n_i - i0 : KZ_EXACT_CHUNK);
    double* ov = cand_v + ((int64_t)b * n_chunks + c) * k_eff;
    int* oi = cand_i + ((int64_t)b * n_chunks + c) * k_eff;
    unsigned long long x[PER];
    unsigned long long all_or = 0ull, all_and = ~0ull;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + 256 * u;
        const bool in = e < nvalid;
        x[u] = in ? (unsigned long long)__double_as_longlong(v[i0 + e]) : ~0ull;   // (places past the end: above every value)
        all_or |= in ? x[u] : 0ull;
        all_and &= x[u];
    }
    if (nvalid <= k_eff) {   // (a short last chunk: every entry survives; the unused places hold (+inf, INT_MAX))
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int e = tid + 256 * u;
            if (e < nvalid) {
                ov[e] = __longlong_as_double((long long)x[u]);
                oi[e] = (int)(i0 + e);
            }
        }
        for (int e = nvalid + tid; e < k_eff; e += 256) {
            ov[e] = INFINITY;
            oi[e] = 0x7fffffff;
        }
        return;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    if (lane == 0) {
        s_or[wave] = all_or;
        s_and[wave] = all_and;
    }
    if (tid == 0) s_pos = 0;
    __syncthreads();
    all_or = s_or[0] | s_or[1] | s_or[2] | s_or[3];
    all_and = s_and[0] & s_and[1] & s_and[2] & s_and[3];
    auto block_sum = [&](int cnt) {   // (every thread gets the workgroup's total)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        __syncthreads();   // (the previous round's readers are done with s_cnt)
        if (lane == 0) s_cnt[wave] = cnt;
        __syncthreads();
        return s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    };
    const unsigned long long differ = all_or ^ all_and;
    const int top = differ ? 63 - __clzll(differ) : -1;
    // thr = the k_eff-th smallest pattern: the largest prefix with fewer than k_eff entries below it, bit by bit
    unsigned long long thr = top >= 63 ? 0ull : (top < 0 ? all_and : (all_and & ~((2ull << top) - 1ull)));
    for (int bit = top; bit >= 0; --bit) {
        const unsigned long long cand = thr | (1ull << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) cnt += x[u] < cand ? 1 : 0;
        if (block_sum(cnt) < k_eff) thr = cand;
    }
    // everything below thr
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (x[u] < thr) {
            const int pos = atomicAdd(&s_pos, 1);
            ov[pos] = __longlong_as_double((long long)x[u]);
            oi[pos] = (int)(i0 + tid + 256 * u);
        }
    }
    int ties = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) ties += x[u] == thr ? 1 : 0;
    const int T = block_sum(ties);   // (its barriers also publish s_pos)
    const int L = s_pos;
    const int m = k_eff - L;         // places left for entries equal to thr: 1 <= m <= T
    if (T == m) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (x[u] == thr) {
                const int pos = atomicAdd(&s_pos, 1);
                ov[pos] = __longlong_as_double((long long)thr);
                oi[pos] = (int)(i0 + tid + 256 * u);
            }
        }
        return;
    }
    // more ties at the k_eff-th place than places: those with the smallest index rows, one per round
    int last = -1;
    for (int r = 0; r < m; ++r) {
        int best = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = (int)(i0 + tid + 256 * u);
            if (x[u] == thr && id > last && id < best) best = id;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
        __syncthreads();
        if (lane == 0) s_cnt[wave] = best;
        __syncthreads();
        best = min(min(s_cnt[0], s_cnt[1]), min(s_cnt[2], s_cnt[3]));
        if (tid == 0) {
            ov[L + r] = __longlong_as_double((long long)thr);
            oi[L + r] = best;
        }
        last = best;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void kz_exact_select_kernel(const int* __restrict__ fail_list, int batch0, int64_t q_begin,
                                                              const double* __restrict__ vals, const int* __restrict__ cand_idx,
                                                              int64_t n_entries, int64_t n_i, int k,
                                                              int exclude_self, const int64_t* __restrict__ self_ids,
                                                              int metric, double p, double* __restrict__ out_dist,
                                                              int64_t* __restrict__ out_ind, const int* __restrict__ dyn_n = nullptr,
                                                              const long long* __restrict__ seg_off = nullptr, int* __restrict__ left = nullptr,
                                                              int* __restrict__ left_cnt = nullptr, const long long* __restrict__ idx_off = nullptr,
                                                              const int* __restrict__ seg_len = nullptr) {
    __shared__ double s_v[4];
    __shared__ int s_i[4];
    extern __shared__ __attribute__((aligned(16))) char sel_sm[];   // k_eff doubles + k_eff ints (any k the host admits)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    if (dyn_n && !kz_spec_row_live(dyn_n, b, (int)gridDim.x)) return;
    const int q = fail_list[batch0 + b];
    if (q < 0) return;   // (grouped ranges, kz_range.h: a padding slot of a block)
    // the row's values: all n_i of them (cand_idx == nullptr: entry i is index row i), or the survivors of kz_exact_chunk_kernel
    // (n_entries (value, index row) pairs; unused places hold (+inf, INT_MAX) and are never reached: k_eff <= n_i real entries exist)
    const double* v = vals + (int64_t)b * n_entries;
    const int* vid = cand_idx ? cand_idx + (int64_t)b * n_entries : nullptr;
    const int k_eff = (int)min((int64_t)(k + (exclude_self ? 1 : 0)), n_i);
    if (seg_off) {
        // range re-search (kz_range.h): row b's entries are the segment [seg_off[b], seg_off[b + 1]) of vals / cand_idx; a segment
        // with fewer than k entries cannot answer its row -- the row is handed back (left)
        const long long s0 = seg_off[b];
        n_entries = seg_len ? (int64_t)seg_len[b] : seg_off[b + 1] - s0;   // (seg_len: the segments are not adjacent)
        if (n_entries < k_eff) {
            if (tid == 0) left[atomicAdd(left_cnt, 1)] = q;
            return;
        }
        v = vals + s0;
        vid = cand_idx + (idx_off ? idx_off[b] : s0);   // (idx_off: the rows of a group share one list of index rows)
    }
    // LONG SEGMENTS (a group's range: thousands of values per row, kz_range.h): k passes over all of them -- 82 k rows x 10 x 5 000
    // loads, 4 ms of a 79 ms search -- become two.  Pass 1: every thread's smallest value; the k-th smallest T of those 256 minima is
    // at or above the k-th smallest value of the segment.  Pass 2: the entries <= T (all ties included) go to a list in LDS; the k
    // rounds below then run over that list.  The k smallest by (value, row) all lie at or below T: the same selection.  A list
    // that would not fit (values dense at the bottom, duplicates) leaves the segment where it is.
    constexpr int SEL_CAP = 1536;
    __shared__ double c_v[SEL_CAP];
    __shared__ int c_i[SEL_CAP];
    __shared__ double s_min[256];
    __shared__ double s_T;
    __shared__ int s_cnt;
    if (seg_off && n_entries >= 2048 && k_eff <= 256) {   // (uniform; every thread then owns >= 8 entries)
        double m = INFINITY;
        for (int64_t i = tid; i < n_entries; i += 256) m = fmin(m, v[i]);
        s_min[tid] = m;
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int rank = 0;
        for (int o = 0; o < 256; ++o) {
            const double om = s_min[o];
            rank += (om < m || (om == m && o < tid)) ? 1 : 0;
        }
        if (rank == k_eff - 1) s_T = m;
        __syncthreads();
        const double Tv = s_T;
        for (int64_t i = tid; i < n_entries; i += 256) {
            const double x = v[i];
            if (x <= Tv) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < SEL_CAP) {
                    c_v[pos] = x;
                    c_i[pos] = vid ? vid[i] : (int)i;
                }
            }
        }
        __syncthreads();
        if (s_cnt <= SEL_CAP) {   // (uniform)
            v = c_v;
            vid = c_i;
            n_entries = s_cnt;
        }
    }
    double* s_sv = reinterpret_cast<double*>(sel_sm);
    int* s_si = reinterpret_cast<int*>(s_sv + k_eff);
    double pv = -1.0;  // values are >= 0
    int pi = -1;
    for (int r = 0; r < k_eff; ++r) {
        double bv = INFINITY;
        int bi = 0x7fffffff;
        for (int64_t i = tid; i < n_entries; i += 256) {
            const double x = v[i];
            const int id = vid ? vid[i] : (int)i;
            const bool after = (x > pv) || (x == pv && id > pi);
            if (after && (x < bv || (x == bv && id < bi))) {
                bv = x;
                bi = id;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov < bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_v[wave] = bv;
            s_i[wave] = bi;
        }
        __syncthreads();
        bv = s_v[0];
        bi = s_i[0];
        for (int ww = 1; ww < 4; ++ww) {
            if (s_v[ww] < bv || (s_v[ww] == bv && s_i[ww] < bi)) {
                bv = s_v[ww];
                bi = s_i[ww];
            }
        }
        if (tid == 0) {
            s_sv[r] = bv;
            s_si[r] = bi;
        }
        pv = bv;
        pi = bi;
        __syncthreads();
    }
    if (wave == 0)
        kz_emit_sorted<T>(s_sv, s_si, k_eff, k, exclude_self, self_ids ? self_ids[q] : q_begin + q, metric,
                          out_dist + (int64_t)q * k, out_ind + (int64_t)q * k, lane, p);
}

// ---------------------------------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------------------------------
static int kz_pick_list_len(int k_eff) {
    if (k_eff <= 12) return 16;
    if (k_eff <= 26) return 32;
    if (k_eff <= 54) return 64;
    if (k_eff <= 110) return 128;
    return 0;
}

template <int KP>
static int kz_cand_occupancy(int* blocks_per_cu) {
    auto kern = kz_knn_cand_kernel<KP, 0>;
    int nb = 0;
    KZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, KZ_CAND_LDS));
    *blocks_per_cu = nb < 1 ? 1 : nb;
    return KZ_OK;
}

template <int KP>
static int kz_launch_cand(kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    hipLaunchKernelGGL((kz_knn_cand_kernel<KP, 0>), dim3(n_blocks), dim3(256), KZ_CAND_LDS, ctx->stream, p);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

#define KZ_DISPATCH_CAND(rc, fn, args)          \
    do {                                        \
        switch (KP) {                           \
            case 16: rc = fn<16> args; break;   \
            case 32: rc = fn<32> args; break;   \
            case 64: rc = fn<64> args; break;   \
            default: rc = fn<128> args; break;  \
        }                                       \
    } while (0)

// fp16 and split-bf16 kernels: instantiated per list length in kz_knn_h_kp*.hip / kz_knn_bf_kp*.hip (parallel compilation)
int kz_h_occupancy_kp16(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_h_occupancy_kp32(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_h_occupancy_kp64(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_h_occupancy_kp128(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_h_launch_kp16(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_h_launch_kp32(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_h_launch_kp64(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_h_launch_kp128(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_hd_occupancy_kp16(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_hd_occupancy_kp32(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_hd_occupancy_kp64(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_hd_occupancy_kp128(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad);
int kz_hd_launch_kp16(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_hd_launch_kp32(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_hd_launch_kp64(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
int kz_hd_launch_kp128(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide);
bool kz_h64_supports(int n_slices);   // kz_knn_h64.hip: 64 queries per wave, K' = 16, ordinary (dual = 0) and dual-pass builds
int kz_h64_occupancy(int n_slices, int dual, int* blocks_per_cu, int lds_pad);
int kz_h64_launch(int n_slices, int dual, kz_ctx* ctx, const KnnCandParams& p, int n_blocks);
int kz_bf_occupancy_kp16(int n_slices_bf, int* blocks_per_cu, int lds_pad);
int kz_bf_occupancy_kp32(int n_slices_bf, int* blocks_per_cu, int lds_pad);
int kz_bf_occupancy_kp64(int n_slices_bf, int* blocks_per_cu, int lds_pad);
int kz_bf_occupancy_kp128(int n_slices_bf, int* blocks_per_cu, int lds_pad);
int kz_bf_launch_kp16(int n_slices_bf, kz_ctx* ctx, const KnnCandParams& p, int n_blocks);
int kz_bf_launch_kp32(int n_slices_bf, kz_ctx* ctx, const KnnCandParams& p, int n_blocks);
int kz_bf_launch_kp64(int n_slices_bf, kz_ctx* ctx, const KnnCandParams& p, int n_blocks);
int kz_bf_launch_kp128(int n_slices_bf, kz_ctx* ctx, const KnnCandParams& p, int n_blocks);
#define KZ_DISPATCH_KP(rc, fn, args)                 \
    do {                                             \
        switch (KP) {                                \
            case 16: rc = fn##_kp16 args; break;     \
            case 32: rc = fn##_kp32 args; break;     \
            case 64: rc = fn##_kp64 args; break;     \
            default: rc = fn##_kp128 args; break;    \
        }                                            \
    } while (0)

// (kz_plan_rounds, kz_plan_pass, kz_plan_fill_work: kz_plan.h)
extern "C" int kz_knn_plan(int64_t n_query_rows, int64_t n_index_rows, int k_eff, int slots, int force_splits, int min_splits,
                           int* n_rounds, int* round_qtiles, int* round_pieces, int* round_piece_tiles) {
    KZ_REQUIRE(n_rounds && round_qtiles && round_pieces && round_piece_tiles, "kz_knn_plan: null argument");
    KZ_REQUIRE(n_query_rows > 0 && n_index_rows > 0 && slots > 0, "kz_knn_plan: sizes must be positive");
    const int KP = kz_pick_list_len(k_eff);
    KZ_REQUIRE(KP > 0, "kz_knn_plan: k=%d exceeds the supported maximum of 110 neighbours per query", k_eff);
    const int n_qtiles = (int)((n_query_rows + KZ_TILE - 1) / KZ_TILE);
    const int n_ytiles = (int)((n_index_rows + KZ_TILE - 1) / KZ_TILE);
    int sp[KZ_MAX_REGIONS];
    kz_plan_rounds(n_qtiles, n_ytiles, slots, kz_max_pieces(KP, 1), force_splits, min_splits, n_rounds, round_qtiles, sp);
    for (int r = 0; r < *n_rounds; ++r) {
        const int len = (n_ytiles + sp[r] - 1) / sp[r];
        round_piece_tiles[r] = len;
        round_pieces[r] = (n_ytiles + len - 1) / len;
    }
    return KZ_OK;
}

// Rows of a query matrix gathered into a dense block (escalation of uncertified rows to the float32-operand kernel)
__global__ __launch_bounds__(256) void kz_gather_rows_kernel(const char* __restrict__ raw, const int* __restrict__ rows,
                                                             int64_t row0, int n_rows, int64_t row_bytes,
                                                             char* __restrict__ out, int64_t* __restrict__ self_ids,
                                                             const int64_t* __restrict__ parent_self) {
    const int r = blockIdx.x;
    if (r >= n_rows) return;
    const int64_t src = row0 + rows[r];
    const char* sp = raw + src * row_bytes;
    char* dp = out + (int64_t)r * row_bytes;
    for (int64_t b = threadIdx.x * 4; b < row_bytes; b += 256 * 4) *reinterpret_cast<int*>(dp + b) = *reinterpret_cast<const int*>(sp + b);
    // index row to strip for this query: its own row, or (subset of a subset) what its parent recorded for it
    if (self_ids && threadIdx.x == 0) self_ids[r] = parent_self ? parent_self[src] : src;
}

__global__ void kz_iota_kernel(int* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ void kz_strided_rows_kernel(int* __restrict__ out, int n, int64_t stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)((int64_t)i * stride);
}

__global__ __launch_bounds__(256) void kz_scatter_rows_kernel(const double* __restrict__ sd, const int64_t* __restrict__ si,
                                                              const int* __restrict__ rows, int n_rows, int k,
                                                              double* __restrict__ od, int64_t* __restrict__ oi) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_rows * k) return;
    const int r = (int)(t / k), c = (int)(t - (int64_t)r * k);
    od[(int64_t)rows[r] * k + c] = sd[t];
    oi[(int64_t)rows[r] * k + c] = si[t];
}

// Operand precision of the fused kernel ("tier").  kz_knn starts at the highest tier the shapes allow; rows whose
// candidate set cannot be certified under that tier's bound go down: fp16 / split-bf16 -> float32 operands (gathered into
// a dense query block) -> exact float64 brute force.  The result is the float64 neighbour order at every tier.
enum { KZ_TIER_F32 = 0, KZ_TIER_BF = 1, KZ_TIER_H = 2 };
constexpr int KZ_EXACT_MAX_K = 4096;   // neighbours per query on the exact-only route (selection state: 48 KiB of LDS)

// The finalize launches of one pass: one per list region (the dynamic LDS follows the region's entry count: occupancy of
// the gather).  fp.q_first / q_last / max_m are filled here.
static int kz_launch_finalize(kz_ctx* ctx, KnnFinParams& fp, const KzListLayout& lay, int KP, int64_t q_count, int dtype) {
    // the launches of this pass: [first query, last query), entries per query
    struct Group { int64_t lo, hi; int max_m; } groups[KZ_MAX_REGIONS];
    int n_groups = 0;
    for (int rg = 0; rg < lay.n_regions; ++rg) {
        const int64_t lo = (int64_t)(rg > 0 ? lay.qt_end[rg - 1] : 0) * KZ_TILE - fp.list_row0;
        // (neighbouring regions with the same number of ranges -- forced ranges: all of them -- go out as ONE launch)
        while (rg + 1 < lay.n_regions && lay.pieces[rg + 1] == lay.pieces[rg]) ++rg;
        const int64_t hi = (int64_t)lay.qt_end[rg] * KZ_TILE - fp.list_row0;
        Group g = {lo < 0 ? 0 : lo, hi > q_count ? q_count : hi, lay.pieces[rg] * lay.halves * KP};
        if (g.hi > g.lo) groups[n_groups++] = g;
    }
    // The SMALL launches -- the last query tiles of a pass, swept in many short ranges so that they fill the chip: a few hundred
    // queries with hundreds of list entries each, all latency (100k x 100k: 117 us after the 284 us of the main launch) -- go to
    // the context's second stream and run BESIDE the large one (fork / join by events), unless that stream is busy with the
    // reverse chain of a shared sweep or is the stream this call runs on.
    const hipStream_t main_stream = ctx->stream;
    const bool fork = n_groups >= 2 && ctx->stream2 && ctx->stream2 != main_stream && !ctx->stream2_busy;
    int big = 0;
    for (int g = 1; g < n_groups; ++g)
        if (groups[g].hi - groups[g].lo > groups[big].hi - groups[big].lo) big = g;
    if (fork) {
        KZ_HIP(hipEventRecord(ctx->ev[7], main_stream));
        KZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev[7], 0));
    }
    for (int pass = 0; pass < 2; ++pass) {   // (fork: the small launches first, on the second stream; then the large one)
        for (int g = 0; g < n_groups; ++g) {
            const bool side = fork && g != big;
            if (fork ? (side != (pass == 0)) : pass == 1) continue;
            const hipStream_t st = side ? ctx->stream2 : main_stream;
            fp.q_first = groups[g].lo;
            fp.q_last = groups[g].hi;
            fp.max_m = groups[g].max_m;
            fp.fast_div = ctx->fin_fast_div;
            const int fin_blocks = (int)((fp.q_last - fp.q_first + KZ_FIN_QPB - 1) / KZ_FIN_QPB);
            size_t fin_lds = (size_t)4 * kz_fin_wave_bytes(fp.max_m, fp.KSEL > 0 ? fp.KSEL : KP);
            const bool wide = (fp.KSEL > 0 ? fp.KSEL : KP) > 160;   // the long-k route
            // (many selected candidates + float32 rows on the fp16 tier, ordinary direction: kz_knn_fin_wide.h -- option "fin_wide")
            // ("fin_wide" = 2: every launch that selects from several lists, KSEL > 0 -- the short-list routes -- takes it too)
            const bool rows_vec = fp.d <= 256 && (fp.d & 3) == 0 && (((uintptr_t)fp.qraw | (uintptr_t)fp.yraw) & 15u) == 0;
            const bool wide2 = (wide || (KZ_K_FIN_WIDE >= 2 && fp.KSEL > 0)) && dtype == KZ_F32 && fp.tier_h && !fp.excl_floor && KZ_K_FIN_WIDE && rows_vec;
            // (float32 rows of 260 .. 512 elements, 16-byte aligned: the build whose pipelined re-rank takes two loads per lane and row)
            const bool two_chunks = !wide && !wide2 && dtype == KZ_F32 && fp.d > 256 && fp.d <= 512 && (fp.d & 3) == 0 &&
                                    (((uintptr_t)fp.qraw | (uintptr_t)fp.yraw) & 15u) == 0;
            const void* fk = two_chunks ? (const void*)kz_knn_finalize_kernel<float, KZ_FIN_ROWS, KZ_FIN_WAVES_2, 2> : dtype == KZ_F32 ? (wide2 ? (const void*)kz_knn_finalize_wide_kernel<float, 4> : (wide ? (const void*)kz_knn_finalize_kernel<float, 8, 2> : (const void*)kz_knn_finalize_kernel<float, KZ_FIN_ROWS, KZ_FIN_WAVES>))
                                             : (wide ? (const void*)kz_knn_finalize_kernel<double, 4, 2> : (const void*)kz_knn_finalize_kernel<double, KZ_FIN_ROWS, KZ_FIN_WAVES>);
            if (wide2) fin_lds = (size_t)4 * kz_fin_wide_wave_bytes(fp.max_m, fp.KSEL);
            if (fin_lds > 65536) KZ_HIP(hipFuncSetAttribute(fk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
            if (wide2)
                hipLaunchKernelGGL((kz_knn_finalize_wide_kernel<float, 4>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
            else if (dtype == KZ_F32 && wide)
                hipLaunchKernelGGL((kz_knn_finalize_kernel<float, 8, 2>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
            else if (dtype == KZ_F32 && two_chunks)
                hipLaunchKernelGGL((kz_knn_finalize_kernel<float, KZ_FIN_ROWS, KZ_FIN_WAVES_2, 2>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
            else if (dtype == KZ_F32)
                hipLaunchKernelGGL((kz_knn_finalize_kernel<float, KZ_FIN_ROWS, KZ_FIN_WAVES>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
            else if (wide)
                hipLaunchKernelGGL((kz_knn_finalize_kernel<double, 4, 2>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
            else
                hipLaunchKernelGGL((kz_knn_finalize_kernel<double, KZ_FIN_ROWS, KZ_FIN_WAVES>), dim3(fin_blocks), dim3(256), fin_lds, st, fp);
        }
    }
    KZ_HIP(hipGetLastError());
    if (fork) {
        KZ_HIP(hipEventRecord(ctx->ev[11], ctx->stream2));
        KZ_HIP(hipStreamWaitEvent(main_stream, ctx->ev[11], 0));
    }
    return KZ_OK;
}

// One launch of a fused kernel: list layout, scratch carve-up and the uploaded work table.
struct KzPass {
    KzListLayout lay;
    int W;            // workgroups
    int W0;           // boot_first: the first W0 items of the table are the items of index range 0 (else 0)
    float* out_key;   // candidate lists (scratch)
    int* out_idx;
    int* fail_list;   // [fail_rows] (scratch)
    double* fail_tau; // [fail_rows] (scratch; KnnFinParams::fail_tau)
    int4* d_work;     // [W] (scratch)
};

// Plans the rounds (kz_plan_rounds), carves the context's scratch block and uploads the work table.  tier decides the
// list layout (fp16: K' contiguous entries per list; float32 kernels: two lane-half lists per query and range).
// tpw = query tiles per workgroup (wide fp16 builds: 2 or 3, kz_knn_h16.h "WIDE"): the plan is made for UNITS of tpw consecutive
// query tiles -- one work item = one unit x one index range, w4.x = its first tile -- and converted back to tiles for the list
// layout (a region ends on a unit boundary, the last one at the last tile).
static int kz_prepare_pass(kz_ctx* ctx, int n_qtiles, int n_ytiles, int slots, int max_pieces, int KP, int tier, int64_t fail_rows,
                           KzPass* out, int tpw = 1, int force_pieces = 0, int min_pieces = 0, bool boot_first = false) {
    KzPlan pl;
    kz_plan_pass(n_qtiles, n_ytiles, slots, max_pieces, (tier == KZ_TIER_H ? 1 : 2) * KP, tier == KZ_TIER_F32 ? 2 : 1,
                 tier == KZ_TIER_H ? 1 : 0, tpw, force_pieces > 0 ? force_pieces : ctx->force_splits, min_pieces > KZ_K_MIN_SPLITS ? min_pieces : KZ_K_MIN_SPLITS, &pl);
    const KzListLayout& lay = pl.lay;
    const int W = pl.W;
    const size_t list_elems = pl.list_elems;
    // (the kernels address a launch's lists with 32-bit element offsets -- kz_knn_h16.h "off_u"; a K' = 16 chunk of 2 M rows over
    //  KZ_MAX_PIECES = 128 ranges is exactly 2^32 elements: refused here instead of wrapping there)
    KZ_REQUIRE(list_elems < ((size_t)1 << 32), "kz_knn: the candidate lists of one launch exceed 2^32 entries (%zu): fewer rows per chunk (option chunk_rows)", list_elems);
    const size_t key_bytes = (list_elems * 4 + 255) & ~(size_t)255;
    const size_t fail_bytes = ((size_t)fail_rows * 4 + 255) & ~(size_t)255;
    const size_t tau_bytes = ((size_t)fail_rows * 8 + 255) & ~(size_t)255;
    const size_t work_bytes = ((size_t)W * sizeof(int4) + 255) & ~(size_t)255;
    void* scratch = nullptr;
    int rc = kz_scratch(ctx, key_bytes * 2 + fail_bytes + tau_bytes + work_bytes, &scratch);
    if (rc != KZ_OK) return rc;
    out->lay = lay;
    out->W = W;
    out->out_key = (float*)scratch;
    out->out_idx = (int*)((char*)scratch + key_bytes);
    out->fail_list = (int*)((char*)scratch + 2 * key_bytes);
    out->fail_tau = (double*)((char*)scratch + 2 * key_bytes + fail_bytes);
    out->d_work = (int4*)((char*)scratch + 2 * key_bytes + fail_bytes + tau_bytes);
    {
        // host-side table (pinned staging grows on demand)
        const size_t need = work_bytes;
        if (need * 2 > ctx->h_stage_bytes) {
            KZ_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->h_stage) KZ_HIP(hipHostFree(ctx->h_stage));
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            KZ_HIP(hipHostMalloc(&ctx->h_stage, need * 4, hipHostMallocDefault));
            ctx->h_stage_bytes = need * 4;
        }
        // The table of the PREVIOUS pass may still be in flight out of h_stage when two passes follow each other without a
        // read-back in between (dual pass: sample sweep, then the main sweep): passes alternate between the two halves of
        // the staging buffer, and every second pass is followed by a stream synchronisation in any case (kz_knn_impl's
        // fail-counter read).
        ctx->h_stage_flip ^= 1;
        int4* hw = (int4*)((char*)ctx->h_stage + (ctx->h_stage_flip ? ctx->h_stage_bytes / 2 : 0));
        static_assert(sizeof(KzWorkItem) == sizeof(int4), "work items are uploaded as int4");
        // (fp16 kernel: query groups of four times what an XCD holds, see kz_plan_fill_work; "qgroup" overrides.  500k x 500k,
        //  ten ranges per query tile, main kernel: 24: 109.3 ms, 96: 107.6, 384: 102.5, 768 .. 4096: 101.9 .. 103.2)
        const int per_xcd = (slots + 7) / 8;
        const int qgroup = KZ_K_QGROUP > 0 ? KZ_K_QGROUP : (tier == KZ_TIER_H && 4 * per_xcd > KZ_QGROUP ? 4 * per_xcd : KZ_QGROUP);
        kz_plan_fill_work(pl, n_ytiles, tpw, (KzWorkItem*)hw, qgroup);
        out->W0 = 0;
        if (boot_first) {
            // RANGE-0 BOOTSTRAP (kz_knn_impl): the items of index range 0 first -- they are launched on their own, the others
            // behind them with a floor read off range 0's lists.  (Stable: both groups keep the XCD-aware order among themselves.)
            KzWorkItem* w = (KzWorkItem*)hw;
            out->W0 = (int)(std::stable_partition(w, w + W, [](const KzWorkItem& it) { return it.w == 0; }) - w);
        }
        KZ_HIP(hipMemcpyAsync(out->d_work, hw, (size_t)W * sizeof(int4), hipMemcpyHostToDevice, ctx->stream));
    }
    return KZ_OK;
}

// ---------------------------------------------------------------------------------------------------
// POPULATION FLOOR (seeded lists).  A list that starts empty (threshold -inf) takes K' (1 + ln(n / K')) events per query over a
// sweep, half of them in the first few tiles.  A strided probe of the query rows (an escalation-style sub-search: the tier
// probe of an ordinary search, a probe of its own in kz_knn_dual; exact float64 results, written to their places) shows where
// the k-th best key of a row lies as a function of |q_c|^2: least squares over the probe, the floor = the model minus the
// largest shortfall seen (times "floor_margin").  Every list of the main sweep then STARTS at its row's floor
// (KnnCandParams::qfloor), and the finalize kernel counts the floor into its bound on the rows outside the lists
// (KnnFinParams::list_floor).  A row whose k-th key lies below its floor ends with fewer than k candidates, is not certified
// and is searched again like any other uncertified row: the floor decides how many rows take that path, never a result.  Whatever
// the data looks like, a row falls short of the largest shortfall among P probe rows with probability 1 / (P + 1) (the rows are
// exchangeable): at most ~n / P rows of a call take the detour.
// ---------------------------------------------------------------------------------------------------
// probe row i = matrix row i * stride: (|q_c|^2, exact key of its k-th neighbour) from the probe's float64 distances.
// key = (|q_c|^2 - d^2) / 2 with d^2 = the squared distance (cosine: 2 x distance, rows are unit vectors).
__global__ void kz_floor_pairs_kernel(const double* __restrict__ dist, const double* __restrict__ rowq, int n_probe, int64_t stride, int k,
                                      int metric, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_probe) return;
    const int64_t r = (int64_t)i * stride;
    const double v = dist[r * k + k - 1];
    const double d2 = metric == KZ_EUCLIDEAN ? v * v : (metric == KZ_COSINE ? 2.0 * v : v);
    const double x = rowq[r * 3];
    out[2 * i] = x;
    out[2 * i + 1] = 0.5 * (x - d2);
}
// floor of list row p (matrix row row_map[p]) in the units of the sweep's approximate keys: the model's key minus the
// margin, minus the rounding bound of this row's approximate keys (the finalize kernel's eps_q), scaled and rounded down.
// Pad rows: +inf (no events at all).
__global__ void kz_floor_rows_kernel(const int* __restrict__ row_map, int64_t n, int64_t n_pad, const double* __restrict__ rowq,
                                     const double* __restrict__ y_hmax, const double* __restrict__ hscale, double alpha, double beta,
                                     double margin, double eps_mult, double gamma_acc, float* __restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pad) return;
    const int64_t r = row_map ? (int64_t)row_map[p] : (p < n ? p : -1);   // (no map: the rows in their own order)
    if (r < 0) {
        out[p] = INFINITY;
        return;
    }
    const double qc2 = rowq[r * 3 + 0], qh = rowq[r * 3 + 1], qr = rowq[r * 3 + 2];
    const double Yh = y_hmax[0], Ry = y_hmax[1], Yc2 = y_hmax[2];
    const double qc = sqrt(qc2), yc = sqrt(Yc2);
    const double eps = eps_mult * (qr * Yh + qh * Ry + qr * Ry + gamma_acc * (0.5 * Yc2 + qh * Yh) +
                                   1.1920928955078125e-07 * (qc + yc) * (qc + yc) + 1e-12 * (0.5 * Yc2 + qc2));
    const double f = (alpha + beta * qc2 - margin - eps) / hscale[1];
    float ff = (float)f;
    if ((double)ff > f) ff = nextafterf(ff, -INFINITY);
    out[p] = ff;
}

static inline double kz_gamma_acc_h(int kg) { return 2.0 * (double)(kg * 4 + 16) * 5.9604644775390625e-08; }
// model = {alpha, beta, margin}; *ok = false when the probe's values are not finite (no floor then).  `dist`: [.., k] results whose
// row i * stride is probe row i; rowq: the query image's per-row statistics, same row numbering.  Waits for the stream.
static int kz_floor_model(kz_ctx* ctx, const double* dist, const double* rowq, int n_probe, int64_t stride, int k, int metric, double* model,
                          bool* ok) {
    double* d_pairs = nullptr;
    int rc = kz_pool_alloc(ctx, (size_t)n_probe * 16, (void**)&d_pairs);
    if (rc != KZ_OK) return rc;
    std::vector<double> hp((size_t)n_probe * 2);
    hipLaunchKernelGGL(kz_floor_pairs_kernel, dim3((unsigned)((n_probe + 255) / 256)), dim3(256), 0, ctx->stream, dist, rowq, n_probe, stride, k,
                       metric, d_pairs);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(hp.data(), d_pairs, (size_t)n_probe * 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    kz_pool_free(ctx, d_pairs, 0);
    if (e != hipSuccess) {
        kz_set_error("kz_knn: floor probe failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    *ok = kz_floor_fit(hp.data(), n_probe, ctx->floor_margin, model);
    return KZ_OK;
}

// RANGE-0 BOOTSTRAP of the short-list routes (round 5).  A query keeps one list of 16 per index RANGE of the row-dealt image, and
// range 0 of a dealt image is a systematic 1 / P sample of the index: the smallest key of its full list -- the 16th best over the
// sample, about rank 16 P over everything -- is a lower bound of the query's 16 P-th best key, known after 1 / P of the sweep.  The
// other P - 1 ranges are swept behind it with their lists STARTING at that floor (KnnCandParams::qfloor, as the population floor of
// the seeded lists does): keys at or below it never become events.  A list that starts empty takes 16 (1 + ln(n / (16 P)))
// events; with the floor a range sees about the 16 P / P = 16 rows above it -- P = 32 lists on 300 k rows: ~3 800 events per query
// without, ~700 with.  The finalize kernel counts the floor into its bound (KnnFinParams::list_floor) exactly as for seeded lists:
// a floor can only make a row uncertified (searched again), never change a result.
__global__ void kz_boot_floor_kernel(const float* __restrict__ in_key, const int* __restrict__ in_idx, KzListLayout lay, int KP, int64_t list_row0,
                                     int64_t q_begin, int64_t q_count, const float* __restrict__ prev, float* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q_count) return;
    const int64_t l0 = kz_list_contig_off(list_row0 + q, lay, KP, 0);
    float mn = INFINITY;
    bool full = true;
    for (int e = 0; e < KP; ++e) {
        full = full && in_idx[l0 + e] >= 0;
        mn = fminf(mn, in_key[l0 + e]);
    }
    float f = full ? mn : -INFINITY;
    if (prev) f = fmaxf(f, prev[q_begin + q]);
    out[q_begin + q] = f;
}

__global__ void kz_floor_nudge_kernel(float* __restrict__ f, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && f[i] > -INFINITY) f[i] = nextafterf(f[i], -INFINITY);
}

// Escalation of uncertified rows: gather rows cq_begin + fail_list[0 .. n_fail) of `query` into a dense block, search it
// again (kz_knn_impl with the given precision / minimum list length; that call sends ITS uncertified rows further down)
// and scatter the results into out_dist / out_ind at the rows' positions.  Ends with a stream synchronisation.
// Dual pass (kz_knn_dual.h): what the main sweep needs to report the events of the index rows besides its own lists.
struct KzDualPass {
    int probed;                    // the caller's tier probe has looked at this data (and chose the shared fp16 sweep): no ladder after the fact
    const float* qpack;            // fp16 image of the query rows in a load-balanced order (kz_knn_dual.h "stratified deal") ...
    const int* row_map;            // ... and [query tiles * 128] the matrix row of each of its rows
    const float* ypack;            // fp16 image of the index rows SORTED by their event threshold (kz_himage_pack_permuted) ...
    const float* ybias;            // ... and its accumulator-init rows
    const int* perm;               // [index rows] matrix row of image row r: list entries are translated by the finalize kernel
    const float* theta;            // [index tiles * 128] per row of the sorted image: the smallest threshold of its tile
    const float* qnbias;           // [query tiles * 128]
    const float* qfloor;           // [query tiles * 128] or nullptr: seeded forward lists (kz_knn_dual.h "population floor")
    void* log_keys;
    void* log_meta;
    unsigned long long* log_cnt;   // device counter
    long long log_cap;
    int broken;                    // set by kz_knn_impl when a chunk did not run the dual build (tier change): events incomplete
    int short_pieces, short_ksel, short_kp;  // > 0: the forward lists are short_pieces lists of 16 per query (the image interleaves the index tiles
                                   // over the ranges, kz_knn_dual.h), the finalize kernel selects short_ksel of their entries
    double main_ms;
    // called by kz_knn_impl right behind the launch of the LAST chunk's sweep (the event log is complete once that kernel has
    // run): kz_knn_dual enqueues the reverse direction's chain on the context's second stream there, so that it runs beside the
    // forward direction's finalize kernel, read-back and re-search instead of behind them
    int (*post_sweep)(void* user);
    void* post_user;
    int post_called;
    // NESTED sample sweep (kz_knn_dual.h "NESTED"): the pass is the sweep of b x sample(a) -- its index side is only the first
    // n_ytiles tiles of `ypack` (the sorted sample image; `index` stays the whole matrix a), its forward lists are wanted RAW
    // (thresholds are read off them by the caller's post_sweep hook: lists_* below are filled in before the hook runs) and are
    // never finalized; at most max_entries list entries per query (the threshold kernel ranks <= 256).
    int n_ytiles;
    int raw_lists;
    int max_entries;
    int no_q64;                    // the 32-queries-per-wave build whatever "h_q64" says (kz_range.h reads that build's log format)
    const float* lists_key;
    const int* lists_idx;
    KzListLayout lists_lay;
    int lists_KP;
};
// query rows one launch of the fused kernels takes (the candidate lists of a launch stay below ~1 GiB)
static inline int64_t kz_rows_per_chunk(const kz_ctx* ctx, int KP_mem, bool wide_route) {
    return ctx->chunk_rows > 0 ? ctx->chunk_rows : (int64_t)128 * 4096 * (wide_route ? 1 : (KP_mem <= 16 ? 4 : (KP_mem <= 32 ? 2 : 1)));
}
// SPECULATIVE RESCUE (round 6).  A pass on data that is fine still leaves a HANDFUL of rows uncertified (near-ties around the k-th
// place, a crowded index range: 2 - 20 rows of a 15 k .. 1 M row search), and what they cost was never the arithmetic: the host had to
// learn the count (read-back + stream synchronisation), gather the rows into a sub-matrix, pack it, sweep the whole index for one
// query tile (a latency-bound launch: 108 us on a 15 k-row index), finalize, scatter, synchronise again -- 0.25 ms per search of a
// 15 k x 15 k step that takes 1.7 ms (bench.py "ea15k"), 1.3 ms per step on a 1 M-row index.  The exact float64 kernels answer a row
// for n d multiply-adds whatever the data: they are launched BEHIND the finalize kernel, before the host knows anything, for up to R
// rows -- grid sized for R, the count read from device memory (kz_spec_row_live), every workgroup of a dead row returns at once.  R is
// 4 .. "spec_rows" (64), as many as "spec_elems" / (n d) allows (a row of a 1 M x 200 index is 0.2 G multiply-adds: R = 8).  The
// read-back that follows tells the host whether that was all (count <= R: the results are in place -- the exact float64 order, what
// every route returns) or whether the ordinary re-search has to run (count > R: the speculative launches did nothing).
struct KzSpec {
    int R = 0;            // rows the speculative launches cover (0: not launched)
    double* vals = nullptr;
    double* cand_v = nullptr;
    int* cand_i = nullptr;
    double* qd = nullptr;   // float64 operand rows of the (up to R) query rows: kz_exact_lanes.h
};
static void kz_spec_release(kz_ctx* ctx, KzSpec& sp) {
    kz_pool_free(ctx, sp.qd, 0);
    kz_pool_free(ctx, sp.vals, 0);
    kz_pool_free(ctx, sp.cand_v, 0);
    kz_pool_free(ctx, sp.cand_i, 0);
    sp = KzSpec();
}
static int kz_spec_rows(const kz_ctx* ctx, const kz_matrix* index, int k_eff) {
    if (ctx->spec_rows <= 0 || index->metric >= KZ_MANHATTAN || k_eff > 64) return 0;
    const double nd = (double)index->n * (double)index->d;
    int R = (int)(KZ_K_SPEC_ELEMS / (nd > 1.0 ? nd : 1.0)) & ~3;
    // (where the one-pair-per-lane kernel takes the launch -- kz_spec_rescue -- the index is staged once per block of 16 query rows
    //  whatever their number: 32 rows cost little more than 4; 1 M x 200: 0.53 ms for 4 rows, the re-search of 16 took 3.5 ms)
    if (R < 32 && nd <= 1.0e9 && ctx->exact_rows >= 2 && index->dtype == KZ_F32 && (index->d & 3) == 0 && index->d <= 512 &&   // (<= 4 GB of rows staged, 1.3 GB of values)
        index->n >= (int64_t)4 * 64 * ctx->n_cus && ((uintptr_t)index->raw & 15u) == 0)
        R = 32;
    if (R > ctx->spec_rows) R = ctx->spec_rows & ~3;
    return R < 4 ? 0 : R;
}
// The buffers of a speculation for R rows (ahead of the launches where those run on another stream than the pool's: the reverse
// chains of kz_knn_dual -- a buffer handed out here may have been released by work that is still in flight on the context's stream).
// No memory: sp stays empty and nothing is speculated (the ordinary re-search will do).
static int kz_spec_alloc(kz_ctx* ctx, KzSpec& sp, int R, const kz_matrix* index, int k_eff) {
    const int k_sel = (int)(k_eff < index->n ? k_eff : index->n);
    const int n_chunks = (int)((index->n + KZ_EXACT_CHUNK - 1) / KZ_EXACT_CHUNK);
    const bool two_level = n_chunks >= 2 && k_sel <= KZ_EXACT_CHUNK;
    int rc = kz_pool_alloc(ctx, (size_t)R * (size_t)index->n * 8, (void**)&sp.vals);
    if (rc == KZ_OK && two_level) rc = kz_pool_alloc(ctx, (size_t)R * n_chunks * k_sel * 8, (void**)&sp.cand_v);
    if (rc == KZ_OK && two_level) rc = kz_pool_alloc(ctx, (size_t)R * n_chunks * k_sel * 4, (void**)&sp.cand_i);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, kz_exact_lanes_qd_bytes(R, (int)index->d), (void**)&sp.qd);
    if (rc != KZ_OK) {
        kz_spec_release(ctx, sp);
        return rc == KZ_ERR_NOMEM ? KZ_OK : rc;
    }
    return KZ_OK;
}
// fail_list / fail_count: the finalize kernel's device-side list and counter (fail_list holds rows relative to q0)
static int kz_spec_rescue(kz_ctx* ctx, KzSpec& sp, int R, const kz_matrix* query, int64_t q0, const int* fail_list, const int* fail_count,
                          const kz_matrix* index, int k, int exclude_self, const int64_t* d_self_ids, double* out_dist, int64_t* out_ind) {
    const int metric = index->metric;
    const int k_eff = k + (exclude_self ? 1 : 0);
    const int k_sel = (int)(k_eff < index->n ? k_eff : index->n);
    const size_t sel_lds = (size_t)k_sel * 12 + 16;
    const int n_chunks = (int)((index->n + KZ_EXACT_CHUNK - 1) / KZ_EXACT_CHUNK);
    // (two selection levels from two chunks on: the single-level kernel passes k_eff times over the whole row with ONE workgroup --
    //  135 us for 15 k values, k = 10; the chunk kernel selects from registers)
    const bool two_level = n_chunks >= 2 && k_sel <= KZ_EXACT_CHUNK;
    if (!sp.vals) {   // (not allocated ahead by the caller: kz_spec_alloc)
        const int rc = kz_spec_alloc(ctx, sp, R, index, k_eff);
        if (rc != KZ_OK || !sp.vals) return rc;
    }
    const int dist_blocks = (int)((index->n + 3) / 4 < 256 ? (index->n + 3) / 4 : 256);   // (grid-stride; dead rows cost their dispatch)
    const double* sel_v = two_level ? (const double*)sp.cand_v : (const double*)sp.vals;
    const int* sel_i = two_level ? (const int*)sp.cand_i : (const int*)nullptr;
    const int64_t n_entries = two_level ? (int64_t)n_chunks * k_sel : index->n;
    bool lanes = false;
    // (... from ~4 workgroups of 64 index rows per CU on: on a 15 k-row index its 235 workgroups run one per CU, all latency -- 73 us
    //  against the cooperative kernel's 60)
    if (index->dtype == KZ_F32 && index->n >= (int64_t)4 * KZ_XL_ROWS * ctx->n_cus) {
        // (one pair per lane where that kernel applies: a pass over a 500 k x 200 index per FOUR rows made the cooperative kernel 2 ms
        //  for 16 rows -- on the critical path behind the forward finalize; 0.3 ms)
        const int rcl = kz_launch_exact_lanes(ctx, fail_list, 0, R, q0, query, index, metric, sp.vals, &lanes, fail_count, sp.qd);
        if (rcl != KZ_OK) {
            kz_spec_release(ctx, sp);
            return rcl;
        }
    }
    if (lanes) {
    } else if (index->dtype == KZ_F32) {
        // (a handful of rows: short stretches of index rows per wave, so that the launch is wide -- 15 k rows: 235 x R / 4 workgroups)
        int rpw = (int)(index->n / 1024);
        rpw = rpw < 16 ? 16 : (rpw > 256 ? 256 : rpw);
        if (!(ctx->exact_rows && kz_launch_exact_rows(ctx, fail_list, 0, R, q0, query, index, metric, sp.vals, fail_count, rpw)))
            hipLaunchKernelGGL(kz_exact_dist_kernel<float>, dim3(dist_blocks, R), dim3(256), 0, ctx->stream, fail_list, 0, q0,
                               (const float*)query->raw, (const float*)index->raw, query->sqn, index->sqn, index->n, (int)index->d, metric,
                               index->mink_p, sp.vals, fail_count);
    } else {
        hipLaunchKernelGGL(kz_exact_dist_kernel<double>, dim3(dist_blocks, R), dim3(256), 0, ctx->stream, fail_list, 0, q0,
                           (const double*)query->raw, (const double*)index->raw, query->sqn, index->sqn, index->n, (int)index->d, metric,
                           index->mink_p, sp.vals, fail_count);
    }
    if (two_level)
        hipLaunchKernelGGL(k_sel >= 24 && ctx->exact_rows ? kz_exact_chunk_radix_kernel : kz_exact_chunk_kernel, dim3(n_chunks, R), dim3(256), 0,
                           ctx->stream, (const double*)sp.vals, index->n, k_sel, n_chunks, sp.cand_v, sp.cand_i, fail_count);
    if (index->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_exact_select_kernel<float>, dim3(R), dim3(256), sel_lds, ctx->stream, fail_list, 0, q0, sel_v, sel_i, n_entries,
                           index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p, out_dist, out_ind, fail_count);
    else
        hipLaunchKernelGGL(kz_exact_select_kernel<double>, dim3(R), dim3(256), sel_lds, ctx->stream, fail_list, 0, q0, sel_v, sel_i, n_entries,
                           index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p, out_dist, out_ind, fail_count);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kz_spec_release(ctx, sp);
        kz_set_error("kz_knn: speculative exact re-search failed to launch: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    sp.R = R;
    return KZ_OK;
}

static int kz_knn_impl(kz_ctx* ctx, kz_matrix* query, int64_t q_begin, int64_t q_count, kz_matrix* index, int k,
                       int exclude_self, const int64_t* d_self_ids, int precision_override, int kp_min, double* d_dist,
                       int64_t* d_ind, kz_knn_stats* stats, KzDualPass* dual);
static int kz_escalate_rows(kz_ctx* ctx, kz_matrix* query, int64_t cq_begin, const int* fail_list, int n_fail, kz_matrix* index, int k,
                            int exclude_self, const int64_t* d_self_ids, int precision_override, int kp_min, double* out_dist,
                            int64_t* out_ind, kz_knn_stats* st2, float* ms_out) {
    KZ_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
    const size_t row_bytes = (size_t)query->d * (query->dtype == KZ_F32 ? 4 : 8);
    int* fl = nullptr;
    void* sub_raw = nullptr;
    int64_t* sub_self = nullptr;
    double* sub_dist = nullptr;
    int64_t* sub_ind = nullptr;
    kz_matrix* qsub = nullptr;
    auto release = [&]() {
        if (qsub) kz_matrix_destroy(qsub);
        kz_pool_free(ctx, fl, 0);
        kz_pool_free(ctx, sub_raw, 0);
        kz_pool_free(ctx, sub_self, 0);
        kz_pool_free(ctx, sub_dist, 0);
        kz_pool_free(ctx, sub_ind, 0);
    };
    int rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&fl);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * row_bytes, &sub_raw);
    if (rc == KZ_OK && exclude_self) rc = kz_pool_alloc(ctx, (size_t)n_fail * 8, (void**)&sub_self);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * k * 8, (void**)&sub_dist);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * k * 8, (void**)&sub_ind);
    if (rc != KZ_OK) {
        release();
        return rc;
    }
    hipError_t e = hipMemcpyAsync(fl, fail_list, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kz_gather_rows_kernel, dim3(n_fail), dim3(256), 0, ctx->stream, (const char*)query->raw, fl, cq_begin,
                           n_fail, (int64_t)row_bytes, (char*)sub_raw, sub_self, d_self_ids);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        release();
        kz_set_error("kz_knn: gathering the escalated rows failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    rc = kz_matrix_create(ctx, sub_raw, 2, n_fail, query->d, query->dtype, query->metric, &qsub);
    memset(st2, 0, sizeof(*st2));
    if (rc == KZ_OK)
        rc = kz_knn_impl(ctx, qsub, 0, n_fail, index, k, exclude_self, sub_self, precision_override, kp_min, sub_dist, sub_ind, st2, nullptr);
    if (rc == KZ_OK) {
        hipLaunchKernelGGL(kz_scatter_rows_kernel, dim3((unsigned)(((int64_t)n_fail * k + 255) / 256)), dim3(256), 0, ctx->stream,
                           sub_dist, sub_ind, fl, n_fail, k, out_dist, out_ind);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ctx->ev[4], ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            kz_set_error("kz_knn: scattering the escalated rows failed: %s", hipGetErrorString(e));
            rc = KZ_ERR_HIP;
        }
    }
    release();
    if (rc != KZ_OK) return rc;
    KZ_HIP(hipEventElapsedTime(ms_out, ctx->ev[3], ctx->ev[4]));
    return KZ_OK;
}

// d_self_ids (device, optional): index row to strip per query when exclude_self is set and the query matrix is not the
// index matrix itself (escalated subsets).  precision_override: -1 = the context's setting, 1 = float32 operands only.
// kp_min: smallest list length to use (escalated subsets of the fp16 tier are first re-done with LONGER lists on the same
// operand images: the certification compares the K'-th approximate key with the k-th exact one, so more margin in ranks
// is usually all a failed row needs, and unlike the float32 tier it costs no new image of the index).
// LADDER AFTER THE FACT (round 5).  A pass that leaves MORE THAN HALF of its rows uncertified without a tier probe having looked at
// the data first (searches below the probe's size gates; tools/cliff_probe.py: 100k x 101k x 128, tight clusters, 60 - 100 ms
// against 5.6 on uniform rows) used to hand all of them to the split-bf16 operands, where rows of a tight cluster fail again --
// what they lack is margin in ranks.  Now a strided sample of the failed rows goes through the fp16 tier's WIDE route first (their
// results are final either way); at most a quarter of the sample uncertified there: every failed row takes that route, else the
// caller's choice (prec, kp_min).  Fewer than 4096 failed rows: the caller's choice at once.
__global__ void kz_strided_pick_kernel(const int* __restrict__ in, int n_out, int64_t stride, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) out[i] = in[(int64_t)i * stride];
}
static int kz_escalate_ladder(kz_ctx* ctx, kz_matrix* query, int64_t cq_begin, const int* fail_list, int n_fail, kz_matrix* index, int k,
                              int exclude_self, const int64_t* d_self_ids, int prec, int kp_min, double* out_dist, int64_t* out_ind,
                              kz_knn_stats* st, float* ms_out) {
    float ms_probe = 0;
    // (the wide route wants at least 8 index tiles per list: a small index gets fewer lists, down to 8 -- 128 entries per query)
    int P = ctx->wide_lists;
    if ((int64_t)index->n_tiles < (int64_t)8 * P) P = (int)(index->n_tiles / 8);
    int* keep = nullptr;   // (the caller's list lives in the pass's scratch block, which the probe's own search reuses: a private copy)
    if (P >= 8 && ctx->esc_ladder && n_fail >= 4096) {
        const int n_probe = 1024;
        int* plist = nullptr;
        int rc = kz_pool_alloc(ctx, (size_t)n_probe * sizeof(int), (void**)&plist);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&keep);
        if (rc != KZ_OK) {
            kz_pool_free(ctx, plist, 0);
            return rc;
        }
        if (hipMemcpyAsync(keep, fail_list, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
            kz_pool_free(ctx, plist, 0);
            kz_pool_free(ctx, keep, 0);
            kz_set_error("kz_knn: copying the list of uncertified rows failed");
            return KZ_ERR_HIP;
        }
        hipLaunchKernelGGL(kz_strided_pick_kernel, dim3((n_probe + 255) / 256), dim3(256), 0, ctx->stream, keep, n_probe, (int64_t)(n_fail / n_probe), plist);
        kz_knn_stats stw;
        memset(&stw, 0, sizeof(stw));
        rc = kz_escalate_rows(ctx, query, cq_begin, plist, n_probe, index, k, exclude_self, d_self_ids, 0, -P, out_dist, out_ind, &stw, &ms_probe);
        kz_pool_free(ctx, plist, 0);
        if (rc != KZ_OK) {
            kz_pool_free(ctx, keep, 0);
            return rc;
        }
        if (stw.wide_lists > 0 && (int64_t)stw.n_first_pass_fail * 4 <= n_probe) {
            prec = 0;
            kp_min = -P;
        }
        fail_list = keep;
    }
    const int rc = kz_escalate_rows(ctx, query, cq_begin, fail_list, n_fail, index, k, exclude_self, d_self_ids, prec, kp_min, out_dist, out_ind, st, ms_out);
    kz_pool_free(ctx, keep, 0);
    if (ms_out) *ms_out += ms_probe;
    return rc;
}

#include "kz_range.h"

static int kz_knn_impl(kz_ctx* ctx, kz_matrix* query, int64_t q_begin, int64_t q_count, kz_matrix* index, int k,
                       int exclude_self, const int64_t* d_self_ids, int precision_override, int kp_min, double* d_dist,
                       int64_t* d_ind, kz_knn_stats* stats, KzDualPass* dual) {
    KZ_REQUIRE(ctx && query && index && d_dist && d_ind, "kz_knn: null argument");
    KZ_REQUIRE(query->ctx == ctx && index->ctx == ctx, "kz_knn: matrices belong to a different context");
    KZ_REQUIRE(!query->raw_only && !index->raw_only, "kz_knn: a rows-only matrix (kz_matrix_create rows_on_device = 3) cannot be searched");
    KZ_REQUIRE(query->d == index->d, "kz_knn: feature dimensions differ (%lld vs %lld)", (long long)query->d,
               (long long)index->d);
    KZ_REQUIRE(query->dtype == index->dtype, "kz_knn: query and index must have the same dtype");
    KZ_REQUIRE(query->metric == index->metric && query->mink_p == index->mink_p, "kz_knn: query and index were packed for different metrics");
    KZ_REQUIRE(q_begin >= 0 && q_count >= 0 && q_begin + q_count <= query->n, "kz_knn: query row range out of bounds");
    KZ_REQUIRE(k >= 1, "kz_knn: Expected k > 0. Got %d", k);
    // kp_min >= 1000: lists of kp_min - 1000, and NOT the short-list route (the re-search of rows that route could not certify:
    // every re-search must differ from the pass that failed)
    const bool no_short = kp_min >= 1000;
    if (no_short) kp_min -= 1000;
    // kp_min <= -2: the WIDE fp16 route with -kp_min lists of 16 per query (below, "WIDE ROUTE"): a caller's probe has found that
    // this data needs more margin in ranks, not better operands
    int forced_lists = 0;
    if (kp_min <= -2) {
        forced_lists = -kp_min;
        kp_min = 0;
    }
    const int k_eff = k + (exclude_self ? 1 : 0);
    KZ_REQUIRE((int64_t)k_eff <= index->n, "kz_knn: Expected n_neighbors %s n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld",
               exclude_self ? "<" : "<=", k, (long long)index->n);
    if (exclude_self && !d_self_ids)
        KZ_REQUIRE(query->n == index->n, "kz_knn: exclude_self needs query and index of equal length");
    // More than 110 neighbours per query: no fused kernel keeps a list that long -- but the kernels keep ONE list per query and
    // index RANGE, and the finalize kernel merges them.  LONG-k ROUTE (111 .. ~540 neighbours): lists of 128 over S >= k / 24
    // index ranges (a range then holds ~24 of a query's k nearest rows on average; a range that holds more than its list does is
    // seen by the certification -- kz_finalize_query: piece_bound -- and the row goes down the tiers), from which the finalize
    // kernel selects and re-ranks KSEL = k + max(16, k / 8) candidates.  Beyond that (or an index too short to cut into S
    // ranges of >= 4 tiles) the call runs entirely on the exact float64 kernels (k selection rounds over the full distance row
    // per query: correct for any k <= n, and slow -- the reference's scikit-learn path has no such limit either,
    // sklearn_nearest_neighbors.py:51-65; INTEGRATION.md "Deviations").
    // The Minkowski family beyond p = 2 (KZ_MANHATTAN, KZ_CHEBYSHEV, KZ_MINKOWSKI) has no inner-product form: no MFMA kernel, the
    // call runs entirely on the exact float64 kernels (scikit-learn's own generic DatasetsPair path is the slow one there too).
    const bool no_gemm_form = index->metric >= KZ_MANHATTAN;
    int KP = no_gemm_form ? 0 : kz_pick_list_len(k_eff);
    int KSEL = 0, long_pieces = 0;
    if (KP == 0 && !dual && KZ_K_LONG_K && !no_gemm_form) {
        const int S = k_eff / 24 + 1 > 4 ? k_eff / 24 + 1 : 4;
        const int sel = k_eff + (k_eff / 8 > 16 ? k_eff / 8 : 16);
        // (finalize: 4 waves x (S 128 entries x 8 B + KSEL x 28 B) of LDS per workgroup)
        // (float32-operand tier: two lane-half lists per range, hence at most 8 ranges = 16 lists, see kz_prepare_pass below)
        if (S <= KZ_FIN_MAXM / 128 && 4 * kz_fin_wave_bytes(S * 128, sel) <= 160 * 1024 &&
            4 * kz_fin_wave_bytes((S < 8 ? S : 8) * 256, sel) <= 160 * 1024 && (int64_t)index->n_tiles >= (int64_t)4 * S) {
            KP = 128;
            KSEL = sel;
            long_pieces = S;
        }
    }
    // SHORT-LIST ROUTE of the dual pass (13 .. 110 neighbours): the same construction the other way round -- lists
    // of 16 over k / 5 index ranges instead of one list of 32 / 64 / 128 per query.  The K' = 16 kernel keeps three workgroups per CU
    // and its merges short; a range that holds 16 or more of a query's nearest rows is seen by the certification (piece_bound).
    // The caller has dealt the index tiles over the ranges (kz_knn_dual.h): near rows of a query sit in ALL ranges alike.
    const int KP_class = KP;
    if (dual && dual->short_pieces > 0 && KP > dual->short_kp && kp_min <= dual->short_kp) {
        KP = dual->short_kp;
        KSEL = dual->short_ksel;
        long_pieces = dual->short_pieces;
    }
    const bool exact_only = KP == 0;
    if (exact_only) {
        if (k_eff > KZ_EXACT_MAX_K) {
            kz_set_error("kz_knn: k=%d exceeds the supported maximum of %d neighbours per query", k_eff, KZ_EXACT_MAX_K);
            return KZ_ERR_UNSUPPORTED;
        }
        KP = 128;   // (list geometry of the scratch block only; no list kernel runs)
    }
    if (KP < kp_min) KP = kp_min < 128 ? kp_min : 128;   // (list lengths are 16 / 32 / 64 / 128)
    // kp_min = -1 (escalated rows of a K' = 16 pass): MORE LISTS instead of longer ones -- a list of 16 per index range over at
    // least four ranges, the finalize kernel selecting k + 16 of their entries.  The bound of the certification becomes the
    // largest 16th-best key of a RANGE (a quarter of the index or less) instead of the 16th-best key of the whole index: the
    // margin in ranks a failed row needs, with the K' = 16 kernel and a quarter of the entries to merge (14 rows of a 1M-row
    // index: 0.75 + 0.62 ms with lists of 64 over 64 ranges).
    int min_pieces_call = 0;
    if (kp_min == -1) {
        if (KP == 16 && !exact_only && KSEL == 0 && index->n_tiles >= 16) {
            KSEL = k_eff + 48;   // (<= 60 of the >= 64 entries)
            min_pieces_call = 4;
        } else if (!exact_only && KP < 64) {
            KP = 64;   // (an index of a few tiles: lists of 64 -- every re-search must ask for MORE than the pass that failed)
        }
    }
    if (stats) memset(stats, 0, sizeof(*stats));
    if (q_count == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(ctx->device));

    const int metric = index->metric;
    const int n_ytiles = (dual && dual->n_ytiles > 0) ? dual->n_ytiles : (int)index->n_tiles;
    const int n_slices = index->kg / 4;
    // rounding bound factors.  float32 operands: (d_pad + 16) 2^-24 covers the d+1 step fma chain, the float32 rounding of
    // the bias and (float64 inputs) of the operands; 1e-12 covers the float64 re-rank's own rounding.  fp16 operands: the
    // float32 accumulation of d_pad exact products + bias in an unspecified order, (n + 16) u doubled to allow for
    // truncating internal adds (the operand rounding is measured per row, kz_pack.hip).
    const double gamma_f32 = ((double)(index->kg * 4 + 16) * 5.9604644775390625e-08 + 1e-12) * ctx->eps_scale;
    const double gamma_bf = kz_bf16_gamma(index->kg_bf * 4) * ctx->eps_scale;
    const double gamma_acc_h = kz_gamma_acc_h(index->kg);

    // ---- tier of this call ------------------------------------------------------------------------------------------
    const int precision = precision_override >= 0 ? precision_override : ctx->precision;
    int tier = KZ_TIER_F32;
    if (precision != 1 && n_slices >= 2 && n_slices <= 24 && query->kg == index->kg) tier = precision == 2 ? KZ_TIER_BF : KZ_TIER_H;
    if (exact_only) tier = KZ_TIER_F32;   // (nothing is packed or launched for it below)
    if (tier == KZ_TIER_H) {
        const int rc = kz_himage_ensure(query, index);
        if (rc != KZ_OK) return rc;
    }
    // SHORT-LIST ROUTE of the ordinary kernel: as in the dual pass, lists of 16 over P index ranges instead of one list of 32 /
    // 64 / 128 per query -- on a second image of the index whose ROWS are dealt over the ranges (kz_himage_dealt; in the caller's
    // row order the near rows of a query may all sit in one stretch).  P lists hold at least as many entries as the list they
    // replace; taken when a range has at least 48 tiles (measured down to 49: k = 100 on 125k index rows 15.3 -> 8.2 ms, k = 50 on
    // 83k rows 11.3 -> 9.6, k = 26 on 60k rows 6.9 -> 6.5).  (The long lists' kernels stay for small indexes.)
    const int KP_long = KP, KSEL_long = KSEL, pieces_long = long_pieces;   // (the list geometry this call would use without the route)
    bool short_ord = false;
    int route_P = 0;      // ranges the short-list route dealt the index over (kz_himage_dealt)
    float probe_ms = 0;   // (tier probe, below: reported with the fallback time)
    // (the long-k route up to 320 neighbours as well: k / 5 <= 64 lists of 16 instead of 4 .. 14 lists of 128 -- 50k x 500k x 200, main
    //  kernel: k = 128 32.4 -> 12.5 ms, k = 160 35.4 -> 13.3; beyond 32 lists the finalize kernel selects by repeated arg-max)
    const bool longk_lists = KSEL > 0 && long_pieces > 0 && KP == 128 && kp_min <= 0 && k_eff <= 320;
    if (!dual && !no_short && tier == KZ_TIER_H && ctx->short_ord && KP > 16 && (KSEL == 0 || longk_lists) && !exact_only) {
        int P = (k_eff + KZ_K_DUAL_SHORT_DIV - 1) / KZ_K_DUAL_SHORT_DIV;
        if (P < KP / 16) P = KP / 16;
        if (kp_min >= 128) P = 16;   // (a re-search that asks for lists of 128: all the ranges the finalize kernel's fast selection takes)
        const int sel = k_eff + (KP >= 128 ? 80 : 48) < P * 16 ? k_eff + (KP >= 128 ? 80 : 48) : P * 16;
        if (P <= (longk_lists ? 64 : 32) && (int64_t)index->n_tiles >= (int64_t)ctx->short_ord_min_tiles * P && sel >= k_eff &&
            4 * kz_fin_wave_bytes(P * 16, sel) <= 160 * 1024) {   // (<= 512 entries: kz_rank_select<8>; up to 1024 -- 33 .. 64 lists of the long-k route -- the radix select)
            const int rc = kz_himage_dealt(index, P);
            if (rc == KZ_OK) {
                short_ord = true;
                KP = 16;
                KSEL = sel;
                long_pieces = P;
                route_P = P;
            } else if (rc != KZ_ERR_NOMEM) {
                return rc;
            }   // (no memory for the second image: the long list)
        }
    }
    // WIDE ROUTE (round 5): many lists of 16 -- "wide_lists" (32) of them over as many ranges of the row-dealt image, the finalize
    // kernel selecting "wide_sel" (256) of their entries.  For data whose keys are DENSE around the k-th neighbour (tight clusters:
    // hundreds of rows of a cluster lie within the rounding bound of the k-th key).  What such a row needs to be certified is margin
    // in RANKS -- the bound on the rows outside the candidate set must fall 2 eps below the k-th key, i.e. the set must reach down
    // to the ~250th key -- and that costs list events and re-ranked rows, not MFMA products: the fp16 kernel with one product per
    // multiply-add stays, where the split-bf16 tier pays three (bench.py "hard": every row failed the fp16 pass with lists worth
    // ~100 ranks; with 256 they are certified).  Taken when the tier probe says so (below) or a caller asks for it (forced_lists).
    bool wide_route = false;
    auto wide_geometry = [&](int P, int* sel_out) -> bool {
        int sel = ctx->wide_sel < P * 16 ? ctx->wide_sel : P * 16;
        if (sel < k_eff + 16) sel = k_eff + 16 < P * 16 ? k_eff + 16 : P * 16;
        *sel_out = sel;
        return !dual && tier == KZ_TIER_H && !exact_only && KP_long <= 128 && KSEL_long == 0 && P >= 2 && P <= 32 && P * 16 > KP_long &&
               sel >= k_eff && (int64_t)index->n_tiles >= (int64_t)8 * P && 4 * kz_fin_wave_bytes(P * 16, sel) <= 160 * 1024;
    };
    auto take_wide = [&](int P) -> int {
        int sel = 0;
        if (!wide_geometry(P, &sel)) return KZ_ERR_UNSUPPORTED;
        const int rc = kz_himage_dealt(index, P);
        if (rc != KZ_OK) return rc;
        short_ord = true;
        wide_route = true;
        KP = 16;
        KSEL = sel;
        long_pieces = P;
        return KZ_OK;
    };
    if (forced_lists > 0) {
        const int rc = take_wide(forced_lists);
        if (rc != KZ_OK && rc != KZ_ERR_UNSUPPORTED && rc != KZ_ERR_NOMEM) return rc;   // (not available: this call's ordinary route)
    }
    // TIER PROBE.  Data that is hard for fp16 as a whole (tight clusters far from the centre: nearly every row fails the first pass'
    // certification) used to pay for a complete fp16 sweep and its finalize before anything went down the tiers (bench.py "hard":
    // 2 x 31.6 of 150 ms per step).  A large ordinary search therefore first sends a STRIDED sample of its query rows (1024 rows since the end of round 4, 4096 before:
    // representative whatever the row order) through the fp16 pass as an escalation-style sub-search; if more than half of them
    // cannot be certified, the call starts at the split-bf16 tier.  The sample's results are written to their places (the main
    // pass writes the same values again).  Cost on data that is fine: ~0.35 % of a 300k-row sweep + ~0.3 ms (230k x 230k x 128: 13.5 -> 13.2 ms
    // with 1024 instead of 4096 rows; 200k x 400k x 200, cosine, k = 50: 32.4 -> 31.8); only top-level searches of >= 5e10 distance pairs
    // and >= 16 probe sizes of query rows take it (C1 / C2 do not).  Option "tier_probe" = 0: off.
    float* qfloor_ord = nullptr;   // seeded lists of an ordinary search (the context's buffer: nothing to release)
    bool probed = false;           // the tier probe below has run: its verdict stands (no ladder after the fact)
    if (tier == KZ_TIER_H && !dual && precision_override < 0 && kp_min == 0 && forced_lists == 0 && !exact_only && ctx->tier_probe > 0 && ctx->esc_bf &&
        q_count >= (int64_t)16 * ctx->tier_probe && ctx->chunk_rows == 0 &&
        ((double)q_count * (double)index->n >= ctx->probe_min_pairs ||
         2.0 * (double)q_count * (double)index->n * (double)(index->kg * 4) / 1e12 >= KZ_K_PROBE_MIN_MS)) {
        const int n_probe = ctx->tier_probe;
        int* plist = nullptr;
        int rc = kz_pool_alloc(ctx, (size_t)n_probe * sizeof(int), (void**)&plist);
        if (rc != KZ_OK) return rc;
        hipLaunchKernelGGL(kz_strided_rows_kernel, dim3((unsigned)((n_probe + 255) / 256)), dim3(256), 0, ctx->stream, plist, n_probe,
                           q_count / n_probe);
        kz_knn_stats stp;
        float pms = 0;
        rc = kz_escalate_rows(ctx, query, q_begin, plist, n_probe, index, k, exclude_self, d_self_ids, 0, 0, d_dist, d_ind, &stp, &pms);
        int* plist_keep = plist;
        if (rc != KZ_OK) {
            kz_pool_free(ctx, plist, 0);
            return rc;
        }
        probe_ms = pms;
        probed = true;
        // (the verdict counts the rows that left the probe's FIRST pass uncertified, once each -- not the cumulative count of the
        //  levels below it, which counted a row that went two levels down twice)
        bool hard = (int64_t)stp.n_first_pass_fail * 2 > n_probe;
        if ((int64_t)stp.n_first_pass_fail * 8 > n_probe && ctx->wide_lists >= 2) {
            // LADDER: before better operands, more margin in ranks on the SAME operands -- the probe rows again through the wide
            // route.  Tried from an eighth of the probe uncertified on: re-searching a quarter of the rows one by one costs more
            // than the sweep itself (300 k x 300 k x 96, clusters of very different spread, k = 10: 24 % of the rows re-searched,
            // 24 ms of sweep in a 93 ms call).  Taken when at most a quarter of the probe stays uncertified there AND that is less
            // than half of what the ordinary lists left.
            int sel = 0;
            if (wide_geometry(ctx->wide_lists, &sel)) {
                kz_knn_stats stw;
                float wms = 0;
                rc = kz_escalate_rows(ctx, query, q_begin, plist_keep, n_probe, index, k, exclude_self, d_self_ids, 0, -ctx->wide_lists, d_dist, d_ind,
                                      &stw, &wms);
                if (rc != KZ_OK) {
                    kz_pool_free(ctx, plist_keep, 0);
                    return rc;
                }
                probe_ms += wms;
                if ((int64_t)stw.n_first_pass_fail * 4 <= n_probe && (int64_t)stw.n_first_pass_fail * 2 < stp.n_first_pass_fail &&
                    take_wide(ctx->wide_lists) == KZ_OK)
                    hard = false;
            }
        }
        kz_pool_free(ctx, plist_keep, 0);
        if (hard) {
            tier = KZ_TIER_BF;
            // (data this hard for fp16 is hard for the split-bf16 operands, too, wherever the keys are dense: lists of 64 from the
            //  start -- 300k x 300k x 96, k = 10, clusters of very different spread: rows searched again 135 k -> 51 k, call 170 -> 110 ms)
            if (KSEL == 0 && KP < 64) KP = 64;
            if (short_ord) {   // (the other tiers' kernels keep one list of K' per query)
                short_ord = false;
                KP = KP_long;
                KSEL = KSEL_long;
                long_pieces = pieces_long;
            }
        } else if (ctx->list_floor && !wide_route) {
            // POPULATION FLOOR (above kz_escalate_rows): the probe's results are the model's input
            double model[3];
            bool ok = false;
            rc = kz_floor_model(ctx, d_dist, query->himg->rowq + q_begin * 3, n_probe, q_count / n_probe, k, metric, model, &ok);
            if (rc != KZ_OK) return rc;
            const int64_t n_pad = (int64_t)query->n_tiles * KZ_TILE;
            if (ok) rc = kz_floor_buf(ctx, (size_t)n_pad * 4, &qfloor_ord);
            if (rc != KZ_OK) return rc;
            if (qfloor_ord) {
                hipLaunchKernelGGL(kz_floor_rows_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, ctx->stream, (const int*)nullptr, query->n,
                                   n_pad, query->himg->rowq, index->himg->d_max, index->himg->center->d_scale, model[0], model[1], model[2],
                                   ctx->eps_scale, gamma_acc_h, qfloor_ord);
                KZ_HIP(hipGetLastError());
            }
        }
    }
    // 64 QUERIES PER WAVE (kz_knn_h64.h): K' = 16 sweeps of 4 .. 13 slices -- half the LDS fragment reads per MFMA and half the
    // LDS-DMA volume per query of the 32-query kernel at two waves per SIMD instead of three; a work item = a unit of two query
    // tiles (tpw = 2).  Measured (profiles/r04_ablation.md section 2): the shared sweep at 13 slices -1.5 % (250k x 1M x 200: 90.6 ->
    // 89.2 ms), the ordinary kernel +0.8 % there, +8 % at 8 slices, and a launch of fewer than ~4 rounds of units does not fill the
    // chip (100k x 100k: +30 %).  Option "h_q64": 2 (default) = the shared sweep from 9 slices on over >= 4 rounds of units,
    // 1 = wherever the kernel is built for (tests), 0 = never.
    const bool q64_ok = tier == KZ_TIER_H && KP == 16 && kz_h64_supports(n_slices) && !exact_only;
    const bool q64 = q64_ok && !(dual && dual->no_q64) && (ctx->h_q64 == 1 || (ctx->h_q64 == 2 && dual && n_slices >= 9 &&
                                                     (q_count + 2 * KZ_TILE - 1) / (2 * KZ_TILE) >= (int64_t)4 * 2 * ctx->n_cus));
    int slots_cache[3] = {0, 0, 0};
    int tpw_h = 1;   // query tiles per workgroup of the fp16 kernel this call runs (wide builds: 2 or 3)
    auto slots_for = [&](int t, int* out) -> int {
        if (slots_cache[t] == 0) {
            int blocks_per_cu = 1;
            int rc0;
            if (t == KZ_TIER_H && q64) {
                rc0 = kz_h64_occupancy(n_slices, dual ? 1 : 0, &blocks_per_cu, KZ_K_LDS_PAD);
                tpw_h = 2;
            } else if (t == KZ_TIER_H && dual)
                KZ_DISPATCH_KP(rc0, kz_hd_occupancy, (n_slices, &blocks_per_cu, &tpw_h, KZ_K_H_WPS, KZ_K_H_WIDE, KZ_K_LDS_PAD));
            else if (t == KZ_TIER_H)
                KZ_DISPATCH_KP(rc0, kz_h_occupancy, (n_slices, &blocks_per_cu, &tpw_h, KZ_K_H_WPS, KZ_K_H_WIDE, KZ_K_LDS_PAD));
            else if (t == KZ_TIER_BF)
                KZ_DISPATCH_KP(rc0, kz_bf_occupancy, (n_slices, &blocks_per_cu, KZ_K_LDS_PAD));
            else
                KZ_DISPATCH_CAND(rc0, kz_cand_occupancy, (&blocks_per_cu));
            if (rc0 != KZ_OK) return rc0;
            slots_cache[t] = blocks_per_cu * ctx->n_cus;
        }
        *out = slots_cache[t];
        return KZ_OK;
    };
    // query rows are processed in chunks so that the candidate lists stay below ~1 GiB: 524288 rows for K' >= 64, up to four
    // times as many for shorter lists (1 M x 250 k, K' = 16: one launch instead of two -- one tail round, one read-back, one
    // re-search of the uncertified rows)
    const int KP_mem = KP_class > KP ? KP_class : KP;   // (short-list route: several lists of 16 -- the chunk of the replaced list length)
    // (the wide route keeps 32 lists of 16 per query -- 4 KiB: 524288 rows)
    int64_t max_rows_per_chunk = kz_rows_per_chunk(ctx, KP_mem, wide_route);   // (halved below where a chunk's lists would pass 2^32 entries)
    if (dual && dual->raw_lists && q_count > max_rows_per_chunk) {
        kz_set_error("kz_knn: internal: a raw-list pass must be one launch");
        return KZ_ERR_INVALID;
    }
    double main_ms = 0, fin_ms = 0, fb_ms = 0;   // (the tier probe's time is reported under its own field, kz_knn_stats.probe_ms)
    int64_t n_fail_total = 0, n_escalated = 0, n_first_fail = 0, n_spec = 0, n_range = 0, n_range_pairs = 0, n_range_group = 0;
    double max_err_ratio = 0.0;
    int last_splits = 1, last_blocks = 0, first_tier = tier;
    for (int64_t c0 = 0; c0 < q_count;) {
        if (tier == KZ_TIER_BF) {
            int rc = kz_matrix_image_bf(query);
            if (rc == KZ_OK) rc = kz_matrix_image_bf(index);
            if (rc != KZ_OK) return rc;
        } else if (tier == KZ_TIER_F32 && !exact_only) {
            int rc = kz_matrix_image_f32(query);
            if (rc == KZ_OK) rc = kz_matrix_image_f32(index);
            if (rc != KZ_OK) return rc;
        }
        if (tier == KZ_TIER_H && short_ord && long_pieces >= 2) {
            // (a re-search of the previous chunk's uncertified rows may have selected -- or packed -- the index dealt over another
            //  number of ranges: this route's own image again; cached, two are kept)
            // (any failure ends the call: launching on whatever image the last sub-search selected would certify against the wrong
            //  range layout; unreachable while a slot exists once the route is chosen)
            const int rcd = kz_himage_dealt(index, wide_route || forced_lists > 0 ? long_pieces : route_P);
            if (rcd != KZ_OK) {
                if (rcd == KZ_ERR_NOMEM) kz_set_error("kz_knn: out of device memory for the row-dealt image of the index");
                return rcd;
            }
        }
        int slots = 0;
        {
            const int rcs = slots_for(tier, &slots);
            if (rcs != KZ_OK) return rcs;
        }
        const int64_t cq_begin = q_begin + c0;
        const int qt0 = (int)(cq_begin / KZ_TILE);
        int64_t cq_count = 0;
        int n_qtiles = 0, force_pieces = 0, max_pieces = 0;
        // ---- schedule: which workgroup sweeps which (query tile, index-tile range): kz_prepare_pass above --------------
        for (;;) {
            cq_count = (q_count - c0 < max_rows_per_chunk) ? (q_count - c0) : max_rows_per_chunk;
            n_qtiles = (int)((cq_begin + cq_count - 1) / KZ_TILE) - qt0 + 1;
            // (long-k route: exactly long_pieces index ranges per query tile, one round)
            force_pieces = (tier == KZ_TIER_F32 && long_pieces > 8) ? 8 : long_pieces;
            if (tier == KZ_TIER_H && short_ord && force_pieces > 0) {
                // a SMALL launch on the short-list route (a re-search of a few hundred uncertified rows, a probe): P ranges per query
                // tile leave most of the chip idle (216 rows x 10 ranges: 20 workgroups sweeping 390 tiles each, 1.98 ms) -- every
                // range is cut further, s P lists of 16 per query (the finalize kernel selects from any number of lists), as long as
                // a piece keeps at least 8 tiles and a query at most KZ_MAX_PIECES (128) lists
                const int units = (n_qtiles + tpw_h - 1) / tpw_h;
                int sub = slots / (units * force_pieces);
                if (sub > KZ_MAX_PIECES / force_pieces) sub = KZ_MAX_PIECES / force_pieces;
                if (sub > n_ytiles / (8 * force_pieces)) sub = n_ytiles / (8 * force_pieces);
                if (sub > 1 && 4 * kz_fin_wave_bytes(force_pieces * sub * 16, KSEL) <= 160 * 1024) force_pieces *= sub;
            }
            max_pieces = kz_max_pieces(KP, tier == KZ_TIER_F32 ? 2 : 1);
            if (dual && dual->max_entries > 0 && max_pieces > dual->max_entries / KP) max_pieces = dual->max_entries / KP;
            // The kernels address a launch's lists with 32-bit element offsets: a chunk whose plan would pass 2^32 entries (K' = 16,
            // 2 M rows over 128 ranges) is halved -- the plan is host arithmetic, made here once more than kz_prepare_pass makes it.
            KzPlan pl;
            kz_plan_pass(n_qtiles, n_ytiles, slots, max_pieces, (tier == KZ_TIER_H ? 1 : 2) * KP, tier == KZ_TIER_F32 ? 2 : 1, tier == KZ_TIER_H ? 1 : 0,
                         tier == KZ_TIER_H ? tpw_h : 1, force_pieces > 0 ? force_pieces : ctx->force_splits,
                         min_pieces_call > KZ_K_MIN_SPLITS ? min_pieces_call : KZ_K_MIN_SPLITS, &pl);
            if (pl.list_elems < ((size_t)1 << 32) || cq_count <= 8 * KZ_TILE || (dual && dual->raw_lists)) break;
            max_rows_per_chunk = ((cq_count / 2 + KZ_TILE - 1) / KZ_TILE) * KZ_TILE;
        }
        KzPass ps;
        // (range-0 bootstrap: the short-list routes of the ordinary 32-query kernel, from four ranges on)
        const bool boot = tier == KZ_TIER_H && short_ord && !dual && !q64 && KZ_K_RANGE_BOOT && force_pieces >= 4;
        int rc = kz_prepare_pass(ctx, n_qtiles, n_ytiles, slots, max_pieces, KP, tier, cq_count, &ps,
                                 tier == KZ_TIER_H ? tpw_h : 1, force_pieces, min_pieces_call, boot);
        if (rc != KZ_OK) return rc;
        const KzListLayout& lay = ps.lay;
        const int W = ps.W;
        float* out_key = ps.out_key;
        int* out_idx = ps.out_idx;
        int* fail_list = ps.fail_list;
        int4* d_work = ps.d_work;
        int* fail_count = ctx->d_counters + 8;
        KZ_HIP(hipMemsetAsync(fail_count, 0, 4 * sizeof(int), ctx->stream));  // fail counter, (unused), error-ratio bits
        const float* boot_floor = nullptr;   // (this chunk's range-0 floor, if any: the finalize kernel must know it)
        unsigned long long* stamp_buf = nullptr;   // (diagnostic "abl_stamp")
        KnnCandParams cp;
        memset(&cp, 0, sizeof(cp));
        if (tier == KZ_TIER_H) {
            cp.qpack = (const float*)query->himg->packed;
            cp.ypack = (const float*)(short_ord ? index->himg->dealt_packed : index->himg->packed);
            cp.ybias = short_ord ? index->himg->dealt_bias : index->himg->bias;
        } else {
            cp.qpack = tier == KZ_TIER_BF ? (const float*)query->packed_bf : query->packed;
            cp.ypack = tier == KZ_TIER_BF ? (const float*)index->packed_bf : index->packed;
            cp.ybias = index->bias;
        }
        cp.work = d_work;
        cp.qt0 = qt0;
        cp.n_ytiles = n_ytiles;
        cp.n_qtiles = n_qtiles;
        cp.lay = lay;
        cp.kg = index->kg;
        cp.out_key = out_key;
        cp.out_idx = out_idx;
        if (tier == KZ_TIER_H && !dual) cp.qfloor = qfloor_ord;
        KZ_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
        if (exact_only) {
            // every row of the chunk goes to the exact kernels: the "fail list" is 0 .. cq_count-1
            hipLaunchKernelGGL(kz_iota_kernel, dim3((unsigned)((cq_count + 255) / 256)), dim3(256), 0, ctx->stream, fail_list, (int)cq_count);
            KZ_HIP(hipGetLastError());
        } else if (tier == KZ_TIER_H && dual) {
            cp.qpack = dual->qpack;
            cp.ypack = dual->ypack;
            cp.ybias = dual->ybias;
            cp.theta = dual->theta;
            cp.qnbias = dual->qnbias;
            cp.qfloor = dual->qfloor;
            cp.log_keys = dual->log_keys;
            cp.log_meta = dual->log_meta;
            cp.log_cnt = dual->log_cnt;
            cp.log_cap = dual->log_cap;
            if (q64)
                rc = kz_h64_launch(n_slices, 1, ctx, cp, W);
            else
                KZ_DISPATCH_KP(rc, kz_hd_launch, (n_slices, ctx, cp, W, KZ_K_H_WPS, KZ_K_H_WIDE));
        } else if (tier == KZ_TIER_H && q64)
            rc = kz_h64_launch(n_slices, 0, ctx, cp, W);
        else if (tier == KZ_TIER_H && boot && ps.W0 > 0 && ps.W0 < W) {
            // range 0 of every query tile, the floor off its lists, then the other ranges
            KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, ps.W0, KZ_K_H_WPS, KZ_K_H_WIDE));
            if (rc != KZ_OK) return rc;
            float* bfloor = nullptr;
            const int64_t n_pad = (int64_t)query->n_tiles * KZ_TILE;
            rc = kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&bfloor);   // (a buffer of this chunk: escalated sub-searches boot too)
            if (rc != KZ_OK) return rc;
            hipLaunchKernelGGL(kz_boot_floor_kernel, dim3((unsigned)((cq_count + 255) / 256)), dim3(256), 0, ctx->stream, out_key, out_idx, lay, KP,
                               cq_begin - (int64_t)qt0 * KZ_TILE, cq_begin, cq_count, cp.qfloor, bfloor);
            KZ_HIP(hipGetLastError());
            cp.qfloor = bfloor;
            cp.work = d_work + ps.W0;
            KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, W - ps.W0, KZ_K_H_WPS, KZ_K_H_WIDE));
            cp.work = d_work;
            boot_floor = bfloor;
        } else if (tier == KZ_TIER_H) {
            if ((ctx->abl & 2) && getenv("KZ_STAMP_FILE")) {   // (diagnostic: a -DKZ_ABL_STAMP build of the fp16 units fills it)
                rc = kz_pool_alloc(ctx, (size_t)W * (16 + 1024), (void**)&stamp_buf);
                if (rc != KZ_OK) return rc;
                KZ_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)W * (16 + 1024), ctx->stream));
                cp.log_meta = stamp_buf;
                cp.log_keys = stamp_buf + 2 * (size_t)W;
            }
            KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, W, KZ_K_H_WPS, KZ_K_H_WIDE));
            cp.log_meta = nullptr;
            cp.log_keys = nullptr;
            if (rc == KZ_OK && (ctx->abl & 1) && !short_ord && ps.lay.n_regions == 1 && ps.lay.pieces[0] == 1) {
                // DIAGNOSTIC ("abl_refloor", profiles/r06_event_ablation.md): the same sweep AGAIN with every list starting at the
                // threshold it ENDED on (the K'-th best key of the first sweep's list): K' insertions per query instead of
                // K' (1 + ln(n / K')) -- the time any seeding of the lists could at best reach.  The second sweep is the one timed.
                float* bfloor = nullptr;
                const int64_t n_pad = (int64_t)query->n_tiles * KZ_TILE;
                rc = kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&bfloor);
                if (rc != KZ_OK) return rc;
                hipLaunchKernelGGL(kz_boot_floor_kernel, dim3((unsigned)((cq_count + 255) / 256)), dim3(256), 0, ctx->stream, out_key, out_idx, lay, KP,
                                   cq_begin - (int64_t)qt0 * KZ_TILE, cq_begin, cq_count, cp.qfloor, bfloor);
                // (one ulp below: entries equal to the threshold must get in again)
                hipLaunchKernelGGL(kz_floor_nudge_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, ctx->stream, bfloor, n_pad);
                KZ_HIP(hipGetLastError());
                cp.qfloor = bfloor;
                KZ_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
                KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, W, KZ_K_H_WPS, KZ_K_H_WIDE));
                boot_floor = bfloor;
            }
        } else if (tier == KZ_TIER_BF)
            KZ_DISPATCH_KP(rc, kz_bf_launch, (n_slices, ctx, cp, W));
        else
            KZ_DISPATCH_CAND(rc, kz_launch_cand, (ctx, cp, W));
        if (rc != KZ_OK) return rc;
        if (dual && tier != KZ_TIER_H) dual->broken = 1;   // this chunk's pairs were not scanned for events
        KZ_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
        if (dual) {
            dual->lists_key = out_key;
            dual->lists_idx = out_idx;
            dual->lists_lay = lay;
            dual->lists_KP = KP;
        }
        if (dual && dual->post_sweep && !dual->broken && c0 + cq_count >= q_count) {
            dual->post_called = 1;
            rc = dual->post_sweep(dual->post_user);
            if (rc != KZ_OK) return rc;
        }

        KnnFinParams fp;
        memset(&fp, 0, sizeof(fp));
        fp.in_key = out_key;
        fp.in_idx = out_idx;
        fp.lay = lay;
        fp.KP = KP;
        fp.KSEL = KSEL;
        fp.list_row0 = cq_begin - (int64_t)qt0 * KZ_TILE;
        fp.q_begin = cq_begin;
        fp.q_count = cq_count;
        fp.qraw = query->raw;
        fp.yraw = index->raw;
        fp.qsqn = query->sqn;
        fp.ysqn = index->sqn;
        fp.n_i = index->n;
        fp.d = (int)index->d;
        fp.metric = metric;
        fp.k = k;
        fp.exclude_self = exclude_self ? 1 : 0;
        fp.self_ids = d_self_ids;
        fp.gamma = tier == KZ_TIER_BF ? gamma_bf : gamma_f32;
        fp.ystats = index->d_stats;
        fp.tier_h = tier == KZ_TIER_H ? 1 : 0;
        if (fp.tier_h) {
            fp.eps_mult = ctx->eps_scale;
            fp.gamma_acc = gamma_acc_h;
            fp.q_rowq = query->himg->rowq;
            fp.y_hmax = index->himg->d_max;
            fp.hscale = index->himg->center->d_scale;
        }
        fp.out_dist = d_dist + c0 * (int64_t)k;
        fp.out_ind = d_ind + c0 * (int64_t)k;
        if (tier == KZ_TIER_H && short_ord) fp.idx_map = index->himg->dealt_perm;   // the lists hold rows of the dealt index image
        if (tier == KZ_TIER_H && !dual) fp.list_floor = boot_floor ? boot_floor : qfloor_ord;
        if (tier == KZ_TIER_H && dual) {
            fp.idx_map = dual->perm;      // the lists hold rows of the sorted index image
            fp.row_map = dual->row_map;   // the chunk is a range of IMAGE rows: results and failures go by matrix row
            fp.list_floor = dual->qfloor;
            fp.out_dist = d_dist;
            fp.out_ind = d_ind;
        }
        fp.fail_count = fail_count;
        fp.fail_list = fail_list;
        fp.fail_tau = ps.fail_tau;
        fp.err_ratio_bits = (unsigned long long*)(ctx->d_counters + 10);
        if (fp.tier_h && metric == KZ_COSINE && (fp.KSEL > 0 ? fp.KSEL : KP) > 160 && KZ_K_FIN_WIDE && !(dual && dual->raw_lists)) {
            // (hundreds of re-ranked candidates per query: the normalised float64 rows of the index, built once -- kz_pack.hip)
            rc = kz_matrix_norm64(index);
            if (rc != KZ_OK) return rc;
            fp.ynorm64 = index->norm64;
        }
        if (!exact_only && !(dual && dual->raw_lists)) {   // (raw lists: the caller's hook has read them; nothing is finalized)
            rc = kz_launch_finalize(ctx, fp, lay, KP, cq_count, index->dtype);
            if (rc != KZ_OK) return rc;
        }
        if (boot_floor) kz_pool_free(ctx, const_cast<float*>(boot_floor), 0);   // (stream-ordered: the launches above have it)
        KZ_HIP(hipGetLastError());
        KZ_HIP(hipEventRecord(ctx->ev[2], ctx->stream));
        // SPECULATIVE RESCUE (above kz_escalate_rows): the exact kernels for up to R uncertified rows, before the count is known
        KzSpec spec;
        if (!exact_only && !(dual && dual->raw_lists) && tier != KZ_TIER_F32) {
            const int R = kz_spec_rows(ctx, index, k_eff);
            if (R > 0) {
                rc = kz_spec_rescue(ctx, spec, R, query, fp.row_map ? 0 : cq_begin, fail_list, fail_count, index, k, exclude_self, d_self_ids,
                                    fp.out_dist, fp.out_ind);
                if (rc != KZ_OK) return rc;
                if (spec.R > 0) KZ_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
            }
        }
        KZ_HIP(hipMemcpyAsync(ctx->h_counters + 8, fail_count, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        // matrices created from device rows have not had their finiteness verdict read yet (kz_matrix_create waits for
        // nothing): it rides on this call's read-back
        kz_matrix* unchecked[2] = {query->checked ? nullptr : query, (index->checked || index == query) ? nullptr : index};
        for (int u = 0; u < 2; ++u)
            if (unchecked[u])
                KZ_HIP(hipMemcpyAsync(ctx->h_counters + 44 + 10 * u, unchecked[u]->d_stats, 40, hipMemcpyDeviceToHost, ctx->stream));
        {
            const hipError_t es = hipStreamSynchronize(ctx->stream);
            const int spec_R = spec.R;
            kz_spec_release(ctx, spec);   // (stream-ordered pool: the launches that used the buffers are on the stream)
            spec.R = spec_R;
            KZ_HIP(es);
        }
        if (stamp_buf) {   // (diagnostic: start / end of every workgroup of the sweep, in work-table order, appended to the file)
            std::vector<unsigned long long> hs((size_t)W * 130);
            KZ_HIP(hipMemcpy(hs.data(), stamp_buf, (size_t)W * (16 + 1024), hipMemcpyDeviceToHost));
            kz_pool_free(ctx, stamp_buf, 0);
            if (FILE* f = fopen(getenv("KZ_STAMP_FILE"), "a")) {
                fprintf(f, "# launch W=%d n_qtiles=%d n_ytiles=%d slices=%d\n", W, n_qtiles, n_ytiles, n_slices);
                for (int w = 0; w < W; ++w) {
                    fprintf(f, "%d %llu %llu", w, hs[2 * (size_t)w], hs[2 * (size_t)w + 1]);
                    if (w < 8 || w % 97 == 0)   // (per-tile stamps of a few workgroups: the first 64 tiles, then every 16th)
                        for (int t = 0; t < 128; ++t) fprintf(f, " %llu", hs[2 * (size_t)W + (size_t)w * 128 + t]);
                    fprintf(f, "\n");
                }
                fclose(f);
            }
        }
        for (int u = 0; u < 2; ++u) {
            if (!unchecked[u]) continue;
            if (ctx->h_counters[44 + 10 * u + 8] != 0) {
                kz_set_error("kz_matrix_create: input contains NaN, infinity or a value too large for float32");
                return KZ_ERR_NONFINITE;
            }
            memcpy(&unchecked[u]->max_norm, ctx->h_counters + 44 + 10 * u, 8);
            unchecked[u]->checked = true;
        }
        int n_fail = exact_only ? (int)cq_count : ctx->h_counters[8];
        n_first_fail += n_fail;
        {
            double ratio;
            memcpy(&ratio, ctx->h_counters + 10, 8);
            if (ratio > max_err_ratio) max_err_ratio = ratio;
        }
        float ms = 0;
        KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        main_ms += ms;
        KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
        fin_ms += ms;
        last_splits = lay.pieces[0];
        last_blocks = W;
        // More than a quarter of the chunk's rows uncertified: this data needs better operands -- the REST of the call starts at
        // the next tier (not in the dual pass: its kernel exists for the fp16 tier only).  THIS chunk's uncertified rows go down
        // like any others: searching a quarter (or all) of the rows again at the next tier is never more work than redoing the
        // whole chunk there, which is what an earlier version did -- and abandoned the shared sweep altogether (500k x 62.5k, k = 50, 40
        // tight clusters: 579 -> 540 ms cluster by cluster, 528 -> 348 ms shuffled; nearly every row of that set needs better operands).
        int tier_next = tier;
        if (!dual && tier != KZ_TIER_F32 && (int64_t)n_fail * 4 > cq_count)
            tier_next = (tier == KZ_TIER_H && ctx->esc_bf && long_pieces == 0) ? KZ_TIER_BF : KZ_TIER_F32;
        // A HANDFUL of rows left by the split-bf16 operands skips the float32-operand kernel: that kernel sweeps the whole index for
        // one query tile in at most eight pieces -- 2.2 ms on 300 k rows of d = 64 whatever the row count -- while the exact kernels
        // cost ~35 us a row there (both scale with n d): bench.py "hard", ~20 rows per direction and step: 60.6 -> see r05_notes.
        // (... and so do a few hundred to a few thousand rows where the range re-search applies and the index is large: that kernel's
        //  sweep of the whole index per launch -- 3 ms on 200 k x 200 -- against one fp16 sweep of the failed rows and their pairs;
        //  the tier probe's 1 024 rows paid 2 x 3 ms there on hard data)
        const bool range_direct = tier == KZ_TIER_BF && !dual && !exact_only && n_fail >= KZ_RANGE_MIN_ROWS && n_fail < KZ_RG_MIN_ROWS &&
                                  index->n >= 65536 && kz_range_shapes_ok(ctx, query, index);
        const bool exact_direct = (tier == KZ_TIER_BF && !dual && n_fail > 0 && n_fail <= KZ_K_EXACT_DIRECT_ROWS) || range_direct;
        // (the speculative launches behind the finalize kernel have answered them all)
        const bool rescued = spec.R > 0 && n_fail > 0 && n_fail <= spec.R;
        if (rescued) {
            KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
            fb_ms += ms;
            n_spec += n_fail;
        }
        if (tier != KZ_TIER_F32 && n_fail > 0 && !exact_direct && !rescued) {
            // Escalate only the uncertified rows: gather them into a dense query block and search it again -- fp16 tier
            // with lists shorter than 128: same operands, lists four times as long (no new image of the index: 14 rows
            // of a 1M-row index cost 0.4 ms this way against 7 ms for packing its float32 image); otherwise the split-bf16
            // operands (from the fp16 tier), then the float32-operand kernel.  The inner call sends its own uncertified rows further down (float32 operands,
            // exact float64 kernels).  Results are scattered back.
            // (more than half of the chunk uncertified: the fp16 operands are the wrong tool for this data, longer lists of the
            //  same keys will not help most of them -- straight to the split-bf16 operands)
            // (... and so are the rows the WIDE route leaves: it already is the fp16 tier's largest margin in ranks)
            const bool fp16_hard = tier == KZ_TIER_H && ctx->esc_bf && (wide_route || ((int64_t)n_fail * 2 > cq_count && (long_pieces == 0 || short_ord || (dual && dual->short_pieces > 0))));
            const bool widen = tier == KZ_TIER_H && KP < 128 && !fp16_hard;
            // (short-list route: the rows it cannot certify are mostly the ones a list of K' could not certify either -- they go
            //  where that list's failures would have gone, lists four times K', not through a list of K' first)
            const int KP_esc = short_ord ? KP_long : (KP_class > KP ? KP_class : KP);
            // (... when they are many: lists of 128 -- which the callee turns into 16 lists of 16 on the dealt image when the index
            //  is large: more ranges than this pass had, so not the same search again.  A handful -- uniform data: ~2e-4 of the
            //  queries, those whose near rows crowd one range -- is certified by ONE list of K' at a quarter of the cost: 500k x
            //  500k, k = 50: 4.7 -> 1.7 ms per step)
            const bool crowding_only = KP_esc > KP && n_fail <= KZ_ESC_SHORT_MAX_ROWS;
            // what the re-search asks for: (operand tier, list length) -- never what this pass just tried
            int next_prec, next_kp;
            if (!widen) {   // fp16 with its longest lists, or fp16 altogether, has failed: better operands, this call's own list length
                next_prec = (tier == KZ_TIER_H && ctx->esc_bf && (long_pieces == 0 || fp16_hard)) ? 2 : 1;
                // (the float32 operands are the LAST approximate tier and their a-priori bound is the loosest: with this call's own
                //  list length -- 16 for k = 10 -- the K'-th key lies a handful of keys below the k-th and inside the bound wherever
                //  the keys are dense; lists of 64 certify such rows instead of handing them to the exact kernels at ~60 us a row:
                //  300k x 300k x 96, clusters of very different spread: 9 968 rows to the exact kernels and 726 ms per call before, none and
                //  169 ms now; lists of 128 for every call: bench.py "hard", k = 50, 118 -> 225 ms -- its lists of 64 were long enough)
                next_kp = ((next_prec == 1 || wide_route) && KP_class < 64) ? 64 : 0;
            } else if (KP == 16 && KSEL == 0 && KZ_K_ESC_SHORT && n_fail <= KZ_ESC_SHORT_MAX_ROWS) {
                next_prec = 0;
                next_kp = -1;   // a handful of rows of a K' = 16 pass: more lists of 16
            } else {
                const int len = crowding_only ? KP_esc : (KP_esc * 4 < 128 ? KP_esc * 4 : 128);
                // (after a short-list pass: one LONG list -- unless many rows failed and the callee can still add ranges)
                const bool long_only = KP_esc > KP && (crowding_only || long_pieces >= 16);
                next_prec = 0;
                next_kp = (long_only ? 1000 : 0) + len;
            }
            // EARLY RANGE RE-SEARCH (kz_range.h "grouped"): thousands of rows the split-bf16 operands could not certify are, on data
            // with clusters far tighter than its extent, rows the float32 operands cannot certify either -- they used to cost a sweep
            // of the whole index there (85 of 180 ms per direction, 200 k x 200 k x 200) before the exact kernels got them.  The
            // groups are tried HERE: rows that share a representative's range are answered by the exact kernels at once; the others
            // go on to the next tier as before.
            int* early_left = nullptr;
            const int* esc_list = fail_list;
            // (NOT on what an fp16 pass leaves: rows that only lack margin in ranks -- 40 tight clusters, 100 k x 101 k x 128 -- are
            //  cheaper on the ladder's wide route than as 2.4e8 exact pairs: 26.5 -> 33.7 ms with the groups tried there, removed)
            if (tier == KZ_TIER_BF && !dual && n_fail >= KZ_RG_MIN_ROWS && kz_range_shapes_ok(ctx, query, index)) {
                int* fl0 = nullptr;
                double* tau0 = nullptr;
                rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&fl0);
                if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * 8, (void**)&tau0);
                if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&early_left);
                if (rc == KZ_OK && (hipMemcpyAsync(fl0, fail_list, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
                                    hipMemcpyAsync(tau0, ps.fail_tau, (size_t)n_fail * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)) {
                    kz_set_error("kz_knn: copying the uncertified rows failed");
                    rc = KZ_ERR_HIP;
                }
                int n_after = n_fail;
                long long pairs = 0, grouped = 0;
                const bool no_mem = rc == KZ_ERR_NOMEM;   // (no room for the lists: the next tier as before -- the step is an optimisation)
                if (rc == KZ_OK) {
                    KZ_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
                    rc = kz_range_rescue(ctx, query, fp.row_map ? 0 : cq_begin, fl0, tau0, n_fail, index, k, exclude_self, d_self_ids, fp.out_dist,
                                         fp.out_ind, early_left, &n_after, &pairs, &grouped, true, KZ_RANGE_EARLY_PER_ROW);
                }
                kz_pool_free(ctx, fl0, 0);
                kz_pool_free(ctx, tau0, 0);
                if (no_mem) {
                    kz_pool_free(ctx, early_left, 0);
                    early_left = nullptr;
                    rc = KZ_OK;
                } else {
                    if (rc != KZ_OK) {
                        kz_pool_free(ctx, early_left, 0);
                        return rc;
                    }
                    KZ_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
                    KZ_HIP(hipStreamSynchronize(ctx->stream));
                    KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
                    fb_ms += ms;
                    n_range += n_fail - n_after;
                    n_range_group += grouped;
                    n_range_pairs += pairs;
                    n_fail_total += n_fail - n_after;   // (answered by the exact kernels)
                    n_fail = n_after;
                    esc_list = early_left;
                }
            }
            kz_knn_stats st2;
            memset(&st2, 0, sizeof(st2));
            // (fp16 found hard after the fact, no probe beforehand: the ladder on the failed rows -- top-level calls only)
            const bool ladder = fp16_hard && !wide_route && next_prec == 2 && !probed && kp_min == 0 && forced_lists == 0 && precision_override < 0 &&
                                !(dual && dual->probed);
            ms = 0;
            if (n_fail == 0) {
            } else if (ladder)
                rc = kz_escalate_ladder(ctx, query, fp.row_map ? 0 : cq_begin, esc_list, n_fail, index, k, exclude_self, d_self_ids, next_prec,
                                        next_kp, fp.out_dist, fp.out_ind, &st2, &ms);
            else
                rc = kz_escalate_rows(ctx, query, fp.row_map ? 0 : cq_begin, esc_list, n_fail, index, k, exclude_self, d_self_ids, next_prec,
                                      next_kp, fp.out_dist, fp.out_ind, &st2, &ms);
            kz_pool_free(ctx, early_left, 0);
            if (rc != KZ_OK) return rc;
            fb_ms += ms;
            n_escalated += n_fail + st2.n_escalated_rows;
            n_fail_total += st2.n_fallback_rows;
            n_range += st2.n_range_rows;
            n_range_pairs += st2.n_range_pairs;
            n_range_group += st2.n_range_group_rows;
            if (st2.max_err_ratio > max_err_ratio) max_err_ratio = st2.max_err_ratio;
            c0 += max_rows_per_chunk;
            if (tier_next != tier && short_ord) {   // (the other tiers' kernels keep one list of K' per query)
                short_ord = false;
                KP = KP_long;
                KSEL = KSEL_long;
                long_pieces = pieces_long;
            }
            tier = tier_next;
            continue;
        }
        n_fail_total += n_fail;

        if (n_fail > 0 && !rescued) {
            // exact brute force in batches; the fail list lives at the end of the scratch block, the value matrix
            // goes to a separate allocation so that the list is not overwritten by a scratch regrow.
            KZ_HIP(hipEventRecord(ctx->ev[3], ctx->stream));
            int* fl = nullptr;
            rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&fl);  // stream-ordered pool: no device sync
            if (rc != KZ_OK) return rc;
            KZ_HIP(hipMemcpyAsync(fl, fail_list, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
            // RANGE RE-SEARCH (kz_range.h): the exact kernels on the pairs that can matter; the rows it hands back -- and every
            // row where it does not apply -- go on against the whole index below
            int n_dense = n_fail;
            if (!exact_only && n_fail >= KZ_RANGE_MIN_ROWS && kz_range_shapes_ok(ctx, query, index)) {
                double* tau = nullptr;
                int* left = nullptr;
                rc = kz_pool_alloc(ctx, (size_t)n_fail * 8, (void**)&tau);
                if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&left);
                if (rc == KZ_OK && hipMemcpyAsync(tau, ps.fail_tau, (size_t)n_fail * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
                    kz_set_error("kz_knn: copying the bounds of the uncertified rows failed");
                    rc = KZ_ERR_HIP;
                }
                long long pairs = 0, grouped = 0;
                const bool no_mem = rc == KZ_ERR_NOMEM;   // (no room for the lists: the whole-index kernels as before)
                if (rc == KZ_OK)
                    rc = kz_range_rescue(ctx, query, cq_begin, fl, tau, n_fail, index, k, exclude_self, d_self_ids, fp.out_dist, fp.out_ind, left,
                                         &n_dense, &pairs, &grouped);
                kz_pool_free(ctx, tau, 0);
                if (no_mem) {
                    kz_pool_free(ctx, left, 0);
                    rc = KZ_OK;
                } else {
                    if (rc != KZ_OK) {
                        kz_pool_free(ctx, left, 0);
                        kz_pool_free(ctx, fl, 0);
                        return rc;
                    }
                    n_range += n_fail - n_dense;
                    n_range_pairs += pairs;
                    n_range_group += grouped;
                    kz_pool_free(ctx, fl, 0);   // (the rows handed back take the list's place)
                    fl = left;
                }
            }
            n_fail = n_dense;
            if (n_fail == 0) {
                kz_pool_free(ctx, fl, 0);
                KZ_HIP(hipEventRecord(ctx->ev[4], ctx->stream));
                KZ_HIP(hipStreamSynchronize(ctx->stream));
                KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
                fb_ms += ms;
                c0 += max_rows_per_chunk;
                continue;
            }
            if (metric == KZ_COSINE && n_fail >= 64 && ctx->exact_rows) {   // (many rows: the normalised float64 index rows, once)
                rc = kz_matrix_norm64(index);
                if (rc != KZ_OK) return rc;
            }
            int64_t batch = ((int64_t)256 << 20) / (index->n * 8);
            if (batch < 1) batch = 1;
            if (batch > n_fail) batch = n_fail;
            if (batch > 65535) batch = 65535;
            void* vals = nullptr;
            rc = kz_scratch(ctx, (size_t)batch * (size_t)index->n * 8, &vals);
            if (rc != KZ_OK) {
                kz_pool_free(ctx, fl, 0);
                return rc;
            }
            const int dist_blocks = (int)((index->n + 3) / 4);
            const int k_sel = (int)(k_eff < index->n ? k_eff : index->n);
            const size_t sel_lds = (size_t)k_sel * 12 + 16;
            if (sel_lds > 65536) {
                kz_pool_free(ctx, fl, 0);
                kz_set_error("kz_knn: k=%d is too large for the exact selection kernel", k_eff);
                return KZ_ERR_UNSUPPORTED;
            }
            // rows of more than four chunks: the selection in two levels (kz_exact_chunk_kernel)
            const int n_chunks = (int)((index->n + KZ_EXACT_CHUNK - 1) / KZ_EXACT_CHUNK);
            const bool two_level = n_chunks > 4 && k_sel <= KZ_EXACT_CHUNK;
            double* cand_v = nullptr;
            int* cand_i = nullptr;
            if (two_level) {
                rc = kz_pool_alloc(ctx, (size_t)batch * n_chunks * k_sel * 8, (void**)&cand_v);
                if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)batch * n_chunks * k_sel * 4, (void**)&cand_i);
                if (rc != KZ_OK) {
                    kz_pool_free(ctx, cand_v, 0);
                    kz_pool_free(ctx, fl, 0);
                    return rc;
                }
            }
            for (int b0 = 0; b0 < n_fail; b0 += (int)batch) {
                const int nb = (n_fail - b0 < batch) ? (n_fail - b0) : (int)batch;
                if (index->dtype == KZ_F32) {
                    bool lanes = false;
                    if (!no_gemm_form) {
                        rc = kz_launch_exact_lanes(ctx, fl, b0, nb, cq_begin, query, index, metric, (double*)vals, &lanes);
                        if (rc != KZ_OK) {
                            kz_pool_free(ctx, fl, 0);
                            kz_pool_free(ctx, cand_v, 0);
                            kz_pool_free(ctx, cand_i, 0);
                            return rc;
                        }
                    }
                    if (lanes) {
                    } else if (no_gemm_form)
                        kz_launch_family_dist<float>(ctx, fl, b0, nb, cq_begin, query, index, (double*)vals);
                    else if (ctx->exact_rows && kz_launch_exact_rows(ctx, fl, b0, nb, cq_begin, query, index, metric, (double*)vals)) {
                    } else
                        hipLaunchKernelGGL(kz_exact_dist_kernel<float>, dim3(dist_blocks, nb), dim3(256), 0, ctx->stream, fl, b0,
                                           cq_begin, (const float*)query->raw, (const float*)index->raw, query->sqn, index->sqn,
                                           index->n, (int)index->d, metric, index->mink_p, (double*)vals);
                    if (two_level)
                        hipLaunchKernelGGL(k_sel >= 24 && ctx->exact_rows ? kz_exact_chunk_radix_kernel : kz_exact_chunk_kernel, dim3(n_chunks, nb), dim3(256), 0,
                                           ctx->stream, (const double*)vals, index->n, k_sel, n_chunks, cand_v, cand_i, (const int*)nullptr);
                    hipLaunchKernelGGL(kz_exact_select_kernel<float>, dim3(nb), dim3(256), sel_lds, ctx->stream, fl, b0, cq_begin,
                                       two_level ? (const double*)cand_v : (const double*)vals, two_level ? (const int*)cand_i : (const int*)nullptr,
                                       two_level ? (int64_t)n_chunks * k_sel : index->n, index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p,
                                       fp.out_dist, fp.out_ind);
                } else {
                    if (no_gemm_form)
                        kz_launch_family_dist<double>(ctx, fl, b0, nb, cq_begin, query, index, (double*)vals);
                    else
                        hipLaunchKernelGGL(kz_exact_dist_kernel<double>, dim3(dist_blocks, nb), dim3(256), 0, ctx->stream, fl, b0,
                                           cq_begin, (const double*)query->raw, (const double*)index->raw, query->sqn, index->sqn,
                                           index->n, (int)index->d, metric, index->mink_p, (double*)vals);
                    if (two_level)
                        hipLaunchKernelGGL(k_sel >= 24 && ctx->exact_rows ? kz_exact_chunk_radix_kernel : kz_exact_chunk_kernel, dim3(n_chunks, nb), dim3(256), 0,
                                           ctx->stream, (const double*)vals, index->n, k_sel, n_chunks, cand_v, cand_i, (const int*)nullptr);
                    hipLaunchKernelGGL(kz_exact_select_kernel<double>, dim3(nb), dim3(256), sel_lds, ctx->stream, fl, b0, cq_begin,
                                       two_level ? (const double*)cand_v : (const double*)vals, two_level ? (const int*)cand_i : (const int*)nullptr,
                                       two_level ? (int64_t)n_chunks * k_sel : index->n, index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p,
                                       fp.out_dist, fp.out_ind);
                }
            }
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(ctx->ev[4], ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            kz_pool_free(ctx, fl, 0);
            kz_pool_free(ctx, cand_v, 0);
            kz_pool_free(ctx, cand_i, 0);
            if (e != hipSuccess) {
                kz_set_error("kz_knn: exact fallback failed: %s", hipGetErrorString(e));
                return KZ_ERR_HIP;
            }
            KZ_HIP(hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
            fb_ms += ms;
        }
        c0 += max_rows_per_chunk;
    }
    if (stats) {
        stats->main_kernel_ms = main_ms;
        stats->finalize_ms = fin_ms;
        stats->fallback_ms = fb_ms;
        stats->probe_ms = probe_ms;
        stats->n_fallback_rows = n_fail_total;
        stats->list_len = KP;
        stats->n_splits = last_splits;
        stats->n_blocks = last_blocks;
        stats->first_pass = first_tier;
        stats->n_escalated_rows = n_escalated;
        stats->max_err_ratio = max_err_ratio;
        stats->n_first_pass_fail = n_first_fail > 0x7fffffff ? 0x7fffffff : (int32_t)n_first_fail;
        stats->wide_lists = wide_route ? long_pieces : 0;
        stats->n_spec_rows = n_spec > 0x7fffffff ? 0x7fffffff : (int32_t)n_spec;
        stats->n_range_rows = n_range;
        stats->n_range_pairs = n_range_pairs;
        stats->n_range_group_rows = n_range_group;
    }
    return KZ_OK;
}

extern "C" int kz_knn(kz_ctx* ctx, const kz_matrix* query, int64_t q_begin, int64_t q_count, const kz_matrix* index, int k,
                      int exclude_self, double* d_dist, int64_t* d_ind, kz_knn_stats* stats) {
    // (the matrices are logically const for the caller: kz_knn only attaches lazily built operand images to them)
    return kz_knn_impl(ctx, const_cast<kz_matrix*>(query), q_begin, q_count, const_cast<kz_matrix*>(index), k, exclude_self, nullptr,
                       -1, 0, d_dist, d_ind, stats, nullptr);
}

#include "kz_knn_dual.h"
