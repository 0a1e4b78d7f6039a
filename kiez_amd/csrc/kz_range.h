// RANGE RE-SEARCH (round 6): the exact float64 kernels on the PAIRS that can matter instead of on every pair.
//
// A row that no tier could certify used to end on the exact kernels against the WHOLE index -- n d multiply-adds per row: on
// data whose clusters are orders of magnitude tighter than its extent (tools/cliff_probe.py, last kind) a tenth of the rows, 265 of
// a 609 ms call on 200 k x 200 k x 200.  But the pass that failed the row has re-ranked its candidates in float64: the value tau of
// its k-th best CANDIDATE bounds the value of its k-th NEIGHBOUR from above (any k exact values do), so a neighbour y has
//     value(x, y) <= tau   =>   key(x, y) >= key(tau)   =>   key~(x, y) >= key(tau) - eps(x)
// for the approximate keys key~ of ANY tier with its rounding bound eps (kz_finalize_query).  The rows at or above that threshold
// are a few thousand rows of the row's own cluster, not the index.  So:
//   1. kz_range_thr_kernel: thr(x) = (key(tau) - eps) / scale in the fp16 tier's key units, rounded down;
//   2. the fp16 sweep kernel in its dual-pass build (kz_knn_h16.h, DUAL) over the failed rows: that build already logs, besides
//      its lists, every group of four keys whose maximum reaches a per-(query, index tile) threshold qnbias(q) + theta(tile) -- the
//      events of the shared sweep.  With theta = 0 and qnbias = thr it logs the range { y : key~(x, y) >= thr(x) }; its lists start
//      at +inf and stay empty.  No new sweep kernel;
//   3. kz_range_count_kernel / kz_range_scan_kernel / kz_range_fill_kernel: the log (unordered, groups of four) into one segment of
//      index rows per failed row;
//   4. kz_exact_pairs_kernel: the exact values of those pairs -- kz_exact_dist_rows_kernel's arithmetic operation for operation
//      (bit-identical values), the index rows gathered through the segment;
//   5. kz_exact_select_kernel over the segment: the k smallest by (value, row), written like every other route writes them.
// A row whose segment holds fewer than k entries (tau = +inf: it never had k candidates), a batch whose log overflows its buffer
// even at the smallest batch size: those rows go to the exact kernels against the whole index as before (`left`).  Correct for any
// data; what the data decides is how many pairs are left -- uniform noise inside eps of everything keeps them all.
// GROUPS (below, "GROUPED ranges"): from 2 048 failed rows on, rows of one tight cluster share the range of a representative row and
// become one dense block for the one-pair-per-lane kernel.  WHERE (kz_knn_impl): at the end of the ladder (all of the above), and
// early -- groups plus at most KZ_RANGE_EARLY_PER_ROW rows of their own -- on what a split-bf16 pass leaves, before the float32-operand
// tier sweeps the index for it.
// Reference: the brute-force search it replaces row by row, kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101.
#pragma once

constexpr int KZ_RANGE_MIN_ROWS = 128;      // fewer rows: the whole-index kernels (a sweep for a handful of rows costs more than it saves)
constexpr int KZ_RANGE_BATCH = 32768;       // failed rows per sweep
constexpr int KZ_RANGE_MIN_BATCH = 1024;    // a log that overflows at this batch size: the batch goes to `left`
constexpr int KZ_RANGE_PPW = 256;           // pairs per wave of kz_exact_pairs_kernel
constexpr int KZ_RANGE_EARLY_PER_ROW = 4096;   // the early call: at most this many rows outside the groups take their own ranges there

// thr [n_pad]: the threshold of batch row i in the fp16 tier's key units (+inf: nothing is logged -- pad rows, tau = +inf; diagnostic
// "abl" bit 8: every row, as if no row had k candidates);
// inf_floor [n_pad] = +inf: the floor the sweep's lists start from (they stay empty).  The bound is kz_finalize_query's, term by term.
__global__ void kz_range_thr_kernel(const double* __restrict__ tau, int nb, int64_t n_pad, const double* __restrict__ rowq,
                                    const double* __restrict__ qsqn, const double* __restrict__ y_hmax, const double* __restrict__ hscale,
                                    const double* __restrict__ ystats, int metric, double eps_mult, double gamma_acc,
                                    float* __restrict__ thr, float* __restrict__ inf_floor) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    inf_floor[i] = INFINITY;
    float t = INFINITY;
    if (i < nb) {
        const double tv = tau[i];
        if (tv < (double)INFINITY) {
            const double qc2 = rowq[i * 3 + 0], qh = rowq[i * 3 + 1], qr = rowq[i * 3 + 2];
            const double Yh = y_hmax[0], Ry = y_hmax[1], Yc2 = y_hmax[2];
            const double qc = sqrt(qc2), yc = sqrt(Yc2);
            const double ymax = ystats[0];
            const double raw2 = metric == KZ_COSINE ? 2.0 : qsqn[i] + ymax * ymax;
            const double eps = eps_mult * (qr * Yh + qh * Ry + qr * Ry + gamma_acc * (0.5 * Yc2 + qh * Yh) +
                                           1.1920928955078125e-07 * (qc + yc) * (qc + yc) + 1e-12 * (0.5 * Yc2 + qc2) + 1e-14 * raw2);
            const double key = 0.5 * (qc2 - (metric == KZ_COSINE ? 2.0 * tv : tv));
            const double want = (key - eps) / hscale[1];
            float f = (float)want;
            if ((double)f > want) f = nextafterf(f, -INFINITY);
            t = nextafterf(f, -INFINITY);   // (one more: the threshold errs towards MORE pairs)
            if (!(t == t)) t = -INFINITY;   // (never expected; a NaN threshold would log nothing)
        }
    }
    thr[i] = t;
}

// Keys of a logged group that really are in the range of their row (the kernel logs a group when its MAXIMUM is): index rows
// row0 .. row0 + 3 of the image = of the matrix (the sweep runs on the undealt image), query = batch row mt.y.
__device__ __forceinline__ int kz_range_group(const f32x4e kv, const i32x2e mt, const float* __restrict__ thr, int64_t n_i, int* row0_out) {
    const int ql = mt.x & 63, tg = mt.x >> 6;
    const int row0 = (tg >> 4) * KZ_TILE + ((tg >> 2) & 3) * 32 + (tg & 3) * 8 + 4 * (ql >> 5);
    const float t = thr[mt.y];
    int m = 0;
    m |= (kv.x >= t && row0 + 0 < n_i) ? 1 : 0;
    m |= (kv.y >= t && row0 + 1 < n_i) ? 2 : 0;
    m |= (kv.z >= t && row0 + 2 < n_i) ? 4 : 0;
    m |= (kv.w >= t && row0 + 3 < n_i) ? 8 : 0;
    *row0_out = row0;
    return m;
}
__global__ __launch_bounds__(256) void kz_range_count_kernel(const f32x4e* __restrict__ log_keys, const i32x2e* __restrict__ log_meta,
                                                             long long n_groups, const float* __restrict__ thr, int64_t n_i,
                                                             int* __restrict__ cnt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += (long long)gridDim.x * blockDim.x) {
        const i32x2e mt = log_meta[i];
        int row0;
        const int m = kz_range_group(log_keys[i], mt, thr, n_i, &row0);
        if (m) atomicAdd(cnt + mt.y, __popc(m));
    }
}
// seg_off [nb + 1]: exclusive prefix sums of cnt [nb] (one workgroup of 1024 threads; nb <= KZ_RANGE_BATCH)
__global__ __launch_bounds__(1024) void kz_range_scan_kernel(const int* __restrict__ cnt, int nb, long long* __restrict__ seg_off) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = part[tid] - s;
    for (int i = lo; i < hi; ++i) {
        seg_off[i] = run;
        run += cnt[i];
    }
    if (tid == 1023) seg_off[nb] = part[1023];
}
__global__ __launch_bounds__(256) void kz_range_fill_kernel(const f32x4e* __restrict__ log_keys, const i32x2e* __restrict__ log_meta,
                                                            long long n_groups, const float* __restrict__ thr, int64_t n_i,
                                                            const long long* __restrict__ seg_off, int* __restrict__ cur,
                                                            int* __restrict__ pair_idx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += (long long)gridDim.x * blockDim.x) {
        const i32x2e mt = log_meta[i];
        int row0;
        const int m = kz_range_group(log_keys[i], mt, thr, n_i, &row0);
        if (!m) continue;
        long long o = seg_off[mt.y] + atomicAdd(cur + mt.y, __popc(m));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (m & (1 << u)) pair_idx[o++] = row0 + u;
    }
}

// The exact values of the pairs (batch row r, index row pair_idx[p]) for p in [seg_off[r], seg_off[r + 1]): a wave takes
// pairs_per_wave consecutive pairs of the flat array -- finds the row of its first pair by bisection, keeps that row's elements in
// registers, and moves on to the next row where the segment ends.  A pair's arithmetic is kz_exact_dist_rows_kernel's (above):
// the lane's four fma per 256-element chunk in element order, the butterfly inside the lane group, the same last line.
template <int LPR, bool NORM, int NV = 1>
__global__ __launch_bounds__(256) void kz_exact_pairs_kernel(const long long* __restrict__ seg_off, int n_rows, const int* __restrict__ fail_list,
                                                             int batch0, int64_t q_begin, const float* __restrict__ qraw,
                                                             const float* __restrict__ yraw, const double* __restrict__ ynorm64,
                                                             const double* __restrict__ qsqn, const double* __restrict__ ysqn, int d, int metric,
                                                             const int* __restrict__ pair_idx, double* __restrict__ pair_val, int pairs_per_wave) {
    static_assert(NV == 1 || LPR == 64, "two chunks per lane: the whole wave owns one row");
    constexpr int G = 64 / LPR;
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
    const long long total = seg_off[n_rows];
    long long p0 = ((long long)blockIdx.x * 4 + wave) * pairs_per_wave;
    if (p0 >= total) return;
    const long long p1 = p0 + pairs_per_wave < total ? p0 + pairs_per_wave : total;
    int r;
    {   // the row of pair p0: seg_off[r] <= p0 < seg_off[r + 1]  (seg_off[0] = 0 <= p0 < total = seg_off[n_rows])
        int lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seg_off[mid] <= p0) lo = mid; else hi = mid;
        }
        r = lo;
    }
    // U pairs per lane group and step, two steps in flight, the index rows of the step after those already read (a pair is two
    // dependent loads: its index row's number, then the row).  Measured at d = 200, 4.1e7 pairs: 5.65 ms = 5.8 TB/s of gathered rows
    // -- what the memory side delivers for 800-byte rows in random order; one pair per step: 5.82 ms; the segments ordered so that
    // rows with the same range run together (their index rows then fit the memory-side cache, not an XCD's L2): no change.
    constexpr int U = NV == 1 ? 4 : 2;
    constexpr int STEP = U * G;
    struct Buf {
        float4 f[U][NV];
        double ys[U];
        double2 n0[U][NV], n1[U][NV];
    };
    while (p0 < p1) {
        while (seg_off[r + 1] <= p0) ++r;   // (empty segments)
        const long long seg_end = seg_off[r + 1];
        const long long rend = seg_end < p1 ? seg_end : p1;
        const int64_t qrow = q_begin + fail_list[batch0 + r];
        const double qs = qsqn[qrow];
        double qk[4 * NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};
            if (act[c]) {
                kz_row4(qraw + qrow * (int64_t)d, k0r[c], d, true, t);
                if (metric == KZ_COSINE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = t[e] / qs;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) qk[4 * c + e] = t[e];
        }
        auto load_idx = [&](long long i, int (&ix)[U]) {   // (pairs past the end: the last pair again, nothing is written for them)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long p = i + u * G + grp;
                ix[u] = pair_idx[p < rend ? p : rend - 1];
            }
        };
        auto issue = [&](const int (&ix)[U], Buf& b) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t yi = ix[u];
                if (NORM) {
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const double* row = ynorm64 + yi * (int64_t)d + k0r[c];
                        b.n0[u][c] = *reinterpret_cast<const double2*>(row);
                        b.n1[u][c] = *reinterpret_cast<const double2*>(row + 2);
                    }
                } else {
                    b.ys[u] = ysqn[yi];
#pragma unroll
                    for (int c = 0; c < NV; ++c) b.f[u][c] = *reinterpret_cast<const float4*>(yraw + yi * (int64_t)d + k0r[c]);
                }
            }
        };
        auto reduce = [&](long long i, const Buf& b) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double yv[4 * NV];
#pragma unroll
                for (int c = 0; c < NV; ++c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) yv[4 * c + e] = 0.0;
                    if (act[c]) {
                        if (NORM) {
                            yv[4 * c] = b.n0[u][c].x, yv[4 * c + 1] = b.n0[u][c].y, yv[4 * c + 2] = b.n1[u][c].x, yv[4 * c + 3] = b.n1[u][c].y;
                        } else {
                            const double yk[4] = {(double)b.f[u][c].x, (double)b.f[u][c].y, (double)b.f[u][c].z, (double)b.f[u][c].w};
                            if (metric == KZ_COSINE) {
                                const double rcp = 1.0 / b.ys[u];
                                const bool fin = (((unsigned long long)__double_as_longlong(rcp) >> 52) & 0x7ff) != 0x7ff;
#pragma unroll
                                for (int e = 0; e < 4; ++e) yv[4 * c + e] = fin ? kz_div_shared(yk[e], b.ys[u], rcp) : yk[e] / b.ys[u];
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) yv[4 * c + e] = yk[e];
                            }
                        }
                    }
                }
                double a = 0.0;
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    if (act[c]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a = fma(qk[4 * c + e], yv[4 * c + e], a);
                    }
                }
#pragma unroll
                for (int off = LPR >> 1; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
                double v;
                if (metric == KZ_COSINE)
                    v = fmin(fmax(1.0 - a, 0.0), 2.0);
                else
                    v = fmax((qs + b.ys[u]) - 2.0 * a, 0.0);
                const long long p = i + u * G + grp;
                if (sl == 0 && p < rend) pair_val[p] = v;
            }
        };
        Buf ba, bb;
        int ia[U], ib[U];
        load_idx(p0, ia);
        issue(ia, ba);
        load_idx(p0 + STEP, ib);
        for (long long i = p0; i < rend;) {   // (two steps in flight, the rows of a third known; the conditions are wave-uniform)
            issue(ib, bb);
            load_idx(i + 2 * STEP, ia);
            reduce(i, ba);
            i += STEP;
            if (i >= rend) break;
            issue(ia, ba);
            load_idx(i + 2 * STEP, ib);
            reduce(i, bb);
            i += STEP;
        }
        p0 = rend;
    }
}
static bool kz_range_shapes_ok(const kz_ctx* ctx, const kz_matrix* query, const kz_matrix* index) {
    const int d = (int)index->d;
    const int n_slices = index->kg / 4;
    return ctx->exact_rows >= 3 && ctx->precision != 1 && index->dtype == KZ_F32 && (d & 3) == 0 && d <= 512 && index->metric <= KZ_COSINE &&
           (((uintptr_t)query->raw | (uintptr_t)index->raw) & 15u) == 0 && n_slices >= 2 && n_slices <= 24 && query->kg == index->kg;
}
static void kz_launch_exact_pairs(kz_ctx* ctx, const long long* seg_off, int nb, const int* fl, int b0, int64_t q_begin, const kz_matrix* query,
                                  const kz_matrix* index, long long total, const int* pair_idx, double* pair_val) {
    const int d = (int)index->d, metric = index->metric;
    const bool norm = metric == KZ_COSINE && index->norm64 != nullptr;
    const long long waves = (total + KZ_RANGE_PPW - 1) / KZ_RANGE_PPW;
    const dim3 grid((unsigned)((waves + 3) / 4));
    const int lanes = (d + 3) >> 2;
#define KZ_EXACT_PAIRS(L, NVV)                                                                                                               \
    do {                                                                                                                                     \
        if (norm)                                                                                                                            \
            hipLaunchKernelGGL((kz_exact_pairs_kernel<L, true, NVV>), grid, dim3(256), 0, ctx->stream, seg_off, nb, fl, b0, q_begin,         \
                               (const float*)query->raw, (const float*)index->raw, index->norm64, query->sqn, index->sqn, d, metric,         \
                               pair_idx, pair_val, KZ_RANGE_PPW);                                                                            \
        else                                                                                                                                 \
            hipLaunchKernelGGL((kz_exact_pairs_kernel<L, false, NVV>), grid, dim3(256), 0, ctx->stream, seg_off, nb, fl, b0, q_begin,        \
                               (const float*)query->raw, (const float*)index->raw, (const double*)nullptr, query->sqn, index->sqn, d, metric, \
                               pair_idx, pair_val, KZ_RANGE_PPW);                                                                            \
    } while (0)
    if (lanes <= 8)
        KZ_EXACT_PAIRS(8, 1);
    else if (lanes <= 16)
        KZ_EXACT_PAIRS(16, 1);
    else if (lanes <= 32)
        KZ_EXACT_PAIRS(32, 1);
    else if (lanes <= 64)
        KZ_EXACT_PAIRS(64, 1);
    else
        KZ_EXACT_PAIRS(64, 2);
#undef KZ_EXACT_PAIRS
}

// The log of a sweep and what the sweep needs around it (allocated once per call of kz_range_rescue)
struct KzRangeLog {
    float* theta0 = nullptr;               // [index tiles * 128] zeros: the per-tile part of the kernel's threshold
    void* keys = nullptr;                  // [cap] groups of four keys
    void* meta = nullptr;                  // [cap] {lane | 16 tile + group, batch row}
    unsigned long long* counters = nullptr;   // [0] groups logged, [1] (as int) rows handed back
    long long cap = 0;
};
// The sweep of the rows of `qsub` (nb rows; tau [nb] on the device) against the whole index: thr / inff [n_pad] are filled, the log
// holds the groups, *n_groups their number; *over: the log overflowed or the fp16 tier did not run -- nothing usable was logged.
static int kz_range_sweep_rows(kz_ctx* ctx, kz_matrix* qsub, const double* tau, int nb, kz_matrix* index, const KzRangeLog& lg, float* thr,
                               float* inff, double* out_dist, int64_t* out_ind, unsigned long long* n_groups, bool* over) {
    if (ctx->chunk_rows > 0 && nb > ctx->chunk_rows) {   // (test knob "chunk_rows": the sweep must be ONE launch -- not taken)
        *n_groups = 0;
        *over = true;
        return KZ_OK;
    }
    int rc = kz_himage_ensure(qsub, index);
    if (rc != KZ_OK) return rc;
    const int64_t n_pad = qsub->n_tiles * KZ_TILE;
    hipLaunchKernelGGL(kz_range_thr_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, ctx->stream, tau, (ctx->abl & 8) ? 0 : nb, n_pad,
                       qsub->himg->rowq, qsub->sqn, index->himg->d_max, index->himg->center->d_scale, index->d_stats, index->metric, ctx->eps_scale,
                       kz_gamma_acc_h(index->kg), thr, inff);
    hipError_t e = hipMemsetAsync(lg.counters, 0, 8, ctx->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    KzDualPass dp;
    memset(&dp, 0, sizeof(dp));
    dp.probed = 1;
    dp.qpack = (const float*)qsub->himg->packed;
    dp.ypack = (const float*)index->himg->packed;
    dp.ybias = index->himg->bias;
    dp.theta = lg.theta0;
    dp.qnbias = thr;
    dp.qfloor = inff;
    dp.log_keys = lg.keys;
    dp.log_meta = lg.meta;
    dp.log_cnt = lg.counters;
    dp.log_cap = lg.cap;
    dp.raw_lists = 1;
    dp.no_q64 = 1;
    kz_knn_stats st;
    // (k = 1: lists of 16 -- the build with the most workgroups per CU; nothing is finalized, out_dist / out_ind are not written)
    rc = kz_knn_impl(ctx, qsub, 0, nb, index, 1, 0, nullptr, 0, 0, out_dist, out_ind, &st, &dp);
    if (rc != KZ_OK) return rc;
    e = hipMemcpyAsync(n_groups, lg.counters, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_knn: range re-search: reading the log counter failed");
        return KZ_ERR_HIP;
    }
    *over = dp.broken || st.first_pass != KZ_TIER_H || *n_groups > (unsigned long long)lg.cap;
    return KZ_OK;
}

// ---- GROUPED ranges ----------------------------------------------------------------------------------------------------------
// Thousands of failed rows are rows of a few tight clusters, and the rows of one cluster all have the same range -- the cluster.
// Sweeping the index once per ROW to find it (4.25 ms per 8 k rows, most of it the log's traffic) and gathering an index row per
// PAIR (5.8 TB/s, HBM-bound) does that work hundreds of times over.  Instead:
//   * every KZ_RG_STRIDE-th failed row is a REPRESENTATIVE; every failed row x is assigned to its nearest representative r(x)
//     (exact float64, one search of the failed rows against the representatives on the exact kernels);
//   * a neighbour y of x has |x - y| <= a(x) = the distance of x's k-th candidate, so |r - y| <= |r - x| + a(x): the range of r
//     with the radius R(r) = max over its rows of that sum holds every neighbour of every row assigned to r (cosine: the same on
//     the chords sqrt(2 dist) of the unit vectors);
//   * the representatives -- a 256th of the rows -- are swept and logged as above; a group (r, its rows M, its range B) is then a
//     DENSE block of |M| x |B| pairs for the one-pair-per-lane kernel (kz_exact_lanes.h, GATHER: a workgroup stages 64 rows of B
//     once for all rows of M; 3.8e10 pairs/s against the per-pair gather's 7e9); kz_exact_select_kernel picks every row's k best
//     of its group's block.
// Where the data has no such structure the radius R(r) covers most of the index: groups whose range holds more than half of it, or
// whose blocks do not fit the memory budget, are not taken -- their rows go through the per-row range above.  Values and order are
// the exact kernels' as everywhere.
constexpr int KZ_RG_MIN_ROWS = 2048;   // failed rows from which representatives are tried
constexpr int KZ_RG_STRIDE = 256;      // one representative per this many failed rows (at least KZ_XL_MIN_ROWS of them) ...
constexpr int KZ_RG_MAX_REPS = 1024;   // ... at most this many

// vals [n_rep][n]: the exact values between the representatives and the failed rows.  grp [n]: the nearest representative of failed
// row i (-1: no bound, the row is not grouped); rbits [n_rep]: the radius (bits of a positive float, rounded up) = max over the rows
// of a representative; mcount [n_rep]: its rows
// The values are the exact kernels' -- float64 evaluations of |x|^2 + |y|^2 - 2 x.y (cosine: 1 - x^.y^), a few 1e-13 of the squared
// norms away from the geometry the triangle inequality speaks about: every value that enters it is taken E larger, E = 4e-11 x the
// largest squared norm of either side (cosine: 1e-11) -- nothing next to a range's radius, ~100 x the round-off it covers.
__device__ __forceinline__ double kz_rg_slack(const double* __restrict__ stats_a, const double* __restrict__ stats_b, int metric) {
    if (metric == KZ_COSINE) return 1e-11;
    const double m = fmax(stats_a[0], stats_b[0]);   // (d_stats[0]: the largest row norm of a matrix)
    return 4e-11 * m * m;
}
__global__ void kz_rg_assign_kernel(const double* __restrict__ tau, const double* __restrict__ vals, int n_rep, int n, int metric,
                                    const double* __restrict__ stats_a, const double* __restrict__ stats_b, int* __restrict__ grp,
                                    int* __restrict__ rbits, int* __restrict__ mcount) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double E = kz_rg_slack(stats_a, stats_b, metric);
    const double tv = tau[i];
    int g = -1;
    if (tv < (double)INFINITY) {
        double best = INFINITY;
        g = 0;
        for (int j = 0; j < n_rep; ++j) {
            const double v = vals[(size_t)j * n + i];
            if (v < best) {
                best = v;
                g = j;
            }
        }
        const double s = metric == KZ_COSINE ? 2.0 : 1.0;   // (values: squared distances / cosine distances; chords: sqrt(s value))
        // (a row whose nearest representative is more than four times its own k-th candidate away -- a row of a cluster without a
        //  representative -- would blow up the radius for every row of that group: it stays on its own)
        if (best > 16.0 * tv) {
            g = -1;
        } else {
            const double r = (sqrt(s * (tv + E)) + sqrt(s * (best + E))) * (1.0 + 1e-9) + 1e-300;
            float rf = (float)r;
            if ((double)rf < r) rf = nextafterf(rf, INFINITY);
            atomicMax(rbits + g, __float_as_int(rf));
            atomicAdd(mcount + g, 1);
        }
    }
    grp[i] = g;
}
// tau_rep [n_rep]: the bound of a representative's range in value units (squared distance / cosine distance), +inf: no rows
__global__ void kz_rg_tau_kernel(const int* __restrict__ rbits, const int* __restrict__ mcount, int n_rep, int metric,
                                 const double* __restrict__ stats_a, const double* __restrict__ stats_b, double* __restrict__ tau_rep) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_rep) return;
    double t = INFINITY;
    if (mcount[j] > 0) {
        const double R = (double)__int_as_float(rbits[j]);
        t = R * R * (1.0 + 1e-6) / (metric == KZ_COSINE ? 2.0 : 1.0) + kz_rg_slack(stats_a, stats_b, metric);
        if (metric == KZ_COSINE && t > 2.0) t = 2.0;
    }
    tau_rep[j] = t;
}
// acc_of [n_rep]: the group's ordinal among the accepted groups, or -1.  Accepted rows: slot mate_off[a] + its arrival; the others
// are appended to rest / rest_tau.
__global__ void kz_rg_mates_kernel(const int* __restrict__ grp, const int* __restrict__ acc_of, const int* __restrict__ mate_off, int n,
                                   const int* __restrict__ fl, const double* __restrict__ tau, int* __restrict__ mcur, int* __restrict__ mates,
                                   int* __restrict__ slot_grp, int* __restrict__ rest, double* __restrict__ rest_tau, int* __restrict__ rest_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = grp[i];
    const int a = g >= 0 ? acc_of[g] : -1;
    if (a >= 0) {
        const int slot = mate_off[a] + atomicAdd(mcur + a, 1);
        mates[slot] = fl[i];
        slot_grp[slot] = a;
    } else {
        const int p = atomicAdd(rest_cnt, 1);
        rest[p] = fl[i];
        rest_tau[p] = tau[i];
    }
}
// per slot b (operand row b of the launch: the m-th row of an accepted group a, or padding behind its rows): where its values and its
// group's range rows lie.  Slots of a group start at mate_off[a] (a multiple of 16), slot_grp was set for the real rows only.
__global__ void kz_rg_slots_kernel(const int* __restrict__ slot_grp, const int* __restrict__ mate_off, const int* __restrict__ blen,
                                   const long long* __restrict__ ball_off, const long long* __restrict__ val_off, int n_slots,
                                   long long* __restrict__ seg_off, int* __restrict__ seg_len, long long* __restrict__ idx_off) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_slots) return;
    const int a = slot_grp[b];
    if (a < 0) {   // (padding: an empty segment; kz_exact_select_kernel skips the slot by its row number -1)
        seg_off[b] = idx_off[b] = 0;
        seg_len[b] = 0;
        return;
    }
    seg_off[b] = val_off[a] + (long long)(b - mate_off[a]) * blen[a];
    seg_len[b] = blen[a];
    idx_off[b] = ball_off[a];
}
// The grouped path over fl / tau [n_fail]: rows it answers are written (or, a segment short of k entries, appended to `left`); the
// others come back as rest / rest_tau [*n_rest] for the per-row path.  *pairs: pairs evaluated.
static int kz_range_grouped(kz_ctx* ctx, kz_matrix* query, int64_t q0, const int* fl, const double* tau, int n_fail, kz_matrix* index, int k,
                            int exclude_self, const int64_t* d_self_ids, double* out_dist, int64_t* out_ind, const KzRangeLog& lg, int* rest,
                            double* rest_tau, int* n_rest, int* left, int* left_cnt, long long* pairs, long long pair_cap) {
    const int metric = index->metric, d = (int)index->d;
    const int k_eff = k + (exclude_self ? 1 : 0);
    const int k_sel = (int)(k_eff < index->n ? k_eff : index->n);
    const size_t sel_lds = (size_t)k_sel * 12 + 16;
    const size_t row_bytes = (size_t)d * 4;
    int n_rep = (n_fail + KZ_RG_STRIDE - 1) / KZ_RG_STRIDE;
    if (n_rep > KZ_RG_MAX_REPS) n_rep = KZ_RG_MAX_REPS;
    if ((size_t)n_rep * n_fail > ((size_t)1 << 27)) n_rep = (int)(((size_t)1 << 27) / n_fail);   // (representatives x rows: at most 1 GiB of values)
    *pairs = 0;
    // every buffer of the path (released together, stream-ordered)
    int *rep_fl = nullptr, *ibuf = nullptr, *pair_idx = nullptr, *mates = nullptr, *slot_grp = nullptr;
    void *rep_raw = nullptr, *all_raw = nullptr;
    kz_matrix *rm = nullptr, *fm = nullptr;
    double *rv = nullptr, *tau_rep = nullptr, *vals = nullptr;
    float* thr = nullptr;   // [2 n_pad]: thresholds, +inf floors
    long long *seg_rep = nullptr, *lbuf = nullptr;
    auto release = [&]() {
        if (rm) kz_matrix_destroy(rm);
        if (fm) kz_matrix_destroy(fm);
        kz_pool_free(ctx, rep_fl, 0); kz_pool_free(ctx, ibuf, 0); kz_pool_free(ctx, pair_idx, 0); kz_pool_free(ctx, mates, 0);
        kz_pool_free(ctx, rep_raw, 0); kz_pool_free(ctx, all_raw, 0); kz_pool_free(ctx, rv, 0);
        kz_pool_free(ctx, tau_rep, 0); kz_pool_free(ctx, vals, 0); kz_pool_free(ctx, thr, 0);
        kz_pool_free(ctx, seg_rep, 0); kz_pool_free(ctx, lbuf, 0);
    };
    // "nothing grouped": every row is the per-row path's
    auto give_up = [&](int rc_in) -> int {
        release();
        if (rc_in != KZ_OK && rc_in != KZ_ERR_NOMEM) return rc_in;
        hipError_t e0 = hipMemcpyAsync(rest, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(rest_tau, tau, (size_t)n_fail * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(ctx->stream);
        if (e0 != hipSuccess) {
            kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e0));
            return KZ_ERR_HIP;
        }
        *n_rest = n_fail;
        return KZ_OK;
    };
    if (n_rep < KZ_XL_MIN_ROWS) return give_up(KZ_OK);
    const int64_t stride = n_fail / n_rep;
    const int64_t rep_pad = (int64_t)((n_rep + KZ_TILE - 1) / KZ_TILE) * KZ_TILE;
    // ibuf: grp [n_fail] | rbits | mcount | cnt | cur | acc_of | mate_off | blen | mcur | iota [n_rep each] | rest_cnt [1]
    int rc = kz_pool_alloc(ctx, (size_t)n_rep * sizeof(int), (void**)&rep_fl);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, ((size_t)n_fail + 9 * (size_t)n_rep + 4) * sizeof(int), (void**)&ibuf);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_rep * row_bytes, &rep_raw);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * row_bytes, &all_raw);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_rep * (size_t)n_fail * 8, (void**)&rv);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_rep * 8, (void**)&tau_rep);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)rep_pad * 2 * sizeof(float), (void**)&thr);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)(n_rep + 1) * sizeof(long long), (void**)&seg_rep);
    // (slots: a group's rows, padded to whole blocks of 16 operand rows -- at most n_fail + 16 n_rep of them)
    const size_t slots_cap = (size_t)n_fail + 16 * (size_t)n_rep;
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, slots_cap * 3 * sizeof(int), (void**)&mates);   // mates | slot_grp | seg_len
    // lbuf: ball_off [n_rep] | val_off [n_rep] | seg_off [slots] | idx_off [slots] | group descriptors [n_rep]
    if (rc == KZ_OK)
        rc = kz_pool_alloc(ctx, (2 * (size_t)n_rep + 2 * slots_cap + 2) * sizeof(long long) + (size_t)n_rep * sizeof(KzXlGroup), (void**)&lbuf);
    if (rc != KZ_OK) return give_up(rc);
    int *grp = ibuf, *rbits = ibuf + n_fail, *mcount = rbits + n_rep, *cnt = mcount + n_rep, *cur = cnt + n_rep, *acc_of = cur + n_rep,
        *mate_off = acc_of + n_rep, *blen = mate_off + n_rep, *mcur = blen + n_rep, *iota = mcur + n_rep, *rest_cnt = iota + n_rep;
    long long *ball_off = lbuf, *val_off = lbuf + n_rep, *seg_off = val_off + n_rep, *idx_off = seg_off + slots_cap + 1;
    KzXlGroup* d_groups = (KzXlGroup*)(idx_off + slots_cap + 1);
    slot_grp = mates + slots_cap;
    int* seg_len = slot_grp + slots_cap;
    hipError_t e = hipMemsetAsync(rbits, 0, (9 * (size_t)n_rep + 4) * sizeof(int), ctx->stream);
    if (e != hipSuccess) return give_up(KZ_ERR_NOMEM);
    // ---- representatives; every failed row's nearest one (exact values representatives x failed rows, the dense kernels) --------
    hipLaunchKernelGGL(kz_strided_pick_kernel, dim3((n_rep + 255) / 256), dim3(256), 0, ctx->stream, fl, n_rep, stride, rep_fl);
    hipLaunchKernelGGL(kz_iota_kernel, dim3((n_rep + 255) / 256), dim3(256), 0, ctx->stream, iota, n_rep);
    hipLaunchKernelGGL(kz_gather_rows_kernel, dim3(n_rep), dim3(256), 0, ctx->stream, (const char*)query->raw, rep_fl, q0, n_rep, (int64_t)row_bytes,
                       (char*)rep_raw, (int64_t*)nullptr, (const int64_t*)nullptr);
    hipLaunchKernelGGL(kz_gather_rows_kernel, dim3(n_fail), dim3(256), 0, ctx->stream, (const char*)query->raw, fl, q0, n_fail, (int64_t)row_bytes,
                       (char*)all_raw, (int64_t*)nullptr, (const int64_t*)nullptr);
    rc = kz_matrix_create(ctx, rep_raw, 2, n_rep, d, query->dtype, query->metric, &rm);
    if (rc == KZ_OK) rc = kz_matrix_create(ctx, all_raw, 2, n_fail, d, query->dtype, query->metric, &fm);
    if (rc != KZ_OK) return give_up(rc);
    {
        bool took = false;
        rc = kz_launch_exact_lanes(ctx, iota, 0, n_rep, 0, rm, fm, metric, rv, &took);
        if (rc != KZ_OK) return give_up(rc);
        if (!took && !kz_launch_exact_rows(ctx, iota, 0, n_rep, 0, rm, fm, metric, rv))
            hipLaunchKernelGGL(kz_exact_dist_kernel<float>, dim3((unsigned)((n_fail + 3) / 4), n_rep), dim3(256), 0, ctx->stream, iota, 0, (int64_t)0,
                               (const float*)rm->raw, (const float*)fm->raw, rm->sqn, fm->sqn, (int64_t)n_fail, d, metric, index->mink_p, rv,
                               (const int*)nullptr);
    }
    hipLaunchKernelGGL(kz_rg_assign_kernel, dim3((n_fail + 255) / 256), dim3(256), 0, ctx->stream, tau, rv, n_rep, n_fail, metric, fm->d_stats,
                       index->d_stats, grp, rbits, mcount);
    hipLaunchKernelGGL(kz_rg_tau_kernel, dim3((n_rep + 255) / 256), dim3(256), 0, ctx->stream, rbits, mcount, n_rep, metric, fm->d_stats, index->d_stats,
                       tau_rep);
    // ---- the representatives' ranges -----------------------------------------------------------------------------------------
    unsigned long long n_groups = 0;
    bool over = false;
    rc = kz_range_sweep_rows(ctx, rm, tau_rep, n_rep, index, lg, thr, thr + rep_pad, out_dist, out_ind, &n_groups, &over);
    if (rc != KZ_OK) return give_up(rc);
    if (over) return give_up(KZ_OK);
    const int gb = (int)((n_groups + 255) / 256 < 8192 ? (n_groups + 255) / 256 : 8192);
    if (n_groups > 0)
        hipLaunchKernelGGL(kz_range_count_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const f32x4e*)lg.keys, (const i32x2e*)lg.meta, (long long)n_groups,
                           thr, index->n, cnt);
    hipLaunchKernelGGL(kz_range_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, n_rep, seg_rep);
    std::vector<int> h_cnt(n_rep), h_m(n_rep);
    long long rep_total = 0;
    e = hipMemcpyAsync(h_cnt.data(), cnt, (size_t)n_rep * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_m.data(), mcount, (size_t)n_rep * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&rep_total, seg_rep + n_rep, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return give_up(KZ_ERR_NOMEM);
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return give_up(KZ_ERR_NOMEM);
    if ((size_t)rep_total * 4 > mem_free / 8) return give_up(KZ_OK);
    // ---- which groups are taken: at least a block of rows for the dense kernel, a range of at most half the index, all blocks
    //      within a third of the free memory (and 2^31 pairs) ------------------------------------------------------------------
    std::vector<int> h_acc(n_rep, -1), h_moff, h_blen;
    std::vector<long long> h_boff, h_voff;
    std::vector<KzXlGroup> h_groups;
    long long budget = (long long)(mem_free / 3 / 8);
    if (budget > (1ll << 31)) budget = 1ll << 31;
    if (pair_cap >= 0 && budget > pair_cap) budget = pair_cap;
    long long tot_pairs = 0;
    int n_slots = 0, rows_max = 0, q_max = 0;
    {
        long long boff = 0;
        for (int j = 0; j < n_rep; ++j) {
            const long long pj = (long long)h_m[j] * h_cnt[j];
            if (h_m[j] >= KZ_XL_MIN_ROWS && h_cnt[j] >= k_sel && (int64_t)h_cnt[j] * 2 <= index->n && tot_pairs + pj <= budget) {
                h_acc[j] = (int)h_moff.size();
                h_moff.push_back(n_slots);
                h_blen.push_back(h_cnt[j]);
                h_boff.push_back(boff);
                h_voff.push_back(tot_pairs);
                KzXlGroup g;
                g.nb = h_m[j];
                g.n_rows = h_cnt[j];
                g.qd_off = n_slots;
                g.gather_off = boff;
                g.val_off = tot_pairs;
                h_groups.push_back(g);
                n_slots += (h_m[j] + 15) / 16 * 16;
                tot_pairs += pj;
                if (h_cnt[j] > rows_max) rows_max = h_cnt[j];
                if (h_m[j] > q_max) q_max = h_m[j];
            }
            boff += h_cnt[j];   // (the range lists of ALL representatives are filled: seg_rep)
        }
    }
    const int n_acc = (int)h_moff.size();
    if (n_acc == 0) return give_up(KZ_OK);
    rc = kz_pool_alloc(ctx, (size_t)rep_total * 4 + 4, (void**)&pair_idx);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)tot_pairs * 8 + 8, (void**)&vals);
    if (rc != KZ_OK) return give_up(rc);
    // (pageable host arrays: waited for below, before the vectors go out of scope)
    e = hipMemcpyAsync(acc_of, h_acc.data(), (size_t)n_rep * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(mate_off, h_moff.data(), (size_t)n_acc * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(blen, h_blen.data(), (size_t)n_acc * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ball_off, h_boff.data(), (size_t)n_acc * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(val_off, h_voff.data(), (size_t)n_acc * sizeof(long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_groups, h_groups.data(), (size_t)n_acc * sizeof(KzXlGroup), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(mates, 0xff, (size_t)n_slots * sizeof(int), ctx->stream);      // (-1: padding rows)
    if (e == hipSuccess) e = hipMemsetAsync(slot_grp, 0xff, (size_t)n_slots * sizeof(int), ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return give_up(KZ_ERR_NOMEM);
    if (n_groups > 0)
        hipLaunchKernelGGL(kz_range_fill_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const f32x4e*)lg.keys, (const i32x2e*)lg.meta, (long long)n_groups,
                           thr, index->n, seg_rep, cur, pair_idx);
    hipLaunchKernelGGL(kz_rg_mates_kernel, dim3((n_fail + 255) / 256), dim3(256), 0, ctx->stream, grp, acc_of, mate_off, n_fail, fl, tau, mcur, mates, slot_grp,
                       rest, rest_tau, rest_cnt);
    hipLaunchKernelGGL(kz_rg_slots_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, ctx->stream, slot_grp, mate_off, blen, ball_off, val_off, n_slots,
                       seg_off, seg_len, idx_off);
    // the groups = dense blocks, ONE launch: their rows against the rows of their ranges, one pair per lane (kz_exact_lanes.h, GATHER)
    {
        bool took = false;
        rc = kz_launch_exact_lanes(ctx, mates, 0, n_slots, q0, query, index, metric, vals, &took, nullptr, nullptr, pair_idx, d_groups, n_acc, rows_max,
                                   q_max);
        if (rc == KZ_OK && !took) {
            kz_set_error("kz_knn: internal: the dense kernel refused the groups of the range re-search");
            rc = KZ_ERR_INVALID;
        }
        if (rc != KZ_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            release();
            return rc;
        }
    }
    hipLaunchKernelGGL(kz_exact_select_kernel<float>, dim3(n_slots), dim3(256), sel_lds, ctx->stream, mates, 0, q0, (const double*)vals, (const int*)pair_idx,
                       (int64_t)0, index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p, out_dist, out_ind, (const int*)nullptr,
                       (const long long*)seg_off, left, left_cnt, (const long long*)idx_off, (const int*)seg_len);
    int h_rest = 0;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&h_rest, rest_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release();
    if (e != hipSuccess) {
        kz_set_error("kz_knn: grouped range re-search failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    *n_rest = h_rest;
    *pairs = tot_pairs;
    return KZ_OK;
}

// fl / tau [n_fail] (device; the caller's copies -- not in the context's scratch block, which the sweep below re-carves): rows
// q0 + fl[i] of `query`.  Results of the rows it answers go to out_dist / out_ind at row fl[i]; the others are appended to
// left [n_fail] (device), *n_left = their number.  Ends synchronised with the stream.  n_pairs_out: pairs evaluated (statistics).
static int kz_range_rescue(kz_ctx* ctx, kz_matrix* query, int64_t q0, const int* fl, const double* tau, int n_fail, kz_matrix* index, int k,
                           int exclude_self, const int64_t* d_self_ids, double* out_dist, int64_t* out_ind, int* left, int* n_left,
                           long long* n_pairs_out, long long* n_grouped_out, bool grouped_only = false, int per_row_max = 0) {
    *n_left = 0;
    if (n_pairs_out) *n_pairs_out = 0;
    if (n_grouped_out) *n_grouped_out = 0;
    const int metric = index->metric;
    const int k_eff = k + (exclude_self ? 1 : 0);
    const int k_sel = (int)(k_eff < index->n ? k_eff : index->n);
    const size_t sel_lds = (size_t)k_sel * 12 + 16;
    const size_t row_bytes = (size_t)query->d * 4;
    const int64_t y_pad = index->n_tiles * KZ_TILE;
    // the log: 24 bytes per group of four keys -- at most a quarter of the free memory, at most 2^28 groups (6 GiB)
    size_t mem_free = 0, mem_total = 0;
    KZ_HIP(hipMemGetInfo(&mem_free, &mem_total));
    long long log_cap = (long long)(mem_free / 4 / 24);
    if (log_cap > (1ll << 28)) log_cap = 1ll << 28;
    if ((ctx->abl & 4) && log_cap > 4096) log_cap = 4096;   // (diagnostic: a log that overflows -- every batch is handed back)
    if (log_cap < (1ll << 16) && !(ctx->abl & 4)) {   // (no room for a log worth sweeping for: everything stays with the caller)
        KZ_HIP(hipMemcpyAsync(left, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        KZ_HIP(hipStreamSynchronize(ctx->stream));
        *n_left = n_fail;
        return KZ_OK;
    }
    if (metric == KZ_COSINE && n_fail >= 1024) {   // (the normalised float64 rows of the index, once -- as the whole-index kernels take
        const int rc = kz_matrix_norm64(index);    //  them; for a few hundred rows the image -- 800 MB for 500 k x 200 -- costs more than
        if (rc != KZ_OK) return rc;                //  the divisions it saves: the kernels divide the raw rows, the same values)
    }
    KzRangeLog lg;
    int* rest = nullptr;        // the rows the grouped path leaves to the per-row path, their bounds
    double* rest_tau = nullptr;
    // The log is sized for what is about to be swept -- 65 536 groups per representative (a range beyond that is not taken anyway),
    // 4 096 per row of a per-row batch -- and grown when the next sweep needs more: a 6 GiB log for every call (the limit above)
    // kept the context's buffer cache turning over gigabytes.
    auto ensure_log = [&](long long want) -> int {
        if (want > log_cap) want = log_cap;
        if (want < (1ll << 16) && !(ctx->abl & 4)) want = 1ll << 16;
        if (lg.cap >= want) return KZ_OK;
        kz_pool_free(ctx, lg.keys, 0);
        kz_pool_free(ctx, lg.meta, 0);
        lg.keys = lg.meta = nullptr;
        lg.cap = 0;
        int rc2 = kz_pool_alloc(ctx, (size_t)want * 16, &lg.keys);
        if (rc2 == KZ_OK) rc2 = kz_pool_alloc(ctx, (size_t)want * 8, &lg.meta);
        if (rc2 == KZ_OK) lg.cap = want;
        return rc2;
    };
    const bool try_groups = n_fail >= KZ_RG_MIN_ROWS && !(ctx->abl & 16);
    int rc = kz_pool_alloc(ctx, (size_t)y_pad * 4, (void**)&lg.theta0);
    if (rc == KZ_OK)
        rc = ensure_log(try_groups ? (long long)(n_fail / KZ_RG_STRIDE + 1 < KZ_RG_MAX_REPS ? n_fail / KZ_RG_STRIDE + 1 : KZ_RG_MAX_REPS) * 65536
                                   : (long long)(n_fail < KZ_RANGE_BATCH ? n_fail : KZ_RANGE_BATCH) * 4096);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, 64, (void**)&lg.counters);
    auto release_all = [&]() {
        kz_pool_free(ctx, lg.theta0, 0);
        kz_pool_free(ctx, lg.keys, 0);
        kz_pool_free(ctx, lg.meta, 0);
        kz_pool_free(ctx, lg.counters, 0);
        kz_pool_free(ctx, rest, 0);
        kz_pool_free(ctx, rest_tau, 0);
    };
    if (rc != KZ_OK) {
        release_all();
        if (rc != KZ_ERR_NOMEM) return rc;
        KZ_HIP(hipMemcpyAsync(left, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
        KZ_HIP(hipStreamSynchronize(ctx->stream));
        *n_left = n_fail;
        return KZ_OK;
    }
    hipError_t e = hipMemsetAsync(lg.theta0, 0, (size_t)y_pad * 4, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(lg.counters, 0, 64, ctx->stream);
    if (e != hipSuccess) {
        release_all();
        kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    int* left_cnt = (int*)(lg.counters + 1);
    long long pairs_total = 0;
    auto read_back = [&](void* dst, const void* src, size_t bytes) -> hipError_t {   // (behind everything queued on the context's stream)
        const hipError_t e1 = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
        return e1 != hipSuccess ? e1 : hipStreamSynchronize(ctx->stream);
    };
    // ---- groups first: rows of one tight cluster share a representative's range ("abl" bit 16: off) ------------------------------
    // (grouped_only -- the EARLY call, rows that have tiers left to try: groups only, and only while their blocks hold at most an eighth
    //  of the pairs the whole index would; the rows that are not grouped come back in `left` for the next tier)
    if (try_groups) {
        rc = kz_pool_alloc(ctx, (size_t)n_fail * sizeof(int), (void**)&rest);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_fail * 8, (void**)&rest_tau);
        if (rc == KZ_OK) {
            int n_rest = 0;
            long long gp = 0;
            rc = kz_range_grouped(ctx, query, q0, fl, tau, n_fail, index, k, exclude_self, d_self_ids, out_dist, out_ind, lg, rest, rest_tau, &n_rest,
                                  left, left_cnt, &gp, grouped_only ? (long long)((double)n_fail * (double)index->n / 8.0) : -1);
            if (rc != KZ_OK) {
                release_all();
                return rc;
            }
            pairs_total += gp;
            if (n_grouped_out) *n_grouped_out = n_fail - n_rest;
            fl = rest;
            tau = rest_tau;
            n_fail = n_rest;
        } else if (rc != KZ_ERR_NOMEM) {
            release_all();
            return rc;
        }
    }
    // (... unless only a few are left: the next tier would sweep the whole index for them -- 3.3 ms per launch of the float32-operand
    //  kernel on 200 k rows whatever the row count -- where their own ranges cost a fraction of that)
    if (grouped_only && fl == rest && n_fail <= per_row_max) grouped_only = false;
    if (grouped_only) {   // (what the groups did not take: handed back behind the rows their selection handed back)
        if (fl != rest) {   // (no group was tried: nothing was answered)
            release_all();
            KZ_HIP(hipMemcpyAsync(left, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
            KZ_HIP(hipStreamSynchronize(ctx->stream));
            *n_left = n_fail;
            return KZ_OK;
        }
        int have = 0;
        hipError_t e2 = read_back(&have, left_cnt, sizeof(int));
        if (e2 == hipSuccess && n_fail > 0) e2 = hipMemcpyAsync(left + have, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
        release_all();
        if (e2 != hipSuccess) {
            kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e2));
            return KZ_ERR_HIP;
        }
        *n_left = have + n_fail;
        if (n_pairs_out) *n_pairs_out = pairs_total;
        return KZ_OK;
    }
    int batch = n_fail < KZ_RANGE_BATCH ? n_fail : KZ_RANGE_BATCH;
    if (ctx->chunk_rows > 0 && batch > ctx->chunk_rows) batch = (int)ctx->chunk_rows;
    if (n_fail > 0) {
        rc = ensure_log((long long)batch * 4096);
        if (rc != KZ_OK) {   // (no room for the per-row log: the rows stay with the caller)
            int have = 0;
            hipError_t e2 = rc == KZ_ERR_NOMEM ? read_back(&have, left_cnt, sizeof(int)) : hipErrorUnknown;
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(left + have, fl, (size_t)n_fail * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
            release_all();
            if (e2 != hipSuccess) return rc == KZ_ERR_NOMEM ? KZ_ERR_HIP : rc;
            *n_left = have + n_fail;
            if (n_pairs_out) *n_pairs_out = pairs_total;
            return KZ_OK;
        }
    }
    for (int b0 = 0; b0 < n_fail;) {
        const int nb = n_fail - b0 < batch ? n_fail - b0 : batch;
        // ---- the batch's rows as a matrix of their own, their thresholds, the sweep ------------------------------------
        void* sub_raw = nullptr;
        kz_matrix* qsub = nullptr;
        float *thr = nullptr, *inff = nullptr;
        int* cnt = nullptr;   // [nb] pairs per row, then [nb] fill cursors
        long long* seg_off = nullptr;
        int* pair_idx = nullptr;
        double* pair_val = nullptr;
        auto release = [&]() {
            if (qsub) kz_matrix_destroy(qsub);
            kz_pool_free(ctx, sub_raw, 0);
            kz_pool_free(ctx, thr, 0);
            kz_pool_free(ctx, inff, 0);
            kz_pool_free(ctx, cnt, 0);
            kz_pool_free(ctx, seg_off, 0);
            kz_pool_free(ctx, pair_idx, 0);
            kz_pool_free(ctx, pair_val, 0);
        };
        auto fail = [&](int code) {
            release();
            release_all();
            return code;
        };
        const int64_t n_pad = (int64_t)((nb + KZ_TILE - 1) / KZ_TILE) * KZ_TILE;
        rc = kz_pool_alloc(ctx, (size_t)nb * row_bytes, &sub_raw);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&thr);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&inff);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)nb * 2 * sizeof(int), (void**)&cnt);
        if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)(nb + 1) * sizeof(long long), (void**)&seg_off);
        if (rc != KZ_OK) return fail(rc);
        hipLaunchKernelGGL(kz_gather_rows_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const char*)query->raw, fl + b0, q0, nb, (int64_t)row_bytes,
                           (char*)sub_raw, (int64_t*)nullptr, (const int64_t*)nullptr);
        rc = kz_matrix_create(ctx, sub_raw, 2, nb, query->d, query->dtype, query->metric, &qsub);
        if (rc != KZ_OK) return fail(rc);
        e = hipMemsetAsync(cnt, 0, (size_t)nb * 2 * sizeof(int), ctx->stream);
        if (e != hipSuccess) {
            kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e));
            return fail(KZ_ERR_HIP);
        }
        unsigned long long n_groups = 0;
        bool over = false;
        rc = kz_range_sweep_rows(ctx, qsub, tau + b0, nb, index, lg, thr, inff, out_dist, out_ind, &n_groups, &over);
        if (rc != KZ_OK) return fail(rc);
        long long total = 0;
        if (!over && n_groups > 0) {
            const int gb = (int)((n_groups + 255) / 256 < 8192 ? (n_groups + 255) / 256 : 8192);
            hipLaunchKernelGGL(kz_range_count_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const f32x4e*)lg.keys, (const i32x2e*)lg.meta,
                               (long long)n_groups, thr, index->n, cnt);
        }
        if (!over) {
            hipLaunchKernelGGL(kz_range_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, cnt, nb, seg_off);
            if (hipGetLastError() != hipSuccess || read_back(&total, seg_off + nb, 8) != hipSuccess) {
                kz_set_error("kz_knn: range re-search: counting the pairs failed");
                return fail(KZ_ERR_HIP);
            }
            // (idx + value: 12 bytes a pair, at most a quarter of what is free now)
            size_t f2 = 0, t2 = 0;
            KZ_HIP(hipMemGetInfo(&f2, &t2));
            if ((size_t)total * 12 > f2 / 4) over = true;
        }
        if (over) {
            release();
            if (batch > KZ_RANGE_MIN_BATCH && nb > KZ_RANGE_MIN_BATCH) {   // the same rows again, fewer per sweep
                batch = batch / 4 > KZ_RANGE_MIN_BATCH ? batch / 4 : KZ_RANGE_MIN_BATCH;
                continue;
            }
            // (its rows stay with the caller: appended to `left` by a copy -- left_cnt is only read by this stream)
            int have = 0;
            hipError_t e2 = read_back(&have, left_cnt, sizeof(int));
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(left + have, fl + b0, (size_t)nb * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
            have += nb;
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(left_cnt, &have, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
            if (e2 != hipSuccess) {
                release_all();
                kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e2));
                return KZ_ERR_HIP;
            }
            b0 += nb;
            continue;
        }
        if (total > 0) {
            rc = kz_pool_alloc(ctx, (size_t)total * 4, (void**)&pair_idx);
            if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)total * 8, (void**)&pair_val);
            if (rc != KZ_OK) return fail(rc);
            const int gb = (int)((n_groups + 255) / 256 < 8192 ? (n_groups + 255) / 256 : 8192);
            hipLaunchKernelGGL(kz_range_fill_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const f32x4e*)lg.keys, (const i32x2e*)lg.meta,
                               (long long)n_groups, thr, index->n, seg_off, cnt + nb, pair_idx);
            kz_launch_exact_pairs(ctx, seg_off, nb, fl, b0, q0, query, index, total, pair_idx, pair_val);
        }
        hipLaunchKernelGGL(kz_exact_select_kernel<float>, dim3(nb), dim3(256), sel_lds, ctx->stream, fl, b0, q0, (const double*)pair_val,
                           (const int*)pair_idx, (int64_t)0, index->n, k, exclude_self ? 1 : 0, d_self_ids, metric, index->mink_p, out_dist, out_ind,
                           (const int*)nullptr, (const long long*)seg_off, left, left_cnt);
        e = hipGetLastError();
        if (e != hipSuccess) {
            kz_set_error("kz_knn: range re-search: %s", hipGetErrorString(e));
            return fail(KZ_ERR_HIP);
        }
        pairs_total += total;
        release();   // (stream-ordered pool: the launches above have the buffers)
        b0 += nb;
    }
    int have = 0;
    e = read_back(&have, left_cnt, sizeof(int));
    release_all();
    if (e != hipSuccess) {
        kz_set_error("kz_knn: range re-search failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    *n_left = have;
    if (n_pairs_out) *n_pairs_out = pairs_total;
    return KZ_OK;
}
