// Dual pass: both directions of a kNN search between two matrices from ONE sweep of the distance matrix
// (included by kz_knn.hip).
//
// The hubness reductions need the neighbours of every target among the sources (fit, kiez/hubness_reduction/base.py:
// 48-58) AND of every source among the targets (kneighbors, base.py:95-112): the reference runs two brute-force searches
// over the same n_s x n_t distances.  Here the fused fp16 kernel sweeps  A (queries) x B (index)  once and reports
//   * per row of A its K' best rows of B -- the candidate lists it always keeps -- and
//   * per row t of B every row q of A whose key for the REVERSE direction,
//         key'(t, q) = q_h.t_h + bias(q) = acc(q, t) - bias(t) + bias(q),
//     reaches a threshold tau(t): the "events" of t (kz_knn_epi3.h "Dual pass").
// The test the sweep itself performs is coarser and nearly free: B's rows are SORTED by their threshold, so the 128 rows of
// a tile have almost the same one, and a group of four keys is logged when its maximum (which the list scan has already
// computed) reaches the tile's smallest threshold -- one extra compare per tile on the common path.
// tau(t) is fixed before the sweep: the (k+1)-th best key of t against a SAMPLE of A (every s-th tile of A's fp16 image,
// copied into a small image and swept by the ordinary kernel with B as the query side: 1/s of a full sweep).  By
// construction about k s rows of A pass tau(t), whatever the data looks like (the count of population members above
// the k-th order statistic of a sample is negative binomial: mean k (s - 1), deviation sqrt(k) s), so the event
// buffers are small and of predictable size.
//
// After the sweep: kz_dual_scatter_kernel files the logged groups per index row (per-key test, atomic slot),
// kz_dual_select_kernel keeps the K' best events of every row as an ordinary candidate list, and the ordinary finalize
// kernel certifies and re-ranks it in float64 with B as the query side -- with one change: rows of A outside the list are
// bounded by max(K'-th list key, tau(t)) instead of the K'-th list key alone.  Rows of B whose event buffer overflowed or
// whose list cannot be certified are searched again the ordinary way (kz_escalate_rows).  The result is the float64
// neighbour order in both directions, as from two separate kz_knn calls.
#pragma once

// Event counters of the rows owning events: ints per counter.  (Round 5, measured and left at 1: the event log is nearly sorted by
// index tile, so the scatter kernel's threads that run together hit the counters of the same 128 rows -- but ONE counter per
// 128-byte line, KZ_EVC = 32, changed nothing: C3 scatter 4.7 ms / select 5.2 ms either way.  The atomics are not what these two
// kernels wait for.)
constexpr int KZ_EVC = 1;

// ---- thresholds from the sample sweep --------------------------------------------------------------------------------
// One wave per row t of B.  The row's lists from the sample sweep hold pieces x K' <= 256 entries; tau = the `rank`-th best of
// them (by key, ties by entry order; rank = k + 1: the sample rows are rows of A, so k rows above tau are there by
// construction, and about rank x stride in all -- the list length K' is for the certification's margin, not for the threshold).
// theta(t) = tau + bias(t) - margin, rounded DOWN to float32.  The certification of the reverse direction needs: a (q, t) pair
// that is NOT filed as an event has  acc - bias(t) + bias(q) < tau  in exact arithmetic on the float32 values.  A pair can miss
// the events in two ways, with Mx = S^2 (Ah Bh + Ac2 + Bc2) >= |acc|, |theta|, |bias(q)|:
//   * the per-key test of kz_dual_scatter_kernel fails: fl(acc - theta) < -bias(q), hence acc - theta < -bias(q) + 2^-24 |acc - theta|
//     <= -bias(q) + 2^-23 Mx;
//   * its group fails the kernel's per-tile test  max(group) >= fl(-bias(q) + theta_min)  (theta_min <= theta(t): the tile's
//     smallest threshold): acc < (-bias(q) + theta_min)(1 + 2^-24), hence acc - theta(t) < -bias(q) + 2^-23 Mx as well.
// Either way  acc + bias(q) < theta + 2^-23 Mx <= tau + bias(t)  once margin >= 2^-23 Mx; the kernel uses 2^-21 Mx (four times
// that; in key units a few 1e-7 of the squared scale -- no visible effect on the event counts).
__global__ __launch_bounds__(256) void kz_dual_theta_kernel(const float* __restrict__ in_key, const int* __restrict__ in_idx,
                                                            KzListLayout lay, int KP, int rank, int64_t n_b, int64_t n_b_pad,
                                                            const float* __restrict__ bias_b, const double* __restrict__ a_hmax,
                                                            const double* __restrict__ b_hmax, const double* __restrict__ hscale,
                                                            float* __restrict__ theta, float* __restrict__ floor_,
                                                            const int* __restrict__ idx_to_row, int sev_cap, int* __restrict__ sev_cnt,
                                                            uint2* __restrict__ sev) {
    // NESTED sample (sev_cnt != nullptr; below "NESTED"): the lists ARE this row's events among the sample rows -- the main sweep no
    // longer visits those.  Two changes: (1) the entries at or above the threshold are written out as events (key, matrix row of
    // the sample row); (2) the BOUND on the rows that are not events (floor_, read by the certification) is raised to the smallest
    // key of any FULL list: a full list may have evicted sample rows, and those lie at or below its smallest key -- which can
    // exceed the threshold when a row's near sample rows crowd one range.  The event threshold itself (theta) stays at the rank-th
    // best key: raising it too would only cost events (cluster-ordered rows, 12 k x 25 k, k = 50: 3 500 rows short of k events and
    // searched again against 760).
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_b_pad) return;
    if (t >= n_b) {
        if (lane == 0) theta[t] = INFINITY;
        return;
    }
    const int M = lay.pieces[kz_list_region(t, lay)] * KP;
    const int64_t l0 = kz_list_contig_off(t, lay, KP, 0);
    float x[4];
    int valid[4];
    unsigned ux[4];   // sortable patterns (0 = no entry)
    unsigned all_or = 0u, all_and = 0xffffffffu;
    int nv = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = lane + 64 * u;
        x[u] = e < M ? in_key[l0 + e] : -INFINITY;
        valid[u] = e < M && in_idx[l0 + e] >= 0;
        const unsigned b = __float_as_uint(x[u]);
        ux[u] = valid[u] ? (b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u)) : 0u;
        all_or |= ux[u];
        all_and &= valid[u] ? ux[u] : 0xffffffffu;
        nv += valid[u] ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nv += __shfl_xor(nv, off, 64);
    // tau = the rank-th best valid key (its VALUE: ties do not matter) -- a radix selection on the entries in registers (round 5;
    // rounds 3 - 4 ranked all M entries against each other through v_readlane: M^2 / 64 compares per lane, 0.6 ms on ns's 250 k rows)
    float tau = -INFINITY;
    if (nv >= rank) {   // (uniform)
        const unsigned thr = kz_radix_kth_u32_regs<4>(ux, all_or, all_and, rank);
        tau = __uint_as_float(thr ^ ((thr >> 31) ? 0x80000000u : 0xffffffffu));
    }
    float tau_bound = tau;
    if (sev_cnt) {
        if (KP <= 32) {
            const bool okv[4] = {valid[0] != 0, valid[1] != 0, valid[2] != 0, valid[3] != 0};
            tau_bound = fmaxf(tau_bound, kz_full_lists_bound<4>(x, okv, M, KP, lane));
        } else {
            for (int p0 = 0; p0 < M; p0 += KP) {   // (uniform; at most 256 / 64 lists)
                int c = 0;
                float mn = INFINITY;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = lane + 64 * u;
                    const bool in = e >= p0 && e < p0 + KP && valid[u];
                    c += in ? 1 : 0;
                    mn = in ? fminf(mn, x[u]) : mn;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    c += __shfl_xor(c, off, 64);
                    mn = fminf(mn, __shfl_xor(mn, off, 64));
                }
                if (c == KP) tau_bound = fmaxf(tau_bound, mn);
            }
        }
        int base = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool evt = valid[u] && x[u] >= tau && tau > -INFINITY;
            const unsigned long long mask = __ballot(evt);
            if (evt) {
                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (pos < sev_cap) sev[t * (int64_t)sev_cap + pos] = make_uint2(__float_as_uint(x[u]), (unsigned)idx_to_row[in_idx[l0 + lane + 64 * u]]);
            }
            base += (int)__popcll(mask);
        }
        if (lane == 0) sev_cnt[t] = base;
    }
    if (lane == 0) {
        const double S2 = hscale[0] * hscale[0];
        const double margin = 4.76837158203125e-07 * S2 * (a_hmax[0] * b_hmax[0] + a_hmax[2] + b_hmax[2]);   // 2^-21 Mx
        const double th = (double)tau + (double)bias_b[t] - margin;
        float tf = (float)th;
        if ((double)tf > th) tf = nextafterf(tf, -INFINITY);
        theta[t] = tf;
        floor_[t] = tau_bound;
    }
}

// Sample image: the rows of every stride-th tile of A's image, dealt over P parts ROW BY ROW (data stored cluster by cluster has
// all near rows of a query in one stretch of A; dealt, every part holds its share of them).  One workgroup per image tile; a
// row of the fp16 image is 2 nsr fragments of 16 bytes, one per (slice, plane) block of its tile.
__global__ __launch_bounds__(256) void kz_dual_sample_kernel(const uint4* __restrict__ packed, const float* __restrict__ bias,
                                                             int nsr, int stride, int s_tiles, int P,
                                                             uint4* __restrict__ s_packed, float* __restrict__ s_bias) {
    const int jp = blockIdx.x;
    const int n_frag = 2 * nsr * KZ_TILE;   // 16-byte fragments of a tile: [slice][plane][row]
    for (int e = threadIdx.x; e < n_frag; e += 256) {
        const int row = e & (KZ_TILE - 1), blk = e >> 7;
        const int64_t j = kz_dealt_row((int64_t)jp * KZ_TILE + row, (int64_t)s_tiles * KZ_TILE, P);
        const int64_t src_tile = (j >> 7) * stride;
        s_packed[(int64_t)jp * n_frag + e] = packed[src_tile * n_frag + blk * KZ_TILE + (j & (KZ_TILE - 1))];
        if (blk == 0) s_bias[(int64_t)jp * KZ_TILE + row] = bias[src_tile * KZ_TILE + (j & (KZ_TILE - 1))];
    }
}

// Short-list route of the main sweep: the TILES of the sorted order are dealt over P index ranges (rows with neighbouring
// thresholds tend to be neighbours of the same queries); the last, possibly partial, tile stays the last.  Whole tiles, not
// rows: the sweep's per-tile test wants the 128 rows of a tile to be neighbours in threshold (dealt row by row a tile spans
// 128 P sorted ranks -- 500k x 500k, k = 50: the log of passed groups overflowed its 1.5 x estimate).
__global__ void kz_dual_interleave_kernel(const int* __restrict__ perm, const float* __restrict__ theta_s, int64_t n, int64_t n_pad,
                                          int P, int* __restrict__ perm_out, float* __restrict__ theta_out) {
    const int64_t rp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rp >= n_pad) return;
    const int64_t tiles = n_pad / KZ_TILE - 1;   // the tiles that move
    const int64_t jp = rp / KZ_TILE;
    const int64_t src = jp < tiles ? kz_dealt_row(jp, tiles, P) * KZ_TILE + (rp & (KZ_TILE - 1)) : rp;
    perm_out[rp] = src < n ? perm[src] : -1;
    theta_out[rp] = src < n ? theta_s[src] : INFINITY;
}

__global__ void kz_dual_fill_kernel(float* __restrict__ out, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}

// per-tile minimum of the thresholds in image order, broadcast over the tile's rows (the kernel copies the first 64 floats of a
// tile and reads one).  (Rows behind the end carry +inf.)
__global__ void kz_dual_tilemin_kernel(const float* __restrict__ theta_sorted, int64_t n, int64_t n_pad, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    const int64_t t0 = (i / KZ_TILE) * KZ_TILE;
    const int lane = threadIdx.x & 63;
    // (blocks of 256 threads = two whole tiles; the 64 lanes of a wave share a tile)
    float m = fminf(theta_sorted[t0 + lane], theta_sorted[t0 + 64 + lane]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
    out[i] = m;
}

// Load balance of the query side.  How many events a query row takes part in is heavy-tailed (hubness: rows near the data
// centre are near neighbours of many index rows; simulated on uniform data: mean 10, deviation 14, maximum 388 per row, and
// the sums over the 128 rows of a tile still range over a factor 2.3), and the count correlates with |q_c|^2 at -0.75.  A
// workgroup with an event-rich tile falls behind the others of its XCD; once it is more than an L2's worth of index tiles
// behind, its index stream misses L2 (measured: hit rate 96 % -> 58 %, 171 GB of fabric reads per launch).  So the rows of A
// are dealt into tiles like cards: sorted by |q_c|^2, full tile t takes the ranks t, t + T, t + 2T, ... -- every tile gets the
// same mix.  (The ragged tail keeps the last ranks.)
__global__ void kz_dual_c2key_kernel(const double* __restrict__ rowq, int64_t n, float* __restrict__ key) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = (float)rowq[i * 3];
}
__global__ void kz_dual_deal_kernel(const int* __restrict__ sorted_rows, int64_t n, int64_t n_pad, int* __restrict__ row_map) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pad) return;
    const int64_t T = n / KZ_TILE, n_full = T * KZ_TILE;
    int r = -1;
    if (p < n_full)
        r = sorted_rows[(p % KZ_TILE) * T + p / KZ_TILE];
    else if (p < n)
        r = sorted_rows[p];
    row_map[p] = r;
}

__global__ void kz_dual_natural_kernel(int64_t n, int64_t n_pad, int* __restrict__ row_map) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pad) row_map[p] = p < n ? (int)p : -1;
}

// NESTED sample: small kernels.  rows_of: out[i] = row_map[perm[i]] (sorted sample image row -> matrix row);
// scatter_f32: out[map[j]] = in[j]; inject: the sample-row events of matrix row perm[t] become the first entries of sorted row
// t's event buffer (the main sweep's scatter kernel appends behind them).
__global__ void kz_dual_rows_of_kernel(const int* __restrict__ perm, const int* __restrict__ row_map, int64_t n, int* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = row_map[perm[i]];
}
__global__ void kz_dual_scatter_f32_kernel(const float* __restrict__ in, const int* __restrict__ map, int64_t n, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && map[j] >= 0) out[map[j]] = in[j];
}
// range number of every sorted sample row (as a float key for the stable radix sort) + its sorted position as the value
__global__ void kz_dual_rangekey_kernel(const int* __restrict__ perm, int64_t n, int64_t range_rows, float* __restrict__ key, int* __restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (float)(perm[i] / range_rows);
    pos[i] = (int)i;
}
__global__ void kz_dual_gather2_kernel(const int* __restrict__ order, const int* __restrict__ perm, const float* __restrict__ theta, int64_t n,
                                       int* __restrict__ perm_out, float* __restrict__ theta_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    perm_out[i] = perm[order[i]];
    theta_out[i] = theta[order[i]];
}
__global__ void kz_dual_inject_kernel(const int* __restrict__ perm, int64_t n_b, const int* __restrict__ sev_cnt, const uint2* __restrict__ sev,
                                      int sev_cap, int* __restrict__ ev_cnt, uint2* __restrict__ ev, int ev_cap) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_b) return;
    const int64_t orig = perm[t];
    if (orig < 0) return;
    const int filed = sev_cnt[orig];
    const int c = filed < sev_cap ? filed : sev_cap;
    for (int i = 0; i < c && i < ev_cap; ++i) ev[t * (int64_t)ev_cap + i] = sev[orig * (int64_t)sev_cap + i];
    // (more sample events than either buffer holds: the count says so, the select kernel sends the row to the ordinary search)
    ev_cnt[t * KZ_EVC] = filed > sev_cap ? ev_cap + 1 : c;
}

// -bias of the query side (pad rows: +inf, never an event)
__global__ void kz_dual_negbias_kernel(const float* __restrict__ bias, int64_t n, int64_t n_pad, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) out[i] = i < n ? -bias[i] : INFINITY;
}

// ---- logged groups -> per-row event buffers --------------------------------------------------------------------------
// One thread per logged group of four keys: the kernel's test again per key (the same float32 expression), an atomic slot
// in the row's buffer, one 8-byte store.  A row with more events than its buffer holds keeps counting (kz_dual_select_kernel
// sees the overflow and sends the row to the ordinary search).
__global__ __launch_bounds__(256) void kz_dual_scatter_kernel(const f32x4e* __restrict__ log_keys, const i32x2e* __restrict__ log_meta,
                                                              const unsigned long long* __restrict__ log_cnt, long long log_cap,
                                                              const float* __restrict__ theta, const float* __restrict__ qnb,
                                                              const float* __restrict__ bias_b_sorted,
                                                              const int* __restrict__ row_map, int* __restrict__ ev_cnt,
                                                              uint2* __restrict__ ev, int ev_cap) {
    const unsigned long long filled = *log_cnt;
    // (an overflowed or poisoned log holds entries that were never written: nothing is filed, the host falls back)
    const long long n = filled <= (unsigned long long)log_cap ? (long long)filled : 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const f32x4e kv = log_keys[i];
        const i32x2e mt = log_meta[i];
        const int ql = mt.x & 63, tg = mt.x >> 6;   // lane of the wave, 16 tile + group
        const int row0 = (tg >> 4) * KZ_TILE + ((tg >> 2) & 3) * 32 + (tg & 3) * 8 + 4 * (ql >> 5);
        const float nb = qnb[mt.y];
        const float4 th = *reinterpret_cast<const float4*>(theta + row0);
        const float4 bt = *reinterpret_cast<const float4*>(bias_b_sorted + row0);
        const float kk[4] = {kv.x, kv.y, kv.z, kv.w};
        const float tt[4] = {th.x, th.y, th.z, th.w};
        const float bb[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (kk[u] - tt[u] >= nb) {
                const int slot = atomicAdd(ev_cnt + (int64_t)(row0 + u) * KZ_EVC, 1);
                // The event is filed as the reverse-direction key  key'(t, q) = acc - bias(t) + bias(q), evaluated in float64
                // on the float32 values (exact) and rounded once to float32 (the list format), under the MATRIX row of the
                // query (mt.y is a row of the dealt query image; -nb is its bias).
                const float kf = (float)(((double)kk[u] - (double)bb[u]) - (double)nb);
                if (slot < ev_cap) ev[(int64_t)(row0 + u) * ev_cap + slot] = make_uint2(__float_as_uint(kf), (unsigned)row_map[mt.y]);
            }
        }
    }
}

constexpr int KZ_DUAL_SPREAD = 1024;                                   // counter pairs of the select kernel's statistics (a power of two)
constexpr size_t KZ_DUAL_CNT_BYTES = 128 + (size_t)KZ_DUAL_SPREAD * 16;   // [0] log counter, [1] events, [2] overflowing rows, [16 ..] the pairs
// ---- the K' best events of a row -> an ordinary candidate list ---------------------------------------------------------
// One wave per row t of B.  Selection by (key' descending, q ascending) is a total order, so the list does not depend on
// the order the atomics filed the events in.  K'-th key by radix select on the sortable bit pattern.
__global__ __launch_bounds__(256) void kz_dual_select_kernel(const int* __restrict__ ev_cnt, const uint2* __restrict__ ev, int ev_cap,
                                                             int64_t n_b, const int* __restrict__ perm, int KP, float* __restrict__ out_key,
                                                             int* __restrict__ out_idx, float* __restrict__ floor_,
                                                             unsigned long long* __restrict__ totals) {
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + wave;   // row of the sorted image: events were filed under it
    if (t >= n_b) return;
    const int64_t orig = perm[t];                        // the matrix row: lists and floors are kept under it
    unsigned* su = reinterpret_cast<unsigned*>(ssm) + (size_t)wave * 2 * ev_cap;   // sortable key bits
    int* sq = reinterpret_cast<int*>(su + ev_cap);
    const int filed = ev_cnt[t * KZ_EVC];
    const int n = filed < ev_cap ? filed : ev_cap;
    if (lane == 0) {
        // (statistics only: spread over KZ_DUAL_SPREAD counter pairs -- one pair for all 250 k waves serialised the whole kernel on
        //  a single L2 atomic: 3.1 ms of which 2 were this line; 32 pairs still queued 15 000 atomics per address on C3's 500 k rows)
        unsigned long long* tot = totals + 2 * (blockIdx.x & (KZ_DUAL_SPREAD - 1));
        atomicAdd(tot + 0, (unsigned long long)filed);
        if (filed > ev_cap) {
            floor_[orig] = INFINITY;   // incomplete events: the certification must fail, the row is searched again
            atomicAdd(tot + 1, 1ull);
        }
    }
    if (n <= KP) {
        // the usual case where the list is longer than the events are many (reverse lists of 2 K'): every event is a list entry --
        // straight from the buffer to the list, no LDS, no reductions
        float* okd = out_key + orig * (int64_t)KP;
        int* oid = out_idx + orig * (int64_t)KP;
        for (int e = lane; e < KP; e += 64) {
            uint2 v = make_uint2(0u, 0u);
            if (e < n) v = ev[t * (int64_t)ev_cap + e];
            okd[e] = e < n ? __uint_as_float(v.x) : -INFINITY;
            oid[e] = e < n ? (int)v.y : -1;
        }
        return;
    }
    unsigned all_or = 0u, all_and = 0xffffffffu;
    for (int e = lane; e < n; e += 64) {
        const uint2 v = ev[t * (int64_t)ev_cap + e];
        const unsigned b = v.x;   // the reverse-direction key as filed by kz_dual_scatter_kernel
        const unsigned u = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
        su[e] = u;
        sq[e] = (int)v.y;
        all_or |= u;
        all_and &= u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        all_or |= __shfl_xor(all_or, off, 64);
        all_and &= __shfl_xor(all_and, off, 64);
    }
    kz_wave_sync();
    float* ok = out_key + orig * (int64_t)KP;
    int* oi = out_idx + orig * (int64_t)KP;
    auto key_of = [](unsigned u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); };
    if (n <= KP) {
        for (int e = lane; e < KP; e += 64) {
            ok[e] = e < n ? key_of(su[e]) : -INFINITY;
            oi[e] = e < n ? sq[e] : -1;
        }
        return;
    }
    // thr = the largest value with at least K' entries >= it: the K'-th largest key (kz_radix_kth_u32: below the common prefix of
    // the keys, and -- round 5 -- on the row's entries held in REGISTERS when there are at most 512: the counting passes of the
    // LDS version each waited for their LDS reads, bit after bit; C3's 500 k rows: 3.3 -> see profiles/r05_notes.md section 10)
    (void)all_or;
    (void)all_and;
    const unsigned thr = kz_radix_kth_u32<true>(su, n, KP, lane);
    int base = 0;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        const bool sel = e < n && su[e] > thr;
        const unsigned long long mask = __ballot(sel);
        if (sel) {
            const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            ok[pos] = key_of(su[e]);
            oi[pos] = sq[e];
        }
        base += (int)__popcll(mask);
    }
    // the remaining slots go to the entries equal to thr with the smallest query rows
    int last = -1;
    for (; base < KP; ++base) {
        int best = 0x7fffffff;
        for (int e = lane; e < n; e += 64)
            if (su[e] == thr && sq[e] > last && sq[e] < best) best = sq[e];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
        if (lane == 0) {
            ok[base] = best != 0x7fffffff ? key_of(thr) : -INFINITY;
            oi[base] = best != 0x7fffffff ? best : -1;   // (cannot happen: at least K' entries are >= thr; an empty slot, not a wild row, if it ever does)
        }
        last = best;
    }
}

__global__ void kz_dual_sum_kernel(const unsigned long long* __restrict__ spread, unsigned long long* __restrict__ out) {
    unsigned long long a = 0ull, b = 0ull;
    for (int i = threadIdx.x; i < KZ_DUAL_SPREAD; i += 64) {
        a += spread[2 * i];
        b += spread[2 * i + 1];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = b;
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------
static const double KZ_DUAL_MAX_BYTES = 32.0 * (double)(1ull << 30);   // transient footprint the shared sweep may claim (of 288 GB)
static void kz_dual_fill_stats(kz_knn_stats* st, const kz_knn_stats& v) {
    if (st) *st = v;
}

// Both directions by two ordinary searches (shapes or settings the dual pass does not cover)
static int kz_knn_dual_separately(kz_ctx* ctx, kz_matrix* a, kz_matrix* b, int k, double* d_dist_ab, int64_t* d_ind_ab,
                                  double* d_dist_ba, int64_t* d_ind_ba, kz_knn_stats* stats_ab, kz_knn_stats* stats_ba,
                                  int precision_override = -1, int wide = 0) {
    // (precision_override = 2: the caller's probe has found the data hard for fp16 -- both searches start at the split-bf16 tier
    //  without probing again, with lists of at least 64 as an ordinary search's own probe would have chosen;
    //  wide > 0: the probe has found that the fp16 tier's WIDE route certifies this data -- both searches take it, kz_knn_impl kp_min = -wide)
    const int kp_min = wide > 0 ? -wide : (precision_override == 2 ? 64 : 0);
    int rc = kz_knn_impl(ctx, a, 0, a->n, b, k, 0, nullptr, precision_override, kp_min, d_dist_ab, d_ind_ab, stats_ab, nullptr);
    if (rc == KZ_OK) rc = kz_knn_impl(ctx, b, 0, b->n, a, k, 0, nullptr, precision_override, kp_min, d_dist_ba, d_ind_ba, stats_ba, nullptr);
    return rc;
}

// Rank of the sample key that becomes a row's event threshold (kz_knn_dual "rank").  Model, per candidate rank r: stride s =
// sqrt(T / (|B| r c_ev)) clamped as kz_knn_dual clamps it; cost = T / s (sample sweep) + c_ev |B| r s (events: log, scatter, select,
// the slower sweep) + P_fail |B| c_row (rows with fewer than k events, searched again; c_row = three times a row's share of a
// sweep: small batches run well below the sweep's rate); P_fail = P(Gamma(r) < (k - r + 1) / (s - 1)) -- the lower tail of the
// negative binomial count of non-sample rows above the r-th best of an s-fold sample.  Ranks with P_fail > 1e-3 are out.
// P(a row gets fewer than k events) at threshold rank r of an st-fold sample: P(Gamma(r) < (k - r + 1) / (st - 1))
static inline double kz_dual_p_fail(int k, int r, double st) {
    if (r >= k + 1) return 0.0;
    const double x = (double)(k - r + 1) / (st - 1.0);   // Gamma(r, 1) must reach this
    // P(Gamma(r) < x) = e^-x sum_{j >= r} x^j / j!
    double term = exp(-x);
    for (int j = 1; j <= r; ++j) term *= x / j;
    double sum = 0.0;
    for (int j = r; j < r + 200 && term > 1e-300; ++j) {
        sum += term;
        term *= x / (j + 1);
    }
    return sum < 1.0 ? sum : 1.0;
}
static inline int kz_dual_pick_rank(int k, int rank_safe, double t_sweep_ms, double a_n, double b_n) {
    const double c_ev = 0.10e-6;   // ms per event (kz_knn_dual's stride model)
    auto stride_of = [&](int r) {
        const double s_opt = sqrt(t_sweep_ms / (b_n * r * c_ev));
        double st = s_opt < 4.0 ? 4.0 : (s_opt > 32.0 ? 32.0 : floor(s_opt + 0.5));
        const double s_max = floor(4096.0 / ((double)r + 7.0 * sqrt((double)r) + 1.0));
        return st > s_max ? s_max : st;
    };
    auto p_fail = [&](int r, double st) { return kz_dual_p_fail(k, r, st); };
    const double c_row = 3.0 * t_sweep_ms / a_n;
    int best = rank_safe;
    double best_cost = 1e300;
    // (floor: a quarter of the safe rank and 8 -- below, the model's gains were not there when measured: 250k x 1M, k = 10: rank
    //  8 -> 106.9 ms per step, rank 6 -> 108.6, k + 1 = 11 -> 108.7; a sample of a few hundred tiles is a short sweep)
    int r_min = (rank_safe + 3) / 4 > 8 ? (rank_safe + 3) / 4 : 8;
    if (r_min > rank_safe) r_min = rank_safe;
    for (int r = rank_safe; r >= r_min; --r) {
        const double st = stride_of(r);
        const double pf = p_fail(r, st);
        if (pf > 1e-3) break;   // (P_fail grows as r falls)
        const double cost = t_sweep_ms / st + c_ev * b_n * r * st + pf * b_n * c_row;
        if (cost < best_cost) {
            best_cost = cost;
            best = r;
        }
    }
    return best;
}

// The chain that turns a sweep's event log into results for the rows that OWN the events: scatter -> select -> ordinary finalize
// with those rows as the query side.  Main sweep: the rows of b (events from the rows of a).  NESTED sample sweep: the sample rows of
// a (events from the rows of b); their lists are kept under their position j in the dealt image (perm = sorted row -> j, fin_row_map =
// j -> matrix row of a), the floor the finalize kernel reads goes by matrix row (floor_fin, filled from floor_sel through fin_row_map).
struct KzRevChain {
    kz_ctx* ctx;
    kz_matrix *qm, *im;            // rows owning the events (query side of the finalize) / rows the events come from (index side)
    const kz_himage *qi, *ii;
    void *log_keys, *log_meta;
    unsigned long long* d_cnt;
    long long log_cap;
    float *theta_s, *qnb, *p_bias, *col_key, *floor_sel, *floor_fin;
    int *row_map, *ev_cnt, *perm, *col_idx, *fail_list, *fail_count;
    const int* fin_row_map;
    uint2* ev;
    int ev_cap, KP, k;
    int64_t n_rows, n_tiles;
    double* d_dist;
    int64_t* d_ind;
    int timed, h_fail, h_cnt, second_stream;
    KzSpec* spec;   // speculative exact re-search behind this chain's finalize (kz_knn.hip "SPECULATIVE RESCUE"); buffers owned by the caller
};
static int kz_dual_enqueue_chain(KzRevChain& r) {
    kz_ctx* ctx = r.ctx;
    hipStream_t first = ctx->stream;
    if (r.second_stream) {
        // (the sweep was recorded as ev[1] of the first stream by kz_knn_impl just now)
        KZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->ev[1], 0));
        ctx->stream = ctx->stream2;   // every launch helper below enqueues on ctx->stream
    }
    auto body = [&]() -> int {
        if (r.timed) KZ_HIP(hipEventRecord(ctx->ev[8], ctx->stream));
        hipLaunchKernelGGL(kz_dual_scatter_kernel, dim3(ctx->n_cus * 8), dim3(256), 0, ctx->stream, (const f32x4e*)r.log_keys,
                           (const i32x2e*)r.log_meta, r.d_cnt, r.log_cap, r.theta_s, r.qnb, r.p_bias, r.row_map, r.ev_cnt, r.ev, r.ev_cap);
        const size_t sel_lds = (size_t)4 * 2 * r.ev_cap * 4;
        if (sel_lds > 65536)
            KZ_HIP(hipFuncSetAttribute((const void*)kz_dual_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_lds));
        hipLaunchKernelGGL(kz_dual_select_kernel, dim3((unsigned)((r.n_rows + 3) / 4)), dim3(256), sel_lds, ctx->stream, r.ev_cnt, r.ev, r.ev_cap,
                           r.n_rows, r.perm, r.KP, r.col_key, r.col_idx, r.floor_sel, r.d_cnt + 16);
        hipLaunchKernelGGL(kz_dual_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, r.d_cnt + 16, r.d_cnt + 1);
        if (r.fin_row_map && r.floor_fin != r.floor_sel)
            hipLaunchKernelGGL(kz_dual_scatter_f32_kernel, dim3((unsigned)((r.n_rows + 255) / 256)), dim3(256), 0, ctx->stream, r.floor_sel,
                               r.fin_row_map, r.n_rows, r.floor_fin);
        KZ_HIP(hipGetLastError());
        if (r.timed) KZ_HIP(hipEventRecord(ctx->ev[9], ctx->stream));
        KzListLayout lay;
        memset(&lay, 0, sizeof(lay));
        lay.n_regions = 1;
        lay.qt_end[0] = (int)r.n_tiles;
        lay.pieces[0] = 1;
        lay.halves = 1;
        lay.contig = 1;
        KZ_HIP(hipMemsetAsync(r.fail_count, 0, 4 * sizeof(int), ctx->stream));
        KnnFinParams fp;
        memset(&fp, 0, sizeof(fp));
        fp.in_key = r.col_key;
        fp.in_idx = r.col_idx;
        fp.lay = lay;
        fp.KP = r.KP;
        fp.list_row0 = 0;
        fp.q_begin = 0;
        fp.q_count = r.n_rows;
        fp.row_map = r.fin_row_map;
        fp.qraw = r.qm->raw;
        fp.yraw = r.im->raw;
        fp.qsqn = r.qm->sqn;
        fp.ysqn = r.im->sqn;
        fp.n_i = r.im->n;
        fp.d = (int)r.im->d;
        fp.metric = r.im->metric;
        fp.k = r.k;
        fp.ystats = r.im->d_stats;
        fp.tier_h = 1;
        fp.eps_mult = ctx->eps_scale;
        fp.gamma_acc = 2.0 * (double)(r.im->kg * 4 + 16) * 5.9604644775390625e-08;
        fp.q_rowq = r.qi->rowq;
        fp.y_hmax = r.ii->d_max;
        fp.hscale = r.ii->center->d_scale;
        fp.excl_floor = r.floor_fin;
        fp.dual_col = 1;
        fp.out_dist = r.d_dist;
        fp.out_ind = r.d_ind;
        fp.fail_count = r.fail_count;
        fp.fail_list = r.fail_list;
        fp.err_ratio_bits = (unsigned long long*)(r.fail_count + 2);
        const int rc2 = kz_launch_finalize(ctx, fp, lay, r.KP, r.n_rows, r.im->dtype);
        if (rc2 != KZ_OK) return rc2;
        if (r.timed) KZ_HIP(hipEventRecord(ctx->ev[10], ctx->stream));
        if (r.spec) {
            // (the handful of rows this chain leaves uncertified -- short of events, an overflowing buffer, a near-tie -- answered by the
            //  exact kernels on this chain's stream before the host knows the count; fail_list holds matrix rows)
            const int R = kz_spec_rows(ctx, r.im, r.k);
            if (R > 0 && r.spec->vals) {   // (the buffers were allocated ahead: kz_knn_dual)
                const int rc3 = kz_spec_rescue(ctx, *r.spec, R, r.qm, 0, r.fail_list, r.fail_count, r.im, r.k, 0, nullptr, r.d_dist, r.d_ind);
                if (rc3 != KZ_OK) return rc3;
            }
        }
        KZ_HIP(hipMemcpyAsync(ctx->h_counters + r.h_fail, r.fail_count, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        KZ_HIP(hipMemcpyAsync(ctx->h_counters + r.h_cnt, r.d_cnt, 32, hipMemcpyDeviceToHost, ctx->stream));
        return KZ_OK;
    };
    const int rc2 = body();
    ctx->stream = first;
    return rc2;
}

extern "C" int kz_knn_dual(kz_ctx* ctx, const kz_matrix* a_c, const kz_matrix* b_c, int k, double* d_dist_ab, int64_t* d_ind_ab,
                           double* d_dist_ba, int64_t* d_ind_ba, kz_knn_stats* stats_ab, kz_knn_stats* stats_ba) {
    kz_matrix* a = const_cast<kz_matrix*>(a_c);
    kz_matrix* b = const_cast<kz_matrix*>(b_c);
    KZ_REQUIRE(ctx && a && b && d_dist_ab && d_ind_ab && d_dist_ba && d_ind_ba, "kz_knn_dual: null argument");
    KZ_REQUIRE(a != b, "kz_knn_dual: the two matrices must be different objects (a single matrix is searched with kz_knn)");
    KZ_REQUIRE(a->ctx == ctx && b->ctx == ctx, "kz_knn_dual: matrices belong to a different context");
    KZ_REQUIRE(!a->raw_only && !b->raw_only, "kz_knn_dual: a rows-only matrix (kz_matrix_create rows_on_device = 3) cannot be searched");
    KZ_REQUIRE(a->d == b->d, "kz_knn_dual: feature dimensions differ (%lld vs %lld)", (long long)a->d, (long long)b->d);
    KZ_REQUIRE(a->dtype == b->dtype, "kz_knn_dual: the matrices must have the same dtype");
    KZ_REQUIRE(a->metric == b->metric, "kz_knn_dual: the matrices were packed for different metrics");
    KZ_REQUIRE(k >= 1, "kz_knn_dual: Expected k > 0. Got %d", k);
    KZ_REQUIRE((int64_t)k <= a->n && (int64_t)k <= b->n,
               "kz_knn_dual: Expected n_neighbors <= n_samples_fit, but n_neighbors = %d, n_samples_fit = %lld", k,
               (long long)(a->n < b->n ? a->n : b->n));
    if (stats_ab) memset(stats_ab, 0, sizeof(*stats_ab));
    if (stats_ba) memset(stats_ba, 0, sizeof(*stats_ba));
    KZ_HIP(hipSetDevice(ctx->device));

    const int KP = kz_pick_list_len(k);
    // the threshold of a row is its (k + 1)-th best sample key: the sample rows are rows of a, so k rows STRICTLY above the
    // threshold are there by construction (at rank k a row whose k best all happen to be sample rows -- probability stride^-k:
    // 4-25 % of the rows at k = 1 -- could not be certified and was searched again)
    // ... that is the SAFE rank.  Round 4: the threshold sits at a LOWER rank r of a thinner sample (stride ~ 1 / sqrt(r)): sample
    // sweep and events per row (~ r stride) both shrink with sqrt(r).  The k rows are then no longer there by construction: the
    // number of non-sample rows above the r-th best sample key is negative binomial (r, 1 / stride) -- ~ Gamma(r) x (stride - 1),
    // whatever the data looks like -- and a row that gets fewer than k events comes out of the finalize kernel uncertified (V < k)
    // and is searched again like any other.  r = the cheapest rank whose share of such rows stays below 1e-3 (kz_dual_pick_rank);
    // 500k x 500k, k = 50: rank 51 -> 12, stride 6 -> 12: step 162.9 -> 142.7 ms, rows searched again unchanged (~750); k = 10:
    // rank 11 -> 8: ns 108.7 -> 106.9 ms.  Option "dual_rank": 0 = automatic, -1 = k + 1, > 0 = that rank.
    const int rank_safe = k + 1 < KP ? k + 1 : KP;
    int rank = rank_safe;
    // (small sweeps with lists of 16 keep the safe rank: the few hundred rows a lower rank sends to a re-search cost a fixed ~0.3 ms,
    //  which the model of kz_dual_pick_rank does not know -- 100k x 100k, d 128, k 10 through the API: rank 8 / stride 6 6.01 ms,
    //  rank 8 / stride 4 5.95 with 1 640 rows searched again, rank 11 / stride 4 5.72, two searches 6.10)
    const bool small_sweep = KP == 16 && 2.0 * (double)a->n * (double)b->n * (double)(a->kg * 4) / 1e12 < 4.0;
    if (ctx->dual_rank > 0)
        rank = ctx->dual_rank < rank_safe ? ctx->dual_rank : rank_safe;
    else if (ctx->dual_rank == 0 && ctx->dual_stride == 1 && !small_sweep)
        rank = kz_dual_pick_rank(k, rank_safe, 2.0 * (double)a->n * (double)b->n * (double)(a->kg * 4) / 1e12, (double)a->n, (double)b->n);
    // list length of the REVERSE direction (the K' best events of a row): twice K' -- the events are there anyway
    // (~(k + 1) stride per row), the re-rank gathers only the candidates within 2 eps of the k-th key whatever the list length
    // is, and the certification's bound falls from the 64th to the 128th best key: on clustered data (many near-equal
    // distances) far fewer rows are searched again (400k x 400k, k = 50, 40 clusters: 42.7k -> 0 rows, call 162 -> 134 ms; uniform 500k x 500k: +0.9 ms; k = 10: 12.4k -> 0 rows, 104 -> 97 ms, and ns 105.9 -> 104.4 ms per step:
    // its ~25 uncertified reverse rows per step are gone)
    const int KPr = ctx->dual_rev_long ? (2 * KP < 128 ? 2 * KP : 128) : KP;
    const int n_slices = b->kg / 4;
    // every stride-th tile of A is in the sample.  Automatic (dual_stride = 1): the sample sweep costs T / stride, the events
    // (log, scatter, select, slower sweep) ~0.10 ns each with |B| k stride of them: stride = sqrt(T / (|B| k 0.07 ns)), T ~ 2 |A| |B| d / 1e15 s
    // (ns: 20, measured flat between 16 and 28; 500k x 500k, k = 50: 6)
    int stride = ctx->dual_stride;
    if (stride == 1) {
        const double t_ms = 2.0 * (double)a->n * (double)b->n * (double)(a->kg * 4) / 1e12;
        // (0.10 ns per event: round 3, same box, 500k x 500k, k = 50: stride 4 / 5 / 6 / 7 / 8 -> 186.0 / 184.8 / 183.8 / 183.8 /
        //  186.8 ms per step -- every event also slows the sweep itself, 117.6 -> 126.4 ms; 250k x 1M, k = 10: flat from 16 to 28)
        const double s_opt = sqrt(t_ms / ((double)b->n * rank * (small_sweep ? 0.20e-6 : 0.10e-6)));   // (small sweeps: an event costs relatively more)
        stride = s_opt < 4.0 ? 4 : (s_opt > 32.0 ? 32 : (int)(s_opt + 0.5));
        // With the NESTED sample (below; round 5) the sample rows are no longer swept twice: the sample sweep costs little more than
        // it saves the main sweep, while an event costs what it did -- the balance moves to HALF the stride (twice the sample, half
        // the events), as long as the rank's share of rows that end up short of k events stays below 1e-3 (kz_dual_p_fail).
        // Same box, rank 8, ms per step: ns stride 9 / 10 / 11 / 12 / 13 / 14 / 16 / 18 / 23 (the old choice) -> 98.5 / 98.4 / 98.0 /
        // 97.5 / 98.1 / 97.8 / 98.6 / 99.6 / 99.8; C4's share 10 / 12 / 14 / 16 / 28 -> 147.3 / 146.1 / 148.2 / 147.5 / 149.7; gmm
        // (200k x 200k x 300) 6 / 8 / 10 / 12 -> 34.0 / 34.7 / 34.9 / 35.7; C3 (rank 12) 8 / 10 / 13 / 16 -> 127.7 / 124.9 / 125.8 / 126.8.
        if (ctx->dual_nested && !small_sweep && b->n <= kz_rows_per_chunk(ctx, KP, false) && t_ms / stride >= KZ_K_NESTED_MIN_MS) {
            int half = (int)(0.5 * (s_opt > 32.0 ? 32.0 : s_opt) + 0.5);
            if (half < 4) half = 4;
            while (half < stride && kz_dual_p_fail(k, rank, (double)half) > 1e-3) ++half;
            if (half < stride) stride = half;
        }
    }
    // (a row's event buffer -- k stride + 7 sqrt(k) stride entries -- is selected from LDS, 8 B per entry and four rows per
    //  workgroup: at most 4096 entries)
    if (stride > 1 && KP > 0) {
        const int s_max = (int)(4096.0 / ((double)rank + 7.0 * sqrt((double)rank) + 1.0));
        if (stride > s_max) stride = s_max;
    }
    const int64_t a_tiles = a->n_tiles, b_tiles = b->n_tiles;
    const int64_t s_tiles = stride > 0 ? (a_tiles + stride - 1) / stride : 0;
    // rows of A in the sample (the last tile of A may be partial and may or may not be part of it)
    int64_t s_rows = s_tiles * KZ_TILE;
    if (stride > 0 && (a_tiles - 1) % stride == 0) s_rows -= a_tiles * KZ_TILE - a->n;
    // Does sharing the sweep pay?  It saves one sweep (T ~ 2 |A| |B| d / 1e15 s at the rate the kernel reaches) and costs: the
    // sweep itself ~20-25 % slower, the sample sweep T / stride, ~0.15 ns per event (log, scatter, select; |B| k stride events)
    // and ~2 ms of fixed work (sort, permuted image, small kernels).  Measured: 1M x 250k, d 200, K' 16: 134 against 186 ms per
    // fit + kneighbors; 500k x 500k, K' 64: 207 against 249 ms; 100k x 100k, d 128: 11.0 against 6.8 ms -- the last one is what
    // the margin below keeps out.  "dual_force" (test knob) skips this gate.
    // Round 4 (threshold at a lower sample rank, seeded lists, cheaper reverse chain), tools/dual_gate.py, two searches against the
    // forced shared sweep, ms: 100k x 100k, d 128, k 10: 6.03 / 5.77; d 300: 12.1 / 9.3; 150k x 60k, d 128: 6.07 / 5.44; 200k x 50k, d 200:
    // 8.80 / 6.44; 150k x 150k, d 128: 12.7 / 10.7; 100k x 100k, d 128, k 50: 12.6 / 10.6; 60k x 60k, d 200, k 50: 10.6 / 8.9 -- and on the
    // losing side 100k x 100k, d 64: 4.39 / 4.89; 70k x 70k, d 128: 4.13 / 4.17; 40k x 200k, d 128: 5.93 / 7.52; 50k x 50k: 2.30 / 2.74.
    // Fitted: the saving is ~0.7 T (twice that where the ordinary searches keep lists longer than 16) minus ~2 x the event term
    // minus 0.3 ms.  (Through the API the gain is larger than between the bare calls: C2 6.49 -> 5.61 ms per fit + kneighbors.)
    const double t_sweep_ms = 2.0 * (double)a->n * (double)b->n * (double)(a->kg * 4) / 1e12;
    const double t_events_ms = (double)b->n * rank * stride * 0.15e-6;
    const bool pays = ctx->dual_force || 0.7 * t_sweep_ms * (KP > 16 ? 2.0 : 1.0) > 2.0 * t_events_ms + 0.3;
    const bool eligible = pays && stride >= 2 && ctx->precision == 0 && KP > 0 && a->metric < KZ_MANHATTAN && n_slices >= 2 && n_slices <= 24 && a->kg == b->kg &&
                          s_rows >= (int64_t)8 * KP && b->n >= 1024 && b_tiles < (1 << 20);
    if (!eligible) return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba);
    {
        // Footprint gate: the event buffers (8 B x capacity per row of B), the log (24 B per expected group) and three permuted
        // fp16 images live for the duration of the call and stay in the context's buffer cache afterwards (kz_ctx_trim drops
        // it).  1M x 1M at k = 100 would ask for ~30 GB: beyond a budget -- or what the device has free -- search twice.
        const double rk = (double)(k + 1 < KP ? k + 1 : KP);
        const double ev_b = (double)b->n_tiles * KZ_TILE * (rk * stride + 7.0 * sqrt(rk) * stride + 64.0) * 8.0;
        const double log_b = ((double)b->n * rk * stride * 1.5 + (double)(1 << 20)) * 24.0;
        const double img_b = ((double)s_tiles + (double)a_tiles + (double)b_tiles) * (double)n_slices * 4096.0;
        size_t free_b = 0, total_b = 0;
        KZ_HIP(hipMemGetInfo(&free_b, &total_b));
        const double avail = (double)free_b + (double)ctx->pool_bytes - 2.0 * (double)(1ull << 30);
        const double budget = ctx->dual_max_gb > 0 ? ctx->dual_max_gb * (double)(1ull << 30) : KZ_DUAL_MAX_BYTES;
        if (ev_b + log_b + img_b > budget || ev_b + log_b + img_b > avail)
            return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba);
    }

    int rc = kz_himage_ensure(a, b);
    if (rc != KZ_OK) return rc;
    const kz_himage* ia = a->himg;
    const kz_himage* ib = b->himg;
    const int64_t b_pad = b_tiles * KZ_TILE, a_pad = a_tiles * KZ_TILE;

    // expected events per row of B: k (stride - 1) + k, deviation sqrt(k) stride; the buffer takes mean + ~7 deviations (a row
    // that overflows is searched again on its own: ~1.5 ms for a single row against a million index rows)
    const int ev_cap = (int)(((int64_t)rank * stride + (int64_t)(7.0 * sqrt((double)rank) * stride) + 63) & ~(int64_t)63);
    // logged groups: about one per event (rarely two events share a group) plus the groups that pass the tile's smallest
    // threshold but not their own rows' (few: the rows of a tile are neighbours in threshold order); the TOTAL over all rows
    // is sharply concentrated around |B| k stride -- 1.5 times that, plus slack for small inputs (24 B per entry).  An
    // overflowing log is detected and the direction redone.
    const long long log_cap = (long long)((double)b->n * rank * stride * 1.5) + (1 << 20);

    unsigned short *s_packed = nullptr, *p_packed = nullptr, *q_packed = nullptr;
    float *s_bias = nullptr, *p_bias = nullptr, *q_bias = nullptr, *q_key = nullptr, *q_key_s = nullptr, *theta = nullptr, *theta_s = nullptr, *theta_min = nullptr, *floor_ = nullptr, *qnb = nullptr, *col_key = nullptr, *qfloor = nullptr;
    int *ev_cnt = nullptr, *col_idx = nullptr, *fail_list = nullptr, *iota = nullptr, *perm = nullptr, *q_iota = nullptr, *q_sorted = nullptr, *row_map = nullptr;
    uint2* ev = nullptr;
    void *log_keys = nullptr, *log_meta = nullptr;
    unsigned long long* d_cnt = nullptr;   // [0] log counter, [1] events filed, [2] rows with an overflowing buffer, [16 ..] 32 spread pairs of [1], [2]
    // buffers of the nested stages (released with everything else)
    unsigned short *s3_packed = nullptr, *ss_packed = nullptr;
    float *s3_bias = nullptr, *ss_bias = nullptr, *theta3 = nullptr, *theta3_s = nullptr, *theta3_min = nullptr, *floor3 = nullptr, *floor3m = nullptr,
          *qnb_b = nullptr, *col3_key = nullptr;
    int *iota3 = nullptr, *perm3 = nullptr, *rm3 = nullptr, *row_map_b = nullptr, *ev3_cnt = nullptr, *col3_idx = nullptr, *fail3 = nullptr, *sev_cnt = nullptr;
    uint2 *ev3 = nullptr, *sev = nullptr;
    void *log3_keys = nullptr, *log3_meta = nullptr;
    unsigned long long* d_cnt3 = nullptr;
    auto release_nested = [&]() {
        void* bufs[] = {s3_packed, ss_packed, s3_bias, ss_bias, theta3, theta3_s, theta3_min, floor3, floor3m, qnb_b, col3_key, iota3, perm3, rm3,
                        row_map_b, ev3_cnt, col3_idx, fail3, sev_cnt, ev3, sev, log3_keys, log3_meta, d_cnt3};
        for (void* q : bufs) kz_pool_free(ctx, q, 0);
        s3_packed = ss_packed = nullptr; s3_bias = ss_bias = theta3 = theta3_s = theta3_min = floor3 = floor3m = qnb_b = col3_key = nullptr;
        iota3 = perm3 = rm3 = row_map_b = ev3_cnt = col3_idx = fail3 = sev_cnt = nullptr; ev3 = sev = nullptr; log3_keys = log3_meta = nullptr; d_cnt3 = nullptr;
    };
    KzSpec spec_ba, spec_s3;   // (buffers of the two chains' speculative exact launches: released with everything else, behind the second stream's sync)
    auto release = [&]() {
        release_nested();
        kz_spec_release(ctx, spec_ba);
        kz_spec_release(ctx, spec_s3);
        kz_pool_free(ctx, s_packed, 0);
        kz_pool_free(ctx, s_bias, 0);
        kz_pool_free(ctx, p_packed, 0);
        kz_pool_free(ctx, p_bias, 0);
        kz_pool_free(ctx, q_packed, 0);
        kz_pool_free(ctx, q_bias, 0);
        kz_pool_free(ctx, q_key, 0);
        kz_pool_free(ctx, q_key_s, 0);
        kz_pool_free(ctx, q_iota, 0);
        kz_pool_free(ctx, q_sorted, 0);
        kz_pool_free(ctx, row_map, 0);
        kz_pool_free(ctx, theta_s, 0);
        kz_pool_free(ctx, theta_min, 0);
        kz_pool_free(ctx, iota, 0);
        kz_pool_free(ctx, perm, 0);
        kz_pool_free(ctx, theta, 0);
        kz_pool_free(ctx, floor_, 0);
        kz_pool_free(ctx, qnb, 0);
        kz_pool_free(ctx, qfloor, 0);
        kz_pool_free(ctx, col_key, 0);
        kz_pool_free(ctx, ev_cnt, 0);
        kz_pool_free(ctx, col_idx, 0);
        kz_pool_free(ctx, fail_list, 0);
        kz_pool_free(ctx, ev, 0);
        kz_pool_free(ctx, log_keys, 0);
        kz_pool_free(ctx, log_meta, 0);
        kz_pool_free(ctx, d_cnt, 0);
    };
    const size_t tile_bytes = (size_t)n_slices * 4096;
    rc = kz_pool_alloc(ctx, (size_t)s_tiles * tile_bytes + 32 * 4096, (void**)&s_packed);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)s_tiles * KZ_TILE * 4, (void**)&s_bias);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_tiles * tile_bytes + 32 * 4096, (void**)&p_packed);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_tiles * tile_bytes + 32 * 4096, (void**)&q_packed);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&q_bias);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&q_key);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&q_key_s);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&q_iota);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&q_sorted);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&row_map);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&p_bias);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&theta_s);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&theta_min);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&iota);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&perm);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&theta);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4, (void**)&floor_);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&qnb);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b->n * KPr * 4, (void**)&col_key);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b->n * KPr * 4, (void**)&col_idx);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * 4 * KZ_EVC, (void**)&ev_cnt);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b->n * 4, (void**)&fail_list);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)b_pad * ev_cap * 8, (void**)&ev);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)log_cap * 16, &log_keys);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)log_cap * 8, &log_meta);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, KZ_DUAL_CNT_BYTES, (void**)&d_cnt);
    if (rc != KZ_OK) {
        release();
        // (not enough memory for the event buffers: the two ordinary searches need far less)
        return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba);
    }
#define KZ_DUAL_HIP(call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            kz_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);   \
            release();                                                                                 \
            return KZ_ERR_HIP;                                                                         \
        }                                                                                              \
    } while (0)
#define KZ_DUAL_RC(expr)      \
    do {                      \
        rc = (expr);          \
        if (rc != KZ_OK) {    \
            release();        \
            return rc;        \
        }                     \
    } while (0)

    // ---- POPULATION FLOOR of the forward lists (kz_knn.hip "POPULATION FLOOR"): a strided probe of A's rows -- an escalation-style
    // sub-search, exact float64 results written to their places -- gives the model.  It runs FIRST: it is this call's tier probe as well (below), and a call
    // that is handed to two ordinary searches should not have enqueued a sample sweep.
    double floor_model[3] = {0, 0, 0};
    bool have_floor = false;
    bool tier_probed = false;   // the tier probe below has looked at the data
    // (the floor takes ~2 % off a sweep with one list of 16 per query, ~4.5 % with ten; its probe costs 0.4 - 1 ms: sweeps from
    //  T ~ 25 model-ms on -- C2's shared sweep, T = 2.6: 5.94 ms per step with the floor, 5.58 without)
    bool want_floor = ctx->list_floor && KZ_K_FLOOR_PROBE > 0 && a->n >= (int64_t)16 * KZ_K_FLOOR_PROBE &&
                      (ctx->dual_force || t_sweep_ms >= (KP > 16 ? 12.0 : 25.0));   // ("dual_force", the test knob, skips this gate too)
    // The same probe is this call's TIER PROBE (kz_knn_impl): a shared sweep of a size at which an ordinary search would first
    // look whether the data is hard for fp16 as a whole looks too -- more than half of the probe rows uncertified: two ordinary
    // searches instead, each of which starts at the split-bf16 tier (bench.py "hard", 300k x 301k clustered rows: 127 ms per step
    // that way, 201 ms through a shared fp16 sweep whose rows nearly all go down the tiers afterwards).
    const bool want_tier = !ctx->dual_force && ctx->tier_probe > 0 && ctx->esc_bf && a->n >= (int64_t)16 * ctx->tier_probe &&
                           ((double)a->n * (double)b->n >= ctx->probe_min_pairs || t_sweep_ms >= KZ_K_PROBE_MIN_MS);
    // (a tier probe that runs anyway gives the floor for nothing: sweeps of 12 .. 25 model-ms with lists of 16)
    if (want_tier && ctx->list_floor && KZ_K_FLOOR_PROBE > 0 && a->n >= (int64_t)16 * KZ_K_FLOOR_PROBE) want_floor = true;
    if (want_floor || want_tier) {
        const int n_probe = want_floor ? KZ_K_FLOOR_PROBE : (ctx->tier_probe < 1024 ? ctx->tier_probe : 1024);
        const int64_t pstride = a->n / n_probe;
        int* plist = nullptr;
        rc = kz_pool_alloc(ctx, (size_t)n_probe * sizeof(int), (void**)&plist);
        if (rc == KZ_OK && want_floor) rc = kz_pool_alloc(ctx, (size_t)a_pad * 4, (void**)&qfloor);
        kz_knn_stats stp;
        memset(&stp, 0, sizeof(stp));
        float pms = 0;
        if (rc == KZ_OK) {
            hipLaunchKernelGGL(kz_strided_rows_kernel, dim3((unsigned)((n_probe + 255) / 256)), dim3(256), 0, ctx->stream, plist, n_probe, pstride);
            rc = kz_escalate_rows(ctx, a, 0, plist, n_probe, b, k, 0, nullptr, 0, 0, d_dist_ab, d_ind_ab, &stp, &pms);
        }
        if (rc == KZ_OK && want_tier) tier_probed = true;
        // (the rows the probe's FIRST pass left uncertified, once each)
        if (rc == KZ_OK && want_tier && (int64_t)stp.n_first_pass_fail * 8 > n_probe) {
            // LADDER (kz_knn_impl "WIDE ROUTE"), from an EIGHTH of the probe uncertified on, as an ordinary search's: re-searching
            // a quarter of the rows costs more than the sweep (round 5, tools/cliff_probe.py: 300k x 301k x 64, cosine, k = 50, 40
            // tight clusters in random row order -- 44 % of the probe uncertified, below the old "more than half": shared sweep with
            // 314 k rows searched again, 153 ms; wide route 50 ms).  The probe rows again with many lists of 16 on the same fp16
            // operands: at most a quarter uncertified and less than half of what the ordinary lists left -- two ordinary searches on
            // that route; else, with more than half uncertified, two that start at the split-bf16 tier; else the shared sweep.
            int wide = 0;
            if (ctx->wide_lists >= 2) {
                kz_knn_stats stw;
                memset(&stw, 0, sizeof(stw));
                float wms = 0;
                rc = kz_escalate_rows(ctx, a, 0, plist, n_probe, b, k, 0, nullptr, 0, -ctx->wide_lists, d_dist_ab, d_ind_ab, &stw, &wms);
                if (rc == KZ_OK && stw.wide_lists > 0 && (int64_t)stw.n_first_pass_fail * 4 <= n_probe &&
                    (int64_t)stw.n_first_pass_fail * 2 < stp.n_first_pass_fail)
                    wide = ctx->wide_lists;
            }
            if (rc != KZ_OK || wide > 0 || (int64_t)stp.n_first_pass_fail * 2 > n_probe) {
                kz_pool_free(ctx, plist, 0);
                release();
                if (rc != KZ_OK) return rc;
                return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba, wide > 0 ? -1 : 2, wide);
            }
        }
        kz_pool_free(ctx, plist, 0);
        if (rc == KZ_OK && want_floor) rc = kz_floor_model(ctx, d_dist_ab, ia->rowq, n_probe, pstride, k, a->metric, floor_model, &have_floor);
        if (rc != KZ_OK) {
            release();
            return rc;
        }
        if (!have_floor) {
            kz_pool_free(ctx, qfloor, 0);
            qfloor = nullptr;
        }
    }

    // ---- sample sweep's lists.  The threshold is the rank-th best sample key, and the rank-th best of ANY set of distinct sample
    // rows is a valid (lower) threshold.  The sweep therefore never needs lists of K' entries: the sample is cut into `pieces`
    // parts with a list of 16 (32) each, 2 rank entries in all.  A part holds rank / pieces +- sqrt(rank / pieces) of a row's rank
    // best -- provided the parts are alike: part p takes the sample ROWS p, p + pieces, p + 2 pieces, ... (a contiguous range
    // would hold ALL the near rows of a query when the data is stored cluster by cluster, its list of 16 would overflow and the
    // threshold fall to the far rows).  The K' = 16 kernel keeps three workgroups per CU and short merges: 500k x 500k, k = 50:
    // reverse direction 51.0 -> 46.4 ms per step, same event counts.
    auto sample_lists = [&](int rank_, int64_t tiles_, int* kps_out, int* force_out, int head = 2) {   // (entries per row = head x rank)
        int kps = KP, force = 0;
        if (ctx->dual_sample_short && KP > 16) {
            kps = rank_ > 96 ? 32 : 16;
            const int need = (head * rank_ + kps - 1) / kps, cap = 256 / kps;
            force = need < cap ? need : cap;
            if ((int64_t)force * 4 > tiles_ || force * kps < rank_) {   // (a sample of a few tiles: one list of K')
                kps = KP;
                force = 0;
            }
        }
        *kps_out = kps;
        *force_out = force;
    };
    int KPs = KP, force_s = 0;
    sample_lists(rank, s_tiles, &KPs, &force_s);
    KZ_DUAL_HIP(hipEventRecord(ctx->ev[5], ctx->stream));
    KZ_DUAL_HIP(hipMemsetAsync(ev_cnt, 0, (size_t)b_pad * 4 * KZ_EVC, ctx->stream));
    KZ_DUAL_HIP(hipMemsetAsync(d_cnt, 0, KZ_DUAL_CNT_BYTES, ctx->stream));
    // ---- query side: rows dealt into tiles by |q_c|^2 (load balance), its image and its offsets in that order --------------
    hipLaunchKernelGGL(kz_dual_c2key_kernel, dim3((unsigned)((a->n + 255) / 256)), dim3(256), 0, ctx->stream, ia->rowq, a->n, q_key);
    hipLaunchKernelGGL(kz_iota_kernel, dim3((unsigned)((a_pad + 255) / 256)), dim3(256), 0, ctx->stream, q_iota, (int)a_pad);
    KZ_DUAL_HIP(hipGetLastError());
    if (KZ_K_DUAL_DEAL) {
        KZ_DUAL_RC(kz_sort_pairs_f32_i32(ctx, q_key, q_key_s, q_iota, q_sorted, (int)a->n, 0));
        hipLaunchKernelGGL(kz_dual_deal_kernel, dim3((unsigned)((a_pad + 255) / 256)), dim3(256), 0, ctx->stream, q_sorted, a->n, a_pad, row_map);
    } else {   // tuning knob "dual_deal" = 0: the query rows in their natural order
        hipLaunchKernelGGL(kz_dual_natural_kernel, dim3((unsigned)((a_pad + 255) / 256)), dim3(256), 0, ctx->stream, a->n, a_pad, row_map);
    }
    KZ_DUAL_HIP(hipGetLastError());
    KZ_DUAL_RC(kz_himage_pack_permuted(a, row_map, q_packed, q_bias));
    hipLaunchKernelGGL(kz_dual_negbias_kernel, dim3((unsigned)((a_pad + 255) / 256)), dim3(256), 0, ctx->stream, q_bias, a->n, a_pad, qnb);

    // ---- NESTED sample (round 5).  The sample sweep b x sample(a) used to be work the main sweep repeated: the main sweep swept ALL
    // of a, the sampled rows included.  Now the sample S is the FIRST s_tiles tiles of the DEALT image of a (a dealt tile is a
    // stratified draw of a's rows -- sorted by |q_c|^2 and dealt round-robin -- whatever order the caller stored them in), the main
    // sweep covers the other tiles only, and everything the sampled rows need comes out of the sample sweep, which is a shared
    // sweep itself (b as its query side, S sorted by threshold as its index side):
    //   * b's lists over S give tau(t) as before -- and ARE the events of t among the sample rows (kz_dual_theta_kernel, sev);
    //   * the events of the rows of S (thresholds from a third, small sweep S x sample(b)) become their forward lists, certified
    //     and re-ranked by the ordinary finalize kernel with S as the query side (KzRevChain), failures searched again.
    // Saves 1 / stride of the main sweep (ns: stride 20, C3: 11) for a sample sweep that runs the dual build (+5 .. 10 %) and a
    // 1 / stride^2 pre-sample.  Option "dual_nested" (1).
    int rank3 = 0, stride3 = 0, KPs3 = KP, force_s3 = 0, ev_cap3 = 0;
    int64_t s3_tiles = 0;
    long long log_cap3 = 0;
    const int64_t s_img_rows = s_tiles * KZ_TILE;   // rows of S: image rows [0, s_img_rows) of the dealt image
    // (it saves t_sweep / stride and costs ~1 ms of extra launches, sorts and a host synchronisation: C2's shared sweep, 2.6 model-ms at
    //  stride 4, went from 5.7 to 6.7 ms per step with it -- taken from 2 model-ms of saving on; "dual_force" keeps it for the tests)
    bool nested = ctx->dual_nested && s_tiles >= 8 && s_tiles < a->n / KZ_TILE && s_img_rows >= (int64_t)8 * KP &&
                  b->n <= kz_rows_per_chunk(ctx, KP, false) && (ctx->dual_force || t_sweep_ms / stride >= KZ_K_NESTED_MIN_MS);
    if (nested) {
        const double t2_ms = 2.0 * (double)b->n * (double)s_img_rows * (double)(a->kg * 4) / 1e12;
        const int rank3_safe = k + 1 < KP ? k + 1 : KP;
        rank3 = rank3_safe;
        if (ctx->dual_rank > 0)
            rank3 = ctx->dual_rank < rank3_safe ? ctx->dual_rank : rank3_safe;
        else if (ctx->dual_rank == 0 && t2_ms >= 4.0)
            rank3 = kz_dual_pick_rank(k, rank3_safe, t2_ms, (double)b->n, (double)s_img_rows);
        const double s_opt = sqrt(t2_ms / ((double)s_img_rows * rank3 * 0.20e-6));
        stride3 = s_opt < 4.0 ? 4 : (s_opt > 32.0 ? 32 : (int)(s_opt + 0.5));
        const int s_max = (int)(4096.0 / ((double)rank3 + 7.0 * sqrt((double)rank3) + 1.0));
        if (stride3 > s_max) stride3 = s_max;
        s3_tiles = (b_tiles + stride3 - 1) / stride3;
        int64_t s3_rows = s3_tiles * KZ_TILE;
        if ((b_tiles - 1) % stride3 == 0) s3_rows -= b_tiles * KZ_TILE - b->n;
        if (s3_rows < (int64_t)8 * KP || s3_tiles < 2) nested = false;
        sample_lists(rank3, s3_tiles, &KPs3, &force_s3);
        ev_cap3 = (int)(((int64_t)rank3 * stride3 + (int64_t)(7.0 * sqrt((double)rank3) * stride3) + 63) & ~(int64_t)63);
        log_cap3 = (long long)((double)s_img_rows * rank3 * stride3 * 1.5) + (1 << 20);
    }
    // (events of a row among the sample rows: the entries at or above its rank-th best key -- `rank` of them but for ties; a row
    //  with more than the buffer holds is flagged and searched again)
    const int sev_cap = ((2 * rank + 16 + 15) & ~15) < 256 ? ((2 * rank + 16 + 15) & ~15) : 256;
    if (nested) {
        const int KPr3 = ctx->dual_rev_long ? (2 * KP < 128 ? 2 * KP : 128) : KP;
        int rcn = kz_pool_alloc(ctx, (size_t)s3_tiles * tile_bytes + 32 * 4096, (void**)&s3_packed);
        auto al = [&](size_t bytes, void** out) { if (rcn == KZ_OK) rcn = kz_pool_alloc(ctx, bytes, out); };
        al((size_t)s3_tiles * KZ_TILE * 4, (void**)&s3_bias);
        al((size_t)s_tiles * tile_bytes + 32 * 4096, (void**)&ss_packed);
        al((size_t)s_img_rows * 4, (void**)&ss_bias);
        al((size_t)s_img_rows * 4, (void**)&theta3);
        al((size_t)s_img_rows * 4, (void**)&theta3_s);
        al((size_t)s_img_rows * 4, (void**)&theta3_min);
        al((size_t)s_img_rows * 4, (void**)&floor3);
        al((size_t)a_pad * 4, (void**)&floor3m);
        al((size_t)b_pad * 4, (void**)&qnb_b);
        al((size_t)s_img_rows * KPr3 * 4, (void**)&col3_key);
        al((size_t)s_img_rows * KPr3 * 4, (void**)&col3_idx);
        al((size_t)s_img_rows * 4, (void**)&iota3);
        al((size_t)s_img_rows * 4, (void**)&perm3);
        al((size_t)s_img_rows * 4, (void**)&rm3);
        al((size_t)b_pad * 4, (void**)&row_map_b);
        al((size_t)s_img_rows * 4 * KZ_EVC, (void**)&ev3_cnt);
        al((size_t)s_img_rows * 4, (void**)&fail3);
        al((size_t)b_pad * 4, (void**)&sev_cnt);
        al((size_t)s_img_rows * ev_cap3 * 8, (void**)&ev3);
        al((size_t)b->n * sev_cap * 8, (void**)&sev);
        al((size_t)log_cap3 * 16, &log3_keys);
        al((size_t)log_cap3 * 8, &log3_meta);
        al(KZ_DUAL_CNT_BYTES, (void**)&d_cnt3);
        if (rcn != KZ_OK) {   // (no memory for the nested stages: the classic sample sweep)
            release_nested();
            nested = false;
        }
    }
    double nested_sweep_ms = 0;   // the sample sweep b x S: part of the distance matrix the dominant kernel covers (reported with the main sweep)
    KzRevChain rv3;
    memset(&rv3, 0, sizeof(rv3));
    KzDualPass dp2;
    memset(&dp2, 0, sizeof(dp2));
    dp2.probed = 1;   // (the nested sample sweep: its failures are a sample's, the main sweep decides)
    struct Hook2 {
        kz_ctx* ctx; KzDualPass* dp2; KzRevChain* rv3; int rank; int64_t b_n, b_pad; const kz_himage *ia, *ib;
        float *theta, *floor_; const int* rm3; int sev_cap; int* sev_cnt; uint2* sev; int overlap;
    } hk2 = {ctx, &dp2, &rv3, rank, b->n, b_pad, ia, ib, theta, floor_, nullptr, sev_cap, nullptr, nullptr, ctx->dual_overlap};
    // (the nested sample is STRATIFIED by |q_c|^2 -- which correlates with the keys at -0.75: rows near the centre are near
    //  neighbours of everybody -- and the count of rows above the r-th best of a stratified sample is not negative binomial
    //  (r, 1 / stride) but lower: sampled systematically along the key order, the r-th best sample row is the (stride (r - 1) + 1)-th
    //  best overall, (stride - 1)(r - 1) other rows above it against (stride - 1) r at random.  Measured: 5 - 8 % fewer events per
    //  row than the strided-tile sample at every rank, 100 x the rows short of k events on cluster-ordered data (758 against 6 of
    //  12 000).  The threshold of the nested sample therefore sits one rank lower
    //  ... two where a row keeps more than 16 neighbours: cluster-ordered data still came up 5 % short with one.)
    if (nested) {
        const int lower = KP > 16 ? 2 : 1, room = (force_s > 0 ? force_s : 1) * KPs;
        hk2.rank = rank + lower <= room ? rank + lower : (rank + 1 <= room ? rank + 1 : rank);
    }
    if (nested) {
        // ---- third level: S x sample(b) with the ordinary kernel -> the event thresholds of the rows of S ----
        hipLaunchKernelGGL(kz_dual_sample_kernel, dim3((unsigned)s3_tiles), dim3(256), 0, ctx->stream, (const uint4*)ib->packed, ib->bias,
                           n_slices, stride3, (int)s3_tiles, force_s3 > 0 ? force_s3 : 1, (uint4*)s3_packed, s3_bias);
        KZ_DUAL_HIP(hipGetLastError());
        {
            const int KP = KPs3;   // (KZ_DISPATCH_KP switches on `KP`)
            int blocks_per_cu = 1, tpw = 1;
            KZ_DISPATCH_KP(rc, kz_h_occupancy, (n_slices, &blocks_per_cu, &tpw, KZ_K_H_WPS, KZ_K_H_WIDE, KZ_K_LDS_PAD));
            if (rc != KZ_OK) {
                release();
                return rc;
            }
            KzPass ps;
            rc = kz_prepare_pass(ctx, (int)s_tiles, (int)s3_tiles, blocks_per_cu * ctx->n_cus, 256 / KP, KP, KZ_TIER_H, 0, &ps, tpw, force_s3);
            if (rc == KZ_OK) {
                KnnCandParams cp;
                memset(&cp, 0, sizeof(cp));
                cp.qpack = (const float*)q_packed;     // S = the first s_tiles tiles of the dealt image of a
                cp.ypack = (const float*)s3_packed;
                cp.ybias = s3_bias;
                cp.work = ps.d_work;
                cp.qt0 = 0;
                cp.n_ytiles = (int)s3_tiles;
                cp.n_qtiles = (int)s_tiles;
                cp.lay = ps.lay;
                cp.kg = b->kg;
                cp.out_key = ps.out_key;
                cp.out_idx = ps.out_idx;
                KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, ps.W, KZ_K_H_WPS, KZ_K_H_WIDE));
            }
            if (rc != KZ_OK) {
                release();
                return rc;
            }
            // (the rows owning these thresholds are rows of S, position j of the dealt image: bias q_bias[j]; their events come from b)
            hipLaunchKernelGGL(kz_dual_theta_kernel, dim3((unsigned)((s_img_rows + 3) / 4)), dim3(256), 0, ctx->stream, ps.out_key, ps.out_idx,
                               ps.lay, KP, rank3, s_img_rows, s_img_rows, q_bias, ib->d_max, ia->d_max, ia->center->d_scale, theta3, floor3,
                               (const int*)nullptr, 0, (int*)nullptr, (uint2*)nullptr);
            KZ_DUAL_HIP(hipGetLastError());
        }
        // S sorted by descending threshold: perm3 (sorted row -> j), its image from the raw rows, per-tile minima
        hipLaunchKernelGGL(kz_iota_kernel, dim3((unsigned)((s_img_rows + 255) / 256)), dim3(256), 0, ctx->stream, iota3, (int)s_img_rows);
        KZ_DUAL_HIP(hipGetLastError());
        rc = kz_sort_pairs_f32_i32(ctx, theta3, theta3_s, iota3, perm3, (int)s_img_rows, 1);
        if (rc == KZ_OK && force_s > 1) {
            // b's lists over S are kept per index RANGE of this image (force_s lists of 16), and the ranges must be ALIKE: a range
            // that holds most of a row's near sample rows truncates them, the threshold is read too low and the row's event
            // buffer overflows.  Dealing the sorted TILES over the ranges (as the main sweep does) is not enough here: the most
            // central rows of a cluster have the highest thresholds AND are the nearest rows of everybody in it -- one tile, one
            // range (cluster-ordered rows, 12 k x 25 k, k = 50: +53 % events, 1 217 overflowing buffers).  So range p = the S rows
            // with dealt position j in [p L, (p + 1) L), L = the planner's range length -- a contiguous stretch of the DEALT image
            // is a stratified draw of a -- sorted by threshold INSIDE the range: a stable sort by range number on top of the sort
            // by threshold.  Tiles stay coherent in threshold (the per-tile test), ranges are fair.
            const int64_t range_rows = ((s_tiles + force_s - 1) / force_s) * KZ_TILE;
            hipLaunchKernelGGL(kz_dual_rangekey_kernel, dim3((unsigned)((s_img_rows + 255) / 256)), dim3(256), 0, ctx->stream, perm3, s_img_rows, range_rows,
                               theta3, iota3);     // (theta3, iota3: free since the sort above) range number as the key, sorted position as the value
            if (hipGetLastError() != hipSuccess) rc = KZ_ERR_HIP;
            if (rc == KZ_OK) rc = kz_sort_pairs_f32_i32(ctx, theta3, floor3m, iota3, rm3, (int)s_img_rows, 0);   // (floor3m, rm3: scratch here, written later)
            if (rc == KZ_OK) {
                hipLaunchKernelGGL(kz_dual_gather2_kernel, dim3((unsigned)((s_img_rows + 255) / 256)), dim3(256), 0, ctx->stream, rm3, perm3, theta3_s,
                                   s_img_rows, iota3, theta3);
                if (hipGetLastError() != hipSuccess) rc = KZ_ERR_HIP;
                int* ti = perm3; perm3 = iota3; iota3 = ti;
                float* tf = theta3_s; theta3_s = theta3; theta3 = tf;
            }
        }
        if (rc == KZ_OK) {
            hipLaunchKernelGGL(kz_dual_rows_of_kernel, dim3((unsigned)((s_img_rows + 255) / 256)), dim3(256), 0, ctx->stream, perm3, row_map, s_img_rows, rm3);
            hipLaunchKernelGGL(kz_dual_tilemin_kernel, dim3((unsigned)((s_img_rows + 255) / 256)), dim3(256), 0, ctx->stream, theta3_s, s_img_rows, s_img_rows, theta3_min);
            hipLaunchKernelGGL(kz_dual_natural_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, b->n, b_pad, row_map_b);
            hipLaunchKernelGGL(kz_dual_negbias_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, ib->bias, b->n, b_pad, qnb_b);
            rc = kz_himage_pack_rows(a, rm3, s_img_rows, s_img_rows, ss_packed, ss_bias);
        }
        if (rc == KZ_OK && hipMemsetAsync(ev3_cnt, 0, (size_t)s_img_rows * 4 * KZ_EVC, ctx->stream) != hipSuccess) rc = KZ_ERR_HIP;
        if (rc == KZ_OK && hipMemsetAsync(d_cnt3, 0, KZ_DUAL_CNT_BYTES, ctx->stream) != hipSuccess) rc = KZ_ERR_HIP;
        if (rc != KZ_OK) {
            release();
            return rc;
        }
        // ---- second level: b x S with the DUAL build.  Forward lists (raw) -> tau(t), theta(t) and the sample-row events of t;
        // event log -> the forward results of the rows of S (chain rv3, second stream) ----
        rv3.ctx = ctx; rv3.qm = a; rv3.im = b; rv3.qi = ia; rv3.ii = ib;
        rv3.log_keys = log3_keys; rv3.log_meta = log3_meta; rv3.d_cnt = d_cnt3; rv3.log_cap = log_cap3;
        rv3.theta_s = theta3_s; rv3.qnb = qnb_b; rv3.p_bias = ss_bias; rv3.col_key = col3_key; rv3.floor_sel = floor3; rv3.floor_fin = floor3m;
        rv3.row_map = row_map_b; rv3.ev_cnt = ev3_cnt; rv3.perm = perm3; rv3.col_idx = col3_idx; rv3.fail_list = fail3;
        rv3.fail_count = ctx->d_counters + 24; rv3.fin_row_map = row_map;
        rv3.ev = ev3; rv3.ev_cap = ev_cap3; rv3.KP = ctx->dual_rev_long ? (2 * KP < 128 ? 2 * KP : 128) : KP; rv3.k = k;
        rv3.n_rows = s_img_rows; rv3.n_tiles = s_tiles; rv3.d_dist = d_dist_ab; rv3.d_ind = d_ind_ab;
        rv3.timed = 0; rv3.h_fail = 24; rv3.h_cnt = 28; rv3.second_stream = ctx->dual_overlap ? 1 : 0;
        if (kz_spec_rows(ctx, b, k) > 0) KZ_DUAL_RC(kz_spec_alloc(ctx, spec_s3, kz_spec_rows(ctx, b, k), b, k));   // (ahead of the chain: see kz_spec_alloc)
        rv3.spec = &spec_s3;
        dp2.qpack = (const float*)ib->packed;
        dp2.row_map = row_map_b;
        dp2.ypack = (const float*)ss_packed;
        dp2.ybias = ss_bias;
        dp2.perm = rm3;
        dp2.theta = theta3_min;
        dp2.qnbias = qnb_b;
        dp2.log_keys = log3_keys;
        dp2.log_meta = log3_meta;
        dp2.log_cnt = d_cnt3;
        dp2.log_cap = log_cap3;
        dp2.short_pieces = force_s;       // (lists of KPs over force_s parts of S, as the classic sample sweep keeps them)
        dp2.short_kp = KPs;
        dp2.short_ksel = k;
        dp2.n_ytiles = (int)s_tiles;
        dp2.raw_lists = 1;
        dp2.max_entries = 256;
        hk2.rm3 = rm3;
        hk2.sev_cnt = sev_cnt;
        hk2.sev = sev;
        dp2.post_user = &hk2;
        dp2.post_sweep = +[](void* user) -> int {
            Hook2& h = *(Hook2*)user;
            kz_ctx* ctx = h.ctx;
            const KzDualPass& d2 = *h.dp2;
            // thresholds of b's rows off the raw lists (first stream: the lists live in the scratch block until the next pass)
            hipLaunchKernelGGL(kz_dual_theta_kernel, dim3((unsigned)((h.b_pad + 3) / 4)), dim3(256), 0, ctx->stream, d2.lists_key, d2.lists_idx,
                               d2.lists_lay, d2.lists_KP, h.rank, h.b_n, h.b_pad, h.ib->bias, h.ia->d_max, h.ib->d_max, h.ib->center->d_scale,
                               h.theta, h.floor_, h.rm3, h.sev_cap, h.sev_cnt, h.sev);
            KZ_HIP(hipGetLastError());
            return kz_dual_enqueue_chain(*h.rv3);   // the sample rows' own results, beside what follows on the first stream
        };
        kz_knn_stats st2;
        ctx->stream2_busy = 1;
        rc = kz_knn_impl(ctx, b, 0, b->n, a, k, 0, nullptr, -1, 0, d_dist_ba, d_ind_ba, &st2, &dp2);
        if (rc == KZ_OK && (dp2.broken || !dp2.post_called)) {
            // the sweep left the fp16 tier (cannot happen for a dual pass today) -- no thresholds: give the call to two ordinary searches
            (void)hipStreamSynchronize(ctx->stream2);
            ctx->stream2_busy = 0;
            release();
            return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba);
        }
        if (rc != KZ_OK) {
            (void)hipStreamSynchronize(ctx->stream2);
            ctx->stream2_busy = 0;
            release();
            return rc;
        }
        nested_sweep_ms = st2.main_kernel_ms;
    } else {
        // ---- classic sample image: every stride-th tile of A's fp16 image (tiles are contiguous runs of n_slices x 4 KiB), dealt over the parts
        hipLaunchKernelGGL(kz_dual_sample_kernel, dim3((unsigned)s_tiles), dim3(256), 0, ctx->stream, (const uint4*)ia->packed, ia->bias,
                           n_slices, stride, (int)s_tiles, force_s > 0 ? force_s : 1, (uint4*)s_packed, s_bias);
        KZ_DUAL_HIP(hipGetLastError());
        // ---- sample sweep: B x sample(A) with the ordinary kernel, lists of at most 256 entries per row -----------------------
        const int KP = KPs;   // (KZ_DISPATCH_KP switches on `KP`)
        int blocks_per_cu = 1, tpw = 1;
        KZ_DISPATCH_KP(rc, kz_h_occupancy, (n_slices, &blocks_per_cu, &tpw, KZ_K_H_WPS, KZ_K_H_WIDE, KZ_K_LDS_PAD));
        if (rc != KZ_OK) {
            release();
            return rc;
        }
        KzPass ps;
        KZ_DUAL_RC(kz_prepare_pass(ctx, (int)b_tiles, (int)s_tiles, blocks_per_cu * ctx->n_cus, 256 / KP, KP, KZ_TIER_H, 0, &ps, tpw, force_s));
        KnnCandParams cp;
        memset(&cp, 0, sizeof(cp));
        cp.qpack = (const float*)ib->packed;
        cp.ypack = (const float*)s_packed;
        cp.ybias = s_bias;
        cp.work = ps.d_work;
        cp.qt0 = 0;
        cp.n_ytiles = (int)s_tiles;
        cp.n_qtiles = (int)b_tiles;
        cp.lay = ps.lay;
        cp.kg = b->kg;
        cp.out_key = ps.out_key;
        cp.out_idx = ps.out_idx;
        KZ_DISPATCH_KP(rc, kz_h_launch, (n_slices, ctx, cp, ps.W, KZ_K_H_WPS, KZ_K_H_WIDE));
        if (rc != KZ_OK) {
            release();
            return rc;
        }
        hipLaunchKernelGGL(kz_dual_theta_kernel, dim3((unsigned)((b_pad + 3) / 4)), dim3(256), 0, ctx->stream, ps.out_key, ps.out_idx,
                           ps.lay, KP, rank, b->n, b_pad, ib->bias, ia->d_max, ib->d_max, ib->center->d_scale, theta, floor_,
                           (const int*)nullptr, 0, (int*)nullptr, (uint2*)nullptr);
        KZ_DUAL_HIP(hipGetLastError());
    }
    // ---- B's rows in DESCENDING order of their threshold: permutation, sorted thresholds (+inf behind them), sorted image.
    // Descending = rows with a near K'-th neighbour first: those are the rows that ARE near neighbours of many queries, so the
    // forward lists meet their best candidates early and their thresholds tighten at once (ascending order is the
    // adversarial one: candidates keep improving over the whole sweep -- 4.7 against 2.7 list inserts per wave and tile) ----
    hipLaunchKernelGGL(kz_iota_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, iota, (int)b_pad);
    hipLaunchKernelGGL(kz_dual_fill_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, theta_s, b_pad, INFINITY);
    KZ_DUAL_HIP(hipGetLastError());
    KZ_DUAL_RC(kz_sort_pairs_f32_i32(ctx, theta, theta_s, iota, perm, (int)b->n, 1));
    // ---- short-list route of the main sweep (kz_knn_impl): k / 5 lists of 16 per query instead of one of 32 / 64 / 128.  Rows with
    // neighbouring thresholds tend to be neighbours of the same queries, so the sorted tiles are dealt over the index ranges.
    int main_pieces = 0;
    if (ctx->dual_short_main && KP > 16) {
        // (a range of at least 64 tiles: the k nearest rows of a query must be spread over many more tiles than there are ranges)
        const int per = KZ_K_DUAL_SHORT_DIV * (KZ_K_DUAL_SHORT_KP / 16);
        const int P = (k + per - 1) / per;
        if (P >= 2 && P * KZ_K_DUAL_SHORT_KP <= 512 && KP > KZ_K_DUAL_SHORT_KP && b_tiles - 1 >= (int64_t)ctx->dual_short_min_tiles * P) main_pieces = P;
    }
    if (main_pieces > 0) {
        // (iota and theta -- the sort's inputs -- are free now: they take the dealt order)
        hipLaunchKernelGGL(kz_dual_interleave_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, perm, theta_s, b->n,
                           b_pad, main_pieces, iota, theta);
        KZ_DUAL_HIP(hipGetLastError());
        int* ti = perm; perm = iota; iota = ti;
        float* tf = theta_s; theta_s = theta; theta = tf;
    }
    hipLaunchKernelGGL(kz_dual_tilemin_kernel, dim3((unsigned)((b_pad + 255) / 256)), dim3(256), 0, ctx->stream, theta_s, b->n, b_pad, theta_min);
    KZ_DUAL_HIP(hipGetLastError());
    KZ_DUAL_RC(kz_himage_pack_permuted(b, perm, p_packed, p_bias));
    if (nested) {
        // the sample-row events of every row of b open its event buffer (the main sweep's scatter kernel appends behind them)
        // (rows [0, b->n) of the sorted image: the permutation is only defined there without the dealt tiles, and they keep the
        //  ragged last tile last)
        hipLaunchKernelGGL(kz_dual_inject_kernel, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, ctx->stream, perm, b->n, sev_cnt, sev, sev_cap,
                           ev_cnt, ev, ev_cap);
        KZ_DUAL_HIP(hipGetLastError());
    }
    KZ_DUAL_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
    float sample_ms = 0;
    if (qfloor) {
        hipLaunchKernelGGL(kz_floor_rows_kernel, dim3((unsigned)((a_pad + 255) / 256)), dim3(256), 0, ctx->stream, row_map, a->n, a_pad, ia->rowq,
                           ib->d_max, ib->center->d_scale, floor_model[0], floor_model[1], floor_model[2], ctx->eps_scale,
                           kz_gamma_acc_h(b->kg), qfloor);
        KZ_DUAL_HIP(hipGetLastError());
    }

    // ---- main sweep: A x B, lists of A's rows + event log of B's rows ---------------------------------------------------------
    KzDualPass dp;
    memset(&dp, 0, sizeof(dp));
    dp.probed = tier_probed ? 1 : 0;
    dp.qpack = (const float*)q_packed;
    dp.row_map = row_map;
    dp.ypack = (const float*)p_packed;
    dp.ybias = p_bias;
    dp.perm = perm;
    dp.theta = theta_min;
    dp.qnbias = qnb;
    dp.qfloor = qfloor;
    dp.log_keys = log_keys;
    dp.log_meta = log_meta;
    dp.log_cnt = d_cnt;
    dp.log_cap = log_cap;
    dp.short_pieces = main_pieces;
    dp.short_ksel = k + ctx->dual_short_extra < main_pieces * KZ_K_DUAL_SHORT_KP ? k + ctx->dual_short_extra : main_pieces * KZ_K_DUAL_SHORT_KP;
    dp.short_kp = KZ_K_DUAL_SHORT_KP;
    // ---- the reverse direction's chain: events -> lists -> ordinary finalize with B as the query side.  Enqueued on the
    // context's SECOND stream from inside kz_knn_impl, right behind the sweep (KzDualPass::post_sweep): it shares no buffer
    // with what the first stream does meanwhile (finalize of A's lists, fail-counter read-back, re-search of uncertified rows)
    int* fail_count_b = ctx->d_counters + 12;   // {fail counter, -, error-ratio bits x 2}: a set of its own beside the forward direction's
    KzRevChain rv;
    memset(&rv, 0, sizeof(rv));
    rv.ctx = ctx; rv.qm = b; rv.im = a; rv.qi = ib; rv.ii = ia;
    rv.log_keys = log_keys; rv.log_meta = log_meta; rv.d_cnt = d_cnt; rv.log_cap = log_cap;
    rv.theta_s = theta_s; rv.qnb = qnb; rv.p_bias = p_bias; rv.col_key = col_key; rv.floor_sel = floor_; rv.floor_fin = floor_;
    rv.row_map = row_map; rv.ev_cnt = ev_cnt; rv.perm = perm; rv.col_idx = col_idx; rv.fail_list = fail_list; rv.fail_count = fail_count_b;
    rv.ev = ev; rv.ev_cap = ev_cap; rv.KP = KPr; rv.k = k; rv.n_rows = b->n; rv.n_tiles = b_tiles; rv.d_dist = d_dist_ba; rv.d_ind = d_ind_ba;
    rv.timed = 1; rv.h_fail = 12; rv.h_cnt = 16; rv.second_stream = 1;
    if (kz_spec_rows(ctx, a, k) > 0) KZ_DUAL_RC(kz_spec_alloc(ctx, spec_ba, kz_spec_rows(ctx, a, k), a, k));   // (ahead of the chain: see kz_spec_alloc)
    rv.spec = &spec_ba;
    auto enqueue_reverse = [](void* user) -> int { return kz_dual_enqueue_chain(*(KzRevChain*)user); };
    if (ctx->dual_overlap) {
        dp.post_sweep = +enqueue_reverse;
        dp.post_user = &rv;
    }
    kz_knn_stats st_ab;
    ctx->stream2_busy = 1;   // (from here to the synchronisation below the second stream belongs to the reverse chain)
    // (nested: the main sweep covers the image rows behind the sample -- in the dual pass a query range is a range of IMAGE rows)
    rc = kz_knn_impl(ctx, a, nested ? s_img_rows : 0, nested ? a->n - s_img_rows : a->n, b, k, 0, nullptr, -1, 0, d_dist_ab, d_ind_ab, &st_ab, &dp);
    if (rc == KZ_OK && !ctx->dual_overlap && !dp.broken) {   // ("dual_overlap" = 0: the same chain, behind the forward direction)
        dp.post_called = 1;
        rc = enqueue_reverse(&rv);
    }
    // whatever happened on the first stream: the second one is done with the buffers before anything is released
    {
        const hipError_t e2 = hipStreamSynchronize(ctx->stream2);
        ctx->stream2_busy = 0;
        if (rc == KZ_OK && e2 != hipSuccess) {
            kz_set_error("kz_knn_dual: second stream failed: %s", hipGetErrorString(e2));
            rc = KZ_ERR_HIP;
        }
    }
    if (rc != KZ_OK) {
        release();
        return rc;
    }
    // (kz_knn_impl ends with a stream synchronisation: ev[5] and ev[6] around the sample sweep have completed)
    KZ_DUAL_HIP(hipEventElapsedTime(&sample_ms, ctx->ev[5], ctx->ev[6]));
    st_ab.dual = 1;
    st_ab.main_kernel_ms += nested_sweep_ms;   // (nested: S x B is swept by the sample sweep, the rest of A x B by the main sweep)
    if (nested) {
        // the sample rows' own chain (second stream, synchronised above): rows it could not certify -- or all of them when its
        // event log overflowed -- are searched again the ordinary way, like the uncertified rows of the main sweep
        int n_fail3 = ctx->h_counters[24];
        unsigned long long hc3[4];
        memcpy(hc3, ctx->h_counters + 28, 32);
        const int* list3 = fail3;
        if (hc3[0] > (unsigned long long)log_cap3) {
            n_fail3 = (int)s_img_rows;
            list3 = rm3;   // (the matrix rows of S)
        }
        const bool rescued3 = hc3[0] <= (unsigned long long)log_cap3 && spec_s3.R > 0 && n_fail3 > 0 && n_fail3 <= spec_s3.R;
        if (rescued3) {   // (the speculative exact launches behind the chain's finalize have answered them)
            st_ab.n_fallback_rows += n_fail3;
            st_ab.n_spec_rows += n_fail3;
        } else if (n_fail3 > 0) {
            const int kp_min3 = ((int64_t)n_fail3 * 8 > s_img_rows || KPr >= 128) ? 0 : (KPr == 16 && KZ_K_ESC_SHORT && n_fail3 <= KZ_ESC_SHORT_MAX_ROWS ? -1 : (KPr * 4 < 128 ? KPr * 4 : 128));
            kz_knn_stats st3;
            float ms3 = 0;
            const int prec3 = ((int64_t)n_fail3 * 2 > s_img_rows && ctx->esc_bf) ? 2 : -1;
            KZ_DUAL_RC(kz_escalate_rows(ctx, a, 0, list3, n_fail3, b, k, 0, nullptr, prec3, prec3 == 2 ? 0 : kp_min3, d_dist_ab, d_ind_ab, &st3, &ms3));
            st_ab.fallback_ms += ms3;
            st_ab.n_escalated_rows += n_fail3 + st3.n_escalated_rows;
            st_ab.n_fallback_rows += st3.n_fallback_rows;
            if (st3.max_err_ratio > st_ab.max_err_ratio) st_ab.max_err_ratio = st3.max_err_ratio;
        }
        double r3;
        memcpy(&r3, ctx->h_counters + 26, 8);
        if (r3 > st_ab.max_err_ratio) st_ab.max_err_ratio = r3;
        st_ab.n_first_pass_fail += n_fail3;
    }
    kz_dual_fill_stats(stats_ab, st_ab);

    kz_knn_stats st_ba;
    memset(&st_ba, 0, sizeof(st_ba));
    if (!dp.broken && !dp.post_called) dp.broken = 1;   // (the sweep never got as far as its last chunk in the fp16 tier)
    if (!dp.broken) {
        const int n_fail = ctx->h_counters[12];
        unsigned long long hc[4];
        memcpy(hc, ctx->h_counters + 16, 32);
        memcpy(&st_ba.max_err_ratio, ctx->h_counters + 14, 8);
        float ms = 0;
        KZ_DUAL_HIP(hipEventElapsedTime(&ms, ctx->ev[8], ctx->ev[9]));
        st_ba.main_kernel_ms = sample_ms - (float)nested_sweep_ms + ms;   // sample stage (without the part of the matrix it covers for both directions) + scatter + select: what this direction cost besides the shared sweep
        KZ_DUAL_HIP(hipEventElapsedTime(&ms, ctx->ev[9], ctx->ev[10]));
        st_ba.finalize_ms = ms;
        st_ba.list_len = KPr;
        st_ba.n_splits = 1;
        st_ba.first_pass = KZ_TIER_H;
        st_ba.dual = 1;
        st_ba.n_events = (int64_t)hc[1];
        st_ba.n_overflow_rows = (int64_t)hc[2];
        st_ba.n_logged_groups = (int64_t)(hc[0] & ((1ull << 62) - 1));
        if (hc[0] > (unsigned long long)log_cap) {
            dp.broken = 1;   // the log itself overflowed: events are missing for unknown rows
        } else if (spec_ba.R > 0 && n_fail > 0 && n_fail <= spec_ba.R) {
            st_ba.n_fallback_rows = n_fail;   // (answered by the speculative exact launches behind the chain's finalize)
            st_ba.n_spec_rows = n_fail;
        } else if (n_fail > 0) {
            // rows of B with an overflowing buffer or an uncertified list: the ordinary search, longer lists when they are few
            // (K' = 16: more lists instead of longer ones, kz_knn_impl kp_min = -1)
            const int kp_min = ((int64_t)n_fail * 8 > b->n || KPr >= 128) ? 0 : (KPr == 16 && KZ_K_ESC_SHORT && n_fail <= KZ_ESC_SHORT_MAX_ROWS ? -1 : (KPr * 4 < 128 ? KPr * 4 : 128));
            kz_knn_stats st2;
            // (more than half of B's rows: fp16 is the wrong tier for this data -- the split-bf16 operands at once)
            const int prec = ((int64_t)n_fail * 2 > b->n && ctx->esc_bf) ? 2 : -1;
            if (prec == 2 && !tier_probed)   // (no probe has looked at this data: the ladder on the failed rows, kz_knn.hip "LADDER AFTER THE FACT")
                KZ_DUAL_RC(kz_escalate_ladder(ctx, b, 0, fail_list, n_fail, a, k, 0, nullptr, prec, 0, d_dist_ba, d_ind_ba, &st2, &ms));
            else
                KZ_DUAL_RC(kz_escalate_rows(ctx, b, 0, fail_list, n_fail, a, k, 0, nullptr, prec, prec == 2 ? 0 : kp_min, d_dist_ba, d_ind_ba, &st2, &ms));
            st_ba.fallback_ms = ms;
            st_ba.n_escalated_rows = n_fail + st2.n_escalated_rows;
            st_ba.n_fallback_rows = st2.n_fallback_rows;
            if (st2.max_err_ratio > st_ba.max_err_ratio) st_ba.max_err_ratio = st2.max_err_ratio;
        }
    }
    release();
    if (dp.broken == 2) {
        // the forward sweep gave up on a chunk (too many uncertified rows for the fp16 tier): both directions the ordinary way
        return kz_knn_dual_separately(ctx, a, b, k, d_dist_ab, d_ind_ab, d_dist_ba, d_ind_ba, stats_ab, stats_ba);
    }
    if (dp.broken) {
        // the sweep left the fp16 tier on the way, or the log overflowed: this direction the ordinary way
        rc = kz_knn_impl(ctx, b, 0, b->n, a, k, 0, nullptr, -1, 0, d_dist_ba, d_ind_ba, &st_ba, nullptr);
        if (rc != KZ_OK) return rc;
    }
    kz_dual_fill_stats(stats_ba, st_ba);
    return KZ_OK;
#undef KZ_DUAL_HIP
#undef KZ_DUAL_RC
}
