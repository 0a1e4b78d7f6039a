// Finalize of a query with MANY selected candidates (KSEL > 160: the fp16 tier's wide route -- 32 lists of 16, 256 selected -- and
// the long-k route).  Same contract and same certification as kz_finalize_query (kz_knn.hip), which it falls back to for rows it
// is not built for; what differs is the cost per candidate.  Round 5: on bench.py "hard" (300k queries, 512 list entries each, ~250
// candidates within 2 eps of the k-th key) the generic path took 47 ms per launch, three quarters of it in three O(n^2) rank sorts
// and in a re-rank that used 16 of 64 lanes.  Here:
//   * NOTHING is sorted that does not have to be: the KS best approximate keys by radix select + compaction (unordered); the k-th
//     best of them by a second radix select; the candidates within 2 eps of it by compaction; after the float64 re-rank the
//     (k + 1) smallest exact values by a radix select over the float64 bit patterns, and only those few are rank-sorted;
//   * the re-rank works on SEVERAL candidate rows per wave step when a row needs fewer than 64 lanes (d <= 128: two, d <= 64:
//     four, d <= 32: eight): lane group g takes candidate c + g, lane s of a group the elements 4 s .. 4 s + 3, and the butterfly
//     sum runs inside the group.  Bit-identical to kz_wave_dot: there the lanes past the row hold exact zeros, which the first
//     butterfly steps add without changing a bit; the steps inside the group are the same additions in the same order.
#pragma once

template <typename T>
__device__ __forceinline__ void kz_finalize_query_wide(const KnnFinParams& p, const int64_t q, const int lane, char* wbase) {
    const int KS = p.KSEL;
    const int64_t qrow = p.row_map ? (int64_t)p.row_map[p.q_begin + q] : p.q_begin + q;
    const T* qptr = reinterpret_cast<const T*>(p.qraw) + qrow * (int64_t)p.d;
    const T* yraw = reinterpret_cast<const T*>(p.yraw);
    // what this path is built for is decided per LAUNCH on the host (kz_launch_finalize: float32 rows on the fp16 tier, ordinary
    // direction, d <= 256, 16-byte aligned rows); the generic path used to be inlined here as a per-query fallback and cost the
    // kernel 57 spilled VGPRs at its three waves per SIMD
    // LDS of a wave (kz_fin_wide_wave_bytes: 8 max_m + 20 KS bytes -- 9 KiB at 512 entries / 256 selected: FOUR workgroups per CU;
    // with every array on its own, 11.3 KiB, it was three, and the kernel is bound by latency; five -- si in the free half of eidx --
    // measured no faster):
    //   cv   [KS] float64   exact values of the re-ranked candidates; before the re-rank: scratch for sortable key patterns (su)
    //   ekey / eidx [max_m] the list copy; after the selection: the candidates within 2 eps of the k-th key; eidx later the ids of
    //                       the long-k sort's output
    //   ck / ci [KS]        the KS selected keys and rows -- dead after the 2-eps compaction, then sv [KS] float64: the k + 1 smallest values
    //   si   [KS]           their rows
    double* cv = reinterpret_cast<double*>(wbase);
    float* ekey = reinterpret_cast<float*>(cv + KS);
    int* eidx = reinterpret_cast<int*>(ekey + p.max_m);
    float* ck = reinterpret_cast<float*>(eidx + p.max_m);
    int* ci = reinterpret_cast<int*>(ck + KS);
    double* sv = reinterpret_cast<double*>(ck);
    int* si = ci + KS;
    const int KP = p.KP;
    const int k_eff = p.k + (p.exclude_self ? 1 : 0);
    const int64_t qout = p.row_map ? qrow : q;
    const double qs = p.qsqn[qrow];
    auto fail = [&](const double tau = (double)INFINITY) {   // (tau: KnnFinParams::fail_tau)
        if (lane == 0) {
            const int pos = atomicAdd(p.fail_count, 1);
            p.fail_list[pos] = (int)qout;
            if (p.fail_tau) p.fail_tau[pos] = tau;
        }
    };
    const int64_t lrow = p.list_row0 + q;
    const int M = p.lay.pieces[kz_list_region(lrow, p.lay)] * p.lay.halves * KP;
    {
        const int64_t l0 = kz_list_contig_off(lrow, p.lay, KP, 0);   // (fp16 tier: contiguous lists)
        for (int e = lane; e < M; e += 64) {
            ekey[e] = p.in_key[l0 + e];
            int r = p.in_idx[l0 + e];
            if (p.idx_map && r >= 0) r = p.idx_map[r];
            eidx[e] = r;
        }
    }
    kz_wave_sync();
    // a FULL list may have evicted rows (at or below its smallest key); the lists' own floor
    float piece_bound = p.list_floor ? p.list_floor[p.q_begin + q] : -INFINITY;
    if (KP <= 32 && M <= 512) {
        // (all lists of a 64-entry block at once, inside their lane groups: kz_full_lists_bound)
        float kk[8];
        bool okv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            okv[i] = e < M && eidx[e] >= 0;
            kk[i] = e < M ? ekey[e] : INFINITY;
        }
        piece_bound = fmaxf(piece_bound, kz_full_lists_bound<8>(kk, okv, M, KP, lane));
    } else {
        for (int l0 = 0; l0 < M; l0 += KP) {
            float mn = INFINITY;
            int cnt = 0;
            for (int e = lane; e < KP; e += 64) {
                const bool ok = eidx[l0 + e] >= 0;
                cnt += ok ? 1 : 0;
                mn = ok ? fminf(mn, ekey[l0 + e]) : mn;
            }
            if (KP >= 64) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    cnt += __shfl_xor(cnt, off, 64);
                    mn = fminf(mn, __shfl_xor(mn, off, 64));
                }
            } else {   // (lists of 16 / 32: the entries sit in the first lanes)
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    cnt += __shfl_xor(cnt, off, 64);
                    mn = fminf(mn, __shfl_xor(mn, off, 64));
                }
                cnt = __builtin_amdgcn_readfirstlane(cnt);
                mn = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(mn)));
            }
            if (cnt == KP) piece_bound = fmaxf(piece_bound, mn);
        }
    }
    // ---- the KS best approximate keys, unordered, into ck / ci; sel_bound = the largest key left behind -----------------
    unsigned* uk = reinterpret_cast<unsigned*>(ekey);
    auto key_of = [](unsigned u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); };
    int nv = 0;
    for (int e = lane; e < M; e += 64) {
        unsigned bts = __float_as_uint(ekey[e]);
        if (bts == 0x80000000u) bts = 0u;
        const bool valid = eidx[e] >= 0;
        uk[e] = valid ? (bts ^ ((bts >> 31) ? 0xffffffffu : 0x80000000u)) : 0u;   // (a valid key is never the pattern 0: that is -NaN)
        nv += valid ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) nv += __shfl_xor(nv, off, 64);
    kz_wave_sync();
    float sel_bound = -INFINITY;
    int V = 0;
    {
        const bool all = nv <= KS;
        unsigned thr = 0u;
        if (!all) thr = kz_radix_kth_u32<true>(uk, M, KS, lane);   // (invalid entries carry the smallest pattern: they never reach rank KS)
        for (int e0 = 0; e0 < M; e0 += 64) {
            const int e = e0 + lane;
            const bool sel = e < M && eidx[e] >= 0 && (all || uk[e] > thr);
            const unsigned long long mask = __ballot(sel);
            if (sel) {
                const int pos = V + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                ck[pos] = key_of(uk[e]);
                ci[pos] = eidx[e];
            }
            V += (int)__popcll(mask);
        }
        if (!all) {
            // entries AT the threshold fill the remaining places, smallest rows first (as kz_finalize_query does); whatever is left
            // behind has a key <= key_of(thr)
            int last = -1;
            while (V < KS) {
                int best = 0x7fffffff;
                for (int e = lane; e < M; e += 64) {
                    const int xi = eidx[e];
                    if (xi >= 0 && uk[e] == thr && xi > last && xi < best) best = xi;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) best = min(best, __shfl_xor(best, off, 64));
                if (best == 0x7fffffff) break;
                if (lane == 0) {
                    ck[V] = key_of(thr);
                    ci[V] = best;
                }
                last = best;
                ++V;
            }
            sel_bound = key_of(thr);
        }
    }
    kz_wave_sync();
    if (V < k_eff) {   // (cannot be certified: fewer candidates than neighbours asked for)
        fail();
        return;
    }
    // ---- rounding bound (the fp16 tier's, kz_finalize_query) -----------------------------------------------------------
    const double qc2 = p.q_rowq[qrow * 3 + 0], qh = p.q_rowq[qrow * 3 + 1], qr = p.q_rowq[qrow * 3 + 2];
    const double Yh = p.y_hmax[0], Ry = p.y_hmax[1], Yc2 = p.y_hmax[2];
    const double qc = sqrt(qc2), yc = sqrt(Yc2);
    const double ymax = p.ystats[0];
    const double raw2 = p.metric == KZ_COSINE ? 2.0 : qs + ymax * ymax;
    const double eps_q = p.eps_mult * (qr * Yh + qh * Ry + qr * Ry + p.gamma_acc * (0.5 * Yc2 + qh * Yh) +
                                       1.1920928955078125e-07 * (qc + yc) * (qc + yc) + 1e-12 * (0.5 * Yc2 + qc2) + 1e-14 * raw2);
    const double key_scale = p.hscale[1];
    auto exact_key = [&](double v) { return 0.5 * (qc2 - (p.metric == KZ_COSINE ? 2.0 * v : v)); };
    // ---- the k-th best approximate key; candidates within 2 eps of it go to the front (ekey / eidx: the list copy is spent) ----
    unsigned* su = reinterpret_cast<unsigned*>(cv);   // (scratch: cv is written by the re-rank)
    for (int c = lane; c < V; c += 64) {
        const unsigned b = __float_as_uint(ck[c]);
        su[c] = (b == 0x80000000u ? 0u : b) ^ (((b == 0x80000000u ? 0u : b) >> 31) ? 0xffffffffu : 0x80000000u);
    }
    kz_wave_sync();
    const float key_k = key_of(kz_radix_kth_u32<true>(su, V, k_eff, lane));
    const double thr2 = (double)key_k * key_scale - 2.0 * eps_q;
    int Vr = 0;
    float left_max = -INFINITY;
    for (int c0 = 0; c0 < V; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < V && (double)ck[c] * key_scale >= thr2;
        const unsigned long long mask = __ballot(in);
        if (in) {
            const int pos = Vr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            ekey[pos] = ck[c];
            eidx[pos] = ci[c];
        } else if (c < V) {
            left_max = fmaxf(left_max, ck[c]);
        }
        Vr += (int)__popcll(mask);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) left_max = fmaxf(left_max, __shfl_xor(left_max, off, 64));
    kz_wave_sync();
    // ---- exact float64 values of the Vr candidates: G rows per wave step ---------------------------------------------------
    auto rerank = [&](auto lpr_c, auto norm_c) {
        constexpr int LPR = decltype(lpr_c)::value;   // lanes per row
        constexpr bool NORM = decltype(norm_c)::value;   // cosine: the index rows come normalised in float64 (kz_matrix_norm64)
        constexpr int G = 64 / LPR;
        const int grp = lane / LPR, sl = lane & (LPR - 1);
        const int k0 = 4 * sl;
        const bool act = k0 < p.d;
        const int k0r = act ? k0 : 0;
        double qk[4] = {0.0, 0.0, 0.0, 0.0};
        if (act) {
            kz_row4(qptr, k0, p.d, true, qk);
            if (p.metric == KZ_COSINE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) qk[e] = qk[e] / qs;
            }
        }
        const int steps = (Vr + G - 1) / G;
        struct Buf {
            float4 f;        // raw row elements               (!NORM)
            double ys;       // the row's norm / squared norm  (!NORM)
            double2 n0, n1;  // normalised row elements        (NORM)
        };
        auto issue = [&](int st, Buf& b) {
            // (no load under a condition: steps and candidates past the end repeat the last candidate, see kz_finalize_query)
            const int yi = eidx[min(min(st, steps - 1) * G + grp, Vr - 1)];
            if (NORM) {
                const double* row = p.ynorm64 + (int64_t)yi * p.d + k0r;
                b.n0 = *reinterpret_cast<const double2*>(row);
                b.n1 = *reinterpret_cast<const double2*>(row + 2);
            } else {
                b.ys = p.ysqn[yi];
                b.f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(yraw) + (int64_t)yi * p.d + k0r);
            }
        };
        auto reduce = [&](int st, const Buf& b) {
            double a = 0.0;
            if (NORM) {
                if (act) {
                    const double yn[4] = {b.n0.x, b.n0.y, b.n1.x, b.n1.y};   // = y_e / |y|: the quotients of the branch below
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fma(qk[e], yn[e], a);
                }
            } else if (act) {
                const double yk[4] = {(double)b.f.x, (double)b.f.y, (double)b.f.z, (double)b.f.w};
                const double ysb = b.ys;
                if (p.metric == KZ_COSINE) {
                    bool done = false;
                    if (p.fast_div) {
                        const double rcp = 1.0 / ysb;
                        if ((((unsigned long long)__double_as_longlong(rcp) >> 52) & 0x7ff) != 0x7ff) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a = fma(qk[e], kz_div_shared(yk[e], ysb, rcp), a);
                            done = true;
                        }
                    }
                    if (!done) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a = fma(qk[e], yk[e] / ysb, a);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = fma(qk[e], yk[e], a);
                }
            }
            // the butterfly of kz_wave_sum inside the lane group (its first log2(G) steps add the zeros of the lanes past the row)
#pragma unroll
            for (int off = LPR >> 1; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
            double v;
            if (p.metric == KZ_COSINE)
                v = fmin(fmax(1.0 - a, 0.0), 2.0);
            else
                v = fmax((qs + (NORM ? 0.0 : b.ys)) - 2.0 * a, 0.0);
            const int c = st * G + grp;
            if (sl == 0 && c < Vr) cv[c] = v;
        };
        Buf b0, b1, b2;
        issue(0, b0);
        issue(1, b1);
        for (int st = 0;;) {   // (all conditions wave-uniform; three steps in flight)
            issue(st + 2, b2);
            reduce(st, b0);
            if (++st >= steps) break;
            issue(st + 2, b0);
            reduce(st, b1);
            if (++st >= steps) break;
            issue(st + 2, b1);
            reduce(st, b2);
            if (++st >= steps) break;
        }
    };
    {
        const int lanes_needed = (p.d + 3) >> 2;   // (wave-uniform)
        const bool norm = p.metric == KZ_COSINE && p.ynorm64 != nullptr;
        auto go = [&](auto lpr_c) {
            if (norm)
                rerank(lpr_c, std::true_type{});
            else
                rerank(lpr_c, std::false_type{});
        };
        if (lanes_needed <= 8)
            go(std::integral_constant<int, 8>{});
        else if (lanes_needed <= 16)
            go(std::integral_constant<int, 16>{});
        else if (lanes_needed <= 32)
            go(std::integral_constant<int, 32>{});
        else
            go(std::integral_constant<int, 64>{});
    }
    kz_wave_sync();
    // ---- self-check of the rounding bound on every re-ranked candidate -------------------------------------------------------
    bool bound_violated = false;
    if (eps_q > 0.0 && eps_q < INFINITY && p.err_ratio_bits) {
        double worst = 0.0;
        for (int c = lane; c < Vr; c += 64) {
            const double v = cv[c];
            if (v > 0.0) worst = fmax(worst, fabs((double)ekey[c] * key_scale - exact_key(v)) / eps_q);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off, 64));
        bound_violated = worst > 1.0;
        if (lane == 0 && worst > 0.0) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(worst);
            if (bits > *(const volatile unsigned long long*)p.err_ratio_bits) atomicMax(p.err_ratio_bits, bits);
        }
    }
    // ---- the (k_eff + 1) smallest exact values, sorted by (value, row): everything the output and the certification read ------
    const int need = Vr < k_eff + 1 ? Vr : k_eff + 1;
    const unsigned long long tbits = kz_radix_kth_small_f64<true>(cv, Vr, need, lane);
    int ns = 0;
    for (int c0 = 0; c0 < Vr; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < Vr && (unsigned long long)__double_as_longlong(cv[c]) <= tbits;
        const unsigned long long mask = __ballot(in);
        const int pos = ns + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (in && pos < KS) {
            sv[pos] = cv[c];
            si[pos] = eidx[c];
        }
        ns += (int)__popcll(mask);
    }
    kz_wave_sync();
    if (ns > KS) {   // (more exact ties at the k-th place than the buffers hold: the exact kernels order those)
        fail();
        return;
    }
    const double* out_v = sv;
    const int* out_i = si;
    if (ns <= 256) {
        // rank sort of the ns (~k + 1) entries in place through registers: lane c holds entries c, c + 64, ...
        double v[4];
        int id[4], rk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = lane + 64 * u;
            v[u] = c < ns ? sv[c] : 0.0;
            id[u] = c < ns ? si[c] : 0;
            rk[u] = 0;
        }
        for (int o = 0; o < ns; ++o) {
            const double ov = sv[o];
            const int oid = si[o];
#pragma unroll
            for (int u = 0; u < 4; ++u) rk[u] += (ov < v[u] || (ov == v[u] && oid < id[u])) ? 1 : 0;
        }
        kz_wave_sync();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (lane + 64 * u < ns) {
                sv[rk[u]] = v[u];
                si[rk[u]] = id[u];
            }
        }
    } else {
        // long k (more than 255 neighbours): out of place into cv / eidx, which are spent (ci shares its bytes with sv)
        for (int c = lane; c < ns; c += 64) {
            const double v = sv[c];
            const int id = si[c];
            int rk = 0;
            for (int o = 0; o < ns; ++o) {
                const double ov = sv[o];
                const int oid = si[o];
                rk += (ov < v || (ov == v && oid < id)) ? 1 : 0;
            }
            cv[rk] = v;
            eidx[rk] = id;
        }
        out_v = cv;
        out_i = eidx;
    }
    kz_wave_sync();
    // ---- certification (kz_finalize_query): rows outside the re-ranked set are the selected candidates left behind (<= left_max),
    // the entries the selection left behind (<= sel_bound), rows evicted from a full list or never above the floor (<= piece_bound)
    const float bound = fmaxf(fmaxf(piece_bound, sel_bound), left_max);
    bool certified;
    if (bound == -INFINITY)
        certified = (Vr >= min((int64_t)k_eff, p.n_i));
    else
        certified = Vr >= k_eff && (double)bound * key_scale + eps_q < exact_key(out_v[k_eff - 1]);
    if (bound_violated) certified = false;
    if (!certified) {
        fail(Vr >= k_eff ? out_v[k_eff - 1] : (double)INFINITY);
        return;
    }
    kz_emit_sorted<T>(out_v, out_i, ns, p.k, p.exclude_self, p.self_ids ? p.self_ids[q] : qrow, p.metric,
                      p.out_dist + qout * (int64_t)p.k, p.out_ind + qout * (int64_t)p.k, lane);
}

template <typename T, int MINW>
__global__ __launch_bounds__(256, MINW) void kz_knn_finalize_wide_kernel(KnnFinParams p) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* wbase = fsm + (size_t)wave * kz_fin_wide_wave_bytes(p.max_m, p.KSEL);
    const int64_t q = p.q_first + (int64_t)blockIdx.x * 4 + wave;
    if (q >= p.q_last) return;
    kz_finalize_query_wide<T>(p, q, lane, wbase);
}
