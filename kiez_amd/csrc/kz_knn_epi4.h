// Tile epilogue of the 64-queries-per-wave fp16 kernel (kz_knn_h64.h): the candidate scan of kz_knn_epi3.h for a lane that owns
// TWO queries.
//
// A wave of that kernel holds 64 queries x 64 index rows (half a tile) in four 32x32 accumulators acc[rb][qh]: rb = 32-row
// block of the half tile, qh = query half -- lane (j, h) holds, for query 32 qh + j of the wave, the keys of rows
// 32 rb + 8 g4 + 4 h + 0..3 (the C layout of the MFMA).  Per query the scan is the one of kz_knn_epi3.h over TWO row blocks
// instead of four (8 group maxima -> 2 block maxima -> one maximum -> one compare); the two queries of a lane share the wave's
// event pool (entries are linked into one chain per (lane, query)), the bookkeeping of a tile (merge schedule, workgroup merge
// flags, capacity check) and every merge.  Entry codes, row arithmetic, list inserts and the dual pass' column entries are those
// of kz_knn_epi3.h: code = 16 tile + 4 mt + g4 with mt = 2 hf + rb, the 32-row block inside the 128-row tile.
// Lists: K' = 16, in LDS, [entry][128 queries of a tile] (KzListRef<1>); the list of a lane's second query sits 32 floats
// behind the first one's.  Block minima are re-read at every merge (RECOMP of kz_knn_epi3.h: no registers between merges).
#pragma once
#include "kz_knn_epi3.h"

struct KzCandState4 {
    KzListRef<1> list;   // list of query half 0 (lanes < 32 own it); query half 1: list.k + 32
    float tau[2];        // K'-th best key of each query's list as of the last merge (same value in both lane halves)
    int head[2];         // newest entry of this lane's chain per query, -1 = empty
};

struct KzDualRef4 {
    int qrow0;   // global row of the wave's query 0 (its query half 1 starts 32 rows further)
};
constexpr int KZ_COL_QH = 0x40000000;   // column entry of the lane's SECOND query (bit 30; the code proper stays below 2^30: < 2^20 index tiles)

// column entries -> global log (kz_flush_col3 with the query half of an entry taken from bit 30 of its code)
__device__ __forceinline__ void kz_flush_col4(const KzWavePool& pool, const KzDualRef4& du) {
    const int lane = threadIdx.x & 63;
    int n_col = 0;
    for (int e0 = 0; e0 < pool.cnt; e0 += 64) {   // (uniform trip count)
        const int e = e0 + lane;
        const bool col = e < pool.cnt && pool.meta[e].x < 0;
        n_col += (int)__popcll(__builtin_amdgcn_ballot_w64(col));
    }
    if (n_col == 0) return;
    typedef __attribute__((address_space(4))) const volatile unsigned long long kz_karg_u64;
    const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    f32x4e* log_keys = (f32x4e*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_keys));
    i32x2e* log_meta = (i32x2e*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_meta));
    unsigned long long* log_cnt = (unsigned long long*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_cnt));
    const long long log_cap = (long long)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_cap));
    unsigned long long base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(log_cnt, (unsigned long long)n_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (int e0 = 0; e0 < pool.cnt; e0 += 64) {
        const int e = e0 + lane;
        i32x2e mt;
        mt.x = 0;
        if (e < pool.cnt) mt = pool.meta[e];
        const bool col = mt.x < 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(col);
        if (col) {
            const unsigned long long pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if ((long long)pos < log_cap) {
                i32x2e mo;
                mo.x = mt.x & 0x3fffffff;
                mo.y = du.qrow0 + ((mt.x & KZ_COL_QH) ? 32 : 0) + (mt.x & 31);
                __builtin_nontemporal_store(pool.keys[e], log_keys + pos);
                __builtin_nontemporal_store(mo, log_meta + pos);
            }
        }
        base += __popcll(mask);
    }
}

// Merge: for each of the lane's two queries, lane l < 32 walks its own chain, then its partner's (lane l + 32: the other half of the
// same query), inserting every key that still beats the list's threshold (kz_merge_pool3, twice, on one pool).
template <bool DUAL>
__device__ __forceinline__ void kz_merge_pool4(KzCandState4& st, KzWavePool& pool, const KzDualRef4& du) {
    constexpr int KP = 16;
    const int lane = threadIdx.x & 63;
    if constexpr (DUAL) kz_flush_col4(pool, du);
#pragma unroll 1
    for (int qh = 0; qh < 2; ++qh) {
        const int own = qh ? st.head[1] : st.head[0];
        const int other = __shfl_xor(own, 32, 64);
        float tau = qh ? st.tau[1] : st.tau[0];
        if (lane < 32) {
            KzListRef<1> L = st.list;
            L.k += 32 * qh;
            KzBlockMin3<KP> bs;
            constexpr int NB = KzBlockMin3<KP>::NB, BS = KzBlockMin3<KP>::BS, S = KzListRef<1>::KSTRIDE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float m = INFINITY;
#pragma unroll
                for (int jj = 0; jj < BS; ++jj) m = fminf(m, L.kp()[(b * BS + jj) * S]);
                bs.bm[b] = m;
            }
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                int e = half ? other : own;
#pragma unroll 1
                while (e >= 0) {
                    auto* entry = pool.keys + e;
                    const auto* meta = pool.meta + e;
                    e = meta->y;
                    kz_insert_group3<KP, 1>(L, bs, entry, meta, half, tau);
                }
            }
        }
        tau = __shfl(tau, lane & 31, 64);
        if (qh) {
            st.tau[1] = tau;
            st.head[1] = -1;
        } else {
            st.tau[0] = tau;
            st.head[0] = -1;
        }
    }
    pool.cnt = 0;
}

// One group of four keys of accumulator block `blk` (group g4 of it; code gi within the 128-row tile): the lanes in `mask`
// append it to the pool and link it in front of their chain `head`
#define KZ_EPI4_APPEND(blk, g4, gi, head, ev, mask)                                                                   \
    do {                                                                                                              \
        if (ev) {                                                                                                     \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                       \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u)); \
            f32x4e kv;                                                                                                \
            kv.x = (blk)[4 * (g4)];                                                                                   \
            kv.y = (blk)[4 * (g4) + 1];                                                                               \
            kv.z = (blk)[4 * (g4) + 2];                                                                               \
            kv.w = (blk)[4 * (g4) + 3];                                                                               \
            pool.keys[pos] = kv;                                                                                      \
            { i32x2e mv_; mv_.x = tile * 16 + (gi); mv_.y = (head); pool.meta[pos] = mv_; }                           \
            (head) = pos;                                                                                             \
        }                                                                                                             \
        pool.cnt += (int)__popcll(mask);                                                                              \
    } while (0)

// ... and as a column entry of the dual pass (code < 0, not linked; qflag = KZ_COL_QH for the lane's second query)
#define KZ_EPI4_APPEND_COL(blk, g4, gi, qflag, ev, mask)                                                              \
    do {                                                                                                              \
        if (ev) {                                                                                                     \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                       \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u)); \
            f32x4e kv;                                                                                                \
            kv.x = (blk)[4 * (g4)];                                                                                   \
            kv.y = (blk)[4 * (g4) + 1];                                                                               \
            kv.z = (blk)[4 * (g4) + 2];                                                                               \
            kv.w = (blk)[4 * (g4) + 3];                                                                               \
            pool.keys[pos] = kv;                                                                                      \
            i32x2e mv_;                                                                                               \
            int lane_;                                                                                                \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));              \
            mv_.x = KZ_COL_FLAG | (qflag) | ((tile * 16 + (gi)) << 6) | lane_;                                        \
            mv_.y = -1;                                                                                               \
            pool.meta[pos] = mv_;                                                                                     \
        }                                                                                                             \
        pool.cnt += (int)__popcll(mask);                                                                              \
    } while (0)

// The scan of ONE query of the lane over its two accumulator blocks b0 (rows 0-31 of the half tile) and b1 (rows 32-63);
// mt0 = 2 hf: the first block's number inside the 128-row tile.  Appends list events (key group maximum > tau) to the chain
// `head` and, in the dual pass, column events (group maximum >= cthr).  Merges (both queries of the lane, the whole pool) when a
// group does not fit, and resumes.
template <int CAP, bool DUAL, int QH>
__device__ __forceinline__ void kz_scan_query4(const f32x16& b0, const f32x16& b1, KzCandState4& st, KzWavePool& pool, const int tile, const int mt0,
                                               const KzDualRef4& du, const float cthr) {
    float gm[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const f32x16& b = g < 4 ? b0 : b1;
        const int g4 = g & 3;
        gm[g] = fmaxf(fmaxf(b[4 * g4], b[4 * g4 + 1]), fmaxf(b[4 * g4 + 2], b[4 * g4 + 3]));
    }
    float bmx[2];
    bmx[0] = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
    bmx[1] = fmaxf(fmaxf(gm[4], gm[5]), fmaxf(gm[6], gm[7]));
    const float m = fmaxf(bmx[0], bmx[1]);
    float tau_a = st.tau[QH];
    const unsigned long long anym = __builtin_amdgcn_ballot_w64(m > tau_a);
    unsigned long long anyc = 0ull;
    if constexpr (DUAL) anyc = __builtin_amdgcn_ballot_w64(m >= cthr);
    if ((anym | anyc) == 0ull) return;
    int worst = 0;
    if (anym != 0ull) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(bmx[rb] > tau_a));
    }
    if constexpr (DUAL) {
        if (anyc != 0ull) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(bmx[rb] >= cthr));
        }
    }
    if (pool.cnt + worst <= CAP) {
        // common path: whatever the groups hold, the pool takes it
        if (anym != 0ull) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                if (__builtin_amdgcn_ballot_w64(bmx[rb] > tau_a) == 0ull) continue;
                const f32x16& b = rb ? b1 : b0;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const bool ev = gm[4 * rb + g4] > tau_a;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
                    if (mask != 0ull) KZ_EPI4_APPEND(b, g4, 4 * (mt0 + rb) + g4, st.head[QH], ev, mask);
                }
            }
        }
        if constexpr (DUAL) {
            if (anyc != 0ull) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    if (__builtin_amdgcn_ballot_w64(bmx[rb] >= cthr) == 0ull) continue;
                    const f32x16& b = rb ? b1 : b0;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const bool evc = gm[4 * rb + g4] >= cthr;
                        const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                        if (maskc != 0ull) KZ_EPI4_APPEND_COL(b, g4, 4 * (mt0 + rb) + g4, QH ? KZ_COL_QH : 0, evc, maskc);
                    }
                }
            }
        }
        return;
    }
    // slow path (first tiles of a sweep, bursts, a pool that is nearly full): merge when a group does not fit and resume at that
    // step with the fresher threshold (step = 2 group + kind: list entry, then column entry)
    int resume = 0;
    float cthr_l = cthr;
    for (;;) {
        bool need_room = false;
        int tile_l = tile;
        asm volatile("" : "+s"(tile_l));   // (entry codes made on the spot: hoisted, the 16 of them sit in scalar registers for the whole epilogue)
        const int tile = tile_l;
        if constexpr (DUAL) asm volatile("" : "+v"(cthr_l));
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x16& b = g < 4 ? b0 : b1;
            const int g4 = g & 3, gi = 4 * mt0 + g;
            const bool ev = gm[g] > tau_a;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
            if (2 * g >= resume && !need_room && mask != 0ull) {
                if (pool.cnt + (int)__popcll(mask) > CAP) {
                    need_room = true;
                    resume = 2 * g;
                } else {
                    KZ_EPI4_APPEND(b, g4, gi, st.head[QH], ev, mask);
                }
            }
            if constexpr (DUAL) {
                const bool evc = gm[g] >= cthr_l;
                const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                if (2 * g + 1 >= resume && !need_room && maskc != 0ull) {
                    if (pool.cnt + (int)__popcll(maskc) > CAP) {
                        need_room = true;
                        resume = 2 * g + 1;
                    } else {
                        KZ_EPI4_APPEND_COL(b, g4, gi, QH ? KZ_COL_QH : 0, evc, maskc);
                    }
                }
            }
        }
        if (!need_room) break;
        kz_merge_pool4<DUAL>(st, pool, du);
        tau_a = st.tau[QH];
    }
}

// Epilogue of one HALF tile (64 index rows): acc[rb][qh].  hf = which half of the 128-row tile.  sync = 4 LDS words of the
// workgroup (merge flags, protocol of kz_tile_epilogue3: written during epilogue t, read during t + 1, cleared during t + 2, a
// workgroup barrier between any two epilogues -- every half tile contains a slice barrier).  cthr0 / cthr1: the dual pass'
// column thresholds of the lane's two queries (+inf otherwise).
template <int CAP, bool DUAL>
__device__ __forceinline__ void kz_half_epilogue4(f32x16 (&acc)[2][2], KzCandState4& st, KzWavePool& pool, const int tile, const int hf,
                                                  const bool last, kz_lds_i32* sync, const KzDualRef4& du, const float cthr0, const float cthr1) {
    const int t = ++pool.tiles_done;
    const bool sched = (t == pool.next_merge) || last;   // block-uniform
    if (t == pool.next_merge) {
        // geometric schedule (kz_tile_epilogue3: the next merge after tiles * 4 / K' more tiles; K' = 16, t counts half tiles)
        const int step = t / 4;
        pool.next_merge = t + (step > 0 ? step : 1);
    }
    {
        const bool together = __builtin_amdgcn_readfirstlane(sync[(t - 1) & 3]) != 0;
        if ((threadIdx.x & 63) == 0) {
            int zero = 0;
            asm volatile("" : "+v"(zero));
            sync[(t + 1) & 3] = zero;
        }
        if (together) kz_merge_pool4<DUAL>(st, pool, du);
    }
    kz_scan_query4<CAP, DUAL, 0>(acc[0][0], acc[1][0], st, pool, tile, 2 * hf, du, cthr0);
    kz_scan_query4<CAP, DUAL, 1>(acc[0][1], acc[1][1], st, pool, tile, 2 * hf, du, cthr1);
    if (pool.cnt > CAP / 2 && (threadIdx.x & 63) == 0) {
        int one = 1;
        asm volatile("" : "+v"(one));
        sync[t & 3] = one;
    }
    if (sched) kz_merge_pool4<DUAL>(st, pool, du);
}
