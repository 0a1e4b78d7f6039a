// Tile epilogue of the 16x16x32 build of the fp16 kernel (kz_knn_hx.h): the candidate scan of kz_knn_epi3.h for the C layout of
// v_mfma_f32_16x16x32_f16.
//
// A wave still holds 32 queries x 128 index rows, now in SIXTEEN 16x16 accumulators acc[rb][qb] (rb = 16-row block 0 .. 7, qb =
// 16-query block 0 / 1).  Lane l = (c = l & 15, g = l >> 4) holds, for query 16 qb + c of the wave, the keys of rows
// 16 rb + 4 g + 0..3 -- one GROUP of four consecutive rows per accumulator.  So a lane owns TWO queries (8 groups each), and a
// query is spread over FOUR lanes (g = 0 .. 3) instead of two.  The scan per query: 8 group maxima -> 2 half maxima -> one maximum
// -> one compare (kz_knn_epi4.h's shape); events are blind group appends to the wave's pool, linked into one chain per (lane,
// query); a merge is done by the lanes g = 0, each walking the four chains of its query.
// Entry code of a list event: 16 tile + rb (the group's rows follow from the chain's lane: 4 g).  Column entries of the dual pass
// carry the code kz_dual_scatter_kernel decodes -- tile, group and lane half of the 32x32 layout -- chosen so that it names the
// same four rows: rows 16 rb + 4 g = 32 m + 8 n + 4 hh with m = rb >> 1, n = 2 (rb & 1) + (g >> 1), hh = g & 1; the lane field
// holds hh and the query's number in the wave (16 qb + c), which is what kz_flush_col3 adds to the wave's first query row.
#pragma once
#include "kz_knn_epi3.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));

struct KzCandState5 {
    KzListRef<1> list;   // list of query 16*0 + c (the lanes g = 0 own it); query 16 + c: list.k + 16
    float tau[2];
    int head[2];
};

// Merge: lane (c, g = 0) walks, for each of its two queries, the chains of the four lanes (c, 0 .. 3)
template <bool DUAL>
__device__ __forceinline__ void kz_merge_pool5(KzCandState5& st, KzWavePool& pool, const KzDualRef& du) {
    constexpr int KP = 16;
    const int lane = threadIdx.x & 63;
    if constexpr (DUAL) kz_flush_col3(pool, du);
#pragma unroll 1
    for (int qb = 0; qb < 2; ++qb) {
        const int own = qb ? st.head[1] : st.head[0];
        int chain[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) chain[gg] = __shfl(own, (lane & 15) + 16 * gg, 64);
        float tau = qb ? st.tau[1] : st.tau[0];
        if (lane < 16) {
            KzListRef<1> L = st.list;
            L.k += 16 * qb;
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
            for (int gg = 0; gg < 4; ++gg) {
                int e = gg == 0 ? chain[0] : (gg == 1 ? chain[1] : (gg == 2 ? chain[2] : chain[3]));
#pragma unroll 1
                while (e >= 0) {
                    const f32x4e kv = pool.keys[e];
                    const i32x2e mt = pool.meta[e];
                    e = mt.y;
                    const int code = mt.x;
                    const int row0 = (code >> 4) * KZ_TILE + (code & 7) * 16 + 4 * gg;
                    if (kv.x > tau) kz_list_insert3<KP, 1>(L, bs, kv.x, row0, tau);
                    if (kv.y > tau) kz_list_insert3<KP, 1>(L, bs, kv.y, row0 + 1, tau);
                    if (kv.z > tau) kz_list_insert3<KP, 1>(L, bs, kv.z, row0 + 2, tau);
                    if (kv.w > tau) kz_list_insert3<KP, 1>(L, bs, kv.w, row0 + 3, tau);
                }
            }
        }
        tau = __shfl(tau, lane & 15, 64);
        if (qb) {
            st.tau[1] = tau;
            st.head[1] = -1;
        } else {
            st.tau[0] = tau;
            st.head[0] = -1;
        }
    }
    pool.cnt = 0;
}

// the four keys of accumulator `blk` (row block rb): the lanes in `mask` append them and link the entry in front of their chain
#define KZ_EPI5_APPEND(blk, rb, head, ev, mask)                                                                       \
    do {                                                                                                              \
        if (ev) {                                                                                                     \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                       \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u)); \
            f32x4e kv;                                                                                                \
            kv.x = (blk)[0];                                                                                          \
            kv.y = (blk)[1];                                                                                          \
            kv.z = (blk)[2];                                                                                          \
            kv.w = (blk)[3];                                                                                          \
            pool.keys[pos] = kv;                                                                                      \
            { i32x2e mv_; mv_.x = tile * 16 + (rb); mv_.y = (head); pool.meta[pos] = mv_; }                           \
            (head) = pos;                                                                                             \
        }                                                                                                             \
        pool.cnt += (int)__popcll(mask);                                                                              \
    } while (0)

// ... and as a column entry of the dual pass, in the code of the 32x32 layout (see the header)
#define KZ_EPI5_APPEND_COL(blk, rb, qb, ev, mask)                                                                     \
    do {                                                                                                              \
        if (ev) {                                                                                                     \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                       \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u)); \
            f32x4e kv;                                                                                                \
            kv.x = (blk)[0];                                                                                          \
            kv.y = (blk)[1];                                                                                          \
            kv.z = (blk)[2];                                                                                          \
            kv.w = (blk)[3];                                                                                          \
            pool.keys[pos] = kv;                                                                                      \
            i32x2e mv_;                                                                                               \
            int lane_;                                                                                                \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));              \
            const int g_ = lane_ >> 4;                                                                                \
            const int gi_ = 4 * ((rb) >> 1) + 2 * ((rb) & 1) + (g_ >> 1);                                             \
            mv_.x = KZ_COL_FLAG | ((tile * 16 + gi_) << 6) | ((g_ & 1) << 5) | (16 * (qb)) | (lane_ & 15);            \
            mv_.y = -1;                                                                                               \
            pool.meta[pos] = mv_;                                                                                     \
        }                                                                                                             \
        pool.cnt += (int)__popcll(mask);                                                                              \
    } while (0)

// the scan of ONE query of the lane (query block QB) over its eight accumulators acc[0..7][QB]
template <int CAP, bool DUAL, int QB>
__device__ __forceinline__ void kz_scan_query5(const f32x4a (&acc)[8][2], KzCandState5& st, KzWavePool& pool, const int tile, const KzDualRef& du,
                                               const float cthr) {
    float gm[8];
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) gm[rb] = fmaxf(fmaxf(acc[rb][QB][0], acc[rb][QB][1]), fmaxf(acc[rb][QB][2], acc[rb][QB][3]));
    float hm[2];
    hm[0] = fmaxf(fmaxf(gm[0], gm[1]), fmaxf(gm[2], gm[3]));
    hm[1] = fmaxf(fmaxf(gm[4], gm[5]), fmaxf(gm[6], gm[7]));
    const float m = fmaxf(hm[0], hm[1]);
    float tau_a = st.tau[QB];
    const unsigned long long anym = __builtin_amdgcn_ballot_w64(m > tau_a);
    unsigned long long anyc = 0ull;
    if constexpr (DUAL) anyc = __builtin_amdgcn_ballot_w64(m >= cthr);
    if ((anym | anyc) == 0ull) return;
    int worst = 0;
    if (anym != 0ull) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(hm[hb] > tau_a));
    }
    if constexpr (DUAL) {
        if (anyc != 0ull) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(hm[hb] >= cthr));
        }
    }
    if (pool.cnt + worst <= CAP) {
        if (anym != 0ull) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                if (__builtin_amdgcn_ballot_w64(hm[hb] > tau_a) == 0ull) continue;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rb = 4 * hb + r4;
                    const bool ev = gm[rb] > tau_a;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
                    if (mask != 0ull) KZ_EPI5_APPEND(acc[rb][QB], rb, st.head[QB], ev, mask);
                }
            }
        }
        if constexpr (DUAL) {
            if (anyc != 0ull) {
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    if (__builtin_amdgcn_ballot_w64(hm[hb] >= cthr) == 0ull) continue;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rb = 4 * hb + r4;
                        const bool evc = gm[rb] >= cthr;
                        const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                        if (maskc != 0ull) KZ_EPI5_APPEND_COL(acc[rb][QB], rb, QB, evc, maskc);
                    }
                }
            }
        }
        return;
    }
    // slow path (first tiles of a sweep, bursts, a pool that is nearly full): merge when a group does not fit and resume there
    int resume = 0;
    float cthr_l = cthr;
    for (;;) {
        bool need_room = false;
        int tile_l = tile;
        asm volatile("" : "+s"(tile_l));
        const int tile = tile_l;
        if constexpr (DUAL) asm volatile("" : "+v"(cthr_l));
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
            const bool ev = gm[rb] > tau_a;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
            if (2 * rb >= resume && !need_room && mask != 0ull) {
                if (pool.cnt + (int)__popcll(mask) > CAP) {
                    need_room = true;
                    resume = 2 * rb;
                } else {
                    KZ_EPI5_APPEND(acc[rb][QB], rb, st.head[QB], ev, mask);
                }
            }
            if constexpr (DUAL) {
                const bool evc = gm[rb] >= cthr_l;
                const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                if (2 * rb + 1 >= resume && !need_room && maskc != 0ull) {
                    if (pool.cnt + (int)__popcll(maskc) > CAP) {
                        need_room = true;
                        resume = 2 * rb + 1;
                    } else {
                        KZ_EPI5_APPEND_COL(acc[rb][QB], rb, QB, evc, maskc);
                    }
                }
            }
        }
        if (!need_room) break;
        kz_merge_pool5<DUAL>(st, pool, du);
        tau_a = st.tau[QB];
    }
}

// Epilogue of one tile (128 index rows): acc[rb][qb].  sync = 4 LDS words of the workgroup (merge flags; protocol of
// kz_tile_epilogue3).  cthr0 / cthr1: the dual pass' column thresholds of the lane's two queries (+inf otherwise).
template <int CAP, bool DUAL>
__device__ __forceinline__ void kz_tile_epilogue5(const f32x4a (&acc)[8][2], KzCandState5& st, KzWavePool& pool, const int tile, const bool last,
                                                  kz_lds_i32* sync, const KzDualRef& du, const float cthr0, const float cthr1) {
    const int t = ++pool.tiles_done;
    const bool sched = (t == pool.next_merge) || last;   // block-uniform
    if (t == pool.next_merge) {
        const int step = t / 4;   // (tiles * 4 / K', K' = 16)
        pool.next_merge = t + (step > 0 ? step : 1);
    }
    {
        const bool together = __builtin_amdgcn_readfirstlane(sync[(t - 1) & 3]) != 0;
        if ((threadIdx.x & 63) == 0) {
            int zero = 0;
            asm volatile("" : "+v"(zero));
            sync[(t + 1) & 3] = zero;
        }
        if (together) kz_merge_pool5<DUAL>(st, pool, du);
    }
    kz_scan_query5<CAP, DUAL, 0>(acc, st, pool, tile, du, cthr0);
    kz_scan_query5<CAP, DUAL, 1>(acc, st, pool, tile, du, cthr1);
    if (pool.cnt > CAP / 2 && (threadIdx.x & 63) == 0) {
        int one = 1;
        asm volatile("" : "+v"(one));
        sync[t & 3] = one;
    }
    if (sched) kz_merge_pool5<DUAL>(st, pool, du);
}
