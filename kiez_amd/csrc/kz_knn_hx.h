// fp16 first pass on v_mfma_f32_16x16x32_f16 (K' = 16): the kernel of kz_knn_h16.h with the matrix instruction the chip can hold a
// higher clock on.
//
// Why: the sweeps of kz_knn_h16.h / kz_knn_h64.h sit on the chip's power limit -- matrix-pipe busy x clock is the same 1.05 GHz for
// both structures, and the same launch on zero operands runs 34 % shorter (profiles/r04_ablation.md section 3).  What is left is
// energy per multiply-add, and the guide's DVFS note 7 (and round 3's bare-loop probe) give the 16x16x32 shape 12 - 15 % more
// FLOP/s than 32x32x16 at equal cycles: the chip holds a higher clock on it.
//
// Same workgroup geometry as kz_knn_h16.h: 4 waves x 32 queries, tile = 128 index rows, the SAME fp16 image and 4-KiB slices, the
// same LDS-DMA ring (4 slots) -- but the k loop advances in STEPS of 32 k = two slices:
//   A (index rows): lane l reads row 16 rb + (l & 15), k chunk l >> 4 of the step = plane (l >> 4) & 1 of slice 2 s + (l >> 5):
//     one ds_read_b128 per 16-row block rb, conflict-free on the existing image; lanes 32 .. 63 read the step's second slice.
//   B (queries, stationary): 2 query blocks x 4 VGPRs per step.  An odd slice count leaves the last step half empty: its second
//     slice does not exist -- the B registers of the lanes that would hold it are ZERO, so whatever (finite) bytes those lanes
//     read for A from the ring's next slot contribute exactly 0.
//   C: sixteen 16x16 accumulators acc[rb][qb]; lane (c, g) holds rows 16 rb + 4 g + 0..3 of query 16 qb + c (kz_knn_epi5.h).
// A step = two half steps of 4 fragment reads + 8 MFMAs (4 row blocks x 2 query blocks): the cadence of a slice of kz_knn_h16.h.
// The bias rows are read once per tile and enter the second query block through the C operand of its first MFMA.
// Ring protocol, in slices: slice G lives in slot G & 3; a workgroup barrier at every step boundary; behind it the copies run
// ahead to slice (first unread slice) + 3.  A slice is waited for (vmcnt(0) of its issuing wave + the barrier) one step after it
// was issued and read the step after that.
// Lists in LDS, results, entry codes of the dual pass' log: those of kz_knn_h16.h (tests/test_gpu_hx.py forces this build).
#pragma once
#include <type_traits>

#include "kz_knn_epi5.h"
#include "kz_knn_h16.h"   // kz_f16x8

template <int NSR, int WPS, bool DUAL>
struct KzHxCfg {
    static constexpr int KP = 16;
    static constexpr int RING = 4;
    static constexpr int NST = (NSR + 1) / 2;                          // steps of 32 k per tile
    static constexpr int CAP = WPS == 3 ? (DUAL ? 176 : 192) : 256;    // event-pool entries per wave (24 B each)
    static constexpr int RING_BYTES = RING * 4096;
    static constexpr int BIAS_OFF = RING_BYTES;                        // 2 x 128 floats
    static constexpr int SYNC_OFF = BIAS_OFF + 1024;
    static constexpr int THETA_OFF = SYNC_OFF + 256;                   // dual pass: 3 x 64 threshold floats + 128 query offsets
    static constexpr int POOLK_OFF = THETA_OFF + (DUAL ? 768 + 768 : 0);
    static constexpr int POOLM_OFF = POOLK_OFF + 4 * CAP * 16;
    static constexpr int LIST_OFF = POOLM_OFF + 4 * CAP * 8;           // keys [16][128], then rows [16][128]
    static constexpr int LDS_BYTES = LIST_OFF + KP * 128 * 8;
    static_assert(WPS == 2 || LDS_BYTES <= 42 * 1280, "three workgroups per CU: 42 LDS granules of 1280 B each");
};

template <int NSR, int WPS, bool DUAL>
__global__ __launch_bounds__(256, WPS) void kz_knn_cand_hx_kernel(KnnCandParams p) {
    using Cfg = KzHxCfg<NSR, WPS, DUAL>;
    constexpr int R = Cfg::RING, CAP = Cfg::CAP, KP = Cfg::KP, NST = Cfg::NST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);
    float* bbuf = reinterpret_cast<float*>(smem + Cfg::BIAS_OFF);
    float* tbuf = reinterpret_cast<float*>(smem + Cfg::THETA_OFF);
    kz_lds_i32* msync = (kz_lds_i32*)(smem + Cfg::SYNC_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int gq = lane >> 4;   // k chunk of the operands, row quad of the results
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;

    auto out_list_offset = [&](const int qb) { return kz_list_contig_off((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + 16 * qb + c, p.lay, KP, s); };
    KzCandState5 st;
    st.list.k = (kz_lds_f32*)(smem + Cfg::LIST_OFF) + 32 * (tid >> 6) + c;
    st.list.i_off = KP * 128;
    KzWavePool pool;
    pool.keys = (__attribute__((address_space(3))) f32x4e*)(smem + Cfg::POOLK_OFF) + wave * CAP;
    pool.meta = (__attribute__((address_space(3))) i32x2e*)(smem + Cfg::POOLM_OFF) + wave * CAP;
    if (gq == 0) {
#pragma unroll 4
        for (int e = 0; e < KP; ++e) {
            st.list.kp()[e * 128] = -INFINITY;
            st.list.kp()[e * 128 + 16] = -INFINITY;
            st.list.ip()[e * 128] = -1;
            st.list.ip()[e * 128 + 16] = -1;
        }
    }
    if (t_end <= t_begin) {
        if (gq == 0) {
#pragma unroll 1
            for (int qb = 0; qb < 2; ++qb) {
                const int64_t listoff = out_list_offset(qb);
                for (int e = 0; e < KP; ++e) {
                    p.out_key[listoff + e] = -INFINITY;
                    p.out_idx[listoff + e] = -1;
                }
            }
        }
        return;
    }
    st.tau[0] = st.tau[1] = -INFINITY;
    st.head[0] = st.head[1] = -1;
    pool.cnt = 0;
    pool.tiles_done = 0;
    pool.next_merge = 1;

    // LDS-DMA of 4-KiB slices, 1 KiB per wave, strictly in order (kz_knn_h16.h); `issued` = slices issued so far
    const char* dma_src = reinterpret_cast<const char*>(p.ypack) + ((int64_t)t_begin * NSR) * 4096;   // uniform
    int issued = 0;
    auto dma_next = [&]() {
        unsigned lane16;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
        kz_glds16_s(dma_src + wave * 1024, lane16, ybuf + (issued & (R - 1)) * 1024 + wave * 256);
        dma_src += 4096;
        ++issued;
    };
#pragma unroll
    for (int i = 0; i < R; ++i) dma_next();
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    KzDualRef du;
    du.qrow0 = (p.qt0 + qt) * KZ_TILE + 32 * wave;
    if constexpr (DUAL) {
        if (tid < 64) tbuf[tid] = p.theta[(int64_t)t_begin * KZ_TILE + tid];
        if (lane < 32) tbuf[192 + 32 * (tid >> 6) + lane] = p.qnbias[du.qrow0 + lane];
    }
    if (tid < 4) msync[tid] = 0;
    // stationary query operands: lane (c, gq) holds, per step and query block, k chunk gq of the step = plane gq & 1 of slice
    // 2 st + (gq >> 1), query row 32 wave + 16 qb + c; the chunks of a slice that does not exist (odd NSR, last step) are zero
    kz_f16x8 qf[NST][2];
    {
        const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 1024 + ((gq & 1) * KZ_TILE + 32 * (tid >> 6) + c) * 4;
#pragma unroll
        for (int sx = 0; sx < NST; ++sx) {
            const bool whole = 2 * sx + 1 < NSR;   // (compile time)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                kz_f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (whole || gq < 2) v = *reinterpret_cast<const kz_f16x8*>(qbase + (2 * sx + (whole ? (gq >> 1) : 0)) * 1024 + 16 * qb * 4);
                qf[sx][qb] = v;
            }
        }
    }
    __syncthreads();   // (drains vmcnt(0): the whole prologue ring has landed)

    // this lane's fragment offset inside a slot: plane gq & 1, row c (+ 16 rb); lanes gq >= 2 read the step's second slice
    const int frag_off = ((gq & 1) * KZ_TILE + c) * 4;   // floats
    int g = 0;        // slices consumed (uniform)
    int th_cur = 0;   // dual pass: threshold buffer of the current tile (uniform)
    f32x4a acc[8][2];

    auto run_tile = [&](const int tile) {
        // accumulator init: bias of rows 16 rb + 4 gq + 0..3, for query block 0; query block 1 takes it through the C operand
        {
            int gq_now;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 4, %0" : "=v"(gq_now));
            const float* bp = bbuf + (tile & 1) * 128 + 4 * gq_now;
#pragma unroll
            for (int rb = 0; rb < 8; ++rb) {
                const float4 v = *reinterpret_cast<const float4*>(bp + 16 * rb);
                acc[rb][0][0] = v.x;
                acc[rb][0][1] = v.y;
                acc[rb][0][2] = v.z;
                acc[rb][0][3] = v.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            // bias rows of the next tile (waves 0, 1) and its smallest thresholds (dual pass, wave 2) by 4-byte LDS-DMA: visible
            // behind the step barriers every tile contains
            unsigned lane4;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(lane4));
            const int nt = min(tile + 1, p.n_ytiles - 1);
            if (wave < 2)
                kz_glds4_s(p.ybias + (int64_t)nt * KZ_TILE + wave * 64, lane4, bbuf + ((tile + 1) & 1) * 128 + wave * 64);
            else if (DUAL && wave == 2)
                kz_glds4_s(p.theta + (int64_t)nt * KZ_TILE, lane4, tbuf + (th_cur == 2 ? 0 : th_cur + 1) * 64);
        }
#pragma unroll
        for (int sx = 0; sx < NST; ++sx) {
            const int nsl = (2 * sx + 1 < NSR) ? 2 : 1;   // slices this step consumes (compile time)
            // per-lane slot base: lanes gq < 2 read slice g, the others slice g + 1 (the ring's next slot even where that slice
            // belongs to the next tile or is not this step's: its B operand is zero then)
            const int slotA = (g & (R - 1)) * 1024, slotB = ((g + 1) & (R - 1)) * 1024;
            int gq_now;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 5, %0" : "=v"(gq_now));
            const float* fb = ybuf + frag_off + (gq_now ? slotB : slotA);
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                kz_f16x8 f[4];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) f[r4] = *reinterpret_cast<const kz_f16x8*>(fb + 64 * (4 * hs + r4));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rb = 4 * hs + r4;
                    if (sx == 0) {
                        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[r4], qf[0][1], acc[rb][0], 0, 0, 0);
                        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[r4], qf[0][0], acc[rb][0], 0, 0, 0);
                    } else {
                        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[r4], qf[sx][0], acc[rb][0], 0, 0, 0);
                        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[r4], qf[sx][1], acc[rb][1], 0, 0, 0);
                    }
                }
            }
            g += nsl;
            // step barrier: every wave has the step's fragments in registers; the slots of the slices < g take the copies up to
            // slice g + R - 1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            while (issued < g + R) dma_next();   // (uniform: one or two copies)
        }
        __builtin_amdgcn_sched_barrier(0);
        float cthr0 = INFINITY, cthr1 = INFINITY;
        if constexpr (DUAL) {
            const float th = tbuf[th_cur * 64];
            cthr0 = tbuf[192 + 32 * (tid >> 6) + c] + th;
            cthr1 = tbuf[192 + 32 * (tid >> 6) + 16 + c] + th;
            th_cur = th_cur == 2 ? 0 : th_cur + 1;
        }
        kz_tile_epilogue5<CAP, DUAL>(acc, st, pool, tile, tile == t_end - 1, msync, du, cthr0, cthr1);
    };

    for (int tile = t_begin; tile < t_end; ++tile) run_tile(tile);

    int lane_now;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
    if (lane_now < 16) {
#pragma unroll 1
        for (int qb = 0; qb < 2; ++qb) {
            const int64_t listoff = out_list_offset(qb);
#pragma unroll 4
            for (int e = 0; e < KP; ++e) {
                p.out_key[listoff + e] = st.list.kp()[e * 128 + 16 * qb];
                p.out_idx[listoff + e] = st.list.ip()[e * 128 + 16 * qb];
            }
        }
    }
}
