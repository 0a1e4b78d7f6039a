// fp16 first pass, 64 QUERIES PER WAVE (K' = 16): the kernel of kz_knn_h16.h with half the LDS traffic per multiply-add.
//
// kz_knn_h16.h gives a wave 32 queries x 128 index rows: per 16-k slice it reads four 1 KiB fragments of the index from LDS for
// four MFMAs, and a workgroup's 4 KiB LDS-DMA copy of the slice serves 128 queries; round 3's ablation had the DMA ring cost
// 14-17 % in proportion to its volume and a bare-loop probe of 64-query waves 7-14 % faster per flop.  What this kernel measured
// once it was a whole kernel (profiles/r04_ablation.md sections 2, 3): a tie -- -1.5 % on the shared sweep at 13 slices, +1 ... +9 %
// on ordinary sweeps -- because the sweeps sit on the chip's power limit, not on the LDS pipeline: matrix-pipe busy x clock is the
// same 1.05 GHz for both structures (this one: an emptier pipe at a higher clock).  It runs where it wins (kz_knn_impl: the shared
// sweep from 9 slices on, large launches) and stays otherwise as that evidence.
//
// Here a wave owns 64 queries x 64 index rows (HALF a tile): four accumulators acc[rb][qh] (rb = 32-row block, qh = 32-query
// half), and per 16-k step of a half tile TWO 1 KiB fragments feed FOUR MFMAs (each index fragment meets both query halves):
// half the fragment bytes per MFMA.  A workgroup = 4 waves = 256 queries (two query tiles: a "unit" of the work table, tpw = 2)
// on ONE ring: a 2 KiB copy (64 rows x 16 k) serves 256 queries -- half the DMA volume per query.  The accumulator-init
// (bias) rows are read once per half tile and enter the second query half through the C operand of its first MFMA.
// Both query tiles stay stationary in registers (8 VGPRs per slice; 13 slices = 104), the accumulators take 64, the two
// fragment sets 16: 2 waves per SIMD (256 VGPRs), 2 workgroups per CU (79 KiB of LDS each).
//
// Index image: unchanged (kz_pack.hip: [tile][slice][plane][128 rows][8] fp16, 4 KiB per slice).  A half slice (tile t, half hf,
// slice u) is two 1 KiB runs of it -- rows 64 hf .. 64 hf + 63 of plane 0 and of plane 1 -- copied by the two waves of a pair
// (waves 0/1 take the even half slices of the sweep, waves 2/3 the odd ones: one copy instruction per wave and TWO half
// slices); inside a ring slot they form the conflict-free ds_read_b128 image again ([plane][64 rows][16 B]).
// Order of a sweep: tile t: half 0 slices 0 .. NSR-1, half 1 slices 0 .. NSR-1, tile t + 1 ...  Ring of 8 slots (16 KiB), one
// workgroup barrier per 4 half slices, fragments of half slice g + 1 fetched under the MFMAs of half slice g (two static sets) --
// the protocol of kz_knn_h16.h at two workgroups per CU.  Requires 4 <= NSR (every half tile contains a barrier) and NSR <= 13.
// Candidate selection: kz_knn_epi4.h (the scan of kz_knn_epi3.h for a lane that owns two queries).  Lists in LDS, flushed to the
// output arrays in the layout kz_knn_finalize_kernel reads; entry codes, dual-pass log and results are those of kz_knn_h16.h.
#pragma once
#include <type_traits>

#include "kz_knn_epi4.h"
#include "kz_knn_h16.h"   // kz_f16x8

template <int NSR, bool DUAL>
struct KzH64Cfg {
    static constexpr int KP = 16;
    static constexpr int RING = 8, PERIOD = 4;
    static constexpr int SLOT_FLOATS = 512;                            // 2 KiB: [2 planes][64 rows][16 B]
    static constexpr int CAP = 288;                                    // event-pool entries per wave (64 queries; 24 B each)
    static constexpr int RING_BYTES = RING * SLOT_FLOATS * 4;
    static constexpr int BIAS_OFF = RING_BYTES;                        // 2 x 64 floats
    static constexpr int SYNC_OFF = BIAS_OFF + 512;                    // 4 merge flags (+ padding)
    static constexpr int THETA_OFF = SYNC_OFF + 256;                   // dual pass: 3 x 64 threshold floats, then -bias of the 256 queries
    static constexpr int POOLK_OFF = THETA_OFF + (DUAL ? 768 + 1024 : 0);
    static constexpr int POOLM_OFF = POOLK_OFF + 4 * CAP * 16;
    static constexpr int LIST_OFF = POOLM_OFF + 4 * CAP * 8;           // per query tile: keys [16][128], then rows [16][128]
    static constexpr int LIST_BLOCK = KP * 128 * 8;
    static constexpr int LDS_BYTES = LIST_OFF + 2 * LIST_BLOCK;
    static_assert(LDS_BYTES <= 64 * 1280, "two workgroups per CU: 64 LDS granules of 1280 B each");
    static_assert(NSR >= PERIOD && NSR <= 13, "every half tile must contain a slice barrier; 13 stationary slices = 104 VGPRs");
};

template <int NSR, bool DUAL>
__global__ __launch_bounds__(256, 2) void kz_knn_cand_h64_kernel(KnnCandParams p) {
    using Cfg = KzH64Cfg<NSR, DUAL>;
    constexpr int R = Cfg::RING, P = Cfg::PERIOD, CAP = Cfg::CAP, KP = Cfg::KP;
    static_assert(R == 2 * P && (R & (R - 1)) == 0, "a power-of-two ring of two periods");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);
    float* bbuf = reinterpret_cast<float*>(smem + Cfg::BIAS_OFF);
    float* tbuf = reinterpret_cast<float*>(smem + Cfg::THETA_OFF);
    kz_lds_i32* msync = (kz_lds_i32*)(smem + Cfg::SYNC_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int t_begin = wd.y, t_end = wd.z, s = wd.w;
    // this wave's 64 queries: rows r0 .. r0 + 63 of query tile wd.x + (wave >> 1).  A unit at the end of the launch may reach past
    // the last tile: those waves sweep along (copies and barriers need them) on the last valid tile's rows with every threshold
    // at +inf -- no event, no list traffic, no output
    const int tb = wave >> 1;
    const int r0 = 64 * (wave & 1);
    const int r0_v = 64 * ((tid >> 6) & 1);
    const bool valid = wd.x + tb < p.n_qtiles;
    const int qt = valid ? wd.x + tb : p.n_qtiles - 1;

    auto out_list_offset = [&](const int qh) { return kz_list_contig_off((int64_t)qt * KZ_TILE + r0_v + 32 * qh + j, p.lay, KP, s); };
    KzCandState4 st;
    st.list.k = (kz_lds_f32*)(smem + Cfg::LIST_OFF + tb * Cfg::LIST_BLOCK) + r0_v + j;
    st.list.i_off = KP * 128;
    KzWavePool pool;
    pool.keys = (__attribute__((address_space(3))) f32x4e*)(smem + Cfg::POOLK_OFF) + wave * CAP;
    pool.meta = (__attribute__((address_space(3))) i32x2e*)(smem + Cfg::POOLM_OFF) + wave * CAP;
    // (seeded lists: KnnCandParams::qfloor)
    const int64_t frow = (int64_t)(p.qt0 + qt) * KZ_TILE + r0_v + j;
    const float fl0 = p.qfloor ? p.qfloor[frow] : -INFINITY, fl1 = p.qfloor ? p.qfloor[frow + 32] : -INFINITY;
    if (h == 0 && valid) {
#pragma unroll 4
        for (int e = 0; e < KP; ++e) {
            st.list.kp()[e * 128] = fl0;
            st.list.kp()[e * 128 + 32] = fl1;
            st.list.ip()[e * 128] = -1;
            st.list.ip()[e * 128 + 32] = -1;
        }
    }
    if (t_end <= t_begin) {
        if (h == 0 && valid) {
#pragma unroll 1
            for (int qh = 0; qh < 2; ++qh) {
                const int64_t listoff = out_list_offset(qh);
                for (int e = 0; e < KP; ++e) {
                    p.out_key[listoff + e] = -INFINITY;
                    p.out_idx[listoff + e] = -1;
                }
            }
        }
        return;
    }
    st.tau[0] = valid ? fl0 : INFINITY;
    st.tau[1] = valid ? fl1 : INFINITY;
    st.head[0] = st.head[1] = -1;
    pool.cnt = 0;
    pool.tiles_done = 0;
    pool.next_merge = 1;

    // LDS-DMA of half slices: the waves of pair (wave >> 1) copy the half slices G of the sweep with G % 2 == pair, wave (wave & 1)
    // of the pair the rows' plane (wave & 1) -- 64 rows x 16 B = 1 KiB per instruction.  (d_tile, d_hf, d_u) = the next half slice
    // of this wave, d_slot its ring slot; the sequence runs past the end of the sweep into the next tiles of the image or the
    // padding behind it (kz_himage_build), those slots are never read.
    int d_tile = t_begin, d_hf = 0, d_u = tb, d_slot = tb;   // (NSR >= 4: half slices 0 and 1 are slices 0 and 1 of half 0)
    const char* ybase = reinterpret_cast<const char*>(p.ypack) + (wave & 1) * 2048;
    auto dma_next = [&]() {
        const char* src = ybase + ((int64_t)d_tile * NSR + d_u) * 4096 + d_hf * 1024;   // uniform
        unsigned lane16;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 4, %0" : "=v"(lane16));
        kz_glds16_s(src, lane16, ybuf + d_slot * Cfg::SLOT_FLOATS + (wave & 1) * 256);
        d_u += 2;
        if (d_u >= NSR) {
            d_u -= NSR;
            d_hf ^= 1;
            if (d_hf == 0) ++d_tile;
        }
        d_slot = (d_slot + 2) & (R - 1);
    };
#pragma unroll
    for (int i = 0; i < R / 2; ++i) dma_next();
    if (tid < 64) bbuf[tid] = p.ybias[(int64_t)t_begin * KZ_TILE + tid];
    KzDualRef4 du;
    du.qrow0 = (p.qt0 + qt) * KZ_TILE + r0;
    if constexpr (DUAL) {
        if (tid < 64) tbuf[tid] = p.theta[(int64_t)t_begin * KZ_TILE + tid];
        // the queries' own offsets: read back from LDS in every epilogue
        if (h == 0) {
            tbuf[192 + 64 * (tid >> 6) + j] = valid ? p.qnbias[du.qrow0 + j] : INFINITY;
            tbuf[192 + 64 * (tid >> 6) + 32 + j] = valid ? p.qnbias[du.qrow0 + 32 + j] : INFINITY;
        }
    }
    if (tid < 4) msync[tid] = 0;
    // stationary query fragments: lane (j, h) holds k = 16 u + 8 h + 0..7 of query rows r0 + j (qh = 0) and r0 + 32 + j (qh = 1)
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 1024 + (h * KZ_TILE + r0_v + j) * 4;
    kz_f16x8 qf[2][NSR];
#pragma unroll
    for (int u = 0; u < NSR; ++u) {
        qf[0][u] = *reinterpret_cast<const kz_f16x8*>(qbase + u * 1024);
        qf[1][u] = *reinterpret_cast<const kz_f16x8*>(qbase + u * 1024 + 32 * 4);
    }
    // the whole prologue ring must have landed before anyone reads it.  The copies are inline asm: the compiler's barrier does not
    // know them, and only waves with a load-dependent LDS store of their own in front of it (wave 0's bias rows; every wave in the
    // dual build) would wait for them by accident -- at 4 slices (8 query-fragment loads) waves 1 .. 3 of the ordinary build did
    // reach the barrier with copies in flight: 1 - 4 wrong rows in ~1 % of randomised runs (tools/fuzz_dual.py, round 4).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float* fbase = ybuf + (h * 64 + j) * 4;   // this lane's fragment inside a slot: plane h, row j (+ 32 rb)
    kz_f16x8 f0[2], f1[2];
    auto fetch_frags = [&](kz_f16x8 (&f)[2], const int gi) {
        const float* fb = fbase + (gi & (R - 1)) * Cfg::SLOT_FLOATS;
        f[0] = *reinterpret_cast<const kz_f16x8*>(fb);
        f[1] = *reinterpret_cast<const kz_f16x8*>(fb + 128);
    };
    fetch_frags(f0, 0);
    int g = 0;        // half slices done (uniform)
    int seq = 0;      // half tiles done (uniform)
    int th_cur = 0;   // dual pass: threshold buffer of the current half tile (uniform)
    f32x16 acc[2][2];

    // one half tile (64 index rows) whose first half slice has global parity P0
    auto run_half = [&](const int tile, const int hf, auto start_parity) {
        constexpr int P0 = decltype(start_parity)::value;
        f32x16 binit[2];
        {
            const float* bp = bbuf + (seq & 1) * 64 + 4 * h;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 v = *reinterpret_cast<const float4*>(bp + 32 * rb + 8 * g4);
                    binit[rb][4 * g4 + 0] = v.x;
                    binit[rb][4 * g4 + 1] = v.y;
                    binit[rb][4 * g4 + 2] = v.z;
                    binit[rb][4 * g4 + 3] = v.w;
                }
            }
        }
        // bias rows of the next half tile (wave 0) and, in the dual pass, the smallest thresholds of its tile (wave 2; THREE
        // buffers: the value is read at the END of a half tile) by 4-byte LDS-DMA -- made visible by the slice barrier every half
        // tile contains.  Pinned behind the accumulator-init reads (hipcc orders a ds_read after an LDS-DMA behind vmcnt(0)).
        __builtin_amdgcn_sched_barrier(0);
        {
            const int nt = hf ? min(tile + 1, p.n_ytiles - 1) : tile;   // tile of the next half tile (behind the last one: any valid tile)
            unsigned lane4;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0" : "=v"(lane4));
            if (wave == 0)
                kz_glds4_s(p.ybias + (int64_t)nt * KZ_TILE + (hf ? 0 : 64), lane4, bbuf + ((seq + 1) & 1) * 64);
            else if (DUAL && wave == 2)
                kz_glds4_s(p.theta + (int64_t)nt * KZ_TILE, lane4, tbuf + (th_cur == 2 ? 0 : th_cur + 1) * 64);
        }
#pragma unroll
        for (int u = 0; u < NSR; ++u) {
            const bool odd = ((P0 + u) & 1) != 0;
            kz_f16x8 (&cur)[2] = odd ? f1 : f0;
            // fragments of the next half slice, under this one's MFMAs (it landed at least one barrier ago)
            __builtin_amdgcn_sched_barrier(0);
            if (odd)
                fetch_frags(f0, g + 1);
            else
                fetch_frags(f1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) {
                // (the second query half takes the bias rows through the C operand before the first overwrites them)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[rb], qf[1][0], binit[rb], 0, 0, 0);
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[rb], qf[0][0], binit[rb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[rb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[rb], qf[0][u], acc[rb][0], 0, 0, 0);
                    acc[rb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[rb], qf[1][u], acc[rb][1], 0, 0, 0);
                }
            }
            // One barrier per P half slices, after the half slices g with (g + 2) % P == 0.  Every wave that passes it has the
            // fragments of all half slices <= g + 1 in registers, so the slots of g-P+2 .. g+1 take g+P+2 .. g+2P+1 (this wave's
            // share: the two of its parity).  The next period prefetches g+2 .. g+P+1: issued at the PREVIOUS barrier, hence
            // vmcnt(0).
            if (((g + 2) & (P - 1)) == 0) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int i = 0; i < P / 2; ++i) dma_next();   // (issued one half slice later, behind the next MFMAs: +3 ... +4 % -- measured, not kept)
            }
            ++g;
        }
        __builtin_amdgcn_sched_barrier(0);
        float cthr0 = INFINITY, cthr1 = INFINITY;
        if constexpr (DUAL) {
            const float th = tbuf[th_cur * 64];   // the tile's smallest theta (rows sorted by theta; valid for both halves of the tile)
            cthr0 = tbuf[192 + 64 * (tid >> 6) + j] + th;
            cthr1 = tbuf[192 + 64 * (tid >> 6) + 32 + j] + th;
            th_cur = th_cur == 2 ? 0 : th_cur + 1;
        }
        kz_half_epilogue4<CAP, DUAL>(acc, st, pool, tile, hf, hf == 1 && tile == t_end - 1, msync, du, cthr0, cthr1);
        ++seq;
    };

    int tile = t_begin;
    for (;;) {
        run_half(tile, 0, std::integral_constant<int, 0>{});
        run_half(tile, 1, std::integral_constant<int, (NSR & 1)>{});
        if (++tile >= t_end) break;
    }
    // the sweep is over: the lists go to the output arrays in the layout kz_knn_finalize_kernel reads
    int lane_now;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
    if (lane_now < 32 && valid) {
#pragma unroll 1
        for (int qh = 0; qh < 2; ++qh) {
            const int64_t listoff = out_list_offset(qh);
#pragma unroll 4
            for (int e = 0; e < KP; ++e) {
                p.out_key[listoff + e] = st.list.kp()[e * 128 + 32 * qh];
                p.out_idx[listoff + e] = st.list.ip()[e * 128 + 32 * qh];
            }
        }
    }
}
