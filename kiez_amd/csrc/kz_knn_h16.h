// fp16 first pass of the fused kNN kernel: ONE v_mfma_f32_32x32x16_f16 per 16 k (the split-bf16 pass needs three).
//
// Same contract as the other fused kernels -- per-query unsorted lists of the K' best APPROXIMATE keys, certified and
// re-ranked in float64 by kz_knn_finalize_kernel -- but the operands are the centred, scaled fp16 image of kz_pack.hip
// (kz_himage):   x_h = half(S (x - mu)),   key~ = q_h . y_h - S^2 |y_c|^2 / 2.
// fp16 x fp16 products are exact in float32, so the only errors are the operand rounding (whose per-row residual norms
// |x_c - x_h| are MEASURED at pack time, not bounded a priori) and the float32 accumulation; the certification in
// kz_knn_finalize_kernel uses exactly those.  Centring is what makes one fp16 product enough:
// distances are translation invariant, the rounding error scales with |q_c||y_c|, and embeddings with a large common
// mean (rng.rand: |x|^2 = d/3, |x - mu|^2 = d/12) lose a factor four of it.
//
// Structure: 4 waves x 32 queries per workgroup, the query tile STATIONARY in registers (4 VGPRs per 16-k slice), the
// index image streamed global -> LDS by LDS-DMA (one 16-k slice of 128 index rows = 4 KiB: two planes k 0-7 / 8-15 of
// 128 rows x 16 B, copied linearly because the image already is the conflict-free ds_read_b128 layout) through a ring of
// R = 2 P slots with one workgroup barrier per P slices; fragments of slice g+1 are read under the MFMAs of slice g.
// Candidate selection: kz_knn_epi3.h.  Three workgroups per CU (168 VGPRs) up to d = 128, two beyond.
#pragma once
#include <type_traits>

#include "kz_knn_epi3.h"

typedef _Float16 kz_f16x8 __attribute__((ext_vector_type(8)));

// Per (list length, occupancy class) configuration: ring slots, event-pool capacity per wave and where the lists live.  LDS per workgroup: 53.3 KiB at three workgroups per CU, 80 KiB at two.
// (Every tile must contain a slice barrier -- the bias double buffer and the merge flags rely on it -- so the barrier
// period RING / 2 never exceeds the slice count; NSR >= 2 is required by the host.)
// DUAL: the dual-pass build (kz_knn_epi3.h "Dual pass"): + 1.5 KiB (thresholds of three tiles, the queries' offsets).
// LDS-DMA copies that a barrier releases are issued one slice LATER, behind that slice's MFMAs, where this wave has no LDS read in
// flight (the guide prices an LDS-DMA issued among LDS reads at 100-185 cycles of the issuing wave, 25-60 in a gap without them).
// 1 (default) = at two workgroups per CU (8-slot ring, four slices per barrier: the copies still lead their use by three slices);
// 2 = everywhere; 0 = never.  Same box, main kernel, default -> late: C4 share 140.2 -> 136.8 ms, C3 122.7 -> 119.2 ms; at three
// workgroups per CU (4-slot ring: the lead shrinks to one slice) ns 91.0 -> 92.1, ordinary kernel 87.9 -> 91.1; C1 2.75 -> 2.75.
#ifndef KZ_H_DMA_LATE
#define KZ_H_DMA_LATE 1
#endif
// WIDE: ONE workgroup of 4 x WPS waves per CU instead of WPS workgroups of 4 waves: its WPS query tiles share one ring, so
// every index slice is copied into the CU's LDS once instead of WPS times.  Measured on the scan-less diagnostic kernel of round 3 (profiles/r03_ablation.md section 2; builds
// 1 5 6 7; 250k x 1M x 200, same box): bare loop 71.2 ms, + slice barriers 71.2 -> 72.9, + the LDS-DMA traffic 83.4 -- and
// with a third of the DMA volume 75.3: the copies, not the barriers, are what the ring costs, in proportion to their volume.
// What the wide build gives back: all waves of the CU now run in lockstep -- they reach every barrier and every tile epilogue
// together, so nothing covers them (narrow: the three workgroups of a CU are in different phases).  Net, same-box: ordinary
// kernel 250k x 1M x 200 -3 % ... +0.5 %, x 300 -1 % ... -4 %, shared sweep +1 % ... +10 %; 100k x 100k x 128 +3 %; K' = 64 +8 %.
// Off by default (context option "h_wide" = 1 turns it on for K' = 16 with more than 8 slices); parity-tested both ways.
template <int KP, int WPS, int NSR, bool DUAL = false, bool WIDE = false>
struct KzHCfg {
    static constexpr int TPW = WIDE ? WPS : 1;                         // query tiles (of 128 rows) per workgroup
    static constexpr bool LDS_LIST = KP <= 32;
    static constexpr bool LISTS_FIT = WPS == 2 || KP == 16;   // K' = 32 lists do not fit beside the ring at 3 per CU
    // where the lists live (KzListRef, kz_knn_epi3.h): 1 = LDS, 2 = keys in LDS + rows in the output arrays, 0 = output arrays.
    // The hybrid needs K' x 512 B: K' = 64 at two per CU (with a 4-slot ring), K' = 32 at three per CU (with a smaller pool).
    static constexpr int LMODE = (LDS_LIST && LISTS_FIT) ? 1 : (((KP == 64 && WPS == 2) || (KP == 32 && WPS == 3)) ? 2 : 0);
    static constexpr bool IN_LDS = LMODE == 1;
    // WIDE: eight slots, one barrier per four slices -- a barrier of a wide workgroup stops every wave of the CU (same-box,
    // 250k x 1M x 200, ordinary kernel: 4 slots / 2 slices per barrier 91.8 ms, 8 / 4: 85.0 ms, 12 / 6: 90.0 ms, narrow 88-89.6;
    // the wave groups staggered by a period with a barrier per 2 slices: 106 ms -- profiles/r03_ablation.md)
    static constexpr int RING = WIDE ? 8 : ((WPS == 3 || KP == 32 || NSR < 4) ? 4 : 8);
    static constexpr int PERIOD = RING / 2;                            // slices per barrier: a slot is refilled one period before it is read
    // (three per CU: the workgroup must stay within 42 LDS granules of 1280 B -- 52.5 KiB with the lists of K' = 16 or the
    //  keys of K' = 32; the dual-pass build pays for its 1.5 KiB of thresholds and query offsets with 16 pool entries)
    static constexpr int CAP = WPS == 3 ? (KP <= 32 ? (DUAL ? 176 : 192) : 256) : (LMODE == 2 && RING == 8 ? (DUAL ? 140 : 156) : 256);     // event-pool entries per wave (24 B each)
    static constexpr int RING_BYTES = RING * 4096;
    static constexpr int BIAS_OFF = RING_BYTES;                        // 2 x 128 floats
    static constexpr int SYNC_OFF = BIAS_OFF + 1024;                   // 4 merge flags (+ padding)
    static constexpr int THETA_OFF = SYNC_OFF + 256;                   // dual pass: 3 x 64 threshold floats + 128 query offsets per tile
    static constexpr int POOLK_OFF = THETA_OFF + (DUAL ? 768 + 768 * TPW : 0);    // [4 TPW waves][CAP] x 4 floats
    static constexpr int POOLM_OFF = POOLK_OFF + 4 * TPW * CAP * 16;   // [4 TPW waves][CAP] x {code, next}
    static constexpr int LIST_OFF = POOLM_OFF + 4 * TPW * CAP * 8;     // per tile: keys [KP][128], then rows [KP][128]
    static constexpr int LIST_BLOCK = LMODE == 1 ? KP * 128 * 8 : (LMODE == 2 ? KP * 128 * 4 : 0);
    static constexpr int LDS_BYTES = LIST_OFF + TPW * LIST_BLOCK;
    static_assert(LDS_BYTES <= 160 * 1024, "workgroup exceeds the CU's LDS");
};

template <int KP, int NSR, int WPS, bool DUAL = false, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? 256 * WPS : 256, WIDE ? 1 : WPS) void kz_knn_cand_h_kernel(KnnCandParams p) {
    using Cfg = KzHCfg<KP, WPS, NSR, DUAL, WIDE>;
    constexpr int TPW = Cfg::TPW;
    constexpr int R = Cfg::RING, P = Cfg::PERIOD, CAP = Cfg::CAP;
    static_assert((R & (R - 1)) == 0 && R == 2 * P, "slot arithmetic below: a power-of-two ring of two periods");
    constexpr bool LATE = KZ_H_DMA_LATE == 2 || (KZ_H_DMA_LATE == 1 && WPS == 2 && !WIDE);
    constexpr int IN_LDS = Cfg::LMODE;   // list storage mode (KzListRef)
    // at three waves per SIMD (168 VGPRs) the first fragments of the next tile are NOT fetched across the epilogue: the 16
    // registers they would occupy there are what keeps the stationary query tile out of scratch memory.  Exception: a tile
    // whose LAST slice carries the barrier (even global slice index: odd NSR, tile starting at parity 0) must have read
    // slice g+1 before that barrier hands its slot to the DMA engine -- that tile carries, and its successor does not re-fetch.
    constexpr bool CARRY = WPS != 3;
    // beyond 8 slices the stationary query tile leaves no room for a second fragment set at three waves per SIMD: the
    // fragments of a slice are then fetched right before its MFMAs (two other waves of the SIMD cover the LDS latency)
    constexpr bool ONE_SET = WPS == 3 && NSR > 8;
    constexpr bool RECOMP = DUAL && ONE_SET;   // kz_merge_pool3: block minima re-read per merge instead of carried
    constexpr int LAG = ONE_SET ? 1 : 2;   // slices between a barrier and the oldest slot it may hand to the DMA engine
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);                       // R slots x 1024 floats
    float* bbuf = reinterpret_cast<float*>(smem + Cfg::BIAS_OFF);       // 2 x 128 bias floats
    float* tbuf = reinterpret_cast<float*>(smem + Cfg::THETA_OFF);      // dual pass: 3 x 64 thresholds (first half of a tile's rows), then -bias of the 128 queries
    kz_lds_i32* msync = (kz_lds_i32*)(smem + Cfg::SYNC_OFF);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
#ifdef KZ_ABL_STAMP   // (diagnostic build, tools/ab_build.sh + option "abl_stamp": when every workgroup started and ended, 100 MHz clock)
    const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int total = (t_end - t_begin) * NSR;
    // WIDE: waves 4 b .. 4 b + 3 take query tile wd.x + b; a workgroup at the end of the launch may reach past its last
    // tile: those waves sweep along (barriers, and nothing else, need them) on the last valid tile's rows with every
    // threshold at +inf -- no event, no list traffic, no output
    const int wq = WIDE ? (wave & 3) : wave;            // wave within its query tile (uniform)
    const int wq_v = WIDE ? ((tid >> 6) & 3) : (tid >> 6);
    const int tb = WIDE ? (wave >> 2) : 0;              // tile within the workgroup (uniform)
    const bool valid = !WIDE || wd.x + tb < p.n_qtiles;
    const int qt = WIDE ? (valid ? wd.x + tb : p.n_qtiles - 1) : wd.x;

    // this query's list in the output arrays (ONE list per query and index range, K' contiguous entries)
    auto out_list_offset = [&]() { return kz_list_contig_off((int64_t)qt * KZ_TILE + 32 * wq_v + j, p.lay, KP, s); };
    KzCandState3<IN_LDS> st;
    if constexpr (IN_LDS == 1) {
        st.list.k = (kz_lds_f32*)(smem + Cfg::LIST_OFF + tb * Cfg::LIST_BLOCK) + 32 * wq_v + j;
        st.list.i_off = KP * 128;
    } else if constexpr (IN_LDS == 2) {
        st.list.k = (kz_lds_f32*)(smem + Cfg::LIST_OFF + tb * Cfg::LIST_BLOCK) + 32 * wq_v + j;
        st.list.ib = p.out_idx;
        {
            // (uniform: the offsets of this wave's query 0 and of its query 1 -- lists of consecutive queries are equally spaced)
            const int64_t row0 = (int64_t)qt * KZ_TILE + 32 * wq;
            const int64_t o0 = kz_list_contig_off(row0, p.lay, KP, s);
            st.list.off_u = (unsigned)o0;
            st.list.stride = (unsigned)(kz_list_contig_off(row0 + 1, p.lay, KP, s) - o0);
        }
    } else {
        // (uniform bases + a 32-bit per-lane element offset: no 64-bit per-lane pointers to keep alive; a launch's lists stay
        //  below 2^32 elements -- kz_prepare_pass refuses a launch whose lists reach that)
        st.list.kb = p.out_key;
        st.list.ib = p.out_idx;
        st.list.off = (unsigned)out_list_offset();
    }
    KzWavePool pool;
    pool.keys = (__attribute__((address_space(3))) f32x4e*)(smem + Cfg::POOLK_OFF) + wave * CAP;
    pool.meta = (__attribute__((address_space(3))) i32x2e*)(smem + Cfg::POOLM_OFF) + wave * CAP;
    // (seeded lists: KnnCandParams::qfloor)
    const float fl = p.qfloor ? p.qfloor[(int64_t)(p.qt0 + qt) * KZ_TILE + 32 * wq_v + j] : -INFINITY;
    if (h == 0 && valid) {  // the list belongs to the query: lane-half 0 owns it (kz_merge_logs3)
#pragma unroll 4
        for (int e = 0; e < KP; ++e) {
            st.list.kp()[e * KzListRef<IN_LDS>::KSTRIDE] = fl;
            st.list.ip()[e * KzListRef<IN_LDS>::ISTRIDE] = -1;
        }
    }
    if (total <= 0) {
        if constexpr (IN_LDS != 0) {
            const int64_t listoff = out_list_offset();
            if (h == 0 && valid)
                for (int e = 0; e < KP; ++e) {
                    p.out_key[listoff + e] = -INFINITY;
                    if constexpr (IN_LDS == 1) p.out_idx[listoff + e] = -1;   // (hybrid: the rows were initialised in place above)
                }
        }
        return;
    }
    st.tau = valid ? fl : INFINITY;
    KzBlockMin3<KP> bmin;
    bmin.init(fl);
    st.head = -1;
    pool.cnt = 0;
    pool.tiles_done = 0;
    pool.next_merge = 1;

    // LDS-DMA of one 4 KiB slice: lane l of wave w copies 16 B from slice base + (64 w + l) * 16 to the same offset of the
    // slot; one wave-instruction per wave and slice.  The source is a wave-uniform running pointer (scalar base + lane
    // offset: two scalar adds per slice instead of a clamped 64-bit index computation); slices are issued strictly in
    // order, and the ring runs up to R + P slices past the end of the sweep -- into the next tiles of the image or into
    // the padding kz_himage_build allocates behind it (those slots are never read).
    const char* dma_src = reinterpret_cast<const char*>(p.ypack) + ((int64_t)t_begin * NSR) * 4096;   // uniform
    int dma_slot = 0;   // uniform: slot of the next slice to issue
    int dma_turn = 0;   // WIDE: the group of four waves that copies the next slice (the groups take turns)
    const int lane_off = (WIDE ? (tid & 255) : tid) * 16;
    auto dma_next = [&]() {
        float* dst = ybuf + dma_slot * 1024 + wq * 256;  // wave-uniform LDS base (floats)
        if (!WIDE || tb == dma_turn) kz_glds16_s(dma_src, (unsigned)lane_off, dst);
        if constexpr (WIDE) dma_turn = dma_turn + 1 == TPW ? 0 : dma_turn + 1;
        dma_src += 4096;
        dma_slot = (dma_slot + 1) & (R - 1);
    };
#pragma unroll
    for (int i = 0; i < R; ++i) dma_next();
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    KzDualRef du;
    if constexpr (DUAL) {
        if (tid < 64) tbuf[tid] = p.theta[(int64_t)t_begin * KZ_TILE + tid];
        du.qrow0 = (p.qt0 + qt) * KZ_TILE + 32 * wq;
        // this query's own offset: read back from LDS in every epilogue (a register held for the whole sweep was spilled at
        // three workgroups per CU, and reloaded behind a wait for the DMA ring)
        if (h == 0) tbuf[192 + 32 * (tid >> 6) + j] = valid ? p.qnbias[du.qrow0 + j] : INFINITY;
    }
    if (tid < 4) msync[tid] = 0;
    // stationary query fragments: lane (j, h) holds k = 16 u + 8 h + 0..7 of query row 32 wave + j
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 1024 + (h * KZ_TILE + 32 * wq_v + j) * 4;
    kz_f16x8 qf[NSR];
#pragma unroll
    for (int u = 0; u < NSR; ++u) qf[u] = *reinterpret_cast<const kz_f16x8*>(qbase + u * 1024);
    // the whole prologue ring must have landed before anyone reads it: said explicitly (the copies are inline asm, invisible to the
    // compiler's barrier; every wave's bias-row store above happens to wait for them too -- kz_knn_h64.h shows what happens without)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float* fbase = ybuf + (h * KZ_TILE + j) * 4;  // this lane's fragment inside a slot: plane h, row j (+ 32 mt)
    // two static fragment sets selected by the parity of the global slice counter (no register copies)
    kz_f16x8 f0[4], f1[4];
    int dma_due = 0;   // KZ_H_DMA_LATE: a barrier has released slots whose copies are still to be issued (uniform)
    auto fetch_frags = [&](kz_f16x8 (&f)[4], const int gi) {
        const float* fb = fbase + (gi & (R - 1)) * 1024;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) f[mt] = *reinterpret_cast<const kz_f16x8*>(fb + 128 * mt);
    };
    if (CARRY) fetch_frags(f0, 0);
    int g = 0;
    int th_cur = 0;   // dual pass: threshold buffer of the current tile (uniform)
    f32x16 acc[4];

    // one tile whose first slice has global parity P0 (compile time: the parity alternates from tile to tile when NSR is odd)
    auto run_tile = [&](const int tile, auto start_parity) {
        constexpr int P0 = decltype(start_parity)::value;
        {
            int h_now = h;
            if constexpr (WPS == 3) {
                // (lane half re-made here: the base address below, kept in a register across the tile, is what gets spilled at
                //  three workgroups per CU -- and reloaded behind a wait for the whole DMA ring)
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 5, %0" : "=v"(h_now));
            }
            const float* bp = bbuf + (tile & 1) * 128 + 4 * h_now;
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
        // bias rows of the next tile: 128 floats by LDS-DMA (waves 0 and 1, one dword per lane), issued at the start of this
        // tile -- every tile contains a slice barrier behind this point, whose vmcnt(0) + s_barrier make them visible to
        // all waves before the next tile reads them.  No VGPR round trip, no extra wait.  Pinned BEHIND the accumulator
        // init (hipcc orders every ds_read after an LDS-DMA behind s_waitcnt vmcnt(0)).
        __builtin_amdgcn_sched_barrier(0);
        {
            // lane offset tid * 4 of the 4-byte copies: derived from the 16-byte one HERE, by an instruction the compiler cannot
            // hoist -- kept alive across the tile it was spilled at three workgroups per CU and reloaded (scratch_load +
            // s_waitcnt vmcnt(0): a wait for the whole DMA ring) at the top of every tile
            unsigned off4;
            asm volatile("v_lshrrev_b32 %0, 2, %1" : "=v"(off4) : "v"(lane_off));
            if (wave < 2)
                kz_glds4_s(p.ybias + (int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE, off4, bbuf + ((tile + 1) & 1) * 128 + wave * 64);
            else if (DUAL && wave == 2)
                // ... and its smallest thresholds (the rows are sorted by threshold: the epilogue only reads the first), by a
                // third wave (the -128 floats of its lane offset are folded into the scalar base).  THREE buffers: the value is
                // read at the END of a tile, so a wave that is already here may not overwrite what a slower wave still reads
                // for the previous tile; the buffer written here was last read two tiles ago, with a slice barrier in between.
                kz_glds4_s(p.theta + ((int64_t)min(tile + 1, p.n_ytiles - 1) - 1) * KZ_TILE, off4, tbuf + (th_cur == 2 ? 0 : th_cur + 1) * 64);
        }
        constexpr bool carry_in = !ONE_SET && (CARRY || (P0 == 1 && (NSR & 1)));
        constexpr bool carry_out = !ONE_SET && (CARRY || (((P0 + NSR) & 1) != 0));
        if (!carry_in && !ONE_SET) {
            if (P0)
                fetch_frags(f1, g);
            else
                fetch_frags(f0, g);
        }
#pragma unroll
        for (int u = 0; u < NSR; ++u) {
            const bool odd = ((P0 + u) & 1) != 0;
            kz_f16x8 (&cur)[4] = (odd && !ONE_SET) ? f1 : f0;
            // fragments of the next slice, under this slice's MFMAs (it landed at least one barrier ago)
            __builtin_amdgcn_sched_barrier(0);
            if (ONE_SET) {
                fetch_frags(f0, g);
            } else if (carry_out || u + 1 < NSR) {
                if (odd)
                    fetch_frags(f0, g + 1);
                else
                    fetch_frags(f1, g + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[mt], qf[u], acc[mt], 0, 0, 0);
            if constexpr (LATE) {
                // (the copies the PREVIOUS slice's barrier released, issued here: see KZ_H_DMA_LATE)
                if (dma_due) {
#pragma unroll
                    for (int i = 0; i < P; ++i) dma_next();
                    dma_due = 0;
                }
            }
            // One barrier per P slices, after the slices g with (g + 2) % P == 0.  Every wave that passes it has the
            // fragments of all slices <= g + 1 in registers (lgkmcnt(0)), so the slots of slices g-P+2 .. g+1 take
            // slices g+P+2 .. g+2P+1.  The next period prefetches slices g+2 .. g+P+1: those were issued at the PREVIOUS
            // barrier and are this wave's youngest DMAs, hence vmcnt(0) (any other outstanding operation -- bias load,
            // list traffic of a merge -- only has to finish too).
            // (ONE_SET: nothing is prefetched, a wave at the barrier has read the slices <= g only: the barrier sits one
            //  slice later in the period -- (g + 1) % P == 0 -- and hands out the slots of slices g-P+1 .. g.)
            if ((ONE_SET ? odd : !odd) && (P == 2 || ((g + LAG) & (P - 1)) == 0)) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if constexpr (LATE) {
                    dma_due = 1;   // (issued behind the next slice's MFMAs, see KZ_H_DMA_LATE)
                } else {
                    {
#pragma unroll
                        for (int i = 0; i < P; ++i) dma_next();   // slices g+P+LAG .. g+2P+LAG-1, in order
                    }
                }
            }
            ++g;
        }
        __builtin_amdgcn_sched_barrier(0);
        float cthr = INFINITY;
        if constexpr (DUAL) {
            cthr = tbuf[192 + 32 * (tid >> 6) + j] + tbuf[th_cur * 64];   // this query's offset + the tile's smallest theta (rows sorted by theta)
            th_cur = th_cur == 2 ? 0 : th_cur + 1;
        }
        kz_tile_epilogue3<KP, CAP, IN_LDS, DUAL, RECOMP>(acc, st, pool, bmin, tile, tile == t_end - 1, msync, du, cthr);
    };

    int tile = t_begin;
#ifdef KZ_ABL_STAMP
    // (per-tile stamps of this workgroup's first 64 tiles, then of every 16th: [W][128] in log_keys)
    auto tile_stamp = [&](int t) {
        if constexpr (!DUAL) {
            const int rel = t - t_begin;
            const int slot = rel < 64 ? rel : 64 + ((rel - 64) >> 4);
            if (p.log_keys && threadIdx.x == 0 && slot < 128 && (rel < 64 || ((rel - 64) & 15) == 0))
                ((unsigned long long*)p.log_keys)[(size_t)blockIdx.x * 128 + slot] = __builtin_amdgcn_s_memrealtime();
        }
    };
#else
    auto tile_stamp = [](int) {};
#endif
    for (;;) {
        tile_stamp(tile);
        run_tile(tile, std::integral_constant<int, 0>{});
        if (++tile >= t_end) break;
        tile_stamp(tile);
        run_tile(tile, std::integral_constant<int, (NSR & 1)>{});
        if (++tile >= t_end) break;
    }
    if constexpr (IN_LDS != 0) {
        // the sweep is over: what lived in LDS goes to the output arrays in the layout kz_knn_finalize_kernel reads
        const int64_t listoff = out_list_offset();
        // (lane number re-made here: the `h == 0` mask of the prologue, kept for this one use, cost a VGPR as SGPR spill space)
        int lane_now;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_now));
        if (lane_now < 32 && valid) {
#pragma unroll 4
            for (int e = 0; e < KP; ++e) {
                p.out_key[listoff + e] = st.list.kp()[e * 128];
                if constexpr (IN_LDS == 1) p.out_idx[listoff + e] = st.list.ip()[e * 128];
            }
        }
    }
#ifdef KZ_ABL_STAMP
    if constexpr (!DUAL) {
        if (p.log_meta && threadIdx.x == 0) {
            unsigned long long* o = (unsigned long long*)p.log_meta + 2 * (size_t)blockIdx.x;
            o[0] = stamp0;
            o[1] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}
