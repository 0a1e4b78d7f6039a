// Split-bf16 first pass of the fused kNN kernel (included by kz_knn.hip).
//
// Same contract as kz_knn_cand_kernel -- per-(query, lane-half) unsorted candidate lists of approximate keys
// key~ = q.y + bias(y), later certified and re-ranked in float64 by kz_knn_finalize_kernel -- but the products run on
// the bf16 matrix pipe, 16x the float32 MFMA rate.  Every operand is split as x = hi + lo + r (kz_pack.hip) and
//     q.y  ~=  hi_q.hi_y + hi_q.lo_y + lo_q.hi_y                  (3 x v_mfma_f32_32x32x16_bf16 per 16 k)
// accumulated in float32.  bf16 x bf16 products are exact in float32; the dropped terms are bounded by
// 3.1 * 2^-16 |q||y| (kz_bf16_gamma below), which only widens the certification margin: rows whose candidate set
// cannot be certified under the wider margin are re-done by the float32-MFMA kernel / the exact float64 kernels,
// so the result is still the float64 neighbour order.
//
// Structure (d_pad = 16 * NSR; NSR <= 16 at two workgroups per CU, NSR <= 24 (d <= 384) at one):
//   * the query tile is STATIONARY: the hi/lo fragments of all NSR slices stay in registers (8 VGPRs per slice),
//     so the only global stream is the index image -- which is what made the float32 kernel lose ~20 % (section 7);
//   * index slices go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write) into a ring
//     of four 8 KiB slots; ONE workgroup barrier per two slices, placed on the global slice counter so that tile
//     boundaries (epilogues) and barriers are independent;
//   * 2 workgroups per CU (the register budget of the stationary tile), 4 waves x 32 queries each.
#pragma once
#include <type_traits>

typedef __bf16 kz_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KZ_BF_LDS_BASE = 4 * 8192 + 1024 + 256;   // ring of 4 slices + 2 x 128 bias floats + merge flags
constexpr int KZ_BF_CAP = 20;    // log rows per lane; a group of four values is only scanned while every lane has 4 free
constexpr int KZ_BF_LDS = KZ_BF_LDS_BASE + (KZ_BF_CAP + 1) * 256 * 8;   // log rows 0..CAP-1 + one scratch row

// Relative rounding bound of the split-bf16 key (multiplies |y|max^2/2 + |q||y|max like the float32 bound does):
//   split:        |x - hi - lo| <= 2^-16 (1 + 2^-7) |x|  per operand  ->  dropped terms <= 3.1 * 2^-16 |q||y|
//   accumulation: 3 d_pad products + bias summed in float32 by the matrix pipe in an unspecified order; (n + 16) u
//                 is the any-order bound for round-to-nearest, doubled to allow for truncating internal adds.
static inline double kz_bf16_gamma(int d_pad) {
    return 3.1 * 1.52587890625e-05 + 2.0 * (double)(3 * d_pad + 16) * 5.9604644775390625e-08 + 1e-12;
}

// WPS = waves per SIMD the kernel is compiled for (always 2 here: up to 16 slices the stationary query fragments, 8 VGPRs
// per slice, fit beside the accumulators; beyond that kz_knn_cand_bf_ov_kernel runs one workgroup per CU).
template <int KP, int NSR, int WPS>
__global__ __launch_bounds__(256, WPS) void kz_knn_cand_bf_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);   // 4 slots x 2048 floats (8 KiB: planes hi0, hi1, lo0, lo1)
    float* bbuf = ybuf + 4 * 2048;                   // 2 x 128 bias floats
    int* msync = reinterpret_cast<int*>(bbuf + 256);  // 4 merge flags (kz_tile_epilogue2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int total = (t_end - t_begin) * NSR;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + j, p.lay, KP, s) + j;  // ONE list per query (halves = 1)
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_BF_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_BF_LDS_BASE + (KZ_BF_CAP + 1) * 256 * 4) + tid;
    if (h == 0) {  // the list belongs to the query: lane-half 0 owns it (kz_merge_logs_shared)
#pragma unroll 4
        for (int e = 0; e < KP; ++e) {
            st.lk[e * KZ_LSTRIDE] = -INFINITY;
            st.li[e * KZ_LSTRIDE] = -1;
        }
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    KzBlockMin<KP> bmin;
    bmin.init();
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    // LDS-DMA of one 8 KiB slice: lane l of wave w copies 16 B from src + (64 (w + 4c) + l) * 16 to the same offset
    // of the slot, c = 0, 1 (the image is copied linearly: it already is the LDS layout)
    const float* ysrc = p.ypack + ((int64_t)t_begin * NSR) * 2048 + tid * 4;
    auto dma_slice = [&](int gi) {
        const float* src = ysrc + (int64_t)min(gi, total - 1) * 2048;
        float* dst = ybuf + (gi & 3) * 2048 + wave * 256;  // wave-uniform LDS base (floats)
        kz_glds16(src, dst);
        kz_glds16(src + 1024, dst + 1024);
    };
    dma_slice(0);
    dma_slice(1);
    dma_slice(2);
    dma_slice(3);
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    if (tid < 4) msync[tid] = 0;
    // stationary query fragments: lane (j, h) holds k = 16 u + 8 h + 0..7 of query row 32 wave + j, hi and lo
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 2048 + (h * KZ_TILE + 32 * (tid >> 6) + j) * 4;
    kz_bf16x8 qh[NSR], ql[NSR];
#pragma unroll
    for (int u = 0; u < NSR; ++u) {
        qh[u] = *reinterpret_cast<const kz_bf16x8*>(qbase + u * 2048);
        ql[u] = *reinterpret_cast<const kz_bf16x8*>(qbase + u * 2048 + 1024);
    }
    __syncthreads();

    const float* fbase = ybuf + (h * KZ_TILE + j) * 4;  // this lane's fragment inside a slot: plane h, row j (+ 32 mt)
    auto load_frags = [&](kz_bf16x8 (&fh)[4], kz_bf16x8 (&fl)[4], const int gi) {
        const float* fb = fbase + (gi & 3) * 2048;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            fh[mt] = *reinterpret_cast<const kz_bf16x8*>(fb + 128 * mt);
            fl[mt] = *reinterpret_cast<const kz_bf16x8*>(fb + 128 * mt + 1024);
        }
    };
    int g = 0;
    f32x16 acc[4];
    // Even NSR: software pipeline over pairs of slices.  Fragment set A holds slice g (even), set B slice g+1; B is
    // fetched under A's MFMAs, the workgroup barrier sits in the middle of B's MFMAs (which only need registers), and
    // the first fragments of the NEXT pair are fetched right behind the barrier, under B's remaining MFMAs -- also
    // across a tile boundary, where they stay in registers during the epilogue.
    kz_bf16x8 ah[4], al[4], bh[4], bl[4];
    constexpr bool PIPE = (NSR % 2 == 0) && NSR <= 10;  // the second fragment set costs 32 VGPRs: beyond 10 slices it spills
    if (PIPE) load_frags(ah, al, 0);
    for (int tile = t_begin; tile < t_end; ++tile) {
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
        // bias rows of the next tile (one 4-byte load per thread per tile; parked in LDS after the first slice).
        // Pinned BEHIND the accumulator init: hipcc orders every ds_read after an LDS-DMA with s_waitcnt vmcnt(0); hoisted
        // above the init (as its scheduler does) this fresh load would be waited for at every tile start.
        __builtin_amdgcn_sched_barrier(0);
        const float bn = p.ybias[(int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE + (tid & 127)];
        if (PIPE) {
#pragma unroll
            for (int u = 0; u < NSR; u += 2) {
                __builtin_amdgcn_sched_barrier(0);
                load_frags(bh, bl, g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], qh[u], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], ql[u], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], qh[u], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[mt], qh[u + 1], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[mt], ql[u + 1], acc[mt], 0, 0, 0);
                if (u == 0) bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
                __builtin_amdgcn_sched_barrier(0);
                // slices g and g+1 are consumed (their fragments are in registers): after the barrier their slots take
                // slices g+4 and g+5, while g+2 and g+3 (issued one barrier ago, drained by the fence) become readable
                __syncthreads();
                dma_slice(g + 4);
                dma_slice(g + 5);
                load_frags(ah, al, g + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 2; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[mt], ql[u + 1], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[mt], qh[u + 1], acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                g += 2;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NSR; ++u) {
                kz_bf16x8 ch[4], cl[4];
                load_frags(ch, cl, g);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[mt], qh[u], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[mt], ql[u], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[mt], qh[u], acc[mt], 0, 0, 0);
                if (u == 0) bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
                if (g & 1) {
                    // slices g-1 and g are consumed: after the barrier their slots take slices g+3 and g+4, while g+1
                    // and g+2 (issued one barrier ago, drained by the fence of __syncthreads) are ready to be read
                    __syncthreads();
                    dma_slice(g + 3);
                    dma_slice(g + 4);
                }
                ++g;
            }
        }
        kz_tile_epilogue2<KP, KZ_BF_CAP>(acc, st, bmin, tile, tile == t_end - 1, h, msync, 0);
    }
}


// ---------------------------------------------------------------------------------------------------
// Overlapped form for one workgroup per CU (one wave per SIMD, nothing else to hide the epilogue behind): two accumulator
// sets; while the MFMAs of tile t run into one set, the candidate scan of tile t-1 is issued between the MFMA groups on
// the other set -- a wave issues in order, but an issued MFMA executes for 32 cycles in the matrix pipe, and the scan's
// VALU / LDS instructions fit into that shadow.  After the slice loop only the rare parts remain serial: merges and the
// groups that found a full log (kz_tile_epilogue2 with resume0).  512 VGPRs per lane at one wave per SIMD: 128
// accumulators + 8 NSR query registers + 32 fragment registers.
// ---------------------------------------------------------------------------------------------------
constexpr int KZ_OV_RING = 8;                                   // index-slice slots of the overlapped kernel
constexpr int KZ_OV_LDS_BASE = KZ_OV_RING * 8192 + 1024 + 256;  // ring + 2 x 128 bias floats + merge flags
constexpr int KZ_OV_LDS = KZ_OV_LDS_BASE + (KZ_BF_CAP + 1) * 256 * 8;

template <int KP, int NSR>
__global__ __launch_bounds__(256, 1) void kz_knn_cand_bf_ov_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);          // KZ_OV_RING slots x 2048 floats
    float* bbuf = ybuf + KZ_OV_RING * 2048;
    int* msync = reinterpret_cast<int*>(bbuf + 256);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int total = (t_end - t_begin) * NSR;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + j, p.lay, KP, s) + j;  // ONE list per query (halves = 1)
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_OV_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_OV_LDS_BASE + (KZ_BF_CAP + 1) * 256 * 4) + tid;
    if (h == 0) {  // the list belongs to the query: lane-half 0 owns it (kz_merge_logs_shared)
#pragma unroll 4
        for (int e = 0; e < KP; ++e) {
            st.lk[e * KZ_LSTRIDE] = -INFINITY;
            st.li[e * KZ_LSTRIDE] = -1;
        }
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    KzBlockMin<KP> bmin;
    bmin.init();
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    const float* ysrc = p.ypack + ((int64_t)t_begin * NSR) * 2048 + tid * 4;
    auto dma_slice = [&](int gi) {
        const float* src = ysrc + (int64_t)min(gi, total - 1) * 2048;
        float* dst = ybuf + (gi & (KZ_OV_RING - 1)) * 2048 + wave * 256;
        kz_glds16(src, dst);
        kz_glds16(src + 1024, dst + 1024);
    };
#pragma unroll
    for (int i = 0; i < KZ_OV_RING; ++i) dma_slice(i);
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    if (tid < 4) msync[tid] = 0;
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 2048 + (h * KZ_TILE + 32 * (tid >> 6) + j) * 4;
    kz_bf16x8 qh[NSR], ql[NSR];
#pragma unroll
    for (int u = 0; u < NSR; ++u) {
        qh[u] = *reinterpret_cast<const kz_bf16x8*>(qbase + u * 2048);
        ql[u] = *reinterpret_cast<const kz_bf16x8*>(qbase + u * 2048 + 1024);
    }
    __syncthreads();

    const float* fbase = ybuf + (h * KZ_TILE + j) * 4;
    int g = 0;
    constexpr int NG = 3 * NSR;  // MFMA groups (of four) per tile
    // fragments of the NEXT slice are fetched under the current slice's MFMAs (one wave per SIMD: nobody else hides the
    // LDS latency); with 8 ring slots slice g+1 is always visible while slice g is computed
    // Two static fragment sets, selected by the parity of the global slice counter (no register copies): a tile starts
    // at parity (tile index * NSR) & 1, which alternates exactly like the accumulator roles when NSR is odd and is
    // always 0 when NSR is even -- so each of the two run_tile instantiations knows its parity at compile time.
    kz_bf16x8 f0h[4], f0l[4], f1h[4], f1l[4];
    auto fetch_frags = [&](kz_bf16x8 (&fh)[4], kz_bf16x8 (&fl)[4], const int gi) {
        const float* fb = fbase + (gi & (KZ_OV_RING - 1)) * 2048;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            fh[mt] = *reinterpret_cast<const kz_bf16x8*>(fb + 128 * mt);
            fl[mt] = *reinterpret_cast<const kz_bf16x8*>(fb + 128 * mt + 1024);
        }
    };
    fetch_frags(f0h, f0l, 0);

    // one tile: MFMAs into `cur`; scan of the previous tile's keys in `prev` between the MFMA groups; then its tail
    auto run_tile = [&](f32x16 (&cur)[4], f32x16 (&prev)[4], const bool have_prev, const int tile, auto start_parity) {
        constexpr int P0 = decltype(start_parity)::value;
        {
            const float* bp = bbuf + (tile & 1) * 128 + 4 * h;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 v = *reinterpret_cast<const float4*>(bp + 32 * mt + 8 * g4);
                    cur[mt][4 * g4 + 0] = v.x;
                    cur[mt][4 * g4 + 1] = v.y;
                    cur[mt][4 * g4 + 2] = v.z;
                    cur[mt][4 * g4 + 3] = v.w;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // (see kz_knn_cand_bf_kernel: keep this load behind the LDS reads above)
        const float bn = p.ybias[(int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE + (tid & 127)];
        int stop = have_prev ? 16 : 0;
        const float tau_a = fmaxf(st.tau, __shfl_xor(st.tau, 32, 64));
        const int rowbase = (tile - 1) * KZ_TILE + 4 * h;
        // the previous tile's MFMAs issued long ago, but inline asm is invisible to the hazard recognizer: fence once
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(prev[0]), "+v"(prev[1]), "+v"(prev[2]), "+v"(prev[3]));
        unsigned long long mask = have_prev ? kz_epi_group_mask<KZ_BF_CAP>(prev, 0, tau_a) : 0ull;
#pragma unroll
        for (int u = 0; u < NSR; ++u) {
            kz_bf16x8 (&ch)[4] = ((P0 + u) & 1) ? f1h : f0h;
            kz_bf16x8 (&cl)[4] = ((P0 + u) & 1) ? f1l : f0l;
#pragma unroll
            for (int pg = 0; pg < 3; ++pg) {
                const int gidx = 3 * u + pg;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sidx = (gidx * 16) / NG; sidx < ((gidx + 1) * 16) / NG; ++sidx) {
                    const unsigned long long mnext = sidx + 1 < 16 ? kz_epi_group_mask<KZ_BF_CAP>(prev, (sidx + 1) & 15, tau_a) : 0ull;
                    kz_epi_step<KZ_BF_CAP>(prev, st, tau_a, rowbase, sidx, mask, stop);
                    mask = mnext;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pg == 0) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[mt], qh[u], cur[mt], 0, 0, 0);
                    // The next slice's fragments are requested AFTER the first MFMA group has consumed this slice's: the
                    // s_waitcnt hipcc places in front of that group then only covers reads issued a whole slice ago
                    // (behind the scan's branches it falls back to lgkmcnt(0), which would also wait for a fresh prefetch).
                    __builtin_amdgcn_sched_barrier(0);
                    if ((P0 + u) & 1)
                        fetch_frags(f0h, f0l, g + 1);
                    else
                        fetch_frags(f1h, f1l, g + 1);
                } else if (pg == 1) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[mt], ql[u], cur[mt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) cur[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[mt], qh[u], cur[mt], 0, 0, 0);
                }
            }
            if (u == 0) bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
            if (g & 1) {
                // slices g-1 and g are consumed: their slots take slices g-1+RING and g+RING, three barrier periods ahead
                // of their use (with one wave per SIMD nothing else covers the L2 latency of a late DMA)
                // Until the next barrier this wave READS slices up to g+3 (g+1, g+2 are computed, the fragments of g+3 are
                // prefetched during g+2).  g+3 and g+4 were issued two barriers ago, so only the 4 wave-loads of the LAST
                // barrier (slices g+5, g+6) may stay in flight: vmcnt counts in issue order, anything older than the 4
                // newest operations is complete.  (__syncthreads() would drain vmcnt(0): wait for DMAs not needed for two
                // more periods.  vmcnt(8) -- forgetting the one-slice prefetch -- raced: 3 wrong rows in 3000 on a
                // 1M-row index, caught by the rounding-bound self-check, tests/test_gpu_fullsize.py.)
                asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                dma_slice(g - 1 + KZ_OV_RING);
                dma_slice(g + KZ_OV_RING);
            }
            ++g;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (have_prev) kz_tile_epilogue2<KP, KZ_BF_CAP>(prev, st, bmin, tile - 1, false, h, msync, stop);
    };

    f32x16 acc0[4], acc1[4];
    int tile = t_begin;
    bool have_prev = false;
    for (;;) {
        run_tile(acc0, acc1, have_prev, tile, std::integral_constant<int, 0>{});
        have_prev = true;
        if (++tile >= t_end) {
            kz_tile_epilogue2<KP, KZ_BF_CAP>(acc0, st, bmin, tile - 1, true, h, msync, 0);
            break;
        }
        run_tile(acc1, acc0, true, tile, std::integral_constant<int, (NSR & 1)>{});
        if (++tile >= t_end) {
            kz_tile_epilogue2<KP, KZ_BF_CAP>(acc1, st, bmin, tile - 1, true, h, msync, 0);
            break;
        }
    }
}
