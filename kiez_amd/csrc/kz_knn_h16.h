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
// Structure: the stationary-query / LDS-DMA-ring design of the split-bf16 kernel with half the bytes per slice
// (one 16-k slice of 128 index rows = 4 KiB: two planes k 0-7 / 8-15 of 128 rows x 16 B), a ring of eight slots filled
// two barrier periods ahead and a counted vmcnt in front of the slice barrier.  Registers per lane: 64 accumulators +
// 4 NSR query + 2 x 16 fragment VGPRs, so three workgroups per CU fit up to d = 208 and two up to d = 384.
#pragma once
#include <type_traits>

typedef _Float16 kz_f16x8 __attribute__((ext_vector_type(8)));

constexpr int KZ_H_RING = 8;                                       // 4 KiB slots
constexpr int KZ_H_LDS_BASE = KZ_H_RING * 4096 + 1024 + 256;       // ring + 2 x 128 bias floats + merge flags
template <int CAP>
constexpr int kz_h_lds_bytes() { return KZ_H_LDS_BASE + (CAP + 1) * 256 * 8; }   // log rows 0..CAP-1 + one scratch row

template <int KP, int NSR, int WPS, int CAP>
__global__ __launch_bounds__(256, WPS) void kz_knn_cand_h_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);          // KZ_H_RING slots x 1024 floats
    float* bbuf = ybuf + KZ_H_RING * 1024;                 // 2 x 128 bias floats
    int* msync = reinterpret_cast<int*>(bbuf + 256);       // 4 merge flags (kz_tile_epilogue2)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int total = (t_end - t_begin) * NSR;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + j, p.lay, KP, s) + j;  // ONE list per query
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_H_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_H_LDS_BASE + (CAP + 1) * 256 * 4) + tid;
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

    // LDS-DMA of one 4 KiB slice: lane l of wave w copies 16 B from src + (64 w + l) * 16 to the same offset of the slot
    // (the image is copied linearly: it already is the LDS layout); one wave-instruction per wave and slice
    const float* ysrc = p.ypack + ((int64_t)t_begin * NSR) * 1024 + tid * 4;
    auto dma_slice = [&](int gi) {
        const float* src = ysrc + (int64_t)min(gi, total - 1) * 1024;
        float* dst = ybuf + (gi & (KZ_H_RING - 1)) * 1024 + wave * 256;  // wave-uniform LDS base (floats)
        kz_glds16(src, dst);
    };
#pragma unroll
    for (int i = 0; i < KZ_H_RING; ++i) dma_slice(i);
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    if (tid < 4) msync[tid] = 0;
    // stationary query fragments: lane (j, h) holds k = 16 u + 8 h + 0..7 of query row 32 wave + j
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * NSR) * 1024 + (h * KZ_TILE + 32 * (tid >> 6) + j) * 4;
    kz_f16x8 qf[NSR];
#pragma unroll
    for (int u = 0; u < NSR; ++u) qf[u] = *reinterpret_cast<const kz_f16x8*>(qbase + u * 1024);
    __syncthreads();   // (drains vmcnt(0): the whole prologue ring has landed)

    const float* fbase = ybuf + (h * KZ_TILE + j) * 4;  // this lane's fragment inside a slot: plane h, row j (+ 32 mt)
    // two static fragment sets selected by the parity of the global slice counter (no register copies)
    kz_f16x8 f0[4], f1[4];
    auto fetch_frags = [&](kz_f16x8 (&f)[4], const int gi) {
        const float* fb = fbase + (gi & (KZ_H_RING - 1)) * 1024;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) f[mt] = *reinterpret_cast<const kz_f16x8*>(fb + 128 * mt);
    };
    fetch_frags(f0, 0);
    int g = 0;
    f32x16 acc[4];
#ifdef KZ_STAMP
    unsigned long long c_slices = 0, c_epi = 0, c_merge = 0, n_pass = 0, n_ins = 0, c_dma = 0, c_bar = 0, c_e1 = 0, c_e2 = 0;
#endif

    // one tile whose first slice has global parity P0 (compile time: the parity alternates from tile to tile when NSR is odd)
    auto run_tile = [&](const int tile, auto start_parity) {
        constexpr int P0 = decltype(start_parity)::value;
        KZ_T(t0);
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
        // bias rows of the next tile: one 4-byte load per thread and tile, parked in LDS after the first slice; pinned
        // BEHIND the accumulator init (hipcc orders every ds_read after an LDS-DMA behind s_waitcnt vmcnt(0): hoisted
        // above the init this fresh load would be waited for at every tile start)
        __builtin_amdgcn_sched_barrier(0);
        const float bn = p.ybias[(int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE + (tid & 127)];
#pragma unroll
        for (int u = 0; u < NSR; ++u) {
            constexpr int dummy = 0;
            (void)dummy;
            const bool odd = ((P0 + u) & 1) != 0;
            kz_f16x8 (&cur)[4] = odd ? f1 : f0;
            // fragments of the next slice, under this slice's MFMAs (it landed at least one barrier ago)
            __builtin_amdgcn_sched_barrier(0);
            if (odd)
                fetch_frags(f0, g + 1);
            else
                fetch_frags(f1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[mt], qf[u], acc[mt], 0, 0, 0);
            if (u == 0) bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
            if (odd) {
                // Slices g-1 and g are consumed by every wave once it passes this barrier (their fragments were read one
                // slice ago); their slots take slices g-1+RING and g+RING.  Until the NEXT barrier this wave reads slices
                // g+2 (now prefetching g+1 is done) and g+3: both were issued two barriers ago, so only the two
                // wave-loads of the LAST barrier (slices g+5, g+6) may still be in flight -- vmcnt counts in issue order,
                // and any younger operation (bias load, list traffic of a merge) only makes the wait stricter.
#ifdef KZ_STAMP
                {
                    __builtin_amdgcn_sched_barrier(0);
                    const unsigned long long w0 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                    const unsigned long long w1 = __builtin_amdgcn_s_memtime();
                    asm volatile("s_barrier" ::: "memory");
                    c_dma += w1 - w0;
                    c_bar += __builtin_amdgcn_s_memtime() - w1;
                    __builtin_amdgcn_sched_barrier(0);
                }
#else
                asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
                dma_slice(g - 1 + KZ_H_RING);
                dma_slice(g + KZ_H_RING);
            }
            ++g;
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef KZ_STAMP
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        kz_tile_epilogue2<KP, CAP>(acc, st, bmin, tile, tile == t_end - 1, h, msync, 0, c_merge, n_pass, n_ins, c_e1, c_e2);
        __builtin_amdgcn_sched_barrier(0);
        c_slices += t1 - t0;
        c_epi += __builtin_amdgcn_s_memtime() - t1;
#else
        kz_tile_epilogue2<KP, CAP>(acc, st, bmin, tile, tile == t_end - 1, h, msync, 0);
#endif
    };

    int tile = t_begin;
    for (;;) {
        run_tile(tile, std::integral_constant<int, 0>{});
        if (++tile >= t_end) break;
        run_tile(tile, std::integral_constant<int, (NSR & 1)>{});
        if (++tile >= t_end) break;
    }
#ifdef KZ_STAMP
    if (lane == 0 && p.dbg) {
        atomicAdd(p.dbg + 0, c_slices);
        atomicAdd(p.dbg + 1, c_epi);
        atomicAdd(p.dbg + 3, (unsigned long long)(t_end - t_begin));
        atomicAdd(p.dbg + 4, c_merge);
        atomicAdd(p.dbg + 5, n_pass);
        atomicAdd(p.dbg + 6, n_ins);
        atomicAdd(p.dbg + 7, c_dma);
        atomicAdd(p.dbg + 8, c_bar);
        atomicAdd(p.dbg + 2, c_e1);
        atomicAdd(p.dbg + 9, c_e2);
    }
#endif
}
