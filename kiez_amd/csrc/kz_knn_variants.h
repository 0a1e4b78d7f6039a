// Experimental variants of the fused kernel (opt-in through kz_ctx_set_option("kernel_variant", n)); all produce the
// same candidate lists as kz_knn_cand_kernel and pass the same parity tests.  DESIGN.md section 7 records what each one
// was built to test and what it measured on C1:
//   1  barrier-free, A fragments straight from L1/L2            84 TF
//   2  32-k macro slices (half the workgroup barriers)          106 TF
//   3  LDS ring + per-wave progress words, no s_barrier          122 TF
//   4  stationary query tile in registers + LDS-DMA staging     120 TF
// (shipped default, variant 0: 122 TF)
#pragma once
#include "kz_knn_device.h"

// ---------------------------------------------------------------------------------------------------
// Variant 2: the LDS-staged kernel with 32-k macro slices (two 16-k slices per workgroup barrier).
// The barrier ablation priced the per-slice barrier at ~16 % (waves of a workgroup drift by the data-dependent
// epilogue and by SIMD arbitration); twice the MFMA work between barriers halves their number.  Needs an even
// number of 16-k slices per tile (d_pad % 32 == 0); other shapes use the 16-k kernel.
// LDS: 2 x 16 KiB index macro slices + bias + the candidate log.
// ---------------------------------------------------------------------------------------------------
constexpr int KZ_CAND2_LDS_BASE = 32768 + 1024;
constexpr int KZ_CAND2_LDS = KZ_CAND2_LDS_BASE + KZ_LOG_CAP * 256 * 8;

template <int KP>
__global__ __launch_bounds__(256, 2) void kz_knn_cand2_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);  // 2 x 4096 floats, then 2 x 128 bias floats
    float* bbuf = ybuf + 8192;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int NM = p.kg >> 3;  // 32-k macro slices per tile
    const int total = (t_end - t_begin) * NM;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * wave + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_CAND2_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_CAND2_LDS_BASE + KZ_LOG_CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    const float4* ysrc = reinterpret_cast<const float4*>(p.ypack + ((int64_t)t_begin * p.kg) * 512);  // 1024 float4 per macro slice
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * p.kg + h) * 512 + (32 * wave + j) * 4;
    auto load_q4 = [&](float4 (&q)[4], int m) {  // fragments of macro slice m: k-groups 8m + 2u + h, u = 0..3
        const float* src = qbase + (int64_t)m * 8 * 512;
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = kz_nt_load4(reinterpret_cast<const float4*>(src + u * 1024));
    };
    {
        float4* nb = reinterpret_cast<float4*>(ybuf);
#pragma unroll
        for (int c = 0; c < 4; ++c) nb[tid + 256 * c] = ysrc[tid + 256 * c];
        bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    }
    float4 qb[4];
    load_q4(qb, 0);
    __syncthreads();

    int g = 0;
    f32x16 acc[4];
    const float* bias_n = p.ybias + (tid & 127);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_sched_barrier(0);
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
        int m = 0;
        do {
            // prefetch the next macro slice (unconditional; the clamp re-reads the last one at the very end)
            const int gn = min(g + 1, total - 1);
            const float4* src = ysrc + (int64_t)gn * 1024;
            float4 ya[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) ya[c] = src[tid + 256 * c];
            const int tile_n = min(tile + 1, p.n_ytiles - 1);
            const float bn = bias_n[(int64_t)tile_n * KZ_TILE];
            float4 qn[4];
            load_q4(qn, (m + 1 == NM) ? 0 : m + 1);
            __builtin_amdgcn_sched_barrier(0);
            const float* buf = ybuf + (g & 1) * 4096;
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // k-groups 2u + h of the macro slice
                float4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    a[mt] = *reinterpret_cast<const float4*>(buf + ((2 * u + h) * KZ_TILE + 32 * mt + j) * 4);
                const float4 bq = qb[u];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float4* nb = reinterpret_cast<float4*>(ybuf + ((g + 1) & 1) * 4096);
#pragma unroll
                for (int c = 0; c < 4; ++c) nb[tid + 256 * c] = ya[c];
                bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
#pragma unroll
                for (int u = 0; u < 4; ++u) qb[u] = qn[u];
            }
            __syncthreads();
            ++g;
        } while (++m < NM);
        __builtin_amdgcn_sched_barrier(0);
        kz_tile_epilogue<KP>(acc, st, tile, tile == t_end - 1, h, (float)p.kg);
    }
}

// ---------------------------------------------------------------------------------------------------
// Variant 3: LDS ring with per-wave progress counters instead of workgroup barriers.
// Same tiling as kz_knn_cand_kernel, but the four waves of a workgroup are decoupled: the index slices go through a
// ring of 4 LDS buffers, slice g+2 is staged during slice g, and a wave may start slice g as soon as EVERY wave has
// completed slice g-2 (then all quarters of slice g are in LDS and nobody still reads the buffer that slice g+2
// overwrites).  Progress is one LDS word per wave, written after the wave's own LDS traffic of the slice has
// retired (LDS executes a wave's operations in order).  No s_barrier in the sweep; spins are bounded and a time-out
// raises an error on the host.  (Barrier ablation: 139 vs 122 TF on C1.)
// LDS: 4 x 8 KiB ring + bias rows + progress words + an 8-entry candidate log per lane.
// ---------------------------------------------------------------------------------------------------
constexpr int KZ_RING_CAP = 8;
constexpr int KZ_RING_BIAS = 32768;                 // byte offset of the 2 x 128 bias floats
constexpr int KZ_RING_PROG = KZ_RING_BIAS + 1024;   // 4 progress words (+ padding)
constexpr int KZ_RING_LOG = KZ_RING_PROG + 64;
constexpr int KZ_RING_LDS = KZ_RING_LOG + KZ_RING_CAP * 256 * 8;

template <int KP>
__global__ __launch_bounds__(256, 3) void kz_knn_cand_ring_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);
    float* bbuf = reinterpret_cast<float*>(smem + KZ_RING_BIAS);
    int* prog = reinterpret_cast<int*>(smem + KZ_RING_PROG);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int NS = p.kg >> 2;  // >= 4 (host)
    const int total = (t_end - t_begin) * NS;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_RING_LOG) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_RING_LOG + KZ_RING_CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    // Circular sweep: workgroup b starts a third of the range further than b-1, so workgroups that are co-resident
    // on one CU stream DIFFERENT index tiles at any time (no same-line pending stalls in the CU's L1); tiles are
    // visited in the order t_begin + (i + off) % nt.  The candidate logic is order-independent.
    const int nt = t_end - t_begin;
    const int off = p.phase_tiles > 0 ? (int)(((int64_t)(blockIdx.x % 3) * nt) / 3) : 0;
    const float4* ysrc = reinterpret_cast<const float4*>(p.ypack + ((int64_t)t_begin * NS) * 2048);
    auto slice_src = [&](int gi) {  // global slice counter -> address of that slice under the circular tile order
        if (off == 0) return ysrc + (int64_t)gi * 512;
        const int ti = gi / NS;
        const int sli = gi - ti * NS;
        int tp = ti + off;
        if (tp >= nt) tp -= nt;
        return ysrc + ((int64_t)tp * NS + sli) * 512;
    };
    auto tile_of = [&](int ti) {
        int tp = ti + off;
        if (tp >= nt) tp -= nt;
        return t_begin + tp;
    };
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * p.kg) * 512 + (32 * (tid >> 6) + j) * 4;
    // prologue: slices 0 and 1, bias rows of the first tile, progress words; ONE workgroup barrier
    {
        float4* nb = reinterpret_cast<float4*>(ybuf);
        const float4* s0 = slice_src(0);
        nb[tid] = s0[tid];
        nb[tid + 256] = s0[256 + tid];
        const float4* s1 = slice_src(min(1, total - 1));
        nb[512 + tid] = s1[tid];
        nb[512 + tid + 256] = s1[256 + tid];
        bbuf[(tid & 127)] = p.ybias[(int64_t)tile_of(0) * KZ_TILE + (tid & 127)];
        if (tid < 4) prog[tid] = 0;
    }
    // index slices are loaded THREE slices ahead and parked in registers for one more slice before they go to LDS:
    // the load -> LDS-write distance is two slices of MFMA time instead of one
    float4 yp0, yp1;   // slice g+2 (loaded during slice g-1), written to LDS at the end of slice g
    {
        const float4* s2 = slice_src(min(2, total - 1));
        yp0 = s2[tid];
        yp1 = s2[256 + tid];
    }
    // query fragments: current slice (qb) and the next one (qn) in registers, the one after that in flight
    float4 qb0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (0 + h) * 512));
    float4 qb1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (2 + h) * 512));
    float4 qn0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 + h) * 512));   // slice 1 (NS >= 4)
    float4 qn1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (6 + h) * 512));
    __syncthreads();

    int g = 0;
    f32x16 acc[4];
    const float* bias_n = p.ybias + (tid & 127);
    for (int ti = 0; ti < nt; ++ti) {
        const int tile = tile_of(ti);
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* bp = bbuf + (ti & 1) * 128 + 4 * h;
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
        // bias rows of the next tile: ONE load per tile (parked in LDS during the first slice, see (d))
        const float bn = bias_n[(int64_t)tile_of(min(ti + 1, nt - 1)) * KZ_TILE];
        int sl = 0;
        do {
            // (a) every wave must have completed slice g-2
            if (g >= 2) {
                int spins = 0;
                for (;;) {
                    const int p0 = __hip_atomic_load(prog + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int p1 = __hip_atomic_load(prog + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int p2 = __hip_atomic_load(prog + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int p3 = __hip_atomic_load(prog + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int pm = min(min(p0, p1), min(p2, p3));
                    if (__builtin_amdgcn_readfirstlane(pm) >= g - 1) break;
                    if (++spins > (1 << 22)) {  // ~seconds: give up loudly instead of hanging the GPU
                        if (lane == 0) atomicOr(p.err, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");  // compiler-only ordering: LDS itself is in order per wave, no caches
            }
            // (b) prefetch: index slice g+2 (-> registers), bias rows of the next tile, query fragments of slice g+1
            const int gn = min(g + 3, total - 1);
            const float4* src = slice_src(gn);
            const float4 ya0 = src[tid];     // slice g+3: written to LDS at the end of slice g+1
            const float4 ya1 = src[256 + tid];
            const int sl2 = (sl + 2 >= NS) ? sl + 2 - NS : sl + 2;  // query fragments TWO slices ahead
            const float4 qm0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sl2 + h) * 512));
            const float4 qm1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sl2 + 2 + h) * 512));
            __builtin_amdgcn_sched_barrier(0);
            // (c) 32 MFMAs out of ring buffer g % 4
            const float* buf = ybuf + (g & 3) * 2048;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    a[mt] = *reinterpret_cast<const float4*>(buf + ((2 * t + h) * KZ_TILE + 32 * mt + j) * 4);
                const float4 bq = t ? qb1 : qb0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (d) stage slice g+2 into ring buffer (g+2) % 4
            {
                float4* nb = reinterpret_cast<float4*>(ybuf + ((g + 2) & 3) * 2048);
                nb[tid] = yp0;               // slice g+2, loaded one slice ago
                nb[tid + 256] = yp1;
                yp0 = ya0;
                yp1 = ya1;
                if (sl == 0) bbuf[((ti + 1) & 1) * 128 + (tid & 127)] = bn;  // every wave passes slice 0 before any starts tile ti+1
                qb0 = qn0;
                qb1 = qn1;
                qn0 = qm0;
                qn1 = qm1;
            }
            // (e) publish: this wave has completed slice g (its reads of buffer g%4 and its writes are retired)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS traffic retired; do NOT drain vmcnt (query loads in flight)
            if (lane == 0) __hip_atomic_store(prog + wave, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++g;
        } while (++sl < NS);
        __builtin_amdgcn_sched_barrier(0);
        kz_tile_epilogue<KP, KZ_RING_CAP>(acc, st, tile, ti == nt - 1, h, (float)p.kg);
    }
}

// ---------------------------------------------------------------------------------------------------
// Variant 4: stationary query tile.  For d_pad == 16*NSR (NSR = 4 or 8, i.e. d <= 64 / d <= 128) the query fragments of
// all slices stay in registers for the whole sweep (64 VGPRs at NSR = 8), so the per-slice query-fragment loads of the
// streaming kernel disappear (diagnostic build without them: 149 vs 122 TF on C1).  To keep three waves per SIMD the
// index slices are staged with LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write); the packed image is
// copied linearly, which is exactly the lane-linear layout LDS-DMA writes.  One workgroup barrier per slice; its
// fence also retires the DMA of the next slice.
// ---------------------------------------------------------------------------------------------------
template <int KP, int NSR>
__global__ __launch_bounds__(256, 3) void kz_knn_cand_res_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ybuf = reinterpret_cast<float*>(smem);            // 2 x 2048 floats
    float* bbuf = ybuf + 4096;                                // 2 x 128 bias floats
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int total = (t_end - t_begin) * NSR;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * (tid >> 6) + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KZ_CAND_LDS_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_CAND_LDS_BASE + KZ_LOG_CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    // LDS-DMA: lane l of wave w copies 16 B from gsrc + (64*(w + 4c) + l)*16 to LDS slice + (64*(w + 4c))*16 + l*16, c = 0, 1
    const float* ysrc = p.ypack + ((int64_t)t_begin * NSR) * 2048 + tid * 4;   // per-lane source of slice 0, chunk 0
    auto dma_slice = [&](int gi, int buf) {
        const float* src = ysrc + (int64_t)gi * 2048;
        float* dst = ybuf + buf * 2048 + wave * 256;  // wave-uniform LDS base (floats)
        kz_glds16(src, dst);
        kz_glds16(src + 1024, dst + 1024);
    };
    // prologue: slice 0 by DMA, bias rows of the first tile, resident query fragments
    dma_slice(0, 0);
    bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * (4 * NSR) + h) * 512 + (32 * (tid >> 6) + j) * 4;
    float4 qres[NSR][2];
#pragma unroll
    for (int u = 0; u < NSR; ++u) {
        qres[u][0] = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * u) * 512));
        qres[u][1] = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * u + 2) * 512));
    }
    __syncthreads();

    int g = 0;
    f32x16 acc[4];
    for (int tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_sched_barrier(0);
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
        // bias rows of the next tile (one 4-byte load per thread per tile; parked in LDS after slice 0)
        const float bn = p.ybias[(int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE + (tid & 127)];
#pragma unroll
        for (int u = 0; u < NSR; ++u) {
            dma_slice(min(g + 1, total - 1), (g + 1) & 1);   // next slice lands in the other buffer while we compute
            __builtin_amdgcn_sched_barrier(0);
            const float* buf = ybuf + (g & 1) * 2048;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float4 a[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    a[mt] = *reinterpret_cast<const float4*>(buf + ((2 * t + h) * KZ_TILE + 32 * mt + j) * 4);
                const float4 bq = qres[u][t];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);  // keep only one t-group of A fragments live (register budget)
            }
            if (u == 0) bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
            __syncthreads();   // fence drains vmcnt: the DMA of slice g+1 has landed; everyone is done with buffer g&1
            ++g;
        }
        __builtin_amdgcn_sched_barrier(0);
        kz_tile_epilogue<KP>(acc, st, tile, tile == t_end - 1, h, (float)p.kg);
    }
}

// ---------------------------------------------------------------------------------------------------
// Barrier-free variant: every wave feeds its MFMAs straight from L1/L2.
// The LDS-staged kernel above shares one index slice among its 4 waves and pays one workgroup barrier per slice;
// removing only those barriers (diagnostic build) raised C1 from 119 to 139 TF, i.e. the waves of a workgroup drift
// on their SIMDs and the barrier stalls cost ~16 %.  Here each lane loads its own A fragments from the packed image
// (512-B coalesced segments, the 4 waves of a workgroup and the co-resident workgroups hit the same lines in L1/L2),
// double-buffered one half-slice (16 MFMAs) ahead, so waves never wait for each other.  LDS only holds the
// candidate logs and a per-wave copy of the tile's bias rows.
// ---------------------------------------------------------------------------------------------------
constexpr int KZ_DIRECT_LDS = KZ_LOG_CAP * 256 * 8 + 4 * 2 * 128 * 4;

template <int KP>
__global__ __launch_bounds__(256, 3) void kz_knn_cand_direct_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int HS = p.kg >> 1;  // half-slices (8 k each side of the lane halves = 16 MFMAs) per tile
    const int total = (t_end - t_begin) * HS;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * wave + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem) + tid;
    st.si = reinterpret_cast<int*>(smem + KZ_LOG_CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    float* bbuf = reinterpret_cast<float*>(smem + KZ_LOG_CAP * 256 * 8) + wave * 256;  // this wave's 2 x 128 bias floats
    // Linear stream of half-slices: G = (tile - t_begin) * HS + hs; lane (j, h) reads k-group 2*hs + h of rows 32*mt + j:
    //   address(G, mt) = ybase + G*1024 + h*512 + (32*mt + j)*4   (floats)
    const float* ybase = p.ypack + ((int64_t)t_begin * p.kg) * 512 + h * 512 + j * 4;
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * p.kg + h) * 512 + (32 * wave + j) * 4;
    auto load_a = [&](float4 (&a)[4], int G) {
        const float* src = ybase + (int64_t)G * 1024;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const float4*>(src + mt * 128);
    };
    auto load_q = [&](int hs) { return kz_nt_load4(reinterpret_cast<const float4*>(qbase + (int64_t)hs * 1024)); };
    auto mfma16 = [&](f32x16 (&acc)[4], const float4 (&a)[4], const float4& bq) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].x, bq.x, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].y, bq.y, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].z, bq.z, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt].w, bq.w, acc[mt], 0, 0, 0);
    };

    // prologue: fragments of half-slice 0, bias rows of the first tile (each wave keeps its own copy: no barriers)
    float4 a0[4], a1[4];
    load_a(a0, 0);
    float4 q0 = load_q(0), q1;
    {
        const float* bsrc = p.ybias + (int64_t)t_begin * KZ_TILE;
        bbuf[(t_begin & 1) * 128 + lane] = bsrc[lane];
        bbuf[(t_begin & 1) * 128 + 64 + lane] = bsrc[64 + lane];
    }
    kz_wave_sync();

    int G = 0;
    f32x16 acc[4];
    for (int tile = t_begin; tile < t_end; ++tile) {
        __builtin_amdgcn_sched_barrier(0);
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
        // bias rows of the next tile: loaded now, parked in LDS after the slices
        const int tile_n = min(tile + 1, p.n_ytiles - 1);
        const float bn0 = p.ybias[(int64_t)tile_n * KZ_TILE + lane];
        const float bn1 = p.ybias[(int64_t)tile_n * KZ_TILE + 64 + lane];
        int hs = 0;
        do {  // two half-slices per trip so that the fragment buffers alternate without register moves (HS is even)
            {
                const int Gn = min(G + 1, total - 1);
                load_a(a1, Gn);
                q1 = load_q(hs + 1);  // hs + 1 < HS always (hs even)
                __builtin_amdgcn_sched_barrier(0);
                mfma16(acc, a0, q0);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const int Gn = min(G + 2, total - 1);
                load_a(a0, Gn);
                q0 = load_q(hs + 2 == HS ? 0 : hs + 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma16(acc, a1, q1);
                __builtin_amdgcn_sched_barrier(0);
            }
            G += 2;
            hs += 2;
        } while (hs < HS);
        bbuf[((tile + 1) & 1) * 128 + lane] = bn0;
        bbuf[((tile + 1) & 1) * 128 + 64 + lane] = bn1;
        kz_tile_epilogue<KP>(acc, st, tile, tile == t_end - 1, h, (float)p.kg);
        kz_wave_sync();  // bias rows visible to this wave's own lanes before the next tile's init
    }
}


// ---------------------------------------------------------------------------------------------------
// Variant 5: interleaved instruction stream.  Same data flow as the shipped kernel (index slices staged through LDS,
// query fragments streamed one slice ahead, one workgroup barrier per slice), but the per-slice global loads, the
// LDS reads of the second k-half and the LDS refill are placed BETWEEN groups of four MFMAs instead of in front of /
// behind the 32-MFMA block.  A wave issues in order: a vector-memory instruction that has to wait for a slot in the
// CU's address path holds back everything behind it, so in the shipped stream [5 loads][4 ds_read][wait][32 MFMA] a
// busy memory path delays the wave's MFMAs; here each memory instruction issues in the shadow of the MFMAs that were
// issued just before it.  PMC: the shipped kernel keeps the MFMA pipe 78.5 % busy, the same build without its global
// loads 89.7 % at the same clock (tools/pmc_clock.sh).
// NB = 2: two slice buffers; the first k-half's fragments are read right after the barrier.
// NB = 3: three slice buffers (log capacity 12 instead of 16 to stay at three workgroups per CU); slices are staged
//         two ahead, so the next slice's first fragments are read before the barrier, under this slice's MFMAs.
// ---------------------------------------------------------------------------------------------------
template <int NB>
struct KzIl {
    static constexpr int CAP = NB == 3 ? 12 : 16;
    static constexpr int LOG_BASE = NB * 8192 + 1024;
    static constexpr int LDS = LOG_BASE + CAP * 256 * 8;
};

#define KZ_MFMA4(A, C, B)                                                                                     \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) acc[mt] =                                                \
        __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt].C, B.C, acc[mt], 0, 0, 0)
#define KZ_SB() __builtin_amdgcn_sched_barrier(0)

template <int KP, int NB>
__global__ __launch_bounds__(256, 3) void kz_knn_cand_il_kernel(KnnCandParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CAP = KzIl<NB>::CAP;
    float* ybuf = reinterpret_cast<float*>(smem);   // NB x 2048 floats
    float* bbuf = ybuf + NB * 2048;                  // 2 x 128 bias floats
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int h = lane >> 5;
    const int4 wd = p.work[blockIdx.x];
    const int qt = wd.x, t_begin = wd.y, t_end = wd.z, s = wd.w;
    const int NS = p.kg >> 2;
    const int total = (t_end - t_begin) * NS;

    const int64_t listoff = kz_list_wave_base((int64_t)qt * KZ_TILE + 32 * wave + j, p.lay, KP, s) + h * 32 + j;
    KzCandState st;
    st.lk = p.out_key + listoff;
    st.li = p.out_idx + listoff;
    st.sk = reinterpret_cast<float*>(smem + KzIl<NB>::LOG_BASE) + tid;
    st.si = reinterpret_cast<int*>(smem + KzIl<NB>::LOG_BASE + CAP * 256 * 4) + tid;
#pragma unroll 4
    for (int e = 0; e < KP; ++e) {
        st.lk[e * KZ_LSTRIDE] = -INFINITY;
        st.li[e * KZ_LSTRIDE] = -1;
    }
    st.tau = -INFINITY;
    st.minpos = 0;
    st.cnt = 0;
    st.tiles_done = 0;
    st.next_merge = 1;
    if (total <= 0) return;

    const float4* ysrc = reinterpret_cast<const float4*>(p.ypack + ((int64_t)t_begin * NS) * 2048);
    const float* qbase = p.qpack + ((int64_t)(p.qt0 + qt) * p.kg) * 512 + (32 * wave + j) * 4;
    // prologue: the first NB-1 slices and the bias rows of the first tile
    {
        float4* nb = reinterpret_cast<float4*>(ybuf);
#pragma unroll
        for (int u = 0; u < NB - 1; ++u) {
            const float4* s0 = ysrc + (int64_t)min(u, total - 1) * 512;
            nb[u * 512 + tid] = s0[tid];
            nb[u * 512 + 256 + tid] = s0[256 + tid];
        }
        bbuf[(t_begin & 1) * 128 + (tid & 127)] = p.ybias[(int64_t)t_begin * KZ_TILE + (tid & 127)];
    }
    // query fragments: qb = current slice, qn = next slice (loaded early in the slice, copied at its end)
    float4 qb0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (0 + h) * 512));
    float4 qb1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (2 + h) * 512));
    __syncthreads();

    // fragment address of this lane inside a slice buffer: k-half t -> k-group 2t + h, rows 32 mt + j
    const float* fbase = ybuf + (h * KZ_TILE + j) * 4;
    auto frag = [&](float4 (&a)[4], const int b, const int t) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            a[mt] = *reinterpret_cast<const float4*>(fbase + b * 2048 + (2 * t * KZ_TILE + 32 * mt) * 4);
    };
    float4 a0[4], a1[4];
    int g = 0;
    int bcur = 0;  // buffer holding slice g
    if (NB == 3) frag(a0, 0, 0);
    f32x16 acc[4];
    const float* bias_n = p.ybias + (tid & 127);
    for (int tile = t_begin; tile < t_end; ++tile) {
        KZ_SB();
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
        // bias rows of the next tile: one load per tile, parked in LDS by every slice's refill (same value)
        const float bn = bias_n[(int64_t)min(tile + 1, p.n_ytiles - 1) * KZ_TILE];
        KZ_SB();
        int sl = 0;
        do {
            const int sln = (sl + 1 == NS) ? 0 : sl + 1;
            const int gn = min(g + NB - 1, total - 1);
            const float4* src = ysrc + (int64_t)gn * 512;
            const int bnext = bcur + 1 == NB ? 0 : bcur + 1;                        // buffer of slice g+1
            const int bfill = NB == 2 ? bnext : (bnext + 1 == NB ? 0 : bnext + 1);  // buffer of slice g+NB-1
            if (NB == 2) frag(a0, bcur, 0);
            KZ_SB();
            KZ_MFMA4(a0, x, qb0);
            KZ_SB();
            const float4 ya0 = src[tid];
            const float4 ya1 = src[256 + tid];
            KZ_SB();
            KZ_MFMA4(a0, y, qb0);
            KZ_SB();
            const float4 qn0 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sln + h) * 512));
            const float4 qn1 = kz_nt_load4(reinterpret_cast<const float4*>(qbase + (4 * sln + 2 + h) * 512));
            KZ_SB();
            KZ_MFMA4(a0, z, qb0);
            KZ_SB();
            frag(a1, bcur, 1);
            KZ_SB();
            KZ_MFMA4(a0, w, qb0);
            KZ_SB();
            KZ_MFMA4(a1, x, qb1);
            KZ_SB();
            if (NB == 3) frag(a0, bnext, 0);   // slice g+1 became visible at the previous barrier
            KZ_SB();
            KZ_MFMA4(a1, y, qb1);
            KZ_SB();
            KZ_MFMA4(a1, z, qb1);
            KZ_SB();
            {
                float4* nb = reinterpret_cast<float4*>(ybuf + bfill * 2048);
                nb[tid] = ya0;
                nb[tid + 256] = ya1;
                bbuf[((tile + 1) & 1) * 128 + (tid & 127)] = bn;
            }
            KZ_SB();
            KZ_MFMA4(a1, w, qb1);
            KZ_SB();
            // keep the current fragments live up to here: the next ones then get registers of their own and the
            // copies (and the wait for the loads behind them) stay at the end of the slice
            asm volatile("" ::"v"(qb0.x), "v"(qb0.y), "v"(qb0.z), "v"(qb0.w), "v"(qb1.x), "v"(qb1.y), "v"(qb1.z), "v"(qb1.w));
            qb0 = qn0;
            qb1 = qn1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            bcur = bnext;
            ++g;
        } while (++sl < NS);
        KZ_SB();
        kz_tile_epilogue<KP, CAP>(acc, st, tile, tile == t_end - 1, h, (float)p.kg);
    }
}
