// kz_matrix: HBM-resident embedding matrix = exact rows + float64 norms + MFMA operand images (built on first use).
//
// kz_matrix_create only touches the rows once: float64 norms, the accumulator-init row (bias) and the maxima the
// certification needs.  The operand images of the fused distance kernel are built by the first kz_knn call that needs
// them (kz_matrix_image_*), all with the same geometry (DESIGN.md "Data layout in HBM"): tile = 128 rows, slice = 16 k,
// d padded with zeros to a multiple of 16, rows padded to a multiple of 128; one slice of one tile is ONE contiguous
// block and consecutive slices (also across tiles) are consecutive in memory, so the kernel's stream is linear and
// every wave-load is 1 KiB coalesced.  Inside a block the bytes are exactly the conflict-free ds_read_b128 image of the
// MFMA A/B operand, which is why the kernels copy them to LDS with a linear LDS-DMA.
//
//   fp16   (first pass, kz_knn_h16.h)   packed_h[tile][slice][plane][row][8]      4 KiB per slice
//          plane p = k 8p..8p+7: the 16-byte fragment of lane-half p of v_mfma_f32_32x32x16_f16 (lane l: row l&31,
//          k 8(l>>5)+j).  Values are x_h = fp16(S (x - mu)): centred with the pair's common shift mu and scaled by a
//          power of two S (kz_center); per row the image keeps |x_c|^2, |x_h| and the MEASURED residual |x_c - x_h|.
//   bf16x2 (second tier, kz_knn_bf16.h) packed_bf[tile][slice][plane][row][8]     8 KiB per slice
//          planes hi(k 0-7), hi(k 8-15), lo(k 0-7), lo(k 8-15) with hi = bf16(x), lo = bf16(x - hi).
//   fp32   (third tier, kz_knn.hip)     packed[tile][kgroup][row][4]              8 KiB per slice (4 kgroups)
#include "kz_common.h"

// round-to-nearest-even float32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned short kz_bf16_rn(float f) {
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float kz_bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

// operand value of element k of a row: the raw value (euclidean family) or the row-normalised one (cosine), in float64
template <typename T>
__device__ __forceinline__ double kz_operand(const T* x, int k, int d, int metric, double nrm) {
    if (k >= d) return 0.0;
    return metric == KZ_COSINE ? (double)x[k] / nrm : (double)x[k];
}

#ifndef KZ_NORM_ROWS
#define KZ_NORM_ROWS 2   // (1M x 200 float32: 1 row 247 us, 2 rows 238, 4 rows 285, 8 rows 376)
#endif
#ifndef KZ_ROWGROUP_MAX_BLOCKS
#define KZ_ROWGROUP_MAX_BLOCKS 4096
#endif

// ---- create: norms, bias, maxima, finiteness ------------------------------------------------------------------
// stats[0] = max row norm, stats[1] = max |operand element| (bits of non-negative doubles order as integers)
template <typename T>
__global__ __launch_bounds__(256) void kz_norms_kernel(const T* __restrict__ raw, int64_t n, int d, int metric, int64_t n_pad,
                                                       float* __restrict__ bias, double* __restrict__ sqn,
                                                       unsigned long long* __restrict__ stats, int* __restrict__ bad_flag) {
    __shared__ double s_max[4], s_abs[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double wmax = 0.0, amax = 0.0;
    const bool vec = kz_row_vec_ok(raw, d);
    // KZ_NORM_ROWS consecutive rows per wave and iteration, their loads issued together (one row at a time left one memory
    // round trip in flight per wave: 1M x 200 float32 took 490 us, 1.6 TB/s).  Per row the arithmetic IS kz_wave_dot(x, x):
    // the same lane owns the same elements, the same fma chain, the same butterfly -- sqn keeps its bits.
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * KZ_NORM_ROWS; row0 < n_pad; row0 += (int64_t)gridDim.x * 4 * KZ_NORM_ROWS) {
        double acc[KZ_NORM_ROWS], am[KZ_NORM_ROWS];
#pragma unroll
        for (int r = 0; r < KZ_NORM_ROWS; ++r) acc[r] = am[r] = 0.0;
        for (int k0 = 4 * lane; k0 < d; k0 += 256) {
            double x[KZ_NORM_ROWS][4];
#pragma unroll
            for (int r = 0; r < KZ_NORM_ROWS; ++r) {
                if (row0 + r < n) {
                    kz_row4(raw + (row0 + r) * (int64_t)d, k0, d, vec, x[r]);
                } else {
                    x[r][0] = x[r][1] = x[r][2] = x[r][3] = 0.0;
                }
            }
#pragma unroll
            for (int r = 0; r < KZ_NORM_ROWS; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[r] = fma(x[r][u], x[r][u], acc[r]);
                    const double a = fabs(x[r][u]);
                    if (a <= 1e300) am[r] = fmax(am[r], a);
                }
        }
        // (n_pad is a multiple of 128: a group never straddles it.)  Float64 sqrt and division are long instruction sequences the
        // whole wave executes: the euclidean family takes ONE sqrt per workgroup (of the largest squared norm -- sqrt is
        // monotone), cosine one sqrt and one division per GROUP (lane r works for row r of the group).
        double sq[KZ_NORM_ROWS];
#pragma unroll
        for (int r = 0; r < KZ_NORM_ROWS; ++r) {
            sq[r] = row0 + r < n ? kz_wave_sum(acc[r]) : 0.0;
            if (!(sq[r] <= 1e30)) {  // NaN, inf, or too large for the float32 operand image
                if (lane == 0) atomicOr(bad_flag, 1);
            }
        }
        const int sel = lane & (KZ_NORM_ROWS - 1);
        double mine = sq[0];
#pragma unroll
        for (int r = 1; r < KZ_NORM_ROWS; ++r) mine = sel == r ? sq[r] : mine;
        const int64_t my_row = row0 + sel;
        if (metric == KZ_COSINE) {
            double a_mine = 0.0;
#pragma unroll
            for (int r = 0; r < KZ_NORM_ROWS; ++r) {
                double a = am[r];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) a = fmax(a, __shfl_xor(a, off, 64));
                a_mine = sel == r ? a : a_mine;
            }
            double nrm = sqrt(mine);
            if (nrm == 0.0) nrm = 1.0;  // sklearn normalize(): zero rows stay zero
            // max |operand element| of the row: |x| / nrm is monotone in |x|, so the largest |x| divided once is the largest quotient
            const double v = a_mine / nrm;
            if (my_row < n) {
                if (v <= 1e300) amax = fmax(amax, v);
                wmax = 1.0;
                if (lane < KZ_NORM_ROWS) {
                    sqn[my_row] = nrm;
                    bias[my_row] = 0.0f;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < KZ_NORM_ROWS; ++r) {
                wmax = fmax(wmax, sq[r]);      // (squared; the root is taken once, below)
                amax = fmax(amax, am[r]);
            }
            if (lane < KZ_NORM_ROWS && my_row < n) {
                sqn[my_row] = mine;
                bias[my_row] = (float)(-0.5 * mine);
            }
        }
        if (lane < KZ_NORM_ROWS && my_row >= n) bias[my_row] = -INFINITY;   // rows past the end
    }
    if (metric != KZ_COSINE) wmax = sqrt(wmax);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) {
        s_max[wave] = wmax;
        s_abs[wave] = amax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // thousands of workgroups, two addresses: read first (an ordinary load; a stale value only costs a redundant atomic) and
        // pay for the atomic only when this workgroup raises the maximum -- same-address atomics serialise in the L2
        // (100k x 128: the kernel took 108 us for 51 MB, most of it here)
        double m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (m <= 1e300 && bits > *(const volatile unsigned long long*)stats) atomicMax(stats, bits);
        m = fmax(fmax(s_abs[0], s_abs[1]), fmax(s_abs[2], s_abs[3]));
        bits = (unsigned long long)__double_as_longlong(m);
        if (bits > *(const volatile unsigned long long*)(stats + 1)) atomicMax(stats + 1, bits);
    }
}

// ---- float32 and split-bf16 images (lower tiers) ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kz_pack_f32_kernel(const T* __restrict__ raw, const double* __restrict__ sqn, int64_t n,
                                                          int d, int metric, int kg, int64_t n_pad, float* __restrict__ packed) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int d_pad = kg * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_pad; row += (int64_t)gridDim.x * 4) {
        const int64_t tile = row >> 7;
        const int r = (int)(row & 127);
        float* dst = packed + (tile * kg) * (int64_t)(KZ_TILE * 4) + r * 4;
        const T* x = raw + row * (int64_t)d;
        const double nrm = (row < n && metric == KZ_COSINE) ? sqn[row] : 1.0;
        for (int k = lane; k < d_pad; k += 64) {
            const double vd = row < n ? kz_operand(x, k, d, metric, nrm) : 0.0;
            dst[(int64_t)(k >> 2) * (KZ_TILE * 4) + (k & 3)] = (float)vd;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void kz_pack_bf_kernel(const T* __restrict__ raw, const double* __restrict__ sqn, int64_t n,
                                                         int d, int metric, int kg_bf, int64_t n_pad,
                                                         unsigned short* __restrict__ packed_bf) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int d_pad = kg_bf * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_pad; row += (int64_t)gridDim.x * 4) {
        const int64_t tile = row >> 7;
        const int r = (int)(row & 127);
        // element k of this row -> slice k/16, plane (k/8)&1 (+2 for lo), 8 values per row and plane
        unsigned short* dbf = packed_bf + (tile * kg_bf) * (int64_t)(KZ_TILE * 8) + r * 8;
        const T* x = raw + row * (int64_t)d;
        const double nrm = (row < n && metric == KZ_COSINE) ? sqn[row] : 1.0;
        for (int k = lane; k < d_pad; k += 64) {
            const double vd = row < n ? kz_operand(x, k, d, metric, nrm) : 0.0;
            const unsigned short hi = kz_bf16_rn((float)vd);
            const unsigned short lo = kz_bf16_rn((float)(vd - (double)kz_bf16_to_f32(hi)));
            const int64_t off = (int64_t)(k >> 4) * (4 * KZ_TILE * 8) + ((k >> 3) & 1) * (KZ_TILE * 8) + (k & 7);
            dbf[off] = hi;
            dbf[off + 2 * KZ_TILE * 8] = lo;
        }
    }
}

// ---- fp16 image: centre (two-stage deterministic column mean), scale, pack ---------------------------------------
constexpr int KZ_COLSUM_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(256) void kz_colsum_kernel(const T* __restrict__ raw, const double* __restrict__ sqn, int64_t n, int d,
                                                        int metric, double* __restrict__ partial) {
    // block b sums rows b, b + B, ...; thread t the columns t, t + 256, ...  (row-major rows: consecutive threads read
    // consecutive elements); four independent rows in flight per thread
    const int64_t B = gridDim.x;
    for (int k = threadIdx.x; k < d; k += 256) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int64_t row = blockIdx.x;
        for (; row + 3 * B < n; row += 4 * B) {
            const double n0 = metric == KZ_COSINE ? sqn[row] : 1.0, n1 = metric == KZ_COSINE ? sqn[row + B] : 1.0;
            const double n2 = metric == KZ_COSINE ? sqn[row + 2 * B] : 1.0, n3 = metric == KZ_COSINE ? sqn[row + 3 * B] : 1.0;
            a0 += (double)raw[row * (int64_t)d + k] / n0;
            a1 += (double)raw[(row + B) * (int64_t)d + k] / n1;
            a2 += (double)raw[(row + 2 * B) * (int64_t)d + k] / n2;
            a3 += (double)raw[(row + 3 * B) * (int64_t)d + k] / n3;
        }
        for (; row < n; row += B) a0 += (double)raw[row * (int64_t)d + k] / (metric == KZ_COSINE ? sqn[row] : 1.0);
        partial[(int64_t)blockIdx.x * d + k] = (a0 + a1) + (a2 + a3);
    }
}

// One block per column k: mu[k] = float32(mean_k), summed in a fixed order (the images, and with them which rows need a
// lower tier, are reproducible from run to run).  Block 0 also fixes the scale S = 2^(13 - ceil(log2(max |operand|))):
// S |x - mu| <= 2^14 for the matrix the centre was taken from, and a partner matrix with elements up to four times
// larger still fits the fp16 range (beyond that the pack kernel clamps, and the clamped part shows up in the measured
// residual, i.e. in the certification bound).
__global__ __launch_bounds__(256) void kz_center_finish_kernel(const double* __restrict__ partial, int blocks, int64_t n, int d,
                                                               const double* __restrict__ stats, float* __restrict__ mu,
                                                               double* __restrict__ scale) {
    __shared__ double red[256];
    const int k = blockIdx.x;
    double s = 0.0;
    if (k < d)
        for (int b = threadIdx.x; b < blocks; b += 256) s += partial[(int64_t)b * d + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        mu[k] = k < d ? (float)(red[0] / (double)n) : 0.0f;
        if (k == 0) {
            const double amax = stats[1];
            double S = 1.0;
            if (amax > 0.0 && amax < 1e300) {
                int e;
                frexp(amax, &e);        // amax = f 2^e, f in [0.5, 1): amax <= 2^e
                int se = 13 - e;
                if (se > 100) se = 100;    // S and S^2 |x|^2 stay far inside the float32 range
                if (se < -100) se = -100;
                S = ldexp(1.0, se);
            }
            scale[0] = S;
            scale[1] = 1.0 / (S * S);
        }
    }
}

// Four rows per wave (lane layout below).  v = float32(x - mu) is the centred operand (the "exact" vector of the certification: its distance
// to another centred row differs from the true distance only by float32 centring round-off, accounted for in the bound),
// x_h = fp16(S v) / S the operand the matrix pipe multiplies.  Values below the fp16 normal range are stored as zero
// (no dependence on the denormal mode of the matrix pipe), values beyond it are clamped; both end up in the residual.
template <typename T>
__global__ __launch_bounds__(256) void kz_pack_h_kernel(const T* __restrict__ raw, const double* __restrict__ sqn, int64_t n, int d,
                                                        int metric, int nsr, int64_t n_pad, const float* __restrict__ mu,
                                                        const double* __restrict__ scale, unsigned short* __restrict__ packed,
                                                        float* __restrict__ bias, double* __restrict__ rowq,
                                                        unsigned long long* __restrict__ dmax, const int* __restrict__ perm) {
    __shared__ double s_m[3][4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r4 = lane & 3, kb = lane >> 2;
    const int d_pad = nsr * 16;
    const double S = scale[0];
    const double invS = 1.0 / S;   // S is a power of two: x * invS == x / S, bit for bit, without the float64 division sequence
    const float Sf = (float)S;
    const bool vec = kz_row_vec_ok(raw, d);
    double m_h = 0.0, m_r = 0.0, m_c = 0.0;
    // Four consecutive rows per wave: lane 4 kb + r4 owns elements 128 c + 8 kb .. + 7 of row 4 g + r4 -- ONE 16-byte fragment of
    // the image per chunk c, read with two 16-byte loads and written with one 16-byte store; the four lanes of a kb write
    // four consecutive rows = 64 contiguous bytes.  (One row per wave with 2-byte stores: 1M x 200 took 790 us, 1.6 TB/s.)
    for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g * 4 < n_pad; g += (int64_t)gridDim.x * 4) {
        const int64_t row = g * 4 + r4;
        const int64_t tile = row >> 7;
        const int r = (int)(row & 127);
        _Float16* dst = reinterpret_cast<_Float16*>(packed) + (tile * nsr) * (int64_t)(2 * KZ_TILE * 8) + r * 8;
        const bool live = row < n;   // rows past the end: zero image, bias -inf
        // perm (dual pass, kz_knn_dual.h): image row `row` holds matrix row perm[row]; only the image and the bias are written
        const int64_t srow = !live ? 0 : (perm ? (int64_t)perm[row] : row);
        const T* x = raw + srow * (int64_t)d;
        const double nrm = (live && metric == KZ_COSINE) ? sqn[srow] : 1.0;
        double c2 = 0.0, h2 = 0.0, r2 = 0.0;
        for (int k0 = 8 * kb; k0 < d_pad; k0 += 128) {
            double xv[2][4];
            if (live) {
                kz_row4(x, k0, d, vec, xv[0]);       // (zero fill past d)
                kz_row4(x, k0 + 4, d, vec, xv[1]);
            } else {
                xv[0][0] = xv[0][1] = xv[0][2] = xv[0][3] = xv[1][0] = xv[1][1] = xv[1][2] = xv[1][3] = 0.0;
            }
            const float4 mu0 = *reinterpret_cast<const float4*>(mu + k0), mu1 = *reinterpret_cast<const float4*>(mu + k0 + 4);
            const float mk[8] = {mu0.x, mu0.y, mu0.z, mu0.w, mu1.x, mu1.y, mu1.z, mu1.w};
            union { _Float16 h[8]; uint4 q; } frag;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float v = 0.0f;
                if (live && k0 + u < d) {
                    const double op = metric == KZ_COSINE ? xv[u >> 2][u & 3] / nrm : xv[u >> 2][u & 3];   // (kz_operand)
                    v = (float)(op - (double)mk[u]);
                }
                float vs = v * Sf;                                       // exact: S is a power of two
                if (fabsf(vs) < 6.103515625e-05f) vs = 0.0f;             // below the fp16 normal range
                vs = fminf(fmaxf(vs, -65504.0f), 65504.0f);
                const _Float16 hv = (_Float16)vs;                        // round to nearest even
                frag.h[u] = hv;
                const double xh = (double)(float)hv * invS;
                const double res = (double)v - xh;
                c2 = fma((double)v, (double)v, c2);
                h2 = fma(xh, xh, h2);
                r2 = fma(res, res, r2);
            }
            *reinterpret_cast<uint4*>(dst + (int64_t)(k0 >> 4) * (2 * KZ_TILE * 8) + ((k0 >> 3) & 1) * (KZ_TILE * 8)) = frag.q;
        }
        // the row's sums: over the sixteen lanes of its r4 (lane bits 2..5); every lane of the row ends with the same value
#pragma unroll
        for (int off = 32; off >= 4; off >>= 1) {
            c2 += __shfl_xor(c2, off, 64);
            h2 += __shfl_xor(h2, off, 64);
            r2 += __shfl_xor(r2, off, 64);
        }
        // one float64 sqrt sequence for both norms of the four rows: the lanes of even kb take |x_h|, those of odd kb the residual
        const double root = sqrt((kb & 1) ? r2 : h2);
        const double nh = __shfl(root, r4, 64), nr = __shfl(root, 4 + r4, 64);
        if (kb == 0) {
            if (live) {
                if (rowq) {
                    rowq[row * 3 + 0] = c2;
                    rowq[row * 3 + 1] = nh;
                    rowq[row * 3 + 2] = nr;
                }
                bias[row] = (float)(-0.5 * c2 * S * S);
            } else {
                bias[row] = -INFINITY;
            }
        }
        m_h = fmax(m_h, h2);   // (squared: the roots are taken once, below; rows past the end contribute zeros)
        m_r = fmax(m_r, r2);
        m_c = fmax(m_c, c2);
    }
    m_h = sqrt(m_h);
    m_r = sqrt(m_r);
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {   // the four rows of the wave
        m_h = fmax(m_h, __shfl_xor(m_h, off, 64));
        m_r = fmax(m_r, __shfl_xor(m_r, off, 64));
        m_c = fmax(m_c, __shfl_xor(m_c, off, 64));
    }
    if (lane == 0) {
        s_m[0][wave] = m_h;
        s_m[1][wave] = m_r;
        s_m[2][wave] = m_c;
    }
    __syncthreads();
    if (threadIdx.x < 3 && dmax) {
        const double m = fmax(fmax(s_m[threadIdx.x][0], s_m[threadIdx.x][1]), fmax(s_m[threadIdx.x][2], s_m[threadIdx.x][3]));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (bits > *(const volatile unsigned long long*)(dmax + threadIdx.x)) atomicMax(dmax + threadIdx.x, bits);   // (see kz_norms_kernel)
    }
}

static int kz_pack_blocks(int64_t n_pad) {
    const int64_t b = (n_pad + 3) / 4;
    return (int)(b < 4096 ? b : 4096);   // (32768: no faster -- 100k x 128: norms 46 -> 68 us, pack 62 -> 64 us)
}
// kz_norms_kernel / kz_pack_h_kernel: a wave takes KZ_NORM_ROWS / four rows per iteration
static int kz_rowgroup_blocks(int64_t n_pad, int rows_per_wave) {
    const int64_t b = (n_pad / rows_per_wave + 3) / 4;
    return (int)(b < KZ_ROWGROUP_MAX_BLOCKS ? (b < 1 ? 1 : b) : KZ_ROWGROUP_MAX_BLOCKS);
}

// ---- lazy images ----------------------------------------------------------------------------------------------
int kz_matrix_image_f32(kz_matrix* m) {
    if (m->packed) return KZ_OK;
    kz_ctx* ctx = m->ctx;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    if (kz_pool_alloc(ctx, (size_t)n_pad * (size_t)m->kg * 16, (void**)&m->packed) != KZ_OK) return KZ_ERR_NOMEM;
    if (m->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pack_f32_kernel<float>, dim3(kz_pack_blocks(n_pad)), dim3(256), 0, ctx->stream, (const float*)m->raw,
                           m->sqn, m->n, (int)m->d, m->metric, m->kg, n_pad, m->packed);
    else
        hipLaunchKernelGGL(kz_pack_f32_kernel<double>, dim3(kz_pack_blocks(n_pad)), dim3(256), 0, ctx->stream, (const double*)m->raw,
                           m->sqn, m->n, (int)m->d, m->metric, m->kg, n_pad, m->packed);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

// Normalised float64 rows of a cosine matrix: x_e / |x| by the IEEE division -- bit for bit what the re-rank's kz_div_shared (and
// the plain division it stands for) gives per element, computed once per row instead of once per (query, candidate) pair.  Only
// where a finalize launch re-ranks HUNDREDS of candidates per query (the wide route on clustered data, long k: kz_knn_fin_wide.h):
// there the kernel is bound by instructions, and the divisions are 40 % of them (bench.py "hard": 12.5 -> see DESIGN section 3.2).
__global__ void kz_norm64_kernel(const float* __restrict__ raw, const double* __restrict__ nrm, int64_t n_elems, int d, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_elems) out[i] = (double)raw[i] / nrm[i / d];
}
int kz_matrix_norm64(kz_matrix* m) {
    if (m->norm64 || m->metric != KZ_COSINE || m->dtype != KZ_F32 || m->raw_only || (m->d & 3) != 0 || m->d > 256) return KZ_OK;
    const size_t bytes = (size_t)m->n * (size_t)m->d * 8;
    if (bytes > ((size_t)8 << 30)) return KZ_OK;   // (of 288 GB; beyond, the per-pair divisions stay)
    kz_ctx* ctx = m->ctx;
    // (the image lives as long as the matrix and is in no footprint gate of the searches: a large one is only built while it is a
    //  small part of what the device has FREE -- a process that shares the GPU with another allocator keeps the per-pair divisions)
    if (bytes > ((size_t)256 << 20)) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b / 8 < bytes) return KZ_OK;
    }
    if (kz_pool_alloc(ctx, bytes, (void**)&m->norm64) != KZ_OK) {
        m->norm64 = nullptr;
        return KZ_OK;   // (an optimisation only)
    }
    const int64_t n_elems = m->n * m->d;
    hipLaunchKernelGGL(kz_norm64_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)m->raw, m->sqn, n_elems,
                       (int)m->d, m->norm64);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_matrix_image_bf(kz_matrix* m) {
    if (m->packed_bf) return KZ_OK;
    kz_ctx* ctx = m->ctx;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    if (kz_pool_alloc(ctx, (size_t)n_pad * (size_t)m->kg_bf * 16, (void**)&m->packed_bf) != KZ_OK) return KZ_ERR_NOMEM;
    if (m->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pack_bf_kernel<float>, dim3(kz_pack_blocks(n_pad)), dim3(256), 0, ctx->stream, (const float*)m->raw,
                           m->sqn, m->n, (int)m->d, m->metric, m->kg_bf, n_pad, m->packed_bf);
    else
        hipLaunchKernelGGL(kz_pack_bf_kernel<double>, dim3(kz_pack_blocks(n_pad)), dim3(256), 0, ctx->stream, (const double*)m->raw,
                           m->sqn, m->n, (int)m->d, m->metric, m->kg_bf, n_pad, m->packed_bf);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

static void kz_center_release(kz_ctx* ctx, kz_center* c) {
    if (!c) return;
    if (--c->refs > 0) return;
    kz_pool_free(ctx, c->d_mu, 0);
    kz_pool_free(ctx, c->d_scale, 0);
    delete c;
}

void kz_himage_free(kz_matrix* m) {
    kz_himage* im = m->himg;
    if (!im) return;
    kz_pool_free(m->ctx, im->packed, 0);
    kz_pool_free(m->ctx, im->bias, 0);
    kz_pool_free(m->ctx, im->rowq, 0);
    kz_pool_free(m->ctx, im->d_max, 0);
    for (int i = 0; i < 2; ++i) {
        kz_pool_free(m->ctx, im->slot[i].packed, 0);
        kz_pool_free(m->ctx, im->slot[i].bias, 0);
        kz_pool_free(m->ctx, im->slot[i].perm, 0);
    }
    kz_center_release(m->ctx, im->center);
    delete im;
    m->himg = nullptr;
}

// centre taken from matrix `from`: column mean (of the normalised rows for cosine) and the fp16 scale
static int kz_center_create(kz_ctx* ctx, const kz_matrix* from, kz_center** out) {
    kz_center* c = new kz_center();
    memset(c, 0, sizeof(*c));
    c->refs = 1;
    const int d = (int)from->d, d_pad = from->kg * 4;
    double* partial = nullptr;
    if (kz_pool_alloc(ctx, (size_t)d_pad * 4, (void**)&c->d_mu) != KZ_OK || kz_pool_alloc(ctx, 16, (void**)&c->d_scale) != KZ_OK ||
        kz_pool_alloc(ctx, (size_t)KZ_COLSUM_BLOCKS * d * 8, (void**)&partial) != KZ_OK) {
        kz_pool_free(ctx, partial, 0);
        kz_center_release(ctx, c);
        return KZ_ERR_NOMEM;
    }
    const int blocks = (int)(from->n < KZ_COLSUM_BLOCKS ? from->n : KZ_COLSUM_BLOCKS);
    if (from->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_colsum_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)from->raw, from->sqn,
                           from->n, d, from->metric, partial);
    else
        hipLaunchKernelGGL(kz_colsum_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)from->raw, from->sqn,
                           from->n, d, from->metric, partial);
    hipLaunchKernelGGL(kz_center_finish_kernel, dim3(d_pad), dim3(256), 0, ctx->stream, partial, blocks, from->n, d, from->d_stats,
                       c->d_mu, c->d_scale);
    const hipError_t e = hipGetLastError();
    kz_pool_free(ctx, partial, 0);   // stream-ordered pool: reuse is ordered behind the kernels above
    if (e != hipSuccess) {
        kz_center_release(ctx, c);
        kz_set_error("kz_knn: centre kernels failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    *out = c;
    return KZ_OK;
}

static int kz_himage_build(kz_matrix* m, kz_center* center) {
    kz_ctx* ctx = m->ctx;
    kz_himage_free(m);
    kz_himage* im = new kz_himage();
    memset(im, 0, sizeof(*im));
    im->center = center;
    ++center->refs;
    m->himg = im;
    const int nsr = m->kg / 4;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    // (+ 32 slices of padding: the kernel's DMA ring runs a few slices past the end of a sweep, kz_knn_h16.h)
    if (kz_pool_alloc(ctx, (size_t)n_pad * (size_t)nsr * 32 + 32 * 4096, (void**)&im->packed) != KZ_OK ||
        kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&im->bias) != KZ_OK ||
        kz_pool_alloc(ctx, (size_t)m->n * 24, (void**)&im->rowq) != KZ_OK || kz_pool_alloc(ctx, 32, (void**)&im->d_max) != KZ_OK) {
        kz_himage_free(m);
        return KZ_ERR_NOMEM;
    }
    hipError_t e = hipMemsetAsync(im->d_max, 0, 32, ctx->stream);
    if (e == hipSuccess) {
        if (m->dtype == KZ_F32)
            hipLaunchKernelGGL(kz_pack_h_kernel<float>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const float*)m->raw,
                               m->sqn, m->n, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, im->packed, im->bias,
                               im->rowq, (unsigned long long*)im->d_max, nullptr);
        else
            hipLaunchKernelGGL(kz_pack_h_kernel<double>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const double*)m->raw,
                               m->sqn, m->n, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, im->packed, im->bias,
                               im->rowq, (unsigned long long*)im->d_max, nullptr);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        kz_himage_free(m);
        kz_set_error("kz_knn: fp16 pack kernel failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    return KZ_OK;
}

// The fp16 image of m's rows in the order perm[0 .. n) (dual pass: index rows sorted by their event threshold), with m's
// own centre and scale, into caller-provided buffers: packed [n_tiles][nsr] x 4 KiB (+ the DMA ring's padding), bias
// [n_tiles * 128].  Bit-identical operands and biases to m's own image, row for row.
int kz_himage_pack_permuted(kz_matrix* m, const int* d_perm, unsigned short* packed, float* bias) {
    kz_ctx* ctx = m->ctx;
    KZ_REQUIRE(m->himg, "kz_himage_pack_permuted: the matrix has no fp16 image");
    const kz_center* center = m->himg->center;
    const int nsr = m->kg / 4;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    if (m->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pack_h_kernel<float>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const float*)m->raw, m->sqn,
                           m->n, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, packed, bias, (double*)nullptr,
                           (unsigned long long*)nullptr, d_perm);
    else
        hipLaunchKernelGGL(kz_pack_h_kernel<double>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const double*)m->raw, m->sqn,
                           m->n, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, packed, bias, (double*)nullptr,
                           (unsigned long long*)nullptr, d_perm);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

// The same for ANY list of n_rows matrix rows (n_pad image rows, a multiple of 128; rows behind n_rows: zero image, bias -inf).
int kz_himage_pack_rows(kz_matrix* m, const int* d_rows, int64_t n_rows, int64_t n_pad, unsigned short* packed, float* bias) {
    kz_ctx* ctx = m->ctx;
    KZ_REQUIRE(m->himg && n_pad % KZ_TILE == 0 && n_rows <= n_pad, "kz_himage_pack_rows: no fp16 image / bad row counts");
    const kz_center* center = m->himg->center;
    const int nsr = m->kg / 4;
    if (m->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pack_h_kernel<float>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const float*)m->raw, m->sqn,
                           n_rows, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, packed, bias, (double*)nullptr,
                           (unsigned long long*)nullptr, d_rows);
    else
        hipLaunchKernelGGL(kz_pack_h_kernel<double>, dim3(kz_rowgroup_blocks(n_pad, 4)), dim3(256), 0, ctx->stream, (const double*)m->raw, m->sqn,
                           n_rows, (int)m->d, m->metric, nsr, n_pad, center->d_mu, center->d_scale, packed, bias, (double*)nullptr,
                           (unsigned long long*)nullptr, d_rows);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

// A second fp16 image of m with its rows dealt over P index ranges (range p = rows p, p + P, p + 2 P, ...; kz_dealt_row): the
// short-list route of the ordinary kernel keeps one list of 16 per query and RANGE, and the ranges must be alike -- in the
// caller's row order the near rows of a query may all sit in one stretch of the matrix (data stored cluster by cluster).
// Same centre, same scale, bit-identical operands and biases, row for row; cached on the image (one P at a time).
__global__ void kz_dealt_perm_kernel(int64_t n, int64_t n_pad, int P, int* __restrict__ perm) {
    const int64_t rp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rp < n_pad) perm[rp] = rp < n ? (int)kz_dealt_row(rp, n, P) : -1;
}

int kz_himage_dealt(kz_matrix* m, int P) {
    kz_ctx* ctx = m->ctx;
    KZ_REQUIRE(m->himg && P >= 2 && (int64_t)P <= m->n, "kz_himage_dealt: no fp16 image / bad range count");
    kz_himage* im = m->himg;
    auto select = [&](int i) {
        im->slot_cur = i;
        im->dealt_P = im->slot[i].P;
        im->dealt_packed = im->slot[i].packed;
        im->dealt_bias = im->slot[i].bias;
        im->dealt_perm = im->slot[i].perm;
    };
    for (int i = 0; i < 2; ++i)
        if (im->slot[i].P == P && im->slot[i].packed) {
            select(i);
            return KZ_OK;
        }
    // not cached: an empty slot, else the one that was NOT selected last
    int v = !im->slot[0].packed ? 0 : (!im->slot[1].packed ? 1 : 1 - im->slot_cur);
    const int nsr = m->kg / 4;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    if (!im->slot[v].packed) {
        if (kz_pool_alloc(ctx, (size_t)n_pad * (size_t)nsr * 32 + 32 * 4096, (void**)&im->slot[v].packed) != KZ_OK ||
            kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&im->slot[v].bias) != KZ_OK ||
            kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&im->slot[v].perm) != KZ_OK) {
            kz_pool_free(ctx, im->slot[v].packed, 0);
            kz_pool_free(ctx, im->slot[v].bias, 0);
            kz_pool_free(ctx, im->slot[v].perm, 0);
            im->slot[v].packed = nullptr;
            im->slot[v].bias = nullptr;
            im->slot[v].perm = nullptr;
            im->slot[v].P = 0;
            // (no memory for a second image: the other slot is re-packed instead, if there is one)
            v = 1 - v;
            if (!im->slot[v].packed) return KZ_ERR_NOMEM;
        }
    }
    im->slot[v].P = 0;
    hipLaunchKernelGGL(kz_dealt_perm_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, ctx->stream, m->n, n_pad, P, im->slot[v].perm);
    KZ_HIP(hipGetLastError());
    const int rc = kz_himage_pack_permuted(m, im->slot[v].perm, im->slot[v].packed, im->slot[v].bias);
    if (rc != KZ_OK) return rc;
    im->slot[v].P = P;
    select(v);
    return KZ_OK;
}

// Make sure query and index carry fp16 images with ONE common centre.  The centre of an existing image wins (index
// first), so the two passes of a fit (target -> source, then source -> target) pack every matrix once.
int kz_himage_ensure(kz_matrix* query, kz_matrix* index) {
    kz_ctx* ctx = index->ctx;
    if (index->himg && query->himg && index->himg->center == query->himg->center) return KZ_OK;
    kz_center* c = nullptr;
    if (index->himg)
        c = index->himg->center;
    else if (query->himg)
        c = query->himg->center;
    int rc = KZ_OK;
    bool created = false;
    if (!c) {
        rc = kz_center_create(ctx, index, &c);
        if (rc != KZ_OK) return rc;
        created = true;
    }
    if (!index->himg || index->himg->center != c) rc = kz_himage_build(index, c);
    if (rc == KZ_OK && query != index && (!query->himg || query->himg->center != c)) rc = kz_himage_build(query, c);
    if (created) kz_center_release(ctx, c);   // the images hold their own references
    return rc;
}

// Wait for the matrix' norm kernel and read its verdict: max row norm (host copy for the lower tiers' bound) and the
// non-finite flag.  Idempotent; called by kz_matrix_create (host rows) or by the first kz_knn that uses the matrix.
int kz_matrix_check(kz_matrix* m) {
    if (m->checked) return KZ_OK;
    kz_ctx* ctx = m->ctx;
    KZ_HIP(hipMemcpyAsync(ctx->h_counters, m->d_stats, 40, hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->h_counters[8] != 0) {
        kz_set_error("kz_matrix_create: input contains NaN, infinity or a value too large for float32");
        return KZ_ERR_NONFINITE;
    }
    memcpy(&m->max_norm, ctx->h_counters, 8);
    m->checked = true;
    return KZ_OK;
}

extern "C" {

// rows_on_device: 0 = host rows (copied), 1 = device rows (copied), 2 = device rows BORROWED: the matrix keeps the
// caller's pointer, the caller keeps the buffer alive and unchanged until kz_matrix_destroy (zero-copy fit)
int kz_matrix_create(kz_ctx* ctx, const void* rows, int rows_on_device, int64_t n, int64_t d, int dtype, int metric,
                     kz_matrix** out) {
    KZ_REQUIRE(ctx && rows && out, "kz_matrix_create: null argument");
    KZ_REQUIRE(n > 0 && d > 0, "kz_matrix_create: empty matrix (n=%lld, d=%lld)", (long long)n, (long long)d);
    KZ_REQUIRE(d <= 65536, "kz_matrix_create: d=%lld too large", (long long)d);
    KZ_REQUIRE(n < ((int64_t)1 << 31) - 256, "kz_matrix_create: n=%lld exceeds the int32 row-id range", (long long)n);
    KZ_REQUIRE(dtype == KZ_F32 || dtype == KZ_F64, "kz_matrix_create: dtype must be KZ_F32 or KZ_F64");
    KZ_REQUIRE(metric >= KZ_EUCLIDEAN && metric <= KZ_MINKOWSKI, "kz_matrix_create: unknown metric %d", metric);
    KZ_REQUIRE(rows_on_device >= 0 && rows_on_device <= 3, "kz_matrix_create: rows_on_device must be 0, 1, 2 or 3");
    KZ_HIP(hipSetDevice(ctx->device));
    kz_matrix* m = new kz_matrix();
    memset(m, 0, sizeof(*m));
    m->ctx = ctx;
    m->n = n;
    m->d = d;
    m->dtype = dtype;
    m->metric = metric;
    m->mink_p = 2.0;
    m->n_tiles = (n + KZ_TILE - 1) / KZ_TILE;
    const int64_t d_pad = ((d + KZ_KSLICE - 1) / KZ_KSLICE) * KZ_KSLICE;
    m->kg = (int)(d_pad / 4);
    m->kg_bf = m->kg;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    const size_t esz = dtype == KZ_F32 ? 4 : 8;
    const size_t raw_bytes = (size_t)n * (size_t)d * esz;
    auto fail = [&](int code) {
        kz_matrix_destroy(m);
        return code;
    };
    m->raw_borrowed = rows_on_device >= 2;
    if (m->raw_borrowed) m->raw = const_cast<void*>(rows);
    if (rows_on_device == 3) {
        // rows only (a row SOURCE for kz_dsl_fit's centroid gather: the multi-rank DSL path gathers the source shards for that one
        // kernel): nothing is computed, allocated or waited for; every search entry point refuses the matrix
        m->raw_only = true;
        m->checked = true;
        *out = m;
        return KZ_OK;
    }
    if ((!m->raw_borrowed && kz_pool_alloc(ctx, raw_bytes, &m->raw) != KZ_OK) ||
        kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&m->bias) != KZ_OK || kz_pool_alloc(ctx, (size_t)n * 8, (void**)&m->sqn) != KZ_OK ||
        kz_pool_alloc(ctx, 64, (void**)&m->d_stats) != KZ_OK) {
        kz_set_error("kz_matrix_create: out of device memory (%zu B of rows)", raw_bytes);
        return fail(KZ_ERR_NOMEM);
    }
    hipError_t e = hipSuccess;
    if (!m->raw_borrowed)
        e = hipMemcpyAsync(m->raw, rows, raw_bytes, rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: copy failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    // d_stats: [0] max row norm, [1] max |operand element|, int at byte 32: non-finite flag
    e = hipMemsetAsync(m->d_stats, 0, 64, ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: memset failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    auto* st = (unsigned long long*)m->d_stats;
    int* bad = (int*)(m->d_stats + 4);
    if (dtype == KZ_F32)
        hipLaunchKernelGGL(kz_norms_kernel<float>, dim3(kz_rowgroup_blocks(n_pad, KZ_NORM_ROWS)), dim3(256), 0, ctx->stream, (const float*)m->raw, n,
                           (int)d, metric, n_pad, m->bias, m->sqn, st, bad);
    else
        hipLaunchKernelGGL(kz_norms_kernel<double>, dim3(kz_rowgroup_blocks(n_pad, KZ_NORM_ROWS)), dim3(256), 0, ctx->stream, (const double*)m->raw, n,
                           (int)d, metric, n_pad, m->bias, m->sqn, st, bad);
    e = hipGetLastError();
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: norm kernel failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    // Host rows: the copy above was synchronous anyway, so the finiteness verdict is checked here and the caller gets
    // KZ_ERR_NONFINITE from the call that passed the data (scikit-learn rejects such input in fit, too).  Device rows:
    // NOTHING is waited for -- the matrix is usable at once, the verdict stays in d_stats and is checked by the first kz_knn
    // that searches the matrix (its result read-back synchronises anyway): fit() enqueues both matrices and the reverse
    // pass back to back.
    if (rows_on_device == 0) {
        const int rc = kz_matrix_check(m);
        if (rc != KZ_OK) return fail(rc);
    }
    *out = m;
    return KZ_OK;
}

int kz_matrix_destroy(kz_matrix* m) {
    if (!m) return KZ_OK;
    if (m->ctx) {
        (void)hipSetDevice(m->ctx->device);
        kz_himage_free(m);
        if (!m->raw_borrowed) kz_pool_free(m->ctx, m->raw, 0);
        kz_pool_free(m->ctx, m->packed, 0);
        kz_pool_free(m->ctx, m->packed_bf, 0);
        kz_pool_free(m->ctx, m->norm64, 0);
        kz_pool_free(m->ctx, m->bias, 0);
        kz_pool_free(m->ctx, m->sqn, 0);
        kz_pool_free(m->ctx, m->d_stats, 0);
    }
    delete m;
    return KZ_OK;
}

int kz_matrix_set_minkowski_p(kz_matrix* m, double p) {
    KZ_REQUIRE(m != nullptr, "kz_matrix_set_minkowski_p: null matrix");
    KZ_REQUIRE(m->metric == KZ_MINKOWSKI, "kz_matrix_set_minkowski_p: the matrix was not created for KZ_MINKOWSKI");
    KZ_REQUIRE(p >= 1.0 && p < 1e6, "kz_matrix_set_minkowski_p: p must be >= 1 (got %g)", p);
    m->mink_p = p;
    return KZ_OK;
}

int kz_matrix_shape(const kz_matrix* m, int64_t* n, int64_t* d, int* dtype, int* metric) {
    KZ_REQUIRE(m != nullptr, "kz_matrix_shape: null matrix");
    if (n) *n = m->n;
    if (d) *d = m->d;
    if (dtype) *dtype = m->dtype;
    if (metric) *metric = m->metric;
    return KZ_OK;
}

}  // extern "C"
