// kz_matrix: HBM-resident embedding matrix = exact rows + float64 norms + MFMA-packed float32 tile image.
//
// Packed image layout (the layout the fused distance kernel streams, DESIGN.md "Data layout in HBM"):
//     packed[tile][kgroup][row_in_tile][4]      tile = row / 128, kgroup = k / 4, d padded to a multiple of 16
// One 16-k slice of one tile is 4 consecutive kgroups = a contiguous 8 KiB block, and consecutive slices
// (also across tile boundaries) are consecutive in memory, so the kernel's HBM stream is purely linear.
// Inside a kgroup the 128 rows x 16 B are exactly the conflict-free ds_read_b128 image of the MFMA A/B operand.
//
// Split-bf16 image (first-pass operand of the bf16x2 kernel, same geometry and size): x = hi + lo + r with
// hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-16 |x| (1 + 2^-7).  One 16-k slice = four 2 KiB planes
//     packed_bf[tile][slice][plane][row_in_tile][8]     plane 0/1 = hi of k 0-7 / 8-15, plane 2/3 = lo of the same
// i.e. plane p of lane-half h is the 16-byte fragment v_mfma_f32_32x32x16_bf16 expects (lane l: row l&31, k 8(l>>5)+j).
#include "kz_common.h"

// round-to-nearest-even float32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned short kz_bf16_rn(float f) {
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float kz_bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

template <typename T>
__global__ __launch_bounds__(256) void kz_pack_kernel(const T* __restrict__ raw, int64_t n, int d, int metric, int kg,
                                                      int kg_bf, int64_t n_pad, float* __restrict__ packed,
                                                      unsigned short* __restrict__ packed_bf,
                                                      float* __restrict__ bias, double* __restrict__ sqn,
                                                      unsigned long long* __restrict__ maxnorm_bits,
                                                      int* __restrict__ bad_flag) {
    __shared__ double s_max[4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int d_pad = kg * 4;
    const int d_pad_bf = kg_bf * 4;
    double wmax = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n_pad; row += (int64_t)gridDim.x * 4) {
        const int64_t tile = row >> 7;
        const int r = (int)(row & 127);
        float* dst = packed + (tile * kg) * (int64_t)(KZ_TILE * 4) + r * 4;
        // bf16 image: element k of this row -> slice k/16, plane (k/8)&1 (+2 for lo), 8 values per row and plane
        unsigned short* dbf = packed_bf + (tile * kg_bf) * (int64_t)(KZ_TILE * 8) + r * 8;
        auto bf_off = [](int k) { return (int64_t)(k >> 4) * (4 * KZ_TILE * 8) + ((k >> 3) & 1) * (KZ_TILE * 8) + (k & 7); };
        if (row >= n) {
            for (int k = lane; k < d_pad_bf; k += 64) {
                if (k < d_pad) dst[(int64_t)(k >> 2) * (KZ_TILE * 4) + (k & 3)] = 0.0f;
                dbf[bf_off(k)] = 0;
                dbf[bf_off(k) + 2 * KZ_TILE * 8] = 0;
            }
            if (lane == 0) bias[row] = -INFINITY;
            continue;
        }
        const T* x = raw + row * (int64_t)d;
        const double sq = kz_wave_dot(x, x, d, lane);
        if (!(sq <= 1e30)) {  // NaN, inf, or too large for the float32 operand image
            if (lane == 0) atomicOr(bad_flag, 1);
        }
        double scale_div = 1.0;
        if (metric == KZ_COSINE) {
            double nrm = sqrt(sq);
            if (nrm == 0.0) nrm = 1.0;  // sklearn normalize(): zero rows stay zero
            scale_div = nrm;
            if (lane == 0) {
                sqn[row] = nrm;
                bias[row] = 0.0f;
            }
            wmax = fmax(wmax, 1.0);
        } else {
            if (lane == 0) {
                sqn[row] = sq;
                bias[row] = (float)(-0.5 * sq);
            }
            wmax = fmax(wmax, sqrt(sq));
        }
        for (int k = lane; k < d_pad_bf; k += 64) {
            double vd = 0.0;
            if (k < d) vd = (metric == KZ_COSINE) ? (double)x[k] / scale_div : (double)x[k];
            if (k < d_pad) dst[(int64_t)(k >> 2) * (KZ_TILE * 4) + (k & 3)] = (float)vd;
            const unsigned short hi = kz_bf16_rn((float)vd);
            const unsigned short lo = kz_bf16_rn((float)(vd - (double)kz_bf16_to_f32(hi)));
            dbf[bf_off(k)] = hi;
            dbf[bf_off(k) + 2 * KZ_TILE * 8] = lo;
        }
    }
    if (lane == 0) s_max[wave] = wmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        atomicMax(maxnorm_bits, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order as integers
    }
}

extern "C" {

int kz_matrix_create(kz_ctx* ctx, const void* rows, int rows_on_device, int64_t n, int64_t d, int dtype, int metric,
                     kz_matrix** out) {
    KZ_REQUIRE(ctx && rows && out, "kz_matrix_create: null argument");
    KZ_REQUIRE(n > 0 && d > 0, "kz_matrix_create: empty matrix (n=%lld, d=%lld)", (long long)n, (long long)d);
    KZ_REQUIRE(d <= 65536, "kz_matrix_create: d=%lld too large", (long long)d);
    KZ_REQUIRE(n < ((int64_t)1 << 31) - 256, "kz_matrix_create: n=%lld exceeds the int32 row-id range", (long long)n);
    KZ_REQUIRE(dtype == KZ_F32 || dtype == KZ_F64, "kz_matrix_create: dtype must be KZ_F32 or KZ_F64");
    KZ_REQUIRE(metric == KZ_EUCLIDEAN || metric == KZ_SQEUCLIDEAN || metric == KZ_COSINE,
               "kz_matrix_create: unknown metric %d", metric);
    KZ_HIP(hipSetDevice(ctx->device));
    kz_matrix* m = new kz_matrix();
    memset(m, 0, sizeof(*m));
    m->ctx = ctx;
    m->n = n;
    m->d = d;
    m->dtype = dtype;
    m->metric = metric;
    m->n_tiles = (n + KZ_TILE - 1) / KZ_TILE;
    const int64_t d_pad = ((d + KZ_KSLICE - 1) / KZ_KSLICE) * KZ_KSLICE;
    m->kg = (int)(d_pad / 4);
    const int64_t d_pad_bf = d_pad;  // (kept separate from d_pad: the bf16 kernel is free to use its own slice count)
    m->kg_bf = (int)(d_pad_bf / 4);
    const size_t packed_bf_bytes = (size_t)(m->n_tiles * KZ_TILE) * (size_t)d_pad_bf * 4;
    const int64_t n_pad = m->n_tiles * KZ_TILE;
    const size_t esz = dtype == KZ_F32 ? 4 : 8;
    const size_t raw_bytes = (size_t)n * (size_t)d * esz;
    const size_t packed_bytes = (size_t)n_pad * (size_t)d_pad * 4;
    auto fail = [&](int code) {
        kz_matrix_destroy(m);
        return code;
    };
    if (kz_pool_alloc(ctx, raw_bytes, &m->raw) != KZ_OK || kz_pool_alloc(ctx, packed_bytes, (void**)&m->packed) != KZ_OK ||
        kz_pool_alloc(ctx, packed_bf_bytes, (void**)&m->packed_bf) != KZ_OK ||
        kz_pool_alloc(ctx, (size_t)n_pad * 4, (void**)&m->bias) != KZ_OK ||
        kz_pool_alloc(ctx, (size_t)n * 8, (void**)&m->sqn) != KZ_OK) {
        kz_set_error("kz_matrix_create: out of device memory (raw %zu B + 2 x packed %zu B)", raw_bytes, packed_bytes);
        return fail(KZ_ERR_NOMEM);
    }
    hipError_t e = hipMemcpyAsync(m->raw, rows, raw_bytes, rows_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: copy failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    // counters[0..1] = max-norm bits (u64), counters[2] = bad flag
    e = hipMemsetAsync(ctx->d_counters, 0, 4 * sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: memset failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    int64_t blocks64 = (n_pad + 3) / 4;
    int blocks = (int)(blocks64 < 2048 ? blocks64 : 2048);
    auto* mx = (unsigned long long*)ctx->d_counters;
    int* bad = ctx->d_counters + 2;
    if (dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pack_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)m->raw, n, (int)d,
                           metric, m->kg, m->kg_bf, n_pad, m->packed, m->packed_bf, m->bias, m->sqn, mx, bad);
    else
        hipLaunchKernelGGL(kz_pack_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double*)m->raw, n,
                           (int)d, metric, m->kg, m->kg_bf, n_pad, m->packed, m->packed_bf, m->bias, m->sqn, mx, bad);
    e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        kz_set_error("kz_matrix_create: pack kernel failed: %s", hipGetErrorString(e));
        return fail(KZ_ERR_HIP);
    }
    if (ctx->h_counters[2] != 0) {
        kz_set_error("kz_matrix_create: input contains NaN, infinity or a value too large for float32");
        return fail(KZ_ERR_NONFINITE);
    }
    unsigned long long bits;
    memcpy(&bits, ctx->h_counters, 8);
    memcpy(&m->max_norm, &bits, 8);
    *out = m;
    return KZ_OK;
}

int kz_matrix_destroy(kz_matrix* m) {
    if (!m) return KZ_OK;
    if (m->ctx) {
        (void)hipSetDevice(m->ctx->device);
        kz_pool_free(m->ctx, m->raw, 0);
        kz_pool_free(m->ctx, m->packed, 0);
        kz_pool_free(m->ctx, m->packed_bf, 0);
        kz_pool_free(m->ctx, m->bias, 0);
        kz_pool_free(m->ctx, m->sqn, 0);
    }
    delete m;
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
