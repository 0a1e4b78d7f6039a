// k-occurrence histogram and its reductions: the device part of kiez.analysis.hubness_score
// (kiez/analysis/estimation.py:197-351).  "Next" row (f-1) of SURVEY.md §8: it consumes exactly the [n, k] neighbour
// index matrix the hot path produces.  Bound: HBM/atomics, microseconds at n = 1e5..1e6.
#include "kz_common.h"

// np.bincount(nn_ind[:, :k].ravel(), minlength=n_bins), negative ids dropped first (estimation.py:283-292)
__global__ void kz_k_occurrence_kernel(const int64_t* __restrict__ ind, int64_t n_rows, int cols, int k, int64_t n_bins,
                                       unsigned long long* __restrict__ kocc, int* __restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * k) return;
    const int64_t r = e / k;
    const int c = (int)(e - r * k);
    const int64_t id = ind[r * cols + c];
    if (id < 0) return;
    if (id >= n_bins) {
        atomicOr(bad, 1);
        return;
    }
    atomicAdd(kocc + id, 1ull);
}

__global__ void kz_minmax_i64_kernel(const int64_t* __restrict__ in, int64_t count, long long* __restrict__ mn,
                                     long long* __restrict__ mx) {
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const long long v = in[e];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const long long ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mn, lo);
        atomicMax(mx, hi);
    }
}

// the same over the first k columns of a row-major [rows, cols] matrix (element e of the rows x k sub-matrix)
__global__ void kz_minmax_i64_2d_kernel(const int64_t* __restrict__ in, int64_t rows, int cols, int k, long long* __restrict__ mn,
                                        long long* __restrict__ mx) {
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    const int64_t count = rows * k;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / k;
        const long long v = in[r * cols + (e - r * k)];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const long long ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mn, lo);
        atomicMax(mx, hi);
    }
}

// pass 1 (exact integer reductions): sum, max, #zeros, sum and count of entries >= hub threshold
__global__ void kz_kocc_int_stats_kernel(const long long* __restrict__ kocc, int64_t n, double thr,
                                         unsigned long long* __restrict__ out) {
    unsigned long long s = 0, z = 0, hs = 0, hc = 0;
    long long mx = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const long long v = kocc[e];
        s += (unsigned long long)v;
        mx = v > mx ? v : mx;
        z += (v == 0);
        if ((double)v >= thr) {
            hs += (unsigned long long)v;
            hc += 1;
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        z += __shfl_xor(z, off, 64);
        hs += __shfl_xor(hs, off, 64);
        hc += __shfl_xor(hc, off, 64);
        const long long om = __shfl_xor(mx, off, 64);
        mx = om > mx ? om : mx;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, s);
        atomicMax(out + 1, (unsigned long long)mx);
        atomicAdd(out + 2, z);
        atomicAdd(out + 3, hs);
        atomicAdd(out + 4, hc);
    }
}

// pass 2 (float reductions around the mean): sum|x-m|, sum(x-m)^2, sum(x-m)^3, sum sqrt(x)
__global__ void kz_kocc_float_stats_kernel(const long long* __restrict__ kocc, int64_t n, double mean, double* __restrict__ out) {
    double a1 = 0, a2 = 0, a3 = 0, sq = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)kocc[e];
        const double dlt = x - mean;
        a1 += fabs(dlt);
        a2 += dlt * dlt;
        a3 += dlt * dlt * dlt;
        sq += sqrt(x);
    }
    a1 = kz_wave_sum(a1);
    a2 = kz_wave_sum(a2);
    a3 = kz_wave_sum(a3);
    sq = kz_wave_sum(sq);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, a1);
        atomicAdd(out + 1, a2);
        atomicAdd(out + 2, a3);
        atomicAdd(out + 3, sq);
    }
}

// Gini numerator sum_i sum_j |x_i - x_j| (estimation.py:83-97), exact in integers
__global__ void kz_gini_kernel(const long long* __restrict__ kocc, int64_t n, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long xi = kocc[i];
        for (int64_t jj = 0; jj < n; ++jj) {
            const long long dlt = kocc[jj] - xi;
            acc += (unsigned long long)(dlt < 0 ? -dlt : dlt);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

// np.argwhere(cond).ravel(): ascending ids with kocc == 0 (mode 0) or kocc >= thr (mode 1); one workgroup, ordered scan
__global__ __launch_bounds__(1024) void kz_kocc_select_kernel(const long long* __restrict__ kocc, int64_t n, int mode, double thr,
                                                              int64_t* __restrict__ out, unsigned long long* __restrict__ count) {
    __shared__ int s_wave[16];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < n; c0 += 1024) {
        const int64_t e = c0 + tid;
        bool hit = false;
        if (e < n) {
            const long long v = kocc[e];
            hit = mode == 0 ? (v == 0) : ((double)v >= thr);
        }
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += s_wave[w];
            total += s_wave[w];
        }
        if (hit) out[s_base + woff + before] = e;
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) *count = s_base;
}

// hits@k (kiez/evaluate/eval_metrics.py:7-12): position of gold[i] inside nn_ind[i, :], histogrammed over rows
__global__ void kz_hit_positions_kernel(const int64_t* __restrict__ ind, const int64_t* __restrict__ gold, int64_t n, int cols,
                                        unsigned long long* __restrict__ hist) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t gt = gold[r];
    if (gt == INT64_MIN) return;  // row has no gold entry
    int pos = cols;                // "not found"
    for (int c = 0; c < cols; ++c)
        if (ind[r * cols + c] == gt) {
            pos = c;
            break;
        }
    atomicAdd(hist + pos, 1ull);
}

// ---- self-test of kz_div_shared (kz_common.h): pseudo-random pairs, the bits of the shared-reciprocal quotient against a / b ----
__device__ __forceinline__ unsigned long long kz_mix64(unsigned long long z) {   // (splitmix64 finaliser)
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ void kz_selftest_div_kernel(long long count, unsigned long long seed, int mode, unsigned long long* __restrict__ bad) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_bad = 0;
    for (long long i = i0; i < count; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r0 = kz_mix64(seed + 2ull * (unsigned long long)i), r1 = kz_mix64(seed + 2ull * (unsigned long long)i + 1ull);
        // numerator: a float32 value with a random significand, magnitude 2^-40 .. 2^20, either sign (mode 1: every 4th one has an
        // all-ones or a single-bit significand)
        unsigned man = (unsigned)(r0 & 0x7fffffu);
        if (mode == 1 && ((r0 >> 23) & 3u) == 0u) man = ((r0 >> 25) & 1u) ? 0x7fffffu : (1u << ((r0 >> 26) % 23u));
        const unsigned ex = 127u - 40u + (unsigned)((r0 >> 32) % 61u);
        const float af = __uint_as_float((unsigned)((r0 >> 63) << 31) | (ex << 23) | man);
        const double a = (double)af;
        // divisor: |a| times a factor in [1, 2^12) with a random 52-bit significand (a row norm is at least every element)
        unsigned long long bman = r1 & 0xfffffffffffffull;
        if (mode == 1 && ((r1 >> 52) & 3u) == 0u) bman = ((r1 >> 54) & 1u) ? 0xfffffffffffffull : (1ull << ((r1 >> 55) % 52u));
        const double f = __longlong_as_double((long long)(((1023ull + ((r1 >> 58) % 12ull)) << 52) | bman));
        const double b = fabs(a) * f;
        const double rcp = 1.0 / b;
        const double q = kz_div_shared(a, b, rcp), ref = a / b;
        n_bad += (__double_as_longlong(q) != __double_as_longlong(ref)) ? 1ull : 0ull;
    }
    if (n_bad) atomicAdd(bad, n_bad);
}

extern "C" {

int kz_hit_positions(kz_ctx* ctx, const int64_t* d_ind, const int64_t* d_gold, int64_t n, int cols, int64_t* d_hist) {
    KZ_REQUIRE(ctx && d_ind && d_gold && d_hist && n > 0 && cols >= 1, "kz_hit_positions: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    KZ_HIP(hipMemsetAsync(d_hist, 0, (size_t)(cols + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(kz_hit_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_ind, d_gold, n, cols,
                       (unsigned long long*)d_hist);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_minmax_i64(kz_ctx* ctx, const int64_t* d_in, int64_t count, int64_t* h_min, int64_t* h_max) {
    KZ_REQUIRE(ctx && d_in && h_min && h_max && count > 0, "kz_minmax_i64: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    long long* d = (long long*)(ctx->d_counters + 32);
    const long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
    KZ_HIP(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const int blocks = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    hipLaunchKernelGGL(kz_minmax_i64_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_in, count, d, d + 1);
    KZ_HIP(hipGetLastError());
    long long h[2];
    KZ_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    *h_min = h[0];
    *h_max = h[1];
    return KZ_OK;
}

int kz_minmax_i64_2d(kz_ctx* ctx, const int64_t* d_in, int64_t rows, int cols, int k, int64_t* h_min, int64_t* h_max) {
    KZ_REQUIRE(ctx && d_in && h_min && h_max && rows > 0 && cols >= 1 && k >= 1 && k <= cols, "kz_minmax_i64_2d: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    long long* d = (long long*)(ctx->d_counters + 32);
    const long long init[2] = {0x7fffffffffffffffLL, -0x7fffffffffffffffLL - 1};
    KZ_HIP(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    const int64_t count = rows * k;
    const int blocks = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    hipLaunchKernelGGL(kz_minmax_i64_2d_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_in, rows, cols, k, d, d + 1);
    KZ_HIP(hipGetLastError());
    long long h[2];
    KZ_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    *h_min = h[0];
    *h_max = h[1];
    return KZ_OK;
}

int kz_k_occurrence(kz_ctx* ctx, const int64_t* d_ind, int64_t n_rows, int cols, int k, int64_t n_bins, int64_t* d_kocc) {
    KZ_REQUIRE(ctx && d_ind && d_kocc, "kz_k_occurrence: null argument");
    KZ_REQUIRE(n_rows > 0 && cols >= 1 && k >= 1 && k <= cols && n_bins > 0, "kz_k_occurrence: bad shape");
    KZ_HIP(hipSetDevice(ctx->device));
    int* bad = ctx->d_counters + 40;
    KZ_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    KZ_HIP(hipMemsetAsync(d_kocc, 0, (size_t)n_bins * 8, ctx->stream));
    const int64_t total = n_rows * k;
    hipLaunchKernelGGL(kz_k_occurrence_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_ind, n_rows, cols,
                       k, n_bins, (unsigned long long*)d_kocc, bad);
    KZ_HIP(hipGetLastError());
    KZ_HIP(hipMemcpyAsync(ctx->h_counters + 40, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    KZ_REQUIRE(ctx->h_counters[40] == 0, "kz_k_occurrence: neighbour id >= n_bins");
    return KZ_OK;
}

int kz_kocc_stats(kz_ctx* ctx, const int64_t* d_kocc, int64_t n, double hub_threshold, int with_gini, double* h_out) {
    KZ_REQUIRE(ctx && d_kocc && h_out && n > 0, "kz_kocc_stats: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    void* scratch = nullptr;
    int rc = kz_scratch(ctx, 256, &scratch);
    if (rc != KZ_OK) return rc;
    unsigned long long* di = (unsigned long long*)scratch;  // [0..4] int stats, [5] gini
    double* df = (double*)scratch + 8;                      // [8..11] float stats
    KZ_HIP(hipMemsetAsync(scratch, 0, 128, ctx->stream));
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(kz_kocc_int_stats_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const long long*)d_kocc, n, hub_threshold, di);
    KZ_HIP(hipGetLastError());
    unsigned long long hi[8];
    KZ_HIP(hipMemcpyAsync(hi, di, sizeof(hi), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    const double mean = (double)hi[0] / (double)n;
    hipLaunchKernelGGL(kz_kocc_float_stats_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const long long*)d_kocc, n, mean, df);
    if (with_gini)
        hipLaunchKernelGGL(kz_gini_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const long long*)d_kocc, n, di + 5);
    KZ_HIP(hipGetLastError());
    double hf[4];
    KZ_HIP(hipMemcpyAsync(hf, df, sizeof(hf), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipMemcpyAsync(hi, di, sizeof(hi), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    h_out[0] = (double)hi[0];  // sum
    h_out[1] = hf[0];          // sum |x - mean|
    h_out[2] = hf[1];          // sum (x - mean)^2
    h_out[3] = hf[2];          // sum (x - mean)^3
    h_out[4] = hf[3];          // sum sqrt(x)
    h_out[5] = (double)hi[1];  // max
    h_out[6] = (double)hi[2];  // # zeros
    h_out[7] = (double)hi[3];  // sum over hubs
    h_out[8] = (double)hi[4];  // # hubs
    h_out[9] = with_gini ? (double)hi[5] : 0.0;  // gini numerator
    return KZ_OK;
}

int kz_kocc_select(kz_ctx* ctx, const int64_t* d_kocc, int64_t n, int mode, double thr, int64_t* d_out, int64_t* h_count) {
    KZ_REQUIRE(ctx && d_kocc && d_out && h_count && n > 0 && (mode == 0 || mode == 1), "kz_kocc_select: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    unsigned long long* dc = (unsigned long long*)(ctx->d_counters + 44);
    hipLaunchKernelGGL(kz_kocc_select_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const long long*)d_kocc, n, mode, thr, d_out, dc);
    KZ_HIP(hipGetLastError());
    unsigned long long hc = 0;
    KZ_HIP(hipMemcpyAsync(&hc, dc, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    *h_count = (int64_t)hc;
    return KZ_OK;
}

int kz_selftest_div(kz_ctx* ctx, int64_t count, uint64_t seed, int mode, int64_t* h_mismatch) {
    KZ_REQUIRE(ctx && h_mismatch && count >= 0 && (mode == 0 || mode == 1), "kz_selftest_div: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    unsigned long long* dc = (unsigned long long*)(ctx->d_counters + 44);
    KZ_HIP(hipMemsetAsync(dc, 0, 8, ctx->stream));
    hipLaunchKernelGGL(kz_selftest_div_kernel, dim3(ctx->n_cus * 8), dim3(256), 0, ctx->stream, (long long)count, (unsigned long long)seed, mode, dc);
    KZ_HIP(hipGetLastError());
    unsigned long long hc = 0;
    KZ_HIP(hipMemcpyAsync(&hc, dc, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    KZ_HIP(hipStreamSynchronize(ctx->stream));
    *h_mismatch = (int64_t)hc;
    return KZ_OK;
}

}  // extern "C"
