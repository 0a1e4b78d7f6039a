// Device sort used by the dual pass (kz_knn_dual.h): rows ordered by their event threshold -- a STABLE least-significant-digit radix
// sort of (float key, int value) pairs, four passes of eight bits, hand-written (rounds 2 - 4 called rocPRIM here).
// Per pass three launches on the context's stream:
//   kz_sort_hist_kernel     a workgroup counts the digits of its tile of 2 048 pairs (LDS atomics) -> hist[digit][tile]
//   kz_sort_scan_kernel     workgroup d: exclusive scan of digit d's counters over the tiles + the digit's total (the scan of the 256
//                           totals is done by every scatter workgroup for itself)
//   kz_sort_scatter_kernel  the workgroup walks its tile again in input order (eight rounds of 256 pairs: round, wave, lane) and
//                           gives every pair its rank among the tile's pairs of the same digit: lanes of a wave that hold the same
//                           digit find each other with eight ballots, waves and rounds through per-digit counters in LDS
// Keys travel as their order-preserving bit pattern (sign flip; complemented for a descending sort, which keeps the sort stable in
// the descending direction too) and are turned back into floats by the last pass.  n <= 2^31 - 1 pairs; the sorts of a shared sweep
// are 10^5 .. 2 x 10^6 pairs, < 0.2 % of a step.
#include "kz_common.h"

namespace {
constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;

__device__ __forceinline__ unsigned kz_sort_key_bits(float f, int descending) {
    const unsigned b = __float_as_uint(f);
    const unsigned u = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);   // ascending unsigned order == ascending float order (-0 < +0)
    return descending ? ~u : u;
}
__device__ __forceinline__ float kz_sort_key_float(unsigned u, int descending) {
    if (descending) u = ~u;
    return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu));
}

// FIRST: the keys are floats (pass 0 reads the caller's array)
template <bool FIRST>
__device__ __forceinline__ unsigned kz_sort_load(const void* keys, int64_t i, int descending) {
    return FIRST ? kz_sort_key_bits(reinterpret_cast<const float*>(keys)[i], descending) : reinterpret_cast<const unsigned*>(keys)[i];
}

template <bool FIRST>
__global__ __launch_bounds__(SORT_THREADS) void kz_sort_hist_kernel(const void* __restrict__ keys, int n, int shift, int descending, int n_tiles,
                                                                     int* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(kz_sort_load<FIRST>(keys, i, descending) >> shift) & 255u], 1);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// workgroup d: exclusive scan of digit d's counters over the tiles, in place (coalesced chunks of 256 with a carry), and the digit's
// total -> totals[d]; the scatter kernel adds the exclusive scan of the 256 totals itself
__global__ __launch_bounds__(SORT_THREADS) void kz_sort_scan_kernel(int* __restrict__ hist, int n_tiles, int* __restrict__ totals) {
    __shared__ int part[SORT_THREADS];
    int* row = hist + (int64_t)blockIdx.x * n_tiles;
    int carry = 0;
    for (int c0 = 0; c0 < n_tiles; c0 += SORT_THREADS) {
        const int i = c0 + threadIdx.x;
        const int v = i < n_tiles ? row[i] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < SORT_THREADS; off <<= 1) {   // (Hillis-Steele, inclusive)
            const int add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_tiles) row[i] = carry + part[threadIdx.x] - v;
        carry += part[SORT_THREADS - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void kz_sort_scatter_kernel(const void* __restrict__ keys_in, const int* __restrict__ vals_in, int n, int shift,
                                                                        int descending, int n_tiles, const int* __restrict__ offs,
                                                                        const int* __restrict__ totals, void* __restrict__ keys_out,
                                                                        int* __restrict__ vals_out) {
    __shared__ int first[256];        // first output slot of this tile's pairs of a digit
    __shared__ int seen[256];         // pairs of the digit in the rounds already placed
    __shared__ int cnt[4][256];       // this round: pairs of the digit per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        // first slot of digit d in this tile = (pairs of smaller digits, all tiles) + (pairs of digit d in earlier tiles)
        const int mine = totals[threadIdx.x];
        seen[threadIdx.x] = mine;   // (scratch for the scan of the 256 totals)
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int add = threadIdx.x >= off ? seen[threadIdx.x - off] : 0;
            __syncthreads();
            seen[threadIdx.x] += add;
            __syncthreads();
        }
        first[threadIdx.x] = seen[threadIdx.x] - mine + offs[(int64_t)threadIdx.x * n_tiles + blockIdx.x];
        __syncthreads();
    }
    seen[threadIdx.x] = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int64_t i = base + r * SORT_THREADS + threadIdx.x;
        const bool live = i < n;
        unsigned u = 0u;
        int v = 0, digit = 0;
        if (live) {
            u = kz_sort_load<FIRST>(keys_in, i, descending);
            v = vals_in[i];
            digit = (int)((u >> shift) & 255u);
        }
        // the lanes of this wave that hold the same digit (dead lanes: nobody's peer)
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot(live && ((digit >> b) & 1));
            peers &= ((digit >> b) & 1) ? m : ~m;
        }
        const int in_wave = (int)__popcll(peers & lt);
        if (live && in_wave == 0) cnt[wave][digit] = (int)__popcll(peers);
        __syncthreads();
        if (live) {
            int rank = seen[digit] + in_wave;
            for (int w = 0; w < wave; ++w) rank += cnt[w][digit];
            const int64_t o = (int64_t)first[digit] + rank;
            if (LAST)
                reinterpret_cast<float*>(keys_out)[o] = kz_sort_key_float(u, descending);
            else
                reinterpret_cast<unsigned*>(keys_out)[o] = u;
            vals_out[o] = v;
        }
        __syncthreads();
        {
            const int d = threadIdx.x;
            seen[d] += cnt[0][d] + cnt[1][d] + cnt[2][d] + cnt[3][d];
            cnt[0][d] = cnt[1][d] = cnt[2][d] = cnt[3][d] = 0;
        }
        __syncthreads();
    }
}
}  // namespace

int kz_sort_pairs_f32_i32(kz_ctx* ctx, const float* keys_in, float* keys_out, const int* vals_in, int* vals_out, int n, int descending) {
    KZ_REQUIRE(ctx && (n == 0 || (keys_in && keys_out && vals_in && vals_out)), "kz_sort_pairs_f32_i32: null argument");
    KZ_REQUIRE(n >= 0 && (const void*)keys_in != (const void*)keys_out && vals_in != vals_out, "kz_sort_pairs_f32_i32: the sort is not in place");
    if (n == 0) return KZ_OK;
    const int n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    // ping-pong: pass 0  in -> tmp, pass 1  tmp -> out, pass 2  out -> tmp, pass 3  tmp -> out
    unsigned* tmp_k = nullptr;
    int *tmp_v = nullptr, *hist = nullptr;
    int rc = kz_pool_alloc(ctx, (size_t)n * 4, (void**)&tmp_k);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, (size_t)n * 4, (void**)&tmp_v);
    if (rc == KZ_OK) rc = kz_pool_alloc(ctx, ((size_t)256 * n_tiles + 256) * 4, (void**)&hist);   // counters [digit][tile] + 256 digit totals
    if (rc == KZ_OK) {
        const dim3 grid((unsigned)n_tiles), block(SORT_THREADS);
        int* totals = hist + (size_t)256 * n_tiles;
        const int desc = descending ? 1 : 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 8 * pass;
            const void* kin = pass == 0 ? (const void*)keys_in : (pass & 1 ? (const void*)tmp_k : (const void*)keys_out);
            const int* vin = pass == 0 ? vals_in : (pass & 1 ? tmp_v : vals_out);
            void* kout = (pass & 1) ? (void*)keys_out : (void*)tmp_k;
            int* vout = (pass & 1) ? vals_out : tmp_v;
            if (pass == 0)
                hipLaunchKernelGGL(kz_sort_hist_kernel<true>, grid, block, 0, ctx->stream, kin, n, shift, desc, n_tiles, hist);
            else
                hipLaunchKernelGGL(kz_sort_hist_kernel<false>, grid, block, 0, ctx->stream, kin, n, shift, desc, n_tiles, hist);
            hipLaunchKernelGGL(kz_sort_scan_kernel, dim3(256), block, 0, ctx->stream, hist, n_tiles, totals);
            if (pass == 0)
                hipLaunchKernelGGL((kz_sort_scatter_kernel<true, false>), grid, block, 0, ctx->stream, kin, vin, n, shift, desc, n_tiles, hist, totals, kout, vout);
            else if (pass == 3)
                hipLaunchKernelGGL((kz_sort_scatter_kernel<false, true>), grid, block, 0, ctx->stream, kin, vin, n, shift, desc, n_tiles, hist, totals, kout, vout);
            else
                hipLaunchKernelGGL((kz_sort_scatter_kernel<false, false>), grid, block, 0, ctx->stream, kin, vin, n, shift, desc, n_tiles, hist, totals, kout, vout);
        }
        if (hipGetLastError() != hipSuccess) {
            kz_set_error("kz_sort_pairs_f32_i32: a launch failed");
            rc = KZ_ERR_HIP;
        }
    }
    kz_pool_free(ctx, hist, 0);   // stream-ordered pool: reuse is ordered behind the sort
    kz_pool_free(ctx, tmp_v, 0);
    kz_pool_free(ctx, tmp_k, 0);
    return rc;
}
