// Multi-GPU exchange step of the shared sweep (kiez_amd/distributed.py).
//
// With the source row-sharded over the ranks, HubnessReduction.fit's reverse search (kiez/hubness_reduction/base.py:37-42:
// every target row against ALL source rows) is the merge of the per-shard searches.  The single-GPU search orders
// neighbours by the exact float64 value (squared distance / cosine distance), ties by smaller row; the OUTPUT distances are
// rounded ((double)sqrtf((float)d2) for float32 + euclidean), so merging by them would break ties differently.  Hence:
//   kz_pair_values  -- the exact ordering value of given (query row, index row) pairs, the bits the re-rank computed;
//   kz_merge_topk   -- per row, the k smallest of `segs` sorted segments by (value, global row), payload carried along.
// MutualProximity 'empiric' (mutual_proximity.py:185-212) and DisSimLocal (dis_sim.py:96-107) need the reverse INDICES in
// exactly the single-GPU order; CSLS / LocalScaling / MP 'normal' only the distances (merged by the distances themselves).
#include "kz_common.h"

// one wave per query row; the row's K pairs one after the other through the canonical dot product
template <typename T>
__global__ __launch_bounds__(256) void kz_pair_values_kernel(const T* __restrict__ qraw, const double* __restrict__ qsqn,
                                                             int64_t q_begin, int64_t q_count, const T* __restrict__ yraw,
                                                             const double* __restrict__ ysqn, int64_t n_i, int d, int metric, double p,
                                                             const int64_t* __restrict__ ind, int K, double* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= q_count) return;
    const int64_t qrow = q_begin + r;
    const T* q = qraw + qrow * (int64_t)d;
    const double qs = qsqn[qrow];
    if (metric >= KZ_MANHATTAN) {
        // the Minkowski family: one lane per pair, terms in feature order (kz_common.h: kz_family_value_seq) -- the order of the
        // tiled distance kernel the search ranked by, so that the values that travel between GPUs ARE the search's values
        for (int c = threadIdx.x & 63; c < K; c += 64) {
            const int64_t yi = ind[r * (int64_t)K + c];
            val[r * (int64_t)K + c] = (yi >= 0 && yi < n_i) ? kz_family_value_seq<T>(q, yraw + yi * (int64_t)d, d, metric, p) : INFINITY;
        }
        return;
    }
    for (int c = 0; c < K; ++c) {
        const int64_t yi = ind[r * (int64_t)K + c];   // wave-uniform
        double v = INFINITY;
        if (yi >= 0 && yi < n_i) v = kz_exact_value<T>(q, yraw + yi * (int64_t)d, qs, ysqn[yi], d, metric, lane, p);
        if (lane == 0) val[r * (int64_t)K + c] = v;
    }
}

// (key, row) order; NaN keys never occur (distances of finite inputs)
__device__ __forceinline__ bool kz_kr_less(double ka, long long ia, double kb, long long ib) {
    return ka < kb || (ka == kb && ia < ib);
}

// One wave per row.  The row's M = segs * L entries sit in LDS; segment s = columns [s L, (s + 1) L), each sorted ascending
// by (key, row).  The rank of an entry in the merged order (key, row, segment, position) is its position in its own segment
// plus, for every other segment, the number of entries in front of it -- a binary search per segment instead of a compare
// with every entry.  Entries of rank < k go straight to their output slot.
__global__ __launch_bounds__(256) void kz_merge_topk_kernel(const double* __restrict__ key, const int64_t* __restrict__ ind,
                                                            const double* __restrict__ dist, int64_t n, int segs, int L, int k,
                                                            double* __restrict__ odist, int64_t* __restrict__ oind) {
    extern __shared__ __attribute__((aligned(16))) char mg_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * wpb + wave;
    if (r >= n) return;   // whole wave; only wave-level synchronisation below
    const int M = segs * L;
    double* sk = reinterpret_cast<double*>(mg_smem) + (size_t)wave * M;
    long long* si = reinterpret_cast<long long*>(mg_smem + (size_t)wpb * M * 8) + (size_t)wave * M;
    const double* krow = key + r * (int64_t)M;
    const int64_t* irow = ind ? ind + r * (int64_t)M : nullptr;
    for (int e = lane; e < M; e += 64) {
        sk[e] = krow[e];
        si[e] = irow ? (long long)irow[e] : 0ll;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < M; e += 64) {
        const int s = e / L, p = e - s * L;
        const double ke = sk[e];
        const long long ie = si[e];
        int rank = p;
        for (int o = 0; o < segs && rank < k; ++o) {
            if (o == s) continue;
            // entries of segment o in front of e: strictly smaller (key, row), and -- for the segments before s -- equal ones
            const double* ok = sk + o * L;
            const long long* oi = si + o * L;
            int lo = 0, hi = L;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const bool front = o < s ? !kz_kr_less(ke, ie, ok[mid], oi[mid]) : kz_kr_less(ok[mid], oi[mid], ke, ie);
                if (front) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            odist[r * (int64_t)k + rank] = dist ? dist[r * (int64_t)M + e] : ke;
            oind[r * (int64_t)k + rank] = (int64_t)ie;
        }
    }
}

extern "C" {

int kz_pair_values(kz_ctx* ctx, const kz_matrix* query, int64_t q_begin, int64_t q_count, const kz_matrix* index,
                   const int64_t* d_ind, int k, double* d_val) {
    KZ_REQUIRE(ctx && query && index && d_ind && d_val, "kz_pair_values: null argument");
    KZ_REQUIRE(query && index && !query->raw_only && !index->raw_only, "kz_pair_values: null or rows-only matrix");
    KZ_REQUIRE(query->d == index->d && query->dtype == index->dtype && query->metric == index->metric && query->mink_p == index->mink_p,
               "kz_pair_values: query/index mismatch (d %lld vs %lld)", (long long)query->d, (long long)index->d);
    KZ_REQUIRE(q_begin >= 0 && q_count >= 0 && q_begin + q_count <= query->n && k >= 1, "kz_pair_values: bad row range");
    KZ_HIP(hipSetDevice(ctx->device));
    if (q_count == 0) return KZ_OK;
    const dim3 grid((unsigned)((q_count + 3) / 4));
    if (query->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_pair_values_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)query->raw, query->sqn, q_begin,
                           q_count, (const float*)index->raw, index->sqn, index->n, (int)query->d, query->metric, query->mink_p, d_ind, k, d_val);
    else
        hipLaunchKernelGGL(kz_pair_values_kernel<double>, grid, dim3(256), 0, ctx->stream, (const double*)query->raw, query->sqn, q_begin,
                           q_count, (const double*)index->raw, index->sqn, index->n, (int)query->d, query->metric, query->mink_p, d_ind, k, d_val);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_merge_topk(kz_ctx* ctx, const double* d_key, const int64_t* d_ind, const double* d_dist, int64_t n, int segs, int seg_len,
                  int k, double* d_odist, int64_t* d_oind) {
    KZ_REQUIRE(ctx && d_key && d_odist && d_oind, "kz_merge_topk: null argument");
    KZ_REQUIRE(n >= 0 && segs >= 1 && seg_len >= 1 && (int64_t)segs * seg_len <= KZ_MERGE_MAX_ENTRIES,
               "kz_merge_topk: bad shape n=%lld segs=%d seg_len=%d (at most %d entries per row)", (long long)n, segs, seg_len,
               KZ_MERGE_MAX_ENTRIES);
    KZ_REQUIRE(k >= 1 && k <= segs * seg_len, "kz_merge_topk: k=%d must be in [1, %d]", k, segs * seg_len);
    KZ_HIP(hipSetDevice(ctx->device));
    if (n == 0) return KZ_OK;
    const int M = segs * seg_len;
    const int wpb = M <= 1024 ? 4 : (M <= 4096 ? 2 : 1);   // 16 bytes of LDS per entry and wave
    const int lds = wpb * M * 16;
    if (lds > 65536)
        KZ_HIP(hipFuncSetAttribute((const void*)kz_merge_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kz_merge_topk_kernel, dim3((unsigned)((n + wpb - 1) / wpb)), dim3(64 * wpb), lds, ctx->stream, d_key, d_ind,
                       d_dist, n, segs, seg_len, k, d_odist, d_oind);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

}  // extern "C"
