// Hubness-reduction rescaling kernels (gather / reduce over the [n, K] candidate arrays) and the final
// candidate sort.  Compiled with -ffp-contract=off: the reference evaluates these formulas with separate
// numpy operations in float64, so no fused multiply-add may be formed here.
//
//   CSLS            kiez/hubness_reduction/csls.py:85-96
//   LocalScaling    kiez/hubness_reduction/local_scaling.py:129-151
//   MutualProximity kiez/hubness_reduction/mutual_proximity.py:166-212
//   DisSimLocal     kiez/hubness_reduction/dis_sim.py:96-107, 139-181
//   _sort           kiez/hubness_reduction/base.py:72-87
#include "kz_common.h"

// numpy's pairwise summation of a contiguous float64 run (numpy/_core/src/umath/loops_utils.h.src,
// DOUBLE_pairwise_sum): < 8 elements sequential, <= 128 eight interleaved accumulators, else split.
__device__ double kz_np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
            r0 += a[i + 0];
            r1 += a[i + 1];
            r2 += a[i + 2];
            r3 += a[i + 3];
            r4 += a[i + 4];
            r5 += a[i + 5];
            r6 += a[i + 6];
            r7 += a[i + 7];
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return kz_np_pairwise_sum(a, n2) + kz_np_pairwise_sum(a + n2, n - n2);
}

// same summation applied to (a[i] - c)^2 (np.nanstd: subtract mean, square, sum; numpy/lib/_nanfunctions_impl.py)
__device__ double kz_np_pairwise_sumsq_dev(const double* a, int n, double c) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) {
            const double t = a[i] - c;
            res += t * t;
        }
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int u = 0; u < 8; ++u) {
            const double t = a[u] - c;
            r[u] = t * t;
        }
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
            for (int u = 0; u < 8; ++u) {
                const double t = a[i + u] - c;
                r[u] += t * t;
            }
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) {
            const double t = a[i] - c;
            res += t * t;
        }
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return kz_np_pairwise_sumsq_dev(a, n2, c) + kz_np_pairwise_sumsq_dev(a + n2, n - n2, c);
}

__global__ void kz_row_stats_kernel(const double* __restrict__ dist, int64_t n, int K, double* __restrict__ mean,
                                    double* __restrict__ sd, double* __restrict__ last) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* a = dist + r * (int64_t)K;
    if (last) last[r] = a[K - 1];
    if (mean || sd) {
        const double m = kz_np_pairwise_sum(a, K) / (double)K;
        if (mean) mean[r] = m;
        if (sd) sd[r] = sqrt(kz_np_pairwise_sumsq_dev(a, K, m) / (double)K);
    }
}

// CSLS: out = 2*d - mean_K(d[i,:]) - r_train[ind]      (csls.py:90-93)
__global__ void kz_csls_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind, int64_t n, int K,
                               const double* __restrict__ r_train, double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* a = dist + r * (int64_t)K;
    const int64_t* id = ind + r * (int64_t)K;
    const double r_test = kz_np_pairwise_sum(a, K) / (double)K;
    for (int c = 0; c < K; ++c) {
        double v = 2.0 * a[c];
        v = v - r_test;
        v = v - r_train[id[c]];
        out[r * (int64_t)K + c] = v;
    }
}

// LocalScaling (local_scaling.py:135-147)
__global__ void kz_ls_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind, int64_t n, int K,
                             const double* __restrict__ r_t, int nicdm, double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* a = dist + r * (int64_t)K;
    const int64_t* id = ind + r * (int64_t)K;
    if (nicdm) {
        const double r_s = kz_np_pairwise_sum(a, K) / (double)K;
        for (int c = 0; c < K; ++c) out[r * (int64_t)K + c] = a[c] / sqrt(r_s * r_t[id[c]]);
    } else {
        const double r_s = a[K - 1];
        for (int c = 0; c < K; ++c) {
            const double d = a[c];
            const double inner = (-1.0 * (d * d)) / (r_s * r_t[id[c]]);
            out[r * (int64_t)K + c] = 1.0 - exp(inner);
        }
    }
}

// scipy.stats.norm.sf(x, loc, scale) = ndtr(-(x-loc)/scale), cephes ndtr (scipy/special/xsf/cephes/ndtr.h)
__device__ __forceinline__ double kz_ndtr(double a) {
    const double SQRT1_2 = 0.70710678118654752440;
    if (isnan(a)) return a;
    const double x = a * SQRT1_2;
    const double z = fabs(x);
    if (z < SQRT1_2) return 0.5 + 0.5 * erf(x);
    double y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
    return y;
}

__global__ void kz_mp_normal_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind, int64_t n, int K,
                                    const double* __restrict__ mu_t, const double* __restrict__ sd_t,
                                    double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* a = dist + r * (int64_t)K;
    const int64_t* id = ind + r * (int64_t)K;
    const double mu = kz_np_pairwise_sum(a, K) / (double)K;
    const double sd = sqrt(kz_np_pairwise_sumsq_dev(a, K, mu) / (double)K);
    for (int c = 0; c < K; ++c) {
        const double d = a[c];
        const double p1 = kz_ndtr(-((d - mu) / sd));
        const int64_t t = id[c];
        const double p2 = kz_ndtr(-((d - mu_t[t]) / sd_t[t]));
        out[r * (int64_t)K + c] = 1.0 - p1 * p2;
    }
}

// MutualProximity empiric (mutual_proximity.py:185-212); one wave per query row.
//   out[i,j] = 1 - #{m : d[i,m] > d[i,j] and T_j[m] > d[i,j]} / K
//   T_j[m]   = dist_t2s[c_j, p] if ind_t2s[c_j, p] == c_m (c_m is a TARGET id matched against SOURCE ids: the
//              reference's behaviour) else dist_t2s[c_j, Kt-1] + 1e-6
// The K candidate ids of the query go into a hash table in LDS once; for every j the Kt reverse-list ids of c_j are looked
// up in it (K Kt probes instead of K^2 Kt id compares) and their distances scattered into T.
constexpr int KZ_MP_MAXK = 128;

__global__ __launch_bounds__(256) void kz_mp_empiric_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind,
                                                            int64_t n, int K, const double* __restrict__ dist_t2s,
                                                            const int64_t* __restrict__ ind_t2s, int64_t n_t, int Kt,
                                                            double* __restrict__ out) {
    __shared__ long long s_cand[4][KZ_MP_MAXK];   // candidate ids, by candidate
    __shared__ int s_tab[4][2 * KZ_MP_MAXK];      // open-addressing hash table over them: slot -> candidate, -1 = empty
    __shared__ double s_T[4][KZ_MP_MAXK];         // T_j, by candidate
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= n) return;  // whole wave; only wave-level synchronisation below
    const double* d_i = dist + i * (int64_t)K;
    const int64_t* c_i = ind + i * (int64_t)K;
    long long* cand = s_cand[wave];
    int* tab = s_tab[wave];
    double* T = s_T[wave];
    constexpr unsigned HMASK = 2 * KZ_MP_MAXK - 1;
    auto slot_of = [](const long long id) {
        return (((unsigned)(unsigned long long)id ^ (unsigned)((unsigned long long)id >> 32)) * 0x9E3779B1u) >> 24;   // (8 bits: 256 slots)
    };
    static_assert(2 * KZ_MP_MAXK == 256, "slot_of keeps 8 bits");
    // this lane's candidates m = lane, lane + 64: distance and id
    double dm[2] = {0.0, 0.0};
    long long cm[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int m = lane + 64 * u;
        if (m < K) {
            dm[u] = d_i[m];
            cm[u] = c_i[m];
            cand[m] = cm[u];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) tab[lane + 64 * t] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // The candidate ids go into a hash table (at most 128 keys in 256 slots, linear probing): a look-up is one or two probes
    // instead of the log2 K steps of a binary search over the sorted ids -- the kernel is bound by its instruction count (K
    // look-ups of Kt ids per query), not by the lists it gathers (profiles/r04_ablation.md section 11) -- and nothing has to be
    // sorted first (that was K dependent loads of c_i[o] per query).  (The ids of one kNN row are distinct.)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int m = lane + 64 * u;
        if (m < K) {
            unsigned h = slot_of(cm[u]);
            while (atomicCAS(&tab[h], -1, m) != -1) h = (h + 1) & HMASK;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the reverse list of candidate j + 1 (its ids: one or two per lane; the last distance for the fill value) is fetched while
    // candidate j is processed: the loop is a chain of dependent gathers otherwise (one wave per query, 50 round trips to HBM)
    long long idn[2] = {-1, -1};
    double lastn = 0.0;
    const double* rdn = nullptr;
    auto prefetch = [&](const int jn) {
        const int64_t cn = c_i[jn];
        rdn = dist_t2s + cn * (int64_t)Kt;
        const int64_t* rin = ind_t2s + cn * (int64_t)Kt;
        lastn = rdn[Kt - 1];
#pragma unroll
        for (int u = 0; u < 2; ++u) idn[u] = lane + 64 * u < Kt ? rin[lane + 64 * u] : -1;
    };
    prefetch(0);
    for (int j = 0; j < K; ++j) {
        const double dj = d_i[j];
        const double* rd = rdn;
        const double fill = lastn + 1e-6;
        const long long idc[2] = {idn[0], idn[1]};
        if (j + 1 < K) prefetch(j + 1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (lane + 64 * u < K) T[lane + 64 * u] = fill;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // every reverse-list entry p looks its id up among the candidate ids
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pp = lane + 64 * u;
            if (pp >= Kt) continue;
            const long long id = idc[u];
            unsigned h = slot_of(id);
            int pos = -1;
            for (;;) {
                const int m = tab[h];
                if (m < 0) break;
                if (cand[m] == id) {
                    pos = m;
                    break;
                }
                h = (h + 1) & HMASK;
            }
            if (pos >= 0) T[pos] = rd[pp];  // ids inside one kNN row are distinct: at most one writer per position
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (a count over the wave: two ballots, no cross-lane exchange through LDS)
        const int cnt = (int)__popcll(__ballot(lane < K && dm[0] > dj && T[lane] > dj)) +
                        (int)__popcll(__ballot(lane + 64 < K && dm[1] > dj && T[(lane + 64) & (KZ_MP_MAXK - 1)] > dj));
        if (lane == 0) out[i * (int64_t)K + j] = 1.0 - (double)cnt / (double)K;
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for K > 128 candidates per query (no per-lane register copies; ids, their ranks and T in dynamic LDS, 20 bytes per
// candidate and wave).  The reference has no limit on n_candidates (kiez/hubness_reduction/base.py:20-27).
__global__ __launch_bounds__(256) void kz_mp_empiric_wide_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind,
                                                                 int64_t n, int K, const double* __restrict__ dist_t2s,
                                                                 const int64_t* __restrict__ ind_t2s, int64_t n_t, int Kt,
                                                                 double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char mpw_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * wpb + wave;
    if (i >= n) return;  // whole wave; only wave-level synchronisation below
    long long* sorted = reinterpret_cast<long long*>(mpw_smem) + (size_t)wave * K;
    double* T = reinterpret_cast<double*>(mpw_smem + (size_t)wpb * K * 8) + (size_t)wave * K;
    int* rk = reinterpret_cast<int*>(mpw_smem + (size_t)wpb * K * 16) + (size_t)wave * K;
    const double* d_i = dist + i * (int64_t)K;
    const int64_t* c_i = ind + i * (int64_t)K;
    for (int m = lane; m < K; m += 64) {
        const long long cm = c_i[m];
        int r = 0;
        for (int o = 0; o < K; ++o) {
            const long long co = c_i[o];
            r += (co < cm || (co == cm && o < m)) ? 1 : 0;
        }
        rk[m] = r;
        sorted[r] = cm;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = 0; j < K; ++j) {
        const double dj = d_i[j];
        const int64_t cj = c_i[j];
        const double* rd = dist_t2s + cj * (int64_t)Kt;
        const int64_t* ri = ind_t2s + cj * (int64_t)Kt;
        const double fill = rd[Kt - 1] + 1e-6;
        for (int m = lane; m < K; m += 64) T[m] = fill;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int pp = lane; pp < Kt; pp += 64) {
            const long long id = ri[pp];
            int lo = 0, hi = K - 1, pos = -1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const long long v = sorted[mid];
                if (v == id) {
                    pos = mid;
                    break;
                }
                if (v < id) lo = mid + 1; else hi = mid - 1;
            }
            if (pos >= 0) T[pos] = rd[pp];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int cnt = 0;
        for (int m = lane; m < K; m += 64)
            if (d_i[m] > dj && T[rk[m]] > dj) ++cnt;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == 0) out[i * (int64_t)K + j] = 1.0 - (double)cnt / (double)K;
        __builtin_amdgcn_wave_barrier();
    }
}

// DisSimLocal fit (dis_sim.py:96-102): t2c[j] = | target[t_begin+j] - mean_m source[ind_t2s[j,m]] |^2 ; wave per row
template <typename T>
__global__ __launch_bounds__(256) void kz_dsl_fit_kernel(const int64_t* __restrict__ ind_t2s, int64_t n_rows, int Kt,
                                                         const T* __restrict__ source, const T* __restrict__ target,
                                                         int64_t t_begin, int d, double* __restrict__ t2c) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= n_rows) return;
    const int64_t* id = ind_t2s + r * (int64_t)Kt;
    const T* t = target + (t_begin + r) * (int64_t)d;
    double acc = 0.0;
    for (int k = lane; k < d; k += 64) {
        double c = 0.0;
        for (int m = 0; m < Kt; ++m) c += (double)source[id[m] * (int64_t)d + k];
        c = c / (double)Kt;
        const double df = (double)t[k] - c;
        acc += df * df;
    }
    acc = kz_wave_sum(acc);
    if (lane == 0) t2c[r] = acc;
}

__device__ __forceinline__ void kz_atomic_min_double(double* addr, double v) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    unsigned long long old = *a;
    while (v < __longlong_as_double((long long)old)) {
        const unsigned long long assumed = old;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
        if (old == assumed) break;
    }
}

// DisSimLocal transform (dis_sim.py:153-166): out[i,m] = |q_i - t[c_m]|^2 - |q_i - mean_m t[c_m]|^2 - t2c[c_m]
template <typename T>
__global__ __launch_bounds__(256) void kz_dsl_transform_kernel(const int64_t* __restrict__ ind, int64_t n, int K,
                                                               const T* __restrict__ query, int64_t q_begin,
                                                               const T* __restrict__ target, int d,
                                                               const double* __restrict__ t2c, double* __restrict__ out,
                                                               double* __restrict__ gmin) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= n) return;
    const int64_t* id = ind + i * (int64_t)K;
    const T* q = query + (q_begin + i) * (int64_t)d;
    double s2c = 0.0;
    for (int k = lane; k < d; k += 64) {
        double c = 0.0;
        for (int m = 0; m < K; ++m) c += (double)target[id[m] * (int64_t)d + k];
        c = c / (double)K;
        const double df = (double)q[k] - c;
        s2c += df * df;
    }
    s2c = kz_wave_sum(s2c);
    double wmin = INFINITY;
    for (int m = 0; m < K; ++m) {
        const T* t = target + id[m] * (int64_t)d;
        double acc = 0.0;
        for (int k = lane; k < d; k += 64) {
            const double df = (double)q[k] - (double)t[k];
            acc += df * df;
        }
        acc = kz_wave_sum(acc);
        double v = acc - s2c;
        v = v - t2c[id[m]];
        if (lane == 0) out[i * (int64_t)K + m] = v;
        wmin = fmin(wmin, v);
    }
    if (lane == 0) kz_atomic_min_double(gmin, wmin);
}

__global__ void kz_dsl_finalize_kernel(double* __restrict__ out, int64_t count, double shift, int squared) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    double v = out[e] + shift;
    if (!squared) v = sqrt(v);  // numpy `**= 1/2` on float64 is sqrt-accurate for the tolerance we state
    out[e] = v;
}

// HubnessReduction._sort (base.py:81-86): np.argpartition(kth=arange(k)) == selection sort with swaps of the first k
// positions: for i < k pick the FIRST minimum of positions [i, K) and swap it into position i.  The swap
// history decides the order of tied values (MP-empiric rows are full of ties), so the sort is emulated step by step --
// but by ONE LANE PER ROW on a transposed copy of 64 rows in LDS (values [pos][row], original positions as bytes):
// ~5 instructions per compared element and no cross-lane traffic, against a 6-step shuffle butterfly per selection in the
// wave-per-row form (C3, 500k x 50: 6.4 ms -> see profiles/).  Loads and stores are coalesced through the LDS copy;
// neighbour ids are gathered at the end through the sorted positions.  NaN sorts last (numpy's order), so a row with NaN
// (MP-normal with sd = 0, LS / NICDM with radius 0) stays a permutation.
constexpr int KZ_SEL_ROWS = 64;                      // rows per workgroup (one wave)
constexpr int KZ_SEL_LD = KZ_SEL_ROWS + 1;           // padded leading dimension of the value image (doubles)
__host__ __device__ constexpr int kz_sel_lds_bytes(int K) { return K * KZ_SEL_LD * 8 + K * KZ_SEL_ROWS; }

__global__ __launch_bounds__(64) void kz_select_topk_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind,
                                                            int64_t n, int K, int k, double* __restrict__ odist,
                                                            int64_t* __restrict__ oind) {
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    double* v = reinterpret_cast<double*>(sel_smem);                            // v[pos * LD + row]
    unsigned char* ps = reinterpret_cast<unsigned char*>(sel_smem + (size_t)K * KZ_SEL_LD * 8);   // ps[pos * 64 + row]
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * KZ_SEL_ROWS;
    const int rows = (int)((n - row0) < KZ_SEL_ROWS ? (n - row0) : KZ_SEL_ROWS);
    // coalesced load of rows x K values, transposed into LDS
    const double* src = dist + row0 * (int64_t)K;
    const int total = rows * K;
    for (int e = lane; e < total; e += 64) {
        const int r = e / K, j = e - r * K;
        v[j * KZ_SEL_LD + r] = src[e];
    }
    for (int j = 0; j < K; ++j) ps[j * KZ_SEL_ROWS + lane] = (unsigned char)j;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < rows) {
        double* vr = v + lane;
        unsigned char* pr = ps + lane;
        for (int i = 0; i < k; ++i) {
            double m = vr[i * KZ_SEL_LD];
            int mp = i;
            for (int j = i + 1; j < K; ++j) {
                const double x = vr[j * KZ_SEL_LD];
                // total order with NaN last; strict: the first minimum wins
                const bool less = (x < m) || ((m != m) && (x == x));
                m = less ? x : m;
                mp = less ? j : mp;
            }
            if (mp != i) {
                const double vi = vr[i * KZ_SEL_LD];
                vr[mp * KZ_SEL_LD] = vi;
                vr[i * KZ_SEL_LD] = m;
                const unsigned char a = pr[i * KZ_SEL_ROWS], b = pr[mp * KZ_SEL_ROWS];
                pr[i * KZ_SEL_ROWS] = b;
                pr[mp * KZ_SEL_ROWS] = a;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // coalesced store of rows x k results; ids gathered through the sorted positions
    const int out_total = rows * k;
    double* od = odist + row0 * (int64_t)k;
    int64_t* oi = oind + row0 * (int64_t)k;
    const int64_t* id = ind + row0 * (int64_t)K;
    for (int e = lane; e < out_total; e += 64) {
        const int r = e / k, i = e - r * k;
        od[e] = v[i * KZ_SEL_LD + r];
        oi[e] = id[(int64_t)r * K + ps[i * KZ_SEL_ROWS + r]];
    }
}

// The same selection sort for rows of more than 128 candidates: one WAVE per row, the row in LDS (12 bytes per candidate).
// Step i: every lane scans its strided share of positions [i, K) for its first minimum (NaN last), a butterfly picks the
// smallest value at the lowest position -- the first minimum of the whole range -- and lane 0 swaps it into position i.
__global__ __launch_bounds__(256) void kz_select_topk_wide_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind,
                                                                  int64_t n, int K, int k, double* __restrict__ odist,
                                                                  int64_t* __restrict__ oind) {
    extern __shared__ __attribute__((aligned(16))) char selw_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * wpb + wave;
    if (r >= n) return;  // whole wave; only wave-level synchronisation below
    double* v = reinterpret_cast<double*>(selw_smem) + (size_t)wave * K;
    int* ps = reinterpret_cast<int*>(selw_smem + (size_t)wpb * K * 8) + (size_t)wave * K;
    for (int j = lane; j < K; j += 64) {
        v[j] = dist[r * (int64_t)K + j];
        ps[j] = j;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = 0; i < k; ++i) {
        double m = NAN;
        int mp = 0x7fffffff;
        for (int j = i + lane; j < K; j += 64) {
            const double x = v[j];
            const bool less = mp == 0x7fffffff || (x < m) || ((m != m) && (x == x));
            m = less ? x : m;
            mp = less ? j : mp;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double om = __shfl_xor(m, off, 64);
            const int omp = __shfl_xor(mp, off, 64);
            // the other lane's candidate wins if it is smaller (NaN last), or equal / both NaN and at a lower position
            const bool other_less = (om < m) || ((m != m) && (om == om));
            const bool same = (om == m) || ((m != m) && (om != om));
            const bool take = omp != 0x7fffffff && (mp == 0x7fffffff || other_less || (same && omp < mp));
            m = take ? om : m;
            mp = take ? omp : mp;
        }
        if (lane == 0 && mp != i) {
            const double vi = v[i];
            v[mp] = vi;
            v[i] = m;
            const int a = ps[i], b = ps[mp];
            ps[i] = b;
            ps[mp] = a;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (int i = lane; i < k; i += 64) {
        odist[r * (int64_t)k + i] = v[i];
        oind[r * (int64_t)k + i] = ind[r * (int64_t)K + ps[i]];
    }
}

__global__ void kz_cast_f64_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t count) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = (float)in[e];
}

// ---------------------------------------------------------------------------------------------------
static inline dim3 kz_grid1d(int64_t n, int per_block) { return dim3((unsigned)((n + per_block - 1) / per_block)); }

#define KZ_CHECK_NK(fn)                                                                       \
    KZ_REQUIRE(ctx != nullptr, fn ": null context");                                           \
    KZ_REQUIRE(n >= 0 && K >= 1 && K <= KZ_MAX_CANDIDATES, fn ": bad shape n=%lld K=%d (K must be in [1,%d])", (long long)n, K, KZ_MAX_CANDIDATES); \
    KZ_HIP(hipSetDevice(ctx->device));                                                         \
    if (n == 0) return KZ_OK;

extern "C" {

int kz_row_stats(kz_ctx* ctx, const double* d_dist, int64_t n, int K, double* d_mean, double* d_std, double* d_last) {
    KZ_CHECK_NK("kz_row_stats");
    KZ_REQUIRE(d_dist != nullptr, "kz_row_stats: null input");
    hipLaunchKernelGGL(kz_row_stats_kernel, kz_grid1d(n, 256), dim3(256), 0, ctx->stream, d_dist, n, K, d_mean, d_std, d_last);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_csls(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_r_train, double* d_out) {
    KZ_CHECK_NK("kz_csls");
    KZ_REQUIRE(d_dist && d_ind && d_r_train && d_out, "kz_csls: null argument");
    hipLaunchKernelGGL(kz_csls_kernel, kz_grid1d(n, 256), dim3(256), 0, ctx->stream, d_dist, d_ind, n, K, d_r_train, d_out);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_local_scaling(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_r_t, int nicdm,
                     double* d_out) {
    KZ_CHECK_NK("kz_local_scaling");
    KZ_REQUIRE(d_dist && d_ind && d_r_t && d_out, "kz_local_scaling: null argument");
    hipLaunchKernelGGL(kz_ls_kernel, kz_grid1d(n, 256), dim3(256), 0, ctx->stream, d_dist, d_ind, n, K, d_r_t, nicdm, d_out);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_mp_normal(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_mu_t,
                 const double* d_sd_t, double* d_out) {
    KZ_CHECK_NK("kz_mp_normal");
    KZ_REQUIRE(d_dist && d_ind && d_mu_t && d_sd_t && d_out, "kz_mp_normal: null argument");
    hipLaunchKernelGGL(kz_mp_normal_kernel, kz_grid1d(n, 256), dim3(256), 0, ctx->stream, d_dist, d_ind, n, K, d_mu_t, d_sd_t,
                       d_out);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_mp_empiric(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, const double* d_dist_t2s,
                  const int64_t* d_ind_t2s, int64_t n_t, int Kt, double* d_out) {
    KZ_CHECK_NK("kz_mp_empiric");
    KZ_REQUIRE(d_dist && d_ind && d_dist_t2s && d_ind_t2s && d_out, "kz_mp_empiric: null argument");
    KZ_REQUIRE(Kt >= 1 && n_t >= 1, "kz_mp_empiric: bad reverse list shape");
    if (K <= KZ_MP_MAXK && Kt <= KZ_MP_MAXK) {   // (one or two candidates / reverse-list entries per lane)
        hipLaunchKernelGGL(kz_mp_empiric_kernel, kz_grid1d(n, 4), dim3(256), 0, ctx->stream, d_dist, d_ind, n, K, d_dist_t2s,
                           d_ind_t2s, n_t, Kt, d_out);
    } else {
        const int wpb = K <= 1024 ? 4 : 1;
        const int lds = wpb * K * 20;
        if (lds > 65536)
            KZ_HIP(hipFuncSetAttribute((const void*)kz_mp_empiric_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kz_mp_empiric_wide_kernel, kz_grid1d(n, wpb), dim3(64 * wpb), lds, ctx->stream, d_dist, d_ind, n, K,
                           d_dist_t2s, d_ind_t2s, n_t, Kt, d_out);
    }
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_dsl_fit(kz_ctx* ctx, const int64_t* d_ind_t2s, int64_t n_rows, int Kt, const kz_matrix* source, const kz_matrix* target,
               int64_t t_begin, double* d_t2c) {
    KZ_REQUIRE(ctx && d_ind_t2s && source && target && d_t2c, "kz_dsl_fit: null argument");
    KZ_REQUIRE(source->d == target->d && source->dtype == target->dtype, "kz_dsl_fit: source/target mismatch");
    KZ_REQUIRE(n_rows >= 0 && t_begin >= 0 && t_begin + n_rows <= target->n && Kt >= 1, "kz_dsl_fit: bad row range");
    KZ_HIP(hipSetDevice(ctx->device));
    if (n_rows == 0) return KZ_OK;
    if (source->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_dsl_fit_kernel<float>, kz_grid1d(n_rows, 4), dim3(256), 0, ctx->stream, d_ind_t2s, n_rows, Kt,
                           (const float*)source->raw, (const float*)target->raw, t_begin, (int)source->d, d_t2c);
    else
        hipLaunchKernelGGL(kz_dsl_fit_kernel<double>, kz_grid1d(n_rows, 4), dim3(256), 0, ctx->stream, d_ind_t2s, n_rows, Kt,
                           (const double*)source->raw, (const double*)target->raw, t_begin, (int)source->d, d_t2c);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_dsl_transform(kz_ctx* ctx, const int64_t* d_ind, int64_t n, int K, const kz_matrix* query, int64_t q_begin,
                     const kz_matrix* target, const double* d_t2c, double* d_out, double* d_min) {
    KZ_CHECK_NK("kz_dsl_transform");
    KZ_REQUIRE(d_ind && query && target && d_t2c && d_out && d_min, "kz_dsl_transform: null argument");
    KZ_REQUIRE(query->d == target->d && query->dtype == target->dtype, "kz_dsl_transform: query/target mismatch");
    KZ_REQUIRE(q_begin >= 0 && q_begin + n <= query->n, "kz_dsl_transform: bad row range");
    if (query->dtype == KZ_F32)
        hipLaunchKernelGGL(kz_dsl_transform_kernel<float>, kz_grid1d(n, 4), dim3(256), 0, ctx->stream, d_ind, n, K,
                           (const float*)query->raw, q_begin, (const float*)target->raw, (int)query->d, d_t2c, d_out, d_min);
    else
        hipLaunchKernelGGL(kz_dsl_transform_kernel<double>, kz_grid1d(n, 4), dim3(256), 0, ctx->stream, d_ind, n, K,
                           (const double*)query->raw, q_begin, (const double*)target->raw, (int)query->d, d_t2c, d_out, d_min);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_dsl_finalize(kz_ctx* ctx, double* d_out, int64_t count, double min_value, int squared) {
    KZ_REQUIRE(ctx && d_out && count >= 0, "kz_dsl_finalize: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    if (count == 0) return KZ_OK;
    const double shift = (min_value < 0.0) ? -min_value : 0.0;  // dis_sim.py:171-173, _MINIMUM_DIST = 0.0
    hipLaunchKernelGGL(kz_dsl_finalize_kernel, kz_grid1d(count, 256), dim3(256), 0, ctx->stream, d_out, count, shift, squared);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_select_topk(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K, int k, double* d_odist,
                   int64_t* d_oind) {
    KZ_CHECK_NK("kz_select_topk");
    KZ_REQUIRE(d_dist && d_ind && d_odist && d_oind, "kz_select_topk: null argument");
    KZ_REQUIRE(k >= 1 && k <= K, "kz_select_topk: k=%d must be in [1, K=%d]", k, K);
    if (K <= 128) {
        const int lds = kz_sel_lds_bytes(K);
        if (lds > 65536) KZ_HIP(hipFuncSetAttribute((const void*)kz_select_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kz_select_topk_kernel, kz_grid1d(n, KZ_SEL_ROWS), dim3(64), lds, ctx->stream, d_dist, d_ind, n, K, k, d_odist,
                           d_oind);
    } else {
        const int wpb = K <= 1024 ? 4 : 1;
        const int lds = wpb * K * 12;
        hipLaunchKernelGGL(kz_select_topk_wide_kernel, kz_grid1d(n, wpb), dim3(64 * wpb), lds, ctx->stream, d_dist, d_ind, n, K, k,
                           d_odist, d_oind);
    }
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

// Single-source mode: the reverse search of HubnessReduction.fit (explicit query = the source itself: every row keeps itself
// as its first neighbour, base.py:37-42) and the forward search of kneighbors (query = None: the row itself is stripped the way
// sklearn does, neighbors/_base.py:937-965) are both views of ONE search for K + 1 neighbours without stripping: the reverse
// lists are its first K columns, the forward lists are the row minus the entry whose index is the row (or minus the first
// entry when the row is not among its own K + 1) -- exactly what kz_knn's exclude_self does on its K + 1 candidates.
__global__ __launch_bounds__(256) void kz_split_self_kernel(const double* __restrict__ dist, const int64_t* __restrict__ ind, int64_t n,
                                                            int K1, int64_t row0, double* __restrict__ rev_d,
                                                            int64_t* __restrict__ rev_i, double* __restrict__ fwd_d,
                                                            int64_t* __restrict__ fwd_i) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int K = K1 - 1;
    const double* d = dist + r * K1;
    const int64_t* i = ind + r * K1;
    int self_rank = 0;
    for (int c = 0; c < K1; ++c)
        if (i[c] == row0 + r) {
            self_rank = c;
            break;
        }
    for (int c = 0; c < K; ++c) {
        rev_d[r * K + c] = d[c];
        rev_i[r * K + c] = i[c];
        const int s = c < self_rank ? c : c + 1;
        fwd_d[r * K + c] = d[s];
        fwd_i[r * K + c] = i[s];
    }
}

int kz_split_self(kz_ctx* ctx, const double* d_dist, const int64_t* d_ind, int64_t n, int K1, int64_t row0, double* d_rev_dist,
                  int64_t* d_rev_ind, double* d_fwd_dist, int64_t* d_fwd_ind) {
    KZ_REQUIRE(ctx && d_dist && d_ind && d_rev_dist && d_rev_ind && d_fwd_dist && d_fwd_ind, "kz_split_self: null argument");
    KZ_REQUIRE(n >= 0 && K1 >= 2, "kz_split_self: need n >= 0 and at least two columns");
    KZ_HIP(hipSetDevice(ctx->device));
    if (n == 0) return KZ_OK;
    hipLaunchKernelGGL(kz_split_self_kernel, kz_grid1d(n, 256), dim3(256), 0, ctx->stream, d_dist, d_ind, n, K1, row0, d_rev_dist, d_rev_ind,
                       d_fwd_dist, d_fwd_ind);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

int kz_cast_f64_f32(kz_ctx* ctx, const double* d_in, float* d_out, int64_t count) {
    KZ_REQUIRE(ctx && d_in && d_out && count >= 0, "kz_cast_f64_f32: bad argument");
    KZ_HIP(hipSetDevice(ctx->device));
    if (count == 0) return KZ_OK;
    hipLaunchKernelGGL(kz_cast_f64_f32_kernel, kz_grid1d(count, 256), dim3(256), 0, ctx->stream, d_in, d_out, count);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

}  // extern "C"
