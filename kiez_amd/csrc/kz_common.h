// Internal definitions shared by the HIP translation units of libkiez_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/kiez_amd.h"

#define KZ_TILE 128   // rows per packed tile (= MFMA block tile edge)
#define KZ_KSLICE 16  // k elements staged per LDS slice (4 k-groups of 4)
#define KZ_LIVE_MAX 1024
#define KZ_POOL_SLOTS 256   // cached released buffers (a shared-sweep fit + kneighbors releases ~90: 64 slots evicted ~28 a step -- hipFree, a device-wide sync each)

// Former context options that no tool or test set (round 6: one table of <= 30 options): the values every build ran with.
constexpr int KZ_K_LDS_PAD = 0;   // extra dynamic LDS per workgroup (lowers occupancy; diagnostic)
constexpr int KZ_K_LONG_K = 1;   // 111 .. ~540 neighbours on the fused kernels (0: exact kernels)
constexpr int KZ_K_FIN_WIDE = 1;   // finalize of > 160 selected candidates without O(n^2) sorts, several rows per gather step
constexpr int KZ_K_RANGE_BOOT = 1;   // short-list routes: index range 0 first, the other ranges' lists start at the floor read off it
constexpr double KZ_K_NESTED_MIN_MS = 2.0;   // the nested sample of the shared sweep is taken when it saves at least this many model-ms (sweep / stride)
constexpr int KZ_K_DUAL_SHORT_DIV = 5;   // short-list routes: one list of 16 per this many neighbours (k / 5 lists)
constexpr int KZ_K_DUAL_SHORT_KP = 16;   // ... of this length
constexpr int KZ_K_MIN_SPLITS = 1;   // minimum index splits per query tile in the large-item region
constexpr int KZ_K_QGROUP = 0;   // query tiles per group of the work table (0 = automatic)
constexpr int KZ_K_H_WIDE = 0;   // fp16 kernel: wide workgroups on one ring (kz_knn_h16.h "WIDE": measured slower on every shared sweep, round 6: 250 k x 1 M x 300 149.2 -> 151.4 ms)
constexpr int KZ_K_H_WPS = 0;   // fp16 kernel: workgroups per CU (0 = automatic)
constexpr double KZ_K_PROBE_MIN_MS = 12.0;   // searches below "probe_min_pairs" distance pairs take neither the tier probe nor a floor -- unless the sweep is at least this many model-ms (2 n_q n_i d / 1e12) long
constexpr int KZ_K_EXACT_DIRECT_ROWS = 32;   // at most this many rows left by the split-bf16 tier skip the float32-operand kernel and go to the exact kernels
constexpr double KZ_K_SPEC_ELEMS = 1.6e9;   // speculative rescue: at most this / (index rows x d) rows (and at most "spec_rows")
constexpr int KZ_K_FLOOR_PROBE = 1024;   // seeded lists: rows of the probe in kz_knn_dual
constexpr int KZ_K_DUAL_DEAL = 1;   // query rows dealt into load-balanced tiles
constexpr int KZ_K_ESC_SHORT = 1;   // rows a K' = 16 pass cannot certify: more lists of 16 instead of lists of 64

struct kz_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    hipEvent_t ev[12];   // [0..7] work on `stream`, [8..11] the reverse direction of kz_knn_dual on `stream2`
    int stream2_busy;    // kz_knn_dual: the reverse chain is (or is about to be) on stream2 -- nobody else may queue behind it
    hipStream_t stream2; // second stream: kz_knn_dual runs the reverse direction's event chain beside the forward direction's finalize
    double eps_scale;
    int force_splits;
    int chunk_rows;   // test knob: query rows per chunk (0 = default 524288)
    int precision;    // 0: fp16 first pass where eligible (default), 2: split-bf16 first pass, 1: float32 operands only
    int dual_stride;  // kz_knn_dual: every dual_stride-th tile of the query side is in the threshold sample (0: no dual pass, 1: automatic)
    int esc_bf;            // 1 (default): rows the fp16 tier cannot certify with its longest lists go to the split-bf16 operands before the float32 ones
    int short_ord_min_tiles;   // ... when an index range has at least this many tiles (default 48)
    int short_ord;         // 1 (default): the ordinary fp16 kernel takes the short-list route too (a dealt second image of the index)
    int dual_short_main;   // 1 (default): the main sweep of kz_knn_dual keeps k / dual_short_div lists of 16 per query instead of one of 32 / 64; 0: one list of K'
    int dual_rev_long;     // 1 (default): the reverse direction of kz_knn_dual keeps lists of 2 K'
    int dual_short_extra;  // ... of whose entries the finalize kernel selects k + this many (default 48)
    int dual_short_min_tiles;   // ... taken when an index range has at least this many tiles (default 128; test knob)
    int dual_sample_short; // 1 (default): the sample sweep of kz_knn_dual uses lists of 16 (32) over several index ranges whatever K' is; 0: lists of K'
    int dual_overlap; // 1 (default): kz_knn_dual runs the reverse direction's chain on the second stream beside the forward finalize; 0: behind it
    double dual_max_gb; // kz_knn_dual: transient footprint budget in GiB (0 = the built-in 32)
    int dual_force;   // test knob: run the dual pass also where it does not pay (few query rows)
    int h_q64;        // 64-queries-per-wave kernel (kz_knn_h64.h) for K' = 16 sweeps of 4 .. 13 slices: 2 (default) = where it pays (kz_knn_impl), 1 = always, 0 = never
    int fin_fast_div; // finalize kernel, cosine: shared-reciprocal division (kz_div_shared)
    double probe_min_pairs;  // ordinary searches below this many distance pairs take neither the tier probe nor a floor (5e10)
    int list_floor;   // kz_knn_dual: 1 = the forward lists of the shared sweep start at a population floor (kz_knn_dual.h "POPULATION FLOOR")
    int tier_probe;   // rows of the strided sample a large ordinary search sends through the fp16 pass first (0 = off; default 1024): more than half uncertified -> the call starts at split-bf16
    int dual_nested;  // kz_knn_dual: 1 (default) = the sample is the first tiles of the dealt image and is swept ONLY by the sample sweep (kz_knn_dual.h "NESTED")
    int esc_ladder;   // 1 (default): kz_knn.hip "LADDER AFTER THE FACT"
    int exact_rows;   // 1 (default): the exact distance kernel that keeps four query rows in registers and takes 64 / LPR index rows per step
    int abl;          // diagnostics (bit mask): 1 = kz_knn.hip "abl_refloor", 2 = "abl_stamp", 4 / 8 = kz_range.h (overflow / hand-back paths)
    double floor_margin;  // seeded lists: the largest shortfall of the probe below the model, times this (default 1.3; 0 = the model itself: a test knob)
    int dual_rank;    // kz_knn_dual: rank of the sample key that becomes a row's event threshold (0 = automatic, -1 = k + 1, > 0 = that rank; kz_knn_dual.h)
    int spec_rows;    // exact kernels launched speculatively behind every finalize kernel for up to this many uncertified rows (default 64; kz_knn.hip "SPECULATIVE RESCUE")
    int wide_lists;   // fp16 tier's WIDE route (kz_knn_impl): lists of 16 per query when the tier probe finds the keys dense around the k-th neighbour (default 32; 0 = off)
    int wide_sel;     // ... entries of those lists the finalize kernel selects (default 256)
    // scratch (grown on demand, reused across calls)
    void* scratch;
    size_t scratch_bytes;
    float* floor_buf;      // seeded lists of an ordinary search: one float per padded query row (kz_knn.hip "POPULATION FLOOR")
    size_t floor_bytes;
    int* d_counters;  // small device int array (fail counter, flags)
    int* h_counters;  // pinned host mirror
    int n_cus;        // compute units of the device
    void* h_stage;    // pinned host staging for the per-call work table
    size_t h_stage_bytes;
    int h_stage_flip; // which half of h_stage the last pass used (kz_prepare_pass)
    // stream-ordered free list: buffers released by kz_free / kz_matrix_destroy are reused by later allocations of a
    // similar size instead of going through hipFree (device-wide sync) + hipMalloc on every fit()
    struct { void* ptr; size_t bytes; } pool[KZ_POOL_SLOTS];
    int pool_n;
    size_t pool_bytes;
    void* live_ptr[KZ_LIVE_MAX];   // capacity of every buffer handed out (so kz_free needs no size)
    size_t live_bytes[KZ_LIVE_MAX];
};

int kz_pool_alloc(kz_ctx* ctx, size_t bytes, void** out);   // returns KZ_OK / KZ_ERR_NOMEM
void kz_pool_free(kz_ctx* ctx, void* ptr, size_t bytes);

// Shift and scale shared by the fp16 images of the matrices that are searched against each other (kz_pack.hip):
// distances are translation invariant, so both sides are centred with ONE vector mu (the column mean of the first index
// of the pair) and scaled by ONE power of two S into the fp16 range.  Reference counted: images keep it alive.
struct kz_center {
    float* d_mu;      // [d_pad16] float32 shift (zero padded)
    double* d_scale;  // device {S, 1 / S^2}
    int refs;
};

// fp16 operand image of a matrix (first pass of the fused kernel, kz_knn_h16.h), built lazily by kz_knn
// Rows dealt over P parts: position r' of a sequence of N rows <- row j, with j running through the parts one after the other,
// part p = rows p, p + P, p + 2 P, ... (the first N % P parts hold one row more).  P = 1: j = r'.
__host__ __device__ __forceinline__ int64_t kz_dealt_row(int64_t rp, int64_t N, int P) {
    const int64_t q = N / P, r = N - q * P;
    int64_t part, i;
    if (rp < r * (q + 1)) {
        part = rp / (q + 1);
        i = rp - part * (q + 1);
    } else {
        const int64_t j2 = rp - r * (q + 1);
        part = r + j2 / q;
        i = j2 - (part - r) * q;
    }
    return part + P * i;
}

struct kz_himage {
    kz_center* center;
    unsigned short* packed;  // [n_tiles][nsr][2 planes][128 rows][8] fp16: plane p = k 8p..8p+7 of the 16-k slice
    float* bias;             // [n_tiles*128] accumulator init -S^2 |x_c|^2 / 2 (pad rows: -inf)
    double* rowq;            // [n][3] unscaled: |x_c|^2, |x_h|, |x_c - x_h|  (x_c = float32(x - mu), x_h = fp16 operand / S)
    double* d_max;           // device [3]: max |x_h|, max |x_c - x_h|, max |x_c|^2 over the rows
    // a second image with the rows DEALT over dealt_P index ranges (kz_himage_dealt; short-list route of the ordinary kernel).
    // dealt_* = the image kz_himage_dealt selected last; TWO are kept (slot): a search and the re-search of its uncertified rows
    // deal the index over different numbers of ranges, and with one buffer every step re-packed the whole index twice (500 k x
    // 200: 0.68 ms each)
    int dealt_P;
    unsigned short* dealt_packed;
    float* dealt_bias;
    int* dealt_perm;         // [n_tiles*128] matrix row of image row r (-1 behind the end)
    struct { int P; unsigned short* packed; float* bias; int* perm; } slot[2];
    int slot_cur;
};

struct kz_matrix {
    kz_ctx* ctx;
    int64_t n, d;
    int dtype, metric;
    double mink_p;    // KZ_MINKOWSKI: the exponent (kz_matrix_set_minkowski_p; 2 by default)
    int64_t n_tiles;  // ceil(n / 128)
    int kg;           // d_pad / 4 (number of 4-wide k-groups), d_pad = round_up(d, 16)
    int kg_bf;        // same for the split-bf16 image (currently equal to kg)
    void* raw;        // [n, d] dtype, row-major (exact data, used by the float64 re-rank)
    bool raw_borrowed;  // raw is the caller's buffer (kz_matrix_create rows_on_device = 2 / 3), not ours to free
    bool raw_only;      // rows_on_device = 3: rows only -- no norms, no operand images: a row SOURCE (kz_dsl_fit), not searchable
    float* packed;    // [n_tiles][kg][128][4] float32 MFMA operand image (NULL until kz_matrix_image_f32)
    unsigned short* packed_bf;  // [n_tiles][kg/4][4 planes][128][8] bf16 split image: planes hi(k 0-7), hi(k 8-15), lo, lo
                                // (NULL until kz_matrix_image_bf)
    float* bias;      // [n_tiles*128] accumulator init: -|y|^2/2 (euclidean family), 0 (cosine), -inf (pad rows)
    double* sqn;      // [n] float64: squared norms (euclidean family) or norms with 0 -> 1 (cosine)
    double max_norm;  // max_j |y_j|  (host copy, valid once `checked`)
    bool checked;     // the norm kernel's verdict (finite input, max_norm) has been read back (kz_matrix_check)
    double* d_stats;  // device [4]: max |y_j| (as max_norm), max |operand element| (normalised rows for cosine)
    kz_himage* himg;  // fp16 image (NULL until a kz_knn call builds it)
    double* norm64;   // cosine, float32 rows: [n][d] float64 rows x / |x| -- the values the re-rank's per-element divisions produce
                      // (NULL until kz_matrix_norm64: built for finalize launches with many candidates per query, kz_knn_fin_wide.h)
    size_t raw_bytes, packed_bytes, bias_bytes, sqn_bytes;
};

// operand images are built on first use (kz_pack.hip); all enqueue on the context's stream
int kz_matrix_image_f32(kz_matrix* m);
int kz_matrix_image_bf(kz_matrix* m);
int kz_matrix_norm64(kz_matrix* m);   // KZ_OK also when the image is not built (wrong metric / dtype / shape, or too large): m->norm64 stays NULL
int kz_himage_ensure(kz_matrix* query, kz_matrix* index);
int kz_himage_pack_permuted(kz_matrix* m, const int* d_perm, unsigned short* packed, float* bias);
int kz_himage_pack_rows(kz_matrix* m, const int* d_rows, int64_t n_rows, int64_t n_pad, unsigned short* packed, float* bias);
int kz_himage_dealt(kz_matrix* m, int P);
// stable sort of (float key, int value) pairs on the context's stream (kz_sort.hip)
int kz_sort_pairs_f32_i32(kz_ctx* ctx, const float* keys_in, float* keys_out, const int* vals_in, int* vals_out, int n, int descending);
int kz_matrix_check(kz_matrix* m);
void kz_himage_free(kz_matrix* m);

void kz_set_error(const char* fmt, ...);

#define KZ_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t _e = (call);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            kz_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return KZ_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

#define KZ_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            kz_set_error(__VA_ARGS__); \
            return KZ_ERR_INVALID;     \
        }                              \
    } while (0)

int kz_scratch(kz_ctx* ctx, size_t bytes, void** out);
int kz_floor_buf(kz_ctx* ctx, size_t bytes, float** out);

// ---------------------------------------------------------------------------------------------------
// Canonical wave-cooperative float64 dot product.  Every exact distance in the library (row norms, the
// re-rank, the exact fallback) goes through this routine so that |x|^2 + |y|^2 - 2 x.y is EXACTLY 0 for
// x == y and the same pair gives bit-identical values in the forward and the reverse pass.
// All 64 lanes must be active; every lane returns the same value.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double kz_wave_sum(double acc) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}

// Order of the canonical dot product: the row is cut into chunks of 256 elements; inside a chunk lane l owns elements
// 4l .. 4l+3 (one 16-byte load for float32 rows), accumulated in increasing k by an fma chain per lane; then the butterfly
// sum.  kz_row4 fetches a lane's four elements of a chunk (vector load when the row allows it, scalar loads with zero fill
// at the tail: the fma of a zero leaves the chain unchanged, so both paths give identical bits).
template <typename T>
__device__ __forceinline__ void kz_row4(const T* __restrict__ row, int k0, int d, bool vec, double (&out)[4]) {
    if (vec && k0 + 3 < d) {
        if (sizeof(T) == 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + k0);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            const double2 v0 = *reinterpret_cast<const double2*>(row + k0), v1 = *reinterpret_cast<const double2*>(row + k0 + 2);
            out[0] = v0.x; out[1] = v0.y; out[2] = v1.x; out[3] = v1.y;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u] = k0 + u < d ? (double)row[k0 + u] : 0.0;
    }
}
// a row may be read with 16-byte loads when its start and its pitch are 16-byte aligned
template <typename T>
__device__ __forceinline__ bool kz_row_vec_ok(const T* base, int d) {
    return ((reinterpret_cast<uintptr_t>(base) | ((uintptr_t)d * sizeof(T))) & 15u) == 0;
}

template <typename T>
__device__ __forceinline__ double kz_wave_dot(const T* __restrict__ a, const T* __restrict__ b, int d, int lane) {
    const bool vec = kz_row_vec_ok(a, d) && kz_row_vec_ok(b, d);
    double acc = 0.0;
    for (int k0 = 4 * lane; k0 < d; k0 += 256) {
        double x[4], y[4];
        kz_row4(a, k0, d, vec, x);
        kz_row4(b, k0, d, vec, y);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fma(x[u], y[u], acc);
    }
    return kz_wave_sum(acc);
}

// a / b for MANY numerators over one divisor, bit for bit the IEEE quotient: rcp = 1.0 / b (correctly rounded, computed once),
// q0 = RN(a rcp) is within 2 ulp of a / b; one residual step makes it faithful (q1 = RN(q0 + (a - b q0) rcp): the value before
// the rounding is within 2^-51 ulp of a / b); for a faithful q1 the residual a - b q1 is exact and RN(q1 + (a - b q1) rcp) is
// the correctly rounded quotient (Markstein, "Computation of elementary functions on the IBM RISC System/6000 processor", 1990,
// theorem on the final step of a division with a correctly rounded reciprocal).  Five multiply-adds against the eleven
// instructions (one of them a quarter-rate reciprocal) of the compiler's division.  Valid while nothing under- or overflows: here
// a is a float32 value or 0 and b the row's norm >= |a| (or 1.0 for a zero row), so |a / b| <= 1 and the residuals stay far
// above the subnormal range; the caller takes the plain division when rcp is not finite (b = 0).  tests/test_gpu_fast_div.py compares the two on
// 2^30 pairs.
__device__ __forceinline__ double kz_div_shared(double a, double b, double rcp) {
    const double q0 = a * rcp;
    const double q1 = fma(fma(-q0, b, a), rcp, q0);
    const double q2 = fma(fma(-q1, b, a), rcp, q1);
    return q2;
}

// cosine: sum_k (a_k / na) * (b_k / nb) with the per-element divisions scikit-learn's normalize() performs
template <typename T>
__device__ __forceinline__ double kz_wave_dot_normalized(const T* __restrict__ a, double na, const T* __restrict__ b,
                                                         double nb, int d, int lane) {
    const bool vec = kz_row_vec_ok(a, d) && kz_row_vec_ok(b, d);
    double acc = 0.0;
    for (int k0 = 4 * lane; k0 < d; k0 += 256) {
        double x[4], y[4];
        kz_row4(a, k0, d, vec, x);
        kz_row4(b, k0, d, vec, y);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = fma(x[u] / na, y[u] / nb, acc);
    }
    return kz_wave_sum(acc);
}

// The exact float64 value the search ranks an index row by (squared euclidean distance / cosine distance): the re-rank of
// kz_knn_finalize_kernel, the exact fallback and kz_pair_values all evaluate THIS expression on the canonical dot product.
// Minkowski family beyond p = 2 (KZ_MANHATTAN, KZ_CHEBYSHEV, KZ_MINKOWSKI): scikit-learn's generic DistanceMetric
// (sklearn/metrics/_dist_metrics.pyx.tp: ManhattanDistance / ChebyshevDistance / MinkowskiDistance): the difference x_j - y_j in
// the INPUT dtype, |.| (to the power p) accumulated in float64 IN FEATURE ORDER, the result -- the ranking value: sum |.|, max |.|,
// sum |.|^p -- rounded to the input dtype.  Every kernel that evaluates the family (the tiled distance kernel of kz_knn.hip,
// kz_pair_values) adds the terms of a pair in this order, one thread per pair: manhattan and chebyshev values are scikit-learn's
// bit for bit, minkowski's to the last bit of pow().
// One term.  p_int = 3 or 4 (float32 inputs only, kz_family_p_int): the power as a product with ONE rounding -- a is a float32
// value, so a a is exact in float64 (48 bits) and (a a) a, (a a)(a a) round the exact product once: the correctly rounded a^p, what
// a correctly rounded pow() returns.  Anything else -- float64 inputs, p >= 5, fractional p -- calls pow() as scikit-learn's
// MinkowskiDistance does (a longer chain, or a chain on float64 values, rounds more than once: 1 ulp off in a quarter of the terms).
// CHAIN: -1 = decide at run time from p_int; 3 / 4 = that product, unrolled
template <typename T, int METRIC, int CHAIN = -1>
__device__ __forceinline__ double kz_family_term(T x, T y, double p, int p_int) {
    const T df = x - y;
    const double a = fabs((double)df);
    if (METRIC != KZ_MINKOWSKI) return a;
    if (CHAIN == 3 || (CHAIN < 0 && p_int == 3)) return (a * a) * a;
    if (CHAIN == 4 || (CHAIN < 0 && p_int == 4)) return (a * a) * (a * a);
    return pow(a, p);
}
template <int METRIC>
__device__ __forceinline__ double kz_family_add(double acc, double term) {
    return METRIC == KZ_CHEBYSHEV ? fmax(acc, term) : acc + term;
}
__host__ __device__ __forceinline__ int kz_family_p_int(int metric, double p, bool f32_inputs) {
    return (metric == KZ_MINKOWSKI && f32_inputs && (p == 3.0 || p == 4.0)) ? (int)p : 0;
}
template <typename T, int METRIC>
__device__ __forceinline__ double kz_family_value_seq_m(const T* __restrict__ a, const T* __restrict__ b, int d, double p, int p_int) {
    double acc = 0.0;
    for (int j = 0; j < d; ++j) acc = kz_family_add<METRIC>(acc, kz_family_term<T, METRIC>(a[j], b[j], p, p_int));
    return sizeof(T) == 4 ? (double)(float)acc : acc;
}
// (one thread, one pair)
template <typename T>
__device__ __forceinline__ double kz_family_value_seq(const T* __restrict__ a, const T* __restrict__ b, int d, int metric, double p) {
    const int p_int = kz_family_p_int(metric, p, sizeof(T) == 4);
    if (metric == KZ_MANHATTAN) return kz_family_value_seq_m<T, KZ_MANHATTAN>(a, b, d, p, p_int);
    if (metric == KZ_CHEBYSHEV) return kz_family_value_seq_m<T, KZ_CHEBYSHEV>(a, b, d, p, p_int);
    return kz_family_value_seq_m<T, KZ_MINKOWSKI>(a, b, d, p, p_int);
}

// The exact float64 value the search ranks an index row by (squared euclidean distance / cosine distance / the Minkowski family's
// reduced distance): the re-rank of kz_knn_finalize_kernel, the exact fallback and kz_pair_values all evaluate THIS expression.
template <typename T>
__device__ __forceinline__ double kz_exact_value(const T* q, const T* y, double qs, double ys, int d, int metric, int lane, double p = 2.0) {
    if (metric >= KZ_MANHATTAN) return kz_family_value_seq<T>(q, y, d, metric, p);   // (every lane the whole pair: no fused kernel ranks by this metric)
    if (metric == KZ_COSINE) {
        const double sim = kz_wave_dot_normalized(q, qs, y, ys, d, lane);
        double v = 1.0 - sim;  // sklearn cosine_distances: S *= -1; S += 1; clip(0, 2)
        v = fmin(fmax(v, 0.0), 2.0);
        return v;
    }
    const double dot = kz_wave_dot(q, y, d, lane);
    const double d2 = (qs + ys) - 2.0 * dot;  // |x|^2 - 2 x.y + |y|^2 (_argkmin.pyx.tp:494-499)
    return fmax(d2, 0.0);                     // _argkmin.pyx.tp:502
}

