// Exact float64 distances, ONE PAIR PER LANE (round 6) -- for MANY uncertified rows (data whose tightest clusters are orders of
// magnitude tighter than its extent: a tenth of the rows end on the exact kernels, tools/cliff_probe.py last data kind).
//
// kz_exact_dist_rows_kernel spends a lane GROUP on a pair: the lanes each own four elements, and every pair costs a butterfly sum
// -- six rounds of float64 shuffles and adds, ~25 vector instructions per pair on top of its four fma: 2e10 pairs/s, 13 - 30 us per
// query row against 200 k - 300 k index rows.  Here a LANE owns an index row and evaluates the SAME expression tree by itself:
//     the canonical dot product (kz_wave_dot) = per "leaf" l = 0 .. 63 an fma chain over elements 4 l .. 4 l + 3 (then 256 + 4 l ..
//     for rows beyond 256 elements), the 64 leaves summed by the butterfly -- level 32 pairs leaf l with l + 32, level 16 the sums l with
//     l + 16, ... -- whose value does not depend on which lane holds what.  A depth-first walk of that tree (leaves in bit-reversed
//     order) needs seven partial sums at a time and no shuffle: 4 fma per leaf + 63 adds per pair, bit for bit the value of the
//     wave-cooperative kernels (IEEE addition is commutative; leaves past the end of the row are exact zeros there and skipped here:
//     x + 0 = x for every partial sum that can occur -- none is -0).
// A workgroup (four waves) stages 64 index rows in LDS, transposed ([leaf][row], 16 bytes per entry and one entry of padding per
// leaf: conflict-free for the staging stores and for the lanes' reads); a wave takes four query rows at a time from the batch -- their
// float64 (cosine: normalised) values come through the scalar cache, a pre-pass wrote them (kz_exact_qprep_kernel) -- and walks the
// tree for 4 x 64 pairs.  ~66 vector instructions per pair-quad and lane instead of ~100 per pair: see DESIGN section 3.1.
#pragma once

// float64 operand rows of the batch's query rows: qd[b][e] = (double)q[e] (cosine: / |q|, the IEEE division, as kz_wave_dot_normalized);
// d_pad = leaves x 4 elements per row, zero filled.
__global__ __launch_bounds__(256) void kz_exact_qprep_kernel(const int* __restrict__ fail_list, int batch0, int nb, int64_t q_begin,
                                                             const float* __restrict__ qraw, const double* __restrict__ qsqn, int d, int d_pad,
                                                             int metric, double* __restrict__ qd, double* __restrict__ qsq,
                                                             const int* __restrict__ dyn_n = nullptr) {
    const int b = blockIdx.x;   // (the grid covers the batch rounded up to whole blocks of rows: the rows past it are zeros)
    if (dyn_n) {   // (speculative launch, kz_spec_rescue: nb is the capacity, the row count is on the device)
        const int n = *dyn_n;
        if (n > nb || n <= 0) return;
        nb = n;
    }
    const bool live = b < nb;
    const int64_t qrow = live ? q_begin + fail_list[batch0 + b] : 0;
    const double qs = live ? qsqn[qrow] : 1.0;
    if (threadIdx.x == 0) qsq[b] = live ? qs : 0.0;
    for (int e = threadIdx.x; e < d_pad; e += 256) {
        double v = 0.0;
        if (live && e < d) {
            v = (double)qraw[qrow * (int64_t)d + e];
            if (metric == KZ_COSINE) v = v / qs;
        }
        qd[(int64_t)b * d_pad + e] = v;
    }
}

// The same for the rows of MANY blocks at once (kz_range.h, grouped ranges): rows [n_slots] holds the query row of every operand row
// (relative to q_begin), -1 = a padding row of a block (zeros).
__global__ __launch_bounds__(256) void kz_exact_qprep_slots_kernel(const int* __restrict__ rows, int64_t q_begin, const float* __restrict__ qraw,
                                                                   const double* __restrict__ qsqn, int d, int d_pad, int metric,
                                                                   double* __restrict__ qd, double* __restrict__ qsq) {
    const int b = blockIdx.x;
    const int r = rows[b];
    const bool live = r >= 0;
    const int64_t qrow = live ? q_begin + r : 0;
    const double qs = live ? qsqn[qrow] : 1.0;
    if (threadIdx.x == 0) qsq[b] = live ? qs : 0.0;
    for (int e = threadIdx.x; e < d_pad; e += 256) {
        double v = 0.0;
        if (live && e < d) {
            v = (double)qraw[qrow * (int64_t)d + e];
            if (metric == KZ_COSINE) v = v / qs;
        }
        qd[(int64_t)b * d_pad + e] = v;
    }
}

constexpr int KZ_XL_ROWS = 64;   // index rows per workgroup tile = lanes of a wave
constexpr int KZ_XL_Q = 4;       // query rows a wave carries through the tree at once (a workgroup: 4 waves x 4 rows per block)

// ELT = float (raw rows; euclidean family) or double (the normalised float64 rows of a cosine index, kz_matrix_norm64)
// (pointers into the tile and the query block are LDS-address-space pointers throughout: as generic pointers every leaf's five
//  addresses were 64-bit values computed ahead of the walk -- 512 registers and spills -- instead of ds_read immediates)
typedef __attribute__((address_space(3))) const char kz_xl_lds;
typedef float kz_xl_f4 __attribute__((ext_vector_type(4)));     // (native vectors: HIP's float4 / double2 classes cannot be read through
typedef double kz_xl_d2 __attribute__((ext_vector_type(2)));    //  an address-space-qualified reference)
typedef __attribute__((address_space(3))) const kz_xl_f4 kz_xl_lds_f4;
typedef __attribute__((address_space(3))) const kz_xl_d2 kz_xl_lds_d2;
template <typename ELT>
struct KzXlEntry;
template <>
struct KzXlEntry<float> {
    kz_xl_f4 v;
    __device__ __forceinline__ void load(kz_xl_lds* p) { v = *reinterpret_cast<kz_xl_lds_f4*>(p); }
    __device__ __forceinline__ void get(double (&y)[4]) const { y[0] = v.x, y[1] = v.y, y[2] = v.z, y[3] = v.w; }
};
template <>
struct KzXlEntry<double> {
    kz_xl_d2 a, b;
    __device__ __forceinline__ void load(kz_xl_lds* p) {
        a = *reinterpret_cast<kz_xl_lds_d2*>(p);
        b = *reinterpret_cast<kz_xl_lds_d2*>(p + 16);
    }
    __device__ __forceinline__ void get(double (&y)[4]) const { y[0] = a.x, y[1] = a.y, y[2] = b.x, y[3] = b.y; }
};

// operands of one leaf: the lane's index entries (NV chunks) and the four elements of each of the wave's KZ_XL_Q query rows
template <typename ELT, int NV>
struct KzXlOps {
    KzXlEntry<ELT> y[NV];
    double q[NV][KZ_XL_Q][4];   // (wave-uniform values: scalar loads, scalar registers)
};

template <int LOG>
__host__ __device__ constexpr int kz_bitrev(int i) {
    int r = 0;
    for (int b = 0; b < LOG; ++b) r |= ((i >> b) & 1) << (LOG - 1 - b);
    return r;
}

// NLEAF = 16 / 32 / 64 leaves of the first chunk (d <= 64 / 128 / 256; beyond: 64 and NV = 2).  vals[b][i] as kz_exact_dist_kernel.
// LDS: the index tile [leaf entries][KZ_XL_ROWS + 1] x ENTRY bytes, then the query block [4 waves x KZ_XL_Q rows][d_pad] float64.
// COS_RAW: cosine on the RAW float32 rows (no normalised float64 image of the index): the lane divides its row's elements by the
// row's norm as the cooperative kernel does (one reciprocal per row, kz_div_shared: the IEEE quotient) -- per leaf, shared by the
// wave's four query rows.
// GATHER (kz_range.h, grouped ranges): ONE launch for many dense blocks.  Block g = blockIdx.z has its own query rows (nb of them,
// operand rows from qd_off on) and its own "index": the list gather [gather_off .. + n_rows) of index rows -- row i of the block is
// row gather[i] of yrows / ysqn; its values go to vals [val_off + b n_rows + i].  blockIdx.y takes `q_chunk` query rows of the block
// (a multiple of the workgroup's 16): a block of a few hundred query rows x a few thousand index rows fills the chip.
struct KzXlGroup {
    int nb, n_rows;
    long long qd_off, gather_off, val_off;   // operand rows (in rows), list entries, values
};
template <int NLEAF, int NV, typename ELT, bool COS_RAW = false, bool GATHER = false>
__global__ __launch_bounds__(256) void kz_exact_dist_lanes_kernel(int nb, const double* __restrict__ qd, const double* __restrict__ qsq,
                                                                  const ELT* __restrict__ yrows, const double* __restrict__ ysqn, int64_t n_i, int d,
                                                                  int d_pad, int metric, double* __restrict__ vals,
                                                                  const int* __restrict__ dyn_n = nullptr, const int* __restrict__ gather = nullptr,
                                                                  const KzXlGroup* __restrict__ groups = nullptr, int q_chunk = 0) {
    extern __shared__ __attribute__((aligned(16))) char xl_sm[];
    if constexpr (GATHER) {
        const KzXlGroup g = groups[blockIdx.z];
        nb = g.nb;
        n_i = g.n_rows;
        qd += (size_t)g.qd_off * d_pad;
        qsq += g.qd_off;
        gather += g.gather_off;
        vals += g.val_off;
        if ((int64_t)blockIdx.x * KZ_XL_ROWS >= n_i || (int)blockIdx.y * q_chunk >= nb) return;   // (uniform: before anything is staged)
    }
    if (dyn_n) {   // (speculative launch: nb is the capacity; nothing to do -- or too much -- returns before the tile is staged)
        const int n = *dyn_n;
        if (n > nb || n <= 0) return;
        nb = n;
    }
    constexpr int ENTRY = 4 * (int)sizeof(ELT);
    constexpr int LOG = NLEAF == 64 ? 6 : (NLEAF == 32 ? 5 : 4);
    constexpr int QB = 4 * KZ_XL_Q;   // query rows of the workgroup's block
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t i0 = (int64_t)blockIdx.x * KZ_XL_ROWS;
    const int n_groups = d_pad / 4;   // leaf entries per row (both chunks); d_pad = d rounded up to whole leaves
    char* ytile = xl_sm;
    // ---- stage the tile: entry (row r, group g) -> LDS [g][r]; consecutive threads take consecutive groups of one row (coalesced
    //      16- / 32-byte global loads), the padding entry per leaf spreads their stores over the banks ----
    for (int idx = threadIdx.x; idx < KZ_XL_ROWS * n_groups; idx += 256) {
        const int r = idx / n_groups, g = idx - r * n_groups;
        int64_t yi = i0 + r < n_i ? i0 + r : n_i - 1;
        if constexpr (GATHER) yi = gather[yi];
        char* dst = ytile + ((size_t)g * (KZ_XL_ROWS + 1) + r) * ENTRY;
        const ELT* src = yrows + yi * (int64_t)d + 4 * g;
        if constexpr (sizeof(ELT) == 4) {
            *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        } else {
            *reinterpret_cast<double2*>(dst) = *reinterpret_cast<const double2*>(src);
            *reinterpret_cast<double2*>(dst + 16) = *reinterpret_cast<const double2*>(src + 2);
        }
    }
    __syncthreads();   // (the tile is staged)
    const int64_t i = i0 + lane;
    const double ys = i < n_i ? ysqn[GATHER ? (int64_t)gather[i] : i] : 1.0;   // (euclidean family: |y|^2; cosine: the row's norm, used by COS_RAW only)
    const double ys_rcp = 1.0 / ys;
    const bool ys_fin = (((unsigned long long)__double_as_longlong(ys_rcp) >> 52) & 0x7ff) != 0x7ff;
    kz_xl_lds* ylane = (kz_xl_lds*)(ytile + (size_t)lane * ENTRY);
    // A leaf's operands: the lane's index entry from LDS; the four elements of each of the wave's query rows from the float64 operand
    // rows in global memory -- the address is the same in every lane, so they come through the SCALAR cache into scalar registers and
    // enter the fma as its scalar source: no LDS traffic (a first version broadcast them from LDS: eight of the nine ds_read_b128 per
    // leaf, and the kernel was bound by the LDS at 1.8e10 pairs/s).  A leaf past the end of the row (wave-uniform) loads nothing.
    const double* qrow = nullptr;   // (this wave's KZ_XL_Q rows of the block; set per block)
    auto load_ops = [&](const int l, KzXlOps<ELT, NV>& o) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int g = l + 64 * c;
            if (g < n_groups) {   // (uniform)
                o.y[c].load(ylane + g * ((KZ_XL_ROWS + 1) * ENTRY));
#pragma unroll
                for (int j = 0; j < KZ_XL_Q; ++j) {
                    const double* q = qrow + (size_t)j * d_pad + 4 * g;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.q[c][j][e] = q[e];
                }
            }
        }
    };
    const int nb_pad = (nb + QB - 1) / QB * QB;   // (qd / qsq hold whole blocks of rows, zero filled)
    const int bb_first = GATHER ? (int)blockIdx.y * q_chunk : 0;
    const int bb_last = GATHER ? (bb_first + q_chunk < nb_pad ? bb_first + q_chunk : nb_pad) : nb_pad;
    for (int bb = bb_first; bb < bb_last; bb += QB) {   // (uniform)
        // (a wave whose four rows of this block all lie past the batch has nothing to compute: a speculative launch for TWO rows of a
        //  1 M-row index ran the walk in all four waves -- four times the scalar loads the live wave waits for)
        if (bb + wave * KZ_XL_Q >= nb) continue;
        qrow = qd + ((size_t)bb + (size_t)wave * KZ_XL_Q) * d_pad;
        // ---- the tree, depth first: leaves in bit-reversed order; the partial sums of the completed subtrees sit in lev[0 .. LOG),
        //      one per level, like the digits of a binary counter: leaf `it` is added in at level 0 and carried upward through every
        //      level that holds a sum (the set low bits of `it`) -- earlier subtree + later subtree, the butterfly's pairings.  A ROLLED
        //      loop, two leaves per trip (the operand buffers swap roles): unrolled, the compiler hoists every LDS read of the walk
        //      to its top (512 registers, thousands spilled). ----
        // (seven separately named level registers and one explicit case per carry count: as an array of levels behind a lambda, part of
        //  it was left in scratch memory)
        double lev0[KZ_XL_Q], lev1[KZ_XL_Q], lev2[KZ_XL_Q], lev3[KZ_XL_Q], lev4[KZ_XL_Q], lev5[KZ_XL_Q], lev6[KZ_XL_Q];
        KzXlOps<ELT, NV> opa, opb;
        auto leaf_of = [](int it) { return (int)(__builtin_bitreverse32((unsigned)it) >> (32 - LOG)); };
#define KZ_XL_ADD(L)                                               \
    _Pragma("unroll") for (int j = 0; j < KZ_XL_Q; ++j) x[j] = L[j] + x[j]
#define KZ_XL_PUT(L)                                               \
    _Pragma("unroll") for (int j = 0; j < KZ_XL_Q; ++j) L[j] = x[j]
#define KZ_XL_LEAF(IT, O)                                                                                          \
    do {                                                                                                           \
        const int it_ = (IT);                                                                                      \
        double x[KZ_XL_Q];                                                                                         \
        _Pragma("unroll") for (int j = 0; j < KZ_XL_Q; ++j) x[j] = 0.0;                                            \
        _Pragma("unroll") for (int c = 0; c < NV; ++c) {                                                           \
            if (leaf_of(it_) + 64 * c < n_groups) { /* (uniform; a leaf past the row is an exact zero) */          \
                double y[4];                                                                                       \
                (O).y[c].get(y);                                                                                   \
                if (COS_RAW) {                                                                                     \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) y[e] = ys_fin ? kz_div_shared(y[e], ys, ys_rcp) : y[e] / ys; \
                }                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < KZ_XL_Q; ++j) {                                              \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) x[j] = fma((O).q[c][j][e], y[e], x[j]);          \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        /* carry upward through the levels that hold a sum -- as many as `it` has trailing one bits -- and rest in the next */ \
        const int nm_ = __builtin_ctz(~(unsigned)it_);                                                             \
        if (nm_ == 0) {                                                                                            \
            KZ_XL_PUT(lev0);                                                                                       \
        } else if (nm_ == 1) {                                                                                     \
            KZ_XL_ADD(lev0); KZ_XL_PUT(lev1);                                                                      \
        } else if (nm_ == 2) {                                                                                     \
            KZ_XL_ADD(lev0); KZ_XL_ADD(lev1); KZ_XL_PUT(lev2);                                                     \
        } else if (nm_ == 3) {                                                                                     \
            KZ_XL_ADD(lev0); KZ_XL_ADD(lev1); KZ_XL_ADD(lev2); KZ_XL_PUT(lev3);                                    \
        } else if (nm_ == 4) {                                                                                     \
            KZ_XL_ADD(lev0); KZ_XL_ADD(lev1); KZ_XL_ADD(lev2); KZ_XL_ADD(lev3); KZ_XL_PUT(lev4);                   \
        } else if (nm_ == 5) {                                                                                     \
            KZ_XL_ADD(lev0); KZ_XL_ADD(lev1); KZ_XL_ADD(lev2); KZ_XL_ADD(lev3); KZ_XL_ADD(lev4); KZ_XL_PUT(lev5);  \
        } else {                                                                                                   \
            KZ_XL_ADD(lev0); KZ_XL_ADD(lev1); KZ_XL_ADD(lev2); KZ_XL_ADD(lev3); KZ_XL_ADD(lev4); KZ_XL_ADD(lev5);  \
            KZ_XL_PUT(lev6);                                                                                       \
        }                                                                                                          \
    } while (0)
        load_ops(0, opa);
#pragma unroll 1
        for (int it = 0; it < NLEAF; it += 2) {
            load_ops(leaf_of(it + 1), opb);
            KZ_XL_LEAF(it, opa);
            load_ops(leaf_of(it + 2 < NLEAF ? it + 2 : 0), opa);   // (past the last leaf: any entry, never used)
            KZ_XL_LEAF(it + 1, opb);
        }
#undef KZ_XL_LEAF
#undef KZ_XL_ADD
#undef KZ_XL_PUT
        // (the last leaf, it = NLEAF - 1, has LOG trailing ones: the root rests in level LOG)
        double (&root)[KZ_XL_Q] = LOG == 6 ? lev6 : (LOG == 5 ? lev5 : lev4);
        if (i < n_i) {
#pragma unroll
            for (int j = 0; j < KZ_XL_Q; ++j) {
                const int b = bb + wave * KZ_XL_Q + j;
                if (b >= nb) continue;
                double v;
                if (metric == KZ_COSINE)
                    v = fmin(fmax(1.0 - root[j], 0.0), 2.0);
                else
                    v = fmax((qsq[b] + ys) - 2.0 * root[j], 0.0);
                vals[(int64_t)b * n_i + i] = v;
            }
        }
    }
}
