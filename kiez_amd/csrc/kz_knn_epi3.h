// Tile epilogue of the fp16 fused kernel (kz_knn_h16.h), third form: instruction count first.
//
// With one MFMA product per multiply-add a tile of d = 128 is 1024 matrix-pipe cycles per wave, so every VALU / SALU
// instruction of the candidate scan is on the critical path (the second form, kz_tile_epilogue2, executes ~225
// instructions per tile before it has found a single candidate).  What changed:
//   * plain C++ max trees: this translation unit is compiled with -fno-honor-nans (keys are never NaN: operands are
//     finite and clamped, pad rows carry -inf), so fmaxf() is v_max_f32 / v_max3_f32 with the MFMA hazards handled by
//     the compiler -- no canonicalisation, no inline asm, no s_nop padding;
//   * tile-level early out: 16 group maxima -> one tile maximum -> one compare: a tile without an event costs ~45
//     instructions;
//   * BLIND group appends: a lane that has an event in a group of four keys appends all four keys (one ds_write_b128)
//     and {code, next} to the wave's event pool (below); which of the four really beat the threshold is decided at merge
//     time.  No per-key compare / select / counter chain in the scan;
//   * capacity is checked ONCE per tile with scalar arithmetic (pool fill + 16 x lanes with an event): the common path
//     carries no per-group room / resume bookkeeping; tiles that might overflow the pool (the first tiles of a sweep)
//     take a resumable slow path;
//   * for K' <= 32 the candidate lists live in LDS for the whole sweep (16-32 KiB per workgroup) and are written to the
//     output arrays once at the end: a merge insert is a few LDS round trips instead of L2 round trips (~1.2k cycles).
#pragma once

typedef __attribute__((address_space(3))) float kz_lds_f32;
typedef __attribute__((address_space(3))) int kz_lds_i32;

// list storage (template parameter IN_LDS: 0, 1 or 2):
//   1  keys and rows in LDS, [entry][128 queries of the workgroup] (K' <= 32)
//   0  keys and rows in the output arrays, K' contiguous entries per list
//   2  HYBRID: keys in LDS, rows in the output arrays.  A merge insert READS only keys (find the block minimum's position)
//      and WRITES one key and one row: with the keys in LDS the only global traffic of an insert is a fire-and-forget
//      4-byte store, no L2 round trip (K' = 64 at two workgroups per CU: a merge pass cost 26k cycles against 10k with
//      LDS lists).  32 KiB of keys at K' = 64.
template <int IN_LDS>
struct KzListRef;
template <>
struct KzListRef<1> {
    kz_lds_f32* k;
    int i_off;   // rows live i_off floats behind the keys (a constant of the build: one address register, not two)
    static constexpr int KSTRIDE = 128, ISTRIDE = 128;
    __device__ __forceinline__ kz_lds_f32* kp() const { return k; }
    __device__ __forceinline__ kz_lds_i32* ip() const { return (kz_lds_i32*)(k + i_off); }
};
template <>
struct KzListRef<0> {
    float* kb;      // wave-uniform bases of the output arrays ...
    int* ib;
    unsigned off;   // ... and this query's first list entry (element offset)
    static constexpr int KSTRIDE = 1, ISTRIDE = 1;
    __device__ __forceinline__ float* kp() const { return kb + off; }
    __device__ __forceinline__ int* ip() const { return ib + off; }
};
template <>
struct KzListRef<2> {
    kz_lds_f32* k;
    int* ib;
    // this query's first list entry = off_u + (lane & 31) * stride: both parts wave-uniform (scalar registers), the lane
    // part re-made where a row is stored (merges are rare; a per-lane offset held for the whole sweep is a VGPR the
    // three-workgroups-per-CU builds do not have)
    unsigned off_u, stride;
    static constexpr int KSTRIDE = 128, ISTRIDE = 1;
    __device__ __forceinline__ kz_lds_f32* kp() const { return k; }
    __device__ __forceinline__ int* ip() const {
        unsigned l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return ib + off_u + (l & 31u) * stride;
    }
};

// Per-lane candidate state (one lane = one (query, lane-half) pair) ...
template <int IN_LDS>
struct KzCandState3 {
    KzListRef<IN_LDS> list;   // this query's list (lanes < 32 own it)
    float tau;                // K'-th best key of the list as of the last merge (same value in both lane halves)
    int head;                 // newest entry of this lane's chain in the wave's event pool, -1 = empty
};

// ... and the wave's EVENT POOL: one shared log per wave instead of one log per lane.  An entry = the four keys of a group
// in which a lane had an event (16 B) + {code, next} (8 B): code = 16 tile + group, next = the lane's previous entry.
// Lanes append at pool positions handed out by ballot + mbcnt (consecutive 16-byte slots: conflict-free ds_write_b128)
// and link the entry in front of their own chain; a merge lets every lane walk its chain.  Sharing the capacity is the
// point: per-lane logs must be sized for the UNLUCKIEST lane (events are rare and Poisson: with room for 4 events per
// lane a merge was triggered every ~4 tiles by one lane out of 256, and every merge pass runs as long as its fullest
// lane), a pool is sized for the wave's total, its fill level is a scalar, and a merge processes ~100 events at once.
struct KzWavePool {
    __attribute__((address_space(3))) f32x4e* keys;   // [CAP]
    __attribute__((address_space(3))) i32x2e* meta;     // [CAP] {code, next}
    int cnt;                  // entries in use (wave-uniform)
    int tiles_done, next_merge;
};

// Dual pass (kz_knn_dual, kz_dual.hip): besides the per-query lists the kernel reports, for every INDEX row t, the queries
// q whose key for the reverse direction  key'(t, q) = acc(q, t) - bias(t) + bias(q)  reaches a threshold tau(t) fixed
// before the launch (from a sample of the query rows) -- the "events" of t.  The test is  fl(acc - theta(t)) >= -bias(q)
// with theta = tau + bias(t) (rounded down, minus a margin for the one float32 subtraction).  A lane with an event in a
// group of four keys appends the group blindly to the wave's pool (code < 0: not linked into any chain); every merge also
// FLUSHES those entries to a global log (one atomic per flush reserves the range, the entries go out as coalesced 16- and
// 8-byte stores).  Which of the four keys really is an event, and of which index row, is decided by kz_dual_scatter_kernel
// after the launch: the kernel itself carries no per-key test, no per-row atomics and no threshold re-reads.
struct KzDualRef {
    int qrow0;   // global row of this wave's query 0
    // (the log's pointers, counter and capacity are read from the kernel argument segment inside kz_flush_col3, per flush:
    //  carried in scalar registers for the whole sweep they pushed the kernel into spilling SGPRs to a VGPR)
};
constexpr int KZ_COL_FLAG = (int)0x80000000;

__device__ __forceinline__ void kz_flush_col3(const KzWavePool& pool, const KzDualRef& du) {
    const int lane = threadIdx.x & 63;
    // how many of the pool's entries are column entries (wave-uniform), then one reservation for all of them
    int n_col = 0;
    for (int e0 = 0; e0 < pool.cnt; e0 += 64) {   // (uniform trip count)
        const int e = e0 + lane;
        const bool col = e < pool.cnt && pool.meta[e].x < 0;
        n_col += (int)__popcll(__builtin_amdgcn_ballot_w64(col));
    }
    if (n_col == 0) return;
    typedef __attribute__((address_space(4))) const volatile unsigned long long kz_karg_u64;
    const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    f32x4e* log_keys = (f32x4e*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_keys));
    i32x2e* log_meta = (i32x2e*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_meta));
    unsigned long long* log_cnt = (unsigned long long*)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_cnt));
    const long long log_cap = (long long)*(kz_karg_u64*)(ka + offsetof(KnnCandParams, log_cap));
    unsigned long long base = 0;
    if (lane == 0) base = __hip_atomic_fetch_add(log_cnt, (unsigned long long)n_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    for (int e0 = 0; e0 < pool.cnt; e0 += 64) {
        const int e = e0 + lane;
        i32x2e mt;
        mt.x = 0;
        if (e < pool.cnt) mt = pool.meta[e];
        const bool col = mt.x < 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(col);
        if (col) {
            const unsigned long long pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if ((long long)pos < log_cap) {
                // (streaming stores: the log is written once and read by another kernel)
                i32x2e mo;
                mo.x = mt.x & 0x7fffffff;
                mo.y = du.qrow0 + (mt.x & 31);
                __builtin_nontemporal_store(pool.keys[e], log_keys + pos);
                __builtin_nontemporal_store(mo, log_meta + pos);
            }
        }
        base += __popcll(mask);
    }
}

// Two-level minimum of an unsorted K'-entry list: the list is cut into NB blocks whose minima are kept in registers
// (values only: the position of a block's minimum is found again when the block is re-read for the insert).
template <int KP>
struct KzBlockMin3 {
    static constexpr int NB = KP >= 64 ? 8 : 4;
    static constexpr int BS = KP / NB;
    float bm[NB];
    __device__ __forceinline__ void init(const float v) {
#pragma unroll
        for (int i = 0; i < NB; ++i) bm[i] = v;
    }
};

// Insert (v, idx) over the current minimum of the list (caller guarantees v > tau); returns the new minimum in tau.
// Replacing the global minimum touches ONE block: read its K'/NB keys (one round trip), overwrite the first key equal to
// the block minimum, refresh the block minimum -- instead of re-scanning all K' keys.
template <int KP, int IN_LDS>
__device__ __forceinline__ void kz_list_insert3(const KzListRef<IN_LDS>& L, KzBlockMin3<KP>& bs, float v, int idx, float& tau) {
    constexpr int NB = KzBlockMin3<KP>::NB, BS = KzBlockMin3<KP>::BS, S = KzListRef<IN_LDS>::KSTRIDE, SI = KzListRef<IN_LDS>::ISTRIDE;
    float m = bs.bm[0];
    int b = 0;
#pragma unroll
    for (int i = 1; i < NB; ++i) {
        if (bs.bm[i] < m) {
            m = bs.bm[i];
            b = i;
        }
    }
    auto* blk = L.kp() + (b * BS) * S;
    float kk[BS];
#pragma unroll
    for (int jj = 0; jj < BS; ++jj) kk[jj] = blk[jj * S];
    int pos = BS - 1;
#pragma unroll
    for (int jj = BS - 2; jj >= 0; --jj) pos = (kk[jj] == m) ? jj : pos;   // first key equal to the block minimum
    blk[pos * S] = v;
    L.ip()[(b * BS + pos) * SI] = idx;
    float nm = INFINITY;
#pragma unroll
    for (int jj = 0; jj < BS; ++jj) nm = fminf(nm, (jj == pos) ? v : kk[jj]);
    float t = INFINITY;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        bs.bm[i] = (b == i) ? nm : bs.bm[i];
        t = fminf(t, bs.bm[i]);
    }
    tau = t;
}

// The four keys of a logged group against the list: LARGEST FIRST.  A group is logged blindly when its maximum beat the threshold
// of its tile -- at merge time usually ONE of the four keys still beats the list's threshold, and which one differs from lane to
// lane: four tests in row order made the wave execute the insert body (~45 instructions) up to four times per pool entry, once per
// position some lane needed.  Largest first it runs once per entry for nearly every entry; the loop goes on only for the lanes whose
// next-largest key beats the UPDATED threshold (the first tiles of a sweep).  Same set of rows kept (a top-K' set does not depend
// on the order of insertion; equal keys compete for the last place as before: strictly greater wins).
template <int KP, int IN_LDS>
__device__ __forceinline__ void kz_insert_group3(const KzListRef<IN_LDS>& L, KzBlockMin3<KP>& bs,
                                                 __attribute__((address_space(3))) f32x4e* entry,
                                                 const __attribute__((address_space(3))) i32x2e* meta, const int half, float& tau) {
    // (the largest key first -- and struck out of the pool entry; the four are then tested in row order against the UPDATED
    //  threshold: bodies that run only in the first tiles of a sweep, when several keys of a group still beat it.  Keys and entry code
    //  are re-read from LDS behind the first insert instead of being carried across it: at three workgroups per CU the kernel sits
    //  exactly on its 168 registers, and anything more that lives across an insert is spilled)
    auto row_of = [&]() {
        const int code = meta->x;
        return (code >> 4) * KZ_TILE + ((code >> 2) & 3) * 32 + (code & 3) * 8 + 4 * half;
    };
    {
        const f32x4e kv = *entry;
        const float m = fmaxf(fmaxf(kv.x, kv.y), fmaxf(kv.z, kv.w));
        if (!(m > tau)) return;
        const int slot = kv.x == m ? 0 : (kv.y == m ? 1 : (kv.z == m ? 2 : 3));
        reinterpret_cast<__attribute__((address_space(3))) float*>(entry)[slot] = -INFINITY;
        kz_list_insert3<KP, IN_LDS>(L, bs, m, row_of() + slot, tau);
    }
    const f32x4e kv = *entry;
    if (fmaxf(fmaxf(kv.x, kv.y), fmaxf(kv.z, kv.w)) > tau) {
        const int row0 = row_of();
        if (kv.x > tau) kz_list_insert3<KP, IN_LDS>(L, bs, kv.x, row0, tau);
        if (kv.y > tau) kz_list_insert3<KP, IN_LDS>(L, bs, kv.y, row0 + 1, tau);
        if (kv.z > tau) kz_list_insert3<KP, IN_LDS>(L, bs, kv.z, row0 + 2, tau);
        if (kv.w > tau) kz_list_insert3<KP, IN_LDS>(L, bs, kv.w, row0 + 3, tau);
    }
}

// Merge: lane l < 32 walks its own chain, then its partner's (lane l + 32: the other half of the same query), inserting
// every key that still beats the list's threshold.  code = 16 tile + 4 mt + g4; the keys of an entry are index rows
// 128 tile + 32 mt + 8 g4 + 4 half + 0..3 (C layout of the 32x32 MFMA).
// RECOMP: the block minima are NOT carried from merge to merge but re-read from the list at the start of every merge
// (K' reads per merge against K'/4 registers per lane for the whole sweep: what the dual-pass build lacks at three
// workgroups per CU and 13 stationary slices).
template <int KP, int IN_LDS, bool DUAL, bool RECOMP>
__device__ __forceinline__ void kz_merge_pool3(KzCandState3<IN_LDS>& st, KzWavePool& pool, KzBlockMin3<KP>& bs, const KzDualRef& du) {
    const int lane = threadIdx.x & 63;
    if constexpr (DUAL) kz_flush_col3(pool, du);
    const int other = __shfl_xor(st.head, 32, 64);
    if (lane < 32) {
        if constexpr (RECOMP) {
            constexpr int NB = KzBlockMin3<KP>::NB, BS = KzBlockMin3<KP>::BS, S = KzListRef<IN_LDS>::KSTRIDE;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float m = INFINITY;
#pragma unroll
                for (int jj = 0; jj < BS; ++jj) m = fminf(m, st.list.kp()[(b * BS + jj) * S]);
                bs.bm[b] = m;
            }
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            int e = half ? other : st.head;
#pragma unroll 1
            while (e >= 0) {
                if constexpr (RECOMP) {
                    // (the dual build with 13 stationary slices at three workgroups per CU has no register to spare: the four tests
                    //  in row order, as before -- largest-first spilled 2 .. 4 VGPRs there)
                    const f32x4e kv = pool.keys[e];
                    const i32x2e mt = pool.meta[e];
                    e = mt.y;
                    const int code = mt.x;
                    const int row0 = (code >> 4) * KZ_TILE + ((code >> 2) & 3) * 32 + (code & 3) * 8 + 4 * half;
                    if (kv.x > st.tau) kz_list_insert3<KP, IN_LDS>(st.list, bs, kv.x, row0, st.tau);
                    if (kv.y > st.tau) kz_list_insert3<KP, IN_LDS>(st.list, bs, kv.y, row0 + 1, st.tau);
                    if (kv.z > st.tau) kz_list_insert3<KP, IN_LDS>(st.list, bs, kv.z, row0 + 2, st.tau);
                    if (kv.w > st.tau) kz_list_insert3<KP, IN_LDS>(st.list, bs, kv.w, row0 + 3, st.tau);
                } else {
                    auto* entry = pool.keys + e;
                    const auto* meta = pool.meta + e;
                    e = meta->y;
                    kz_insert_group3<KP, IN_LDS>(st.list, bs, entry, meta, half, st.tau);
                }
            }
        }
    }
    st.head = -1;
    pool.cnt = 0;
    st.tau = __shfl(st.tau, lane & 31, 64);
}

#define KZ_EPI3_MERGE() kz_merge_pool3<KP, IN_LDS, DUAL, RECOMP>(st, pool, bs, du)

// One group of four keys: the lanes in `mask` append it to the pool (positions pool.cnt + rank of the lane in the mask)
#define KZ_EPI3_APPEND(gi, ev, mask)                                                                                  \
    do {                                                                                                              \
        if (ev) {                                                                                                     \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                       \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u)); \
            f32x4e kv;                                                                                                \
            kv.x = acc[(gi) >> 2][4 * ((gi) & 3)];                                                                    \
            kv.y = acc[(gi) >> 2][4 * ((gi) & 3) + 1];                                                                \
            kv.z = acc[(gi) >> 2][4 * ((gi) & 3) + 2];                                                                \
            kv.w = acc[(gi) >> 2][4 * ((gi) & 3) + 3];                                                                \
            pool.keys[pos] = kv;                                                                                      \
            { i32x2e mv_; mv_.x = tile * 16 + (gi); mv_.y = st.head; pool.meta[pos] = mv_; }                                                    \
            st.head = pos;                                                                                            \
        }                                                                                                             \
        pool.cnt += __popcll(mask);                                                                                   \
    } while (0)

// Dual pass: a group whose maximum reaches the tile's column threshold goes to the pool as a column entry (code < 0, not
// linked into any chain; kz_flush_col3)
#define KZ_EPI3_APPEND_COL(gi, ev, mask)                                                                                      \
    do {                                                                                                                      \
        if (ev) {                                                                                                             \
            const int pos = pool.cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)((mask) >> 32),                               \
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)(mask), 0u));       \
            f32x4e kv;                                                                                                        \
            kv.x = acc[(gi) >> 2][4 * ((gi) & 3)];                                                                            \
            kv.y = acc[(gi) >> 2][4 * ((gi) & 3) + 1];                                                                        \
            kv.z = acc[(gi) >> 2][4 * ((gi) & 3) + 2];                                                                        \
            kv.w = acc[(gi) >> 2][4 * ((gi) & 3) + 3];                                                                        \
            pool.keys[pos] = kv;                                                                                              \
            i32x2e mv_;                                                                                                       \
            int lane_;   /* the lane number, made HERE (kept in a register for the whole sweep it was spilled) */              \
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_));                      \
            mv_.x = KZ_COL_FLAG | ((tile * 16 + (gi)) << 6) | lane_;                                                          \
            mv_.y = -1;                                                                                                       \
            pool.meta[pos] = mv_;                                                                                             \
        }                                                                                                                     \
        pool.cnt += (int)__popcll(mask);                                                                                      \
    } while (0)

// CAP = pool capacity (entries per wave).  sync = 4 LDS words of the workgroup (merge flags: a wave whose pool passes
// CAP / 2 asks every wave of the workgroup to merge at the start of the next tile, so that no wave merges alone while
// its siblings wait for it at the slice barrier; the flag of tile t is written during epilogue t, read during epilogue
// t+1, cleared during epilogue t+2, and a workgroup barrier lies between any two epilogues).
template <int KP, int CAP, int IN_LDS, bool DUAL, bool RECOMP>
__device__ __forceinline__ void kz_tile_epilogue3(f32x16 (&acc)[4], KzCandState3<IN_LDS>& st, KzWavePool& pool, KzBlockMin3<KP>& bs,
                                                  const int tile, const bool last_tile, kz_lds_i32* sync, const KzDualRef& du,
                                                  const float cthr) {
    const int t = ++pool.tiles_done;
    const bool sched = (t == pool.next_merge) || last_tile;  // block-uniform
    if (t == pool.next_merge) {
        // geometric schedule: the expected number of events until the next scheduled merge stays near 4 per query
        const int step = t * 4 / KP;
        pool.next_merge = t + (step > 0 ? step : 1);
    }
    {
        const bool together = __builtin_amdgcn_readfirstlane(sync[(t - 1) & 3]) != 0;
        if ((threadIdx.x & 63) == 0) {
            int zero = 0;
            asm volatile("" : "+v"(zero));   // materialised here: hipcc otherwise keeps a zero VGPR live across the whole tile loop
            sync[(t + 1) & 3] = zero;
        }
        if (together) KZ_EPI3_MERGE();
    }
    // 16 group maxima, tile maximum
    float gm[16];
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) {
        const int mt = gi >> 2, g4 = gi & 3;
        gm[gi] = fmaxf(fmaxf(acc[mt][4 * g4], acc[mt][4 * g4 + 1]), fmaxf(acc[mt][4 * g4 + 2], acc[mt][4 * g4 + 3]));
    }
    // ... four block maxima (one per 32-row block) on the way to the tile maximum: a tile with an event usually has it in
    // ONE block, and only that block's four groups are then looked at (4 + 4 ballots instead of 16)
    float bmx[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) bmx[mt] = fmaxf(fmaxf(gm[4 * mt], gm[4 * mt + 1]), fmaxf(gm[4 * mt + 2], gm[4 * mt + 3]));
    const float m = fmaxf(fmaxf(bmx[0], bmx[1]), fmaxf(bmx[2], bmx[3]));
    float tau_a = st.tau;
#ifdef KZ_ABL_NO_EVENTS   // (diagnostic build, tools/ab_build.sh: no list event is ever logged -- WRONG results, the sweep's time without events)
    const unsigned long long anym = 0ull;
#else
    const unsigned long long anym = __builtin_amdgcn_ballot_w64(m > tau_a);
#endif
    // Dual pass: the index rows of a tile are sorted by their event threshold, so ONE per-tile value (the tile's smallest
    // theta, plus this query's own offset: cthr) tested against the same maxima finds the groups that may hold an event of
    // an index row -- a second compare on the tile maximum, nothing per key.  Which of the four keys of such a group
    // really is an event of its row is decided after the launch (kz_dual_scatter_kernel).
    unsigned long long anyc = 0ull;
    if constexpr (DUAL) anyc = __builtin_amdgcn_ballot_w64(m >= cthr);
    if ((anym | anyc) != 0ull) {
        // room needed at most: four entries per (lane, block) pair with an event -- counted from the block maxima (a tile with
        // an event usually has ONE such pair; the older bound, 16 entries per lane with an event, sent many tiles down the
        // slow path once the pool carried the column entries of a dual pass)
        int worst = 0;
        if (anym != 0ull) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(bmx[mt] > tau_a));
        }
        if constexpr (DUAL) {
            if (anyc != 0ull) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) worst += 4 * (int)__popcll(__builtin_amdgcn_ballot_w64(bmx[mt] >= cthr));
            }
        }
        if (pool.cnt + worst <= CAP) {
            // common path: whatever the groups hold, the pool takes it -- no checks
            if (anym != 0ull) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (__builtin_amdgcn_ballot_w64(bmx[mt] > tau_a) == 0ull) continue;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int gi = 4 * mt + g4;
                        const bool ev = gm[gi] > tau_a;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
                        if (mask != 0ull) KZ_EPI3_APPEND(gi, ev, mask);
                    }
                }
            }
            if constexpr (DUAL) {
                if (anyc != 0ull) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        if (__builtin_amdgcn_ballot_w64(bmx[mt] >= cthr) == 0ull) continue;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int gi = 4 * mt + g4;
                            const bool evc = gm[gi] >= cthr;
                            const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                            if (maskc != 0ull) KZ_EPI3_APPEND_COL(gi, evc, maskc);
                        }
                    }
                }
            }
        } else {
            // slow path (first tiles of a sweep, bursts, a pool that is nearly full): merge when a group does not fit and
            // resume at that step with the fresher threshold (step = 2 group + kind: list entry, then column entry)
            int resume = 0;
            float cthr_l = cthr;
            for (;;) {
                bool need_room = false;
                // (the tile number re-read through an opaque copy: hoisted out of this loop, the 16 + 16 entry codes made from
                //  it sat in 31 scalar registers for the whole epilogue -- with the dual pass' extra pointers that pushed the
                //  kernel to re-load its arguments from memory at the top of EVERY tile)
                int tile_l = tile;
                asm volatile("" : "+s"(tile_l));
                const int tile = tile_l;
                // (opaque to the compiler: the 16 column masks are loop invariant, and hoisted out of this loop they held 32
                //  scalar registers -- SGPRs spilled into a VGPR the kernel does not have at three workgroups per CU)
                if constexpr (DUAL) asm volatile("" : "+v"(cthr_l));
#pragma unroll
                for (int gi = 0; gi < 16; ++gi) {
                    const bool ev = gm[gi] > tau_a;
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ev);
                    if (2 * gi >= resume && !need_room && mask != 0ull) {
                        if (pool.cnt + (int)__popcll(mask) > CAP) {
                            need_room = true;
                            resume = 2 * gi;
                        } else {
                            KZ_EPI3_APPEND(gi, ev, mask);
                        }
                    }
                    if constexpr (DUAL) {
                        const bool evc = gm[gi] >= cthr_l;
                        const unsigned long long maskc = __builtin_amdgcn_ballot_w64(evc);
                        if (2 * gi + 1 >= resume && !need_room && maskc != 0ull) {
                            if (pool.cnt + (int)__popcll(maskc) > CAP) {
                                need_room = true;
                                resume = 2 * gi + 1;
                            } else {
                                KZ_EPI3_APPEND_COL(gi, evc, maskc);
                            }
                        }
                    }
                }
                if (!need_room) break;
                KZ_EPI3_MERGE();
                tau_a = st.tau;
            }
        }
        if (pool.cnt > CAP / 2 && (threadIdx.x & 63) == 0) {
            int one = 1;
            asm volatile("" : "+v"(one));
            sync[t & 3] = one;
        }
    }
    if (sched) KZ_EPI3_MERGE();
}
