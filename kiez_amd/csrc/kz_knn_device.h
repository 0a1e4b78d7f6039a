// Device-side building blocks shared by the fused distance + candidate-selection kernels (kz_knn.hip: float32 operands,
// kz_knn_bf16.h: split-bf16, kz_knn_h16.h: fp16): parameter block, candidate list / log state, tile epilogues.
#pragma once
#include "kz_common.h"
#include "kz_plan.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// LDS-DMA: each lane copies 16 bytes from its own global address to (wave-uniform LDS base) + lane*16
__device__ __forceinline__ void kz_glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same with the address split the way the instruction takes it -- scalar 64-bit base + 32-bit per-lane byte offset
// (saddr form) -- and the LDS base placed in M0 by hand.  hipcc otherwise materialises base + offset as a 64-bit VGPR pair
// per lane and keeps it alive (and, at 168 VGPRs, spills and reloads it at every barrier).  Inline asm is invisible to
// hipcc's waitcnt pass: it only ever under-counts the wave's outstanding operations, which makes its own vmcnt waits
// stricter, never looser; completion of these copies is awaited explicitly (vmcnt(0) in front of the slice barrier).
__device__ __forceinline__ void kz_glds16_s(const void* sbase_uniform, unsigned lane_byte_off, float* lds_wave_base) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(lds), "v"(lane_byte_off), "s"(sbase_uniform)
                 : "memory", "m0");
}
// ... and the 4-byte form: lane l copies one dword from scalar base + its byte offset to (wave-uniform LDS base) + l*4
__device__ __forceinline__ void kz_glds4_s(const void* sbase_uniform, unsigned lane_byte_off, float* lds_wave_base) {
    const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                 :
                 : "s"(lds), "v"(lane_byte_off), "s"(sbase_uniform)
                 : "memory", "m0");
}
typedef float f32x4e __attribute__((ext_vector_type(4)));
typedef int i32x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 kz_nt_load4(const float4* p) {  // non-temporal 16-byte load (streaming cache policy)
    const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}


// ---------------------------------------------------------------------------------------------------
// Stage 1: fused similarity + candidate selection
// ---------------------------------------------------------------------------------------------------
// similarity key(q, y) = q.y + bias(y)     bias = -|y|^2/2 (euclidean: argmax key == argmin |q-y|^2), 0 (cosine)
//
// Workgroup = 256 threads = 4 waves, tile = 128 index rows (MFMA M) x 128 queries (MFMA N).
// Wave w owns queries [32w, 32w+32) and all 128 index rows of the tile: 4 accumulators of 32x32.
// With the query on the MFMA column (= lane & 31), the 64 keys a lane holds after a tile all belong to ONE
// query, so each lane keeps a PRIVATE candidate list (query, lane-half) and needs no atomics or barriers:
//   list(q, h) sees index rows with (row & 4) == 4h; union of the two halves' top-K' contains the top-K'.
// Index operand: streamed HBM/L2 -> registers -> LDS (double buffered 8 KiB slices, one barrier per slice).
// Query operand: per-lane fragments straight from L2 (no reuse across waves, so no LDS round trip).
// Candidate-list storage.  The host schedule (kz_knn) cuts the query tiles of a launch into a few REGIONS; every
// query of region r owns pieces[r] lists of 2*KP entries (one per index-range piece and lane half).
// (KzListLayout, KZ_MAX_REGIONS, KZ_QGROUP: kz_plan.h -- the host-only planning code, also compiled on the CPU with sanitizers)
__host__ __device__ __forceinline__ int kz_list_region(int64_t list_row, const KzListLayout& L) {
    const int qt = (int)(list_row / KZ_TILE);
    int r = 0;
    while (r + 1 < L.n_regions && qt >= L.qt_end[r]) ++r;
    return r;
}
// Lists are stored interleaved per WAVE: the 64 (query, lane-half) lists a wave owns for one piece form a block
// [KP entries][64 lanes], so entry e of all 64 lists is one 256-byte line -- the fused kernels' merges scan their lists
// with coalesced wave-loads (per-lane contiguous lists cost 64 cache lines per load instruction).
//   offset(list_row, piece, half h, entry e) = kz_list_wave_base(...) + e * 64 + h * 32 + list_row % 32
constexpr int KZ_LSTRIDE = 64;
__host__ __device__ __forceinline__ int64_t kz_list_wave_base(int64_t list_row, const KzListLayout& L, int KP, int piece) {
    const int r = kz_list_region(list_row, L);
    const int64_t row0 = r > 0 ? (int64_t)L.qt_end[r - 1] * KZ_TILE : 0;
    const int64_t wb = (list_row - row0) >> 5;  // 32 queries per wave
    return L.base[r] + ((wb * L.pieces[r] + piece) * (int64_t)KP) * KZ_LSTRIDE;
}

// Contiguous layout (fp16 kernel): offset(list_row, piece, entry e) = kz_list_contig_off(...) + e.  With the event pool a
// merge inserts for a FEW lanes at a time; an insert re-reads one block of K'/8 keys: 32 contiguous bytes here, against
// eight 128-byte lines in the interleaved layout (C3, K' = 64, lists in the output arrays: 337 GB of fabric reads per
// launch, profiles/r02_c3_pmc.jsonl) -- and the finalize kernel gathers a query's entries as one contiguous run.
__host__ __device__ __forceinline__ int64_t kz_list_contig_off(int64_t list_row, const KzListLayout& L, int KP, int piece) {
    const int r = kz_list_region(list_row, L);
    const int64_t row0 = r > 0 ? (int64_t)L.qt_end[r - 1] * KZ_TILE : 0;
    return L.base[r] + ((list_row - row0) * L.pieces[r] + piece) * (int64_t)KP;
}

struct KnnCandParams {
    const float* qpack;   // packed query matrix
    const float* ypack;   // packed index matrix
    const float* ybias;   // accumulator init per index row
    const int4* work;     // one descriptor per workgroup: {query tile (local), first index tile, end index tile, list slot}
    int qt0;              // first query tile of this launch (global tile index into qpack)
    int n_ytiles;         // index tiles
    int n_qtiles;         // query tiles of this launch (wide workgroups: the last one may reach past them)
    KzListLayout lay;     // candidate-list layout of this launch (kz_list_base)
    int kg;               // k-groups (of 4) per row; slices per tile = kg / 4
    float* out_key;       // per region: [query rows][pieces][2 lane halves][KP]
    int* out_idx;
    // dual pass (fp16 kernel, DUAL build; kz_knn_epi3.h "Dual pass")
    const float* theta;            // [n_ytiles * 128] event threshold per index row (+inf on pad rows)
    const float* qnbias;           // [query rows incl. padding, global row numbers] -bias(q) (+inf on pad rows)
    void* log_keys;                // [log_cap] x 16 B
    void* log_meta;                // [log_cap] x 8 B
    unsigned long long* log_cnt;
    long long log_cap;
    // seeded lists (fp16 kernels): [query rows incl. padding, global row numbers] a list starts FULL of (qfloor, no row) entries
    // instead of (-inf, no row): keys at or below the floor never become events.  The finalize kernel gets the same array
    // (KnnFinParams::list_floor) and counts the floor into the bound on the rows outside the lists.  nullptr: -inf.
    const float* qfloor;
};

constexpr int KZ_CAND_LDS_BASE = 16384 + 1024;  // 2 index slices + 2 bias rows
constexpr int KZ_LOG_CAP = 16;                  // per-lane candidate log entries (keys + rows: 32 KiB per workgroup)
constexpr int KZ_CAND_LDS = KZ_CAND_LDS_BASE + KZ_LOG_CAP * 256 * 8;

#ifndef KZ_SCAN_CHUNK
#define KZ_SCAN_CHUNK 16  // list keys fetched per batch in kz_list_replace_min (register temporaries of the merge path)
#endif

// Replace the minimum of an unsorted K'-entry list by (v, idx) and find the new minimum.  All keys are fetched
// before the compare chain starts so that the LDS latency is paid once, not per element.
template <int KP, int LSTRIDE, int CHUNK = KZ_SCAN_CHUNK>
__device__ __forceinline__ void kz_list_replace_min(float* lk, int* li, float v, int idx, float& tau, int& minpos) {
    lk[minpos * LSTRIDE] = v;
    li[minpos * LSTRIDE] = idx;
    float mn = INFINITY;
    int mp = 0;
#pragma unroll
    for (int c0 = 0; c0 < KP; c0 += CHUNK) {
        float kk[CHUNK];
#pragma unroll
        for (int e = 0; e < CHUNK; ++e) kk[e] = lk[(c0 + e) * LSTRIDE];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < CHUNK; ++e) {
            if (kk[e] < mn) {
                mn = kk[e];
                mp = c0 + e;
            }
        }
    }
    tau = mn;
    minpos = mp;
}



__device__ __forceinline__ void kz_wave_sync() {
    // cross-lane exchange through LDS inside ONE wave: LDS ops of a wave execute in order, the fences only stop
    // the compiler from reordering the accesses.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-lane candidate state of one (query, lane-half) pair (see kz_knn_cand_kernel).
struct KzCandState {
    float* lk;   // list keys  (global, K' entries, unsorted)
    int* li;     // list rows
    float* sk;   // log keys   (LDS, stride 256)
    int* si;     // log rows
    float tau;   // K'-th best key of the list as of the last merge
    int minpos;
    int cnt;     // log entries
    int tiles_done, next_merge;
};

// Tile epilogue shared by both fused kernels.  C layout of the 32x32 MFMA: col = lane & 31 (query),
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Lanes l and l+32 hold the two half-lists of ONE query: a key below the OTHER
// half's K'-th best cannot be in the merged top-K' either, so both halves prune with the larger of the two thresholds.
template <int KP, int CAP = KZ_LOG_CAP>
__device__ __forceinline__ void kz_tile_epilogue(f32x16 (&acc)[4], KzCandState& st, const int tile, const bool last_tile,
                                                 const int h) {
    float tau_eff = fmaxf(st.tau, __shfl_xor(st.tau, 32, 64));
    const int rowbase = tile * KZ_TILE + 4 * h;
    ++st.tiles_done;
    const bool sched = (st.tiles_done == st.next_merge) || last_tile;  // block-uniform
    unsigned long long done = 0ull;  // elements of this tile already logged (bit 16*mt + r)
    for (;;) {
        bool ovf = false;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float m4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                m4[g4] = fmaxf(fmaxf(acc[mt][4 * g4], acc[mt][4 * g4 + 1]), fmaxf(acc[mt][4 * g4 + 2], acc[mt][4 * g4 + 3]));
            const float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            if (m > tau_eff) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (m4[g4] > tau_eff) {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = 4 * g4 + r4;
                            const float v = acc[mt][r];
                            const unsigned long long bit = 1ull << (16 * mt + r);
                            if (v > tau_eff && !(done & bit)) {
                                if (st.cnt < CAP) {
                                    int rb = rowbase;
                                    asm volatile("" : "+v"(rb));  // keep the 64 row ids out of registers: computed on demand
                                    st.sk[st.cnt * 256] = v;
                                    st.si[st.cnt * 256] = rb + 32 * mt + (r & 3) + 8 * (r >> 2);
                                    ++st.cnt;
                                    done |= bit;
                                } else {
                                    ovf = true;
                                }
                            }
                        }
                    }
                }
            }
        }
        const bool any_ovf = __any(ovf);
        if (!any_ovf && !sched) break;
        // merge the log into the list (all lanes of the wave take part; trip counts differ per lane)
        for (int e = 0; e < st.cnt; ++e) {
            const float v = st.sk[e * 256];
            if (v > st.tau) kz_list_replace_min<KP, KZ_LSTRIDE>(st.lk, st.li, v, st.si[e * 256], st.tau, st.minpos);
        }
        st.cnt = 0;
        tau_eff = fmaxf(st.tau, __shfl_xor(st.tau, 32, 64));
        if (!any_ovf) break;  // (a scheduled merge after an overflow round happens on the next pass)
    }
    if (st.tiles_done == st.next_merge) {
        const int step = st.tiles_done * CAP / KP;
        st.next_merge = st.tiles_done + (step > 0 ? step : 1);
    }
}


// Two-level minimum of an unsorted K'-entry list: the list is cut into NB blocks, the minimum of every block (value and
// position inside the block) is kept in registers.  Replacing the global minimum then touches ONE block: write the new
// entry, re-read that block (K'/NB keys, one L2 round trip), refresh its minimum -- instead of re-scanning all K' keys.
template <int KP>
struct KzBlockMin {
    static constexpr int NB = KP >= 64 ? 8 : 4;
    static constexpr int BS = KP / NB;
    float bm[NB];
    int bp[NB];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            bm[i] = -INFINITY;
            bp[i] = 0;
        }
    }
};

// Insert (v, idx) over the current minimum (caller guarantees v > tau); returns the new minimum in tau.
template <int KP, int LSTRIDE>
__device__ __forceinline__ void kz_list_insert_blocked(float* lk, int* li, KzBlockMin<KP>& bs, float v, int idx, float& tau) {
    constexpr int NB = KzBlockMin<KP>::NB, BS = KzBlockMin<KP>::BS;
    float m = bs.bm[0];
    int b = 0;
#pragma unroll
    for (int i = 1; i < NB; ++i) {
        if (bs.bm[i] < m) {
            m = bs.bm[i];
            b = i;
        }
    }
    int pos = bs.bp[0];
#pragma unroll
    for (int i = 1; i < NB; ++i) pos = (b == i) ? bs.bp[i] : pos;
    float* blk = lk + (b * BS) * LSTRIDE;
    float kk[BS];
#pragma unroll
    for (int jj = 0; jj < BS; ++jj) kk[jj] = blk[jj * LSTRIDE];   // issued before the store: its slot is patched below
    __builtin_amdgcn_sched_barrier(0);
    blk[pos * LSTRIDE] = v;
    li[(b * BS + pos) * LSTRIDE] = idx;
    float nm = INFINITY;
    int np = 0;
#pragma unroll
    for (int jj = 0; jj < BS; ++jj) {
        const float x = (jj == pos) ? v : kk[jj];
        if (x < nm) {
            nm = x;
            np = jj;
        }
    }
    float t = INFINITY;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        bs.bm[i] = (b == i) ? nm : bs.bm[i];
        bs.bp[i] = (b == i) ? np : bs.bp[i];
        t = fminf(t, bs.bm[i]);
    }
    tau = t;
}

// Merge of the two lane-half logs of a query into ONE list (split-bf16 kernels).  Lanes l and l+32 see disjoint index
// rows of the same query; with a list each, the pruning threshold of a half is the K'-th best of HALF the rows, and a
// query logs about 2 K' ln(N/2K') events.  With one shared list the threshold is the K'-th best of ALL rows seen: about
// K' ln(N/K') events per query -- half the appends, half the merge inserts, half the list bytes finalize reads.
// Lane l (< 32) inserts its own log and then its partner's (the logs are lane-strided in LDS: the partner's entries
// sit 32 words further); lanes >= 32 idle here.  Both halves leave with the same threshold.
template <int KP>
__device__ __forceinline__ void kz_merge_logs_shared(KzCandState& st, KzBlockMin<KP>& bs) {
    const int lane = threadIdx.x & 63;
    const int other = __shfl_xor(st.cnt, 32, 64);
    if (lane < 32) {
        for (int e = 0; e < st.cnt; ++e) {
            const float v = st.sk[e * 256];
            if (v > st.tau) kz_list_insert_blocked<KP, KZ_LSTRIDE>(st.lk, st.li, bs, v, st.si[e * 256], st.tau);
        }
        for (int e = 0; e < other; ++e) {
            const float v = st.sk[e * 256 + 32];
            if (v > st.tau) kz_list_insert_blocked<KP, KZ_LSTRIDE>(st.lk, st.li, bs, v, st.si[e * 256 + 32], st.tau);
        }
    }
    st.cnt = 0;
    st.tau = __shfl(st.tau, lane & 31, 64);
}

// Tile epilogue, second form (used by the split-bf16 kernel, where the epilogue is no longer hidden under MFMA time).
// Work is proportional to the number of candidate EVENTS instead of the number of values:
//   * the 16 groups of four values are tested first, back to back (max3 + max + compare each, 16 wave-level masks in
//     SGPRs), so the tests do not form a dependent compare -> branch chain per group;
//   * only groups in which some lane has an event are entered; inside, the four values are appended by straight-line
//     code executed by all lanes (lanes without an event write to a scratch row of the log) -- the log can never
//     overflow inside a group because a group is only entered when every lane has room for four entries;
//   * if some lane lacks that room the wave leaves the scan at that group, merges its logs (which also refreshes the
//     threshold) and resumes the scan at the same group: no per-lane bookkeeping of what was already logged.
//   * merges are made workgroup-synchronous: the waves of a workgroup meet at a barrier every two slices, so a wave
//     that merges alone (~16k cycles) stalls its three siblings.  A wave whose fullest log passes CAP - 8 raises a flag
//     in LDS (sync[tile & 3]); every wave reads it at the start of the NEXT tile's epilogue and merges then, together.
//     (The flag of tile t is written during epilogue t, read during epilogue t+1, cleared during epilogue t+2; waves of
//     a workgroup are never more than one tile apart, and a workgroup barrier lies between any two epilogues.)
// a wave asks for a workgroup-wide merge at the next tile once its fullest log passes this many rows
#define KZ_MERGE_FLAG(CAP) ((CAP) >= 16 ? (CAP) - 8 : (CAP) / 2)
template <int KP, int CAP>
__device__ __forceinline__ void kz_tile_epilogue2(f32x16 (&acc)[4], KzCandState& st, KzBlockMin<KP>& bs, const int tile,
                                                  const bool last_tile, const int h, int* sync, const int resume0) {
    ++st.tiles_done;
    const bool sched = (st.tiles_done == st.next_merge) || last_tile;  // block-uniform
    {
        const int t = st.tiles_done;
        const bool together = __builtin_amdgcn_readfirstlane(sync[(t - 1) & 3]) != 0;
        if ((threadIdx.x & 63) == 0) sync[(t + 1) & 3] = 0;
        if (together) {
            kz_merge_logs_shared<KP>(st, bs);
        }
    }
    if (resume0 >= 16 && !sched) {
        // nothing left to scan (an overlapped scan covered all 16 groups) and no merge due: only the bookkeeping
        if (__any(st.cnt > KZ_MERGE_FLAG(CAP)) && (threadIdx.x & 63) == 0) sync[st.tiles_done & 3] = 1;
        return;
    }
    float tau_a = fmaxf(st.tau, __shfl_xor(st.tau, 32, 64));
    const int rowbase = tile * KZ_TILE + 4 * h;
    unsigned long long gm[16];
    // Group maxima by v_max3 / v_max in inline asm (fmaxf() would first canonicalise every MFMA result: 4 extra VALU
    // per group).  hipcc's hazard recognizer does not see through inline asm, and an MFMA result must not be read by
    // a VALU instruction for up to 19 wait states after the MFMA issued: the volatile statement below takes all four
    // accumulators as in/out operands -- every MFMA precedes it, every later read of acc follows it -- and spends
    // those wait states explicitly.
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) {
        const int mt = gi >> 2, g4 = gi & 3;
        float m;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[mt][4 * g4]), "v"(acc[mt][4 * g4 + 1]), "v"(acc[mt][4 * g4 + 2]));
        asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(acc[mt][4 * g4 + 3]));
        gm[gi] = __builtin_amdgcn_ballot_w64(m > tau_a);
    }
    int resume = resume0;  // first group not yet scanned (wave-uniform; > 0 when an overlapped scan already did the rest)
    for (;;) {
        bool need_room = false;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) {
            const int mt = gi >> 2, g4 = gi & 3;
            if (gi >= resume && !need_room && gm[gi] != 0ull) {
                if (__any(st.cnt > CAP - 4)) {
                    need_room = true;
                    resume = gi;
                } else {
                    float ta = tau_a;
                    asm volatile("" : "+v"(ta));  // keeps the per-value compares inside the (rarely taken) branch
                    int rb = rowbase;
                    asm volatile("" : "+v"(rb));      // row ids are computed on demand, not kept in registers
                    // straight-line, all lanes: a lane without an event writes to the scratch row CAP (never read),
                    // so there is no exec juggling and no dependent compare -> saveexec -> branch chain per value
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float v = acc[mt][4 * g4 + r4];
                        const bool ev = v > ta;
                        const int slot = ev ? st.cnt : CAP;
                        st.sk[slot * 256] = v;
                        st.si[slot * 256] = rb + 32 * mt + 8 * g4 + r4;
                        st.cnt += ev ? 1 : 0;
                    }
                }
            }
        }
        if (!need_room && !sched) break;
        // merge the log into the list (all lanes of the wave take part; trip counts differ per lane)
        kz_merge_logs_shared<KP>(st, bs);
        if (!need_room) break;
        tau_a = fmaxf(st.tau, __shfl_xor(st.tau, 32, 64));  // fresher threshold for the rest of the tile
    }
    if (__any(st.cnt > KZ_MERGE_FLAG(CAP)) && (threadIdx.x & 63) == 0) sync[st.tiles_done & 3] = 1;
    if (st.tiles_done == st.next_merge) {
        const int step = st.tiles_done * CAP / KP;
        st.next_merge = st.tiles_done + (step > 0 ? step : 1);
    }
}

// One group of the candidate scan as a free-standing step (overlapped kernel: the scan of tile t-1 is issued between
// the MFMA groups of tile t, in the shadow of the matrix pipe).  `mask` is this group's event mask, computed one step
// earlier so that the compare -> branch latency is covered too; `stop` is the first group NOT scanned here because
// some lane had no room left in its log (the tail epilogue merges and resumes there).
template <int CAP>
__device__ __forceinline__ unsigned long long kz_epi_group_mask(const f32x16 (&acc)[4], const int gi, const float tau_a) {
    const int mt = gi >> 2, g4 = gi & 3;
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(acc[mt][4 * g4]), "v"(acc[mt][4 * g4 + 1]), "v"(acc[mt][4 * g4 + 2]));
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(acc[mt][4 * g4 + 3]));
    return __builtin_amdgcn_ballot_w64(m > tau_a);
}

template <int CAP>
__device__ __forceinline__ void kz_epi_step(const f32x16 (&acc)[4], KzCandState& st, const float tau_a, const int rowbase,
                                            const int gi, const unsigned long long mask, int& stop) {
    if (gi < stop && mask != 0ull) {
        if (__any(st.cnt > CAP - 4)) {
            stop = gi;
        } else {
            const int mt = gi >> 2, g4 = gi & 3;
            float ta = tau_a;
            asm volatile("" : "+v"(ta));
            int rb = rowbase;
            asm volatile("" : "+v"(rb));
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float v = acc[mt][4 * g4 + r4];
                const bool ev = v > ta;
                const int slot = ev ? st.cnt : CAP;
                st.sk[slot * 256] = v;
                st.si[slot * 256] = rb + 32 * mt + 8 * g4 + r4;
                st.cnt += ev ? 1 : 0;
            }
        }
    }
}
