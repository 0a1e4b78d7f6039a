// Host-only planning of one fused-kernel launch (no HIP in this header: tests/host/plan_sanitize.cpp compiles it with
// g++ -fsanitize=address,undefined on the CPU).  kz_knn.hip includes it for the product.
//
//   kz_plan_rounds  -- greedy rounds (which query tiles sweep which index ranges)
//   kz_plan_pass    -- rounds -> candidate-list layout (regions, pieces, element offsets) + the work table, one item per
//                      workgroup, in the XCD-aware order the kernels are dispatched in
#pragma once
#include <cstdint>
#include <cstring>

#ifndef KZ_TILE
#define KZ_TILE 128
#endif
constexpr int KZ_MAX_REGIONS = 8;
constexpr int KZ_QGROUP = 24;
constexpr int KZ_PLAN_START_TILES = 24;   // what the start of an item costs, in tiles of steady sweep (kz_plan_rounds)   // query tiles sharing an XCD at a time (x splits ~= resident workgroups)

// Candidate-list storage.  The host schedule cuts the query tiles of a launch into a few REGIONS; every query of region r owns
// pieces[r] lists (one per index-range piece; two per piece -- one per lane half -- for the float32 kernels).
struct KzListLayout {
    int n_regions;
    int qt_end[KZ_MAX_REGIONS];      // region r = query tiles [qt_end[r-1], qt_end[r])  (local tile numbers)
    int pieces[KZ_MAX_REGIONS];      // index-range pieces per query tile
    int halves;                      // lists per (query, piece): 2 = one per lane half (float32 kernels), 1 = one shared
                                     // by both lane halves (split-bf16 kernels; column h = 1 of a list block is unused)
    int contig;                      // 1: every list is K' CONTIGUOUS entries (fp16 kernel, kz_list_contig_off); 0: the
                                     // wave-interleaved layout (kz_list_wave_base)
    long long base[KZ_MAX_REGIONS];  // element offset of the region's first list
};

struct KzWorkItem {   // = int4 on the device: {first query tile (local), first index tile, end index tile, list slot (piece)}
    int x, y, z, w;
};

// Host schedule of one launch: greedy rounds.  slots = workgroups resident on the chip; dispatch is in block-id order, so
// the items of one round start together and sweep the index in lockstep (each index tile is fetched into L2 once per
// round).  With R query tiles left, a round cuts the index into s = ceil(slots / R) ranges and takes slots / s query
// tiles: every round fills the chip with equal-length items, the items shrink from round to round and only the last few
// query tiles get the shortest allowed ranges (>= 8 index tiles, <= max_pieces ranges).  force_splits (test knob) = one
// round with exactly that many ranges; min_splits raises the first round's range count (L2 grouping knob).
// Outputs: per round the number of query tiles and the requested range count (the actual number of ranges is
// ceil(n_ytiles / ceil(n_ytiles / s))).
static inline void kz_plan_rounds(int n_qtiles, int n_ytiles, int slots, int max_pieces, int force_splits, int min_splits,
                                  int* n_rounds, int* round_qtiles, int* round_splits) {
    const int by_len = n_ytiles / 8 > 1 ? n_ytiles / 8 : 1;
    auto clamp_s = [&](int v) {
        if (v > max_pieces) v = max_pieces;
        if (v > by_len) v = by_len;
        if (v < 1) v = 1;
        return v;
    };
    auto split_len = [&](int sp) { return (n_ytiles + sp - 1) / sp; };
    auto split_cnt = [&](int sp) { return (n_ytiles + split_len(sp) - 1) / split_len(sp); };
    int R = n_qtiles, n = 0;
    // ONE ROUND where that is shorter (round 6).  Every item pays the start of a sweep -- its lists fill from -inf: the first tile of an
    // item takes ~35 us, the second ~18, ... ~0.1 ms in all before the steady 2.3 - 4.3 us per tile (per-tile clock stamps, tools/
    // stamp_show.py) -- about KZ_PLAN_START_TILES tiles' worth.  The greedy rounds below cut a SMALL launch (fewer query tiles than
    // slots) into ceil(slots / R) ranges, which leaves a few query tiles for a second round of short items that pay that start
    // again behind the first (15 k x 15 k x 300: 510 items of 24 tiles, then 224 of 9: 336 us, the second round 100 of them);
    // floor(slots / R) ranges put every query tile into one round (472 items of 30 tiles).  Both plans are priced -- start + length
    // per round -- and the cheaper one is taken.
    if (force_splits <= 0 && n_qtiles < slots) {
        const int sp1 = clamp_s(slots / n_qtiles > min_splits ? slots / n_qtiles : min_splits);
        if (split_cnt(sp1) * n_qtiles <= slots) {
            int cost_rounds = 0, Rg = n_qtiles, ng = 0;
            while (Rg > 0) {   // (the greedy rounds, priced only)
                int spg = clamp_s((slots + Rg - 1) / Rg);
                if (ng == 0 && spg < min_splits) spg = clamp_s(min_splits);
                int Ag = slots / split_cnt(spg);
                if (Ag < 1) Ag = 1;
                if (Ag > Rg || ng == KZ_MAX_REGIONS - 1) Ag = Rg;
                cost_rounds += KZ_PLAN_START_TILES + split_len(spg);
                Rg -= Ag;
                ++ng;
            }
            if (KZ_PLAN_START_TILES + split_len(sp1) <= cost_rounds) {
                round_qtiles[0] = n_qtiles;
                round_splits[0] = sp1;
                *n_rounds = 1;
                return;
            }
        }
    }
    while (R > 0) {
        int sp, A;
        if (force_splits > 0) {
            sp = force_splits < n_ytiles ? force_splits : n_ytiles;
            if (sp > max_pieces) sp = max_pieces;
            A = R;
        } else {
            sp = clamp_s((slots + R - 1) / R);
            if (n == 0 && sp < min_splits) sp = clamp_s(min_splits);
            A = slots / split_cnt(sp);
            if (A < 1) A = 1;
            if (A > R || n == KZ_MAX_REGIONS - 1) A = R;
        }
        round_qtiles[n] = A;
        round_splits[n] = sp;
        ++n;
        R -= A;
    }
    *n_rounds = n;
}

// The plan of one pass.  tpw = query tiles per workgroup (wide fp16 builds: 2 or 3, kz_knn_h16.h "WIDE"): rounds are planned for
// UNITS of tpw consecutive query tiles -- one work item = one unit x one index range, x = its first tile -- and converted back
// to tiles for the list layout (a region ends on a unit boundary, the last one at the last tile).  entries_per_list = K' in
// the contiguous layout, 2 K' in the interleaved one.
struct KzPlan {
    KzListLayout lay;
    int W;                  // work items = workgroups
    size_t list_elems;      // candidate-list entries of the whole launch
    int reg_q0[KZ_MAX_REGIONS], reg_nq[KZ_MAX_REGIONS], reg_s[KZ_MAX_REGIONS], reg_w0[KZ_MAX_REGIONS];   // (units)
};

static inline void kz_plan_pass(int n_qtiles, int n_ytiles, int slots, int max_pieces, int entries_per_list, int halves, int contig,
                                int tpw, int force_splits, int min_splits, KzPlan* out) {
    const int n_units = (n_qtiles + tpw - 1) / tpw;
    auto split_len = [&](int sp) { return (n_ytiles + sp - 1) / sp; };
    auto split_cnt = [&](int sp) { return (n_ytiles + split_len(sp) - 1) / split_len(sp); };
    KzListLayout lay;
    memset(&lay, 0, sizeof(lay));
    int W = 0, n_reg = 0;
    size_t list_elems = 0;
    kz_plan_rounds(n_units, n_ytiles, slots, max_pieces, force_splits, min_splits, &n_reg, out->reg_nq, out->reg_s);
    int q0 = 0;   // (units)
    for (int r = 0; r < n_reg; ++r) {
        out->reg_q0[r] = q0;
        out->reg_w0[r] = W;
        const int t0 = q0 * tpw;
        const int t1 = (q0 + out->reg_nq[r]) * tpw < n_qtiles ? (q0 + out->reg_nq[r]) * tpw : n_qtiles;
        lay.qt_end[r] = t1;
        lay.pieces[r] = split_cnt(out->reg_s[r]);
        lay.base[r] = (long long)list_elems;
        list_elems += (size_t)(t1 - t0) * KZ_TILE * (size_t)(lay.pieces[r] * entries_per_list);
        W += out->reg_nq[r] * lay.pieces[r];
        q0 += out->reg_nq[r];
    }
    lay.n_regions = n_reg;
    lay.halves = halves;
    lay.contig = contig;
    out->lay = lay;
    out->W = W;
    out->list_elems = list_elems;
}

// The work table of a plan: hw[W].  Logical order inside a region: groups of KZ_QGROUP query units, inside a group
// split-major.  The workgroups resident on one XCD then cover few query tiles (their fragments stay in the 4 MiB L2) times a
// few index ranges (each index tile is fetched once and hit by the whole group); items are spread over block ids so that
// blocks with equal (id % 8) -- one XCD -- take consecutive items.  `qgroup`: the fp16 kernel keeps its query fragments in
// registers, so its groups are several times what an XCD holds -- with many index ranges per query tile (short-list
// route: 10) the workgroups resident on an XCD then stream ONE range together instead of four.
static inline void kz_plan_fill_work(const KzPlan& pl, int n_ytiles, int tpw, KzWorkItem* hw, int qgroup = KZ_QGROUP) {
    auto split_len = [&](int sp) { return (n_ytiles + sp - 1) / sp; };
    auto split_cnt = [&](int sp) { return (n_ytiles + split_len(sp) - 1) / split_len(sp); };
    for (int r = 0; r < pl.lay.n_regions; ++r) {
        const int off = pl.reg_w0[r], nq = pl.reg_nq[r], q0 = pl.reg_q0[r], sp = pl.reg_s[r];
        const int cnt = nq * pl.lay.pieces[r];
        if (cnt == 0) continue;
        const int len = split_len(sp);
        const int nsp = split_cnt(sp);
        const int G = qgroup < nq ? qgroup : nq;
        int next = 0;
        for (int label = 0; label < 8; ++label) {
            for (int i = 0; i < cnt; ++i) {
                if (((off + i) & 7) != label) continue;
                const int grp = next / (G * nsp);
                const int gq0 = grp * G;
                const int gsz = (nq - gq0) < G ? (nq - gq0) : G;  // last group may be smaller
                const int within = next - grp * G * nsp;
                const int sidx = within / gsz, qt = q0 + gq0 + within % gsz;
                ++next;
                KzWorkItem w4;
                w4.x = qt * tpw;   // first query tile of the unit
                w4.y = sidx * len;
                w4.z = (sidx + 1) * len < n_ytiles ? (sidx + 1) * len : n_ytiles;
                w4.w = sidx;
                hw[off + i] = w4;
            }
        }
    }
}
