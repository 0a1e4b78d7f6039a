// CPU-only check of the host planning code of libkiez_amd.so (kiez_amd/csrc/kz_plan.h), built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_host_sanitize.py (sanitizers run on the CPU build only).  For a sweep of shapes it plans a
// pass, fills the work table into an exactly-sized heap block (ASan sees any out-of-range item) and checks the invariants the
// kernels and the finalize kernel rely on:
//   * every (query tile, index tile) pair is covered by exactly one work item;
//   * regions partition the query tiles, end on unit boundaries (wide workgroups), pieces tile the index exactly;
//   * list offsets of distinct (query row, piece) pairs never overlap and stay inside the element count of the plan;
//   * items of one XCD label (block id % 8) are consecutive in the logical order.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../kiez_amd/csrc/kz_plan.h"

static int fails = 0;
#define CHECK(cond, ...)                                  \
    do {                                                  \
        if (!(cond)) {                                    \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                     \
            std::printf("\n");                            \
            ++fails;                                      \
        }                                                 \
    } while (0)

static int region_of(const KzListLayout& L, int qt) {
    int r = 0;
    while (r + 1 < L.n_regions && qt >= L.qt_end[r]) ++r;
    return r;
}

static void check(int n_qtiles, int n_ytiles, int slots, int KP, int tpw, int force_splits, int min_splits) {
    const int max_pieces = (4096 / KP) < 128 ? (4096 / KP) : 128;   // (kz_max_pieces: KZ_MAX_PIECES = 128)
    KzPlan pl;
    kz_plan_pass(n_qtiles, n_ytiles, slots, max_pieces, KP, 1, 1, tpw, force_splits, min_splits, &pl);
    const KzListLayout& L = pl.lay;
    CHECK(L.n_regions >= 1 && L.n_regions <= KZ_MAX_REGIONS, "regions %d", L.n_regions);
    CHECK(L.qt_end[L.n_regions - 1] == n_qtiles, "last region ends at %d, not %d", L.qt_end[L.n_regions - 1], n_qtiles);
    int prev = 0;
    for (int r = 0; r < L.n_regions; ++r) {
        CHECK(L.qt_end[r] > prev, "region %d empty", r);
        CHECK(r == L.n_regions - 1 || L.qt_end[r] % tpw == 0, "region %d ends inside a unit", r);
        CHECK(L.pieces[r] >= 1 && L.pieces[r] <= max_pieces, "pieces %d", L.pieces[r]);
        prev = L.qt_end[r];
    }
    std::vector<KzWorkItem>* hw = new std::vector<KzWorkItem>((size_t)pl.W);   // exactly W items: ASan guards both ends
    for (auto& w : *hw) w = KzWorkItem{-1, -1, -1, -1};
    kz_plan_fill_work(pl, n_ytiles, tpw, hw->data());
    std::vector<unsigned char> cover((size_t)n_qtiles * n_ytiles, 0);
    for (int i = 0; i < pl.W; ++i) {
        const KzWorkItem w = (*hw)[i];
        CHECK(w.x >= 0 && w.x < n_qtiles && w.x % tpw == 0, "item %d: tile %d", i, w.x);
        CHECK(w.y >= 0 && w.y < w.z && w.z <= n_ytiles, "item %d: range [%d, %d)", i, w.y, w.z);
        const int r = region_of(L, w.x);
        CHECK(w.w >= 0 && w.w < L.pieces[r], "item %d: piece %d of %d", i, w.w, L.pieces[r]);
        for (int b = 0; b < tpw && w.x + b < n_qtiles; ++b) {
            CHECK(region_of(L, w.x + b) == r, "item %d: unit straddles regions", i);
            for (int t = w.y; t < w.z; ++t) ++cover[(size_t)(w.x + b) * n_ytiles + t];
        }
    }
    for (size_t e = 0; e < cover.size(); ++e)
        if (cover[e] != 1) {
            CHECK(false, "pair (%zu, %zu) covered %d times", e / n_ytiles, e % n_ytiles, (int)cover[e]);
            break;
        }
    // list offsets (contiguous layout): [base + ((row - row0) * pieces + piece) * KP, + KP) -- disjoint, inside list_elems
    size_t expect = 0;
    for (int r = 0; r < L.n_regions; ++r) {
        CHECK((size_t)L.base[r] == expect, "region %d base %lld, expected %zu", r, L.base[r], expect);
        const int t0 = r > 0 ? L.qt_end[r - 1] : 0;
        expect += (size_t)(L.qt_end[r] - t0) * KZ_TILE * L.pieces[r] * KP;
    }
    CHECK(expect == pl.list_elems, "list elements %zu vs %zu", expect, pl.list_elems);
    delete hw;
}

int main() {
    unsigned s = 12345;
    auto rnd = [&](int lo, int hi) {
        s = s * 1664525u + 1013904223u;
        return lo + (int)((s >> 8) % (unsigned)(hi - lo + 1));
    };
    const int KPs[4] = {16, 32, 64, 128};
    int n = 0;
    // the BASELINE shapes at every occupancy class, narrow and wide
    const int shapes[][2] = {{782, 782}, {1954, 7813}, {7813, 1954}, {3907, 3907}, {1, 782}, {8, 313}, {1, 1}, {3, 2}, {2000, 8}};
    for (auto& sh : shapes)
        for (int slots : {256, 512, 768})
            for (int tpw : {1, 2, 3})
                for (int KP : KPs) {
                    check(sh[0], sh[1], slots, KP, tpw, 0, 1);
                    ++n;
                }
    for (int i = 0; i < 3000; ++i) {
        const int nq = rnd(1, i % 3 == 0 ? 9000 : 300), ny = rnd(1, i % 5 == 0 ? 9000 : 400);
        if ((long long)nq * ny > 4000000) continue;
        check(nq, ny, rnd(1, 1024), KPs[rnd(0, 3)], rnd(1, 3), i % 7 == 0 ? rnd(1, 80) : 0, rnd(1, 12));
        ++n;
    }
    std::printf("%d plans checked, %d failures\n", n, fails);
    return fails ? 1 : 0;
}
