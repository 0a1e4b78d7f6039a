// The ONE table of context options (kz_ctx_set_option, kz_runtime.hip): name, field of kz_ctx, type, allowed range, default.
//
// PUBLIC options -- the contract of include/kiez_amd.h -- are the four flagged KZ_OPT_PUBLIC: "precision", "dual_stride",
// "dual_max_gb", "eps_scale".  Everything else is an INTERNAL tuning or diagnostic knob of this build (tools/, tests/, the A/B logs
// under profiles/ switch them): accepted by the same entry point, not part of the ABI's promise, free to change or disappear
// between versions.  Whatever the options say, every route returns the same results -- they decide how fast, never what.
#pragma once
#include <cstddef>

#include "kz_common.h"

enum { KZ_OPT_INT = 0, KZ_OPT_BOOL = 1, KZ_OPT_F64 = 2 };
enum { KZ_OPT_PUBLIC = 1, KZ_OPT_SET = 2 /* allowed = the listed values, not the range */ };

struct KzOption {
    const char* name;
    int kind;
    size_t offset;      // of the field in kz_ctx (int for INT / BOOL, double for F64)
    double lo, hi;      // allowed range (inclusive); BOOL: any value, stored as value != 0
    double dflt;
    unsigned flags;
    double allowed[4];  // KZ_OPT_SET: the allowed values (n_allowed of them) -- in addition to the range when n_allowed < 0
    int n_allowed;
    const char* doc;
};

#define KZ_O(field) offsetof(kz_ctx, field)
static const KzOption KZ_OPTIONS[] = {
    // ---- public -----------------------------------------------------------------------------------------------------------------
    {"precision", KZ_OPT_INT, KZ_O(precision), 0, 2, 0, KZ_OPT_PUBLIC, {}, 0,
     "first-pass operands: 0 fp16 on centred operands (default), 2 split-bf16, 1 float32; the neighbour order is the float64 one either way"},
    {"dual_stride", KZ_OPT_INT, KZ_O(dual_stride), 0, 64, 1, KZ_OPT_PUBLIC, {}, 0,
     "kz_knn_dual: every n-th tile of a is in the threshold sample; 1 = chosen from the shapes, 0 = always two ordinary searches"},
    {"dual_max_gb", KZ_OPT_F64, KZ_O(dual_max_gb), 0, 1e6, 0, KZ_OPT_PUBLIC, {}, 0,
     "kz_knn_dual: transient footprint it may claim, GiB (0 = 32); beyond it, or beyond what the device has free, it searches twice"},
    {"eps_scale", KZ_OPT_F64, KZ_O(eps_scale), 1e-300, 1e300, 1.0, KZ_OPT_PUBLIC, {}, 0,
     "multiplies the certification bound (test knob: a huge value sends every row to the exact float64 kernels)"},
    // ---- internal: scheduling / occupancy ---------------------------------------------------------------------------------------
    {"force_splits", KZ_OPT_INT, KZ_O(force_splits), 0, 64, 0, 0, {}, 0, "fixed index split count (0 = automatic)"},
    {"chunk_rows", KZ_OPT_INT, KZ_O(chunk_rows), 0, 1e9, 0, 0, {}, 0, "query rows per chunk (0 = by list length)"},
    {"h_q64", KZ_OPT_INT, KZ_O(h_q64), 0, 2, 2, 0, {}, 0, "64-queries-per-wave build: 2 where it pays, 1 wherever built, 0 never"},
    // ---- internal: list routes ---------------------------------------------------------------------------------------------------
    {"short_ord", KZ_OPT_BOOL, KZ_O(short_ord), 0, 1, 1, 0, {}, 0, "ordinary search: k / 5 lists of 16 on a row-dealt image (13 .. 320 neighbours)"},
    {"short_ord_min_tiles", KZ_OPT_INT, KZ_O(short_ord_min_tiles), 1, 1e9, 48, 0, {}, 0, "... when an index range has at least this many tiles"},
    {"esc_bf", KZ_OPT_BOOL, KZ_O(esc_bf), 0, 1, 1, 0, {}, 0, "split-bf16 operands before the float32 ones for rows the fp16 tier cannot certify"},
    {"tier_probe", KZ_OPT_INT, KZ_O(tier_probe), 0, 65536, 1024, 0, {}, 0, "rows of the strided sample a large search sends through the fp16 pass first (0 = off)"},
    {"probe_min_pairs", KZ_OPT_F64, KZ_O(probe_min_pairs), 0, 1e300, 5e10, 0, {}, 0, "searches of fewer distance pairs take neither the tier probe nor a floor"},
    {"esc_ladder", KZ_OPT_BOOL, KZ_O(esc_ladder), 0, 1, 1, 0, {}, 0, "a pass without a probe that leaves more than half of its rows uncertified tries the wide route on a sample of them before the split-bf16 tier"},
    {"exact_rows", KZ_OPT_INT, KZ_O(exact_rows), 0, 3, 3, 0, {}, 0, "exact float64 distance kernels: 0 one pair per wave, 1 + many pairs per wave step (float32 rows of d <= 512), 2 + one pair per LANE for batches of >= 32 rows (kz_exact_lanes.h), 3 (default) + the RANGE re-search for >= 128 rows (kz_range.h: only the index rows within the rounding bound of a row's k-th candidate); all bit-identical"},
    {"abl", KZ_OPT_INT, KZ_O(abl), 0, 31, 0, 0, {}, 0, "diagnostics, bit mask: 1 = an ordinary one-range fp16 sweep runs twice, the second (timed) one from the first one's final thresholds; 2 = with a -DKZ_ABL_STAMP build and KZ_STAMP_FILE set, clock stamps of every workgroup of an ordinary fp16 sweep; 4 = the range re-search's log holds 4096 groups (overflow path); 8 = the range re-search treats every row as one without k candidates (hand-back path); 16 = no grouped ranges (one range per row)"},
    {"spec_rows", KZ_OPT_INT, KZ_O(spec_rows), 0, 64, 64, 0, {}, 0, "exact kernels launched speculatively behind every finalize for at most this many uncertified rows (0 = off)"},
    {"wide_lists", KZ_OPT_INT, KZ_O(wide_lists), 2, 32, 32, KZ_OPT_SET, {0}, -1, "fp16 tier's wide route: lists of 16 per query (0 = off)"},
    {"wide_sel", KZ_OPT_INT, KZ_O(wide_sel), 16, 512, 256, 0, {}, 0, "... entries of those lists the finalize kernel selects"},
    {"floor_margin", KZ_OPT_F64, KZ_O(floor_margin), 0, 1e6, 1.3, 0, {}, 0, "seeded lists: the largest shortfall of the probe below the model, times this (0: the model itself -- half of the rows are searched again; test knob)"},
    {"dual_rank", KZ_OPT_INT, KZ_O(dual_rank), -1, 128, 0, 0, {}, 0, "rank of the sample key that becomes a row's event threshold (0 automatic, -1 = k + 1; 1: many rows short of events, test knob)"},
    {"list_floor", KZ_OPT_INT, KZ_O(list_floor), 0, 1, 1, 0, {}, 0, "seeded candidate lists (population floor from a probe)"},
    {"fin_fast_div", KZ_OPT_INT, KZ_O(fin_fast_div), 0, 1, 1, 0, {}, 0, "cosine re-rank through one reciprocal per candidate row (bit-identical)"},
    // ---- internal: shared sweep ----------------------------------------------------------------------------------------------------
    {"dual_force", KZ_OPT_BOOL, KZ_O(dual_force), 0, 1, 0, 0, {}, 0, "run the shared sweep also where its cost model says it does not pay (tests)"},
    {"dual_overlap", KZ_OPT_BOOL, KZ_O(dual_overlap), 0, 1, 1, 0, {}, 0, "reverse direction's chain on the second stream"},
    {"dual_nested", KZ_OPT_BOOL, KZ_O(dual_nested), 0, 1, 1, 0, {}, 0, "sampled rows are swept by the sample sweep only (itself a shared sweep)"},
    {"dual_rev_long", KZ_OPT_BOOL, KZ_O(dual_rev_long), 0, 1, 1, 0, {}, 0, "reverse lists of twice the list length"},
    {"dual_sample_short", KZ_OPT_BOOL, KZ_O(dual_sample_short), 0, 1, 1, 0, {}, 0, "sample sweep keeps lists of 16 (32) over several ranges"},
    {"dual_short_main", KZ_OPT_BOOL, KZ_O(dual_short_main), 0, 1, 1, 0, {}, 0, "main sweep keeps k / dual_short_div lists of 16 (13 .. 110 neighbours)"},
    {"dual_short_extra", KZ_OPT_INT, KZ_O(dual_short_extra), 1, 200, 48, 0, {}, 0, "entries selected beyond k on that route"},
    {"dual_short_min_tiles", KZ_OPT_INT, KZ_O(dual_short_min_tiles), 1, 1e9, 128, 0, {}, 0, "... taken when an index range has at least this many tiles"},
};
#undef KZ_O
static const int KZ_N_OPTIONS = (int)(sizeof(KZ_OPTIONS) / sizeof(KZ_OPTIONS[0]));

static inline void kz_option_store(kz_ctx* c, const KzOption& o, double v) {
    char* p = reinterpret_cast<char*>(c) + o.offset;
    if (o.kind == KZ_OPT_F64)
        *reinterpret_cast<double*>(p) = v;
    else
        *reinterpret_cast<int*>(p) = o.kind == KZ_OPT_BOOL ? (v != 0 ? 1 : 0) : (int)v;
}
static inline void kz_options_defaults(kz_ctx* c) {
    for (int i = 0; i < KZ_N_OPTIONS; ++i) kz_option_store(c, KZ_OPTIONS[i], KZ_OPTIONS[i].dflt);
}
