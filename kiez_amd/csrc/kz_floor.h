// Host-only part of the population floor (kz_knn.hip "POPULATION FLOOR"): the model fitted to the probe.  No HIP dependency
// (tests/host/floor_sanitize.cpp builds it with g++ under AddressSanitizer + UBSan).
#pragma once

// pairs[2 i] = |q_c|^2 of probe row i, pairs[2 i + 1] = the exact key of its k-th neighbour.  Least squares key ~ alpha + beta |q_c|^2;
// the floor of a row is  alpha + beta |q_c|^2 - margin  with margin = (largest amount by which a probe row's key falls short of the
// model) x margin_scale.  By construction no probe row lies below its floor when margin_scale >= 1, and -- the rows being
// exchangeable -- another row does with probability <= 1 / (n_probe + 1).  Returns false (no floor) when the probe's values are not
// finite or there are no probe rows; a probe whose |q_c|^2 are all equal gets beta = 0.
static inline bool kz_floor_fit(const double* pairs, int n_probe, double margin_scale, double* model) {
    model[0] = model[1] = model[2] = 0.0;
    if (n_probe <= 0) return false;
    double sx = 0, sy = 0;
    for (int i = 0; i < n_probe; ++i) {
        sx += pairs[2 * i];
        sy += pairs[2 * i + 1];
    }
    const double mx = sx / n_probe, my = sy / n_probe;
    double sxx = 0, sxy = 0;
    for (int i = 0; i < n_probe; ++i) {
        sxx += (pairs[2 * i] - mx) * (pairs[2 * i] - mx);
        sxy += (pairs[2 * i] - mx) * (pairs[2 * i + 1] - my);
    }
    const double beta = sxx > 0 ? sxy / sxx : 0.0, alpha = my - beta * mx;
    double short_max = 0;
    for (int i = 0; i < n_probe; ++i) {
        const double r = alpha + beta * pairs[2 * i] - pairs[2 * i + 1];   // the model above the row's k-th key by r
        if (r > short_max) short_max = r;
    }
    model[0] = alpha;
    model[1] = beta;
    model[2] = short_max * margin_scale;
    return (alpha - alpha == 0.0) && (beta - beta == 0.0) && (model[2] - model[2] == 0.0);
}
