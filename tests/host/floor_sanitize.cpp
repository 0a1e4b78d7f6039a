// CPU-only check of the host part of the population floor (kiez_amd/csrc/kz_floor.h), built with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_host_sanitize.py.  Properties of the fitted model on synthetic probes:
//   * no probe row lies below its own floor (margin scale >= 1), at least one row touches it at scale 1;
//   * a probe that IS a line is recovered (alpha, beta; margin ~ 0);
//   * on fresh rows from the same population the share below the floor is ~ 1 / (n_probe + 1) at scale 1 (exchangeability:
//     checked with a generous factor) and shrinks with the scale;
//   * degenerate probes: all |q_c|^2 equal (beta = 0), one row, non-finite values (no floor), zero rows (no floor).
#include <cmath>
#include <cstdio>
#include <limits>
#include <random>
#include <vector>

#include "../../kiez_amd/csrc/kz_floor.h"

static int fails = 0;
#define CHECK(cond, ...)                                     \
    do {                                                     \
        if (!(cond)) {                                       \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                        \
            std::printf("\n");                               \
            ++fails;                                         \
        }                                                    \
    } while (0)

int main() {
    std::mt19937_64 rng(7);
    std::normal_distribution<double> gauss(0.0, 1.0);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    for (int trial = 0; trial < 200; ++trial) {
        const int n = 16 + (int)(uni(rng) * 4000);
        const double a = 10.0 * gauss(rng), b = gauss(rng), noise = 0.01 + uni(rng);
        const bool heavy = trial % 3 == 0;   // heavy-tailed shortfalls every third trial
        auto draw = [&](double& x, double& y) {
            x = 5.0 + 3.0 * uni(rng);
            double e = noise * gauss(rng);
            if (heavy && uni(rng) < 0.02) e -= 20.0 * noise * uni(rng);
            y = a + b * x + e;
        };
        std::vector<double> pairs((size_t)2 * n);   // exactly sized: ASan sees any out-of-range access
        for (int i = 0; i < n; ++i) draw(pairs[2 * i], pairs[2 * i + 1]);
        double m[3];
        CHECK(kz_floor_fit(pairs.data(), n, 1.0, m), "trial %d: finite probe rejected", trial);
        int below = 0, touching = 0;
        for (int i = 0; i < n; ++i) {
            const double fl = m[0] + m[1] * pairs[2 * i] - m[2];
            below += pairs[2 * i + 1] < fl - 1e-9 * (1.0 + std::fabs(fl)) ? 1 : 0;
            touching += std::fabs(pairs[2 * i + 1] - fl) <= 1e-9 * (1.0 + std::fabs(fl)) ? 1 : 0;
        }
        CHECK(below == 0, "trial %d: %d probe rows below their own floor", trial, below);
        CHECK(touching >= 1, "trial %d: no probe row touches the floor at scale 1", trial);
        // fresh rows of the same population
        const int fresh = 200000;
        double m13[3];
        kz_floor_fit(pairs.data(), n, 1.3, m13);
        int f1 = 0, f13 = 0;
        for (int i = 0; i < fresh; ++i) {
            double x, y;
            draw(x, y);
            f1 += y < m[0] + m[1] * x - m[2] ? 1 : 0;
            f13 += y < m13[0] + m13[1] * x - m13[2] ? 1 : 0;
        }
        CHECK((double)f1 / fresh <= 8.0 / (n + 1) + 1e-4, "trial %d (n %d): %.5f of fresh rows below the floor at scale 1", trial, n, (double)f1 / fresh);
        CHECK(f13 <= f1, "trial %d: a larger margin sent more rows below the floor (%d > %d)", trial, f13, f1);
    }
    {   // an exact line
        std::vector<double> p(2 * 100);
        for (int i = 0; i < 100; ++i) {
            p[2 * i] = 1.0 + 0.25 * i;
            p[2 * i + 1] = -3.0 + 0.5 * p[2 * i];
        }
        double m[3];
        CHECK(kz_floor_fit(p.data(), 100, 1.3, m), "line rejected");
        CHECK(std::fabs(m[0] + 3.0) < 1e-9 && std::fabs(m[1] - 0.5) < 1e-10 && m[2] < 1e-9, "line: alpha %g beta %g margin %g", m[0], m[1], m[2]);
    }
    {   // degenerate probes
        std::vector<double> p = {2.0, 1.0, 2.0, 3.0, 2.0, -1.0};
        double m[3];
        CHECK(kz_floor_fit(p.data(), 3, 1.0, m) && m[1] == 0.0 && std::fabs(m[0] - 1.0) < 1e-12 && std::fabs(m[2] - 2.0) < 1e-12, "equal x: %g %g %g", m[0], m[1], m[2]);
        CHECK(kz_floor_fit(p.data(), 1, 1.0, m) && m[2] == 0.0, "one row");
        CHECK(!kz_floor_fit(p.data(), 0, 1.0, m), "zero rows accepted");
        p[3] = std::numeric_limits<double>::quiet_NaN();
        CHECK(!kz_floor_fit(p.data(), 3, 1.0, m), "NaN accepted");
        p[3] = std::numeric_limits<double>::infinity();
        CHECK(!kz_floor_fit(p.data(), 3, 1.0, m), "inf accepted");
    }
    std::printf("%d failures\n", fails);
    return fails ? 1 : 0;
}
