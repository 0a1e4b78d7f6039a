"""The shared-reciprocal division of the cosine re-rank (`kz_div_shared`, kiez_amd/csrc/kz_common.h) against the plain float64
division, bit for bit: the finalize kernel replaces the four divisions per lane and candidate row by one reciprocal per row and
five multiply-adds per element (option "fin_fast_div"), and the re-rank's values must stay what `kz_pair_values` and the exact
float64 kernels compute with the plain division (sklearn's normalize() divides element by element)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mismatches(count, seed, mode):
    from kiez_amd import _native as N
    ctx = N.Context.get()
    bad = C.c_int64(-1)
    rc = ctx.lib.kz_selftest_div(ctx.handle, count, seed, mode, C.byref(bad))
    assert rc == 0, ctx.lib.kz_last_error()
    return bad.value


@pytest.mark.parametrize("mode", [0, 1])
def test_shared_reciprocal_division_is_the_ieee_quotient(mode):
    # 2^30 pairs per mode: float32-valued numerators of either sign over 60 binades, divisors = |numerator| x [1, 2^12)
    # with random 52-bit significands (mode 1: all-ones and single-bit significands mixed in on both sides)
    for seed in (1, 0x9e3779b97f4a7c15):
        assert _mismatches(1 << 29, seed, mode) == 0


def test_cosine_results_do_not_depend_on_the_division():
    from kiez_amd import _native as N
    rng = np.random.RandomState(5)
    q = rng.rand(3000, 72).astype(np.float32)
    y = rng.rand(5000, 72).astype(np.float32)
    y[17] = 0.0      # a zero row: its norm is replaced by 1 (normalize()), 0 / 1 on either path
    ctx = N.Context.get()
    out = []
    for fast in (0, 1):
        ctx.set_option("fin_fast_div", fast)
        try:
            qm, ym = N.DeviceMatrix(ctx, q, "cosine"), N.DeviceMatrix(ctx, y, "cosine")
            d, i, _ = N.knn(ctx, qm, ym, 50)
            out.append((d.numpy(), i.numpy()))
        finally:
            ctx.set_option("fin_fast_div", 0)
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0].view(np.int64), out[1][0].view(np.int64))
