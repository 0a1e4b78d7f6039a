"""DESIGN.md section 5's residual risk, measured instead of "not observed": index rows built 1/64 .. 16 ulps apart in exact squared
distance (ulps of |q|^2 + |y|^2, the scale at which the reference's dgemm expansion -- sklearn EuclideanArgKmin behind
kiez/neighbors/exact/sklearn_nearest_neighbors.py:96-101 -- and the device's float64 re-rank both round), the reference's own
answer on them as the golden (tools/gen_near_ties.py).

What must hold: the device finds the same two rows for every query; from 2 ulps on it orders every pair as the reference and as
exact arithmetic do (between 1 and 2 ulps a float64 expansion can still be wrong: the numpy oracle is, once in 100 pairs).  Below
1 ulp the reference itself orders only 50-84 % of the pairs as exact arithmetic does (its order there
is a property of the BLAS summation order of the machine that made the fixture); the device's float64 values come from another
summation order, so it may differ there -- the test records how often and bounds it from below by chance level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_near_tie_pairs_against_the_reference_order():
    from tests.near_ties import run_probe
    r = run_probe()
    print(r)
    assert r["pairs"] == 1024 and r["same_two_rows_for_every_query"]
    by = {tuple(b["gap_ulps"]): b for b in r["buckets"]}
    b = by[(2.0, None)]
    assert b["pairs"] > 250
    assert b["device_orders_as_reference"] == 1.0 and b["device_orders_as_exact_arithmetic"] == 1.0, b
    b = by[(1.0, 2.0)]
    assert b["device_orders_as_reference"] >= 0.95 and b["device_orders_as_exact_arithmetic"] >= 0.95, b
    # under one ulp both implementations are right more often than not and never systematically opposed
    b = by[(0.25, 1.0)]
    assert b["device_orders_as_exact_arithmetic"] > 0.6 and b["reference_orders_as_exact_arithmetic"] > 0.6 and b["device_orders_as_reference"] > 0.5, b
    for key in ((0.0, 1 / 16), (1 / 16, 0.25)):
        assert by[key]["device_orders_as_reference"] > 0.3, by[key]
