"""cosine + float32 inputs against the reference run on the float32 inputs themselves (sgemm order), counted
(tests/cosine_f32.py; fixture generated from the imported reference by tools/gen_cosine_f32.py).  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cosine_float32_differs_from_the_reference_only_inside_near_tie_groups():
    from tests.cosine_f32 import run_probe
    out = run_probe()
    print(out)
    n = out["rows"]
    # the convention of this build: the reference's own answer on the exact float64 casts of the inputs, every row
    assert out["rows_identical_to_reference_on_float64_casts"] == n
    # against the reference on the float32 inputs: the rows that differ are the rows the REFERENCE ITSELF orders differently
    # between the two dtypes, and every one of them differs only where float32 cannot tell two candidates apart
    assert out["rows_differing_otherwise"] == 0
    assert out["rows_ordered_as_reference_on_float32"] + out["rows_differing_only_inside_near_tie_groups"] == n
    assert out["rows_differing_only_inside_near_tie_groups"] == out["rows_reference_itself_orders_differently_f32_vs_f64cast"]
    assert out["rows_ordered_as_reference_on_float32"] >= int(0.97 * n)


def test_classifier_sees_a_real_difference():
    from tests.cosine_f32 import classify
    ref = np.array([[0, 1, 2]])
    d = {0: 0.1, 1: 0.2, 2: 0.3, 3: 0.30000001, 4: 0.5}

    def exact(r, ids):
        return np.array([d[int(i)] for i in ids])
    assert classify(ref, np.array([[0, 1, 2]]), exact) == (1, 0, 0)
    assert classify(ref, np.array([[0, 1, 3]]), exact) == (0, 1, 0)     # 2 and 3 are 3e-8 relative apart: a near tie at the k-th place
    assert classify(ref, np.array([[0, 2, 1]]), exact) == (0, 0, 1)     # a swap of clearly different distances
    assert classify(ref, np.array([[0, 1, 4]]), exact) == (0, 0, 1)
