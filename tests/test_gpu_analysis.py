"""kiez_amd.analysis.hubness_score against the reference's own golden fixtures (tests/analysis/test_estimation.py:
tests/nn_ind.npy + expected_k{2,5,10,50}_hub_scores.pkl, re-encoded by tools/gen_golden.py)."""
import json
import warnings

import numpy as np
import pytest

from tests.golden_util import GOLDEN

pytestmark = pytest.mark.gpu

# the reference's literal test vector (tests/analysis/test_estimation.py:18-25)
K_OCC = np.array([3, 0, 5, 3, 0, 5, 4, 1, 0, 1, 1, 0, 0, 2, 0, 1, 0, 2, 2, 1, 0, 2, 0, 5, 2, 1, 0, 1, 0, 0, 4, 2, 3, 6,
                  1, 0, 3, 0, 0, 0, 2, 2, 3, 4, 3, 3, 2, 1, 0, 0, 1, 5, 2, 3, 0, 10, 0, 1, 0, 3, 1, 3, 5, 1, 1, 2, 6,
                  1, 3, 3, 3, 2, 2, 2, 0, 5, 2, 1, 1, 4, 0, 2, 2, 8, 1, 0, 7, 1, 2, 0, 0, 2, 0, 0, 3, 3, 3, 2, 9, 1])


@pytest.fixture(scope="module")
def neighbors():
    return np.load(GOLDEN / "ref_nn_ind.npy")


@pytest.mark.parametrize("k", [2, 5, 10, 50])
def test_all_measures_match_reference(k, neighbors):
    from kiez_amd.analysis import hubness_score
    exp = json.loads((GOLDEN / "ref_hub_scores.json").read_text())[str(k)]
    arrs = np.load(GOLDEN / "ref_hub_scores_arrays.npz")
    m = hubness_score(neighbors, 1000, k=k, return_value="all", store_k_occurrence=True)
    for key, v in exp.items():
        assert m[key] == pytest.approx(v), key
    for key in ("antihubs", "hubs", "k_occurrence"):
        np.testing.assert_array_equal(m[key], arrs[f"k{k}__{key}"])


def test_small_literal_case():
    from kiez_amd.analysis import hubness_score
    neigh = np.array([[0, 2], [1, 0], [2, 0], [3, 1], [4, 0]])
    assert hubness_score(neigh, 5)["k_skewness"] == pytest.approx(0.9128709291752769, abs=1e-10)


@pytest.mark.parametrize("k", [1, 5, 10])
def test_self_consistent(k, neighbors):
    from kiez_amd.analysis import hubness_score
    s = hubness_score(neighbors, 1000, k=k, store_k_occurrence=True)
    occ = s["k_occurrence"]
    occ_true = np.bincount(neighbors[:, :k].ravel(), minlength=1000)
    np.testing.assert_array_equal(occ, occ_true)
    x0 = occ - occ.mean()
    assert s["k_skewness"] == pytest.approx((x0**3).mean() / (x0**2).mean() ** 1.5, rel=1e-12)
    assert "gini" not in s


def test_gini_and_atkinson_on_literal_vector():
    """Build a neighbour matrix whose k-occurrence is the reference's literal K_OCC and compare with numpy formulas."""
    from kiez_amd.analysis import hubness_score
    ids = np.repeat(np.arange(len(K_OCC)), K_OCC)            # each id as often as its k-occurrence
    pad = (-len(ids)) % 4
    neigh = np.concatenate([ids, -np.ones(pad, dtype=np.int64)]).reshape(-1, 4)   # negatives are dropped
    n_rows = neigh.shape[0]
    m = hubness_score(neigh, 100, return_value="all", store_k_occurrence=True)
    kocc = np.bincount(ids, minlength=n_rows)
    np.testing.assert_array_equal(m["k_occurrence"][:len(K_OCC)], K_OCC)
    gini = np.abs(kocc[None, :] - kocc[:, None]).sum() / (2 * kocc.size * kocc.sum())
    assert m["gini"] == pytest.approx(gini, rel=1e-12)
    atk = 1.0 - 1.0 / kocc.mean() * np.mean(kocc ** 0.5) ** 2
    assert m["atkinson"] == pytest.approx(atk, rel=1e-12)
    assert m["robinhood"] == pytest.approx(0.5 * np.abs(kocc - kocc.mean()).sum() / kocc.sum(), rel=1e-12)


def test_negative_indices_k_too_large_and_wrong_neighbors():
    from kiez_amd.analysis import hubness_score
    neigh = np.array([[1, 2, 3], [-1, 4, 5]])
    assert hubness_score(neigh, 5) is not None
    with pytest.warns(UserWarning, match="k > nn_ind.shape"):
        assert hubness_score(neigh, 5, k=10) is not None
    with pytest.raises(ValueError, match="no negative"):
        hubness_score(np.array([[np.inf], [0]]), 1)


def test_device_input_from_the_hot_path():
    from kiez_amd import Kiez
    from kiez_amd.analysis import hubness_score
    rng = np.random.RandomState(0)
    s, t = rng.rand(400, 20), rng.rand(300, 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, hubness="CSLS").fit(s, t)
        d_dev, i_dev = kz.kneighbors_device(5)
        _, i_host = kz.kneighbors(5)
    a = hubness_score(i_dev, 300, store_k_occurrence=True)
    b = hubness_score(i_host, 300, store_k_occurrence=True)
    np.testing.assert_array_equal(a["k_occurrence"], b["k_occurrence"])
    assert a["robinhood"] == b["robinhood"]


# ---- kiez.evaluate.hits: the reference's literal cases (tests/evaluate/test_eval_metrics.py:6-45) -------------------
@pytest.mark.parametrize(("nn_ind", "gold", "k", "expected"), [
    ([[1, 2, 3], [2, 3, 4], [3, 4, 5], [4, 5, 6]], {0: 2, 1: 4, 2: 3, 3: 4}, [1, 2, 3], {1: 0.5, 2: 0.75, 3: 1.0}),
    ([[1, 2, 3], [2, 3, 4], [3, 4, 5], [4, 5, 6]], {0: 5, 1: 6, 2: 7, 3: 8}, None, {1: 0.0, 5: 0.0, 10: 0.0}),
    ({0: [1, 2, 3], 1: [2, 3, 4], 2: [3, 4, 5], 3: [4, 5, 6]}, {0: 2, 1: 4, 2: 3, 3: 4}, [1, 2, 3], {1: 0.5, 2: 0.75, 3: 1.0}),
    ({0: [1, 2, 3], 1: [2, 3, 4], 2: [3, 4, 5], 3: [4, 5, 6]}, {0: 5, 1: 6, 2: 7, 3: 8}, None, {1: 0.0, 5: 0.0, 10: 0.0}),
    ({"0": ["1", "2", "3"], "1": ["2", "3", "4"], "2": ["3", "4", "5"], "3": ["4", "5", "6"]},
     {"0": "2", "1": "4", "2": "3", "3": "4"}, [1, 2, 3], {1: 0.5, 2: 0.75, 3: 1.0}),
])
def test_hits(nn_ind, gold, k, expected):
    from kiez_amd.evaluate import hits
    assert hits(nn_ind, gold, k) == expected


def test_hits_docstring_example_and_device_input():
    from kiez_amd import _native as N
    from kiez_amd.evaluate import hits
    nn = np.array([[1, 2, 3], [2, 3, 4], [3, 4, 5], [4, 5, 6]])
    gold = {0: 2, 1: 4, 2: 3, 3: 4}
    assert hits(nn, gold) == {1: 0.5, 5: 1.0, 10: 1.0}
    assert hits(N.Context.get().to_device(nn.astype(np.int64)), gold) == {1: 0.5, 5: 1.0, 10: 1.0}
    assert hits(nn, {0: 2, 7: 1}) == {1: 0.0, 5: 0.5, 10: 0.5}     # gold rows outside the matrix count in the denominator


def test_histogram_is_sized_by_the_first_k_columns_only():
    """ADVICE (round 1): with k < nn_ind.shape[1] and target ids >= n_train in the DROPPED columns, the k-occurrence vector
    (and every measure derived from its length) must follow np.bincount(nn_ind[:, :k].ravel(), minlength=n_train)
    (kiez/analysis/estimation.py:276-292)."""
    import numpy as np
    from kiez_amd.analysis import hubness_score
    rng = np.random.RandomState(3)
    n_train, cols, k = 40, 8, 3
    nn_ind = rng.randint(0, 35, size=(n_train, cols)).astype(np.int64)
    nn_ind[:, k:] += 500                       # large ids only in the columns k.. (n_target > n_source case)
    want = np.bincount(nn_ind[:, :k].ravel(), minlength=n_train)
    got = hubness_score(nn_ind, 600, k=k, return_value="k_occurrence")
    assert got.shape == want.shape == (n_train,)
    np.testing.assert_array_equal(got, want)
    allm = hubness_score(nn_ind, 600, k=k, return_value="all_but_gini")
    np.testing.assert_array_equal(allm["antihubs"], np.argwhere(want == 0).ravel())
    assert abs(allm["antihub_occurrence"] - (want == 0).mean()) < 1e-12
    # ... and ids >= n_train INSIDE the first k columns do extend it
    nn_ind[0, 0] = 77
    want = np.bincount(nn_ind[:, :k].ravel(), minlength=n_train)
    got = hubness_score(nn_ind, 600, k=k, return_value="k_occurrence")
    np.testing.assert_array_equal(got, want)
