"""The route gates of the search, one test per gate: at a shape on the gate's boundary BOTH sides are run (the option that forces each
side), on uniform, gaussian and mixture data, and the side the library chooses by itself must be within 15 % of the better one
(best of four runs each; a gate fitted on uniform rows that sends another kind of data down the slow side is a cliff).  Every side
returns the same neighbours -- checked too.  Gates: the shared sweep (kz_knn_dual's cost model), its nested sample, the tier probe of
a large ordinary search, the wide route the probe's ladder takes on dense keys, the ladder after the fact.  Reference: the two
searches of a fit, kiez/hubness_reduction/base.py:33-50; the brute-force search, sklearn_nearest_neighbors.py:96-101."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1.15
RESET = (("dual_force", 0), ("dual_stride", 1), ("dual_nested", 1), ("tier_probe", 1024), ("probe_min_pairs", 5e10), ("wide_lists", 32),
         ("esc_ladder", 1))


@pytest.fixture()
def ctx():
    from kiez_amd import _native as N
    c = N.Context.get()
    yield c
    for name, value in RESET:
        c.set_option(name, value)


def _data(kind, n, d, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, d)).astype(np.float32)
    if kind == "normal":
        return rng.standard_normal((n, d)).astype(np.float32)
    if kind == "gmm":      # L2-normalised gaussian mixture, the same centres on both sides (bench.py "gmm")
        cc = np.random.default_rng(7).standard_normal((256, d))
        x = cc[rng.integers(0, 256, n)] + 0.35 * rng.standard_normal((n, d))
        return (x / np.sqrt((x * x).sum(axis=1, keepdims=True))).astype(np.float32)
    if kind == "tight":    # 40 tight clusters far from the centre, rows shuffled (bench.py "hard"): keys dense around the k-th neighbour
        centres = np.random.default_rng(5).standard_normal((40, d)) * 3
        return (centres[rng.integers(0, 40, n)] + 0.4 * rng.standard_normal((n, d))).astype(np.float32)
    raise ValueError(kind)


def _best(ctx, fn, reps=4):
    best, out = 1e9, None
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        out = fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, out


def _sides(ctx, fn, sides):
    """sides: {name: [(option, value), ...]}; returns {name: (ms, result)} with the options of RESET restored in between."""
    res = {}
    for name, opts in sides.items():
        for o, v in RESET:
            ctx.set_option(o, v)
        for o, v in opts:
            ctx.set_option(o, v)
        res[name] = _best(ctx, fn)
    for o, v in RESET:
        ctx.set_option(o, v)
    return res


def _same(r0, r1):
    for (d0, i0, _), (d1, i1, _) in zip(r0, r1):
        np.testing.assert_array_equal(i0.numpy(), i1.numpy())
        np.testing.assert_array_equal(d0.numpy(), d1.numpy())


@pytest.mark.parametrize("kind", ["uniform", "normal", "gmm"])
@pytest.mark.parametrize("n,d", [(60_000, 128), (100_000, 128), (40_000, 300)])
def test_shared_sweep_gate(ctx, kind, n, d):
    """One sweep for both directions or two ordinary searches: the cost model's choice against both forced sides."""
    from kiez_amd import _native as N
    a, b = _data(kind, n, d, 1), _data(kind, n + 1000, d, 2)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    r = _sides(ctx, lambda: N.knn_dual(ctx, am, bm, 10), {"chosen": [], "shared": [("dual_force", 1)], "twice": [("dual_stride", 0)]})
    _same(r["chosen"][1], r["twice"][1])
    _same(r["shared"][1], r["twice"][1])
    assert r["twice"][1][0][2]["dual"] == 0
    better = min(r["shared"][0], r["twice"][0])
    assert r["chosen"][0] <= TOL * better, {k: round(v[0], 3) for k, v in r.items()}


@pytest.mark.parametrize("kind", ["uniform", "normal", "gmm"])
def test_nested_sample_gate(ctx, kind):
    """The sampled rows swept by the sample sweep only (nested) or by both sweeps: taken from 2 model-ms of saving on -- both sides
    at a shape near that boundary."""
    from kiez_amd import _native as N
    a, b = _data(kind, 160_000, 128, 3), _data(kind, 120_000, 128, 4)
    am, bm = N.DeviceMatrix(ctx, a, "euclidean"), N.DeviceMatrix(ctx, b, "euclidean")
    r = _sides(ctx, lambda: N.knn_dual(ctx, am, bm, 10), {"chosen": [], "nested": [("dual_force", 1)], "classic": [("dual_nested", 0)]})
    _same(r["chosen"][1], r["classic"][1])
    _same(r["nested"][1], r["classic"][1])
    better = min(r["nested"][0], r["classic"][0])
    assert r["chosen"][0] <= TOL * better, {k: round(v[0], 3) for k, v in r.items()}


@pytest.mark.parametrize("kind", ["uniform", "normal", "gmm"])
def test_tier_probe_gate(ctx, kind):
    """A large ordinary search sends 1024 strided rows through the fp16 pass first (from 5e10 pairs on): on data that is fine the probe
    must cost next to nothing -- the search just above the gate against the same search with the probe off."""
    from kiez_amd import _native as N
    q, y = _data(kind, 231_000, 32, 5), _data(kind, 231_000, 32, 6)      # (5.3e10 pairs)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    r = _sides(ctx, lambda: (N.knn(ctx, qm, ym, 10),), {"chosen": [], "no probe": [("tier_probe", 0)]})
    _same(r["chosen"][1], r["no probe"][1])
    assert r["chosen"][1][0][2]["first_pass"] == 2                        # (fp16: the probe found nothing wrong)
    assert r["chosen"][0] <= TOL * r["no probe"][0], {k: round(v[0], 3) for k, v in r.items()}


def test_wide_route_gate(ctx):
    """Keys dense around the k-th neighbour (tight clusters): the probe's ladder takes 32 lists of 16 on the same fp16 operands
    instead of the split-bf16 operands -- the chosen route against both forced sides."""
    from kiez_amd import _native as N
    q, y = _data("tight", 120_000, 64, 7), _data("tight", 121_000, 64, 8)
    qm, ym = N.DeviceMatrix(ctx, q, "cosine"), N.DeviceMatrix(ctx, y, "cosine")
    ctx.set_option("probe_min_pairs", 1e9)
    r = _sides(ctx, lambda: (N.knn(ctx, qm, ym, 50),),
               {"chosen": [("probe_min_pairs", 1e9)], "bf16 from the start": [("probe_min_pairs", 1e9), ("wide_lists", 0)],
                "no probe": [("tier_probe", 0)]})
    _same(r["chosen"][1], r["no probe"][1])
    _same(r["bf16 from the start"][1], r["no probe"][1])
    assert r["chosen"][1][0][2]["wide_lists"] == 32
    better = min(v[0] for k, v in r.items() if k != "chosen")
    assert r["chosen"][0] <= TOL * better, {k: round(v[0], 3) for k, v in r.items()}


def test_ladder_after_the_fact_gate(ctx):
    """A search below the probe's size gates that finds out afterwards -- more than half of a pass uncertified -- tries the wide route
    on a sample of the failed rows before the split-bf16 tier: with the ladder against without."""
    from kiez_amd import _native as N
    q, y = _data("tight", 100_000, 128, 9), _data("tight", 101_000, 128, 10)
    qm, ym = N.DeviceMatrix(ctx, q, "euclidean"), N.DeviceMatrix(ctx, y, "euclidean")
    r = _sides(ctx, lambda: (N.knn(ctx, qm, ym, 10),), {"chosen": [], "no ladder": [("esc_ladder", 0)]})
    _same(r["chosen"][1], r["no ladder"][1])
    assert r["chosen"][0] <= TOL * r["no ladder"][0], {k: round(v[0], 3) for k, v in r.items()}
