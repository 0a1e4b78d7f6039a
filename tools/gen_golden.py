#!/usr/bin/env python3
"""Generate golden vectors for the hot path by RUNNING THE REAL REFERENCE (build container only).

    python tools/gen_golden.py            # writes tests/golden/*.npz

The reference (dobraczka/kiez @ /root/reference) is imported through tools/ref_loader.py; its source
is never copied.  Every fixture holds the inputs, the configuration and the reference's outputs
(including the intermediates `dist_t2s/ind_t2s/dist_s2t/ind_s2t` and the unsorted `transform`
output), so the oracle and the HIP path can be checked stage by stage where the reference is absent.

Library versions are recorded in tests/golden/MANIFEST.json.  numpy's SIMD arg-select makes the
k == 1 tie winner CPU-feature dependent, so the generator re-executes itself with
AVX2/AVX512 dispatch disabled: the scalar path picks the first minimum.
"""
import json
import os
import subprocess
import sys
import warnings
from pathlib import Path

_DISABLE = "AVX2 FMA3 AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"
if os.environ.get("NPY_DISABLE_CPU_FEATURES") != _DISABLE:
    env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=_DISABLE)
    sys.exit(subprocess.call([sys.executable, *sys.argv], env=env))

import numpy as np  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))
from ref_loader import load_reference  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"
R = load_reference()
warnings.simplefilter("ignore")

HUBNESS = [
    ("none", None, {}),
    ("csls", "CSLS", {}),
    ("mp_normal", "MutualProximity", {"method": "normal"}),
    ("mp_empiric", "MutualProximity", {"method": "empiric"}),
    ("ls", "LocalScaling", {"method": "standard"}),
    ("nicdm", "LocalScaling", {"method": "nicdm"}),
    ("dsl", "DisSimLocal", {}),
]
CLS = {None: R.NoHubnessReduction, "CSLS": R.CSLS, "MutualProximity": R.MutualProximity,
       "LocalScaling": R.LocalScaling, "DisSimLocal": R.DisSimLocal}


ONLY = [a for a in os.environ.get("KZ_GOLDEN_ONLY", "").split(",") if a]   # (regenerate only the cases whose name starts so)


def run_case(name, source, target, K, ks, metric, p=2, hubs=None, sk_algorithm="auto"):
    """Run the reference for every hubness setting; save one npz."""
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    out = {"source": source, "K": np.int64(K), "ks": np.array([(-1 if k is None else k) for k in ks]),
           "metric": np.array(metric), "p": np.int64(p) if p == int(p) else np.float64(p)}
    if target is not None:
        out["target"] = target
    for tag, hname, kw in HUBNESS:
        if hubs is not None and tag not in hubs:
            continue
        algo_kw = dict(n_candidates=K, metric=metric, p=p, algorithm=sk_algorithm)
        try:
            hub = CLS[hname](nn_algo=R.SklearnNN(**algo_kw), **kw)
        except ValueError as e:  # e.g. DSL with cosine
            out[f"{tag}__raises"] = np.array(str(e))
            continue
        hub.fit(source, target)
        if hname is not None:
            tgt = source if target is None else target
            d_t2s, i_t2s = hub.nn_algo.kneighbors(k=K, query=tgt, s_to_t=False, return_distance=True)
            d_s2t, i_s2t = hub.nn_algo.kneighbors(query=None, k=K, return_distance=True)
            tr, _ = hub.transform(d_s2t.copy(), i_s2t.copy(), hub.nn_algo.source_.copy())
            out[f"{tag}__dist_t2s"], out[f"{tag}__ind_t2s"] = d_t2s, i_t2s
            out[f"{tag}__dist_s2t"], out[f"{tag}__ind_s2t"] = d_s2t, i_s2t
            out[f"{tag}__transformed"] = tr
        for k in ks:
            d, i = hub.kneighbors(k)
            ktag = "None" if k is None else str(k)
            out[f"{tag}__k{ktag}__dist"], out[f"{tag}__k{ktag}__ind"] = d, i
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")), "bytes (raw)")


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    # 1. the reference's shared test fixture (tests/conftest.py:5-11), n_candidates=5 (tests/test_kiez.py:66-79)
    rng = np.random.RandomState(42)
    s, t = rng.rand(20, 5), rng.rand(50, 5)
    run_case("conftest_two_source", s, t, 5, [None, 1, 3], "minkowski")
    run_case("conftest_single_source", s, None, 5, [None, 1, 3], "minkowski")
    # 2. BASELINE config 0: Kiez docstring data (kiez/kiez.py:50-56), float64, default K=10, k=5
    rng = np.random.RandomState(0)
    s, t = rng.rand(100, 50), rng.rand(100, 50)
    run_case("c0_two_source", s, t, 10, [5, None], "minkowski")
    run_case("c0_single_source", s, None, 10, [5], "minkowski")
    # 3. float32 inputs, ragged sizes (not multiples of any tile), euclidean brute force
    rng = np.random.RandomState(7)
    s, t = rng.rand(300, 24).astype(np.float32), rng.rand(257, 24).astype(np.float32)
    run_case("f32_euclidean", s, t, 10, [10, 4, 1], "euclidean", sk_algorithm="brute")
    run_case("f32_sqeuclidean", s, t, 10, [10, 3], "sqeuclidean", hubs={"none", "csls", "dsl"},
             sk_algorithm="brute")
    # 4. gaussian float32 (contrast to uniform), single source
    rng = np.random.RandomState(11)
    s = rng.randn(200, 32).astype(np.float32)
    run_case("f32_gauss_single", s, None, 10, [10, 2], "euclidean", hubs={"none", "csls", "ls", "dsl"},
             sk_algorithm="brute")
    # 5. cosine, K=50 (BASELINE config 3 at fixture scale).  float32 data handed to the reference as its exact
    #    float64 cast (SURVEY.md §8c caution 2) so the order is not decided by sgemm rounding.
    rng = np.random.RandomState(3)
    s32, t32 = rng.rand(200, 24).astype(np.float32), rng.rand(150, 24).astype(np.float32)
    run_case("cosine_k50", s32.astype(np.float64), t32.astype(np.float64), 50, [50, 7, 1], "cosine",
             sk_algorithm="brute")
    run_case("cosine_single", s32.astype(np.float64), None, 12, [12, 5], "cosine",
             hubs={"none", "csls", "mp_empiric"}, sk_algorithm="brute")
    # 5b. the rest of the Minkowski family (SklearnNN.valid_metrics, sklearn_nearest_neighbors.py:49): scikit-learn's generic
    #     DistanceMetric path -- manhattan (and its aliases), chebyshev, minkowski with p = 3 / 1.5; DisSimLocal raises
    rng = np.random.RandomState(21)
    s, t = rng.rand(120, 20), rng.rand(90, 20)
    run_case("f64_manhattan", s, t, 10, [10, 3], "manhattan", sk_algorithm="brute")
    s32, t32 = rng.randn(150, 24).astype(np.float32), rng.randn(131, 24).astype(np.float32)
    run_case("f32_chebyshev_single", s32, None, 8, [8, 2], "chebyshev", hubs={"none", "csls", "ls"}, sk_algorithm="brute")
    run_case("f32_minkowski_p3", s32, t32, 10, [10, 4], "minkowski", p=3, hubs={"none", "csls", "mp_empiric", "nicdm", "dsl"},
             sk_algorithm="brute")
    run_case("f64_minkowski_p1_5_single", s, None, 6, [6, 1], "minkowski", p=1.5, hubs={"none", "mp_normal"},
             sk_algorithm="brute")
    run_case("f32_cityblock", s32, t32, 5, [5], "cityblock", hubs={"none", "csls"})
    # 5c. integer exponents where a product chain and pow() part ways (round-5 advisor finding): float64 inputs with p = 3 / 4 (the
    #     device calls pow() there, as scikit-learn does) and float32 inputs with p = 4 (the one-rounding product (a a)(a a))
    run_case("f64_minkowski_p3", s, t, 8, [8, 2], "minkowski", p=3, hubs={"none", "csls"}, sk_algorithm="brute")
    run_case("f64_minkowski_p4", s, t, 8, [8], "minkowski", p=4, hubs={"none", "ls"}, sk_algorithm="brute")
    run_case("f32_minkowski_p4", s32, t32, 10, [10, 3], "minkowski", p=4, hubs={"none", "csls", "mp_normal"}, sk_algorithm="brute")
    if ONLY:
        return
    # 6. HubnessReduction._sort (kiez/hubness_reduction/base.py:72-87): the reference's own test input
    #    (tests/hubness_reduction/test_hubness_base.py:17-21) plus tie-heavy rows
    rng = np.random.default_rng(42)
    dist = rng.random((100, 10))
    ind = np.vstack([rng.permutation(10) for _ in range(100)]) if False else rng.integers(0, 1000, (100, 10))
    out = {"dist0": dist, "ind0": ind}
    trng = np.random.RandomState(5)
    tie_d = trng.randint(0, 4, size=(400, 12)).astype(np.float64) / 4.0
    tie_i = trng.randint(0, 10000, size=(400, 12)).astype(np.int64)
    tie_d32 = tie_d.astype(np.float32)
    out.update(tie_d=tie_d, tie_i=tie_i)
    for k in (1, 2, 3, 5, 10):
        d, i = R.HubnessReduction._sort(dist, ind, k)
        out[f"sorted0_k{k}_dist"], out[f"sorted0_k{k}_ind"] = d, i
        d, i = R.HubnessReduction._sort(tie_d, tie_i, k)
        out[f"tie_k{k}_dist"], out[f"tie_k{k}_ind"] = d, i
        d, i = R.HubnessReduction._sort(tie_d32, tie_i, k)
        out[f"tie32_k{k}_dist"], out[f"tie32_k{k}_ind"] = d, i
    np.savez_compressed(OUT / "sort.npz", **out)
    print("wrote sort")
    # 7. kiez.analysis.hubness_score: the reference's own test fixtures (tests/analysis/test_estimation.py:14,56-69):
    #    tests/nn_ind.npy is copied as data, the pickled expected scores are re-encoded as json / npz
    import pickle
    import shutil
    ref_tests = Path("/root/reference/tests")
    shutil.copy(ref_tests / "nn_ind.npy", OUT / "ref_nn_ind.npy")
    exp, arrs = {}, {}
    for k in (2, 5, 10, 50):
        with open(ref_tests / f"expected_k{k}_hub_scores.pkl", "rb") as fh:
            d = pickle.load(fh)
        exp[str(k)] = {a: float(b) for a, b in d.items() if not isinstance(b, np.ndarray)}
        arrs.update({f"k{k}__{a}": b for a, b in d.items() if isinstance(b, np.ndarray)})
    (OUT / "ref_hub_scores.json").write_text(json.dumps(exp, indent=1) + "\n")
    np.savez_compressed(OUT / "ref_hub_scores_arrays.npz", **arrs)
    print("wrote hubness-score fixtures")
    import scipy
    import sklearn
    manifest = {
        "generator": "tools/gen_golden.py",
        "reference": "dobraczka/kiez v0.5.0 (/root/reference), hot-path modules loaded by file path",
        "python": sys.version.split()[0],
        "numpy": np.__version__, "scipy": scipy.__version__, "scikit-learn": sklearn.__version__,
        "NPY_DISABLE_CPU_FEATURES": _DISABLE,
    }
    (OUT / "MANIFEST.json").write_text(json.dumps(manifest, indent=2) + "\n")


if __name__ == "__main__":
    main()
