"""Helpers to iterate the committed golden fixtures (tests/golden/*.npz, made by tools/gen_golden.py)."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

HUB = {
    "none": (None, {}),
    "csls": ("CSLS", {}),
    "mp_normal": ("MutualProximity", {"method": "normal"}),
    "mp_empiric": ("MutualProximity", {"method": "empiric"}),
    "ls": ("LocalScaling", {"method": "standard"}),
    "nicdm": ("LocalScaling", {"method": "nicdm"}),
    "dsl": ("DisSimLocal", {}),
}

PIPELINE_CASES = [
    "conftest_two_source", "conftest_single_source", "c0_two_source", "c0_single_source",
    "f32_euclidean", "f32_sqeuclidean", "f32_gauss_single", "cosine_k50", "cosine_single",
]


def load_case(name):
    z = np.load(GOLDEN / f"{name}.npz")
    g = {k: z[k] for k in z.files}
    g["_name"] = name
    g["_K"] = int(g["K"])
    g["_metric"] = str(g["metric"])
    g["_p"] = int(g["p"])
    g["_ks"] = [None if k < 0 else int(k) for k in g["ks"]]
    g["_target"] = g.get("target")
    g["_tags"] = sorted({k.split("__")[0] for k in g if "__" in k and not k.endswith("__raises")})
    return g


def case_params():
    """(case, tag, k) triples for parametrisation."""
    out = []
    for c in PIPELINE_CASES:
        g = load_case(c)
        for tag in g["_tags"]:
            for k in g["_ks"]:
                out.append((c, tag, k))
    return out


def ktag(k):
    return "None" if k is None else str(k)


def tie_tolerant_index_equal(dist_ref, ind_ref, dist_got, ind_got, rtol=1e-9):
    """Indices equal, except that positions inside a run of (numerically) tied reference distances
    may hold the tied ids in any order."""
    if np.array_equal(ind_ref, ind_got):
        return True
    for r in np.flatnonzero((ind_ref != ind_got).any(axis=1)):
        dr = dist_ref[r]
        bad = np.flatnonzero(ind_ref[r] != ind_got[r])
        for p in bad:
            tied = np.isclose(dr, dr[p], rtol=rtol, atol=1e-12)
            if set(ind_ref[r][tied]) != set(ind_got[r][tied]):
                return False
    return True
