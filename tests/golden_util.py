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
    "f64_manhattan", "f32_chebyshev_single", "f32_minkowski_p3", "f64_minkowski_p1_5_single", "f32_cityblock",
    "f64_minkowski_p3", "f64_minkowski_p4", "f32_minkowski_p4",
]


def load_case(name):
    z = np.load(GOLDEN / f"{name}.npz")
    g = {k: z[k] for k in z.files}
    g["_name"] = name
    g["_K"] = int(g["K"])
    g["_metric"] = str(g["metric"])
    g["_p"] = int(g["p"]) if float(g["p"]) == int(g["p"]) else float(g["p"])
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


def knife_edge_rows(ind):
    """Rows whose candidate list holds the query's own id i.  For such a row MP-empiric (mutual_proximity.py:202-212) looks the
    target id i up in the reverse lists (of SOURCE ids) of the candidates c_j; where it is found the reference compares
    d(s_i, t_cj) from the forward pass with the SAME pair's value from the reverse pass -- equal in exact arithmetic, so the
    strict '>' is decided by last-bit rounding inside the reference's BLAS.  Affected: the candidates whose reverse list
    contains i, each by exactly one count (1/K)."""
    n = ind.shape[0]
    return (ind == np.arange(n)[:, None]).any(axis=1)


def _affected(ind_row, row_id, ind_t2s):
    return np.array([row_id in ind_t2s[c] for c in ind_row])


def knife_edge_transform_ok(ref_row, got_row, ind_row, row_id, K, ind_t2s, rtol=1e-5):
    """Unsorted transform output of a knife-edge row: candidates whose reverse list does not contain the query id must
    agree; the others may differ by one count."""
    aff = _affected(ind_row, row_id, ind_t2s)
    if not np.allclose(got_row[~aff], ref_row[~aff], rtol=rtol, atol=1e-9):
        return False
    return bool(np.all(np.abs(got_row[aff] - ref_row[aff]) <= 1.0 / K + 1e-9))


def knife_edge_topk_ok(ref_d, ref_i, got_d, got_i, row_id, K, ind_t2s, rtol=1e-5):
    """Final (sorted) result of a knife-edge row, compared tie-tolerantly: ids returned by both sides carry the same value
    (affected candidates: within one count), at most as many ids are swapped in / out as there are affected candidates,
    and neither side's worst value beats the other's by more than one count."""
    step = 1.0 / K + 1e-9
    ref = dict(zip(ref_i.tolist(), ref_d.tolist()))
    got = dict(zip(got_i.tolist(), got_d.tolist()))
    ids = sorted(set(ref) | set(got))
    aff = dict(zip(ids, _affected(np.array(ids), row_id, ind_t2s).tolist()))
    for idx in set(ref) & set(got):
        tol = step if aff[idx] else rtol * abs(ref[idx]) + 1e-9
        if abs(ref[idx] - got[idx]) > tol:
            return False
    n_aff = max(1, sum(aff.values()))
    if len(set(ref) - set(got)) > n_aff or len(set(got) - set(ref)) > n_aff:
        return False
    return bool(got_d.max() <= ref_d.max() + step and ref_d.max() <= got_d.max() + step)
