"""The binding INTEGRATION.md shows a kiez maintainer (section 2: an `NNAlgorithm` subclass over `ctypes`) is EXECUTED here, not
only printed: the code block is taken from the document as it stands; the two things that differ on this box are substituted --
the reference's base class (its package cannot be imported here, SURVEY 8c: this repository's mirror of the same interface stands
in) and the path of the shared library.  The class is then driven through the reference's own call sequence
(`fit(source, target)`, `kneighbors(...)`, kiez/neighbors/neighbor_algorithm_base.py:40-136) and checked against the oracle."""
import re
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _stub_source():
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    block = next(b for b in blocks if "class MI355XNN(NNAlgorithm)" in b)
    block = block.replace("from kiez.neighbors.neighbor_algorithm_base import NNAlgorithm", "from kiez_amd.neighbors import NNAlgorithm")
    block = block.replace('C.CDLL("libkiez_amd.so")', f'C.CDLL({str(ROOT / "kiez_amd" / "libkiez_amd.so")!r})')
    assert "kiez_amd.neighbors" in block and "kiez_amd/libkiez_amd.so" in block
    return block


def test_the_documented_ctypes_binding_runs_and_matches_the_oracle():
    from oracle import kiez_oracle as O
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#MI355XNN", "exec"), ns)
    MI355XNN = ns["MI355XNN"]
    rng = np.random.RandomState(3)
    for metric, dtype in (("minkowski", np.float64), ("cosine", np.float64), ("sqeuclidean", np.float32)):
        s, t = rng.rand(700, 33).astype(dtype), rng.rand(900, 33).astype(dtype)
        nn = MI355XNN(n_candidates=7, metric=metric)
        nn.fit(s, t)
        d, i = nn.kneighbors(k=7)                                   # source -> target
        od, oi = O.knn_exact(s, t, 7, O.canonical_metric(metric))
        np.testing.assert_array_equal(i, oi)
        np.testing.assert_allclose(d, od, rtol=1e-9, atol=1e-12)
        d2, i2 = nn.kneighbors(k=5, query=t, s_to_t=False)          # the reverse search HubnessReduction.fit issues (base.py:37-42)
        od2, oi2 = O.knn_exact(t, s, 5, O.canonical_metric(metric))
        np.testing.assert_array_equal(i2, oi2)
        # single-source mode: the implicit query strips the row itself (neighbor_algorithm_base.py:119)
        nn1 = MI355XNN(n_candidates=4, metric=metric)
        nn1.fit(s)
        i3 = nn1.kneighbors(k=4, return_distance=False)
        np.testing.assert_array_equal(i3, O.knn_exact(s, s, 4, O.canonical_metric(metric), exclude_self=True)[1])
