"""A bounded, fixed-seed slice of the randomised cross-checks (tools/fuzz_tiers.py: the three first-pass tiers and the oracle on
random shapes, with the tier probe's ladder and the wide route drawn at random; tools/fuzz_dual.py: the shared sweep against two
ordinary searches with its knobs drawn at random) as part of `pytest -m gpu`.  Both synchronisation races of rounds 3 and 4 (ring
slot hand-over; the 64-query kernel's prologue) were invisible to the parity suite and visible only here -- about a minute each.
The reference has no counterpart: its search is scikit-learn's (sklearn_nearest_neighbors.py:96-101); every route must give its
float64 neighbour order."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(tool, *args, timeout=900):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / tool), *map(str, args)], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert " bad 0" in r.stdout.splitlines()[-1], tail
    return r.stdout


@pytest.mark.parametrize("seed", [505, 506])
def test_tiers_agree_on_random_shapes(seed):
    out = _run("fuzz_tiers.py", 45, seed)
    assert "cases 45 bad 0" in out


@pytest.mark.parametrize("seed", [505, 506])
def test_shared_sweep_equals_two_searches_on_random_shapes(seed):
    out = _run("fuzz_dual.py", 20, seed)
    assert "cases 20 bad 0" in out


def test_minkowski_family_on_random_shapes():
    """tools/fuzz_family.py: manhattan / chebyshev / minkowski[p] (the register-tiled VALU kernel + exact selection) against the
    oracle's restatement of scikit-learn's DistanceMetric32 / 64 -- ragged sizes, ties, self queries."""
    out = _run("fuzz_family.py", 8, 505)      # (VALU kernels outside north_star's metrics: a short slice; tools/job_fuzz5.sh runs hundreds)
    assert "cases 8 bad 0" in out
