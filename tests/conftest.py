import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


# `pytest -x` stops at the first failure: the parity tests proper (the GPU path against the oracle and the golden vectors, the
# BASELINE configurations at full size, both directions of a fit) run BEFORE the (f)-row features, the routes' own tests and
# the tooling checks.  Files not named here keep their alphabetical order behind these.
PARITY_FIRST = ["test_gpu_parity.py", "test_gpu_fullsize.py", "test_gpu_northstar.py", "test_gpu_dual.py", "test_gpu_c4_full.py",
                "test_gpu_mp_empiric.py", "test_gpu_spec_rescue.py", "test_gpu_near_ties.py", "test_gpu_cosine_f32.py",
                "test_gpu_integration_stub.py", "test_gpu_bench_contract.py"]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(PARITY_FIRST)}
    items.sort(key=lambda it: (rank.get(Path(str(it.fspath)).name, len(rank)), ))   # (stable: the order inside a file is kept)
