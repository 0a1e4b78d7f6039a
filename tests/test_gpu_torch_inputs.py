"""Torch tensors as inputs ("next" row f-3: zero-copy tensor input).  Runs in a subprocess because torch must be imported
before libkiez_amd.so is loaded (one HIP runtime per process) and the pytest process has usually loaded the library
already."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r"""
import sys, warnings
sys.path.insert(0, %r)
import torch                       # first: its HIP runtime is the one the process uses
import numpy as np
from kiez_amd import Kiez
from oracle import kiez_oracle as O
warnings.simplefilter("ignore")
rng = np.random.RandomState(0)
s, t = rng.rand(500, 32).astype(np.float32), rng.rand(400, 32).astype(np.float32)
for hub in (None, "CSLS", "DisSimLocal"):
    od, oi = O.kiez_pipeline(s, t, 10, 5, "euclidean", 2, hub, {})
    for dev in ("cuda", "cpu"):
        st, tt = torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev)
        d, i = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness=hub).fit(st, tt).kneighbors(5)
        assert isinstance(d, torch.Tensor) and isinstance(i, torch.Tensor), (type(d), type(i))
        assert d.device.type == dev and i.dtype == torch.int64
        assert np.array_equal(i.cpu().numpy(), oi), (hub, dev)
        assert np.allclose(d.cpu().numpy(), od, rtol=1e-5, atol=1e-6), (hub, dev)
# device-resident rows are indexed without waiting for anything (zero-copy, asynchronous kz_matrix_create): a non-finite
# value is reported by the first search instead of by fit's index construction
bad = torch.from_numpy(s).to("cuda").clone()
bad[3, 2] = float("nan")
try:
    Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}).fit(bad, torch.from_numpy(t).to("cuda")).kneighbors(5)
    raise SystemExit("NaN input was not rejected")
except ValueError as e:
    assert "NaN" in str(e), e
print("TORCH_INPUTS_OK")
"""


def test_torch_tensor_inputs_subprocess():
    r = subprocess.run([sys.executable, "-c", SCRIPT % str(ROOT)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TORCH_INPUTS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
