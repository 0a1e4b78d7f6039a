"""The C ABI's own collectives (`kz_comm_*`, kiez_amd/csrc/kz_comm.hip: RCCL loaded with dlopen, run on the context's stream) --
what a host WITHOUT torch.distributed binds to shard the path (include/kiez_amd.h "multi-GPU"; the reference's only multi-device
call is kiez/neighbors/approximate/faiss.py:138).  Single rank (the GPU box has one MI355X), every collective forced:
  * each call against its definition on raw device buffers (broadcast, all-gather, all-to-all with ragged blocks, min all-reduce);
  * the whole sharded pipeline (ShardedKiez + HipEngine) over `RcclComm` -- no process group exists in the process -- against the
    oracle for every hubness kind, through the shared sweep and its exchange step.
Subprocess: the communicator is created before anything else has touched the GPU."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r"""
import os, sys, tempfile, warnings
sys.path.insert(0, %r)
os.environ["KIEZ_AMD_WITH_TORCH"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
from kiez_amd.distributed import HipEngine, RcclComm, ShardedKiez
from oracle import kiez_oracle as O
from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
import torch.distributed as dist
assert not dist.is_initialized()
warnings.simplefilter("ignore")
eng = HipEngine(0)
path = os.path.join(tempfile.mkdtemp(), "kz_comm_id")
comm = RcclComm.from_file(eng, 0, 1, path, always=True, time_collectives=True)
assert comm.rank == 0 and comm.world == 1 and len(open(path, "rb").read()) == 128

# ---- each collective on raw buffers ----
dev = eng.device
a = torch.arange(1000, dtype=torch.float64, device=dev)
b = a.clone()
comm.broadcast(b, 0)
assert torch.equal(a, b)
rows = torch.rand(37, 5, dtype=torch.float32, device=dev)
assert torch.equal(comm.all_gather_rows(rows, [37]), rows)
got = comm.all_to_all_rows(rows, [37])
assert tuple(got.shape) == (1, 37, 5) and torch.equal(got[0], rows)
m = torch.tensor([3.5, -1.0], dtype=torch.float64, device=dev)
comm.all_reduce_min(m)
assert m.tolist() == [3.5, -1.0]
assert comm.all_gather_vec([7, 8, 9], dev) == [[7, 8, 9]]
eng.sync()

# ---- the sharded pipeline over these collectives ----
eng.ctx.set_option("dual_force", 1)
rng = np.random.RandomState(11)
source = rng.rand(1100, 40).astype(np.float32)
target = rng.rand(900, 40).astype(np.float32)
CASES = [("none", None, {}, "euclidean", False), ("csls", "CSLS", {}, "euclidean", False),
         ("ls", "LocalScaling", {"method": "standard"}, "euclidean", False),
         ("mp_normal", "MutualProximity", {"method": "normal"}, "euclidean", False),
         ("mp_empiric", "MutualProximity", {"method": "empiric"}, "euclidean", False),
         ("dsl", "DisSimLocal", {}, "sqeuclidean", False), ("csls_single", "CSLS", {}, "euclidean", True)]
K, k = 10, 5
comm.reset_timers()
for name, hub, kw, metric, single in CASES:
    sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=kw, engine=eng, comm=comm)
    sk.fit(source, None if single else target, single_source=single)
    d, i = sk.kneighbors(k)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    od, oi = O.kiez_pipeline(source, None if single else target, K, k, metric, 2, hub, kw)
    keep = np.ones(len(i), dtype=bool)
    if name == "mp_empiric":
        keep &= ~knife_edge_rows(O.knn_exact(source, target, K, "euclidean")[1])
        ind_t2s = O.knn_exact(target, source, K, "euclidean")[1]
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(od[r], oi[r], d[r], i[r], r, K, ind_t2s), (name, r)
    assert np.array_equal(i[keep], oi[keep]), name
    assert np.allclose(d[keep], od[keep], rtol=1e-5, atol=1e-6), name
eng.sync()
tr = comm.traffic()
assert tr["broadcast"]["calls"] >= 6 and tr["all_to_all"]["calls"] == 5 and tr["all_gather"]["calls"] >= 5 and tr["all_reduce"]["calls"] >= 1, tr
ms = comm.timers_ms()
assert set(ms) >= {"broadcast_exposed", "all_to_all", "all_gather"} and all(v >= 0 for v in ms.values()), ms
assert not dist.is_initialized()          # torch.distributed was never involved
comm.close()
print("KZ_COMM_OK", tr)
"""


def test_c_abi_collectives_and_the_sharded_pipeline_over_them():
    r = subprocess.run([sys.executable, "-c", SCRIPT % str(ROOT)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "KZ_COMM_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def test_without_rccl_the_calls_say_so(tmp_path):
    """A bad unique id / rank is refused with a message, not a crash."""
    import ctypes as C

    from kiez_amd import _native as N
    lib = N.load()
    ctx = N.Context.get()
    h = C.c_void_p()
    buf = C.create_string_buffer(128)
    assert lib.kz_comm_create(ctx.handle, buf, 3, 2, C.byref(h)) != 0 and b"rank 3 of 2" in lib.kz_last_error()
    assert lib.kz_comm_create(None, buf, 0, 1, C.byref(h)) != 0
    assert lib.kz_comm_destroy(None) == 0
