"""`ShardedKiez(HipEngine)` over a REAL RCCL process group (single rank: the GPU box has one MI355X).

The world-size-2 gloo test (tests/test_distributed_cpu.py) covers the sharding / exchange logic with a CPU engine; this one
covers what it cannot: the product engine (C ABI on torch CUDA tensors, one HIP stream shared by torch, RCCL and our
kernels) with every collective of the pipeline actually issued (`KIEZ_AMD_FORCE_COLLECTIVES=1` runs broadcast /
all_gather / all_reduce / all_to_all even at world size 1).  Subprocess: torch must be imported and the process group created before
any other GPU call, and the pytest process has usually loaded libkiez_amd.so already."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r"""
import os, sys, warnings
sys.path.insert(0, %r)
os.environ["KIEZ_AMD_WITH_TORCH"] = "1"
os.environ["KIEZ_AMD_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = "29617"
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # RCCL, before any other GPU call
import numpy as np
from kiez_amd.distributed import Comm, HipEngine, ShardedKiez
from oracle import kiez_oracle as O
from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
warnings.simplefilter("ignore")
eng = HipEngine(0)
eng.ctx.set_option("dual_force", 1)   # the shared sweep (kz_knn_dual) also on these small shapes: its exchange step runs too
comm = Comm()
assert comm.always and comm.world == 1
calls = {"broadcast": 0, "all_gather_into_tensor": 0, "all_reduce": 0, "all_to_all_single": 0}
for name in calls:
    orig = getattr(dist, name)
    def wrap(*a, _o=orig, _n=name, **k):
        calls[_n] += 1
        return _o(*a, **k)
    setattr(dist, name, wrap)
rng = np.random.RandomState(11)
source = rng.rand(1100, 40).astype(np.float32)
target = rng.rand(900, 40).astype(np.float32)
CASES = [("none", None, {}, "euclidean", False), ("csls", "CSLS", {}, "euclidean", False),
         ("ls", "LocalScaling", {"method": "standard"}, "euclidean", False),
         ("nicdm", "LocalScaling", {"method": "nicdm"}, "minkowski", False),
         ("mp_normal", "MutualProximity", {"method": "normal"}, "euclidean", False),
         ("mp_empiric", "MutualProximity", {"method": "empiric"}, "euclidean", False),
         ("dsl", "DisSimLocal", {}, "sqeuclidean", False),
         ("csls_single", "CSLS", {}, "euclidean", True), ("none_single", None, {}, "euclidean", True)]
K, k = 10, 5
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
# (every hubness kind takes its reverse lists out of the shared sweep and merges them after ONE all-to-all: six two-source cases)
assert calls["broadcast"] >= 7 and calls["all_gather_into_tensor"] >= 9 and calls["all_reduce"] >= 1 and calls["all_to_all_single"] == 6, calls
print("collectives", calls)
dist.barrier()
dist.destroy_process_group()
print("SHARDED_RCCL_OK")
"""


def test_sharded_kiez_hip_engine_over_rccl_single_rank():
    r = subprocess.run([sys.executable, "-c", SCRIPT % str(ROOT)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SHARDED_RCCL_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def test_bench_line_of_a_launched_run_carries_check_and_cpu_baseline(tmp_path):
    """`bench.py` as torch.distributed.run starts it (RANK / WORLD_SIZE / MASTER_* in the environment, RCCL process group), one
    rank, every collective forced: the line of a launched run -- the code path of `--gpus 8` -- carries the oracle check
    (fit state + rows of every rank's shard), recall@k and the CPU baseline, for both ways of getting the target onto the ranks."""
    import json
    import os
    for upload, port in (("broadcast", "29631"), ("local", "29633")):
        env = {**os.environ, "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port,
               "KIEZ_AMD_FORCE_COLLECTIVES": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--workload", "c2", "--steps", "2", "--warmup", "1",
                            "--no-others", "--target-upload", upload, "--detail", str(tmp_path / "detail.json")],
                           capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
        assert r.returncode == 0, r.stderr[-6000:]
        short = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][-1]
        assert len(short) < 4096 and json.loads(short)["check"]["index_rows_identical"] == 1024 and json.loads(short)["cpu_baseline"]["value"] > 0
        line = json.loads((tmp_path / "detail.json").read_text())     # (the full record the compact stdout line points to)
        chk = line["check"]
        assert line["recall_at_k"] == chk["recall_at_k"] == 1.0
        assert chk["index_rows_identical"] == chk["rows"] == 1024 and chk["index_rows_identical_per_rank"] == [1024]
        assert chk["fit_state_rows"] == 1024 and chk["fit_state_max_rel_err"] < 1e-12
        assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["workload_rows"]["n_source_total"] == 100_000
        traffic = line["collective_traffic_per_step"]
        assert ("broadcast" in traffic) == (upload == "broadcast") and traffic["all_to_all"]["calls"] == 1
        assert line["config"]["target_upload"] == upload and line["value"] > 1e6
