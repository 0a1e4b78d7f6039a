"""TWO ranks of `ShardedKiez(HipEngine)` on the one GPU of the box: the product engine (C ABI on torch CUDA tensors: kz_knn_dual,
kz_pair_values, kz_merge_topk, the rescaling kernels) with REAL multi-shard data, which neither the gloo tests (CPU engine) nor the
single-rank RCCL test (one segment per merge) exercise.  RCCL refuses two ranks on one device, so the collectives of this test
run over gloo with the tensors staged through host memory (`_StagedComm`, test infrastructure; the collectives themselves are
covered by tests/test_gpu_sharded_rccl.py).  Every hubness kind, uneven shards, against the single-process oracle pipeline."""
import os
import socket
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
import numpy as np
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from kiez_amd.distributed import Comm, HipEngine, ShardedKiez, row_slice
from oracle import kiez_oracle as O
from tests.golden_util import knife_edge_rows, knife_edge_topk_ok
warnings.simplefilter("ignore")


class _StagedComm(Comm):
    # gloo collectives on host copies of the engine's CUDA tensors (two ranks share one GPU: no RCCL)
    def broadcast(self, t, src=0):
        h = t.cpu()
        self.dist.broadcast(h, src=src)
        t.copy_(h)
        return t

    def broadcast_begin(self, t, src=0):
        self.broadcast(t, src)
        return None

    def all_gather_rows(self, t, counts):
        return super().all_gather_rows(t.cpu(), counts).to(t.device)

    def all_to_all_rows(self, t, counts):
        return super().all_to_all_rows(t.cpu(), counts).to(t.device)

    def all_gather_vec(self, values, device):
        return super().all_gather_vec(values, torch.device("cpu"))

    def all_reduce_min(self, t):
        h = t.cpu()
        self.dist.all_reduce(h, op=self.dist.ReduceOp.MIN)
        t.copy_(h)
        return t


eng = HipEngine(0)
eng.ctx.set_option("dual_force", 1)      # the shared sweep also on these small shards
rng = np.random.RandomState(29)
source = rng.rand(9001, 40).astype(np.float32)
target = rng.rand(7001, 40).astype(np.float32)
if world == 2:
    b, c = (0, 5200) if rank == 0 else (5200, 3801)      # uneven shards
else:
    b, c = row_slice(len(source), rank, world)
CASES = [("none", None, {}, "euclidean"), ("csls", "CSLS", {}, "euclidean"), ("ls", "LocalScaling", {"method": "standard"}, "euclidean"),
         ("nicdm", "LocalScaling", {"method": "nicdm"}, "cosine"), ("mp_normal", "MutualProximity", {"method": "normal"}, "sqeuclidean"),
         ("mp_empiric", "MutualProximity", {"method": "empiric"}, "euclidean"), ("mp_empiric_cos", "MutualProximity", {"method": "empiric"}, "cosine"),
         ("dsl", "DisSimLocal", {}, "euclidean"), ("dsl_sq", "DisSimLocal", {}, "sqeuclidean")]
K, k = 10, 6
for name, hub, kw, metric in CASES:
    s_in = source.astype(np.float64) if metric == "cosine" else source
    t_in = target.astype(np.float64) if metric == "cosine" else target
    comm = _StagedComm()
    sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=kw, engine=eng, comm=comm)
    sk.fit(s_in[b:b + c], t_in if rank == 0 else None)
    d, i = sk.kneighbors(k)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    od, oi = O.kiez_pipeline(s_in, t_in, K, k, metric, 2, hub, kw)
    od, oi = od[b:b + c], oi[b:b + c]
    if hub is not None:
        assert sk.shared and eng.last_stats["dual"] == 1, (name, eng.last_stats)
        assert comm.traffic()["all_to_all"]["calls"] == 1, (name, comm.traffic())
    keep = np.ones(len(i), dtype=bool)
    if name.startswith("mp_empiric"):
        mc = O.canonical_metric(metric)
        ri = O.knn_exact(t_in, s_in, K, mc)[1]
        assert np.array_equal(sk.state["ind_t2s"].cpu().numpy(), ri), name + ": merged reverse indices differ from the single-GPU search"
        keep = ~knife_edge_rows(O.knn_exact(s_in, t_in, K, mc)[1])[b:b + c]
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(od[r], oi[r], d[r], i[r], b + r, K, ri), (name, b + r)
    assert np.array_equal(i[keep], oi[keep]), name
    assert np.allclose(d[keep], od[keep], rtol=1e-5, atol=1e-6), name
    print(rank, name, "ok", flush=True)
# the rest of the Minkowski family (exact tiled kernel; the shards' reverse lists merge by kz_pair_values' ordering values where the
# returned distance is a rounded function of them): smaller matrices, the oracle's feature loop is slow
fs, ft = source[:3001], target[:2503]
fb, fc = row_slice(len(fs), rank, world)
for name, hub, kw, metric, p in [("fam_manhattan_csls", "CSLS", {}, "manhattan", 2), ("fam_mink3_empiric", "MutualProximity", {"method": "empiric"}, "minkowski", 3),
                                 ("fam_chebyshev_nicdm", "LocalScaling", {"method": "nicdm"}, "chebyshev", 2), ("fam_mink15_none", None, {}, "minkowski", 1.5)]:
    comm = _StagedComm()
    sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric, "p": p}, hubness=hub, hubness_kwargs=kw, engine=eng, comm=comm)
    sk.fit(fs[fb:fb + fc], ft if rank == 0 else None)
    d, i = sk.kneighbors(k)
    d, i = d.cpu().numpy(), i.cpu().numpy()
    od, oi = O.kiez_pipeline(fs, ft, K, k, metric, p, hub, kw)
    od, oi = od[fb:fb + fc], oi[fb:fb + fc]
    if hub is not None:
        assert sk.shared and comm.traffic()["all_to_all"]["calls"] == 1, (name, comm.traffic())
    keep = np.ones(len(i), dtype=bool)
    if "empiric" in name:
        mc = O.canonical_metric(metric, p)
        ri = O.knn_exact(ft, fs, K, mc)[1]
        assert np.array_equal(sk.state["ind_t2s"].cpu().numpy(), ri), name + ": merged reverse indices differ from the single-GPU search"
        keep = ~knife_edge_rows(O.knn_exact(fs, ft, K, mc)[1])[fb:fb + fc]
        for r in np.flatnonzero(~keep):
            assert knife_edge_topk_ok(od[r], oi[r], d[r], i[r], fb + r, K, ri), (name, fb + r)
    assert np.array_equal(i[keep], oi[keep]), name
    assert np.allclose(d[keep], od[keep], rtol=1e-5, atol=1e-6), name
    print(rank, name, "ok", flush=True)
dist.barrier()
dist.destroy_process_group()
print("TWO_RANKS_OK", rank)
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [3])      # (three ranks: uneven shards and a target slice per rank; two ranks add nothing to it -- and 22 s to the suite)
def test_two_ranks_of_the_hip_engine_on_one_gpu(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, "-c", SCRIPT % str(ROOT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"TWO_RANKS_OK {rank}" in out, f"rank {rank}:\n{out[-2000:]}\n{err[-4000:]}"
