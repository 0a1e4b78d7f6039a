"""BASELINE.json configuration 4 AT ITS STATED SIZE on one GPU: 2M source rows x 1M target rows, d = 300, k = 10, CSLS.

The per-GPU share (250k x 1M) is covered by tests/test_gpu_northstar.py; the full configuration crosses limits the share never
touches: four query chunks of 524288 rows per sweep, event buffers / logs sized for 2M x 1M, the footprint gate of
`kz_knn_dual`.  Reference path: kiez/hubness_reduction/base.py:33-50 (fit: reverse search), :89-105 (kneighbors), csls.py:85-96.

  (a) `Kiez(hubness="CSLS").fit(source, target).kneighbors(10)` through the shared sweep: samples of BOTH raw kNN results (4 096 source rows, 2 048 target rows)
      array_equal to the oracle's exact float64 search, `r_train` bit-equal, the final (dist, ind) equal to the oracle's CSLS
      transform + `_sort` on the sampled rows;
  (b) ALL rows equal to the concatenation of EIGHT `ShardedKiez` ranks (the north-star partitioning: source row-sharded, target
      broadcast from rank 0, one all-to-all of the per-shard reverse lists, kz_merge_topk over eight real segments) run as
      eight processes of the product engine on the one GPU of the box (collectives staged over gloo, tests/staged_comm.py).
"""
import os
import socket
import subprocess
import sys
import warnings
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def c4_single():
    from kiez_amd import Kiez
    from tests import c4_data as D
    s, t = D.full_source(), D.target_rows()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=D.K, algorithm="SklearnNN", algorithm_kwargs={"metric": "euclidean"}, hubness="CSLS")
        kz.fit(s, t)
        nn = kz.algorithm
        stats = (dict(nn.last_stats), dict(nn.last_stats_reverse or {}))
        k_fwd, fd_dev, fi_dev = nn._forward
        out = {"s": s, "t": t, "stats": stats, "k_fwd": k_fwd, "fd": fd_dev.numpy(), "fi": fi_dev.numpy(),
               "rd": kz.hubness.r_dist_train_.numpy(), "ri": kz.hubness.r_ind_train_.numpy(),
               "r_train": kz.hubness._r_train_dev.numpy()}
        out["dist"], out["ind"] = kz.kneighbors(D.K)
    del kz
    return out


def test_c4_full_size_against_the_oracle(c4_single):
    from oracle import kiez_oracle as O
    from tests import c4_data as D
    r = c4_single
    s, t, K = r["s"], r["t"], D.K
    st_f, st_r = r["stats"]
    # one sweep served both directions; nothing fell back to the exact kernels, the rounding bound held
    assert st_f["dual"] == 1 and st_r.get("dual") == 1, (st_f, st_r)
    assert st_f["max_err_ratio"] < 1.0 and st_r["max_err_ratio"] < 1.0
    assert st_f["n_fallback_rows"] == st_f["n_spec_rows"] and st_r["n_fallback_rows"] == st_r["n_spec_rows"]   # (no row beyond the handful the speculative exact launches answer)
    fd, fi, rd, ri, dist, ind = r["fd"], r["fi"], r["rd"], r["ri"], r["dist"], r["ind"]
    assert r["k_fwd"] == K and fd.shape == (D.N_SOURCE, K) and rd.shape == (D.N_TARGET, K) and dist.shape == (D.N_SOURCE, K)
    assert ind.dtype == np.int64 and dist.dtype == np.float64
    assert (np.diff(fd, axis=1) >= 0).all() and (np.diff(rd, axis=1) >= 0).all() and (np.diff(dist, axis=1) >= 0).all()
    assert fi.min() >= 0 and fi.max() < D.N_TARGET and ri.min() >= 0 and ri.max() < D.N_SOURCE
    assert ind.min() >= 0 and ind.max() < D.N_TARGET

    # rows of every 524288-row chunk of the sweep and of every shard, plus both ends
    threads = max(1, min(32, os.cpu_count() or 1))     # (oracle.knn_exact works on its row chunks with this many threads)
    rows = np.unique(np.concatenate([np.random.RandomState(1).choice(D.N_SOURCE, 4088, replace=False),
                                     [0, 524287, 524288, 1048575, 1048576, 1572864, D.N_SOURCE - 1, D.SHARD_ROWS]]))
    od, oi = O.knn_exact(s[rows], t, K, "euclidean", threads=threads)
    np.testing.assert_array_equal(fi[rows], oi)
    np.testing.assert_array_equal(fd[rows], od)                # float32 inputs: bit-identical distances (sqrt rule)
    trows = np.unique(np.concatenate([np.random.RandomState(2).choice(D.N_TARGET, 2046, replace=False), [0, D.N_TARGET - 1]]))
    ord_, ori = O.knn_exact(t[trows], s, K, "euclidean", threads=threads)
    np.testing.assert_array_equal(ri[trows], ori)
    np.testing.assert_array_equal(rd[trows], ord_)
    np.testing.assert_array_equal(r["r_train"][trows], ord_.mean(axis=1))          # csls.py:90
    tr = 2 * od - od.mean(axis=1).reshape(-1, 1) - r["r_train"][oi]                 # csls.py:93-95
    sd, si = O.sort_topk(tr, oi, K)                                                 # base.py:72-87
    np.testing.assert_array_equal(ind[rows], si)
    np.testing.assert_array_equal(dist[rows], sd)


RANK_SCRIPT = r"""
import os, sys, warnings
sys.path.insert(0, %(root)r)
os.environ["KIEZ_AMD_WITH_TORCH"] = "1"
import numpy as np
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from kiez_amd.distributed import HipEngine, ShardedKiez
from tests import c4_data as D
from tests.staged_comm import StagedComm
warnings.simplefilter("ignore")
assert world == D.N_SHARDS
eng = HipEngine(0)
shard = D.source_shard(rank)
target = D.target_rows() if rank == 0 else None
comm = StagedComm()
sk = ShardedKiez(n_candidates=D.K, algorithm_kwargs={"metric": "euclidean"}, hubness="CSLS", engine=eng, comm=comm)
sk.fit(shard, target)
assert sk.shared and eng.last_stats["dual"] == 1 and eng.last_stats_reverse["dual"] == 1, (eng.last_stats, eng.last_stats_reverse)
dd, ii = sk.kneighbors(D.K)
np.save(os.path.join(%(out)r, f"dist_{rank}.npy"), dd.cpu().numpy())
np.save(os.path.join(%(out)r, f"ind_{rank}.npy"), ii.cpu().numpy())
if rank == 0:
    np.save(os.path.join(%(out)r, "r_t.npy"), sk.state["r_t"].cpu().numpy())
dist.barrier()
dist.destroy_process_group()
print("C4_RANK_OK", rank)
"""


def test_c4_full_size_equals_eight_sharded_ranks(c4_single, tmp_path):
    from tests import c4_data as D
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = RANK_SCRIPT % {"root": str(ROOT), "out": str(tmp_path)}
    procs = []
    for r in range(D.N_SHARDS):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(D.N_SHARDS), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=1500))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so_, se_)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"C4_RANK_OK {r}" in so_, f"rank {r}:\n{so_[-2000:]}\n{se_[-4000:]}"
    res = c4_single
    np.testing.assert_array_equal(np.load(tmp_path / "r_t.npy"), res["r_train"])
    for r in range(D.N_SHARDS):
        b, e = r * D.SHARD_ROWS, (r + 1) * D.SHARD_ROWS
        np.testing.assert_array_equal(np.load(tmp_path / f"ind_{r}.npy"), res["ind"][b:e], err_msg=f"shard {r}: indices")
        np.testing.assert_array_equal(np.load(tmp_path / f"dist_{r}.npy"), res["dist"][b:e], err_msg=f"shard {r}: distances")
