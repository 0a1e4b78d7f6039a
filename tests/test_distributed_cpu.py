"""World-size-2 gloo tests of the multi-GPU sharding logic (kiez_amd.distributed.ShardedKiez) with the CPU engine.
The sharded run must equal the single-process oracle pipeline on the concatenated source."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

CASES = [
    ("none", None, {}, "euclidean", False),
    ("csls", "CSLS", {}, "euclidean", False),
    ("ls", "LocalScaling", {"method": "standard"}, "euclidean", False),
    ("nicdm", "LocalScaling", {"method": "nicdm"}, "minkowski", False),
    ("mp_normal", "MutualProximity", {"method": "normal"}, "cosine", False),
    ("mp_empiric", "MutualProximity", {"method": "empiric"}, "euclidean", False),
    ("dsl", "DisSimLocal", {}, "sqeuclidean", False),
    ("csls_single", "CSLS", {}, "euclidean", True),
    ("none_single", None, {}, "euclidean", True),
    ("dsl_single", "DisSimLocal", {}, "euclidean", True),
]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from kiez_amd.distributed import Comm, ShardedKiez, row_slice
        from oracle import kiez_oracle as O
        from tests.cpu_engine import OracleEngine

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        rng = np.random.RandomState(17)
        source = rng.rand(203, 12)
        target = rng.rand(157, 12)
        b, c = row_slice(len(source), rank, world)
        results = {}
        for name, hub, kw, metric, single in CASES:
            sk = ShardedKiez(n_candidates=7, algorithm_kwargs={"metric": metric}, hubness=hub, hubness_kwargs=kw,
                             engine=OracleEngine(), comm=Comm())
            sk.fit(source[b:b + c], None if single else (target if rank == 0 else None), single_source=single)
            d, i = sk.kneighbors(4)
            od, oi = O.kiez_pipeline(source, None if single else target, 7, 4, metric, 2, hub, kw)
            # the kinds that only need reverse DISTANCES share one sweep per rank and merge by all-to-all; the others search twice
            assert getattr(sk, "shared", False) == (name in ("csls", "ls", "nicdm", "mp_normal")), (name, sk.shared)
            results[name] = (bool(np.array_equal(i.numpy(), oi[b:b + c])),
                             bool(np.allclose(d.numpy(), od[b:b + c], rtol=1e-9, atol=1e-9)), tuple(i.shape))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, results, None))
    except Exception:  # pragma: no cover
        q.put((rank, None, traceback.format_exc()))


def test_sharded_pipeline_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, results, err in out:
        assert err is None, f"rank {rank} failed:\n{err}"
        for name, (idx_ok, dist_ok, shape) in results.items():
            assert idx_ok, f"rank {rank} case {name}: indices differ from the single-process oracle"
            assert dist_ok, f"rank {rank} case {name}: distances differ"
            assert shape[1] == 4


def test_row_slice_partitions_everything():
    from kiez_amd.distributed import row_slice
    for n in (0, 1, 7, 100, 101):
        for w in (1, 2, 3, 8):
            parts = [row_slice(n, r, w) for r in range(w)]
            assert sum(c for _, c in parts) == n
            pos = 0
            for b, c in parts:
                assert b == pos
                pos += c


@pytest.mark.parametrize("shared", [True, False])
def test_single_process_pipeline_with_and_without_the_shared_sweep(shared):
    """World size 1, no process group: ShardedKiez on the CPU engine must equal the oracle pipeline whether fit() takes both
    search directions out of one `knn_dual` call (every hubness kind qualifies with one rank) or searches twice."""
    import warnings
    from kiez_amd.distributed import Comm, ShardedKiez
    from oracle import kiez_oracle as O
    from tests.cpu_engine import OracleEngine
    rng = np.random.RandomState(23)
    source, target = rng.rand(151, 9), rng.rand(190, 9)
    for name, hub, kw, metric, single in CASES:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = ShardedKiez(n_candidates=6, algorithm_kwargs={"metric": metric}, hubness=hub,
                             hubness_kwargs=dict(kw, shared_sweep=shared), engine=OracleEngine(), comm=Comm())
            sk.fit(source, None if single else target, single_source=single)
            d, i = sk.kneighbors(3)
            od, oi = O.kiez_pipeline(source, None if single else target, 6, 3, metric, 2, hub, kw)
        assert sk.shared == (shared and hub is not None and not single), (name, sk.shared)
        np.testing.assert_array_equal(i.numpy(), oi, err_msg=name)
        np.testing.assert_allclose(d.numpy(), od, rtol=1e-9, atol=1e-9, err_msg=name)
