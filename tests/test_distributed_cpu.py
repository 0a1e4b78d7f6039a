"""gloo tests (world size 2, 3 and 8) of the multi-GPU sharding logic (kiez_amd.distributed.ShardedKiez) with the CPU engine.
The sharded run must equal the single-process oracle pipeline on the concatenated source -- for every hubness kind, with
uneven shards, target row counts that do not divide, K = 50, exact ties across shards, and every route that leaves the
shared sweep (single source, shards smaller than K, fewer targets than K, the merge capacity)."""
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
SEVEN = [c for c in CASES if not c[4]]
# the rest of the Minkowski family (a metric given as (name, p) carries the exponent): float32 rows, so that the merge across
# shards must go by the exact ordering values where the returned distance is a rounded function of them (minkowski[p])
FAMILY = [
    ("none", None, {}, "manhattan", False),
    ("csls", "CSLS", {}, "cityblock", False),
    ("mp_empiric", "MutualProximity", {"method": "empiric"}, ("minkowski", 3), False),
    ("nicdm", "LocalScaling", {"method": "nicdm"}, "chebyshev", False),
    ("mp_normal", "MutualProximity", {"method": "normal"}, ("minkowski", 1.5), False),
    ("ls_single", "LocalScaling", {"method": "standard"}, ("minkowski", 1), True),
]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scenario(name):
    """(source, target, K, k, cases, engine attributes, expected `shared` per case name or None = by rule)."""
    rng = np.random.RandomState(17)
    if name == "base":          # uneven shards (203 rows), 157 targets: neither divides by 2, 3 or 8
        return rng.rand(203, 12), rng.rand(157, 12), 7, 4, CASES, {}, None
    if name == "family":
        return rng.rand(150, 9).astype(np.float32), rng.rand(113, 9).astype(np.float32), 7, 4, FAMILY, {}, None
    if name == "k50":           # the C3 candidate count: world x K = 100 ... 400 entries per merged row
        return rng.rand(431, 10), rng.rand(333, 10), 50, 50, SEVEN, {}, None
    if name == "ties":          # integer data: all arithmetic exact, MANY exact distance ties across the shards
        return (rng.randint(0, 3, (180, 4)).astype(np.float64), rng.randint(0, 3, (150, 4)).astype(np.float64), 9, 9,
                [c for c in SEVEN if c[0] in ("none", "csls", "ls", "mp_empiric", "dsl")], {}, None)
    if name == "few_targets":   # n_t < K <= every shard (the shared sweep must not be chosen)
        return rng.rand(60, 8), rng.rand(5, 8), 7, 4, [c for c in SEVEN if c[0] in ("csls", "mp_empiric", "dsl", "none")], {}, False
    if name == "small_shards":  # K > rows of a shard: every rank must take the gathered-source route
        return rng.rand(41, 6), rng.rand(64, 6), 7, 3, [c for c in SEVEN if c[0] in ("csls", "mp_empiric", "dsl")], {}, False
    if name == "merge_cap_in":  # world x K == capacity of the merge: shared
        return rng.rand(120, 6), rng.rand(90, 6), 7, 4, [c for c in SEVEN if c[0] in ("csls", "mp_empiric")], {"MAX_MERGE": 14}, True
    if name == "merge_cap_out":  # one more: two searches
        return rng.rand(120, 6), rng.rand(90, 6), 8, 4, [c for c in SEVEN if c[0] in ("csls", "mp_empiric")], {"MAX_MERGE": 14}, False
    raise KeyError(name)


def _worker(rank, world, port, scenario, q):
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import warnings

        import torch.distributed as dist
        from kiez_amd.distributed import Comm, ShardedKiez, row_slice
        from oracle import kiez_oracle as O
        from tests.cpu_engine import OracleEngine
        from tests.golden_util import knife_edge_rows, knife_edge_topk_ok

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        source, target, K, k, cases, attrs, expect_shared = _scenario(scenario)
        b, c = row_slice(len(source), rank, world)
        results = {}
        warnings.simplefilter("ignore")
        for name, hub, kw, metric, single in cases:
            eng = OracleEngine()
            for a, v in attrs.items():
                setattr(eng, a, v)
            comm = Comm()
            metric, p = (metric, 2) if isinstance(metric, str) else metric
            sk = ShardedKiez(n_candidates=K, algorithm_kwargs={"metric": metric, "p": p}, hubness=hub, hubness_kwargs=kw, engine=eng, comm=comm)
            sk.fit(source[b:b + c], None if single else (target if rank == 0 else None), single_source=single)
            d, i = sk.kneighbors(k)
            od, oi = O.kiez_pipeline(source, None if single else target, K, k, metric, p, hub, kw)
            metric = O.canonical_metric(metric, p)
            want = (hub is not None and not single) if expect_shared is None else (expect_shared and hub is not None)
            assert bool(getattr(sk, "shared", False)) == want, (name, sk.shared, want)
            tr = comm.traffic()
            if want and world > 1:
                assert tr["all_to_all"]["calls"] == 1, (name, tr)     # ONE exchange, whatever the kind needs
            elif world > 1:
                assert "all_to_all" not in tr, (name, tr)
            # (atol: in single-source mode a row's distance to itself is sqrt(|x|^2 - 2 x.x + |x|^2) -- 0 or ~1e-8 depending on
            #  how BLAS blocks the CPU engine's gemm for THIS shard's shape; the HIP engine's canonical dot product gives 0)
            got_i, got_d, ref_i, ref_d = i.numpy(), d.numpy(), oi[b:b + c], od[b:b + c]
            keep = np.ones(len(got_i), dtype=bool)
            if name == "mp_empiric":
                # the fit state itself: the merged reverse lists must BE the single-process lists (order included)
                rd, ri = O.knn_exact(target, source, K, metric)
                assert np.array_equal(sk.state["ind_t2s"].numpy(), ri), "merged reverse indices differ from the single-process search"
                assert np.allclose(sk.state["dist_t2s"].numpy(), rd, rtol=1e-12, atol=1e-12)
                # rows whose candidates include the query's own id are decided by last-bit rounding in the reference
                # (tests/golden_util.py: knife_edge_*): compared tie-tolerantly, nothing dropped
                fi = O.knn_exact(source, target, min(K, len(target)), metric)[1]
                keep = ~knife_edge_rows(fi)[b:b + c]
                for r in np.flatnonzero(~keep):
                    assert knife_edge_topk_ok(ref_d[r], ref_i[r], got_d[r], got_i[r], b + r, K, ri), (name, b + r)
            results[name] = (bool(np.array_equal(got_i[keep], ref_i[keep])),
                             bool(np.allclose(got_d[keep], ref_d[keep], rtol=1e-9, atol=1e-7 if single else 1e-9)), tuple(i.shape),
                             int(oi.shape[1]))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, results, None))
    except Exception:  # pragma: no cover
        q.put((rank, None, traceback.format_exc()))


def _run(world, scenario):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, results, err in out:
        assert err is None, f"rank {rank} failed:\n{err}"
        for name, (idx_ok, dist_ok, shape, k_ref) in results.items():
            assert idx_ok, f"world {world} rank {rank} {scenario}/{name}: indices differ from the single-process oracle"
            assert dist_ok, f"world {world} rank {rank} {scenario}/{name}: distances differ"
            assert shape[1] == k_ref


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_pipeline_gloo_every_kind(world):
    _run(world, "base")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_gloo_minkowski_family(world):
    _run(world, "family")


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_pipeline_gloo_k50_all_seven_kinds_share_the_sweep(world):
    _run(world, "k50")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_gloo_exact_ties_across_shards(world):
    _run(world, "ties")


@pytest.mark.parametrize("scenario,world", [("few_targets", 2), ("small_shards", 8), ("merge_cap_in", 2), ("merge_cap_out", 2)])
def test_sharded_pipeline_gloo_routes_that_leave_the_shared_sweep(scenario, world):
    _run(world, scenario)


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
    search directions out of one `knn_dual` call (every hubness kind qualifies) or searches twice."""
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


def _cache_worker(rank, world, port, q):
    """`ShardedKiez(cache_target=True)`: the same target tensor in consecutive fits is broadcast once; a new tensor or an in-place
    write (torch's version counter) is broadcast again; results always those of the target as it is NOW."""
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import warnings

        import torch
        import torch.distributed as dist
        from kiez_amd.distributed import Comm, ShardedKiez, row_slice
        from oracle import kiez_oracle as O
        from tests.cpu_engine import OracleEngine

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        warnings.simplefilter("ignore")
        rng = np.random.RandomState(5)
        source, source2, target = rng.rand(90, 8), rng.rand(70, 8), rng.rand(60, 8)
        tgt = torch.from_numpy(target.copy())       # the caller's tensor on the engine's device (CPU engine: host memory)
        comm = Comm()
        sk = ShardedKiez(n_candidates=6, algorithm_kwargs={"metric": "euclidean"}, hubness="CSLS", engine=OracleEngine(), comm=comm,
                         cache_target=True)
        log = []

        def step(src, t_arg, want_bcast, t_now):
            comm.reset_timers()
            b, c = row_slice(len(src), rank, world)
            sk.fit(src[b:b + c], t_arg if rank == 0 else None)
            d, i = sk.kneighbors(4)
            od, oi = O.kiez_pipeline(src, t_now, 6, 4, "euclidean", 2, "CSLS", {})
            calls = comm.traffic().get("broadcast", {"calls": 0})["calls"]
            log.append((calls == (1 if want_bcast else 0), bool(np.array_equal(i.numpy(), oi[b:b + c])),
                        bool(np.allclose(d.numpy(), od[b:b + c], rtol=1e-9, atol=1e-9))))
        step(source, tgt, True, target)            # first fit: broadcast
        step(source2, tgt, False, target)          # another source batch, the SAME target tensor: no broadcast
        step(source, tgt, False, target)
        if rank == 0:
            tgt.mul_(0.5)                          # in-place write on rank 0: the version counter moves -> broadcast again
        step(source, tgt, True, target * 0.5)
        step(source2, tgt, False, target * 0.5)
        other = rng.rand(60, 8)
        step(source, other, True, other)           # a host array: copied into a new tensor by every fit -> never cached
        step(source, other, True, other)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, log, None))
    except Exception:  # pragma: no cover
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_target_replica_is_reused_while_the_target_tensor_is_unchanged(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cache_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, log, err in out:
        assert err is None, f"rank {rank} failed:\n{err}"
        assert len(log) == 7 and all(all(t) for t in log), (rank, log)
