"""Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

Partitioning (DESIGN.md "Multi-GPU"):
  * every rank owns a ROW SHARD of the source; the target is replicated (RCCL broadcast from rank 0);
  * hubness=None: nothing else moves — each rank searches its shard against the replicated target;
  * hubness != None: `fit` needs the reverse kNN of every target row against ALL source rows.  SHARED SWEEP (default): every
    rank sweeps its shard against the target once (kz_knn_dual): the forward lists of its rows and, for all target rows,
    their K nearest rows INSIDE the shard; an all-to-all hands every rank the per-shard lists of its slice of target rows and
    kz_merge_topk merges them -- by the distances for the kinds that only need reverse distances (CSLS, LocalScaling, MP
    normal), by exact ordering values + global ids (kz_pair_values) for the kinds that need the reverse indices in the
    single-GPU order (MP empiric, DSL).  Then only the per-target-row fit state is all-gathered:
      CSLS / NICDM: mean reverse distance, LS: K-th reverse distance, MP normal: mean + std (8-16 B per row),
      DSL: distance to the local centroid (8 B per row; its centroid gather needs the source shards all-gathered once),
      MP empiric: the merged reverse lists;
    DSL additionally all-reduces ONE scalar (MIN) before its global shift (kiez/hubness_reduction/dis_sim.py:171-173).
    Without the shared sweep (`hubness_kwargs={"shared_sweep": False}`, single-source mode, shards smaller than K): the source
    shards are all-gathered and each rank runs the reverse search for ITS slice of target rows against the full source.
  * `kneighbors` returns the rows of the local shard (global target ids).

The arithmetic is delegated to an *engine*.  `HipEngine` (the product) drives the C ABI on torch CUDA tensors;
tests inject a CPU engine so the sharding / exchange logic runs under gloo with world_size 2.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .neighbors import canonical_metric

_P = C.c_void_p


def _torch():
    import torch
    return torch


# ---------------------------------------------------------------------------------------------------
# engine: the per-rank compute backend
# ---------------------------------------------------------------------------------------------------
class HipEngine:
    """C-ABI calls on torch CUDA tensors (`tensor.data_ptr()` is a plain HBM pointer)."""

    def __init__(self, device: Optional[int] = None):
        os.environ.setdefault("KIEZ_AMD_WITH_TORCH", "1")
        torch = _torch()  # torch first: one HIP runtime per process (kiez_amd/_native.py)
        from . import _native as N
        self.N = N
        self.torch = torch
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        # ONE stream for torch ops, RCCL collectives (they order against the current stream) and our kernels
        self.stream = torch.cuda.Stream(self.device)
        torch.cuda.synchronize(self.device)
        torch.cuda.set_stream(self.stream)
        self.ctx = N.Context.get(device, self.stream.cuda_stream)
        self.lib = self.ctx.lib
        self.last_stats: Dict[str, Any] = {}

    # -- helpers ----------------------------------------------------------------------------------
    def _ptr(self, t):
        assert t.is_cuda and t.is_contiguous()
        return _P(t.data_ptr())

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def to_engine(self, array):
        """numpy / torch (any device) -> contiguous tensor on this engine's device."""
        torch = self.torch
        t = array if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))
        return t.to(self.device).contiguous()

    def to_numpy(self, t):
        return t.cpu().numpy()

    # -- compute ----------------------------------------------------------------------------------
    def matrix(self, rows, metric: str, rows_only: bool = False):
        N = self.N
        assert rows.dtype in (self.torch.float32, self.torch.float64)
        dt = np.float32 if rows.dtype == self.torch.float32 else np.float64
        # zero-copy: the matrix reads the tensor's HBM in place and keeps the tensor alive
        return N.DeviceMatrix(self.ctx, None, metric, device_ptr=rows.data_ptr(), shape=tuple(rows.shape), dtype=dt, borrow=True,
                              keepalive=rows, rows_only=rows_only)

    def knn(self, qm, q_begin: int, q_count: int, im, k: int, exclude_self: bool):
        torch, N = self.torch, self.N
        dist = self.empty((q_count, k), torch.float64)
        ind = self.empty((q_count, k), torch.int64)
        st = N.KnnStats()
        N._check(self.lib.kz_knn(self.ctx.handle, qm.handle, q_begin, q_count, im.handle, int(k), int(bool(exclude_self)),
                                 self._ptr(dist), self._ptr(ind), C.byref(st)), "kz_knn")
        self.last_stats = st.as_dict()
        return dist, ind

    def knn_dual(self, am, bm, k: int):
        """kz_knn_dual: ((dist, ind) of a -> b [a.n, k], (dist, ind) of b -> a [b.n, k]) from one sweep; stats of both in
        `last_stats` (a -> b) and `last_stats_reverse`."""
        torch, N = self.torch, self.N
        d_ab, i_ab = self.empty((am.shape[0], k), torch.float64), self.empty((am.shape[0], k), torch.int64)
        d_ba, i_ba = self.empty((bm.shape[0], k), torch.float64), self.empty((bm.shape[0], k), torch.int64)
        s_ab, s_ba = N.KnnStats(), N.KnnStats()
        N._check(self.lib.kz_knn_dual(self.ctx.handle, am.handle, bm.handle, int(k), self._ptr(d_ab), self._ptr(i_ab),
                                      self._ptr(d_ba), self._ptr(i_ba), C.byref(s_ab), C.byref(s_ba)), "kz_knn_dual")
        self.last_stats = s_ab.as_dict()
        self.last_stats_reverse = s_ba.as_dict()
        return (d_ab, i_ab), (d_ba, i_ba)

    def row_stats(self, dist, mean=False, std=False, last=False):
        torch = self.torch
        n, K = dist.shape
        m = self.empty((n,), torch.float64) if mean else None
        s = self.empty((n,), torch.float64) if std else None
        l_ = self.empty((n,), torch.float64) if last else None
        self.N._check(self.lib.kz_row_stats(self.ctx.handle, self._ptr(dist), n, K, self._ptr(m) if mean else None,
                                            self._ptr(s) if std else None, self._ptr(l_) if last else None), "kz_row_stats")
        return m, s, l_

    def csls(self, dist, ind, r_train):
        out = self.empty(tuple(dist.shape), self.torch.float64)
        self.N._check(self.lib.kz_csls(self.ctx.handle, self._ptr(dist), self._ptr(ind), dist.shape[0], dist.shape[1],
                                       self._ptr(r_train), self._ptr(out)), "kz_csls")
        return out

    def local_scaling(self, dist, ind, r_t, nicdm: bool):
        out = self.empty(tuple(dist.shape), self.torch.float64)
        self.N._check(self.lib.kz_local_scaling(self.ctx.handle, self._ptr(dist), self._ptr(ind), dist.shape[0], dist.shape[1],
                                                self._ptr(r_t), int(nicdm), self._ptr(out)), "kz_local_scaling")
        return out

    def mp_normal(self, dist, ind, mu_t, sd_t):
        out = self.empty(tuple(dist.shape), self.torch.float64)
        self.N._check(self.lib.kz_mp_normal(self.ctx.handle, self._ptr(dist), self._ptr(ind), dist.shape[0], dist.shape[1],
                                            self._ptr(mu_t), self._ptr(sd_t), self._ptr(out)), "kz_mp_normal")
        return out

    def mp_empiric(self, dist, ind, dist_t2s, ind_t2s):
        out = self.empty(tuple(dist.shape), self.torch.float64)
        self.N._check(self.lib.kz_mp_empiric(self.ctx.handle, self._ptr(dist), self._ptr(ind), dist.shape[0], dist.shape[1],
                                             self._ptr(dist_t2s), self._ptr(ind_t2s), dist_t2s.shape[0], dist_t2s.shape[1],
                                             self._ptr(out)), "kz_mp_empiric")
        return out

    def dsl_fit(self, ind_t2s, sm, tm, t_begin: int):
        out = self.empty((ind_t2s.shape[0],), self.torch.float64)
        self.N._check(self.lib.kz_dsl_fit(self.ctx.handle, self._ptr(ind_t2s), ind_t2s.shape[0], ind_t2s.shape[1], sm.handle,
                                          tm.handle, t_begin, self._ptr(out)), "kz_dsl_fit")
        return out

    def dsl_transform(self, ind, qm, q_begin: int, tm, t2c):
        torch = self.torch
        out = self.empty(tuple(ind.shape), torch.float64)
        gmin = torch.full((1,), float("inf"), dtype=torch.float64, device=self.device)
        self.N._check(self.lib.kz_dsl_transform(self.ctx.handle, self._ptr(ind), ind.shape[0], ind.shape[1], qm.handle, q_begin,
                                                tm.handle, self._ptr(t2c), self._ptr(out), self._ptr(gmin)), "kz_dsl_transform")
        return out, gmin

    def dsl_finalize(self, out, min_value: float, squared: bool):
        self.N._check(self.lib.kz_dsl_finalize(self.ctx.handle, self._ptr(out), out.numel(), float(min_value), int(squared)),
                      "kz_dsl_finalize")
        return out

    def select_topk(self, dist, ind, k: int):
        torch = self.torch
        od = self.empty((dist.shape[0], k), torch.float64)
        oi = self.empty((dist.shape[0], k), torch.int64)
        self.N._check(self.lib.kz_select_topk(self.ctx.handle, self._ptr(dist), self._ptr(ind), dist.shape[0], dist.shape[1], k,
                                              self._ptr(od), self._ptr(oi)), "kz_select_topk")
        return od, oi

    def pair_values(self, qm, q_begin: int, q_count: int, im, ind):
        """kz_pair_values: the exact float64 value the search ranks index row ind[r, c] by for query row q_begin + r."""
        out = self.empty(tuple(ind.shape), self.torch.float64)
        self.N._check(self.lib.kz_pair_values(self.ctx.handle, qm.handle, q_begin, q_count, im.handle, self._ptr(ind),
                                              ind.shape[1], self._ptr(out)), "kz_pair_values")
        return out

    MAX_MERGE = 8192   # KZ_MERGE_MAX_ENTRIES

    def merge_topk(self, key, ind, dist, segs: int, seg_len: int, k: int):
        """kz_merge_topk: per row the k smallest of `segs` sorted segments by (key, ind); returns (dist or key, ind)."""
        torch = self.torch
        n = key.shape[0]
        od = self.empty((n, k), torch.float64)
        oi = self.empty((n, k), torch.int64)
        self.N._check(self.lib.kz_merge_topk(self.ctx.handle, self._ptr(key), self._ptr(ind) if ind is not None else None,
                                             self._ptr(dist) if dist is not None else None, n, segs, seg_len, k,
                                             self._ptr(od), self._ptr(oi)), "kz_merge_topk")
        return od, oi

    def sync(self):
        self.torch.cuda.synchronize(self.device)


# ---------------------------------------------------------------------------------------------------
# communicator: thin wrapper over torch.distributed (RCCL on GPUs, gloo in CPU tests)
# ---------------------------------------------------------------------------------------------------
class Comm:
    """`time_collectives=True` brackets every collective with events on the current stream (GPU) or a wall clock (gloo) and
    accumulates milliseconds per kind: bench.py reports them per step so that a scaling run shows what the exchange costs."""

    def __init__(self, group=None, time_collectives: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # test hook: run the collectives even with one rank (exercises the RCCL calls on a single-GPU box)
        self.always = dist.is_initialized() and os.environ.get("KIEZ_AMD_FORCE_COLLECTIVES") == "1"
        self.timed = bool(time_collectives)
        self._events = []   # (kind, start event, end event) -- resolved lazily, no sync inside the step
        self._wall = {}
        self._bytes = {}    # kind -> [calls, payload bytes this rank handed to the collective]

    # -- timing ---------------------------------------------------------------------------------------
    def reset_timers(self):
        self._events = []
        self._wall = {}
        self._bytes = {}

    def traffic(self, steps: int = 1):
        """Per step and kind: number of collectives and payload bytes (this rank's send buffer; all_gather: its own block)."""
        return {k: {"calls": c / max(steps, 1), "bytes": b / max(steps, 1)} for k, (c, b) in self._bytes.items()}

    def _timed(self, kind, t, fn):
        rec = self._bytes.setdefault(kind, [0, 0])
        rec[0] += 1
        rec[1] += t.numel() * t.element_size()
        if not self.timed:
            return fn()
        if t.is_cuda:
            torch = _torch()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = fn()
            b.record()
            self._events.append((kind, a, b))
            return out
        import time
        t0 = time.perf_counter()
        out = fn()
        self._wall[kind] = self._wall.get(kind, 0.0) + (time.perf_counter() - t0) * 1e3
        return out

    def timers_ms(self, steps: int = 1):
        """Milliseconds per step spent in collectives, by kind (call after a device synchronisation)."""
        acc = dict(self._wall)
        for kind, a, b in self._events:
            acc[kind] = acc.get(kind, 0.0) + a.elapsed_time(b)
        return {k: v / max(steps, 1) for k, v in acc.items()}

    def broadcast(self, t, src=0):
        if self.world > 1 or self.always:
            self._timed("broadcast", t, lambda: self.dist.broadcast(t, src=src, group=self.group))
        return t

    def broadcast_begin(self, t, src=0):
        """Start the broadcast of `t` without making the current stream wait for it (RCCL runs it on its own stream); returns a
        token for `broadcast_end`.  What the caller enqueues in between -- work that does not touch `t` -- runs beside the
        transfer.  Two timers: "broadcast" covers begin .. end on the caller's stream (the transfer INCLUDING whatever the caller
        overlapped with it), "broadcast_exposed" only the wait inside `broadcast_end` (what the step pays for the exchange)."""
        if not (self.world > 1 or self.always):
            return None
        rec = self._bytes.setdefault("broadcast", [0, 0])
        rec[0] += 1
        rec[1] += t.numel() * t.element_size()
        start = None
        if self.timed and t.is_cuda:
            torch = _torch()
            start = torch.cuda.Event(enable_timing=True)
            start.record()
        import time
        t0 = time.perf_counter()
        work = self.dist.broadcast(t, src=src, group=self.group, async_op=True)
        return (work, start, t0, t.is_cuda)

    def broadcast_end(self, token):
        if token is None:
            return
        work, start, t0, on_gpu = token
        import time
        if not self.timed:
            work.wait()   # (GPU: the current stream waits for RCCL's stream; the host does not block)
            return
        if on_gpu:
            torch = _torch()
            mid, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            mid.record()
            work.wait()
            end.record()
            self._events.append(("broadcast", start, end))
            self._events.append(("broadcast_exposed", mid, end))
        else:
            t1 = time.perf_counter()
            work.wait()
            t2 = time.perf_counter()
            self._wall["broadcast"] = self._wall.get("broadcast", 0.0) + (t2 - t0) * 1e3
            self._wall["broadcast_exposed"] = self._wall.get("broadcast_exposed", 0.0) + (t2 - t1) * 1e3

    def all_gather_rows(self, t, counts):
        """Concatenate ragged row blocks [n_r, ...] of all ranks in rank order (padded all_gather)."""
        torch = _torch()
        if self.world == 1 and not self.always:
            return t
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        if t.is_cuda:
            out = torch.empty((self.world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self._timed("all_gather", pad, lambda: self.dist.all_gather_into_tensor(out, pad, group=self.group))
            if all(c == mx for c in counts):
                return out
            return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(self.world)], dim=0)
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        self._timed("all_gather", pad, lambda: self.dist.all_gather(parts, pad, group=self.group))
        return torch.cat([parts[r][: counts[r]] for r in range(self.world)], dim=0)

    def gather_rows_to0(self, t, counts):
        """Ragged row blocks of all ranks concatenated in rank order ON RANK 0 ONLY (None elsewhere): for evidence that one rank
        evaluates (bench.py's oracle check and CPU baseline gather the source shards) -- an all_gather would put world x shard on
        every GPU."""
        torch = _torch()
        if self.world == 1 and not self.always:
            return t
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == 0 else None
        self._timed("gather", pad, lambda: self.dist.gather(pad, gather_list=parts, dst=0, group=self.group))
        if self.rank != 0:
            return None
        return torch.cat([parts[r][: counts[r]] for r in range(self.world)], dim=0)

    def all_to_all_rows(self, t, counts):
        """Row blocks of `t` (block r = counts[r] rows, in rank order) go to rank r; returns what this rank received, stacked
        in rank order: [world, counts[rank], ...].  The exchange step of the shared sweep: every rank holds, for ALL
        target rows, their neighbours inside its own source shard; rank r merges the rows of its target slice."""
        torch = _torch()
        mine = counts[self.rank]
        if self.world == 1 and not self.always:
            return t.reshape((1,) + tuple(t.shape))
        out = torch.empty((self.world * mine,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = t.contiguous()
        # one path on every rank and every backend (RCCL and gloo both implement all_to_all_single); an error of the
        # collective propagates -- a rank-local fallback to other collectives would leave the ranks in different calls
        self._timed("all_to_all", t, lambda: self.dist.all_to_all_single(out, t, output_split_sizes=[mine] * self.world,
                                                                        input_split_sizes=list(counts), group=self.group))
        return out.reshape((self.world, mine) + tuple(t.shape[1:]))

    def all_gather_ints(self, value: int, device):
        return [v[0] for v in self.all_gather_vec([int(value)], device)]

    def all_gather_vec(self, values, device):
        """All-gather a short int vector per rank: ONE collective and ONE device->host sync for all the sizes a fit needs
        (shard row counts and, from rank 0, the target's shape) instead of one per quantity."""
        torch = _torch()
        values = [int(v) for v in values]
        if self.world == 1 and not self.always:
            return [values]
        mine = torch.tensor(values, dtype=torch.int64, device=device)
        out = torch.empty((self.world, len(values)), dtype=torch.int64, device=device)
        if mine.is_cuda:
            self.dist.all_gather_into_tensor(out, mine, group=self.group)
        else:
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(parts, mine, group=self.group)
            out = torch.stack(parts)
        return out.cpu().tolist()

    def all_reduce_min(self, t):
        if self.world > 1 or self.always:
            self._timed("all_reduce", t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group))
        return t


class RcclComm(Comm):
    """The same exchange steps through the C ABI's own collectives (`kz_comm_*`, kiez_amd/csrc/kz_comm.hip: RCCL on the context's
    stream) instead of `torch.distributed`: what a host WITHOUT torch binds (INTEGRATION.md "multi-GPU").  Here the buffers are still
    the engine's tensors -- only their device pointers cross the ABI -- so the sharded pipeline above runs unchanged on it.

    The 128-byte unique id is the host's to distribute: `RcclComm.unique_id()` on rank 0, then any channel (a file, MPI, a socket);
    `RcclComm.from_file(engine, rank, world, path)` is the file rendezvous (rank 0 writes, the others poll)."""

    def __init__(self, engine, rank: int, world: int, unique_id: bytes, always: bool = False, time_collectives: bool = False):
        from . import _native as N
        if len(unique_id) != 128:
            raise ValueError("unique_id: the 128 bytes of RcclComm.unique_id() on rank 0")
        self.N = N
        self.engine = engine
        self.lib = engine.ctx.lib
        self.rank, self.world = int(rank), int(world)
        self.always = bool(always) or os.environ.get("KIEZ_AMD_FORCE_COLLECTIVES") == "1"
        self.group = None
        self.timed = bool(time_collectives)
        self._events, self._wall, self._bytes = [], {}, {}
        h = _P()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        N._check(self.lib.kz_comm_create(engine.ctx.handle, buf, self.rank, self.world, C.byref(h)), "kz_comm_create")
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        from . import _native as N
        buf = C.create_string_buffer(128)
        N._check(N.load().kz_comm_unique_id(buf), "kz_comm_unique_id")
        return buf.raw

    @classmethod
    def from_file(cls, engine, rank: int, world: int, path: str, timeout_s: float = 120.0, **kw):
        import time
        t0 = time.time()
        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as fh:
                fh.write(cls.unique_id())
            os.replace(tmp, path)     # (atomic: a reader never sees a partial id)

        def fresh():
            # (a file left behind by an EARLIER job at the same path is older than this process' call: never taken for this job's id;
            #  the ranks of one job call this within `stale_s` of each other)
            try:
                return os.path.getmtime(path) >= t0 - float(kw.get("stale_s", 30.0))
            except OSError:
                return False
        while not fresh():
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"RcclComm.from_file: no fresh {path} within {timeout_s} s")
            time.sleep(0.01)
        kw.pop("stale_s", None)
        with open(path, "rb") as fh:
            uid = fh.read()
        return cls(engine, rank, world, uid, **kw)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.kz_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):   # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _p(t):
        assert t.is_cuda and t.is_contiguous()
        return _P(t.data_ptr())

    @staticmethod
    def _nbytes(t):
        return t.numel() * t.element_size()

    # -- the collectives (same contracts as Comm's) -------------------------------------------------------------------------
    def broadcast(self, t, src=0):
        if self.world > 1 or self.always:
            self._timed("broadcast", t, lambda: self.N._check(self.lib.kz_comm_broadcast(self.handle, self._p(t), self._nbytes(t), int(src)),
                                                               "kz_comm_broadcast"))
        return t

    def broadcast_begin(self, t, src=0):
        # (kz_comm_* enqueue on the context's stream: the transfer is ordered like a kernel, there is nothing to overlap with on
        #  that stream -- begin runs it, end is a no-op; "broadcast_exposed" = the whole transfer)
        if not (self.world > 1 or self.always):
            return None
        self._timed("broadcast_exposed", t, lambda: self.N._check(self.lib.kz_comm_broadcast(self.handle, self._p(t), self._nbytes(t), int(src)),
                                                                   "kz_comm_broadcast"))
        rec = self._bytes.pop("broadcast_exposed")
        tot = self._bytes.setdefault("broadcast", [0, 0])
        tot[0] += rec[0]
        tot[1] += rec[1]
        return ("done",)

    def broadcast_end(self, token):
        return None

    def all_gather_rows(self, t, counts):
        torch = _torch()
        if self.world == 1 and not self.always:
            return t
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        out = torch.empty((self.world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._timed("all_gather", pad, lambda: self.N._check(self.lib.kz_comm_all_gather(self.handle, self._p(pad), self._p(out), self._nbytes(pad)),
                                                              "kz_comm_all_gather"))
        if all(c == mx for c in counts):
            return out
        return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(self.world)], dim=0)

    def gather_rows_to0(self, t, counts):
        out = self.all_gather_rows(t, counts)     # (no gather primitive in the ABI: evidence paths only)
        return out if self.rank == 0 else None

    def all_to_all_rows(self, t, counts):
        torch = _torch()
        mine = counts[self.rank]
        if self.world == 1 and not self.always:
            return t.reshape((1,) + tuple(t.shape))
        t = t.contiguous()
        row_bytes = self._nbytes(t) // max(t.shape[0], 1)
        out = torch.empty((self.world * mine,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        offs = (C.c_size_t * self.world)(*[sum(counts[:r]) * row_bytes for r in range(self.world)])
        lens = (C.c_size_t * self.world)(*[counts[r] * row_bytes for r in range(self.world)])
        self._timed("all_to_all", t, lambda: self.N._check(self.lib.kz_comm_all_to_all(self.handle, self._p(t), offs, lens, self._p(out), mine * row_bytes),
                                                            "kz_comm_all_to_all"))
        return out.reshape((self.world, mine) + tuple(t.shape[1:]))

    def all_gather_vec(self, values, device):
        torch = _torch()
        values = [int(v) for v in values]
        if self.world == 1 and not self.always:
            return [values]
        mine = torch.tensor(values, dtype=torch.int64, device=device)
        out = torch.empty((self.world, len(values)), dtype=torch.int64, device=device)
        self.N._check(self.lib.kz_comm_all_gather(self.handle, self._p(mine), self._p(out), self._nbytes(mine)), "kz_comm_all_gather")
        return out.cpu().tolist()

    def all_reduce_min(self, t):
        if self.world > 1 or self.always:
            torch = _torch()
            assert t.dtype == torch.float64
            self._timed("all_reduce", t, lambda: self.N._check(self.lib.kz_comm_all_reduce_min_f64(self.handle, self._p(t), t.numel()),
                                                                "kz_comm_all_reduce_min_f64"))
        return t


def row_slice(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous balanced partition of n rows: (begin, count) of `rank`."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------------
# the sharded pipeline
# ---------------------------------------------------------------------------------------------------
_HUB = {None: "none", "no": "none", "nohubnessreduction": "none", "csls": "csls", "localscaling": "ls", "ls": "ls",
        "mutualproximity": "mp", "mp": "mp", "dissimlocal": "dsl", "dsl": "dsl"}


class ShardedKiez:
    """`Kiez` over a row-sharded source (one instance per rank).

    fit(source_shard, target): `target` is needed on rank 0 only (other ranks may pass None when `target_from_rank0`);
    pass `single_source=True` for the reference's `fit(source)` mode (target == full source).
    kneighbors(k) -> (dist, ind) of the local shard rows as engine tensors.
    """

    def __init__(self, n_candidates: int = 10, algorithm_kwargs: Optional[Dict[str, Any]] = None, hubness=None,
                 hubness_kwargs: Optional[Dict[str, Any]] = None, engine=None, comm: Optional[Comm] = None,
                 cache_target: bool = False):
        if not np.issubdtype(type(n_candidates), np.integer):
            raise TypeError(f"n_neighbors does not take {type(n_candidates)} value, enter integer value")
        if n_candidates <= 0:
            raise ValueError(f"Expected n_candidates > 0. Got {n_candidates}")
        if n_candidates == 1:
            raise ValueError("Cannot perform hubness reduction with a single candidate per query!")
        akw = dict(algorithm_kwargs or {})
        self.K = int(akw.get("n_candidates", n_candidates))
        self.metric = canonical_metric(akw.get("metric", "minkowski"), akw.get("p", 2))
        key = hubness.lower() if isinstance(hubness, str) else hubness
        if key not in _HUB:
            raise KeyError(f"Invalid hubness reduction: {hubness}")
        self.hub = _HUB[key]
        hkw = dict(hubness_kwargs or {})
        self.method = str(hkw.get("method", "standard" if self.hub == "ls" else "normal")).lower()
        if self.hub == "ls" and self.method not in ("ls", "standard", "nicdm"):
            raise ValueError(f"Internal: Invalid method {self.method}. Try 'ls' or 'nicdm'.")
        if self.hub == "mp":
            if self.method not in ("exact", "empiric", "normal", "gaussi"):
                raise ValueError(f'Mutual proximity method "{self.method}" not recognized. Try "normal" or "empiric".')
            self.method = "empiric" if self.method in ("exact", "empiric") else "normal"
        if self.hub == "dsl" and self.metric not in ("euclidean", "sqeuclidean"):
            raise ValueError(f"DisSimLocal only supports squared Euclidean distances, not metric={akw.get('metric')}.")
        self.engine = engine if engine is not None else HipEngine()
        self.comm = comm if comm is not None else Comm()
        self.state: Dict[str, Any] = {}
        self.shared_sweep = bool(hkw.get("shared_sweep", True))   # False: search twice, as the reference does
        # cache_target (opt-in): a fit whose target on rank 0 is THE SAME engine tensor as in the previous fit -- same object, same
        # storage, same torch version counter (no in-place write since) -- reuses the replica every rank received then and runs no
        # broadcast: the serving pattern (one index, many query batches) pays the 0.8 - 1.2 GB transfer once, not per fit.  The
        # identity travels in the size exchange every fit runs anyway, so all ranks decide alike.  Off by default: a target that is
        # rewritten through a raw pointer (not through torch) keeps its version counter.
        self.cache_target = bool(cache_target)
        self._tgt_identity = None      # identity (as gathered from rank 0) of the target the last broadcast delivered
        self._tgt_replica = None       # ... and the tensor it landed in (rank 0: the caller's tensor, kept alive)

    # -- fit -----------------------------------------------------------------------------------------
    def fit(self, source_shard, target=None, single_source: bool = False, target_from_rank0: bool = True):
        eng, comm = self.engine, self.comm
        src = eng.to_engine(source_shard)
        if src.dim() != 2:
            raise ValueError("Expected 2D array")
        # one collective for every size this fit needs: shard rows of all ranks + (from rank 0) the target's shape
        torch = _torch()
        tgt0 = None
        meta = [src.shape[0], 0, 0, 0, 0, 0, 0]
        multi = comm.world > 1 or comm.always
        bcast_target = (not single_source) and target_from_rank0 and multi
        if bcast_target and comm.rank == 0:
            tgt0 = eng.to_engine(target)
            meta[1:4] = [tgt0.shape[0], tgt0.shape[1], 0 if tgt0.dtype == torch.float32 else 1]
            if self.cache_target and tgt0 is target:
                # (only a tensor the CALLER holds on the engine's device can be recognised again: a host array is copied into a new
                #  tensor by every fit.  63-bit pieces: the vector travels as int64)
                meta[4:] = [id(tgt0) & 0x7fffffffffffffff, tgt0.data_ptr() & 0x7fffffffffffffff, int(getattr(tgt0, "_version", 0)) + 1]
        gathered = comm.all_gather_vec(meta, src.device)
        counts = [g[0] for g in gathered]
        self.counts = counts
        self.s_begin = sum(counts[: comm.rank])
        self.n_local = counts[comm.rank]
        n_s = sum(counts)
        self.single = bool(single_source)
        self._fwd = None
        bcast = None
        # the target's row count is known BEFORE anything is decided (every rank decides the same way)
        tgt = None
        if self.single:
            n_t = n_s
        elif bcast_target:
            n_t = gathered[0][1]
        else:
            tgt = eng.to_engine(target)
            n_t = tgt.shape[0]
        # Shared sweep (kz_knn_dual): the reverse neighbours of ALL targets inside this rank's source shard and the forward
        # neighbours of the shard come out of one sweep of shard x target; the per-shard reverse lists of a target are then
        # merged on the rank that owns its slice (all-to-all + kz_merge_topk).  Every hubness kind qualifies: the kinds
        # that only need the reverse DISTANCES (CSLS, LocalScaling, MP normal) exchange the distances, the kinds that need
        # the reverse INDICES in the single-GPU order (MP empiric, DSL) exchange exact ordering values + global ids too.
        # Every shard must hold K rows (its own search returns K) and the target side K rows (the forward search).
        self.shared = (self.shared_sweep and hasattr(eng, "knn_dual") and not self.single and self.hub != "none"
                       and self.K <= min(counts) and self.K <= n_t
                       and (not multi or comm.world * self.K <= getattr(eng, "MAX_MERGE", 8192)))
        need_full_source = self.single or (self.hub != "none" and not self.shared)
        src_full = comm.all_gather_rows(src, counts) if need_full_source else None
        if self.single:
            tgt = src_full
        else:
            if bcast_target:
                d, code = gathered[0][2], gathered[0][3]
                # (shape and dtype are checked on the gathered sizes BEFORE the transfer starts: an error raised past
                #  broadcast_begin would leave an RCCL work nobody waits for -- and every rank raises the same way)
                if d != src.shape[1]:
                    raise ValueError("Expected source and target to have the same number of features,"
                                     f" but got source.shape: {tuple(src.shape)} and target.shape: {(n_t, d)}")
                if (torch.float32 if code == 0 else torch.float64) != src.dtype:
                    raise ValueError("source and target must have the same dtype")
                identity = tuple(gathered[0][1:7])
                cached = (self.cache_target and identity[5] != 0 and identity == self._tgt_identity and self._tgt_replica is not None)
                if cached:
                    tgt = self._tgt_replica          # (every rank received exactly this target in the previous fit: no transfer)
                else:
                    tgt = tgt0 if comm.rank == 0 else eng.empty((n_t, d), torch.float32 if code == 0 else torch.float64)
                self._tgt_identity, self._tgt_replica = (identity, tgt) if self.cache_target and identity[5] != 0 else (None, None)
                # RCCL broadcast of the replicated target over xGMI, started here and awaited only where the target is first
                # needed: the source shard's own preparation (norms of its rows) runs beside the transfer.  (The sweep itself
                # cannot start on a part of the target: the shared sweep's event thresholds come from a sample that spans all its
                # rows, and its fp16 image is scaled by the largest centred norm of both matrices -- DESIGN.md section 6.)
                bcast = None if cached else comm.broadcast_begin(tgt, 0)
            if tgt.shape[1] != src.shape[1]:
                raise ValueError("Expected source and target to have the same number of features,"
                                 f" but got source.shape: {tuple(src.shape)} and target.shape: {tuple(tgt.shape)}")
            if tgt.dtype != src.dtype:
                raise ValueError("source and target must have the same dtype")
        self._keep = (src, src_full, tgt)  # the engine matrices borrow these tensors
        self.n_s, self.n_t = n_s, tgt.shape[0]
        S_own = None
        try:
            if not need_full_source:
                S_own = eng.matrix(src, self.metric)     # (beside the broadcast: touches the shard only)
        finally:
            comm.broadcast_end(bcast)                    # (also on an error: the transfer is always awaited)
        self.T = eng.matrix(tgt, self.metric)
        if need_full_source:
            self.S = self.T if self.single else eng.matrix(src_full, self.metric)
            self.q_begin = self.s_begin   # forward queries are a row range of the full source matrix
        else:
            self.S = S_own
            self.q_begin = 0
        if self.hub == "none":
            return self
        t_begin, t_count = row_slice(self.n_t, comm.rank, comm.world)
        t_counts = [row_slice(self.n_t, r, comm.world)[1] for r in range(comm.world)]
        needs_ind = self.hub == "dsl" or (self.hub == "mp" and self.method == "empiric")
        S_fit = self.S      # the matrix DSL's centroids gather source rows from (global ids)
        if self.shared:
            # one sweep: larger side as the query side (fewer rows get event buffers)
            if self.n_t >= self.n_local:
                (d_rev, i_rev), self._fwd = eng.knn_dual(self.T, self.S, self.K)
            else:
                self._fwd, (d_rev, i_rev) = eng.knn_dual(self.S, self.T, self.K)
            if multi:
                W, K = comm.world, self.K
                if needs_ind:
                    # distances + global ids, and -- where the OUTPUT distance is a rounded function of the value the search
                    # ranked by (euclidean: sqrt, for float32 inputs through float32) -- the exact ordering values too: merging
                    # by rounded distances would break ties differently from one GPU.  ONE all-to-all either way.
                    gids = (i_rev + self.s_begin).contiguous()
                    lossless = self.metric in ("sqeuclidean", "cosine", "manhattan", "chebyshev")   # the returned distance IS the ordering value
                    planes = [d_rev, gids.view(torch.float64)] if lossless else \
                        [eng.pair_values(self.T, 0, self.n_t, self.S, i_rev), d_rev, gids.view(torch.float64)]
                    parts = comm.all_to_all_rows(torch.stack(planes, dim=1), t_counts)          # [W, t_count, 2 or 3, K]

                    def seg(c):
                        return parts[:, :, c].permute(1, 0, 2).reshape(t_count, W * K).contiguous()
                    if lossless:
                        d_t2s, i_t2s = eng.merge_topk(seg(0), seg(1).view(torch.int64), None, W, K, K)
                    else:
                        d_t2s, i_t2s = eng.merge_topk(seg(0), seg(2).view(torch.int64), seg(1), W, K, K)
                else:
                    parts = comm.all_to_all_rows(d_rev, t_counts)                               # [W, t_count, K]
                    merged = parts.permute(1, 0, 2).reshape(t_count, W * K).contiguous()
                    d_t2s, i_t2s = eng.merge_topk(merged, None, None, W, K, K)                  # (ids unused by these kinds)
                if self.hub == "dsl":
                    # the centroids average SOURCE ROWS of all shards (dis_sim.py:96-101): the shards are gathered after all,
                    # but only for this gather kernel -- the second sweep stays saved
                    src_full = comm.all_gather_rows(src, counts)
                    self._keep = self._keep + (src_full,)
                    # (a row source only: no norms, no operand images of the gathered rows)
                    S_fit = eng.matrix(src_full, self.metric, rows_only=True)
            else:
                d_t2s, i_t2s = d_rev, i_rev
        else:
            # reverse pass, sharded over target rows (explicit query: self is NOT stripped, base.py:37-42)
            Kr = min(self.K, n_s)
            d_t2s, i_t2s = eng.knn(self.T, t_begin, t_count, self.S, Kr, False)
        st = self.state
        if self.hub == "csls" or (self.hub == "ls" and self.method == "nicdm"):
            m, _, _ = eng.row_stats(d_t2s, mean=True)
            st["r_t"] = comm.all_gather_rows(m, t_counts)
        elif self.hub == "ls":
            _, _, last = eng.row_stats(d_t2s, last=True)
            st["r_t"] = comm.all_gather_rows(last, t_counts)
        elif self.hub == "mp" and self.method == "normal":
            m, s, _ = eng.row_stats(d_t2s, mean=True, std=True)
            st["mu_t"] = comm.all_gather_rows(m, t_counts)
            st["sd_t"] = comm.all_gather_rows(s, t_counts)
        elif self.hub == "mp":
            st["dist_t2s"] = comm.all_gather_rows(d_t2s, t_counts)
            st["ind_t2s"] = comm.all_gather_rows(i_t2s, t_counts)
        elif self.hub == "dsl":
            # (single rank without forced collectives: d_t2s covers all targets and t_begin = 0, t_count = n_t)
            t2c = eng.dsl_fit(i_t2s, S_fit, self.T, t_begin)
            st["t2c"] = comm.all_gather_rows(t2c, t_counts)
        return self

    # -- kneighbors ------------------------------------------------------------------------------------
    def kneighbors(self, k: Optional[int] = None):
        eng, comm, st = self.engine, self.comm, self.state
        if k is None or k > self.K:
            k = self.K
        if self.hub == "none":
            kk = min(k, self.n_t)
            return eng.knn(self.S, self.q_begin, self.n_local, self.T, kk, self.single)
        Kf = min(self.K, self.n_t)
        if self._fwd is not None:
            dist, ind = self._fwd   # came out of fit's sweep
        else:
            dist, ind = eng.knn(self.S, self.q_begin, self.n_local, self.T, Kf, self.single)
        if self.hub == "csls":
            out = eng.csls(dist, ind, st["r_t"])
        elif self.hub == "ls":
            out = eng.local_scaling(dist, ind, st["r_t"], self.method == "nicdm")
        elif self.hub == "mp" and self.method == "normal":
            out = eng.mp_normal(dist, ind, st["mu_t"], st["sd_t"])
        elif self.hub == "mp":
            out = eng.mp_empiric(dist, ind, st["dist_t2s"], st["ind_t2s"])
        else:
            out, gmin = eng.dsl_transform(ind, self.S, self.q_begin, self.T, st["t2c"])
            comm.all_reduce_min(gmin)  # the shift uses the GLOBAL minimum (dis_sim.py:171-173)
            out = eng.dsl_finalize(out, float(gmin.cpu()[0]), self.metric == "sqeuclidean")
        return eng.select_topk(out, ind, min(k, Kf))
