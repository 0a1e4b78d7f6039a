"""Test infrastructure: `Comm` whose collectives run over gloo on HOST copies of the engine's CUDA tensors.

RCCL refuses two ranks on one device, so tests that put several ranks of the product engine (`HipEngine`) on the one GPU of
the box route the collectives of `ShardedKiez` through this class: the C-ABI kernels see real multi-shard data, the
exchange itself runs over gloo (the RCCL collectives are covered by tests/test_gpu_sharded_rccl.py and
tests/test_gpu_northstar.py with one rank and every collective forced)."""
import torch

from kiez_amd.distributed import Comm


class StagedComm(Comm):
    def broadcast(self, t, src=0):
        h = t.cpu()
        self.dist.broadcast(h, src=src)
        t.copy_(h)
        return t

    def broadcast_begin(self, t, src=0):
        self.broadcast(t, src)      # staged: nothing to overlap
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
