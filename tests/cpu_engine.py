"""A CPU stand-in for kiez_amd.distributed.HipEngine, backed by the oracle (TEST INFRASTRUCTURE).

It lets the sharding / exchange logic of ShardedKiez run under gloo with world_size 2 in a container without a GPU.
Tensors are CPU torch tensors; every method has the signature of the HIP engine."""
import numpy as np
import torch

from oracle import kiez_oracle as O


class _Mat:
    def __init__(self, rows, metric):
        self.rows = rows.numpy()
        self.metric = metric
        self.shape = tuple(self.rows.shape)   # (as kiez_amd._native.DeviceMatrix)


class OracleEngine:
    def __init__(self):
        self.device = torch.device("cpu")
        self.last_stats = {}

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def to_engine(self, array):
        t = array if isinstance(array, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(array))
        return t.contiguous()

    def to_numpy(self, t):
        return t.numpy()

    def matrix(self, rows, metric, rows_only=False):
        return _Mat(rows, metric)

    def knn(self, qm, q_begin, q_count, im, k, exclude_self):
        q = qm.rows[q_begin:q_begin + q_count]
        if exclude_self:
            # rows of a range query: strip by GLOBAL row id -> run k+1 and drop like sklearn does
            d, i = O.knn_exact(q, im.rows, k + 1, im.metric)
            rid = np.arange(q_begin, q_begin + q_count)[:, None]
            mask = i != rid
            none_self = mask.all(axis=1)
            mask[none_self, 0] = False
            d, i = d[mask].reshape(q_count, k), i[mask].reshape(q_count, k)
        else:
            d, i = O.knn_exact(q, im.rows, k, im.metric)
        return torch.from_numpy(d), torch.from_numpy(i)

    def knn_dual(self, am, bm, k):
        """Both directions (the HIP engine takes them out of one sweep; the results are those of two searches)."""
        class _M:   # knn() only reads .rows / .metric
            pass
        ab = self.knn(am, 0, am.rows.shape[0], bm, k, False)
        ba = self.knn(bm, 0, bm.rows.shape[0], am, k, False)
        return ab, ba

    def row_stats(self, dist, mean=False, std=False, last=False):
        a = dist.numpy()
        m = torch.from_numpy(a.mean(axis=1)) if mean else None
        s = torch.from_numpy(np.nanstd(a, axis=1)) if std else None
        l_ = torch.from_numpy(a[:, -1].copy()) if last else None
        return m, s, l_

    def csls(self, dist, ind, r_train):
        d, i = dist.numpy(), ind.numpy()
        return torch.from_numpy(2 * d - d.mean(axis=1).reshape(-1, 1) - r_train.numpy()[i])

    def local_scaling(self, dist, ind, r_t, nicdm):
        d, i, r = dist.numpy(), ind.numpy(), r_t.numpy()
        if nicdm:
            out = d / np.sqrt(d.mean(axis=1).reshape(-1, 1) * r[i])
        else:
            out = 1.0 - np.exp(-1 * d**2 / (d[:, -1].reshape(-1, 1) * r[i]))
        return torch.from_numpy(out)

    def mp_normal(self, dist, ind, mu_t, sd_t):
        d, i = dist.numpy(), ind.numpy()
        mu, sd = np.nanmean(d, axis=1).reshape(-1, 1), np.nanstd(d, axis=1).reshape(-1, 1)
        p1 = O._norm_sf(d, mu, sd)
        p2 = O._norm_sf(d, mu_t.numpy()[i], sd_t.numpy()[i])
        return torch.from_numpy(1 - p1 * p2)

    def mp_empiric(self, dist, ind, dist_t2s, ind_t2s):
        return torch.from_numpy(O.mp_empiric_transform(dist.numpy(), ind.numpy(), dist_t2s.numpy(), ind_t2s.numpy()))

    def dsl_fit(self, ind_t2s, sm, tm, t_begin):
        n = ind_t2s.shape[0]
        s64, t64 = sm.rows.astype(np.float64), tm.rows.astype(np.float64)
        return torch.from_numpy(O.dsl_fit(ind_t2s.numpy(), s64, t64[t_begin:t_begin + n]))

    def dsl_transform(self, ind, qm, q_begin, tm, t2c):
        i = ind.numpy()
        q = qm.rows[q_begin:q_begin + i.shape[0]].astype(np.float64)
        t = tm.rows.astype(np.float64)
        nb = t[i]
        diff = q[:, None, :] - nb
        hub = np.einsum("nkd,nkd->nk", diff, diff)
        s2c = ((q - nb.mean(axis=1)) ** 2).sum(axis=1)
        hub = hub - s2c.reshape(-1, 1) - t2c.numpy()[i]
        return torch.from_numpy(hub), torch.tensor([hub.min()], dtype=torch.float64)

    def dsl_finalize(self, out, min_value, squared):
        o = out.numpy()
        if min_value < 0:
            o = o + (-min_value)
        if not squared:
            o = o ** 0.5
        return torch.from_numpy(o)

    def select_topk(self, dist, ind, k):
        d, i = O.sort_topk(dist.numpy(), ind.numpy(), k)
        return torch.from_numpy(np.ascontiguousarray(d)), torch.from_numpy(np.ascontiguousarray(i))

    def pair_values(self, qm, q_begin, q_count, im, ind):
        """Exact float64 ordering value of (query row, index row) pairs, the expression of O.knn_exact."""
        i = ind.numpy()
        mc = O.canonical_metric(im.metric)
        if mc in ("manhattan", "chebyshev") or mc.startswith("minkowski["):
            qr = qm.rows[q_begin:q_begin + q_count]
            v = np.stack([O.minkowski_family_rdist(qr[r:r + 1], im.rows[i[r]], mc)[0] for r in range(len(qr))])
            return torch.from_numpy(v)
        q = qm.rows[q_begin:q_begin + q_count].astype(np.float64)
        y = im.rows.astype(np.float64)
        if O.canonical_metric(im.metric) == "cosine":
            qn, yn = np.sqrt((q * q).sum(1)), np.sqrt((y * y).sum(1))
            qn[qn == 0] = 1.0
            yn[yn == 0] = 1.0
            v = np.clip(1.0 - np.einsum("nd,nkd->nk", q / qn[:, None], (y / yn[:, None])[i]), 0.0, 2.0)
        else:
            v = np.maximum((q * q).sum(1)[:, None] - 2.0 * np.einsum("nd,nkd->nk", q, y[i]) + (y * y).sum(1)[i], 0.0)
        return torch.from_numpy(v)

    MAX_MERGE = 8192

    def merge_topk(self, key, ind, dist, segs, seg_len, k):
        """k smallest per row by (key, ind), ties by column (= segment, position): what kz_merge_topk does."""
        kk = key.numpy()
        ii = ind.numpy() if ind is not None else np.zeros(kk.shape, dtype=np.int64)
        dd = dist.numpy() if dist is not None else kk
        assert kk.shape[1] == segs * seg_len
        for s_ in range(segs):   # the contract: every segment sorted by (key, ind)
            a = slice(s_ * seg_len, (s_ + 1) * seg_len)
            assert ((kk[:, a][:, 1:] > kk[:, a][:, :-1]) | ((kk[:, a][:, 1:] == kk[:, a][:, :-1]) & (ii[:, a][:, 1:] >= ii[:, a][:, :-1]))).all()
        order = np.lexsort((ii, kk), axis=1)[:, :k]
        return (torch.from_numpy(np.ascontiguousarray(np.take_along_axis(dd, order, axis=1))),
                torch.from_numpy(np.ascontiguousarray(np.take_along_axis(ii, order, axis=1))))

    def sync(self):
        pass
