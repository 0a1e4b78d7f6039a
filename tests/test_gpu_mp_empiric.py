"""kz_mp_empiric (the candidate ids of a query in a hash table in LDS, the reverse-list ids looked up in it) against the oracle
(mutual_proximity.py:185-212 restated in oracle/kiez_oracle.py) on list shapes the golden cases do not have: one and two entries
per lane on either side, reverse lists from a small id range (many matches, many hash collisions), ids beyond 32 bits and negative
ids (never equal to a candidate id, whatever their low bits are)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lists(rng, n, K, n_ids):
    """n rows of K distinct ids below n_ids with ascending distances (what a kNN search returns)."""
    ind = np.stack([rng.choice(n_ids, K, replace=False) for _ in range(n)]).astype(np.int64)
    dist = np.sort(rng.random((n, K)), axis=1)
    return dist, ind


def _run(dist, ind, dist_t2s, ind_t2s):
    from kiez_amd import _native as N
    ctx = N.Context.get()
    dev = [ctx.empty(a.shape, a.dtype) for a in (dist, ind, dist_t2s, ind_t2s)]
    for d, a in zip(dev, (dist, ind, dist_t2s, ind_t2s)):
        d.fill_from_host(a)
    out = ctx.empty(dist.shape, np.float64)
    N._check(ctx.lib.kz_mp_empiric(ctx.handle, dev[0].ptr, dev[1].ptr, dist.shape[0], dist.shape[1], dev[2].ptr, dev[3].ptr,
                                   dist_t2s.shape[0], dist_t2s.shape[1], out.ptr), "kz_mp_empiric")
    return out.numpy()


@pytest.mark.parametrize("n,K,n_t,Kt", [(3000, 20, 2500, 20), (700, 50, 900, 50), (5000, 7, 300, 13), (1000, 100, 400, 128), (400, 65, 80, 3),
                                        (120, 10, 90, 10)])
def test_values_are_the_oracles(n, K, n_t, Kt):
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(n + K)
    # candidates: target ids; reverse lists: ids of the other side, drawn from a small range so that matches are common
    dist, ind = _lists(rng, n, K, n_t)
    dist_t2s, ind_t2s = _lists(rng, n_t, Kt, max(n_t, Kt + 5))
    got = _run(dist, ind, dist_t2s, ind_t2s)
    want = O.mp_empiric_transform(dist, ind, dist_t2s, ind_t2s)
    np.testing.assert_array_equal(got, want)


def test_ids_beyond_32_bits_and_negative_ids():
    from oracle import kiez_oracle as O
    rng = np.random.default_rng(9)
    n, K, n_t, Kt = 2000, 16, 1500, 16
    dist, ind = _lists(rng, n, K, n_t)
    dist_t2s, ind_t2s = _lists(rng, n_t, Kt, n_t)
    ind_t2s[7, 3] = (1 << 40) + 5        # never equal to a candidate id
    ind_t2s[900, 0] = -3
    got = _run(dist, ind, dist_t2s, ind_t2s)
    want = O.mp_empiric_transform(dist, ind, dist_t2s, ind_t2s)
    np.testing.assert_array_equal(got, want)
    # a low 32-bit alias of a huge id must not match either: 2^32 + c for a candidate id c of the rows that look row 11 up
    ind_t2s[11, 2] = (1 << 32) + int(ind[0, 0])
    got = _run(dist, ind, dist_t2s, ind_t2s)
    np.testing.assert_array_equal(got, O.mp_empiric_transform(dist, ind, dist_t2s, ind_t2s))
