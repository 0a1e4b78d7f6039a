"""Host logic of kz_knn that needs no GPU: the work schedule (`kz_knn_plan`, DESIGN.md section 3.0 'host schedule': greedy rounds)."""
import ctypes as C

import pytest


def _plan(n_q, n_i, k_eff, slots, force_splits=0, min_splits=1):
    from kiez_amd import _native as N
    lib = N.load()
    n = C.c_int(0)
    qt, pc, ln = (C.c_int * 8)(), (C.c_int * 8)(), (C.c_int * 8)()
    rc = lib.kz_knn_plan(n_q, n_i, k_eff, slots, force_splits, min_splits, C.byref(n), qt, pc, ln)
    assert rc == 0, lib.kz_last_error()
    return [(qt[r], pc[r], ln[r]) for r in range(n.value)]


def _makespan(rounds, slots):
    """Tile-times until the last workgroup finishes when items are dispatched in order onto `slots` slots."""
    import heapq
    free = [0] * slots
    heapq.heapify(free)
    for n_qt, pieces, length in rounds:
        for _ in range(n_qt * pieces):
            t = heapq.heappop(free)
            heapq.heappush(free, t + length)
    return max(free)


@pytest.mark.parametrize("n_q,n_i,k_eff,slots", [
    (100_000, 100_000, 10, 512),     # C1 at 2 workgroups per CU
    (100_000, 100_000, 10, 768),     # C1 at 3 per CU (fp16 kernel, d <= 128)
    (250_000, 1_000_000, 10, 256),   # C4 per-GPU share, one workgroup per CU
    (500_000, 500_000, 50, 256),     # C3
    (1000, 1300, 10, 512),           # fewer items than slots
    (128, 100_000, 11, 512),         # a single query tile
    (100_000, 1000, 10, 512),        # short index
])
def test_rounds_cover_everything_and_balance(n_q, n_i, k_eff, slots):
    rounds = _plan(n_q, n_i, k_eff, slots)
    n_qtiles, n_ytiles = -(-n_q // 128), -(-n_i // 128)
    assert 1 <= len(rounds) <= 8
    assert sum(r[0] for r in rounds) == n_qtiles
    kp = 16 if k_eff <= 12 else 32 if k_eff <= 26 else 64 if k_eff <= 54 else 128
    for n_qt, pieces, length in rounds:
        assert n_qt >= 1 and pieces >= 1
        assert (pieces - 1) * length < n_ytiles <= pieces * length          # the ranges tile the index exactly
        assert pieces * kp <= 4096 and pieces <= 128                         # finalize's per-query list budget
        assert length >= min(8, n_ytiles) or pieces == 1                     # no confetti
    lengths = [r[2] for r in rounds]
    assert lengths == sorted(lengths, reverse=True)                          # long items first (LPT order)
    for n_qt, pieces, _ in rounds[:-1]:
        assert slots - pieces < n_qt * pieces <= slots                       # every round but the last fills the chip
    total = n_qtiles * n_ytiles
    if total >= 20 * slots * 8:
        assert _makespan(rounds, slots) <= 1.06 * total / slots              # within 6 % of perfect balance


def test_c1_schedule_is_the_documented_one():
    assert _plan(100_000, 100_000, 10, 512) == [(512, 1, 782), (256, 2, 391), (14, 36, 22)]
    assert _plan(100_000, 100_000, 10, 768) == [(768, 1, 782), (14, 53, 15)]


def test_a_small_launch_takes_one_round_where_that_is_shorter():
    """Every item pays the start of a sweep (~24 tiles' worth): fewer query tiles than slots -> floor(slots / tiles) ranges, ONE round,
    when start + length beats the greedy rounds' sum (kz_plan.h; measured with per-workgroup clock stamps on the 15 k x 15 k shape)."""
    assert _plan(15_000, 15_000, 10, 512) == [(118, 4, 30)]                  # (greedy: [(102, 5, 24), (16, 14, 9)])
    assert _plan(500 * 128, 100_000, 10, 768) == [(384, 2, 391), (109, 7, 112), (7, 87, 9)]   # two thirds of the slots: the rounds stay
    assert _plan(1000, 1300, 10, 512) == [(8, 1, 11)]                        # an index of 11 tiles is not cut below 8


def test_knobs():
    assert _plan(5000, 50_000, 10, 512, force_splits=5) == [(40, 5, 79)]
    r = _plan(100_000, 100_000, 10, 512, min_splits=4)
    assert r[0][1] == 4 and sum(x[0] for x in r) == 782
    from kiez_amd import _native as N
    lib = N.load()
    n = C.c_int(0)
    a = (C.c_int * 8)()
    assert lib.kz_knn_plan(10, 10, 500, 512, 0, 1, C.byref(n), a, a, a) != 0   # k beyond the supported maximum
    assert b"maximum" in lib.kz_last_error()
