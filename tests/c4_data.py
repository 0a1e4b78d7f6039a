"""Synthetic data of BASELINE.json configuration 4 at its stated size: 2M source rows x 1M target rows, d = 300, float32,
`rng.rand` (the reference's docstring data style, kiez/kiez.py:50-52).

The source is generated SHARD BY SHARD (eight blocks of 250k rows, one seed each) so that a rank of the eight-rank run can
make its own shard without materialising 2M x 300 float64 values; the single-process run concatenates the same blocks."""
import numpy as np

N_SOURCE, N_TARGET, D, K = 2_000_000, 1_000_000, 300, 10
N_SHARDS = 8
SHARD_ROWS = N_SOURCE // N_SHARDS


def source_shard(r: int, rows: int = SHARD_ROWS, d: int = D) -> np.ndarray:
    return np.random.RandomState(4000 + r).rand(rows, d).astype(np.float32)


def target_rows(rows: int = N_TARGET, d: int = D) -> np.ndarray:
    out = np.empty((rows, d), dtype=np.float32)
    rng = np.random.RandomState(4999)
    step = 250_000
    for b in range(0, rows, step):
        out[b:b + step] = rng.rand(min(step, rows - b), d)
    return out


def full_source(shards: int = N_SHARDS, rows: int = SHARD_ROWS, d: int = D) -> np.ndarray:
    out = np.empty((shards * rows, d), dtype=np.float32)
    for r in range(shards):
        out[r * rows:(r + 1) * rows] = source_shard(r, rows, d)
    return out
