"""Synthetic dataset in the OpenEA layout (https://github.com/nju-websoft/OpenEA#dataset-description; what
kiez/io/data_loading.py:75-99 reads): ONE embedding matrix for the entities of both knowledge graphs with their rows
interleaved at random, `kg1_ent_ids` / `kg2_ent_ids` (uri <TAB> row) and the gold alignment `ent_links` (uri1 <TAB> uri2).

Entity i of KG2 is entity perm[i] of KG1 seen through a noisy linear map, so the alignment is recoverable by nearest
neighbours (hits@1 well above chance, below 1) and hub entities exist (a few cluster centres).  TEST / BENCH DATA ONLY.

    python tests/data/openea_synth.py OUT_DIR [n_entities] [dim] [seed]   ->  OUT_DIR/emb, OUT_DIR/kg
"""
import os
import sys

import numpy as np


def write_openea(out_dir, n=15000, d=100, seed=0, n_links=None, noise=1.4, dtype=np.float32):
    rng = np.random.RandomState(seed)
    centres = rng.randn(32, d) * 1.5
    e1 = (centres[rng.randint(0, 32, n)] + rng.randn(n, d)).astype(np.float64)
    perm = rng.permutation(n)                         # KG2 entity i corresponds to KG1 entity perm[i]
    e2 = e1[perm] + noise * rng.randn(n, d)
    rows = rng.permutation(2 * n)                     # rows of the common matrix: the two graphs interleaved at random
    rows1, rows2 = rows[:n], rows[n:]
    emb = np.empty((2 * n, d), dtype=dtype)
    emb[rows1] = e1
    emb[rows2] = e2
    emb_dir, kg_dir = os.path.join(out_dir, "emb"), os.path.join(out_dir, "kg")
    os.makedirs(emb_dir, exist_ok=True)
    os.makedirs(kg_dir, exist_ok=True)
    np.save(os.path.join(emb_dir, "ent_embeds.npy"), emb)
    with open(os.path.join(emb_dir, "kg1_ent_ids"), "w") as fh:
        fh.writelines(f"http://kg1.example/e{i}\t{rows1[i]}\n" for i in range(n))
    with open(os.path.join(emb_dir, "kg2_ent_ids"), "w") as fh:
        fh.writelines(f"http://kg2.example/e{i}\t{rows2[i]}\n" for i in range(n))
    n_links = n if n_links is None else n_links      # (a dataset's gold links usually cover a part of the entities)
    linked = np.sort(rng.choice(n, n_links, replace=False))
    with open(os.path.join(kg_dir, "ent_links"), "w") as fh:
        fh.writelines(f"http://kg1.example/e{perm[i]}\thttp://kg2.example/e{i}\n" for i in linked)
    return emb_dir, kg_dir


if __name__ == "__main__":
    a = sys.argv[1:]
    print(write_openea(a[0], *(int(x) for x in a[1:4])))
