"""Loading entity embeddings in the OpenEA layout (reference: kiez/io/data_loading.py:75-99).

An OpenEA embedding directory holds ONE matrix for the entities of both knowledge graphs (`ent_embeds.npy`) plus two
tab-separated maps `entity uri <TAB> row` (`kg1_ent_ids`, `kg2_ent_ids`); the knowledge-graph directory holds the gold
alignment `ent_links` (`uri in KG1 <TAB> uri in KG2`).  `from_openea` returns what `Kiez.fit(source, target)` and
`kiez_amd.evaluate.hits` consume: the two embedding matrices (rows in ascending order of their row in the common
matrix, as the reference's loader produces them), `uri -> new row` for both sides, and the gold links as
`source row -> target row`.

Same function names (the reference's spelling included), arguments and return values as the reference, so
`from kiez.io.data_loading import from_openea` call sites switch by changing the import; the row selection is one numpy
gather per side instead of a Python loop over all embedding rows.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np


def _read_tsv_pairs(path):
    with open(path) as fh:
        for line in fh:
            left, right = line.strip().split("\t")[:2]
            yield left, right


def _read_kg_ids(path) -> Dict[int, str]:
    """row in the common embedding matrix -> entity uri"""
    return {int(row): uri for uri, row in _read_tsv_pairs(path)}


def _read_ent_links(path) -> Dict[str, str]:
    """uri in KG1 -> uri in KG2"""
    return dict(_read_tsv_pairs(path))


def _side(emb: np.ndarray, kg_ids: Dict[int, str]) -> Tuple[np.ndarray, Dict[str, int]]:
    """The rows of `emb` that belong to one knowledge graph, in ascending row order, and uri -> new row."""
    rows = np.array(sorted(r for r in kg_ids if 0 <= r < len(emb)), dtype=np.int64)
    return emb[rows], {kg_ids[int(r)]: new for new, r in enumerate(rows)}


def _seperate_common_embedding(emb: np.ndarray, kg1_ids: Dict[int, str], kg2_ids: Dict[int, str], ent_links: Dict[str, str]
                               ) -> Tuple[np.ndarray, np.ndarray, Dict[str, int], Dict[str, int], Dict[int, int]]:
    """Split the common embedding matrix by knowledge graph (kiez/io/data_loading.py:44-71).

    Returns emb1, emb2, {uri -> row of emb1}, {uri -> row of emb2}, {row of emb1 -> row of emb2} for the linked entities."""
    emb1, ids1 = _side(np.asarray(emb), kg1_ids)
    emb2, ids2 = _side(np.asarray(emb), kg2_ids)
    links = {ids1[a]: ids2[b] for a, b in ent_links.items()}
    return emb1, emb2, ids1, ids2, links


def _read_openea_files(emb_dir_path, kg_path):
    emb = np.load(os.path.join(emb_dir_path, "ent_embeds.npy"))
    return (emb, _read_kg_ids(os.path.join(emb_dir_path, "kg1_ent_ids")), _read_kg_ids(os.path.join(emb_dir_path, "kg2_ent_ids")),
            _read_ent_links(os.path.join(kg_path, "ent_links")))


def from_openea(emb_dir_path: str, kg_path: str):
    """Load OpenEA-type data: `emb1, emb2, kg1_ids, kg2_ids, ent_links = from_openea(embedding_dir, kg_dir)`
    (dataset layout: https://github.com/nju-websoft/OpenEA#dataset-description)."""
    return _seperate_common_embedding(*_read_openea_files(emb_dir_path, kg_path))
