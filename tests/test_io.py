"""the OpenEA loader.  The two literal cases of the reference's tests/io/test_data_loading.py:9-45, then a
round trip through files in the OpenEA layout and into `hits` bookkeeping (CPU only: the loader is host code)."""
import numpy as np
import pytest
from numpy.testing import assert_array_equal

from kiez_amd.io import _seperate_common_embedding, from_openea

EMB = np.array([[1, 2, 3], [2, 3, 4], [4, 5, 6], [5, 6, 7]])


@pytest.mark.parametrize(("values", "expected"), [
    ((EMB, {0: "s1", 1: "s2"}, {2: "t1", 3: "t2"}, {"s1": "t1", "s2": "t2"}),
     (np.array([[1, 2, 3], [2, 3, 4]]), np.array([[4, 5, 6], [5, 6, 7]]), {"s1": 0, "s2": 1}, {"t1": 0, "t2": 1}, {0: 0, 1: 1})),
    ((EMB, {0: "s1", 2: "s2"}, {1: "t1", 3: "t2"}, {"s1": "t1", "s2": "t2"}),
     (np.array([[1, 2, 3], [4, 5, 6]]), np.array([[2, 3, 4], [5, 6, 7]]), {"s1": 0, "s2": 1}, {"t1": 0, "t2": 1}, {0: 0, 1: 1})),
])
def test_seperate_common_embedding(values, expected):
    emb1, emb2, ids1, ids2, ent_links = _seperate_common_embedding(*values)
    assert_array_equal(expected[0], emb1)
    assert_array_equal(expected[1], emb2)
    assert expected[2] == ids1 and expected[3] == ids2 and expected[4] == ent_links


def test_from_openea_files(tmp_path):
    rng = np.random.RandomState(0)
    emb = rng.rand(9, 4).astype(np.float32)
    perm = rng.permutation(9)
    kg1_rows, kg2_rows = perm[:5], perm[5:]
    emb_dir, kg_dir = tmp_path / "emb", tmp_path / "kg"
    emb_dir.mkdir()
    kg_dir.mkdir()
    np.save(emb_dir / "ent_embeds.npy", emb)
    (emb_dir / "kg1_ent_ids").write_text("".join(f"http://a/{r}\t{r}\n" for r in kg1_rows))
    (emb_dir / "kg2_ent_ids").write_text("".join(f"http://b/{r}\t{r}\n" for r in kg2_rows))
    links = list(zip(sorted(kg1_rows)[:4], sorted(kg2_rows)))
    (kg_dir / "ent_links").write_text("".join(f"http://a/{a}\thttp://b/{b}\n" for a, b in links))
    emb1, emb2, ids1, ids2, ent_links = from_openea(str(emb_dir), str(kg_dir))
    assert_array_equal(emb1, emb[np.sort(kg1_rows)])
    assert_array_equal(emb2, emb[np.sort(kg2_rows)])
    assert ids1 == {f"http://a/{r}": i for i, r in enumerate(np.sort(kg1_rows))}
    assert ids2 == {f"http://b/{r}": i for i, r in enumerate(np.sort(kg2_rows))}
    assert ent_links == {ids1[f"http://a/{a}"]: ids2[f"http://b/{b}"] for a, b in links}
    assert emb1.dtype == np.float32 and len(ent_links) == 4
