"""SURVEY 8 f-4 end to end: an OpenEA-layout dataset on disk -> `kiez_amd.io.from_openea` (kiez/io/data_loading.py:75-99)
-> `Kiez(hubness="CSLS").fit(emb1, emb2).kneighbors_device(k)` -> `kiez_amd.evaluate.hits` on the device
(kiez/evaluate/eval_metrics.py:23-61): the neighbour matrix equals the oracle pipeline's, hits@k equals the reference's
formula evaluated on it, and `bench.py --openea` reports the same."""
import json
import subprocess
import sys
import warnings
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _ref_hits(nn_ind, gold, ks):
    """the reference's loop (eval_metrics.py:8-13, 53-61)"""
    return {k: sum(1 for i in range(len(nn_ind)) if i in gold and gold[i] in nn_ind[i][:k]) / len(gold) for k in ks}


@pytest.mark.parametrize("hub,metric", [("CSLS", "euclidean"), (None, "cosine")])
def test_openea_directory_to_hits(tmp_path, hub, metric):
    from kiez_amd import Kiez
    from kiez_amd.evaluate import hits
    from kiez_amd.io import from_openea
    from oracle import kiez_oracle as O
    from tests.data.openea_synth import write_openea
    emb_dir, kg_dir = write_openea(str(tmp_path), n=3000, d=40, seed=5, n_links=2400)
    emb1, emb2, ids1, ids2, links = from_openea(emb_dir, kg_dir)
    assert emb1.shape == emb2.shape == (3000, 40) and len(links) == 2400 and emb1.dtype == np.float32
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kz = Kiez(n_candidates=10, algorithm="SklearnNN", algorithm_kwargs={"metric": metric}, hubness=hub).fit(emb1, emb2)
        d_dev, i_dev = kz.kneighbors_device(10)
        got = hits(i_dev, links)                              # device neighbour matrix, device scan
    ind = i_dev.numpy()
    s64 = emb1.astype(np.float64) if metric == "cosine" else emb1
    t64 = emb2.astype(np.float64) if metric == "cosine" else emb2
    od, oi = O.kiez_pipeline(s64, t64, 10, 10, metric, 2, hub, {})
    np.testing.assert_array_equal(ind, oi)
    ref = _ref_hits(oi, links, [1, 5, 10])
    assert got == pytest.approx(ref, abs=0) and 0.3 < ref[1] < 1.0 and ref[10] > ref[1]
    assert hits(ind, links) == pytest.approx(ref, abs=0)      # host matrix through the same kernel


def test_bench_openea_mode(tmp_path):
    from tests.data.openea_synth import write_openea
    emb_dir, kg_dir = write_openea(str(tmp_path), n=4000, d=32, seed=7)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--openea", emb_dir, kg_dir, "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["data"].startswith("OpenEA") and line["config"]["n_source"] == 4000 and line["value"] > 0
    assert line["check"]["index_rows_identical"] == line["check"]["rows"] and 0.3 < line["hits"]["1"] <= line["hits"]["10"] <= 1.0
    assert line["hits"] == {str(k): v for k, v in line["check"]["hits_reference_formula"].items()}


def test_hits_with_float_valued_gold():
    """gold read with np.loadtxt or from a pandas column holds floats: the reference's `i in gold and gold[i] in nn_ind[i][:k]`
    (eval_metrics.py:8-13) matches integral floats on both sides (4.0 == 4, equal hashes) and never a fractional one."""
    from kiez_amd.evaluate import hits
    rng = np.random.RandomState(3)
    nn_ind = np.stack([rng.permutation(50)[:10] for _ in range(40)])
    gold = {}
    for i in range(0, 40, 2):
        tgt = int(nn_ind[i][rng.randint(0, 10)]) if i % 4 == 0 else int(rng.randint(0, 50))
        key = [i, float(i), np.float64(i), np.int32(i)][(i // 2) % 4]
        gold[key] = [tgt, float(tgt), np.float64(tgt), np.float32(tgt)][(i // 2 + 1) % 4]
    gold[7] = 3.5            # a fractional target never matches, but counts in len(gold)
    gold[9.5] = 1            # a fractional key is no row number
    gold["x"] = 2            # nor is a label
    gold[11] = 1e20          # integral, but no int64 neighbour id can equal it: no hit, no OverflowError (round-4 advisor finding)
    gold[13] = 2 ** 70
    gold[float(2 ** 64)] = 4
    gold[True] = int(nn_ind[1][0])    # True == 1 and hash(True) == hash(1): row 1's gold in the reference as well
    ref = _ref_hits(nn_ind, gold, [1, 5, 10])
    assert ref[10] > 0
    assert hits(nn_ind, gold) == pytest.approx(ref, abs=0)
