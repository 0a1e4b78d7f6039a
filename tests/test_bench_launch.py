"""`python bench.py --gpus N` launches its own ranks (one process per GPU through torch.distributed.run, started as a CHILD of
a parent that never touches a GPU), forwards rank 0's ONE JSON line and propagates a failing rank as a non-zero exit.

Checked here without a GPU through `--launch-check`: the same launch / rendezvous / sharding / collective / timing / reporting code
on the CPU test engine over gloo (the line says so and carries `value: null` -- it is not a measurement)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(*extra, timeout=600):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra], capture_output=True, text=True, timeout=timeout, cwd=str(ROOT))


def _lines(r, detail):
    """(the ONE compact stdout line, the full record it points to).  The driver keeps a bounded tail of stdout: the line must
    stay far below it (round 5's 20 KB line was not parsed)."""
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line on stdout, from rank 0
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])
    assert line["detail"] == str(detail)
    return line, json.loads(Path(detail).read_text())


@pytest.mark.parametrize("gpus,scaling", [(2, "weak"), (3, "strong")])
def test_bench_launches_its_own_ranks(gpus, scaling, tmp_path):
    r = _run("--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--scaling", scaling, "--launch-check", "--detail", str(tmp_path / "d.json"))
    assert r.returncode == 0, r.stderr[-4000:]
    line, full = _lines(r, tmp_path / "d.json")
    assert line["n_gpus"] == gpus and line["scaling"] == scaling and line["steps"] == 1 and line["warmup"] == 1
    assert line["launch_check"] is True and line["value"] is None        # never mistaken for a measurement
    cfg = line["config"]
    if scaling == "weak":
        assert cfg["n_source_total"] == gpus * cfg["n_source_this_rank"]
    else:   # the total is fixed and split over the ranks (row_slice: rank 0 takes the remainder first)
        assert cfg["n_source_total"] == 4001 and cfg["n_source_this_rank"] == -(-4001 // gpus)
    # the exchange steps of the sharded path ran once per step on every rank
    traffic = full["collective_traffic_per_step"]
    assert traffic["broadcast"]["calls"] == 1 and traffic["all_to_all"]["calls"] == 1 and traffic["all_gather"]["calls"] == 1
    assert set(line["collective_ms_per_step"]) == {"broadcast", "broadcast_exposed", "all_to_all", "all_gather"}
    # the line of an N > 1 run carries the evidence a measurement needs: an oracle check that covers the fit state and rows of
    # EVERY rank's shard, recall@k, and the CPU baseline of the whole job (rank 0, after the timed region)
    assert line["check"]["index_rows_identical"] == line["check"]["rows"] and line["check"]["ranks"] == gpus
    chk = full["check"]
    assert chk["ranks"] == gpus and len(chk["index_rows_identical_per_rank"]) == gpus
    assert chk["index_rows_identical_per_rank"] == chk["rows_per_rank"] and chk["index_rows_identical"] == chk["rows"]
    assert chk["fit_state_rows"] >= 1000 and chk["fit_state_max_rel_err"] < 1e-12
    assert line["recall_at_k"] == chk["recall_at_k"] == 1.0
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] in ("reference", "port") and cpu["sample"]
    assert full["cpu_baseline"]["workload_rows"]["n_source_total"] == cfg["n_source_total"]


def test_every_rank_can_upload_the_target_itself(tmp_path):
    """`--target-upload local`: ShardedKiez.fit(target_from_rank0=False) -- no broadcast in the step, same results."""
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check", "--target-upload", "local", "--detail", str(tmp_path / "d.json"))
    assert r.returncode == 0, r.stderr[-4000:]
    line, full = _lines(r, tmp_path / "d.json")
    assert line["config"]["target_upload"] == "local"
    assert "broadcast" not in full["collective_traffic_per_step"] and full["collective_traffic_per_step"]["all_to_all"]["calls"] == 1
    assert line["check"]["index_rows_identical"] == line["check"]["rows"] and line["recall_at_k"] == 1.0


def test_target_broadcast_once_then_cached(tmp_path):
    """`--target-upload cached`: the broadcast runs in the first fit only (1 warm-up + 2 timed steps: none inside the timed region)."""
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-check", "--target-upload", "cached", "--detail", str(tmp_path / "d.json"))
    assert r.returncode == 0, r.stderr[-4000:]
    line, full = _lines(r, tmp_path / "d.json")
    assert line["config"]["target_upload"] == "cached" and "broadcast" not in full["collective_traffic_per_step"]
    assert line["check"]["index_rows_identical"] == line["check"]["rows"] and line["recall_at_k"] == 1.0


def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--launch-check", "--steps", "0")     # zero timed steps: every rank divides by zero
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_world_size_must_match():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--launch-check"], capture_output=True, text=True, timeout=300,
                       cwd=str(ROOT), env={**__import__("os").environ, "RANK": "0", "WORLD_SIZE": "2", "MASTER_PORT": "1", "MASTER_ADDR": "127.0.0.1"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr


def test_the_stdout_line_stays_small_whatever_the_record_holds():
    """`compact_line` on the largest record a run has produced (round 5's default line, 20 KB, kept under profiles/): the
    contract's fields survive, the line stays under 4 KB, and a record bloated further only loses optional parts."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    full = json.loads((ROOT / "profiles" / "r05_bench_default.json").read_text().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 16000
    text = mod.compact_line(full, "bench_detail.json")
    assert len(text) < 4096 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[key] == full[key] or abs(line[key] - full[key]) <= 1e-5 * abs(full[key])
    assert line["config"]["workload"] == full["config"]["workload"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"]
    assert line["roofline"]["frac"] == line["roofline"]["achieved"] / line["roofline"]["peak"]
    assert abs(line["value"] - full["config"]["n_source_total"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-7
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"]
    assert line["check"]["index_rows_identical"] == 1024 and set(line["summary"]) >= {"ns", "c1", "c2", "c3", "c4s", "c4", "ea15k"}
    # a record with a far larger summary: the optional parts go, the contract stays
    full["summary"] = {f"w{i}": [1.0] * 5 for i in range(400)}
    line = json.loads(mod.compact_line(full, "bench_detail.json"))
    assert "summary" not in line and "roofline" in line and "cpu_baseline" in line and line["value"] > 0


def test_synthetic_rows_of_every_workload():
    """bench.synth_rows: float32 rows of the stated shape for every workload name, the same rows for the same seed; `cliff` -- the
    range re-search's case -- has clusters whose spreads differ by a factor of 32."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in bench.WORKLOADS:
        x = bench.synth_rows(name, 3, 700, 24)
        assert x.shape == (700, 24) and x.dtype == np.float32 and np.isfinite(x).all(), name
        np.testing.assert_array_equal(x, bench.synth_rows(name, 3, 700, 24))
    x = bench.synth_rows("cliff", 1, 20000, 16)
    centres = (np.random.RandomState(5).standard_normal((40, 16)) * 3).astype(np.float32)
    c = np.argmin(((x[:, None, :] - centres[None]) ** 2).sum(-1), axis=1)
    spread = np.array([(x[c == j] - centres[j]).std() for j in range(40) if (c == j).sum() > 50])
    assert spread.max() / spread.min() > 8
