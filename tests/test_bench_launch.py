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


@pytest.mark.parametrize("gpus,scaling", [(2, "weak"), (3, "strong")])
def test_bench_launches_its_own_ranks(gpus, scaling):
    r = _run("--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--scaling", scaling, "--launch-check")
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line on stdout, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["scaling"] == scaling and line["steps"] == 1 and line["warmup"] == 1
    assert line["launch_check"] is True and line["value"] is None        # never mistaken for a measurement
    cfg = line["config"]
    if scaling == "weak":
        assert cfg["n_source_total"] == gpus * cfg["n_source_this_rank"]
    else:   # the total is fixed and split over the ranks (row_slice: rank 0 takes the remainder first)
        assert cfg["n_source_total"] == 4001 and cfg["n_source_this_rank"] == -(-4001 // gpus)
    # the exchange steps of the sharded path ran once per step on every rank
    traffic = line["collective_traffic_per_step"]
    assert traffic["broadcast"]["calls"] == 1 and traffic["all_to_all"]["calls"] == 1 and traffic["all_gather"]["calls"] == 1
    assert set(line["collective_ms_per_step"]) == {"broadcast", "broadcast_exposed", "all_to_all", "all_gather"}
    # the line of an N > 1 run carries the evidence a measurement needs: an oracle check that covers the fit state and rows of
    # EVERY rank's shard, recall@k, and the CPU baseline of the whole job (rank 0, after the timed region)
    chk = line["check"]
    assert chk["ranks"] == gpus and len(chk["index_rows_identical_per_rank"]) == gpus
    assert chk["index_rows_identical_per_rank"] == chk["rows_per_rank"] and chk["index_rows_identical"] == chk["rows"]
    assert chk["fit_state_rows"] >= 1000 and chk["fit_state_max_rel_err"] < 1e-12
    assert line["recall_at_k"] == chk["recall_at_k"] == 1.0
    cpu = line["cpu_baseline"]
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] in ("reference", "port")
    assert cpu["workload_rows"]["n_source_total"] == cfg["n_source_total"]


def test_every_rank_can_upload_the_target_itself():
    """`--target-upload local`: ShardedKiez.fit(target_from_rank0=False) -- no broadcast in the step, same results."""
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-check", "--target-upload", "local")
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert line["config"]["target_upload"] == "local"
    assert "broadcast" not in line["collective_traffic_per_step"] and line["collective_traffic_per_step"]["all_to_all"]["calls"] == 1
    assert line["check"]["index_rows_identical"] == line["check"]["rows"] and line["recall_at_k"] == 1.0


def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--launch-check", "--steps", "0")     # zero timed steps: every rank divides by zero
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_world_size_must_match():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--launch-check"], capture_output=True, text=True, timeout=300,
                       cwd=str(ROOT), env={**__import__("os").environ, "RANK": "0", "WORLD_SIZE": "2", "MASTER_PORT": "1", "MASTER_ADDR": "127.0.0.1"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
