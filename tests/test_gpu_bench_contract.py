"""The driver's contract for `bench.py` (task statement): ONE JSON line on stdout with the metric BASELINE.json names, whole-job
throughput, `roofline` and `cpu_baseline` objects -- checked on a short run of the default workload on the GPU."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_default_bench_line_has_what_the_driver_reads(tmp_path):
    """The DRIVER'S command shape (`python3 bench.py --gpus 1 --steps K --warmup W`, the other workloads ON -- one step each here):
    one stdout line, short enough for the driver's bounded tail, that parses and carries `roofline` and `cpu_baseline`."""
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--other-steps", "1",
                        "--other-warmup", "0", "--detail", str(tmp_path / "detail.json")], capture_output=True, text=True,
                       timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096 and len(r.stdout) < 4200
    line = json.loads(lines[0])
    full = json.loads((tmp_path / "detail.json").read_text())
    assert set(full["other_workloads"]) >= {"c1", "c2", "c3", "c4s", "c4", "ea15k"} and not [o for o in full["other_workloads"].values() if "error" in o]
    rows = line["summary"]
    assert set(rows) >= {"columns", "ns", "c1", "c2", "c3", "c4s", "c4", "c1g", "hard", "cliff", "gmm", "ea15k"}
    for name, row in rows.items():
        if name != "columns":
            got, of = row[4].split("/")
            assert got == of, (name, row)                                  # every workload's oracle check: all rows identical
    base = json.loads((ROOT / "BASELINE.json").read_text())
    assert line["metric"] == base["metric"] and line["unit"] == "queries/s" and line["higher_is_better"] is True
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["data"] == "synthetic" and line["dtype"] in ("f16", "bf16x2", "f32")
    cfg = line["config"]
    assert "workload" in cfg and cfg["d"] == 200 and cfg["n_target"] == 1_000_000 and cfg["k"] == 10 and cfg["hubness"] == "CSLS"
    assert abs(line["value"] - cfg["n_source_total"] * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"])) / line["value"] < 1e-6
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.2 < rf["frac"] < 0.7
    assert abs(rf["achieved"] - rf["algorithmic_flop_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) / rf["achieved"] < 1e-6
    assert rf["avg_launch_ms"] < line["ms_per_step"]                      # the dominant kernel is inside the step
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "queries/s"
    chk = line["check"]
    assert chk["rows"] == 1024 and chk["index_rows_identical"] == 1024 and chk["recall_at_k"] == 1.0
    assert line["certification_fallback_rows"] == full["speculative_rescue_rows"] <= 64 * 2 * line["steps"] and full["rounding_bound_self_check"]["max_err_over_eps"] < 1.0
