# full pytest -m gpu suite, then the default bench line with its per-workload summary.   gpurun -- 'bash tools/job_all.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/all; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; python3 tools/show.py $O/bench.json | cut -c1-200
python3 - <<'PY'
import json,sys,os
j=json.load(open(os.environ["O"]+"/bench.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], "shared", j.get("shared_sweep"))
for n,o in j.get("other_workloads", {}).items(): print(n, round(o["ms_per_step"],2), "%.4g"%o["value"], round(o["roofline_frac"],3), o.get("shared_sweeps"), o.get("check"))
PY
