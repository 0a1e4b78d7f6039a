T0=$(date +%s); python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "wall $(( $(date +%s) - T0 )) s"; tail -3 $O/bench_default.err
python3 - <<'PY'
import json,os
j=json.loads(open(os.environ["O"]+"/bench_default.json").read().strip().splitlines()[-1])
print("MAIN", j["config"]["workload"][:40], "ms/step %.2f value %.4g frac %.3f" % (j["ms_per_step"], j["value"], j["roofline"]["frac"]), j.get("check"))
print("cpu", {k:(round(v,3) if isinstance(v,float) else v) for k,v in j["cpu_baseline"].items() if k in ("value","cores","kind","fit_seconds_extrapolated","kneighbors_seconds_extrapolated")})
print("host_api", j["host_api"]["value"])
for k,v in j["other_workloads"].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ("workload",)})
PY
