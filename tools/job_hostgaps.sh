# HIP API calls over MIN us inside one steady-state bench step (host side of tools/gaps.py)   gpurun -- 'WL=ns bash tools/job_hostgaps.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export O=gpurun_out/hostgaps; mkdir -p $O
for w in ${WL:-ns}; do
  timeout 400 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $O/kt_$w -- python3 bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-others --detail bench_detail_$w.json --no-check > $O/$w.json 2> $O/$w.err
  ls $O/kt_$w/*/ | head
  k=$(find $O/kt_$w -name "*kernel_trace.csv" | head -1); h=$(find $O/kt_$w -name "*hip_api_trace.csv" | head -1)
  python3 - "$k" "$h" ${MIN:-40} <<'PY' | tee $O/${w}_host.txt | tail -${TAIL:-70}
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]) for r in rows)
big = max(ev, key=lambda e: e[1] - e[0])[2]
longest = max(e[1] - e[0] for e in ev)
marks = [e for e in ev if e[2] == big and e[1] - e[0] > 0.5 * longest]
pairs = [(b[1] - a[1], a[1], b[1]) for a, b in zip(marks[:-1], marks[1:])]
_, lo, hi = min(pairs)
api = list(csv.DictReader(open(sys.argv[2])))
print("columns", list(api[0].keys()))
mn = float(sys.argv[3]) * 1e3
tot = {}
for r in api:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < lo or e > hi: continue
    tot.setdefault(r["Function"], [0, 0.0]); tot[r["Function"]][0] += 1; tot[r["Function"]][1] += (e - s) / 1e3
    if e - s >= mn:
        print(f"{(s - lo) / 1e6:9.3f} ms  {r['Function']:36s} {(e - s) / 1e3:9.1f} us")
for f, (n, us) in sorted(tot.items(), key=lambda x: -x[1][1])[:14]:
    print(f"   total {f:36s} x{n:5d} {us / 1e3:8.2f} ms")
PY
  rm -rf $O/kt_$w
done
