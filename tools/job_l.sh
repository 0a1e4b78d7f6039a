cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --workload ns --steps 1 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/b.json 2> $O/b.err
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# print the last ~260 kernels (the timed step + host_api leg) compactly, merging runs of identical names
out = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:40]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    if out and out[-1][0] == name and out[-1][3] == g:
        out[-1][1] += 1; out[-1][2] += dur
    else:
        out.append([name, 1, dur, g, (int(r["Start_Timestamp"]) - t0) / 1e6])
for o in out[-120:]:
    print("%9.2f ms  %-40s x%-4d total %9.1f us grid %s" % (o[4], o[0], o[1], o[2], o[3]))
PY
rm -rf $O/kt
