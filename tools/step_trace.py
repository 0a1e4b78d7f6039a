"""One steady-state bench step as a merged timeline of kernels (every stream) and HIP API calls, from a
`rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv` run of `bench.py --workload W --no-others --no-check`.
A step starts at `kz_matrix_create`'s first `kz_norms_kernel` over the full source (the largest grid of that kernel, the first of
each run of them); the LAST complete step is printed: kernels with start offset, duration, stream; API calls over MIN us
(and every synchronising call) interleaved; totals at the end.
    python3 tools/step_trace.py <kernel_trace.csv> <hip_api_trace.csv> [min_api_us=20]"""
import csv
import sys

k = list(csv.DictReader(open(sys.argv[1])))
a = list(csv.DictReader(open(sys.argv[2])))
mn = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 20e3
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in k)
norms = [e for e in ev if "kz_norms_kernel" in e[2]]
gmax = max(e[4] for e in norms)
big = [e for e in norms if e[4] == gmax]
# step starts: a big norms launch not preceded (within 1 ms, no sweep in between) by another big one
starts = []
for e in big:
    if not starts or e[0] - starts[-1][0] > 0 and any("kz_knn_cand" in x[2] for x in ev if starts[-1][0] < x[0] < e[0]):
        starts.append(e)
lo, hi = starts[-2][0], starts[-1][0]
SYNC = ("Synchronize", "hipMemcpy", "hipMalloc", "hipFree", "hipEventQuery")
items = [(e[0], "K", e) for e in ev if lo <= e[0] < hi]
for r in a:
    s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s < hi and (t - s >= mn or ("Synchronize" in r["Function"]) or r["Function"] in ("hipMemcpy", "hipMalloc", "hipFree")):
        items.append((s, "A", (s, t, r["Function"])))
items.sort(key=lambda x: x[0])
busy = 0.0
n_k = 0
for s, kind, e in items:
    if kind == "K":
        n_k += 1
        busy += (e[1] - e[0]) / 1e3
        print(f"{(s - lo) / 1e6:8.3f} ms  K s{e[3]:<3} {(e[1] - e[0]) / 1e3:8.1f} us  {e[2][:90]}")
    else:
        print(f"{(s - lo) / 1e6:8.3f} ms    api {(e[1] - e[0]) / 1e3:8.1f} us  {e[2]}")
tot = {}
for r in a:
    s, t = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s < hi:
        tot.setdefault(r["Function"], [0, 0.0])
        tot[r["Function"]][0] += 1
        tot[r["Function"]][1] += (t - s) / 1e3
print(f"step {(hi - lo) / 1e6:.3f} ms (under the tracer), {n_k} kernel launches, sum of kernel durations {busy / 1e3:.3f} ms")
for f, (n, us) in sorted(tot.items(), key=lambda x: -x[1][1])[:12]:
    print(f"   api {f:40s} x{n:4d} {us / 1e3:8.3f} ms")
