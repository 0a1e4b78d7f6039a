"""Idle gaps of the GPU inside the last bench step of a rocprofv3 --kernel-trace csv: the union of all kernels' busy intervals
(every stream), gaps over MIN us listed with the kernel that ended before and the one that started after.
    python3 tools/gaps.py <kernel_trace.csv> [min_us=15]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:56], r["Stream_Id"]) for r in rows)
# one steady-state step period: from the end of the third-to-last marker kernel (the biggest kernel = the step's sweep) to the end
# of the second-to-last one (the last one is followed by the bench's other legs)
big = max(ev, key=lambda e: e[1] - e[0])[2]
marks = [i for i, e in enumerate(ev) if e[2] == big and (e[1] - e[0]) > 0.5 * max(x[1] - x[0] for x in ev)]
# (the shortest period between two consecutive sweeps' ends is a steady-state step; warm-up -> timed transitions and the bench's
#  other legs make longer ones)
pairs = [(ev[b][1] - ev[a][1], ev[a][1], ev[b][1]) for a, b in zip(marks[:-1], marks[1:])]
_, lo, hi = min(pairs)
last = [e for e in ev if e[0] >= lo and e[1] <= hi]
# drop the tail of the previous step: find the largest gap-free start -- keep it simple, print everything after prev marker
t0 = last[0][0]
busy_end = last[0][1]
total_gap = 0.0
prev = last[0]
print(f"window: {len(last)} kernels, {(last[-1][1] - t0) / 1e6:.2f} ms after the previous step's sweep ended; marker = {big}")
for e in last[1:]:
    if e[0] > busy_end:
        gap = (e[0] - busy_end) / 1e3
        total_gap += gap
        if gap >= min_us:
            print(f"{(busy_end - t0) / 1e6:9.3f} ms  gap {gap:8.1f} us   after [{prev[2]} s{prev[3]}]  before [{e[2]} s{e[3]}]")
    if e[1] > busy_end:
        busy_end = e[1]
        prev = e
print(f"idle inside the window: {total_gap / 1e3:.2f} ms")
if len(sys.argv) > 3:      # third argument: also write the window's kernels in start order (consecutive launches of one kernel merged)
    with open(sys.argv[3], "w") as out:
        prev_key, cnt, acc, line = None, 0, 0.0, ""
        for e in last:
            key = (e[2], e[3])
            dur = (e[1] - e[0]) / 1e3
            if key == prev_key:
                cnt += 1
                acc += dur
                continue
            if prev_key:
                out.write(f"{line}   x{cnt} total {acc:.1f} us\n")
            line = f"{(e[0] - t0) / 1e6:9.3f} ms  {e[2]:56s} s{e[3]}"
            prev_key, cnt, acc = key, 1, dur
        out.write(f"{line}   x{cnt} total {acc:.1f} us\n")
