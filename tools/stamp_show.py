"""Workgroup start / end times of the fp16 sweep (KZ_STAMP_FILE of a -DKZ_ABL_STAMP build, option abl = 2): for the LAST launch in
the file -- when the first and the last workgroup started, how long a workgroup runs, when the first and the last one ended; and a
coarse timeline of how many workgroups are running.   python3 tools/stamp_show.py <file>"""
import sys
import numpy as np
launches, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        cur = {"hdr": ln.strip(), "rows": []}
        launches.append(cur)
    else:
        v = [int(x) for x in ln.split()]
        cur["rows"].append(v[:3])
        if len(v) > 3:
            cur.setdefault("tiles", []).append((v[0], v[1], v[3:]))
for L in launches[-int(sys.argv[2]) if len(sys.argv) > 2 else -1:]:
    a = np.array(L["rows"], dtype=np.int64)
    s, e = (a[:, 1] - a[:, 1].min()) / 100.0, (a[:, 2] - a[:, 1].min()) / 100.0      # us (100 MHz clock)
    dur = e - s
    print(L["hdr"])
    print(f"  starts: first 0.0  median {np.median(s):8.1f}  p90 {np.percentile(s, 90):8.1f}  last {s.max():8.1f} us")
    print(f"  run time of a workgroup: min {dur.min():8.1f}  median {np.median(dur):8.1f}  max {dur.max():8.1f} us")
    print(f"  ends:   first {e.min():8.1f}  median {np.median(e):8.1f}  p90 {np.percentile(e, 90):8.1f}  last {e.max():8.1f} us")
    T = e.max()
    for t in np.linspace(0, T, 21):
        print(f"    t = {t:8.1f} us: {int(((s <= t) & (e > t)).sum()):5d} running")

    for w, t0, ts in L.get("tiles", [])[:6]:
        ts = np.array(ts, dtype=np.int64)
        first = ts[:64][ts[:64] > 0]
        if len(first) < 3:
            continue
        d = np.diff(first) / 100.0
        line = " ".join(f"{x:.1f}" for x in d[:40])
        later = ts[64:][ts[64:] > 0]
        dl = np.diff(later) / 100.0 / 16.0
        print(f"  wg {w}: start -> first tile {(first[0] - t0) / 100.0:.1f} us; us per tile, tiles 0..: {line}")
        if len(dl):
            print(f"         later (per tile, 16-tile means): " + " ".join(f"{x:.2f}" for x in dl[:48]))
