# C3 after the short lists + large query groups: stride and ranges-per-k once more (same box)
mkdir -p gpurun_out/tune
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check "$@" > gpurun_out/tune/${wl}_$lab.json 2> gpurun_out/tune/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/tune/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.3f} frac {j['roofline']['frac']:.3f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f} esc {j.get('escalated_rows')}")
PY
}
run c3 base
for s in 4 5 6 7; do run c3 s$s --opt dual_stride=$s; done
for d in 4 5 6 7; do run c3 d$d --opt dual_short_div=$d; done
run c3 base2
