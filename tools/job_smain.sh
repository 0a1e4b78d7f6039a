# short-list route of the dual main sweep: parity, then C3 A/B
timeout 1500 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -12
mkdir -p gpurun_out/sm
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others "$@" > gpurun_out/sm/${wl}_$lab.json 2> gpurun_out/sm/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/sm/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.2f} frac {j['roofline']['frac']:.3f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f} esc {j.get('escalated_rows')} fin {j['other_kernels_ms'].get('finalize_avg')} check {str(j.get('sample_check'))[:80]}")
PY
}
for r in 1 2; do
run c3 one_$r --opt dual_short_main=0
run c3 short_$r
done
for dv in 6; do run c3 div$dv --opt dual_short_div=$dv; done
