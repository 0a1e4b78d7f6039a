mkdir -p gpurun_out/sm
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others "$@" > gpurun_out/sm/${wl}_$lab.json 2> gpurun_out/sm/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/sm/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.2f} frac {j['roofline']['frac']:.3f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f} esc {j.get('escalated_rows')} fin {j['other_kernels_ms'].get('finalize_avg'):.2f}")
PY
}
run c3 kp16
run c3 kp32_d5 --opt dual_short_kp=32
run c3 kp32_d4 --opt dual_short_kp=32 --opt dual_short_div=4
run c3 kp32_d6 --opt dual_short_kp=32 --opt dual_short_div=6
run c3 kp16b
