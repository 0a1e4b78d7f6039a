# escalation with more lists (esc_short): parity, then ns / c4s A/B
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dual.py -x -q 2>&1 | tail -4
mkdir -p gpurun_out/esc
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-others "$@" > gpurun_out/esc/${wl}_$lab.json 2> gpurun_out/esc/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/esc/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.2f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f} esc {j.get('escalated_rows')} fb_ms {j['other_kernels_ms'].get('fallback_total'):.2f}")
PY
}
for r in 1 2; do
run ns long_$r --opt esc_short=0
run ns short_$r
done
run c4s long --opt esc_short=0
run c4s short
