# sample sweep with short lists over interleaved parts of the sample: parity, then C3 A/B and the stride around the model's choice
timeout 1500 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -12
mkdir -p gpurun_out/ss
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check "$@" > gpurun_out/ss/${wl}_$lab.json 2> gpurun_out/ss/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/ss/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.2f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f} events/row {j['shared_sweep'].get('reverse_events_per_row',0):.0f}")
PY
}
for r in 1 2; do
run c3 long_$r --opt dual_sample_short=0
run c3 short_$r
done
for s in 4 5 6; do run c3 short_s$s --opt dual_stride=$s; done
run ns short_1
run c4s short_1
