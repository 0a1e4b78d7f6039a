# query-group size of the work table (qgroup): C3 (ten index ranges per query tile)
mkdir -p gpurun_out/qg
run() { # workload, label, opts...
  wl=$1; lab=$2; shift; shift
  timeout 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check "$@" > gpurun_out/qg/${wl}_$lab.json 2> gpurun_out/qg/err.txt
  python3 - $wl $lab <<'PY'
import json,sys
wl,lab=sys.argv[1:3]
j=json.loads(open(f'gpurun_out/qg/{wl}_{lab}.json').read().strip().splitlines()[-1])
print(f"{wl} {lab}: ms/step {j['ms_per_step']:.2f} main {j['roofline']['avg_launch_ms']:.3f} frac {j['roofline']['frac']:.3f} rev_extra {j['shared_sweep']['reverse_extra_ms_per_step']:.2f}")
PY
}
for g in 96 384 768 1536 4096 384 96; do run c3 g$g --opt qgroup=$g; done
# a mid-sized shape with a few index ranges per query tile chosen by the planner (40k x 400k): ordinary kernel
for g in 24 96 384 4096; do
KZ=1 python3 tools/shape_ab.py 40000 400000 200 10 qgroup=$g
done
for g in 24 96 384 4096; do
KZ=1 python3 tools/shape_ab.py 8000 1000000 200 10 qgroup=$g
done
