mkdir -p gpurun_out/stride
for s in 4 5 6 7 8; do
  timeout 300 python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check --opt dual_stride=$s > gpurun_out/stride/c3_$s.json 2> gpurun_out/stride/err.txt
  echo "c3 stride=$s: $(python3 tools/show.py gpurun_out/stride/c3_$s.json | cut -c1-120)"
  python3 -c "
import json; j=json.loads(open('gpurun_out/stride/c3_$s.json').read().strip().splitlines()[-1]); print('   reverse_extra', j['shared_sweep'])"
done
