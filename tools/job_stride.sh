mkdir -p gpurun_out/stride
for s in 1 16 20 28 32; do
  timeout 300 python3 bench.py --workload ns --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check --opt dual_stride=$s > gpurun_out/stride/ns_$s.json 2> gpurun_out/stride/err.txt
  echo "ns stride=$s: $(python3 tools/show.py gpurun_out/stride/ns_$s.json | cut -c1-120)"
done
for s in 1 6 10 12 16; do
  timeout 300 python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check --opt dual_stride=$s > gpurun_out/stride/c3_$s.json 2> gpurun_out/stride/err.txt
  echo "c3 stride=$s: $(python3 tools/show.py gpurun_out/stride/c3_$s.json | cut -c1-120)"
done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dual.py tests/test_gpu_merge.py tests/test_gpu_torch_inputs.py tests/test_gpu_sharded_rccl.py -x -q -k "dsl or DisSim or golden or api or torch or rccl" 2>&1 | tail -3
