timeout 1200 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_northstar.py tests/test_gpu_wide.py -x -q 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
mkdir -p gpurun_out/ovl
for r in 1 2; do
for w in 0 1; do
  for wl in ns c3 c4s; do
  timeout 300 python3 bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-others --opt dual_overlap=$w > gpurun_out/ovl/${wl}_o${w}_$r.json 2> gpurun_out/ovl/err.txt
  echo "dual_overlap=$w r$r: $(python3 tools/show.py gpurun_out/ovl/${wl}_o${w}_$r.json | cut -c1-200)"
  done
done
done
