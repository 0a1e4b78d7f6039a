# wide workgroups (one per CU, its query tiles share the ring): correctness, then same-process-family A/B by option
timeout 1200 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_gpu_northstar.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
for r in 1 2; do
for w in 0 1; do
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10 h_wide=$w
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 h_wide=$w
  timeout 100 python3 tools/shape_ab.py 250000 1000000 300 10 h_wide=$w
  timeout 100 python3 tools/shape_ab.py 500000 500000 200 50 h_wide=$w
done
done
mkdir -p gpurun_out/wide
for r in 1 2; do
for w in 0 1; do
  for wl in ns c3; do
  timeout 300 python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check --opt h_wide=$w > gpurun_out/wide/${wl}_w${w}_$r.json 2> gpurun_out/wide/err.txt
  echo "h_wide=$w r$r: $(python3 tools/show.py gpurun_out/wide/${wl}_w${w}_$r.json | cut -c1-130)"
  done
done
done
