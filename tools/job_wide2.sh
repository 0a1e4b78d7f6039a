timeout 1200 python3 -m pytest tests/test_gpu_dual.py -x -q 2>&1 | tail -3
for r in 1 2; do
for w in 0 1; do
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10 h_wide=$w
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 h_wide=$w
  timeout 100 python3 tools/shape_ab.py 500000 500000 200 50 h_wide=$w
done
done
