timeout 1500 python3 -m pytest tests/test_gpu_dual.py tests/test_gpu_parity.py tests/test_gpu_longk.py -x -q 2>&1 | tail -3
for r in 1 2; do
for w in c3 ns; do
timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-others --no-check > gpurun_out/v_$w.json 2> gpurun_out/v_err.txt
python3 tools/show.py gpurun_out/v_$w.json | cut -c1-150
done
done
