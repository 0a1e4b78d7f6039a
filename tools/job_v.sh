timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 1500 python3 tools/fuzz_dual.py 200 701 2>&1 | tail -1
timeout 1500 python3 tools/fuzz_api.py 150 704 2>&1 | tail -1
for w in ns c3 c4s c1 c2; do
timeout 300 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/v_$w.json 2> gpurun_out/v_err.txt
python3 tools/show.py gpurun_out/v_$w.json | cut -c1-150
done
