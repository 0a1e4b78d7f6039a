for r in 1 2; do
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10
  timeout 100 python3 tools/shape_ab.py 250000 1000000 200 10 h_wps=2
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10
  timeout 100 python3 tools/shape_ab.py 100000 100000 128 10 h_wps=2
done
mkdir -p gpurun_out/wps
for r in 1 2; do for w in 0 2; do
  timeout 300 python3 bench.py --workload ns --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check --opt h_wps=$w > gpurun_out/wps/ns_$w_$r.json 2>/dev/null
  echo "h_wps=$w: $(python3 tools/show.py gpurun_out/wps/ns_$w_$r.json | cut -c1-110)"
done; done
