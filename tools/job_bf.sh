# reverse direction with lists of 2 K' (dual_rev_long): clustered data and C3
for rl in 0 1; do
echo "dual_rev_long=$rl"
KZ_OPTS="dual_rev_long=$rl" timeout 1700 python3 tools/short_route_stress.py 400000 200 50 2>&1 | grep "short=1"
done
for rl in 0 1 0 1; do
timeout 300 python3 bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-others --no-check --opt dual_rev_long=$rl | python3 tools/show.py /dev/stdin | cut -c1-120
done
timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q 2>&1 | tail -2
