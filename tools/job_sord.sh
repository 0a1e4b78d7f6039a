# short-list route of the ORDINARY kernel (short_ord): main-kernel time of one search, C3 as two searches, and the hard-data tools
for o in 0 1; do python3 tools/shape_ab.py 250000 500000 200 50 short_ord=$o; done
for o in 0 1; do python3 tools/shape_ab.py 250000 500000 200 26 short_ord=$o; done
for o in 0 1; do python3 tools/shape_ab.py 100000 500000 200 70 short_ord=$o; done
for o in 0 1; do
timeout 300 python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-others --opt dual_stride=0 --opt short_ord=$o | python3 tools/show.py /dev/stdin | cut -c1-250
done
for o in 0 1; do echo "short_ord=$o"; KZ_OPTS="short_ord=$o" timeout 900 python3 tools/short_route_stress.py 400000 200 50 2>&1 | grep "short=1"; done
