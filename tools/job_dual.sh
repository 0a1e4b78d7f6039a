for r in 1 2; do timeout 300 python3 bench.py --workload ns --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/b.json 2> $O/b.err; echo "$(python3 tools/show.py $O/b.json | cut -c1-150)"; done
timeout 300 python3 bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/b.json 2> $O/b.err; echo "$(python3 tools/show.py $O/b.json | cut -c1-150)"
