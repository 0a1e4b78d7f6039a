for r in 1 2; do
for o in "" "--opt chunk_rows=524288"; do
  timeout 300 python3 bench.py --workload ns --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check $o > $O/b.json 2> $O/b.err; echo "[$o] $(python3 tools/show.py $O/b.json | cut -c1-130)"
done; done
timeout 300 python3 bench.py --workload c4s --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-check > $O/b.json 2> $O/b.err; echo "c4s $(python3 tools/show.py $O/b.json | cut -c1-130)"
timeout 300 python3 bench.py --workload c4s --steps 3 --warmup 1 --no-cpu-baseline --no-others --no-check --opt chunk_rows=524288 > $O/b.json 2> $O/b.err; echo "c4s 524k $(python3 tools/show.py $O/b.json | cut -c1-130)"
