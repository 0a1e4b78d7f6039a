set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2first; mkdir -p $O
timeout 600 python3 -m pytest tests -m gpu -x -q -k "parity and not escalat and not tiers_agree" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for w in c1 c2; do
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --check > $O/b_$w.json 2> $O/b_$w.err; tail -c 1500 $O/b_$w.json; tail -3 $O/b_$w.err
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-check --opt h_wps=2 > $O/b_${w}_w2.json 2> $O/b_${w}_w2.err; tail -c 600 $O/b_${w}_w2.json
done
