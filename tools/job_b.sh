timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity and not escalat and not tiers_agree and not below_the_float32" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for w in c1 c2; do
  timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --check > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json; tail -2 $O/b_$w.err
done
timeout 300 python3 bench.py --workload c1 --steps 5 --warmup 2 --no-cpu-baseline --no-check --opt h_wps=2 > $O/b_c1_w2.json 2> $O/b_c1_w2.err; python3 tools/show.py $O/b_c1_w2.json
for w in ns c3 c4s; do
  timeout 300 python3 bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/b_$w.json 2> $O/b_$w.err; python3 tools/show.py $O/b_$w.json; tail -2 $O/b_$w.err
done
S=$PWD/build/abl/libkiez_amd_stamp.so
for w in c1 ns; do
  KIEZ_AMD_LIB=$S timeout 300 python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-check > $O/stamp_$w.json 2> $O/stamp_$w.err
  grep "kz stamp" $O/stamp_$w.err | tail -2
done
