S=$PWD/build/abl/libkiez_amd_stamp.so
for w in c1 ns c3; do
  KIEZ_AMD_LIB=$S timeout 300 python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-check > $O/stamp_$w.json 2> $O/stamp_$w.err
  grep "kz stamp" $O/stamp_$w.err | tail -2
done
for w in ns c3 c4s c3s; do
  timeout 600 python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --check > $O/b_$w.json 2> $O/b_$w.err; tail -c 1300 $O/b_$w.json; tail -2 $O/b_$w.err
done
