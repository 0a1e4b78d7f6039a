S=$PWD/build/abl/libkiez_amd_stamp.so
for w in c1 ns c3; do
  KIEZ_AMD_LIB=$S timeout 300 python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-others > $O/stamp_$w.json 2> $O/stamp_$w.err
  echo "== $w"; grep "fp16 kernel" $O/stamp_$w.err | sort | uniq -c | sort -rn | head -3
done
