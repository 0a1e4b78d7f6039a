# same-box A/B: alternate the two libraries, 2 rounds
for r in 1 2; do
for lib in default $AB; do
  if [ $lib = default ]; then unset KIEZ_AMD_LIB; else export KIEZ_AMD_LIB=$PWD/build/abl/libkiez_amd_$lib.so; fi
  for w in $WL; do
    timeout 300 python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others --no-check > $O/ab_${lib}_${w}_$r.json 2> $O/ab.err; echo "$lib r$r: $(python3 tools/show.py $O/ab_${lib}_${w}_$r.json | cut -c1-110)"
  done
done
done
