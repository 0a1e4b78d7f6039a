# when the workgroups of a sweep start and end (tools/stamp_show.py)    gpurun -- 'bash tools/job_stamp.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stamp; mkdir -p $O; rm -f $O/*.txt
for shape in "100000 100000 128 10" "15000 15000 300 10" "98304 12800 128 10"; do
  f=$O/stamp_$(echo $shape | tr ' ' '_').txt
  for lib in ${LIBS:-stamp}; do
    rm -f $f
    KZ_STAMP_FILE=$f KIEZ_AMD_LIB=build/abl/libkiez_amd_$lib.so python3 tools/shape_ab.py $shape tier_probe=0 abl=2 spec_rows=0 2>&1 | tail -1
    python3 tools/stamp_show.py $f | tee $O/show_${lib}_$(echo $shape | tr ' ' '_').log
  done
done
