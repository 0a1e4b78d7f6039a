# kernel statistics of the range re-search on one shape (tools/range_probe.py)    gpurun -- 'SHAPE="200000 200 10 euclidean" bash tools/job_range_prof.sh'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/range_prof; mkdir -p $O
export ONLY=${ONLY:-3}
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/range_probe.py ${SHAPE:-200000 200 10 euclidean} > $O/probe.log 2>&1
tail -4 $O/probe.log
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv; python3 tools/ks_show.py $O/kernel_stats.csv | head -${TOP:-24}
rm -rf $O/kt
