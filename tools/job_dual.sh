timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/dual_check.py ns > $O/dual.log 2>&1; tail -2 $O/dual.log | head -1 | cut -c1-200
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python3 tools/ks_show.py $f kz_dual_s; rm -rf $O/ks
