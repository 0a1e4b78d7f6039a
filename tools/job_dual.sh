timeout 900 python3 -m pytest tests/test_gpu_dual.py -x -q -m gpu 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/dual_check.py ns c3 > $O/dual.log 2>&1; tail -3 $O/dual.log | cut -c1-250
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python3 tools/ks_show.py $f kz_dual; rm -rf $O/ks
