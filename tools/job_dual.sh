for s in 8 12 16; do echo "stride $s: $(DUAL_STRIDE=$s timeout 600 python3 tools/dual_check.py ns 2>&1 | tail -2 | head -1 | cut -c30-300)"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 tools/dual_check.py ns > $O/dual.log 2>&1
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); cp "$f" $O/ns_dual_kernel_stats.csv; rm -rf $O/ks
python3 tools/ks_show.py $O/ns_dual_kernel_stats.csv | head -12
