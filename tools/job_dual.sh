for dl in 1 0; do for w in c3 ns; do echo "deal=$dl $(DUAL_DEAL=$dl timeout 600 python3 tools/dual_check.py $w 2>&1 | tail -2 | head -1 | cut -c1-200)"; done; done
