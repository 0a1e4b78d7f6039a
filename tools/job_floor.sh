#!/bin/bash
KZ_FUZZ_FLOOR=0 python3 tools/fuzz_dual.py 12 42 2 8 2>&1 | grep "^ok\|^BAD" | cut -c1-220
KZ_FUZZ_FLOOR=1 python3 tools/fuzz_dual.py 12 42 2 8 2>&1 | grep "^ok\|^BAD" | cut -c1-220
