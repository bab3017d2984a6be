#!/bin/bash
# tools/ab_variants.sh NAME...  -- same-box A/B of build/variants/libaule_NAME.so against the in-tree library: the headline
# shapes of tools/ps_check.py bench, default first and last (drift check).
cd "$(dirname "$0")/.."
run() { timeout 100 python tools/ps_check.py bench < /dev/null 2>&1 | grep -v amdgpu | tail -8 | awk '{printf "   %-46s %s %s %s\n", $1" "$2" "$3" "$4" "$5" "$6" "$7, $9, $10, $11}'; }
echo "== default"; unset AULE_LIBRARY_PATH; run
for n in "$@"; do echo "== $n"; export AULE_LIBRARY_PATH=$PWD/build/variants/libaule_$n.so; run; done
echo "== default (again)"; unset AULE_LIBRARY_PATH; run
