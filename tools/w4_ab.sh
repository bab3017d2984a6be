#!/bin/bash
# Same-box A/B of forward libraries: tools/w4_ab.sh LIB_A LIB_B [...]   (paths; "default" = the in-tree library)
# Alternates the libraries over three rounds of the headline shapes (conditioned launches, HIP events) and prints medians.
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset AULE_LIBRARY_PATH; else export AULE_LIBRARY_PATH=$lib; fi
    echo "== round $round lib $lib"
    timeout 300 python - <<'PY' 2>&1 | grep "bf16 B"
import sys
sys.path.insert(0, "tools")
import fwd_check as pc
pc.bench_shape("bf16", 4, 32, 32, 4096, 128, True, warm=150, iters=100)
pc.bench_shape("bf16", 4, 32, 32, 4096, 128, False, warm=80, iters=60)
pc.bench_shape("bf16", 4, 32, 8, 2048, 128, True, warm=200, iters=100)
PY
  done
done
