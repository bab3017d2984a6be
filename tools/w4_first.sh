#!/bin/bash
# First contact of the one-wave-per-SIMD forward with the GPU: probe, one tiny shape (short timeout: a barrier mismatch hangs),
# then the parity list and an A/B timing against the predecessor.  Writes under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 120 ./build/probes/probe_trread > gpurun_out/r3_trread_probe.txt 2>&1
echo "probe rc=$?"; cat gpurun_out/r3_trread_probe.txt
timeout 400 python - > gpurun_out/w4_tiny.log 2>&1 <<'PY'
import sys
sys.path.insert(0, "tools")
import ps_check as pc
ok = pc.check("bf16", 1, 2, 2, 256, 256, 128, False, want_route=8)
ok &= pc.check("bf16", 1, 2, 2, 256, 256, 128, True, want_route=8)
sys.exit(0 if ok else 1)
PY
rc=$?
echo "tiny rc=$rc"; tail -5 gpurun_out/w4_tiny.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "tiny run did not finish: stopping"; exit 1; fi
timeout 900 python tools/w4_check.py check ${1:-quick} > gpurun_out/w4_check.log 2>&1
echo "check rc=$?"; tail -40 gpurun_out/w4_check.log
timeout 600 python tools/w4_check.py bench w4 > gpurun_out/w4_bench.log 2>&1
echo "bench rc=$?"; cat gpurun_out/w4_bench.log
AULE_HIP_FWD_KERNEL=ps timeout 600 python tools/w4_check.py bench ps > gpurun_out/w4_bench_ps.log 2>&1
echo "bench ps rc=$?"; cat gpurun_out/w4_bench_ps.log
