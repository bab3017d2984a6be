#!/bin/bash
# parity list + timelines + A/B timing of the one-wave-per-SIMD forward; writes under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python tools/w4_check.py check ${1:-quick} > gpurun_out/w4_check.log 2>&1
echo "check rc=$?"; grep -v "^ok" gpurun_out/w4_check.log | tail -20; grep -c "^ok" gpurun_out/w4_check.log
timeout 300 python tools/timeline_w4.py 0 4 32 4096 0 3 > gpurun_out/w4_tl_noncausal.txt 2>&1; grep -A6 "wave 3" gpurun_out/w4_tl_noncausal.txt | head -12
timeout 300 python tools/timeline_w4.py 1 4 32 4096 0 3 > gpurun_out/w4_tl_causal.txt 2>&1; grep -A12 "wave 3" gpurun_out/w4_tl_causal.txt | head -16
timeout 600 python tools/w4_check.py bench w4 > gpurun_out/w4_bench.log 2>&1
echo "bench rc=$?"; cat gpurun_out/w4_bench.log
