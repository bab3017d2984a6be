#!/bin/bash
# round 4, GPU session 18: fp32 backward with the P / dS arithmetic in groups of four scores in front of their MFMAs (and the
# tile_store template flags) against the build before it; the fp32 parity files
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bwd.py tests/test_gpu_capi.py tests/test_gpu_window.py tests/test_gpu_bottom_right.py -x -q -k "fp32 or f32 or legacy or capi" > $O/pytest_f32.txt 2>&1; tail -3 $O/pytest_f32.txt

AULE_LIBRARY_PATH=$PWD/build/variants/libaule_f32pre.so timeout 300 python tools/f32_bench.py bwd > $O/f32_before.txt 2>&1
timeout 300 python tools/f32_bench.py bwd > $O/f32_after.txt 2>&1
AULE_LIBRARY_PATH=$PWD/build/variants/libaule_f32pre.so timeout 300 python tools/f32_bench.py bwd > $O/f32_before2.txt 2>&1
timeout 300 python tools/f32_bench.py bwd > $O/f32_after2.txt 2>&1
for f in before after before2 after2; do echo "== $f"; grep "fp32" $O/f32_$f.txt; done
