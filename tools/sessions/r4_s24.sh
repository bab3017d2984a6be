#!/bin/bash
# round 4, GPU session 24: fp16 D = 64 forward with the row sums from the matrix pipe (W4_MSUM): parity files, then kernel against kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_s24; mkdir -p $O
NEW=aule-attention_amd/aule/lib/libaule.so; OLD=build/variants/libaule_premsum.so
{
for sh in "1 2 2 256 256 64 fp16 1" "2 8 8 1111 1111 64 fp16 1" "1 32 1 4096 4096 64 fp16 0"; do
  for lib in $OLD $NEW; do echo "## $lib"; timeout 60 build/cbench $lib fwd $sh 5 2 1 || echo "FAILED rc=$?"; done
done
} > $O/probe.txt 2>&1
grep -E "FAILED|median|o: sum" $O/probe.txt | cut -c1-150
if grep -q FAILED $O/probe.txt; then echo "probe failed; stopping"; exit 0; fi
timeout 1200 python -m pytest tests/test_gpu_fwd.py tests/test_gpu_fwd_variants.py tests/test_gpu_splitkv.py tests/test_gpu_rope.py tests/test_gpu_graph.py -x -q > $O/pytest_fwd.txt 2>&1; tail -3 $O/pytest_fwd.txt
{
for sh in "1 32 1 16384 16384 64 fp16 0" "8 32 32 2048 2048 64 fp16 1" "4 32 8 4096 4096 64 fp16 1" "2 8 8 1111 1111 64 fp16 1" "1 32 32 2048 2048 64 fp16 1" "16 16 16 1024 1024 64 fp16 1"; do
  for lib in $OLD $NEW $OLD $NEW; do echo "## $lib"; timeout 100 build/cbench $lib fwd $sh 20 10 10; done
done
} > $O/cbench_ab.txt 2>&1
grep median $O/cbench_ab.txt | paste - - - - | awk '{print $2,$3,$4,$5,$6,$7,"| old",$9,"new",$(9+13),"old",$(9+26),"new",$(9+39)}'
