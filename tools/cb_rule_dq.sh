#!/bin/bash
# tools/cb_rule.sh -- dQ dispatch rule: both dQ kernels forced on a spread of shapes (whole backward, tools/cbench.cpp, batches of 5)
L=aule-attention_amd/aule/lib/libaule.so
while read -r name shape; do
  for mode in new old; do
    echo -n "$name dq=$mode: "
    AULE_HIP_BWD_DQ=$mode timeout 60 build/cbench $L bwd $shape 8 3 5 | head -1 | sed 's/.*median/median/'
  done
done <<'S'
mha_c_B4H32S4096      4 32 32 4096 4096 128 bf16 1
mha_c_B16H16S1024     16 16 16 1024 1024 128 bf16 1
mha_nc_B2H16S4096     2 16 16 4096 4096 128 bf16 0
gqa_nc_B8_32_4_S2048  8 32 4 2048 2048 128 bf16 0
gqa_c_B2_64_8_S8192   2 64 8 8192 8192 128 bf16 1
mha_c_B1H16S4096      1 16 16 4096 4096 128 bf16 1
mha_c_B2H16S2048      2 16 16 2048 2048 128 bf16 1
fp16_gqa_c_B4_32_8    4 32 8 2048 2048 128 fp16 1
mha_nc_B1H32S2048     1 32 32 2048 2048 128 bf16 0
ragged_c_B2H16S3000   2 16 16 3000 3000 128 bf16 1
S
