#!/bin/bash
# tools/cb_dq4z.sh -- the dQ kernels alone (debug library, AULE_DBG_BWD_ONLY=dq): zero inputs (the schedule without the power cap),
# then real data; checksums of the whole backward against the predecessor
L=build/variants/libaule_dbg.so
SH="4 32 8 4096 4096 128 bf16 1"
for amp in 0 1; do for m in old new; do echo -n "amp=$amp dq=$m: "; CB_AMP=$amp AULE_DBG_BWD_ONLY=dq AULE_HIP_BWD_DQ=$m timeout 30 build/cbench $L bwd $SH 10 3 10 | head -1 | sed 's/.*median/median/'; done; done
for m in old new; do AULE_HIP_BWD_DQ=$m timeout 30 build/cbench aule-attention_amd/aule/lib/libaule.so bwd 2 8 2 1000 1000 128 bf16 1 3 1 | grep "dq:"; done
