cd "$GRAFT_REPO_ROOT"
for env in "" "AULE_HIP_BWD_DKV=old AULE_HIP_BWD_DQ=old AULE_HIP_BWD_MODE=recompute AULE_HIP_W4_UNPAIR=0 AULE_HIP_F32_SPLIT=0"; do
  echo "== env: $env"
  for c in "bf16 1 4 4 2 3000 64 none 100 1.0 1004" "bf16 3 4 2 7 65 128 top -1 none 1097" "bf16 9 1 1 2 1025 32 top 100 0.3 1265" "bf16 1 4 1 2 1025 32 top -1 none 1263"; do
    env $env python tools/fuzz_parity.py one $c 2>&1 | grep route
  done
done
