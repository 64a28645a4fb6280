#!/bin/bash
# fp16 tile epilogue: 1 slab per pass through the scratch (main build) vs 4 (experiment build ps4)
export CBAS_STAMP_SHAPES=up,qkv
for v in main ps4 main ps4; do
  if [ $v = ps4 ]; then export CBAS_EXP_LIB=$GRAFT_REPO_ROOT/cbas_amd/libcbas_exp_ps4.so; else unset CBAS_EXP_LIB; fi
  echo "== $v"
  python scripts/gemm_stamps.py 13,17 3000 2>&1 | grep -E "tile|stamps|us rc"
done
