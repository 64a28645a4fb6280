#!/bin/bash
# Where the GELU epilogue's cycles go: flag bits 1 = no GELU arithmetic, 2 = no global stores (experiment build).
export CBAS_EXP_LIB=$GRAFT_REPO_ROOT/cbas_amd/libcbas_exp_epi.so CBAS_STAMP_SHAPES=up
for M in 12864 768; do
for f in 0 1 2 3; do
  echo "== M=$M flags=$f"
  CBAS_EXP_FLAGS=$f python scripts/gemm_stamps.py 13 2000 $M 2>&1 | grep -E "stamps|us rc"
done; done
