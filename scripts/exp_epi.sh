#!/bin/bash
# Where the GELU epilogue's cycles go: flag bits 1 = no GELU arithmetic, 2 = no global stores, 4 (fp8) = no amax / scale
# search (experiment build: python scripts/build_exp.py epi -DCBAS_EXP_EPI).   exp_epi.sh [f8]
export CBAS_EXP_LIB=$GRAFT_REPO_ROOT/cbas_amd/libcbas_exp_epi.so CBAS_STAMP_SHAPES=up
FLAGS="0 1 2 3"
if [ "$1" = f8 ]; then export CBAS_STAMP_F8=1; FLAGS="0 1 2 4 5 7"; fi
for M in 12864 768; do
for f in $FLAGS; do
  echo "== M=$M flags=$f"
  CBAS_EXP_FLAGS=$f python scripts/gemm_stamps.py 13 2000 $M 2>&1 | grep -E "stamps|us rc"
done; done
