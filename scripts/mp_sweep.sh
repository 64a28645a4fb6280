for mp in 0 30 34 38 40 41 42 43 44 46; do echo "mp=$mp"; CBAS_PP_MAIN_PANELS=$mp python scripts/gemm_tiles.py 17 2>/dev/null | grep -E "^(up|qkv)"; done
