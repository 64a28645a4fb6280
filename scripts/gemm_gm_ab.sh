#!/bin/bash
# Same-lease comparison of the ping-pong GEMMs' tile raster on bench.py: CBAS_GEMM_GM=<n> (n row panels per group, row-fastest
# inside a group; unset = N-fastest) in alternating rounds.  PRECISION (default 4) selects the mode.
#   GMS="0 2 3 4 6" ROUNDS=2 PRECISION=4 bash scripts/gemm_gm_ab.sh          -> gpurun_out/gemm_gm_ab/summary.jsonl
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gemm_gm_ab; mkdir -p $OUT
P=${PRECISION:-4}
: > $OUT/summary.jsonl
for round in $(seq 1 ${ROUNDS:-2}); do
  for gm in ${GMS:-0 2 3 4 6}; do
    V=${GM_VAR:-CBAS_GEMM_GM}                       # GM_VAR=CBAS_GEMM_GM_WIDE: only the shapes with N >= 2048 (precision 4)
    if [ $gm = 0 ]; then unset $V; else export $V=$gm; fi
    python bench.py --precision $P --no-label-exact --no-cpu-baseline --no-host-path --files 0 --steps 40 --warmup 3 > $OUT/bench_p${P}_gm${gm}_r$round.json 2>> $OUT/bench.err
    python3 -c "
import json; d=json.load(open('$OUT/bench_p${P}_gm${gm}_r$round.json')); k=d['roofline']['by_kernel']; g=d.get('gates') or {}
print(json.dumps({'precision': $P, 'group_m': $gm, 'round': $round, 'value': d['value'], 'gemm_us': {n: k[n]['avg_us'] for n in k if 'gemm' in n}, 'cls_rel_err_max': g.get('cls_rel_err_max')}))" | tee -a $OUT/summary.jsonl
  done
done
