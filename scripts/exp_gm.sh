#!/bin/bash
# Experiment: grouped raster of the ping-pong GEMM (CBAS_GEMM_GM row panels per group).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/exp_gm
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-host-path --no-gates"
for g in 1 3 4 6 8 1; do
  CBAS_GEMM_GM=$g python bench.py $Q > $OUT/gm$g.json 2>> $OUT/err.txt
  python - $OUT/gm$g.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']; k=r['by_kernel']
print(sys.argv[1].split('/')[-1], d['value'], r['achieved'], {n:k[n]['avg_us'] for n in ('qkv_gemm','oproj_gemm','up_gemm','down_gemm')})
PY
done
cd /tmp && export TMPDIR=/tmp
export CBAS_GEMM_GM=6
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_gm6 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
