#!/bin/bash
# the grouped raster (CBAS_GEMM_GM = row panels per group) at ViT-L/16 518x518 batch 32 shapes
# (M = 32 928; q|k|v N = 3072, up N = 4096, K = 1024): per-kernel time (HIP events, one batch in flight) and fabric fetch
# bytes per launch (rocprofv3 --pmc FETCH_SIZE, its own pass) for each setting.  Output: gpurun_out/raster_cfg4/.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/raster_cfg4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for GM in 1 4 6 8 12; do
  export CBAS_GEMM_GM=$GM
  python3 $GRAFT_REPO_ROOT/bench.py --model vitl16 --hw 518 --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --no-host-path --no-gates --files 0 > $OUT/bench_gm$GM.json 2> $OUT/bench_gm$GM.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_gm$GM -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitl16 32 2 518 > /dev/null 2>&1
  echo "GM=$GM done"
done
python3 $GRAFT_REPO_ROOT/scripts/raster_cfg4_report.py $OUT
