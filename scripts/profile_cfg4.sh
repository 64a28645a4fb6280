#!/bin/bash
# cfg4 (ViT-L/16, 518x518, batch 32) evidence set: GPU test, bench line, rocprofv3 kernel stats with one batch in
# flight (what the HIP-event pass times), and the MFMA-busy / traffic PMC passes on a 2-iteration loop.
set -e
TAG=${1:-r03_cfg4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --model vitl16 --hw 518 --batch 32 --steps 12 --warmup 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py --model vitl16 --hw 518 --batch 32 --steps 12 --warmup 2 --no-cpu-baseline --no-host-path --no-gates --files 0 --lanes 1 > $OUT/bench_prof_lanes1.json 2> $OUT/bench_prof.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitl16 32 2 518 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitl16 32 2 518 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitl16 32 2 518 > /dev/null 2>&1
find $OUT -name "*.csv" | head -30
