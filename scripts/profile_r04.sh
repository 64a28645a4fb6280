#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): the bench lines (default, precision 3, precision 2 at
# batch 64 / 128, cfg4), rocprofv3 kernel stats of the bench command with one batch in flight for the default and for
# precision 3, and the PMC passes for precision 3 (each in its own run, --kernel-trace only).  Outputs: gpurun_out/prof_r04/.
set -e
TAG=r04
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
echo "bench default done"
python bench.py --precision 3 --steps 40 --warmup 3 > $OUT/${TAG}_bench_fp32.json 2>> $OUT/bench.err
echo "bench p3 done"
python bench.py --precision 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8.json 2>> $OUT/bench.err
python bench.py --precision 2 --batch 128 --steps 80 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8_b128.json 2>> $OUT/bench.err
python bench.py --model vitl16 --hw 518 --batch 32 --steps 12 --warmup 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_cfg4_bench.json 2>> $OUT/bench.err
echo "bench lines done"
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --no-gates --files 0 --preroll-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 > $OUT/bench_prof_lanes1.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp32_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 --precision 3 --steps 40 --warmup 3 > $OUT/bench_prof_fp32_lanes1.json 2>> $OUT/bench_prof.err
echo "stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_fp32 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_fp32 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_fp32 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 3 > /dev/null 2>&1
find $OUT -name "*.csv" | wc -l
