#!/bin/bash
# Round-4 precision-4 profile set after the split GEMMs moved to the ping-pong kernel (run on the GPU box from the repo
# root): bench lines (default with its label_exact leg, precision 4, cfg4 shape at precision 4, MX-fp8 at batch 64 / 128 as
# a regression check of the shared kernel file), rocprofv3 kernel stats of the precision-4 bench with one batch in flight,
# MFMA busy / clock PMC pass, block timelines of the four GEMM shapes.  Outputs: gpurun_out/prof_r04p4/.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04p4
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/r04_bench.json 2> $OUT/bench.err
echo "bench default done"
python bench.py --precision 4 --no-label-exact > $OUT/r04_bench_p4.json 2>> $OUT/bench.err
echo "bench p4 done"
python bench.py --model vitl16 --hw 518 --batch 32 --steps 12 --warmup 2 --no-cpu-baseline --files 0 --precision 4 --no-label-exact > $OUT/r04_cfg4_bench_p4.json 2>> $OUT/bench.err
python bench.py --precision 2 --no-cpu-baseline --files 0 --no-label-exact > $OUT/r04_bench_fp8.json 2>> $OUT/bench.err
python bench.py --precision 2 --batch 128 --steps 80 --no-cpu-baseline --files 0 --no-label-exact > $OUT/r04_bench_fp8_b128.json 2>> $OUT/bench.err
echo "bench lines done"
python scripts/split_stamps.py 0,256,160 20 > $OUT/r04_split_gemm_stamps.txt 2>&1
CBAS_SOAK_PRECISION=4 python scripts/soak_files.py 4 $OUT/r04_soak_files_p4.json > $OUT/soak_p4.log 2>&1
echo "stamps + soak done"
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --no-gates --files 0 --preroll-seconds 0 --no-label-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p4_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 --precision 4 --steps 80 --warmup 3 > $OUT/bench_prof_p4_lanes1.json 2> $OUT/bench_prof.err
echo "stats done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_p4 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 4 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_mfma.py $OUT/pmc_mfma_p4 $OUT/r04_pmc_mfma_p4.json "ViT-B/16 precision 4, 64 frames 224x224 per launch set (scripts/quick_perf.py vitb16 64 3 224 4)" > /dev/null
cp $(find $OUT/stats_p4_lanes1 -name "*kernel_stats.csv" | head -1) $OUT/r04_bench_p4_lanes1_kernel_stats.csv
rm -rf $OUT/stats_p4_lanes1 $OUT/pmc_mfma_p4
ls $OUT
