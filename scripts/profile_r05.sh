#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root, one lease): the bench lines on the PRODUCT library (default
# line with its label_exact leg, labels_identical, value_r3_definition and the ViT-S cfg1 CPU baseline; precision 4; the
# driver's own --steps 20 --warmup 5 form), the two-rank self-launch (gloo rehearsal on this box's one GPU; the RCCL form
# must refuse), rocprofv3 kernel stats of the default and the precision-4 bench with one batch in flight.
# Outputs: gpurun_out/prof_r05/.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/r05_bench.json 2> $OUT/bench.err
echo "bench default done"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_steps20.json 2>> $OUT/bench.err
python bench.py --precision 4 --no-label-exact --no-cpu-baseline > $OUT/r05_bench_p4.json 2>> $OUT/bench.err
echo "bench p4 done"
CBAS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_gpus2_gloo.json 2>> $OUT/bench.err
echo "gloo 2-rank rc=$?"
set +e
python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/r05_bench_gpus2_rccl_one_device.json 2>> $OUT/bench.err
echo "rccl 2-rank on one device rc=$? (2 expected)"
set -e
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --no-gates --files 0 --preroll-seconds 0 --no-label-exact"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 > $OUT/bench_prof_lanes1.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p4_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 --precision 4 --steps 80 --warmup 3 > $OUT/bench_prof_p4_lanes1.json 2>> $OUT/bench_prof.err
cp $(find $OUT/stats_lanes1 -name "*kernel_stats.csv" | head -1) $OUT/r05_bench_lanes1_kernel_stats.csv
cp $(find $OUT/stats_p4_lanes1 -name "*kernel_stats.csv" | head -1) $OUT/r05_bench_p4_lanes1_kernel_stats.csv
rm -rf $OUT/stats_lanes1 $OUT/stats_p4_lanes1
ls $OUT
