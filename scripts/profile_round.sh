#!/bin/bash
# Round profile set (run on the GPU box from the repo root): the bench line, rocprofv3 kernel stats of the same
# command with one batch in flight (= what the bench's HIP-event pass times) and with the default two, the PMC
# passes (each in its own run, --kernel-trace only), and the same for the MX-fp8 mode (precision 2).
# Outputs land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python bench.py --precision 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8.json 2>> $OUT/bench.err
python bench.py --precision 2 --batch 128 --steps 80 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8_b128.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --no-gates --files 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 > $OUT/bench_prof_lanes1.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes2 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --no-kernel-timing > $OUT/bench_prof_lanes2.json 2>> $OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fp8_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py $Q --lanes 1 --precision 2 > $OUT/bench_prof_fp8_lanes1.json 2>> $OUT/bench_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_fp8 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_fp8 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_fp8 -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 2 > /dev/null 2>&1
find $OUT -name "*.csv" | wc -l
