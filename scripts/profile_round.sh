#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bench line, rocprofv3 kernel stats of the
# same command (one batch in flight = what the bench's HIP-event pass times; and the default two in
# flight), and the PMC passes.  Outputs land in gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01e}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --lanes 1 > $OUT/bench_prof_lanes1.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_lanes2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_prof_lanes2.json 2>> $OUT/bench_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 > /dev/null 2>&1
find $OUT -name "*.csv" | head -30
