#!/bin/bash
# PMC passes over bench.py ITSELF (one batch in flight, frames resident, 20 steps): fabric-side bytes per launch and MFMA busy
# cycles of every kernel of the default line's own command - each counter set in its own run, --kernel-trace only.
# Outputs: gpurun_out/prof_pmc_bench/{pmc_fetch,pmc_write,pmc_mfma}/ and pmc_traffic_bench.json (scripts/pmc_traffic.py).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pmc_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P=${PRECISION:-0}       # PRECISION=4: the drop-in's default mode (output: pmc_traffic_bench_p4.json)
Q="--precision $P --no-cpu-baseline --no-host-path --no-gates --no-label-exact --files 0 --lanes 1 --steps 20 --warmup 2 --preroll-seconds 0 --no-kernel-timing"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/bench_fetch.json 2> $OUT/err_fetch.txt
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/bench_write.json 2> $OUT/err_write.txt
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/bench_mfma.json 2> $OUT/err_mfma.txt
echo mfma done
cd $GRAFT_REPO_ROOT
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1); M=$(find $OUT/pmc_mfma -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W $OUT/pmc_traffic_bench$([ $P = 0 ] || echo _p$P).json $M "ViT-B/16, 64 frames 224x224 per step, 2 + 20 steps, one batch in flight (bench.py itself)" "bench.py $Q" | tail -20
# the raw per-dispatch CSVs are large: keep the summary only
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
