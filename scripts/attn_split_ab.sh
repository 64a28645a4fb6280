#!/bin/bash
# Needs docs/experiments/r05_attention_split_pipelined.patch applied (the pipelined form was measured and NOT adopted: the library
# as committed has the serial form only and ignores CBAS_ATTN_SPLIT_FORM).
# Same-lease A/B of precision 4's attention forms (CBAS_ATTN_SPLIT_FORM=0 serial, 1 software-pipelined): bench.py --precision 4
# alternating, then rocprofv3 kernel stats of each with one batch in flight.  Outputs: gpurun_out/attn_ab/.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_ab
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="--precision 4 --no-label-exact --no-cpu-baseline --no-gates --files 0"
for round in 1 2; do
  for form in 0 1; do
    CBAS_ATTN_SPLIT_FORM=$form python bench.py $B > $OUT/bench_form${form}_r${round}.json 2>> $OUT/bench.err
    python3 -c "import json;d=json.load(open('$OUT/bench_form${form}_r${round}.json'));print('form',$form,'round',$round,d['value'],d['hbm_resident']['value'],d['roofline']['by_kernel']['attention'])"
  done
done
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-path --no-gates --files 0 --preroll-seconds 0 --no-label-exact --lanes 1 --precision 4 --steps 60 --warmup 3"
for form in 0 1; do
  export CBAS_ATTN_SPLIT_FORM=$form
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_form$form -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/prof_form$form.json 2>> $OUT/prof.err
  cp $(find $OUT/stats_form$form -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_form$form.csv
  rm -rf $OUT/stats_form$form
  grep -i "attention" $OUT/kernel_stats_form$form.csv | cut -c1-160
done
