#!/bin/bash
# Same-lease A/B of two builds of the product library on bench.py (library files swapped, alternating): scripts/lib_ab.sh OLD.so TAG
# Outputs: gpurun_out/lib_ab_TAG/{old,new}_{p0,p4}_r{1,2}.json and a summary on stdout.
set -e
OLD=$1; TAG=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/lib_ab_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cp cbas_amd/libcbas_mi355x.so /tmp/cbas_new.so
B="--no-label-exact --no-cpu-baseline --files 0"
for round in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp $OLD cbas_amd/libcbas_mi355x.so; else cp /tmp/cbas_new.so cbas_amd/libcbas_mi355x.so; fi
    python bench.py $B > $OUT/${which}_p0_r$round.json 2>> $OUT/bench.err
    python bench.py $B --precision 4 --steps 80 > $OUT/${which}_p4_r$round.json 2>> $OUT/bench.err
    python3 -c "
import json
for p in ('p0','p4'):
    d=json.load(open('$OUT/${which}_'+p+'_r$round.json')); k=d['roofline']['by_kernel']
    print('$which',p,'round $round',d['value'],d['hbm_resident']['value'],'LN us',k['layernorm']['avg_us'],'cls',d['gates']['cls_rel_err_max'],'e2e_long',d['gates']['e2e_long']['label_mismatches'],d['gates']['e2e_long'].get('fp16_elements_differing_pct'))"
  done
done
cp /tmp/cbas_new.so cbas_amd/libcbas_mi355x.so
