#!/bin/bash
# Fabric-side fetch per launch of precision 4's GEMM kernels under the two rasters (CBAS_GEMM_GM=1: N-fastest, 6: the default since
# r5): rocprofv3 --pmc FETCH_SIZE --kernel-trace over scripts/quick_perf.py vitb16 64 3 224 4, one run per raster.
# Output: gpurun_out/pmc_raster/summary.json (KiB-units of FETCH_SIZE summed over the 8 XCD instances, averaged per launch and kernel)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_raster; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for gm in 1 6; do
  export CBAS_GEMM_GM=$gm
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/gm$gm -- python3 $GRAFT_REPO_ROOT/scripts/quick_perf.py vitb16 64 3 224 4 > $OUT/run_gm$gm.txt 2> $OUT/err_gm$gm.txt
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections, os, re
out={}
for gm in (1,6):
    f=glob.glob(f'gpurun_out/pmc_raster/gm{gm}/**/*counter_collection.csv', recursive=True)[0]
    acc=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']!='FETCH_SIZE': continue
        m=re.search(r'(gemm_split_\w+<[^>]*>|attention_split_kernel|layernorm_f32_kernel<\d+>)', r['Kernel_Name'])
        if not m: continue
        k=m.group(1)
        acc[k][0]+=float(r['Counter_Value']); acc[k][1]+=1
    out[f'group_m_{gm}']={k:{'launches':n,'FETCH_SIZE_per_launch':round(v/n,1)} for k,(v,n) in acc.items()}
json.dump(out,open('gpurun_out/pmc_raster/summary.json','w'),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $OUT/gm1 $OUT/gm6
