#!/bin/bash
# rocprofv3 kernel stats of bench.py --precision 4 with TWO batches in flight (the timed configuration) under the two rasters
# (CBAS_GEMM_GM=1: N-fastest everywhere; unset: the default, groups of 6 for N >= 2048): per-kernel average durations under load.
# Output: gpurun_out/raster_stats/{nfastest,default}_kernel_stats.csv + summary on stdout
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/raster_stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--precision 4 --no-cpu-baseline --no-host-path --no-gates --files 0 --preroll-seconds 0 --no-label-exact --no-kernel-timing --steps 80 --warmup 3"
for arm in nfastest default nfastest2 default2; do
  if [[ $arm == nfastest* ]]; then export CBAS_GEMM_GM=1; else unset CBAS_GEMM_GM; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$arm -- python3 $GRAFT_REPO_ROOT/bench.py $Q > $OUT/bench_$arm.json 2> $OUT/err_$arm.txt
  cp $(find $OUT/$arm -name "*kernel_stats.csv" | head -1) $OUT/${arm}_kernel_stats.csv
  rm -rf $OUT/$arm
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, json, re
out={}
for arm in ("nfastest","default","nfastest2","default2"):
    rows=list(csv.DictReader(open(f"gpurun_out/raster_stats/{arm}_kernel_stats.csv")))
    d={}
    for r in rows:
        m=re.search(r"(gemm_split_pp_kernel<[^>]*>|attention_split_kernel|layernorm_f32_kernel<\d+>)", r["Name"])
        if m: d[m.group(1)]=round(float(r["AverageNs"])/1e3,2)
    d["value"]=json.load(open(f"gpurun_out/raster_stats/bench_{arm}.json"))["value"]
    out[arm]=d
json.dump(out,open("gpurun_out/raster_stats/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
