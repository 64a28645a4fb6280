#!/bin/bash
# What does each part of precision 4's attention kernel cost?  Builds the product library with vit_f32.hip compiled under
# -DCBAS_ATTN_ABLATE=<bits> (parts of a key block compiled OUT: results are garbage, only the time means something), runs
# bench.py --precision 4 on each and prints the attention kernel's average duration (HIP events around every launch, one batch
# in flight).  bits: 1 softmax exponentials + hi/lo split, 2 P.V MFMAs + V's LDS reads, 4 V's LDS reads only, 8 S MFMAs + K's LDS
# reads.  Run on the GPU box from the repo root; the library is restored at the end.  Output: gpurun_out/attn_ablate/summary.json
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/attn_ablate; mkdir -p $OUT
cp cbas_amd/libcbas_mi355x.so /tmp/cbas_product.so
python -m cbas_amd.build --force --no-asm-check > /dev/null      # the object files do not travel to the GPU box: compile them here
OBJS=$(ls cbas_amd/build/*.o | grep -v '\.debug\.o' | grep -v 'vit_f32.o' | grep -v api_debug)
echo "[" > $OUT/summary.json
first=1
for a in ${ABLATIONS:-0 1 2 4 8 3 9 10 11 15 31 47 63}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I cbas_amd/csrc -DCBAS_ATTN_ABLATE=$a -c cbas_amd/csrc/vit_f32.hip -o /tmp/vit_f32_ab.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=cbas_amd/csrc/exports.map -o cbas_amd/libcbas_mi355x.so $OBJS /tmp/vit_f32_ab.o
  python bench.py --precision 4 --no-gates --no-label-exact --no-cpu-baseline --no-host-path --files 0 --steps 40 --warmup 3 > $OUT/bench_ab$a.json 2>> $OUT/bench.err
  [ $first = 1 ] || echo "," >> $OUT/summary.json
  first=0
  python3 -c "
import json; d=json.load(open('$OUT/bench_ab$a.json')); k=d['roofline']['by_kernel']
print(json.dumps({'ablate': $a, 'attention_avg_us': k['attention']['avg_us'], 'value': d['value']}))" | tee -a $OUT/summary.json
done
echo "]" >> $OUT/summary.json
cp /tmp/cbas_product.so cbas_amd/libcbas_mi355x.so
