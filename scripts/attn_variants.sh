#!/bin/bash
# Same-lease comparison of compile-time variants of precision 4's attention kernel: builds the product library with vit_f32.hip
# compiled under each entry of VARIANTS (one -D... flag set per entry, ';'-separated; the empty entry = the default build), runs
# bench.py --precision 4 on each in ROUNDS alternating rounds and prints the attention kernel's average duration (HIP events around
# every launch, one batch in flight), the frame rate and the CLS error gate (variants must leave the rows where they were).
#   VARIANTS=";-DCBAS_ATTN_PRIO=1;-DCBAS_ATTN_PRIO=2" ROUNDS=2 bash scripts/attn_variants.sh
#   SRC=vit_kernels.hip PRECISION=0 VARIANTS=";-DCBAS_ATTN16_PRIO=1" ...   the fp16 attention kernels instead
# Run on the GPU box from the repo root; the library is restored at the end.  Output: gpurun_out/attn_variants/summary.jsonl
set -e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/attn_variants; mkdir -p $OUT
cp cbas_amd/libcbas_mi355x.so /tmp/cbas_product.so
python -m cbas_amd.build --force --no-asm-check > /dev/null      # the object files do not travel to the GPU box: compile them here
SRC=${SRC:-vit_f32.hip}; STEM=${SRC%.hip}; P=${PRECISION:-4}
OBJS=$(ls cbas_amd/build/*.o | grep -v '\.debug\.o' | grep -v "/$STEM.o" | grep -v api_debug)
IFS=';' read -ra VARS <<< "${VARIANTS:-;-DCBAS_ATTN_PRIO=1;-DCBAS_ATTN_PRIO=2;-DCBAS_ATTN_PRIO=3;-DCBAS_ATTN_PRIO=4}"
[ ${#VARS[@]} -gt 0 ] || VARS=("")
i=0
for v in "${VARS[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I cbas_amd/csrc $v -c cbas_amd/csrc/$SRC -o /tmp/vit_f32_v$i.o 2>> $OUT/build.err
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=cbas_amd/csrc/exports.map -o /tmp/cbas_v$i.so $OBJS /tmp/vit_f32_v$i.o
  i=$((i+1))
done
: > $OUT/summary.jsonl
for round in $(seq 1 ${ROUNDS:-2}); do
  i=0
  for v in "${VARS[@]}"; do
    cp /tmp/cbas_v$i.so cbas_amd/libcbas_mi355x.so
    python bench.py --precision $P --no-label-exact --no-cpu-baseline --no-host-path --files 0 --steps 40 --warmup 3 > $OUT/bench_v${i}_r$round.json 2>> $OUT/bench.err
    python3 -c "
import json; d=json.load(open('$OUT/bench_v${i}_r$round.json')); k=d['roofline']['by_kernel']; g=d.get('gates') or {}
print(json.dumps({'variant': '$v', 'round': $round, 'attention_avg_us': k['attention']['avg_us'], 'layernorm_avg_us': k['layernorm']['avg_us'], 'value': d['value'], 'cls_rel_err_max': g.get('cls_rel_err_max'), 'e2e_long_label_mismatches': (g.get('e2e_long') or {}).get('label_mismatches')}))" | tee -a $OUT/summary.jsonl
    i=$((i+1))
  done
done
cp /tmp/cbas_product.so cbas_amd/libcbas_mi355x.so
