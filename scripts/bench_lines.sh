#!/bin/bash
# The four bench lines of a round's profile set (no profiler): fp16 default, MX-fp8 batch 64 / 128, cfg4.  bench_lines.sh TAG
TAG=${1:-r03}; OUT=$GRAFT_REPO_ROOT/gpurun_out/lines_$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python bench.py --precision 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8.json 2>> $OUT/bench.err
python bench.py --precision 2 --batch 128 --steps 80 --no-cpu-baseline --files 0 > $OUT/${TAG}_bench_fp8_b128.json 2>> $OUT/bench.err
python bench.py --model vitl16 --hw 518 --batch 32 --steps 12 --warmup 2 --no-cpu-baseline --files 0 > $OUT/${TAG}_cfg4_bench.json 2>> $OUT/bench.err
python - <<PY
import json
for f in ("bench","bench_fp8","bench_fp8_b128","cfg4_bench"):
    d=json.loads(open("$OUT/${TAG}_%s.json" % f).read().strip().split("\n")[-1])
    print(f, "value", d["value"], "hbm", d["hbm_resident"]["value"], "files", (d.get("files_path") or {}).get("value"), "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
