#!/bin/bash
# Pure-MFMA sustained rate and board power (scripts/probes/probe_mfma_power.hip) -> gpurun_out/mfma_power/
# build first (in the container): hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/probe_mfma_power scripts/probes/probe_mfma_power.hip
mkdir -p gpurun_out/mfma_power; O=gpurun_out/mfma_power
for shape in 0 1 2; do
  ( scripts/probes/bin/probe_mfma_power $shape 10 2 > $O/rate_$shape.txt 2>&1 ) &
  BP=$!
  sleep 3
  : > $O/smi_$shape.jsonl
  for i in $(seq 1 12); do rocm-smi --showpower --showclocks --json 2>/dev/null >> $O/smi_$shape.jsonl; echo >> $O/smi_$shape.jsonl; sleep 0.5; kill -0 $BP 2>/dev/null || break; done
  wait $BP
  cat $O/rate_$shape.txt
  python3 - $O/smi_$shape.jsonl <<'PY'
import json, sys, statistics as st
P=[];S=[]
for line in open(sys.argv[1]):
    line=line.strip()
    if not line: continue
    try: d=json.loads(line)
    except Exception: continue
    for k,v in d.items():
        P.append(float(v["Current Socket Graphics Package Power (W)"])); S.append(int(v["sclk clock speed:"].strip("()Mhz")))
print("   power mean %.0f W (min %.0f max %.0f), sclk mean %.0f MHz, %d samples" % (st.mean(P), min(P), max(P), st.mean(S), len(P)))
PY
done
