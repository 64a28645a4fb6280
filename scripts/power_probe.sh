#!/bin/bash
# Board power / clocks while the bench loop runs (rocm-smi polled every 0.5 s) -> gpurun_out/power/
# usage: scripts/power_probe.sh [extra bench args]
mkdir -p gpurun_out/power
( python bench.py --steps 3000 --no-cpu-baseline --no-gates --no-kernel-timing --files 0 "$@" > gpurun_out/power/bench.json 2> gpurun_out/power/bench.err ) &
BP=$!
sleep 12
for i in $(seq 1 30); do
  rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null >> gpurun_out/power/smi.jsonl || rocm-smi --showpower --showclocks 2>&1 | head -30 >> gpurun_out/power/smi.txt
  echo >> gpurun_out/power/smi.jsonl
  sleep 0.5
  kill -0 $BP 2>/dev/null || break
done
wait $BP
rocm-smi --showmaxpower --showpowercap 2>&1 | head -20 > gpurun_out/power/cap.txt
rocm-smi --showpower --showclocks 2>&1 | head -30 > gpurun_out/power/idle.txt
tail -c 600 gpurun_out/power/bench.json
