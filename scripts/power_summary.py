"""Summarise scripts/power_probe.sh's rocm-smi samples (gpurun_out/power/smi.jsonl) beside its bench line -> one JSON record.
usage: python scripts/power_summary.py "<command description>" out.json"""
import json, statistics as st, sys
P, S, T = [], [], []
for line in open("gpurun_out/power/smi.jsonl"):
    line = line.strip()
    if not line:
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    for v in d.values():
        try:
            P.append(float(v["Current Socket Graphics Package Power (W)"]))
            S.append(int(v["sclk clock speed:"].strip("()Mhz")))
            T.append(float(v.get("Temperature (Sensor junction) (C)", "nan")))
        except Exception:
            pass
b = json.load(open("gpurun_out/power/bench.json"))
busy = [(p, s) for p, s in zip(P, S) if p > 600]          # samples taken while the loop runs (idle: ~255 W)
out = {"command": sys.argv[1], "samples": len(P), "samples_under_load": len(busy),
       "power_W_under_load": {"mean": round(st.mean(p for p, _ in busy), 1), "min": min(p for p, _ in busy), "max": max(p for p, _ in busy)} if busy else None,
       "sclk_MHz_under_load": {"mean": round(st.mean(s for _, s in busy)), "min": min(s for _, s in busy), "max": max(s for _, s in busy)} if busy else None,
       "junction_C_max": max(T) if T else None, "bench_value_frames_s": b.get("value"), "dtype": b.get("dtype")}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
