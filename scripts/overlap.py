"""How much of the wall time has a GEMM running, a non-GEMM kernel running, both, or neither - from a rocprofv3
kernel trace of bench.py (two batches in flight):   python scripts/overlap.py <kernel_trace.csv>"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    kind = "gemm" if "gemm_f16_8ph" in n else ("attn" if "attention" in n else ("ln" if "layernorm" in n else "other"))
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
ev.sort()
t0, t1 = ev[int(len(ev) * 0.3)][0], ev[int(len(ev) * 0.9)][0]       # steady state
pts = []
for s, e, k in ev:
    if e < t0 or s > t1:
        continue
    pts += [(max(s, t0), 1, k), (min(e, t1), -1, k)]
pts.sort()
cnt = {"gemm": 0, "attn": 0, "ln": 0, "other": 0}
acc, last = {}, t0
for t, d, k in pts:
    key = (min(cnt["gemm"], 2), "attn" if cnt["attn"] else ("ln" if cnt["ln"] else ("other" if cnt["other"] else "-")))
    acc[key] = acc.get(key, 0) + (t - last)
    last = t
    cnt[k] += d
for k, v in sorted(acc.items()):
    print(f"GEMMs running {k[0]}  with {k[1]:6s} {100.0 * v / (t1 - t0):5.1f} %")
