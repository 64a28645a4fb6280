"""Table for scripts/raster_cfg4.sh: per raster setting, avg us (HIP events) and fabric fetch MB per launch of the q|k|v / up /
o_proj+down GEMMs."""
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]
rows = []
for gm in (1, 4, 6, 8, 12):
    bj = os.path.join(out, f"bench_gm{gm}.json")
    if not os.path.exists(bj):
        continue
    d = json.loads(open(bj).read().strip().split("\n")[-1])
    bk = d["roofline"]["by_kernel"]
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(out, f"pmc_fetch_gm{gm}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            if r["Counter_Name"] != "FETCH_SIZE":
                continue
            m = re.search(r"gemm_f16_8ph_kernelILi(\d)E", r["Kernel_Name"]) or re.search(r"gemm_f16_8ph_kernel<(\d)", r["Kernel_Name"])
            if m:
                k = {"1": "qkv", "2": "resid", "3": "up", "0": "patch"}.get(m.group(1), m.group(1))
                tot[k] += float(r["Counter_Value"]); n[k] += 1
    mb = {k: tot[k] / n[k] * 1024 * 2 / 1e6 for k in tot}
    rows.append({"group_m": gm, "value_fps": d["value"], "qkv_us": bk["qkv_gemm"]["avg_us"], "up_us": bk["up_gemm"]["avg_us"],
                 "oproj_us": bk["oproj_gemm"]["avg_us"], "down_us": bk["down_gemm"]["avg_us"],
                 "gemm_tflops": d["roofline"]["achieved"], "fetch_mb": {k: round(v, 1) for k, v in mb.items()}})
    print(rows[-1])
json.dump({"workload": "ViT-L/16, 518x518, batch 32 (M = 32 928): bench.py --lanes via HIP events; FETCH_SIZE x 1024 x 2 per launch "
                       "(scripts/quick_perf.py vitl16 32 2 518); CBAS_GEMM_GM = row panels per raster group (1 = N-fastest)",
           "rows": rows}, open(os.path.join(out, "raster_cfg4.json"), "w"), indent=1)
