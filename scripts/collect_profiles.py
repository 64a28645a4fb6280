"""Copy a profile_round.sh / profile_cfg4.sh output set from gpurun_out/ into profiles/ under the round's names and print
the recomputed per-kernel table (avg us -> TF/s) for profiles/README.md.

    python scripts/collect_profiles.py r02c r02c_cfg4 r02
"""
import csv, glob, json, os, re, shutil, subprocess, sys

src, src4, dst = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "prof_" + src)
G4 = os.path.join(ROOT, "gpurun_out", "prof_" + src4)
P = os.path.join(ROOT, "profiles")


def one(pattern):
    m = glob.glob(pattern)
    assert len(m) == 1, (pattern, m)
    return m[0]


def cp(a, b):
    shutil.copyfile(a, os.path.join(P, b))
    print("profiles/" + b)


for name in ("bench", "bench_fp8", "bench_fp8_b128"):
    cp(os.path.join(G, f"{src}_{name}.json"), f"{dst}_{name}.json")
cp(one(G + "/stats_lanes1/*/*kernel_stats.csv"), f"{dst}_bench_lanes1_kernel_stats.csv")
cp(one(G + "/stats_lanes2/*/*kernel_stats.csv"), f"{dst}_bench_lanes2_kernel_stats.csv")
cp(one(G + "/stats_fp8_lanes1/*/*kernel_stats.csv"), f"{dst}_bench_fp8_lanes1_kernel_stats.csv")
py = [sys.executable, os.path.join(ROOT, "scripts", "pmc_traffic.py")]
subprocess.check_call(py + [one(G + "/pmc_fetch/*/*counter_collection.csv"), one(G + "/pmc_write/*/*counter_collection.csv"),
                            os.path.join(P, "pmc_traffic.json"), one(G + "/pmc_mfma/*/*counter_collection.csv")])
subprocess.check_call(py + [one(G + "/pmc_fetch_fp8/*/*counter_collection.csv"), one(G + "/pmc_write_fp8/*/*counter_collection.csv"),
                            os.path.join(P, f"{dst}_pmc_traffic_fp8.json"), one(G + "/pmc_mfma_fp8/*/*counter_collection.csv"),
                            "ViT-B/16 precision 2 (MX-fp8), 64 frames 224x224 per launch set (scripts/quick_perf.py vitb16 64 3 224 2)"])
if os.path.isdir(G4):
    cp(os.path.join(G4, f"{src4}_bench.json"), f"{dst}_cfg4_bench.json")
    cp(one(G4 + "/stats_lanes1/*/*kernel_stats.csv"), f"{dst}_cfg4_vitl16_518_b32_lanes1_kernel_stats.csv")
    subprocess.check_call(py + [one(G4 + "/pmc_fetch/*/*counter_collection.csv"), one(G4 + "/pmc_write/*/*counter_collection.csv"),
                                os.path.join(P, f"{dst}_cfg4_pmc_traffic.json"), one(G4 + "/pmc_mfma/*/*counter_collection.csv"),
                                "ViT-L/16, 32 frames 518x518 per launch set (scripts/quick_perf.py vitl16 32 2 518)"])


def table(path, M, D, F, label):
    rows = {}
    for r in csv.DictReader(open(path)):
        rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
    fl = {"1": 2.0 * M * D * 3 * D, "2": None, "3": 2.0 * M * D * F, "4": 2.0 * M * D * F}
    print(f"--- {label}")
    tot_t = tot_f = 0.0
    for name, (calls, us, pct) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
        m = re.search(r"gemm_f16_8ph_kernel<(\d), (\d), (\d), (\d), (\w+)>", name)
        short = re.sub(r"\(anonymous namespace\)::|void |\(GemmParams.*", "", name)[:70]
        if m and m.group(1) in ("1", "2", "3", "4") and calls > 500:
            e = m.group(1)
            f = fl[e] if fl[e] else (2.0 * M * D * D + 2.0 * M * D * F) / 2      # o_proj + down averaged
            n_per_layer = 2 if e == "2" else 1
            tot_t += us * n_per_layer; tot_f += f * n_per_layer
            print(f"{short:72s} calls {calls:6d} avg {us:8.2f} us {pct:6.2f} %  -> {f / us / 1e6:7.0f} TF/s")
        elif pct > 0.5:
            print(f"{short:72s} calls {calls:6d} avg {us:8.2f} us {pct:6.2f} %")
    if tot_t:
        print(f"family: {tot_f / 1e9:.1f} GFLOP / {tot_t:.1f} us = {tot_f / tot_t / 1e6:.0f} TF/s")


table(os.path.join(P, f"{dst}_bench_lanes1_kernel_stats.csv"), 12864, 768, 3072, "fp16 lanes 1")
table(os.path.join(P, f"{dst}_bench_fp8_lanes1_kernel_stats.csv"), 12864, 768, 3072, "fp8 lanes 1")
if os.path.isdir(G4):
    table(os.path.join(P, f"{dst}_cfg4_vitl16_518_b32_lanes1_kernel_stats.csv"), 32 * 1029, 1024, 4096, "cfg4 lanes 1")
