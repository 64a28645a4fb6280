"""Per-kernel MFMA busy share and shader clock from one rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 <workload>
usage: pmc_mfma.py DIR OUT.json "workload text"
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1 024 matrix pipes (256 CUs x 4 SIMDs); GRBM_GUI_ACTIVE sums the
active cycles of the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration, busy share = MFMA cycles / (1024 x clock x duration)."""
import csv, glob, json, re, sys
from collections import defaultdict
d, out, workload = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
busy, gui, dur, n = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f, newline="")):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += float(r["Counter_Value"])
        dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[k] += float(r["Counter_Value"])
per = {}
for k in sorted(busy, key=lambda k: -dur[k])[:10]:
    clock = gui[k] / 8 / dur[k]                       # cycles per ns = GHz
    per[k] = {"launches": n[k], "avg_us_under_pmc": round(dur[k] / n[k] / 1e3, 1), "clock_ghz": round(clock, 2),
              "mfma_busy_of_all_simds": round(busy[k] / (1024 * gui[k] / 8), 3) if gui[k] else None}
json.dump({"workload": workload, "method": __doc__.split("usage")[0].strip().splitlines()[1].strip() + "; clock = GRBM_GUI_ACTIVE / 8 / duration; "
           "busy share against the cycles the chip actually ran", "per_kernel": per}, open(out, "w"), indent=1)
print(json.dumps(per, indent=1))
