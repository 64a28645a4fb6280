"""Per-kernel averages of every counter in a rocprofv3 --pmc counter_collection CSV, for kernels matching a substring."""
import csv, sys
from collections import defaultdict
path, pat = sys.argv[1], sys.argv[2]
tot, n, dur = defaultdict(float), defaultdict(int), defaultdict(float)
for row in csv.DictReader(open(path, newline="")):
    if pat not in row["Kernel_Name"]:
        continue
    c = row["Counter_Name"]
    tot[c] += float(row["Counter_Value"]); n[c] += 1
    dur[c] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for c in sorted(tot):
    print(f"{c:32s} avg {tot[c]/n[c]:16.1f}   launches {n[c]}   avg dur {dur[c]/n[c]/1e3:8.1f} us")
