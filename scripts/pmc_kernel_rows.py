"""Per-kernel means of the counters of a rocprofv3 --pmc pass: pmc_kernel_rows.py DIR [name-substring ...]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if want and not any(w in k for w in want):
        continue
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})")
