"""Per-kernel average durations from a rocprofv3 --kernel-trace results .db (or *_kernel_stats.csv)."""
import glob, sqlite3, sys
root = sys.argv[1]
for db in glob.glob(root + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kt} d "
         f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
    rows = list(c.execute(q))
    tot = sum(r[3] for r in rows)
    for name, n, avg, total in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
        print(f"{n:6d} {avg:9.2f} us  {100*total/tot:5.1f}%  {name[:110]}")
