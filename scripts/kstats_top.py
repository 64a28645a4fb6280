import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.reader(open(f)))
for r in rows[1:int(sys.argv[2]) if len(sys.argv)>2 else 9]:
    print(f"{r[0][:90]:90s} calls {r[1]:>6s} avg {float(r[3])/1e3:9.2f} us  {r[4]}%")
