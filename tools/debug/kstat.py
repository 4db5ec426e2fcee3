import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 30]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f}ms {float(r['AverageNs'])/1e3:8.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print('total ms',tot/1e6)
