import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
steps=int(sys.argv[2])
print("total %.1f ms/step"%(tot/1e6/steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv)>3 else 22]:
    print("%6.2f ms/step %5.1f%% calls/step %6.1f avg %8.1f us  %s"%(int(r['TotalDurationNs'])/1e6/steps, float(r['Percentage']), int(r['Calls'])/steps, float(r['AverageNs'])/1e3, r['Name'][:86]))
