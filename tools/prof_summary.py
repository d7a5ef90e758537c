import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{int(r['TotalDurationNs'])/1e6:9.1f} ms calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%  {r['Name'][:110]}")
