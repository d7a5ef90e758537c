"""Mean counter values per launch of the kernels matching a substring, from rocprofv3 --pmc CSV output directories."""
import csv, glob, sys
from collections import defaultdict
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for k, (c, v) in sorted(acc.items()):
            print(f"{k:32s} launches {c:4d}  mean {v / c:16.1f}")
