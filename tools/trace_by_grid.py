"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid size): launches, mean / min duration.  Separates the P2-
and P1-level launches of the same kernel template.  usage: trace_by_grid.py <dir> [min_total_ms]"""
import csv, glob, os, re, sys
from collections import defaultdict
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:80]
    acc[(name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(sum(v) for v in acc.values())
print("total %.1f ms" % (tot / 1e3))
for (name, grid), v in rows:
    if sum(v) / 1e3 < lim: break
    print(f"{sum(v)/1e3:8.2f} ms  x{len(v):5d}  mean {sum(v)/len(v):8.1f} us  min {min(v):8.1f}  grid {grid:9d}  {name}")
