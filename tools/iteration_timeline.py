"""One LOBPCG iteration of the LAST solve in a rocprofv3 --kernel-trace, kernel by kernel: from the n-th last k_sytrd_regs to the next one
(the Rayleigh-Ritz step's tridiagonalisation runs once per iteration) -- start offset, duration, the gap to the previous kernel's end,
queue, name; then the sums (busy, gaps) of the window.      python tools/iteration_timeline.py <trace dir> [which = 6 (from the end)] [marker kernel]"""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 6
marker = sys.argv[3] if len(sys.argv) > 3 else "k_sytrd_regs"  # (k_sytrd_wide for the 215-pair workloads)
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name[:90]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[3].startswith(marker)]
i0, i1 = marks[-which - 1], marks[-which]
t0 = rows[i0][0]
prev_end = t0
busy = gaps = 0.0
for s, e, q, n in rows[i0:i1]:
    gap = (s - prev_end) / 1e3
    print("%9.1f us  %8.1f us  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n))
    busy += (e - s) / 1e3
    if gap > 0: gaps += gap
    prev_end = max(prev_end, e)
print("window %.1f us: %d kernels, busy %.1f us, idle gaps %.1f us" % ((rows[i1][0] - t0) / 1e3, i1 - i0, busy, gaps))
