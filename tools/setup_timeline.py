"""Timeline of the hierarchy set-up (`factorize`) of the LAST solve in a rocprofv3 --kernel-trace: from its first k_shift_values to the
first LOBPCG kernel after the coarse elimination, per stream (queue): busy time, kernel list in order with start offsets.
    python tools/setup_timeline.py <trace dir>"""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = []
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:60]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[3].startswith("k_shift_values")]
i0 = starts[-2] if len(starts) >= 2 and starts[-1] - starts[-2] < 4 else starts[-1]  # (two levels: two launches per set-up)
t0 = rows[i0][0]
# the elimination ends with its k_symmetrize_lower on the side stream
end = next(i for i in range(i0, len(rows)) if rows[i][3].startswith("k_symmetrize_lower"))
t1 = rows[end][1]
print("set-up window %.2f ms" % ((t1 - t0) / 1e6))
queues = {}
for s, e, q, n in rows[i0:end + 1]:
    queues.setdefault(q, []).append((s, e, n))
for q, ks in queues.items():
    busy = sum(e - s for s, e, _ in ks)
    print("queue %s: %d kernels, busy %.2f ms, from %.2f to %.2f ms" % (q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6))
    agg = {}
    for s, e, n in ks:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("    %-60s x%4d %9.1f us" % (n, c, us))
