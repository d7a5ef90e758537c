"""Which kernels of the process were running when the soak's recorded deviations happened: rocprofv3's kernel trace against the in-kernel
time stamps of tools/probe/sytrd_soak.py (gpurun_out/soak_first_deviations.json).   python tools/probe/soak_correlate.py <trace dir>"""
import csv, glob, json, sys, collections
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("Grid_Size", r.get("Grid_Size_X", ""))))
rows.sort()
victim = [r for r in rows if "k_soak_sytrd" in r[2]]
print(len(rows), "dispatches,", len(victim), "of the soak kernel")
devs = json.load(open("gpurun_out/soak_first_deviations.json"))
seen = collections.Counter()
for d in devs:
    v = victim[d["launch"] - 1]
    t = v[0] + d["pad"] * 10
    print(f"launch {d['launch']} (kernel {v[0]}..{v[1]}, {(v[1] - v[0]) / 1000:.0f} us) deviation at step {d['step']} group {d['group']} +{d['pad'] / 100:.1f} us:")
    names = set()
    for r in rows:
        if r[0] > t + 5000:
            break
        if r[1] >= t - 40000 and "k_soak" not in r[2]:
            print(f"    {(r[0] - t) / 1000:9.1f} .. {(r[1] - t) / 1000:9.1f} us  q{r[3]} lds {r[4]} wg {r[5]} grid {r[6]}  {r[2][:110]}")
            names.add(r[2][:110])
    seen.update(names)
print("kernels near the deviations, by the number of deviations they were near:")
for name, cnt in seen.most_common(40):
    print(f"  {cnt:3d}  {name}")
