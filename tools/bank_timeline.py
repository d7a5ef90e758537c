"""One block of the bank on the device's clock: from a rocprofv3 --kernel-trace CSV of tools/bank_bench.py, the kernels of the
last all-live-sized block (the one with the longest k_bank_modes), with start offsets, durations and the idle gaps between them.
usage: bank_timeline.py <dir>"""
import csv, glob, os, re, sys
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]))[:60]
modes = [i for i, r in enumerate(rows) if "k_bank_modes" in r["Kernel_Name"]]
best = max(modes, key=lambda i: int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
# the block: from the previous block's mix to this block's mix (the last kernel of a block)
lo = best
while lo > 0 and "k_bank_mix" not in rows[lo - 1]["Kernel_Name"]: lo -= 1
hi = best
while hi < len(rows) - 1 and "k_bank_mix" not in rows[hi]["Kernel_Name"]: hi += 1
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = int(rows[lo - 1]["End_Timestamp"]) if lo else t0
print("since the previous block's last kernel ended: %.1f us" % ((t0 - prev_end) / 1e3))
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  +{(s - t0) / 1e3:7.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {name(r)}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}")
    prev_end = e
print("block on the device: %.1f us" % ((prev_end - t0) / 1e3))
