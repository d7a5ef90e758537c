#!/bin/bash
# Kernel-time distribution of one workload's solve (rocprofv3 --kernel-trace --stats), top rows printed and the csv kept in gpurun_out/.
#   bash tools/profile_workload.sh config3_s30k [reps]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
W=$1; REPS=${2:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$W
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$W -- python3 $ROOT/tools/scan_probe.py $W --reps $REPS > /tmp/prof_$W.log 2>&1
tail -1 /tmp/prof_$W.log
F=$(find /tmp/prof_$W -name '*kernel_stats.csv' | head -1)
mkdir -p $ROOT/gpurun_out && cp $F $ROOT/gpurun_out/kernel_stats_$W.csv
python3 - $F <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    print("%-72s %6s calls %9.1f us avg %5.1f %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("kernel time in all: %.1f ms" % (tot / 1e6))
P
