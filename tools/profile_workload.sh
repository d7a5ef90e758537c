#!/bin/bash
# Kernel-time distribution of a solve (rocprofv3 --kernel-trace --stats), top rows printed and the csv kept in gpurun_out/.
#   bash tools/profile_workload.sh TAG REPS workload [workload ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; REPS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $ROOT/tools/scan_probe.py "$@" --reps $REPS > /tmp/prof_$TAG.log 2>&1
tail -1 /tmp/prof_$TAG.log | cut -c1-260
F=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
mkdir -p $ROOT/gpurun_out && cp $F $ROOT/gpurun_out/kernel_stats_$TAG.csv
python3 - $F <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print("%-72s %6s calls %9.1f us avg %5.1f %%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("kernel time in all: %.1f ms" % (tot / 1e6))
P
