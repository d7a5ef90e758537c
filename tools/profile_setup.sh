#!/bin/bash
# Kernel trace of a few headline solves and the timeline of the last set-up phase.   bash tools/profile_setup.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_setup
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_setup -- python3 $ROOT/tools/scan_probe.py cube_s100k --reps 2 > /tmp/prof_setup.log 2>&1
python3 $ROOT/tools/setup_timeline.py /tmp/prof_setup
