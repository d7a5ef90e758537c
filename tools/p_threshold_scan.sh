#!/bin/bash
# iteration counts / times for the P-start threshold over several workloads (GPU box)
for thr in 0 0.5 1 1.5 3; do
  for w in "cube_s100k 65" "skillet_s100k 215" "ball_s10k 65" "cube_s30k 45" "bar_thin 30" "cube_s10k 65"; do
    echo -n "thr=$thr $w: "; MH_P_THRESHOLD=$thr python tools/skillet_probe.py $w 2>&1 | tail -1
  done
done
