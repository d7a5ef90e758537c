# round-6 evidence files: the flat-cell meshes through the solver, the cluster A/B on the scan fills, the iteration timeline
true
true
(echo "# default (element patches on these meshes)"; python tools/scan_probe.py scan_s30k scan_s100k scan_s100k_interior scan_s100k_repaired --reps 2 2>&1 | tail -4 | cut -c1-260
 echo "# MH_CLUSTERS=1 (cluster patches forced, both levels)"; MH_CLUSTERS=1 python tools/scan_probe.py scan_s30k scan_s100k scan_s100k_interior scan_s100k_repaired --reps 2 2>&1 | tail -4 | cut -c1-260
 echo "# MH_CLUSTERS=2 (clusters on the P1 level only)"; MH_CLUSTERS=2 python tools/scan_probe.py scan_s30k scan_s100k scan_s100k_interior scan_s100k_repaired --reps 2 2>&1 | tail -4 | cut -c1-260
 echo "# MH_TEST=last_resort (every solve through the conjugate-gradient search directions)"; MH_TEST=last_resort python tools/scan_probe.py ball_s10k cube_s30k scan_s30k cube_s100k --reps 1 2>&1 | tail -4 | cut -c1-330) > gpurun_out/r06_scan_clusters.txt 2>&1
bash tools/probe/prof_iteration.sh
cp gpurun_out/iteration_timeline.txt gpurun_out/r06_iteration_timeline.txt
tail -3 gpurun_out/r06_iteration_timeline.txt
