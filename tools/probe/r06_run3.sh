python tools/probe/flat_sphere_probe.py 96 48 default MH_CLUSTERS=0 > gpurun_out/r06_flat_sphere.txt 2>&1
python tools/probe/flat_sphere_probe.py 128 64 default MH_CLUSTERS=0 >> gpurun_out/r06_flat_sphere.txt 2>&1
cat gpurun_out/r06_flat_sphere.txt
python tools/scan_probe.py scan_s30k scan_s100k scan_s100k_interior scan_s100k_repaired ball_s10k uvsphere_s10k --reps 2 --json gpurun_out/r06_scan_clusters.json 2>&1 | tail -12
MH_CLUSTERS=0 python tools/scan_probe.py scan_s30k scan_s100k scan_s100k_interior scan_s100k_repaired --reps 2 --json gpurun_out/r06_scan_noclusters.json 2>&1 | tail -8
