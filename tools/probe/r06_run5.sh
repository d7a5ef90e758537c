python tools/probe/flat_sphere_probe.py 128 64 default > gpurun_out/r06_flat_sphere_d.txt 2>&1
grep -v " conv " gpurun_out/r06_flat_sphere_d.txt | tail -30
grep " conv " gpurun_out/r06_flat_sphere_d.txt | tail -12
MH_TEST=last_resort python tools/scan_probe.py ball_s10k cube_s30k scan_s30k --reps 1 2>&1 | tail -4
