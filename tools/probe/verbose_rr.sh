MH_VERBOSE=1 python tools/scan_probe.py cube_s100k --reps 0 2>&1 | grep "\[rr\]" | cut -c1-200 > gpurun_out/verbose_rr.txt
head -60 gpurun_out/verbose_rr.txt
