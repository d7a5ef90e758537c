for w in cube_s100k scan_s100k_repaired; do
MH_VERBOSE=1 python tools/scan_probe.py $w --reps 0 2>&1 | grep "\[lobpcg\] it" | cut -c1-60 > gpurun_out/verbose_$w.txt
cat gpurun_out/verbose_$w.txt
done
