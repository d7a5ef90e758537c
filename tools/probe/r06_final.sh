# the round's evidence, in one go (GPU box, repo root):  bash tools/probe/r06_final.sh
bash tools/collect_profiles.sh r06 > /dev/null 2>&1
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
python tools/probe/quality_sphere_probe.py 96 48 --solve > gpurun_out/r06_sphere_96.txt 2>&1
python tools/probe/quality_sphere_probe.py 128 64 --solve > gpurun_out/r06_sphere_128.txt 2>&1
(FLAT_EPS=1e-6 python tools/probe/flat_sphere_probe.py 96 48 default MH_CLUSTERS=0; FLAT_EPS=1e-6 python tools/probe/flat_sphere_probe.py 128 64 default MH_CLUSTERS=0; FLAT_EPS=1e-7 python tools/probe/flat_sphere_probe.py 96 48 default) > gpurun_out/r06_flat_cells.txt 2>&1
bash tools/probe/r06_records.sh > /dev/null 2>&1
MH_VERBOSE=1 python tools/probe/r06_soak.py 40 60 > gpurun_out/r06_soak.txt 2> gpurun_out/r06_soak.err
(echo "# MH_VERBOSE=1 python tools/probe/r06_soak_large.py 1 12 (one MI355X): the soak's surface kinds at 3-10 times its sizes, default options, default config, 65 pairs"; python tools/probe/r06_soak_large.py 1 12 2>/dev/null | cut -c1-300) > gpurun_out/r06_soak_large.txt
grep -c "once more\|last resort\|dense eigensolve of order" gpurun_out/r06_soak.err
tail -1 gpurun_out/r06_soak.txt | cut -c1-700
