python tools/probe/quality_sphere_probe.py 96 48 --solve > gpurun_out/r06_sphere_96.txt 2>&1
python tools/probe/quality_sphere_probe.py 128 64 --solve > gpurun_out/r06_sphere_128.txt 2>&1
cat gpurun_out/r06_sphere_96.txt gpurun_out/r06_sphere_128.txt
bash tools/probe/run_gpu_suite.sh
