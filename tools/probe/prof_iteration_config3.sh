ROOT=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python3 $ROOT/tools/scan_probe.py config3_s100k_repaired --reps 1 > /tmp/prof_c3.log 2>&1
cd $ROOT
python3 tools/iteration_timeline.py /tmp/prof_c3 12 k_sytrd_wide > gpurun_out/iteration_timeline_config3.txt 2>&1
