ROOT=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_it
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_it -- python3 $ROOT/tools/scan_probe.py cube_s100k --reps 2 > /tmp/prof_it.log 2>&1
cd $ROOT
python3 tools/iteration_timeline.py /tmp/prof_it 8 > gpurun_out/iteration_timeline.txt 2>&1
