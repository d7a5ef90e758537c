export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/soak6 gpurun_out/soak_first_deviations.json
  rocprofv3 --kernel-trace --output-format csv -d /tmp/soak6 -- python3 tools/probe/sytrd_soak.py 0 8000 solve215 > gpurun_out/soak6_run$i.txt 2>&1
  head -1 gpurun_out/soak6_run$i.txt | cut -c1-200
  if [ -f gpurun_out/soak_first_deviations.json ]; then python3 tools/probe/soak_correlate.py /tmp/soak6 > gpurun_out/soak6_correlate$i.txt 2>&1; fi
done
cat gpurun_out/soak6_correlate*.txt | grep -v "^kernels near\|^ *[0-9]* *void\|dispatches" | cut -c1-230
