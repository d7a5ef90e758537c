for tr in 0x00c 0x000 0x00c 0x00c 0x00c; do
  timeout 600 python tools/probe/sytrd_soak.py $tr 30000 solve215 > gpurun_out/soak10_tr$tr.txt 2>&1
  head -2 gpurun_out/soak10_tr$tr.txt | cut -c1-300
done
