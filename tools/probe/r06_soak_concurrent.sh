# Round 6 (the round-5 soaks again, on the final code), with overlap allowed at every block width (no exclusive phase any more).
out=gpurun_out/r06_soak_concurrent.txt
: > $out
echo "== tools/concurrent_solves.py 3 45 12 120 (rounds 2-4: five of eight runs lost a solve), three runs" >> $out
for i in 1 2 3; do MH_CONCURRENT_SOLVES=1 timeout 600 python tools/concurrent_solves.py 3 45 12 120 2>&1 | tail -3 >> $out; done
echo "== tools/concurrent_solves.py 3 18 14 215, two runs" >> $out
for i in 1 2; do MH_CONCURRENT_SOLVES=1 timeout 600 python tools/concurrent_solves.py 3 18 14 215 2>&1 | tail -3 >> $out; done
echo "== tools/probe/one_wide_soak.py (one thread of 120 pairs beside two of 65), three runs, and all three wide" >> $out
for i in 1 2 3; do timeout 600 python tools/probe/one_wide_soak.py 1 2>&1 | tail -1 >> $out; done
timeout 600 python tools/probe/one_wide_soak.py 3 2>&1 | tail -1 >> $out
echo "== tools/probe/mixed_soak.py" >> $out
timeout 900 python tools/probe/mixed_soak.py 2>&1 | tail -3 >> $out
echo "== the product's exchange kernel, 60 000 launches beside 215-pair solves in this process, 30 000 beside another process's" >> $out
timeout 600 python tools/probe/sytrd_soak.py 0x400 60000 solve215 2>&1 | head -2 >> $out
timeout 600 python tools/probe/sytrd_soak.py 0x400 30000 proc215 2>&1 | head -2 >> $out
cat $out
