for w in "ball_s10k 65" "cube_s10k 65" "cube_s30k 45"; do
for cfg in "A=0" "MH_DEG2=3" "MH_DEG2=4" "MH_DEG2=3 MH_DEG1=6" "MH_DEG2=3 MH_DEG1=6 MH_GAMMA=4" "MH_DEG2=4 MH_DEG1=6 MH_GAMMA=4" "MH_DEG2=3 MH_AGG=16" "MH_DEG2=3 MH_DEG1=6 MH_AGG=8" "MH_PRECOND_FP64=1"; do
echo -n "$w | $cfg : "; env $cfg python tools/skillet_probe.py $w 2>&1 | tail -1
done; done
