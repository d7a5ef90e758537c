# every odd-input probe of round 6, one after the other (GPU box, repo root):  bash tools/probe/r06_odd_inputs.sh > gpurun_out/r06_odd_inputs.txt
for p in r06_odd_meshes_probe r06_two_bodies_probe r06_hinge_probe r06_far_and_small_probe r06_odd_arguments_probe r06_odd_seeds_probe r06_floor_parity_probe r06_odd_bank_probe; do
    echo "# tools/probe/$p.py"
    timeout 400 python tools/probe/$p.py 2>/dev/null | cut -c1-400
    echo
done
