"""Forty launches of the register-resident tridiagonalisation at order 222 (for the counter passes of tools/probe/sytrd_regs_pmc.sh)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from mesheditor_amd import api
import lab
ctx = api.Context(0)
rng = np.random.default_rng(5)
b = rng.standard_normal((222, 222))
print(lab.tridiagonalize(ctx, b + b.T, variant=3, reps=40)[2])
