"""The whole mesh2modes path at 280 eigenpairs / 260 kept modes (shape gathers, post-processing, basis export, warm restart) against the oracle
on a plate it can follow.   python tools/probe/wide_path_probe.py"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
from oracle import pyoracle as oracle
ctx = api.Context(0)
m = meshes.MATERIALS["Iron"]
pts, tets = meshes.kuhn_box(24, 24, 2, 0.26, 0.26, 0.012)
ex = pts[::37].astype(np.float32)
kw = dict(num_modes=260, num_fem_modes=280, max_mode_freq=1e7)
ro = oracle.mesh2modes(pts, tets, oracle.material(*m), ex, config=oracle.default_config(**kw))
rg = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(**kw), keep_basis=True)
print("modes", len(ro.freqs), len(rg.freqs), "positions", ro.shapes.shape, rg.shapes.shape)
print("freq rel", np.abs(rg.freqs - ro.freqs).max() / ro.freqs.max(), "t60 rel", np.abs(rg.t60s - ro.t60s).max() / ro.t60s.max())
# shapes up to sign / rotation inside clusters: compare per-mode energy over the sample points
eo, eg = (ro.shapes.astype(float) ** 2).sum(axis=(0, 2)), (rg.shapes.astype(float) ** 2).sum(axis=(0, 2))
print("per-mode shape energy rel (median, max)", np.median(np.abs(eg - eo) / eo), (np.abs(eg - eo) / eo).max())
print("basis", None if rg.basis is None else rg.basis.shape)
# warm restart from the exported basis
rw = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, config=api.default_config(**kw), seed_basis=rg.basis)
print("warm: modes", len(rw.freqs), "freq rel vs cold", np.abs(rw.freqs - rg.freqs).max() / rg.freqs.max(), "iterations cold/warm", rg.profile["restarts"], rw.profile["restarts"])
