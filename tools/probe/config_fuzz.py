"""mesh2modes configurations off the beaten path, device against oracle: kept frequencies, decay times, excitation map and mass properties.
    python tools/probe/config_fuzz.py"""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from mesheditor_amd import api, meshes
from oracle import pyoracle as oracle
ctx = api.Context(0)
pts, tets = meshes.jittered_box(6, 4242)
m = meshes.MATERIALS["Glass"]
rng = np.random.default_rng(3)
ex_full = pts[rng.choice(len(pts), 12, replace=False)].astype(np.float32) + 1e-3

def case(name, ex=ex_full, scale=(1.0, 1.0, 1.0), **kw):
    try:
        ro = oracle.mesh2modes(pts, tets, oracle.material(*m), ex, baked_scale=scale, config=oracle.default_config(**kw))
    except Exception as e:
        ro = e
    try:
        rg = api.mesh2modes(ctx, pts, tets, api.material(*m), ex, baked_scale=scale, config=api.default_config(**kw))
    except Exception as e:
        rg = e
    if isinstance(ro, Exception) or isinstance(rg, Exception):
        print(f"{name:44s} oracle: {str(ro)[:60] if isinstance(ro, Exception) else 'ok'} | device: {str(rg)[:60] if isinstance(rg, Exception) else 'ok'}", flush=True)
        return
    n = (len(ro.freqs), len(rg.freqs))
    if n[0] != n[1]:
        print(f"{name:44s} MODE COUNT DIFFERS oracle {n[0]} device {n[1]}", flush=True)
        return
    df = np.abs(np.asarray(rg.freqs) - np.asarray(ro.freqs)).max() / max(np.abs(ro.freqs).max(), 1e-30) if n[0] else 0.0
    dt = np.abs(np.asarray(rg.t60s) - np.asarray(ro.t60s)).max() / max(np.abs(ro.t60s).max(), 1e-30) if n[0] else 0.0
    same_ex = np.array_equal(np.asarray(rg.sample_point_of_excitation), np.asarray(ro.sample_point_of_excitation))
    dm = abs(rg.mass - ro.mass) / max(abs(ro.mass), 1e-30)
    print(f"{name:44s} modes {n[0]:3d}  freq rel {df:.1e}  t60 rel {dt:.1e}  excitation map {'same' if same_ex else 'DIFFERS'}  mass rel {dm:.1e}", flush=True)

case("defaults")
case("one mode", num_modes=1, num_fem_modes=20)
case("num_modes > num_fem_modes", num_modes=40, num_fem_modes=20)
case("band keeps nothing (max 10 Hz)", num_modes=10, num_fem_modes=25, max_mode_freq=10.0)
case("band starts high (min 20 kHz)", num_modes=10, num_fem_modes=40, min_mode_freq=20000.0, max_mode_freq=1e6)
case("fundamental 440 Hz", num_modes=10, num_fem_modes=25, max_mode_freq=1e6, fundamental_freq=440.0)
case("no excitation points", ex=np.zeros((0, 3), np.float32), num_modes=10, num_fem_modes=25, max_mode_freq=1e6)
case("one excitation point", ex=ex_full[:1], num_modes=10, num_fem_modes=25, max_mode_freq=1e6)
case("repeated excitation points", ex=np.repeat(ex_full[:3], 4, axis=0), num_modes=10, num_fem_modes=25, max_mode_freq=1e6)
case("baked scale (2, 1, 0.5)", scale=(2.0, 1.0, 0.5), num_modes=10, num_fem_modes=25, max_mode_freq=1e6)
case("baked scale (0.01, 0.01, 0.01)", scale=(0.01, 0.01, 0.01), num_modes=10, num_fem_modes=25, max_mode_freq=1e9)
case("max_restarts 1", num_modes=10, num_fem_modes=25, max_mode_freq=1e6, max_restarts=1)
case("num_fem_modes 7 (rigid + 1)", num_modes=10, num_fem_modes=7, max_mode_freq=1e6)
case("num_fem_modes 6 (rigid only)", num_modes=10, num_fem_modes=6, max_mode_freq=1e6)
