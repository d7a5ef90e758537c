import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_gltf_goldens as T
from mesheditor_amd import api
full=np.load('/root/repo/tests/golden/gltf_modal_models_full.npz')
import os
os.environ['MH_VERBOSE']='1'
for name in ("Marble","Solved sphere"):
    g,material,max_freq=T.golden_of(full,name)
    pts,tets,_=T.front_end_mesh(g)
    ctx=api.Context(0)
    got=api.mesh2modes(ctx,pts,tets,api.material(*material),g["positions"],config=api.default_config(num_modes=30,num_fem_modes=45,max_mode_freq=max_freq))
    print(name, len(got.eigenvalues), got.profile)
    ctx.close()
