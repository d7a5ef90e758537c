import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from mesheditor_amd import api, meshes, tets as T
ctx=api.Context(0)
m=meshes.MATERIALS["Ceramic"]
def run(name,v,f,**kw):
    for shell in ("never","when_flat","always"):
        pts,tets,_=T.tetrahedralize(v,f,interior_shell=shell,**kw)
        ex=pts[(np.arange(10)*len(v))//10].astype(np.float32)
        ts=[]
        for _ in range(2):
            t0=time.time(); r=api.mesh2modes(ctx,pts,tets,api.material(*m),ex,config=api.default_config(num_modes=50,num_fem_modes=65)); ctx.synchronize(); ts.append(time.time()-t0)
        print(name,shell,len(pts),len(tets),'dof',r.profile.get('dofs'),'pairs',len(r.eigenvalues),'its',r.profile.get('restarts'),'%.0f ms'%(1e3*ts[-1]),'f1 %.1f'%(np.sqrt(max(r.eigenvalues[6],0))/(2*np.pi) if len(r.eigenvalues)>6 else 0),flush=True)
v,f=meshes.uv_sphere_surface(0.15,80,40); run('uv80x40',v,f)
sys.path.insert(0,'/root/repo/tools/probe')
import solid_scan_probe as S
v,f=meshes.marching_tets_surface(S.bumpy_ball,(-0.14,-0.16,-0.19),(0.14,0.16,0.19),0.012); v=meshes.taubin_smooth(v,f,8); v,f=meshes.largest_component(v,f); run('solid 0.012',v,f)
v,f=meshes.skillet_scan_surface(0.011,0.015); run('skillet 0.011',v,f)
