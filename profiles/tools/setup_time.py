import math, os, sys, time
os.environ.setdefault('CUDDH_SETUP_TIMING', '1')
import numpy as np, torch
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd
cd.use_torch_stream()
nx=1024
torch.zeros(1, device='cuda'); torch.cuda.synchronize()  # the process's one-off HIP initialisation is not set-up work
t0=time.time()
t=time.time(); mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0); print('mesh', time.time()-t)
t=time.time(); fem = cd.H1Space(mesh, cd.Basis(4)); print('h1space', time.time()-t)
n=fem.size()
t=time.time(); F = cd.DDH(math.pi*nx/32, np.ones(n), fem, nx, nx); print('ddh ctor', time.time()-t)
t=time.time(); print('kernel', F.info()['kernel'], 'plan (device tables + structure check)', time.time()-t)
b = torch.zeros(F.size(), dtype=torch.float32, device='cuda'); f=torch.zeros(2*n, dtype=torch.float64, device='cuda')
t=time.time(); F.rhs(f,b); torch.cuda.synchronize(); print('first rhs (plan + kernel)', time.time()-t)
t=time.time(); F.rhs(f,b); torch.cuda.synchronize(); print('second rhs', time.time()-t)
fs = cd.FaceSpace(fem, mesh.boundary_edges())
t=time.time(); A = cd.HelmholtzOperator(1.0, torch.ones(n,dtype=torch.float64,device='cuda'), torch.ones(fs.size(),dtype=torch.float64,device='cuda'), fem, fs); torch.cuda.synchronize(); print('helmholtz operator (3 operators + plan)', time.time()-t)
print('total: mesh + H1Space + DDH constructor + plan', 'see lines above; wall clock since start', time.time()-t0)
