import math, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cuddhelmholtz_amd as cd
cd.use_torch_stream()
dev = torch.device("cuda:0")
nx = 2048
t0=time.time()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0); fem = cd.H1Space(mesh, cd.Basis(4)); n = fem.size()
print("ndof", n, "setup", round(time.time()-t0,1), flush=True)
fs = cd.FaceSpace(fem, mesh.boundary_edges())
os.environ["CUDDH_PLAN_AFFINE"] = "0"
g = torch.Generator(device="cpu").manual_seed(1)
a2 = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
A = cd.HelmholtzOperator(64 * math.pi, a2, torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
x = torch.rand(2 * n, generator=g, dtype=torch.float64).to(dev); z = torch.rand(2 * n, generator=g, dtype=torch.float64).to(dev)
Ax, Az, Ax2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
A.action(x, Ax); A.action(z, Az); A.action(x, Ax2)
s1, s2 = float(torch.dot(z, Ax)), float(torch.dot(x, Az))
print("fused 2048^2: deterministic", bool(torch.equal(Ax, Ax2)), "symmetry rel", abs(s1 - s2) / abs(s1), "bytes", A.bytes_per_apply(), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): A.action(x, Ax)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) * 1e-3 / 10
print(f"fused 2048^2 general: {t*1e6:.0f} us, {A.bytes_per_apply()/t/1e9:.0f} GB/s", flush=True)
del A, x, z, Ax, Az, Ax2, a2
torch.cuda.empty_cache()
F = cd.DDH(64 * math.pi, np.ones(n), fem, nx, nx)
info = F.info(); print("DDH", info, "size", F.size(), flush=True)
lam = torch.rand(F.size(), dtype=torch.float32, device=dev); y1 = torch.zeros_like(lam); y2 = torch.zeros_like(lam)
t0 = time.time(); F.action(lam, y1); torch.cuda.synchronize(); t1 = time.time() - t0
t0 = time.time(); F.action(lam, y2); torch.cuda.synchronize(); t2 = time.time() - t0
print("DDH 2048^2 action", round(t1, 3), round(t2, 3), "deterministic", bool(torch.equal(y1, y2)), "finite", bool(torch.isfinite(y1).all()), "DoF*iter/s", 2 * n / t2)
