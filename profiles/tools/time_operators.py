"""Times the per-class operator applies (reference-layout kernels) and the fused apply; prints algorithmic GB/s."""
import math
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
n, ne = fem.size(), mesh.n_elem()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.zeros_like(x)
coef = torch.ones(n, dtype=torch.float64, device=dev)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


fs = cd.FaceSpace(fem, mesh.boundary_edges())
A = cd.HelmholtzOperator(math.pi * nx / 32, coef, torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
xx = torch.rand(2 * n, dtype=torch.float64, device=dev)
yy = torch.empty_like(xx)
if A.fused():
    t = timeit(lambda: A.action(xx, yy))
    b = A.bytes_per_apply()
    print(f"nx={nx} nb={nb} fused complex apply:          {t * 1e6:9.1f} us  {b / t / 1e9:8.1f} GB/s algorithmic ({b / 1e6:.1f} MB)")
t = timeit(lambda: A.action_unfused(xx, yy))
print(f"nx={nx} nb={nb} unfused composite:            {t * 1e6:9.1f} us")
S = cd.StiffnessMatrix(fem)
M = cd.MassMatrix(fem)
Mw = cd.MassMatrix(fem, coef)
nqS, nqM, nqW = nb + 1, nb + 1, 1 + 3 * nb // 2 + 1
for name, op, nq, comps in (("stiffness", S, nqS, 3), ("mass", M, nqM, 1), ("mass(weighted)", Mw, nqW, 1)):
    t = timeit(lambda: op.action(x, y))
    b = ne * (comps * nq * nq * 8 + nb * nb * 4) + n * 16
    print(f"nx={nx} nb={nb} {name:15s} action(x,y): {t * 1e6:9.1f} us  {b / t / 1e9:8.1f} GB/s algorithmic ({b / 1e6:.1f} MB)")
    t = timeit(lambda: op.action(0.5, x, y))
    print(f"nx={nx} nb={nb} {name:15s} action(c,x,y): {t * 1e6:7.1f} us  {b / t / 1e9:8.1f} GB/s")
