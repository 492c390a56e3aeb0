"""BASELINE config 2: unpreconditioned GMRES(20) on the fused complex Helmholtz operator, omega = 8 pi, 256^2, n_basis 4.
Prints DoF.iter/s (N = 2 ndof complex dofs per matvec)."""
import math
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
cd.use_torch_stream()
omega = math.pi * nx / 32.0
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(4))
n = fem.size()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
a2 = torch.zeros(n, dtype=torch.float64, device=dev)
cd.nodal_values(fem, cd.ALPHA_DISK_SQ, a2)
A = cd.HelmholtzOperator(omega, a2, torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
b = torch.zeros(2 * n, dtype=torch.float64, device=dev)
cd.linear_functional(fem, cd.GAUSSIANS, b[:n], param=omega)
x = torch.zeros_like(b)
solvers = [("gmres(), reference ordering", lambda: cd.gmres(2 * n, x, A, b, 20, cycles + 1, 1e-30))]
if A.has_native():
    solvers.append(("HelmholtzOperator::gmres, plan-native vectors", lambda: A.gmres(x, b, 20, cycles + 1, 1e-30)))
for name, solve in solvers:
    for rep in range(2):  # the first pass warms up
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = solve()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print(f"nx={nx} N={2 * n} {name}: matvecs={out.num_matvec} seconds={t:.4f} DoF*iter/s={2 * n * out.num_matvec / t:.4g} "
          f"us_per_matvec={1e6 * t / out.num_matvec:.1f} rel_res={out.res_norm[-1] / out.res_norm[0]:.3e} kernel {A.kernel()}")
