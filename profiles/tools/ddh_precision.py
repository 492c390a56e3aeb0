"""DDH solve in the reference's fp32 and in the fp64 parity mode on the same problem: iteration counts, times and the
relative l2 difference of the post-processed solutions.  usage: ddh_precision.py [nx] [maxit] [tol] [omega_over_pi]"""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
dev = torch.device("cuda:0")
cd.use_torch_stream()
omega = math.pi * (float(sys.argv[4]) if len(sys.argv) > 4 else nx / 32.0)
fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(4))
n = fem.size()
f = torch.zeros(2 * n, dtype=torch.float64, device=dev)
a = torch.zeros(n, dtype=torch.float64, device=dev)
cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
cd.linear_functional(fem, cd.ALPHA_DISK, a)
cd.DiagInvMassMatrix(fem).action(a, a)
h_a = a.cpu().numpy()
sol = {}
for prec in ("f32", "f64"):
    F = cd.DDH(omega, h_a, fem, nx, nx, precision=prec)
    b = torch.zeros(F.size(), dtype=F.trace_dtype, device=dev)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * n, dtype=torch.float64, device=dev)
    F.rhs(f, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = cd.gmres(F.size(), lam, F, b, 20, maxit, tol)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    F.postprocess(lam, f, u)
    sol[prec] = u.cpu().numpy()
    print(f"nx={nx} {prec}: kernel {F.info()['kernel']}, success={out.success} cycles={out.num_iter} matvecs={out.num_matvec} "
          f"rel_res={out.res_norm[-1] / out.res_norm[0]:.3e} t_gmres={t:.2f}s DoF*iter/s={2 * n * out.num_matvec / t:.4g}")
    del F
d = np.linalg.norm(sol["f32"] - sol["f64"]) / np.linalg.norm(sol["f64"])
print(f"nx={nx}: relative l2 difference of the fp32 and fp64 solutions: {d:.3e}")
