"""Why DDH-GMRES stagnates at BASELINE's configurations: the same mesh, the same solver, different omega.
BASELINE configs 3/4 put omega = pi nx / 32 (64 elements = 16 subdomains per wavelength); the reference's own example puts
omega = 2 pi nx / 10 (5 elements per wavelength, a subdomain is 0.8 wavelengths wide; examples/DDH.cpp:89).  Non-overlapping
Robin (Despres) transmission leaves modes that are evanescent on the scale of a subdomain with a convergence factor of
modulus 1, and with 64 elements per wavelength nearly every interface mode is of that kind.
usage: ddh_frequency_sweep.py [nx=256] [matvec budget=600] [coefficient: disk|one]"""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 600
coef = sys.argv[3] if len(sys.argv) > 3 else "disk"
nb = 4
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
ndof = fem.size()
a = torch.ones(ndof, dtype=torch.float64, device=dev)
if coef == "disk":
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
h_a = a.cpu().numpy()
print(f"coefficient: {coef}")
print(f"{nx}x{nx} quads, n_basis {nb} ({(nx // 4) ** 2} subdomains of 4x4 elements), fp32 DDH, GMRES(20), tol 1e-4, budget {budget} matvecs")
print(f"{'omega/pi':>9} {'elements per wavelength':>24} {'subdomains per wavelength':>26} {'matvecs':>8} {'rel. residual':>14} {'converged':>10} {'nt':>6}")
for elems_per_wavelength in (64.0, 32.0, 16.0, 10.0, 5.0):
    omega = 2 * math.pi / (elems_per_wavelength * 2.0 / nx)
    f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)
    F = cd.DDH(omega, h_a, fem, nx, nx)
    n = F.size()
    b = torch.zeros(n, dtype=torch.float32, device=dev)
    lam = torch.zeros_like(b)
    F.rhs(f, b)
    out = cd.gmres(n, lam, F, b, 20, budget // 21 + 1, 1e-4)
    print(f"{omega / math.pi:9.2f} {elems_per_wavelength:24.0f} {elems_per_wavelength / 4:26.2f} {out.num_matvec:8d} "
          f"{out.res_norm[-1] / out.res_norm[0]:14.3e} {str(out.success):>10} {F.info()['nt']:6d}")
