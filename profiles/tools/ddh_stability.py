"""Is the WaveHoltz time stepping of the reference stable at BASELINE's frequencies?  The explicit midpoint rule amplifies a
mode of the semi-discrete wave operator with frequency mu by |R|^2 = 1 + (mu dt)^4 / 4 per step, and a local solve takes
5 nt steps with dt = 0.1 h / n_basis^2 fixed by the mesh alone (source/DDH.cpp:363-368; the coefficient a(x), which scales the
local wave speed by 1/a, does not enter).  nt = T / dt grows as omega falls, so at low omega the growth factor of the highest
modes overflows.  This tool measures it: norm of rhs(f) relative to f, and a power iteration on T (update = T lambda).
usage: ddh_stability.py [nx=256] [coefficient: disk|one]"""
import math
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
coef = sys.argv[2] if len(sys.argv) > 2 else "disk"
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
print(f"{nx}x{nx} quads, n_basis {nb}, coefficient: {'two-valued disk of examples/DDH.cpp (a = 0.2 inside)' if coef == 'disk' else 'a = 1'}, fp64 DDH")
print(f"{'omega/pi':>9} {'elem/wavelength':>16} {'nt':>6} {'|rhs(f)| / |f|':>15} {'|T^k v| / |T^(k-1) v|, k = 1..8':>40}")
for epw in (64.0, 32.0, 16.0, 10.0, 5.0):
    omega = 2 * math.pi / (epw * 2.0 / nx)
    f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)
    F = cd.DDH(omega, h_a, fem, nx, nx, precision="f64")
    n = F.size()
    b = torch.zeros(n, dtype=torch.float64, device=dev)
    F.rhs(f, b)
    g = torch.Generator(device="cpu").manual_seed(1)
    v = (2 * torch.rand(n, generator=g, dtype=torch.float64) - 1).to(dev)
    v /= v.norm()
    ratios = []
    for _ in range(8):
        w = torch.zeros_like(v)
        F.local_traces(0, F.info()["n_domains"], None, v, w)
        r = float(w.norm())
        ratios.append(r)
        v = w / r if math.isfinite(r) and r > 0 else w
    print(f"{omega / math.pi:9.2f} {epw:16.0f} {F.info()['nt']:6d} {float(b.norm() / f.norm()):15.3e}   " + " ".join(f"{r:9.3e}" for r in ratios))
