"""BASELINE config 3 as a SOLVE, not as steps: DDH at omega = 16 pi on 512 x 512 quads, n_basis 4 (16,384 subdomains),
the flow of examples/DDH.cpp:141-144 (rhs -> gmres on I - T -> postprocess), in fp32 (reference precision, kernel 5) and in
fp64 (DDH64), with GMRES(20) (the example's restart) and a long restart.  Prints the relative residual after every cycle and
the distance between the fp32 and fp64 solutions after the same number of cycles.
usage: ddh_convergence.py [nx=512] [cycles20=60] [cycles_long=6] [m_long=200] [cycles64=15] [coefficient: disk|one]"""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cycles20 = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cycles_long = int(sys.argv[3]) if len(sys.argv) > 3 else 6
m_long = int(sys.argv[4]) if len(sys.argv) > 4 else 200
cycles64 = int(sys.argv[5]) if len(sys.argv) > 5 else 15
coef = sys.argv[6] if len(sys.argv) > 6 else "disk"
nb = 4
omega = math.pi * nx / 32.0
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
ndof = fem.size()
f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
a = torch.ones(ndof, dtype=torch.float64, device=dev)
cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)
if coef == "disk":
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
h_a = a.cpu().numpy()
print(f"DDH solve, {nx}x{nx} quads, n_basis {nb}, omega = {omega / math.pi:g} pi, {2 * ndof} unknowns, forcing of examples/DDH.cpp, coefficient: {coef}")


def solve(precision, m, cycles):
    F = cd.DDH(omega, h_a, fem, nx, nx, precision=precision)
    n = F.size()
    b = torch.zeros(n, dtype=F.trace_dtype, device=dev)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
    F.rhs(f, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = cd.gmres(n, lam, F, b, m, cycles + 1, 1e-4 if precision == "f32" else 1e-10)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    F.postprocess(lam, f, u)
    res = np.asarray(out.res_norm) / out.res_norm[0]
    print(f"\n{precision} DDH (kernel {F.info()['kernel']}), GMRES({m}), {out.num_iter} cycle(s), {out.num_matvec} matvecs, {t:.1f} s "
          f"({2.0 * ndof * out.num_matvec / t / 1e6:.1f} M DoF*iter/s), success = {out.success}, n_traces = {n}")
    print("  relative residual after cycle k: " + " ".join(f"{r:.4f}" for r in res))
    return u, res, out


u32, r32, o32 = solve("f32", 20, cycles20)
u32l, r32l, o32l = solve("f32", m_long, cycles_long)
u64, r64, o64 = solve("f64", 20, cycles64)
if cycles64 < cycles20:
    u32s, r32s, _ = solve("f32", 20, cycles64)
else:
    u32s, r32s = u32, r32
k = min(len(r32s), len(r64))
print(f"\nfp32 vs fp64 after the same {k - 1} GMRES(20) cycles: residual histories differ by at most "
      f"{np.abs(r32s[:k] - r64[:k]).max():.2e}; solutions differ by {float(torch.linalg.norm(u32s - u64) / torch.linalg.norm(u64)):.2e} (relative l2)")
print(f"GMRES(20) after {o32.num_matvec} matvecs: {r32[-1]:.4f};  GMRES({m_long}) after {o32l.num_matvec} matvecs: {r32l[-1]:.4f}")
