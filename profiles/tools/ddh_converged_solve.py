"""A DDH solve carried to the example's tolerance at BASELINE config 3's discretisation (512 x 512 quads, n_basis 4, omega = 16 pi,
16,384 subdomains, fp32 local solves on kernel 5): rhs -> gmres(I - T) -> postprocess as in examples/DDH.cpp:141-144, with the
coefficient a = 1 (with the example's disk the reference's own time step is unstable where a = 0.2: DESIGN 5.2) and a long
restart.  Prints the residual after every cycle (verbose = 2), then the true relative residual of the trace system and the
norm of the postprocessed solution.
usage: ddh_converged_solve.py [nx=512] [m=400] [tol=1e-4] [max_seconds=950] [rule=baseline|example] [coef=one|disk] [also_f64=0|1]
also_f64 = 1: the same solve with fp64 local solves (DDH64, tolerance tol / 100) and the distance between the two solutions.
rule "example": omega = 2 pi nx / 10 (examples/DDH.cpp:109: five elements per wavelength, where the reference's time step is
stable also with its disk coefficient) instead of BASELINE's omega = pi nx / 32."""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
max_seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 950.0
rule = sys.argv[5] if len(sys.argv) > 5 else "baseline"
coef = sys.argv[6] if len(sys.argv) > 6 else "one"
omega = math.pi * nx / 32.0 if rule == "baseline" else 2.0 * math.pi * nx / 10.0
dev = torch.device("cuda:0")
cd.use_torch_stream()
fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(4))
ndof = fem.size()
f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)
a = torch.ones(ndof, dtype=torch.float64, device=dev)
if coef == "disk":  # examples/DDH.cpp:122-125
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
F = cd.DDH(omega, a.cpu().numpy(), fem, nx, nx, precision="f32")
n = F.size()
b = torch.zeros(n, dtype=torch.float32, device=dev)
lam = torch.zeros_like(b)
F.rhs(f, b)
print(f"DDH solve {nx}x{nx}, omega = {omega / math.pi:g} pi ({rule} rule), coefficient {coef}, kernel {F.info()['kernel']}, {n} traces, GMRES({m}), tol {tol:g}", flush=True)
t0 = time.perf_counter()
out = cd.gmres(n, lam, F, b, m, 1000, tol, verbose=2, max_seconds=max_seconds)
torch.cuda.synchronize()
t = time.perf_counter() - t0
r = torch.zeros_like(b)
F.action(lam, r)
true_res = float(torch.linalg.norm((b - r).double()) / torch.linalg.norm(b.double()))
u = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
F.postprocess(lam, f, u)
print(f"success = {out.success}, {out.num_iter} cycles, {out.num_matvec} matvecs, {t:.1f} s ({2.0 * ndof * out.num_matvec / t / 1e6:.1f} M DoF*iter/s)")
print(f"true relative residual ||b - (I - T) lambda|| / ||b|| = {true_res:.3e}; ||u||_2 = {float(torch.linalg.norm(u)):.6e}, finite = {bool(torch.isfinite(u).all())}")
print("relative residual after cycle k: " + " ".join(f"{v / out.res_norm[0]:.2e}" for v in out.res_norm))

if len(sys.argv) > 7 and sys.argv[7] == "1":
    F64 = cd.DDH(omega, a.cpu().numpy(), fem, nx, nx, precision="f64")
    b64 = torch.zeros(n, dtype=torch.float64, device=dev)
    lam64 = torch.zeros_like(b64)
    F64.rhs(f, b64)
    t0 = time.perf_counter()
    out64 = cd.gmres(n, lam64, F64, b64, m, 1000, tol / 100.0, verbose=0, max_seconds=max_seconds)
    torch.cuda.synchronize()
    t64 = time.perf_counter() - t0
    u64 = torch.zeros_like(u)
    F64.postprocess(lam64, f, u64)
    print(f"fp64 local solves (kernel {F64.info()['kernel']}): success = {out64.success}, {out64.num_matvec} matvecs, {t64:.1f} s, "
          f"relative residual {out64.res_norm[-1] / out64.res_norm[0]:.3e}")
    print(f"fp32 solve (tol {tol:g}) vs fp64 solve (tol {tol / 100:g}): ||u32 - u64|| / ||u64|| = {float(torch.linalg.norm(u - u64) / torch.linalg.norm(u64)):.3e}; "
          f"traces: {float(torch.linalg.norm(lam.double() - lam64) / torch.linalg.norm(lam64)):.3e}")
