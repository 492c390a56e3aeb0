import math, sys
import numpy as np, torch
sys.path.insert(0, '.')
import cuddhelmholtz_amd as cd
nx, nb = 256, 4
dev = torch.device("cuda:0"); cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0); fem = cd.H1Space(mesh, cd.Basis(nb)); ndof = fem.size()
a = torch.zeros(ndof, dtype=torch.float64, device=dev); cd.linear_functional(fem, cd.ALPHA_DISK, a); cd.DiagInvMassMatrix(fem).action(a, a); h_a = a.cpu().numpy()
omega = 4 * math.pi
f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev); cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)
for prec, kern in (("f32", 5), ("f32", 3), ("f32", 1), ("f64", 0)):
    F = cd.DDH(omega, h_a, fem, nx, nx, precision=prec, kernel=kern)
    n = F.size(); b = torch.zeros(n, dtype=F.trace_dtype, device=dev); F.rhs(f, b)
    y = torch.zeros_like(b); F.action(b, y)
    print(prec, "kernel", F.info()["kernel"], "nt", F.info()["nt"], "|b|", float(b.double().norm()), "finite b", bool(torch.isfinite(b).all()), "|(I-T)b|", float(y.double().norm()), "max|b|", float(b.abs().max()))
    lam = torch.zeros_like(b)
    out = cd.gmres(n, lam, F, b, 20, 4, 1e-4)
    print("   gmres res", [r / out.res_norm[0] for r in out.res_norm])
