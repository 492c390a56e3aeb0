"""Runs one hot kernel a few times (for rocprofv3 counter passes).
usage: run_kernel.py helm|helmn|stiff|mass|ddh NX [REPS] [KERNEL] [NB=4] [REFINE]      (helmn: the fused apply on plan-native vectors)
REFINE >= 0: the reference's unstructured fixture refined REFINE times instead of uniform_rect(NX) (NX is then ignored).
Prints the kernel instantiation the operator launches (cuddh_hip_helmholtz_plan_describe)."""
import math
import os
import sys
from pathlib import Path

import torch

os.environ.setdefault("CUDDH_PLAN_AFFINE", "0")  # counters are collected on the general-geometry layout (the roofline figure)

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

which, nx = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kernel = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 4
refine = int(sys.argv[6]) if len(sys.argv) > 6 else -1
dev = torch.device("cuda:0")
cd.use_torch_stream()
omega = math.pi * nx / 32.0
if refine >= 0:
    mesh = cd.Mesh2D.load(Path(__file__).resolve().parents[2] / "tests" / "golden" / "unstructured_square").refined(refine)
else:
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
ndof = fem.size()
print("elements", mesh.n_elem(), "n_basis", nb, "ndof", ndof)
if which in ("helm", "helmn"):
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    A = cd.HelmholtzOperator(omega, torch.ones(ndof, dtype=torch.float64, device=dev), torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
    x = torch.rand(2 * ndof, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for _ in range(reps):
        (A.action_native if which == "helmn" else A.action)(x, y)
    print("kernel:", A.kernel(), "(plan-native vectors)" if which == "helmn" else "", "| algorithmic bytes", A.bytes_per_apply(), "| layout bytes",
          A.bytes_native() if which == "helmn" else A.bytes_per_apply(True))
elif which in ("stiff", "mass"):
    op = cd.StiffnessMatrix(fem) if which == "stiff" else cd.MassMatrix(fem, 0.5 + torch.rand(ndof, dtype=torch.float64, device=dev))
    x = torch.rand(ndof, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for _ in range(reps):
        op.action(x, y)
    print("kernel:", op.kernel())
else:
    import numpy as np

    F = cd.DDH(omega, np.ones(ndof), fem, nx, nx, kernel=kernel)
    lam = torch.rand(F.size(), dtype=torch.float32, device=dev)
    out = torch.zeros_like(lam)
    for _ in range(reps):
        F.action(lam, out)
torch.cuda.synchronize()
print("done", which, nx, reps)
