"""Fused Helmholtz apply: reference ordering ([u; v], H1Space numbering) vs plan-native ordering (pairs, owned dofs of a patch
contiguous), same plan, same box, alternating.  usage: native_apply.py [nx=1024] [nb=4] [reps=30] [refine=-1]     (environment: CUDDH_HELM_NB5_MFMA, CUDDH_HELM_LANE ... select the kernel)"""
import math
import os
import sys
from pathlib import Path

import torch

os.environ.setdefault("CUDDH_PLAN_AFFINE", "0")
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
refine = int(sys.argv[4]) if len(sys.argv) > 4 else -1
dev = torch.device("cuda:0")
cd.use_torch_stream()
if refine >= 0:
    mesh = cd.Mesh2D.load(Path(__file__).resolve().parents[2] / "tests" / "golden" / "unstructured_square").refined(refine)
else:
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
n = fem.size()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
omega = math.pi * nx / 32
A = cd.HelmholtzOperator(omega, torch.ones(n, dtype=torch.float64, device=dev), torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
print(f"{mesh.n_elem()} elements, n_basis {nb}, {n} dofs; kernel {A.kernel()}; native ordering: {A.has_native()}")
x = torch.rand(2 * n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
z = torch.empty_like(x)
zy = torch.empty_like(x)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


b_alg = A.bytes_per_apply()
if A.has_native():
    A.to_native(x, z)
for rnd in range(3):
    t = timed(lambda: A.action(x, y))
    line = f"reference ordering {t * 1e6:7.1f} us  {b_alg / t / 1e9:7.1f} GB/s algorithmic ({b_alg / t / 8e12:.3f} of 8 TB/s), layout bytes {A.bytes_per_apply(True) / 1e9:.4f} GB"
    if A.has_native():
        tn = timed(lambda: A.action_native(z, zy))
        line += f" | native ordering {tn * 1e6:7.1f} us  {b_alg / tn / 1e9:7.1f} GB/s ({b_alg / tn / 8e12:.3f}), layout bytes {A.bytes_native() / 1e9:.4f} GB"
    print(line)
if A.has_native():
    tp = timed(lambda: A.to_native(x, z))
    tq = timed(lambda: A.from_native(zy, y))
    print(f"to_native {tp * 1e6:.1f} us, from_native {tq * 1e6:.1f} us (once per solve each)")
