"""Fused complex Helmholtz apply and single operators on the reference's unstructured_square fixture refined r times
(119 * 4^r quads, every element with its own metric tensor): algorithmic GB/s.  usage: unstructured_apply.py [r] [nb]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import cuddhelmholtz_amd as cd  # noqa: E402
from cuddhelmholtz_amd.meshtools import refine_quads  # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = ROOT / "tests" / "golden" / "unstructured_square"
n_pts, n_elem = (int(v) for v in (d / "info.txt").read_text().split())
xy = np.loadtxt(d / "coordinates.txt").reshape(n_pts, 2)
elems = np.loadtxt(d / "elements.txt", dtype=np.int64).reshape(n_elem, 4)
xy, elems = refine_quads(xy, elems, r)
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.from_vertices(xy, elems)
fem = cd.H1Space(mesh, cd.Basis(nb))
n = fem.size()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
g = torch.Generator(device="cpu").manual_seed(1)
a2 = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev)
A = cd.HelmholtzOperator(9.0, a2, torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
x = torch.rand(2 * n, generator=g, dtype=torch.float64).to(dev)
y = torch.empty_like(x)


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


t = timeit(lambda: A.action(x, y))
b = A.bytes_per_apply()
print(f"unstructured r={r}: {len(elems)} quads, nb={nb}, {n} dofs; fused complex apply {t * 1e6:.1f} us, {b / t / 1e9:.1f} GB/s algorithmic "
      f"({b / 1e6:.1f} MB; layout {A.bytes_per_apply(True) / 1e6:.1f} MB)")
S = cd.StiffnessMatrix(fem)
yS = torch.empty(n, dtype=torch.float64, device=dev)
t = timeit(lambda: S.action(x[:n], yS))
bS = len(elems) * (3 * (nb + 1) ** 2 * 8 + nb * nb * 4) + n * 16
print(f"unstructured r={r}: stiffness action(x,y) {t * 1e6:.1f} us, {bS / t / 1e9:.1f} GB/s algorithmic ({bS / 1e6:.1f} MB)")
