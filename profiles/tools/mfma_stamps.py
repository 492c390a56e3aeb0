"""Where a 16-element batch of helm_mfma_kernel spends its life: phase time stamps (100 MHz clock) of wavefront 0 of every batch,
recorded with CUDDH_HELM_STAMPS=1.  usage: [CUDDH_HELM_MFMA_STAGE=1] mfma_stamps.py nx nb [refine=-1]   (native ordering)"""
import ctypes as C
import math
import os
import sys
from pathlib import Path

import numpy as np
import torch

os.environ["CUDDH_HELM_STAMPS"] = "1"
os.environ.setdefault("CUDDH_PLAN_AFFINE", "0")
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402
from cuddhelmholtz_amd import _native as N  # noqa: E402

nx, nb = int(sys.argv[1]), int(sys.argv[2])
refine = int(sys.argv[3]) if len(sys.argv) > 3 else -1
dev = torch.device("cuda:0")
cd.use_torch_stream()
if refine >= 0:
    mesh = cd.Mesh2D.load(Path(__file__).resolve().parents[2] / "tests" / "golden" / "unstructured_square").refined(refine)
else:
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
n = fem.size()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
A = cd.HelmholtzOperator(math.pi * max(nx, 32) / 32, torch.ones(n, dtype=torch.float64, device=dev), torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
x = torch.rand(2 * n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
for _ in range(3):
    A.action_native(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
A.action_native(x, y)
e1.record()
torch.cuda.synchronize()
n_patches = (mesh.n_elem() + 15) // 16
st = np.zeros((n_patches, 8), dtype=np.uint64)
while N.lib.cuddh_helmholtz_read_stamps(A._h, st.ctypes.data_as(C.c_void_p), n_patches) != 0 and n_patches > 1:
    n_patches -= 1  # the plan may hold fewer batches than ceil(n_elem / 16) ... or more: only the first n_patches are read
    st = np.zeros((n_patches, 8), dtype=np.uint64)
t = st.astype(np.float64) * 0.01  # microseconds
t = t[t[:, 7] > 0]
t0 = t[:, 0].min()
names = ["start -> x (and staged metric data) in LDS", "-> element values in registers, LDS cleared", "-> stiffness slices done", "-> mass slices done",
         "-> colour phases done", "-> faces done", "-> write-out done"]
print(f"kernel: {A.kernel()} STAGE={os.environ.get('CUDDH_HELM_MFMA_STAGE', '0')}   launch (with stamps): {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} batches")
print(f"last start {t[:, 0].max() - t0:.1f} us, last end {t[:, 7].max() - t0:.1f} us")
d = np.diff(t, axis=1)
for k, name in enumerate(names):
    print(f"  {name:50s} mean {d[:, k].mean():7.2f} us   median {np.median(d[:, k]):7.2f}   p90 {np.percentile(d[:, k], 90):7.2f}")
life = t[:, 7] - t[:, 0]
print(f"  batch life: mean {life.mean():.2f} us, median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}")
conc = [int(np.sum((t[:, 0] <= s) & (t[:, 7] > s))) for s in np.linspace(t0, t[:, 7].max(), 9)[1:-1]]
print("  batches alive at 1/8 .. 7/8 of the launch:", conc)
