"""Where a wavefront of the fused apply spends its life: phase time stamps (100 MHz clock) recorded by helm_lane_kernel with
CUDDH_HELM_STAMPS=1.  usage: CUDDH_HELM_PRE=0|1 lane_stamps.py [nx=1024] [native]   (native: the plan-native vector ordering)"""
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

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = 4
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(nb))
n = fem.size()
fs = cd.FaceSpace(fem, mesh.boundary_edges())
A = cd.HelmholtzOperator(math.pi * nx / 32, torch.ones(n, dtype=torch.float64, device=dev), torch.ones(fs.size(), dtype=torch.float64, device=dev), fem, fs)
x = torch.rand(2 * n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
native = len(sys.argv) > 2 and sys.argv[2] == "native"
apply = A.action_native if native else A.action
for _ in range(3):
    apply(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
apply(x, y)
e1.record()
torch.cuda.synchronize()
n_patches = (nx // 8) ** 2
st = np.zeros((n_patches, 8), dtype=np.uint64)
rc = N.lib.cuddh_helmholtz_read_stamps(A._h, st.ctypes.data_as(C.c_void_p), n_patches)
assert rc == 0, rc
t = st.astype(np.float64) * 0.01  # microseconds
t0 = t[:, 0].min()
names = ["start -> x in LDS (indices, x gather, first metric requests)", "-> element values in registers", "-> slices done (metric round trips + arithmetic)",
         "-> colour phases done", "-> faces done", "-> write-out done (stores complete)"]
print(f"kernel: {A.kernel()}{' (native ordering)' if native else ''}   launch (with stamps): {e0.elapsed_time(e1) * 1e3:.1f} us, {n_patches} wavefronts")
print(f"first start {0.0:.1f} us, last start {t[:, 0].max() - t0:.1f} us, last end {t[:, 6].max() - t0:.1f} us")
d = np.diff(t[:, :7], axis=1)
for k, name in enumerate(names):
    print(f"  {name:70s} mean {d[:, k].mean():7.2f} us   median {np.median(d[:, k]):7.2f}   p90 {np.percentile(d[:, k], 90):7.2f}")
life = t[:, 6] - t[:, 0]
print(f"  wavefront life: mean {life.mean():.2f} us, median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}")
order = np.argsort(t[:, 0])
conc = [int(np.sum((t[:, 0] <= s) & (t[:, 6] > s))) for s in np.linspace(t0, t[:, 6].max(), 9)[1:-1]]
print("  wavefronts alive at 1/8 .. 7/8 of the launch:", conc)
