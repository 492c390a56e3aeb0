"""One rank's share of the partitioned global Helmholtz apply at 1024^2 on ONE GPU (messages replaced by local copies of
the same size): set-up time, message sizes, local apply and pack/unpack times.  usage: shard_helmholtz.py [nx] [world] [rank]"""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402
from cuddhelmholtz_amd.dist import ShardedHelmholtz  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
cd.use_torch_stream()
mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
fem = cd.H1Space(mesh, cd.Basis(4))
fs = cd.FaceSpace(fem, mesh.boundary_edges())
t0 = time.time()
A = ShardedHelmholtz(cd, math.pi * nx / 32, np.ones(fem.size()), np.ones(fs.size()), mesh, fem, fs, rank, world, device=dev)
t_setup = time.time() - t0
p = A.part
x = torch.rand(2 * A.n_loc, dtype=torch.float64, device=dev)
x.index_fill_(0, A.halo_idx, 0.0)
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


def local_step():
    xs = A._scratch
    xs.copy_(x)
    out = A.pack_x(xs)                                     # what would be sent to the halo holders
    A.unpack_x(xs, {s: torch.zeros(2 * i.numel(), dtype=torch.float64, device=dev) for s, i in A.halo_from.items()})
    A.op.action(xs, y)
    back = A.pack_y(y)                                     # partial sums for the owners
    A.unpack_y(y, {s: torch.zeros(2 * i.numel(), dtype=torch.float64, device=dev) for s, i in A.own_to.items()})
    return out, back


t_all = timeit(local_step)
t_op = timeit(lambda: A.op.action(x, y))
sx = sum(i.numel() for i in A.own_to.values()) * 16
sy = sum(i.numel() for i in A.halo_from.values()) * 16
print(f"nx={nx} rank {rank}/{world}: {len(p.my_elems)} elements, {p.n_loc} local dofs ({p.owned.size} owned, {p.halo.size} halo), "
      f"neighbours x:{sorted(A.own_to)} y:{sorted(A.halo_from)}, messages {sx / 1024:.0f} KiB out (x) + {sy / 1024:.0f} KiB out (y); "
      f"set-up {t_setup:.1f} s; local fused apply {t_op * 1e6:.1f} us; with scratch copy, pack and unpack {t_all * 1e6:.1f} us")
