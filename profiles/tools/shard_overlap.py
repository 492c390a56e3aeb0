"""Times one rank's share of a DDH action at 1024^2 on ONE GPU (messages skipped: dry run) with the exchange posted
after all local solves vs. boundary subdomains first on a second stream.  Usage: shard_overlap.py [nx] [world] [rank]"""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402
from cuddhelmholtz_amd.dist import NeighbourShardedDDH  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
grid = tuple(int(v) for v in sys.argv[4].split("x")) if len(sys.argv) > 4 else None  # e.g. 2x4: rectangles instead of strips
dev = torch.device("cuda:0")
cd.use_torch_stream()
omega = math.pi * nx / 32
fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(4))
F = cd.DDH(omega, np.ones(fem.size()), fem, nx, nx)
nd, n = F.info()["n_domains"], F.size()
x = torch.rand(n, dtype=torch.float32, device=dev)
y = torch.zeros_like(x)
from cuddhelmholtz_amd.dist import rank_grid_map  # noqa: E402

ndx = nx // F.info()["nel1d"]
dom_rank = rank_grid_map(ndx, nd // ndx, *grid) if grid else None
for overlap in (False, True):
    sh = NeighbourShardedDDH(F, nd, rank, world, device=dev, overlap=overlap, set_stream=cd.use_torch_stream, dry_run=True, dom_rank=dom_rank)
    sh.action(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sh.action(x, y)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    msg = sum(i.numel() for i in sh.send_idx.values()) * 4
    print(f"nx={nx} rank {rank}/{world}{' grid ' + sys.argv[4] if grid else ''}: {sh.ex.domains.size} subdomains, {len(sh.send_idx)} neighbours, "
          f"{sum(b - a for a, b in sh.ex.boundary_ranges)} boundary subdomains in {len(sh.ex.boundary_ranges)} ranges, "
          f"{msg / 1024:.1f} KiB sent per action, overlap={overlap}: {t * 1e3:.2f} ms per action")
