"""Times DDH::action for several (nx, n_basis, kernel) and prints subdomain RK2 steps per second and the algorithmic
fp32 FLOP rate (2*nb*2 + 6 + 2*nb*2 + 1 per node per sweep, 26 per node per step for the update).
usage: ddh_rates.py "nx,nb,kernel" ..."""
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402

dev = torch.device("cuda:0")
cd.use_torch_stream()
for spec in sys.argv[1:]:
    nx, nb, kernel = (int(v) for v in spec.split(","))
    omega = math.pi * nx / 32.0
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, np.ones(fem.size()), fem, nx, nx, kernel=kernel)
    info = F.info()
    lam = torch.rand(F.size(), dtype=torch.float32, device=dev)
    out = torch.zeros_like(lam)
    F.action(lam, out)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        F.action(lam, out)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    nodes = (nb * info["nel1d"]) ** 2
    steps = 5 * info["nt"] * info["n_domains"]
    flops = steps * nodes * (2 * (8 * nb + 7) + 26)
    print(f"nx={nx} nb={nb} kernel={info['kernel']} subdomains={info['n_domains']} nodes/subdomain={nodes} nt={info['nt']}: "
          f"{t * 1e3:.1f} ms per action, {steps / t / 1e9:.3f} G subdomain-steps/s, {flops / t / 1e12:.1f} TFLOP/s "
          f"({100 * flops / t / 157.3e12:.0f} % of fp32 vector peak), {2 * fem.size() / t / 1e6:.1f} M DoF*iter/s")
    del F, fem
