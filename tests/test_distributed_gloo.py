"""world_size-2 gloo test (CPU) of the multi-GPU host logic: subdomain sharding + all-reduce
reassembly reproduces the single-rank rhs / action / GMRES solve / postprocess bitwise.
The CPU oracle stands in for the GPU engine (test infrastructure only)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cuddhelmholtz_amd.dist import ShardedDDH, partition


class OracleEngine:
    """DDH sharded entry points backed by the oracle (fp64), on CPU tensors."""

    def __init__(self, O):
        self.O = O

    def local_traces(self, d0, d1, f, lam, update):
        _, upd = self.O.solve(x=None if f is None else f.numpy(), lam=None if lam is None else lam.numpy(), d0=d0, d1=d1)
        update += torch.from_numpy(upd)

    def local_solution(self, d0, d1, lam, f, u, zero_u):
        y, _ = self.O.solve(x=f.numpy(), want_y=True, lam=lam.numpy(), want_update=False, d0=d0, d1=d1)
        if zero_u:
            u.zero_()
        u += torch.from_numpy(y)


def build_case():
    nx, nb = 8, 4
    omega = 2 * math.pi * nx / 10
    d = oracle.Discretization(oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), nb)
    h_a = d.nodal(oracle.alpha_disk)
    f = np.concatenate([oracle.linear_functional(d, oracle.gaussians(omega)), np.zeros(d.ndof)])
    O = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    return O, torch.from_numpy(f), d.ndof


def solve(sh, O, f, ndof):
    n = O.size
    b = torch.zeros(n, dtype=torch.float64)
    sh.rhs(f, b)

    def A(x):
        y = torch.zeros(n, dtype=torch.float64)
        sh.action(torch.from_numpy(np.ascontiguousarray(x)), y)
        return y.numpy()

    lam, info = oracle.gmres(A, b.numpy(), m=10, maxit=3, tol=1e-12)
    u = torch.zeros(2 * ndof, dtype=torch.float64)
    sh.postprocess(torch.from_numpy(lam), f, u)
    return b, torch.from_numpy(lam), u, info["num_matvec"]


def worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O, f, ndof = build_case()
        sh = ShardedDDH(OracleEngine(O), O.t.n_domains, rank, world)
        b, lam, u, nmv = solve(sh, O, f, ndof)
        torch.save({"b": b, "lam": lam, "u": u, "nmv": nmv}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_covers_everything():
    for n in (1, 4, 7, 16384, 65536):
        for w in (1, 2, 3, 8):
            ranges = [partition(n, r, w) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition(4, 4, 4)


def test_two_ranks_reproduce_one_rank(tmp_path):
    O, f, ndof = build_case()
    single = solve(ShardedDDH(OracleEngine(O), O.t.n_domains), O, f, ndof)
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True)
        assert torch.equal(got["b"], single[0])
        assert torch.equal(got["lam"], single[1])
        # y receives floating point sums at nodes shared by subdomains of different ranks: order differs
        assert torch.allclose(got["u"], single[2], rtol=1e-13, atol=1e-15)
        assert got["nmv"] == single[3]
