"""Multi-GPU DDH: subdomains sharded over ranks, one process per GPU.

The DDH local solves of one `action` are independent (one subdomain reads its own
trace slots and writes its neighbours', reference source/DDH.cpp:429-440,222,312),
so the subdomain range is split into `world` contiguous pieces.  Every rank keeps
the whole trace vector (26 MB at 1024^2): a rank's solves fill only the slots its
subdomains write, the rest stays zero, and one all-reduce (sum with zeros: exact,
so the N-rank result is bitwise the 1-rank result) reassembles the vector.  The
Krylov vectors are replicated, so dots and axpys need no communication.

`engine` is anything with the DDH sharded entry points
(`local_traces(d0, d1, f, lam, update)`, `local_solution(d0, d1, lam, f, u, zero_u)`):
`cuddhelmholtz_amd.DDH` on the GPU (torch.distributed backend "nccl" = RCCL over
xGMI), or a CPU stand-in in the gloo tests of this host logic.
"""
from __future__ import annotations


def partition(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced range of `n_items` for `rank` of `world`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


class ShardedDDH:
    def __init__(self, engine, n_domains: int, rank: int = 0, world: int = 1, group=None, always_reduce: bool = False):
        self.engine = engine
        self.rank, self.world, self.group = rank, world, group
        self.always_reduce = always_reduce  # issue the collective even for world == 1 (exercises the RCCL path on one GPU)
        self.d0, self.d1 = partition(n_domains, rank, world)

    def _all_reduce(self, t):
        if self.world > 1 or self.always_reduce:
            import torch.distributed as dist

            dist.all_reduce(t, group=self.group)

    def traces(self, f, lam, out) -> None:
        """out <- outgoing traces of all subdomains for forcing f (or None) and incoming traces lam (or None)."""
        out.zero_()
        self.engine.local_traces(self.d0, self.d1, f, lam, out)
        self._all_reduce(out)

    def rhs(self, f, b) -> None:
        """reference DDH::rhs (source/DDH.cpp:641-667)"""
        self.traces(f, None, b)

    def action(self, x, y) -> None:
        """reference DDH::action (source/DDH.cpp:611-639): y = x - T x"""
        self.traces(None, x, y)
        y.mul_(-1.0).add_(x)

    def postprocess(self, lam, f, u) -> None:
        """reference DDH::postprocess (source/DDH.cpp:669-695); partition-of-unity sums cross ranks at shared nodes"""
        u.zero_()
        self.engine.local_solution(self.d0, self.d1, lam, f, u, False)
        self._all_reduce(u)
