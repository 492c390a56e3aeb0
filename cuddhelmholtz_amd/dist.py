"""Multi-GPU DDH: subdomains sharded over ranks, one process per GPU.

The DDH local solves of one `action` are independent (one subdomain reads its own
trace slots and writes its neighbours', reference source/DDH.cpp:429-440,222,312),
so the subdomain range is split into `world` contiguous pieces.  Every rank keeps
the whole trace vector (26 MB at 1024^2): a rank's solves fill only the slots its
subdomains write, the rest stays zero, and one all-reduce (sum with zeros: exact,
so the N-rank result is bitwise the 1-rank result) reassembles the vector.  The
Krylov vectors are replicated, so dots and axpys need no communication.

`NeighbourShardedDDH` is the partitioned form of the same thing: a rank owns the trace slots
its subdomains READ, every vector it holds is zero outside those slots, the traces its subdomains write
into slots owned by another rank travel by grouped point-to-point messages (ncclGroupStart /
ncclSend / ncclRecv / ncclGroupEnd under `torch.distributed.batch_isend_irecv`) to the (at most two, for
contiguous ranges of a structured block grid) neighbouring ranks, and GMRES reduces its inner products
over the ranks (`cuddhelmholtz_amd.gmres(..., reduce=)`).

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
    def __init__(self, engine, n_domains: int, rank: int = 0, world: int = 1, group=None, always_reduce: bool = False,
                 host_staging: bool = False):
        self.engine = engine
        self.rank, self.world, self.group = rank, world, group
        self.always_reduce = always_reduce  # issue the collective even for world == 1 (exercises the RCCL path on one GPU)
        self.host_staging = host_staging    # process group cannot carry device tensors (gloo rehearsal of the GPU path)
        self.d0, self.d1 = partition(n_domains, rank, world)

    def _all_reduce(self, t):
        if self.world > 1 or self.always_reduce:
            import torch.distributed as dist

            if self.host_staging and t.device.type != "cpu":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
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


def _runs(ids):
    """sorted integer ids -> list of half-open (begin, end) runs"""
    out = []
    for d in ids:
        d = int(d)
        if out and out[-1][1] == d:
            out[-1][1] = d + 1
        else:
            out.append([d, d + 1])
    return [tuple(r) for r in out]


class TraceExchange:
    """Who owns, sends and receives which trace slots, from the slot table B(mx_fdof, 2, n_domains) of the DDH
    constructor (reference source/DDH.cpp:425-440): B(i,0,S) is the slot subdomain S reads for its face dof i,
    B(i,1,S) the slot it writes; slot t stands for entries t and n_lambda + t of a trace vector.

    owner(slot) = rank of the subdomain that reads it (else of the one that writes it; slots nobody touches --
    the reference's orphan slots at cross points -- stay zero everywhere)."""

    def __init__(self, B, n_domains: int, mx_fdof: int, n_lambda: int, rank: int, world: int):
        import numpy as np

        B = np.asarray(B, dtype=np.int64).reshape(n_domains, 2, mx_fdof)
        upper = np.asarray([partition(n_domains, r, world)[1] for r in range(world)])
        dom_rank = np.searchsorted(upper, np.arange(n_domains), side="right")
        dom_of = np.repeat(np.arange(n_domains), mx_fdof)

        def slot_to_domain(col):
            t = B[:, col, :].ravel()
            m = t >= 0
            if t[m].size and (t[m].max() >= n_lambda or np.unique(t[m]).size != int(m.sum())):
                raise ValueError("DDH slot table: a slot is used by two subdomains or is out of range")
            out = np.full(n_lambda, -1, dtype=np.int64)
            out[t[m]] = dom_of[m]
            return out

        reader, writer = slot_to_domain(0), slot_to_domain(1)
        owner_dom = np.where(reader >= 0, reader, writer)
        owner = np.where(owner_dom >= 0, dom_rank[np.maximum(owner_dom, 0)], -1)
        wrank = np.where(writer >= 0, dom_rank[np.maximum(writer, 0)], -1)

        self.rank, self.world, self.n_lambda = rank, world, n_lambda
        self.d0, self.d1 = partition(n_domains, rank, world)
        self.owned_slots = np.flatnonzero(owner == rank)
        self.send_slots, self.recv_slots = {}, {}
        for s in range(world):
            if s == rank:
                continue
            snd = np.flatnonzero((wrank == rank) & (owner == s))
            rcv = np.flatnonzero((owner == rank) & (wrank == s))
            if snd.size:
                self.send_slots[s] = snd
            if rcv.size:
                self.recv_slots[s] = rcv
        sent = np.concatenate(list(self.send_slots.values())) if self.send_slots else np.zeros(0, dtype=np.int64)
        boundary = np.unique(writer[sent])
        self.boundary_ranges = _runs(boundary)
        mask = np.ones(self.d1 - self.d0, dtype=bool)
        mask[boundary - self.d0] = False
        self.interior_ranges = _runs(np.flatnonzero(mask) + self.d0)

    def entries(self, slots):
        """vector entries (lambda and mu halves) of a set of slots"""
        import numpy as np

        return np.concatenate([slots, slots + self.n_lambda])


class NeighbourShardedDDH:
    """DDH over ranks with partitioned trace vectors and neighbour exchange (module docstring).

    engine: as for ShardedDDH, plus `table("B")` and `info()`.  Vectors keep the full length 2*n_lambda so the
    kernels' slot indices stay valid, but on each rank only the owned entries are ever non-zero.
    host_staging: move message payloads through host memory (needed when the process group cannot carry device
    tensors, e.g. gloo; RCCL sends device buffers directly).
    overlap: launch the subdomains that feed other ranks first on a second stream, start the exchange behind them
    and run the interior subdomains meanwhile (`set_stream` must point the engine at torch's current stream)."""

    def __init__(self, engine, n_domains: int, rank: int = 0, world: int = 1, group=None, device=None, host_staging: bool = False,
                 overlap: bool = False, set_stream=None, dry_run: bool = False):
        import torch

        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.host_staging, self.overlap, self.set_stream, self.dry_run = host_staging, overlap, set_stream, dry_run
        info = engine.info()
        self.ex = TraceExchange(engine.table("B"), n_domains, info["mx_fdof"], info["n_lambda"], rank, world)
        self.d0, self.d1 = self.ex.d0, self.ex.d1
        self.device = torch.device("cpu") if device is None else torch.device(device)
        as_idx = lambda a: torch.from_numpy(self.ex.entries(a)).to(self.device)  # noqa: E731
        self.send_idx = {s: as_idx(a) for s, a in self.ex.send_slots.items()}
        self.recv_idx = {s: as_idx(a) for s, a in self.ex.recv_slots.items()}
        self.owned_idx = as_idx(self.ex.owned_slots)
        self._side = torch.cuda.Stream(self.device) if (overlap and self.device.type == "cuda") else None

    # ---- communication
    def reduce(self, t) -> None:
        """sum a small tensor over the ranks in place (inner products of the partitioned Krylov vectors)"""
        if self.world > 1 and not self.dry_run:
            import torch.distributed as dist

            if self.host_staging and t.device.type != "cpu":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.group)

    def _start_exchange(self, out):
        """pack the traces written for other ranks and post the grouped sends/receives"""
        import torch
        import torch.distributed as dist

        stage = (lambda t: t.cpu()) if self.host_staging else (lambda t: t)  # noqa: E731
        sbuf = {s: stage(out.index_select(0, idx)) for s, idx in self.send_idx.items()}
        rbuf = {s: torch.empty(idx.numel(), dtype=out.dtype, device="cpu" if self.host_staging else out.device)
                for s, idx in self.recv_idx.items()}
        if self.dry_run:  # timing one rank's share without peers: buffers are packed, nothing is sent
            for t in rbuf.values():
                t.zero_()
            return [], sbuf, rbuf
        ops = []
        for s in sorted(set(sbuf) | set(rbuf)):
            if s in sbuf:
                ops.append(dist.P2POp(dist.isend, sbuf[s], s, self.group))
            if s in rbuf:
                ops.append(dist.P2POp(dist.irecv, rbuf[s], s, self.group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, sbuf, rbuf

    def _finish_exchange(self, out, pending) -> None:
        reqs, sbuf, rbuf = pending
        for r in reqs:
            r.wait()
        for idx in self.send_idx.values():
            out.index_fill_(0, idx, 0)  # those slots belong to the receiver
        for s, idx in self.recv_idx.items():
            out.index_copy_(0, idx, rbuf[s].to(out.device))
        del sbuf

    # ---- the operator
    def traces(self, f, lam, out) -> None:
        """out <- traces written by this rank's subdomains into slots it owns + traces received from its neighbours"""
        out.zero_()
        if self.world == 1:
            self.engine.local_traces(self.d0, self.d1, f, lam, out)
            return
        if not self.overlap:
            self.engine.local_traces(self.d0, self.d1, f, lam, out)
            self._finish_exchange(out, self._start_exchange(out))
            return
        if self._side is None:  # no streams on this device: same order, nothing to overlap
            for a, b in self.ex.boundary_ranges:
                self.engine.local_traces(a, b, f, lam, out)
            pending = self._start_exchange(out)
            for a, b in self.ex.interior_ranges:
                self.engine.local_traces(a, b, f, lam, out)
            self._finish_exchange(out, pending)
            return
        import torch

        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self.set_stream()
            for a, b in self.ex.boundary_ranges:
                self.engine.local_traces(a, b, f, lam, out)
            pending = self._start_exchange(out)
        self.set_stream()
        for a, b in self.ex.interior_ranges:
            self.engine.local_traces(a, b, f, lam, out)
        main.wait_stream(self._side)
        self._finish_exchange(out, pending)

    def rhs(self, f, b) -> None:
        """reference DDH::rhs (source/DDH.cpp:641-667), owned entries only"""
        self.traces(f, None, b)

    def action(self, x, y) -> None:
        """reference DDH::action (source/DDH.cpp:611-639): y = x - T x on the owned entries"""
        self.traces(None, x, y)
        y.mul_(-1.0).add_(x)

    def postprocess(self, lam, f, u) -> None:
        """reference DDH::postprocess (source/DDH.cpp:669-695); u is summed over the ranks (once per solve)"""
        u.zero_()
        self.engine.local_solution(self.d0, self.d1, lam, f, u, False)
        self.reduce(u)

    def full(self, v):
        """the whole trace vector on every rank (sum of the partitioned copies); for checks and output"""
        w = v.clone()
        self.reduce(w)
        return w

    def solve(self, lam, b, m: int, maxit: int, tol: float, gmres=None, **kw):
        """GMRES on the partitioned vectors; `gmres` defaults to cuddhelmholtz_amd.gmres"""
        if gmres is None:
            from .api import gmres
        return gmres(lam.numel(), lam, self.action, b, m, maxit, tol, reduce=self.reduce, **kw)
