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


def native_trace_exchange(B, n_domains: int, mx_fdof: int, n_lambda: int, rank: int, world: int):
    """The ownership / send / receive slot lists as the C++ multi-GPU host computes them (cuddh::TraceExchangePlan,
    csrc/src/multigpu.cpp): (owned, {peer: send slots}, {peer: recv slots}).  Must equal TraceExchange's."""
    import ctypes as C

    import numpy as np

    from ._native import lib

    B = np.ascontiguousarray(np.asarray(B, dtype=np.int32).reshape(-1))
    pB = B.ctypes.data_as(C.c_void_p)

    def query(which, peer):
        n = lib.cuddh_trace_exchange_query(pB, n_domains, mx_fdof, n_lambda, rank, world, which, peer, None)
        if n < 0:
            raise RuntimeError("cuddh_trace_exchange_query failed")
        out = np.empty(n, dtype=np.int32)
        if n:
            lib.cuddh_trace_exchange_query(pB, n_domains, mx_fdof, n_lambda, rank, world, which, peer, out.ctypes.data_as(C.c_void_p))
        return out

    native_trace_exchange.last_split = (query(3, 0), query(4, 0))  # subdomains of the boundary / interior launch (split schedule)
    owned = query(0, 0)
    send = {p: a for p in range(world) if p != rank and (a := query(1, p)).size}
    recv = {p: a for p in range(world) if p != rank and (a := query(2, p)).size}
    return owned, send, recv


def ddh_solve_multi_gpu(nx: int, nb: int, omega: float, h_a, h_f, world: int, m: int = 20, maxit: int = 100, tol: float = 1e-4,
                        force_rccl=False, split_schedule: bool = False, rank_grid=None):
    """rhs -> gmres -> postprocess on `world` GPUs of this process through the C++ host (cuddh::ddh_solve_multi_gpu: one
    host thread per device, RCCL send/recv for the traces, ncclAllReduce for the inner products).  Host arrays in, (u, info) out.
    force_rccl: False / 0 auto, True / 1 RCCL also for one rank, 2 the loopback test transport (the ranks are threads sharing
    device 0: everything of the N > 1 path except the RCCL calls, on a one-GPU box).  split_schedule: boundary subdomains first
    (one listed launch with issue priority on a second stream), exchange behind them, interior meanwhile."""
    import ctypes as C

    import numpy as np

    from . import _native as N

    h_a = np.ascontiguousarray(h_a, dtype=np.float64)
    h_f = np.ascontiguousarray(h_f, dtype=np.float64)
    u = np.zeros_like(h_f)
    res = N.MultiGpuResult()
    hist = np.zeros(maxit + 2)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N.check_capi(N.lib.cuddh_ddh_solve_multi_gpu(nx, nb, float(omega), vp(h_a), vp(h_f), vp(u), world, m, maxit, float(tol),
                                                int(force_rccl) | (4 if split_schedule else 0) | ((rank_grid[0] << 8) | (rank_grid[1] << 16) if rank_grid else 0),
                                                C.byref(res), vp(hist)), "ddh_solve_multi_gpu")
    info = {k: getattr(res, k) for k, _ in N.MultiGpuResult._fields_}
    info["res_norm"] = hist[: res.n_res].tolist()
    return u, info


def helmholtz_multi_gpu(mesh, nb: int, omega: float, h_a2x, h_ax, h_x, world: int, transport: int = 0, reps: int = 0, m: int = 20,
                        maxit: int = 0, tol: float = 1e-8):
    """The fused Helmholtz operator partitioned over `world` GPUs of this process through the C++ host
    (cuddh::helmholtz_multi_gpu: element partition, sub-mesh operators, halo exchanges through the HIP pack / unpack kernels,
    RCCL send / recv or -- transport 2 -- loopback ranks sharing device 0).  maxit == 0: returns (A h_x, info) and times `reps`
    applies; maxit > 0: returns (GMRES(m) solution of A y = h_x, info).  Host arrays in the global numbering."""
    import ctypes as C

    import numpy as np

    from . import _native as N

    h_a2x = np.ascontiguousarray(h_a2x, dtype=np.float64)
    h_ax = np.ascontiguousarray(h_ax, dtype=np.float64)
    h_x = np.ascontiguousarray(h_x, dtype=np.float64)
    y = np.zeros_like(h_x)
    res = N.HelmholtzMultiGpuResult()
    hist = np.zeros(maxit + 2)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N.check_capi(N.lib.cuddh_helmholtz_multi_gpu(mesh._h, nb, float(omega), vp(h_a2x), vp(h_ax), vp(h_x), vp(y), world, transport, reps, m, maxit,
                                                float(tol), C.byref(res), vp(hist)), "helmholtz_multi_gpu")
    info = {k: getattr(res, k) for k, _ in N.HelmholtzMultiGpuResult._fields_}
    info["res_norm"] = hist[: res.n_res].tolist()
    return y, info


def native_helmholtz_partition(mesh, fem, fs, rank: int, world: int):
    """HelmholtzPartition::build of the C++ host as a dict of numpy arrays (host only; tests compare it with HelmholtzPartition)"""
    import ctypes as C

    import numpy as np

    from . import _native as N

    def q(which, peer=0):
        n = N.lib.cuddh_helmholtz_partition_query(mesh._h, fem._h, fs._h, rank, world, which, peer, None)
        if n < 0:
            raise RuntimeError(N.last_error())
        out = np.zeros(max(n, 1), dtype=np.int32)
        N.lib.cuddh_helmholtz_partition_query(mesh._h, fem._h, fs._h, rank, world, which, peer, out.ctypes.data_as(C.c_void_p))
        return out[:n].astype(np.int64)

    return {"my_elems": q(0), "l2g": q(1), "owned": q(2), "halo": q(3), "face_l2g": q(6), "faces": q(7),
            "own_to": {s: v for s in range(world) if s != rank and (v := q(4, s)).size},
            "halo_from": {s: v for s in range(world) if s != rank and (v := q(5, s)).size}}


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


def rank_grid_map(ndx: int, ndy: int, gx: int, gy: int):
    """Subdomain -> rank for a gx x gy grid of ranks over the ndx x ndy block grid of a structured DDH (subdomain
    bx + ndx * by, reference source/DDH.cpp:341-344): contiguous, balanced rectangles, rank = rx + gx * ry -- at most four face
    neighbours per rank.  gx = 1 gives the strips of block rows `partition` describes."""
    import numpy as np

    if gx < 1 or gy < 1 or gx > ndx or gy > ndy:
        raise ValueError("rank grid does not fit the block grid")
    rx = (np.arange(ndx) * gx) // ndx
    ry = (np.arange(ndy) * gy) // ndy
    return (rx[None, :] + gx * ry[:, None]).reshape(-1).astype(np.int64)


class TraceExchange:
    """Who owns, sends and receives which trace slots, from the slot table B(mx_fdof, 2, n_domains) of the DDH
    constructor (reference source/DDH.cpp:425-440): B(i,0,S) is the slot subdomain S reads for its face dof i,
    B(i,1,S) the slot it writes; slot t stands for entries t and n_lambda + t of a trace vector.

    owner(slot) = rank of the subdomain that reads it (else of the one that writes it; slots nobody touches --
    the reference's orphan slots at cross points -- stay zero everywhere)."""

    def __init__(self, B, n_domains: int, mx_fdof: int, n_lambda: int, rank: int, world: int, dom_rank=None):
        import numpy as np

        B = np.asarray(B, dtype=np.int64).reshape(n_domains, 2, mx_fdof)
        if dom_rank is None:  # contiguous ranges (strips of block rows on a structured block grid)
            upper = np.asarray([partition(n_domains, r, world)[1] for r in range(world)])
            dom_rank = np.searchsorted(upper, np.arange(n_domains), side="right")
        else:  # any assignment, e.g. rank_grid_map
            dom_rank = np.asarray(dom_rank, dtype=np.int64)
            if dom_rank.shape != (n_domains,) or dom_rank.min() < 0 or dom_rank.max() >= world:
                raise ValueError("dom_rank: one rank in [0, world) per subdomain")
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
        self.domains = np.flatnonzero(dom_rank == rank)  # this rank's subdomains, increasing
        contiguous = self.domains.size > 0 and int(self.domains[-1] - self.domains[0]) + 1 == self.domains.size
        # [d0, d1) when the rank's subdomains are one range (always so without dom_rank), else None: use `domains`
        self.d0, self.d1 = (int(self.domains[0]), int(self.domains[-1]) + 1) if contiguous else (None, None)
        if self.domains.size == 0:
            self.d0 = self.d1 = 0
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
        self.interior_ranges = _runs(np.setdiff1d(self.domains, boundary, assume_unique=True))

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
                 overlap: bool = False, set_stream=None, dry_run: bool = False, dom_rank=None):
        import torch

        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.host_staging, self.overlap, self.set_stream, self.dry_run = host_staging, overlap, set_stream, dry_run
        info = engine.info()
        # dom_rank: subdomain -> rank (e.g. rank_grid_map for a gx x gy grid of ranks); default: contiguous ranges
        self.ex = TraceExchange(engine.table("B"), n_domains, info["mx_fdof"], info["n_lambda"], rank, world, dom_rank)
        self.d0, self.d1 = self.ex.d0, self.ex.d1
        self._lists = {}  # device lists of subdomain ids for the one-launch entry point
        # The split schedule launches boundary and interior subdomains separately.  A workgroup of the wavefront kernels
        # holds 4 wavefronts (8 subdomains for n_basis 8), so a boundary count that is not a multiple of that leaves both
        # launches with a partial workgroup -- one workgroup more than the unsplit launch, which at 8,192 subdomains per rank
        # (exactly one resident round) means a second round for it (+5 ms of 44).  Up to 7 interior subdomains therefore
        # join the boundary launch.
        import numpy as np

        boundary = np.asarray([d for a, b in self.ex.boundary_ranges for d in range(a, b)], dtype=np.int64)
        interior = np.asarray([d for a, b in self.ex.interior_ranges for d in range(a, b)], dtype=np.int64)
        pad = min((-boundary.size) % 8, interior.size) if boundary.size else 0
        self._ids = {"all": self.ex.domains, "boundary": np.sort(np.concatenate([boundary, interior[:pad]])), "interior": interior[pad:]}
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
    def _solve(self, which: str, f, lam, out) -> None:
        """local solves of "all" / "boundary" / "interior" subdomains of this rank: one range call when they are one range,
        one listed launch when the engine has that entry point (cuddh_hip_ddh_apply_list_*), else range by range"""
        import torch

        ranges = _runs(self._ids[which])
        if not ranges:
            return
        listed = getattr(self.engine, "local_traces_listed", None)
        if len(ranges) == 1 or listed is None:
            for a, b in ranges:
                self.engine.local_traces(a, b, f, lam, out)
            return
        if which not in self._lists:
            ids = [s for a, b in ranges for s in range(a, b)]
            self._lists[which] = torch.tensor(ids, dtype=torch.int32, device=self.device)
        listed(self._lists[which], f, lam, out)

    def traces(self, f, lam, out) -> None:
        """out <- traces written by this rank's subdomains into slots it owns + traces received from its neighbours"""
        out.zero_()
        if self.world == 1:
            self._solve("all", f, lam, out)
            return
        if not self.overlap:
            self._solve("all", f, lam, out)
            self._finish_exchange(out, self._start_exchange(out))
            return
        if self._side is None:  # no streams on this device: same order, nothing to overlap
            self._solve("boundary", f, lam, out)
            pending = self._start_exchange(out)
            self._solve("interior", f, lam, out)
            self._finish_exchange(out, pending)
            return
        import torch

        # All wavefronts of a rank's local solves are resident at once and advance at the same rate: launched beside the
        # interior, a boundary range finishes WITH it (and a second boundary range queued behind the first then runs alone:
        # profiles/r02/overlap_timeline.txt, 44 + 8 ms).  So the boundary wavefronts take issue priority (s_setprio): they
        # finish first, the exchange starts, and the interior fills the rest of the step.
        # ONE launch for all boundary ranges, so that boundary + interior wavefronts together are exactly the residents of the
        # unsplit launch (queued one behind the other, the pieces leave SIMDs with one subdomain more than the rest: 49 ms).
        prio = getattr(self.engine, "set_wave_priority", None)
        main = torch.cuda.current_stream(self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self.set_stream()
            if prio:
                prio(True)
            self._solve("boundary", f, lam, out)
            if prio:
                prio(False)
            pending = self._start_exchange(out)
        self.set_stream()
        self._solve("interior", f, lam, out)
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
        import torch

        u.zero_()
        runs = _runs(self.ex.domains)
        listed = getattr(self.engine, "local_solution_listed", None)
        if len(runs) > 1 and listed is not None:  # a rectangle of the block grid: one listed launch
            if "all" not in self._lists:
                self._lists["all"] = torch.tensor(self.ex.domains, dtype=torch.int32, device=self.device)
            listed(self._lists["all"], lam, f, u, False)
        else:
            for a, b in runs:
                self.engine.local_solution(a, b, lam, f, u, False)
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


# ---------------------------------------------------------------------------------------------------------------------
# Global operator apply over several GPUs ("next"): element partition + halo exchange of shared dofs
# ---------------------------------------------------------------------------------------------------------------------
class HelmholtzPartition:
    """Host-side description of one rank's share of the fused Helmholtz operator.

    Elements are split into `world` contiguous ranges of the Morton order of their centroids (compact regions).  A rank
    builds an ordinary mesh / H1Space / FaceSpace of its own elements (so the local numbering is the reference's rule on
    the sub-mesh and every existing kernel applies unchanged); `l2g` maps its dofs to the global ones.  A dof touched by
    elements of several ranks is OWNED by the lowest of them; the others hold it as halo.  Vectors are stored per rank in
    local numbering, [u_loc; v_loc], and are ZERO at halo entries, so inner products are the sum of the local ones.

    Before an apply the owners send x at the dofs other ranks hold as halo; after it the halo holders send their partial
    sums of y back to the owners and clear them.  Both exchanges list dofs in increasing global id on both sides."""

    def __init__(self, cd, mesh, fem, fs, rank: int, world: int):
        import numpy as np

        self.rank, self.world = rank, world
        nb = fem.basis.n
        I = fem.global_indices()  # (nb, nb, n_elem), reference layout
        n_elem, ndof = mesh.n_elem(), fem.size()
        xy, elems = mesh.vertices(), mesh.elements().astype(np.int64)
        # compact, equally sized element sets: consecutive runs of the Morton order of the centroids (cuddh::partition_elements,
        # the rule the C++ host's HelmholtzPartition uses too)
        elem_rank = np.asarray(mesh.partition(world), dtype=np.int64)
        self.my_elems = np.flatnonzero(elem_rank == rank)
        if self.my_elems.size == 0:
            raise ValueError("HelmholtzPartition: more ranks than elements")
        If = I.reshape(nb * nb, n_elem, order="F")  # column e = dofs of element e
        owner = np.full(ndof, world, dtype=np.int64)
        np.minimum.at(owner, If.ravel(order="F"), np.repeat(elem_rank, nb * nb))
        self.ndof_global = ndof

        # ---- the sub-mesh and its spaces
        verts = np.unique(elems[self.my_elems])
        remap = np.full(len(xy), -1, dtype=np.int64)
        remap[verts] = np.arange(len(verts))
        self.mesh = cd.Mesh2D.from_vertices(xy[verts], remap[elems[self.my_elems]])
        self.fem = cd.H1Space(self.mesh, cd.Basis(nb))
        n_loc = self.fem.size()
        I_loc = self.fem.global_indices().reshape(nb * nb, len(self.my_elems), order="F")
        l2g = np.full(n_loc, -1, dtype=np.int64)
        l2g[I_loc.ravel(order="F")] = If[:, self.my_elems].ravel(order="F")
        if (l2g < 0).any() or not np.array_equal(l2g[I_loc], If[:, self.my_elems]) or np.unique(l2g).size != n_loc:
            raise RuntimeError("HelmholtzPartition: local and global numberings are inconsistent")
        self.l2g, self.n_loc = l2g, n_loc

        # ---- physical boundary faces of the sub-mesh (its other boundary edges are cuts between ranks)
        ge, le = mesh.edges(), self.mesh.edges()
        nv = len(xy)
        gb = ge[:, 0] == 1
        gkey_b = (np.minimum(ge[gb, 1], ge[gb, 2]).astype(np.int64) * nv + np.maximum(ge[gb, 1], ge[gb, 2]))
        lb = np.flatnonzero(le[:, 0] == 1)
        v0, v1 = verts[le[lb, 1]], verts[le[lb, 2]]
        lkey = np.minimum(v0, v1) * nv + np.maximum(v0, v1)
        self.faces = lb[np.isin(lkey, gkey_b)].astype(np.int32)
        self.fs = cd.FaceSpace(self.fem, self.faces)
        # face-space coefficient map: local face dof -> global face dof
        g2f = np.full(ndof, -1, dtype=np.int64)
        g2f[fs.global_indices()] = np.arange(fs.size())
        self.face_l2g = g2f[l2g[self.fs.global_indices()]] if self.fs.size() else np.zeros(0, dtype=np.int64)
        if (self.face_l2g < 0).any():
            raise RuntimeError("HelmholtzPartition: a local boundary-face dof is not in the global FaceSpace")

        # ---- ownership and the two exchanges
        own_of_local = owner[l2g]
        self.owned = np.flatnonzero(own_of_local == rank)
        self.halo = np.flatnonzero(own_of_local != rank)
        held_by = {}  # which of MY owned dofs other ranks hold
        for s in range(world):
            if s == rank:
                continue
            dofs_s = np.unique(If[:, elem_rank == s])
            mine = dofs_s[owner[dofs_s] == rank]  # sorted by global id
            if mine.size:
                held_by[s] = mine
        g2l = np.full(ndof, -1, dtype=np.int64)
        g2l[l2g] = np.arange(n_loc)
        # owners -> halo holders (x), halo holders -> owners (partial y): local ids, ordered by global id on both sides
        self.own_to = {s: g2l[g] for s, g in held_by.items()}
        self.halo_from = {}
        for s in range(world):
            if s == rank:
                continue
            h = self.halo[own_of_local[self.halo] == s]
            if h.size:
                self.halo_from[s] = h[np.argsort(l2g[h], kind="stable")]

    def both_components(self, ids):
        """entries of a local [u_loc; v_loc] vector for a set of local dofs"""
        import numpy as np

        return np.concatenate([ids, ids + self.n_loc])


class ShardedHelmholtz:
    """The fused complex Helmholtz operator (examples/Helmholtz.hpp:28-56) over `world` ranks, vectors partitioned by dof
    ownership (HelmholtzPartition).  `action(x, y)` works on local vectors of length 2 * n_loc that are zero at halo
    entries; `reduce` sums inner products over the ranks (pass it to cuddhelmholtz_amd.gmres(..., reduce=))."""

    def __init__(self, cd, omega, a2x, ax, mesh, fem, fs, rank: int = 0, world: int = 1, group=None, device="cuda",
                 host_staging: bool = False):
        import numpy as np
        import torch

        self.rank, self.world, self.group, self.host_staging = rank, world, group, host_staging
        self.device = torch.device(device)
        self.part = p = HelmholtzPartition(cd, mesh, fem, fs, rank, world)
        a2x, ax = np.asarray(a2x, dtype=np.float64), np.asarray(ax, dtype=np.float64)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.op = cd.HelmholtzOperator(omega, dev(a2x[p.l2g]), dev(ax[p.face_l2g]), p.fem, p.fs)
        self.n_loc = p.n_loc
        idx = lambda ids: torch.from_numpy(p.both_components(ids)).to(self.device)  # noqa: E731
        self.halo_idx = idx(p.halo)
        self.owned_idx = idx(p.owned)
        # One pack / unpack launch per exchange whatever the number of neighbours (cuddh_hip_halo_pack_f64 / _unpack_f64: the HIP
        # kernels the C++ host uses, include/cuddh_hip.h): concatenated local-dof lists, peers in increasing rank; a dof travels as
        # the pair (u, v), so a neighbour's message is a contiguous piece of the one buffer.
        def cat(d):
            ids = np.concatenate([d[s] for s in sorted(d)]) if d else np.zeros(0, dtype=np.int64)
            return torch.from_numpy(ids.astype(np.int32)).to(self.device), [(s, int(d[s].size)) for s in sorted(d)]

        self._own_ids, self._own_split = cat(p.own_to)
        self._halo_ids, self._halo_split = cat(p.halo_from)

        def views(ids, split):  # {peer: its piece of the concatenated list} (local dof ids)
            out, o = {}, 0
            for s, k in split:
                out[s] = ids[o:o + k]
                o += k
            return out

        self.own_to, self.halo_from = views(self._own_ids, self._own_split), views(self._halo_ids, self._halo_split)
        self._scratch = torch.zeros(2 * p.n_loc, dtype=torch.float64, device=self.device)

    # ---- the two exchanges, as pack / unpack pairs (an in-process replay of several ranks calls them directly)
    def _stream(self):
        import ctypes as C

        import torch

        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _pack(self, v, ids, split, clear: bool):
        """{peer: its piece of the packed buffer} of the entries `ids` of the local vector v (pairs (u, v)); clear: zero them in v"""
        import ctypes as C

        import torch

        from . import _native as N

        n = int(ids.numel())
        buf = torch.empty(2 * n, dtype=torch.float64, device=self.device)
        if n:
            N.check(N.lib.cuddh_hip_halo_pack_f64(n, self.n_loc, C.c_void_p(ids.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(buf.data_ptr()),
                                                   1 if clear else 0, self._stream()))
        out, o = {}, 0
        for s, k in split:
            out[s] = buf[2 * o:2 * (o + k)]
            o += k
        return out

    def _unpack(self, v, ids, split, received, add: bool) -> None:
        import ctypes as C

        from . import _native as N

        o = 0
        for s, k in split:  # one launch per sender, in rank order: partial sums reach a dof in a fixed order
            piece = received[s].to(self.device).contiguous()
            N.check(N.lib.cuddh_hip_halo_unpack_f64(k, self.n_loc, C.c_void_p(ids.data_ptr() + 4 * o), C.c_void_p(piece.data_ptr()),
                                                     C.c_void_p(v.data_ptr()), 1 if add else 0, self._stream()))
            o += k

    def pack_x(self, x):
        return self._pack(x, self._own_ids, self._own_split, False)

    def unpack_x(self, x, received) -> None:
        self._unpack(x, self._halo_ids, self._halo_split, received, False)

    def pack_y(self, y):
        return self._pack(y, self._halo_ids, self._halo_split, True)  # the halo entries of y are handed to their owners and cleared

    def unpack_y(self, y, received) -> None:
        self._unpack(y, self._own_ids, self._own_split, received, True)

    def _exchange(self, outgoing, incoming_split):
        import torch
        import torch.distributed as dist

        stage = (lambda t: t.cpu()) if self.host_staging else (lambda t: t)  # noqa: E731
        sbuf = {s: stage(t.contiguous()) for s, t in outgoing.items()}
        total = sum(n for _, n in incoming_split)
        rall = torch.empty(2 * total, dtype=torch.float64, device="cpu" if self.host_staging else self.device)
        rbuf, o = {}, 0
        for s, k in incoming_split:
            rbuf[s] = rall[2 * o:2 * (o + k)]
            o += k
        ops = []
        for s in sorted(set(sbuf) | set(rbuf)):
            if s in sbuf:
                ops.append(dist.P2POp(dist.isend, sbuf[s], s, self.group))
            if s in rbuf:
                ops.append(dist.P2POp(dist.irecv, rbuf[s], s, self.group))
        for r in (dist.batch_isend_irecv(ops) if ops else []):
            r.wait()
        return rbuf

    def action(self, x, y) -> None:
        """y = A x on the owned entries (halo entries of x are ignored and fetched from their owners; those of y end zero)"""
        xs = self._scratch
        xs.copy_(x)
        if self.world > 1:
            self.unpack_x(xs, self._exchange(self.pack_x(xs), self._halo_split))
        self.op.action(xs, y)
        if self.world > 1:
            self.unpack_y(y, self._exchange(self.pack_y(y), self._own_split))

    def reduce(self, t) -> None:
        if self.world > 1:
            import torch.distributed as dist

            if self.host_staging and t.device.type != "cpu":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.group)

    # ---- moving between the global numbering (on every rank) and this rank's partitioned vector
    def scatter(self, v_global):
        """global [u; v] (2 * ndof) -> local vector, zero at halo entries"""
        import torch

        p = self.part
        g = torch.from_numpy(p.l2g).to(v_global.device)
        nd = p.ndof_global
        z = torch.cat([v_global[:nd].index_select(0, g), v_global[nd:].index_select(0, g)]).to(self.device)
        z.index_fill_(0, self.halo_idx, 0.0)
        return z

    def gather(self, z):
        """local vector -> global [u; v] on every rank (sum over the ranks of the owned entries)"""
        import torch

        p = self.part
        nd = p.ndof_global
        out = torch.zeros(2 * nd, dtype=torch.float64, device=self.device)
        own = torch.from_numpy(p.owned).to(self.device)
        g = torch.from_numpy(p.l2g[p.owned]).to(self.device)
        out.index_copy_(0, g, z.index_select(0, own))
        out.index_copy_(0, g + nd, z.index_select(0, own + p.n_loc))
        self.reduce(out)
        return out
