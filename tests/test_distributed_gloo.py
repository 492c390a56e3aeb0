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
from cuddhelmholtz_amd.dist import NeighbourShardedDDH, ShardedDDH, TraceExchange, partition, rank_grid_map


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

    # what NeighbourShardedDDH asks of an engine
    def table(self, name):
        assert name == "B"
        return np.asarray(self.O.t.B, dtype=np.int32).ravel(order="F")

    def info(self):
        return {"mx_fdof": int(self.O.t.mx_fdof), "n_lambda": int(self.O.t.n_lambda)}


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


# ------------------------------------------------------------------ partitioned vectors + neighbour exchange
def build_case16():
    nx, nb = 16, 4
    omega = 2 * math.pi * nx / 10
    d = oracle.Discretization(oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), nb)
    h_a = d.nodal(oracle.alpha_disk)
    f = np.concatenate([oracle.linear_functional(d, oracle.gaussians(omega)), np.zeros(d.ndof)])
    O = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    return O, torch.from_numpy(f), d.ndof


def solve_partitioned(sh, O, f, ndof):
    n = O.size
    b = torch.zeros(n, dtype=torch.float64)
    sh.rhs(f, b)

    def A(x):
        y = torch.zeros(n, dtype=torch.float64)
        sh.action(torch.from_numpy(np.ascontiguousarray(x)), y)
        return y.numpy()

    def allreduce(v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        sh.reduce(t)
        return float(t[0])

    lam, info = oracle.gmres(A, b.numpy(), m=10, maxit=3, tol=1e-12, allreduce=allreduce)
    lam = torch.from_numpy(lam)
    u = torch.zeros(2 * ndof, dtype=torch.float64)
    sh.postprocess(lam, f, u)
    # every vector is zero outside the entries this rank owns
    mask = torch.ones(n, dtype=torch.bool)
    mask[sh.owned_idx] = False
    assert not bool(b[mask].any()) and not bool(lam[mask].any())
    return sh.full(b), sh.full(lam), u, info["num_matvec"]


def worker_neighbour(rank, world, port, out_dir, overlap, grid=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O, f, ndof = build_case16()
        dom_rank = rank_grid_map(4, 4, *grid) if grid else None  # 16^2 elements, n_basis 4: 4 x 4 blocks
        sh = NeighbourShardedDDH(OracleEngine(O), O.t.n_domains, rank, world, overlap=overlap, dom_rank=dom_rank)
        b, lam, u, nmv = solve_partitioned(sh, O, f, ndof)
        torch.save({"b": b, "lam": lam, "u": u, "nmv": nmv}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_trace_exchange_tables():
    """ownership is a partition of the touched slots; send and receive lists of two ranks mirror each other;
    boundary + interior ranges tile a rank's subdomains."""
    O, _, _ = build_case16()
    t = O.t
    B = np.asarray(t.B, dtype=np.int32).ravel(order="F")
    for world in (1, 2, 3, 5):
        ex = [TraceExchange(B, t.n_domains, t.mx_fdof, t.n_lambda, r, world) for r in range(world)]
        owned = np.concatenate([e.owned_slots for e in ex])
        assert np.unique(owned).size == owned.size
        touched = np.unique(np.asarray(t.B)[np.asarray(t.B) >= 0])
        assert np.array_equal(np.sort(owned), touched)
        assert t.n_lambda - touched.size == t.orphan_slots // 2
        for r in range(world):
            for s in range(world):
                if r != s:
                    a = ex[r].send_slots.get(s, np.zeros(0, dtype=np.int64))
                    b = ex[s].recv_slots.get(r, np.zeros(0, dtype=np.int64))
                    assert np.array_equal(a, b)
            doms = sorted(d for a, b in ex[r].boundary_ranges + ex[r].interior_ranges for d in range(a, b))
            assert doms == list(range(ex[r].d0, ex[r].d1)) == list(ex[r].domains)
            if world > 1:
                assert ex[r].boundary_ranges, "every rank of a connected block grid feeds a neighbour"
        if world == 2:  # 4x4 blocks split into two strips of two block rows: one block row feeds the other rank
            assert ex[0].boundary_ranges == [(4, 8)] and ex[1].boundary_ranges == [(8, 12)]
    # rank grids: rectangles of the block grid, at most four face neighbours, nothing between diagonal ranks
    for gx, gy in ((2, 2), (4, 1), (1, 4), (2, 1), (4, 2)):
        world = gx * gy
        dr = rank_grid_map(4, 4, gx, gy)
        assert np.array_equal(np.bincount(dr, minlength=world), np.full(world, 16 // world))
        ex = [TraceExchange(B, t.n_domains, t.mx_fdof, t.n_lambda, r, world, dr) for r in range(world)]
        owned = np.concatenate([e.owned_slots for e in ex])
        assert np.array_equal(np.sort(owned), touched)
        for r in range(world):
            rx, ry = r % gx, r // gx
            for s in range(world):
                if r != s:
                    a = ex[r].send_slots.get(s, np.zeros(0, dtype=np.int64))
                    assert np.array_equal(a, ex[s].recv_slots.get(r, np.zeros(0, dtype=np.int64)))
                    face_neighbour = abs(s % gx - rx) + abs(s // gx - ry) == 1
                    assert bool(a.size) == face_neighbour, (gx, gy, r, s)
            doms = sorted(d for a, b in ex[r].boundary_ranges + ex[r].interior_ranges for d in range(a, b))
            assert doms == list(ex[r].domains) == list(np.flatnonzero(dr == r))
            assert (ex[r].d0 is None) == (gx > 1 and 16 // world > 4 // gx)  # one range only when a rank's rectangle is one block row
    assert np.array_equal(rank_grid_map(4, 4, 1, 2), [0] * 8 + [1] * 8)  # gx = 1: the strips `partition` gives
    assert np.array_equal(rank_grid_map(5, 3, 2, 3).reshape(3, 5), [[0, 0, 0, 1, 1], [2, 2, 2, 3, 3], [4, 4, 4, 5, 5]])  # uneven split
    for bad in ((0, 1), (5, 1), (1, 5)):
        with pytest.raises(ValueError):
            rank_grid_map(4, 4, *bad)
    with pytest.raises(ValueError):
        TraceExchange(B, t.n_domains, t.mx_fdof, t.n_lambda, 0, 2, np.full(t.n_domains, 2))  # a rank outside [0, world)


@pytest.mark.parametrize("world,overlap,grid", [(2, False, None), (3, True, None), (4, True, (2, 2)), (2, False, (2, 1))])
def test_neighbour_exchange_reproduces_one_rank(tmp_path, world, overlap, grid):
    """grid: SURVEY 8e's gx x gy rectangles of the block grid instead of strips of block rows (a rank's subdomains are then
    not one range; cross points sit between four ranks)."""
    O, f, ndof = build_case16()
    single = solve(ShardedDDH(OracleEngine(O), O.t.n_domains), O, f, ndof)
    mp.spawn(worker_neighbour, args=(world, free_port(), str(tmp_path), overlap, grid), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, f"rank{r}.pt"), weights_only=True)
        assert torch.equal(got["b"], single[0])  # traces are copied, never summed: bitwise
        # inner products are summed rank by rank: same iteration to rounding
        assert got["nmv"] == single[3]
        assert torch.allclose(got["lam"], single[1], rtol=1e-10, atol=1e-13 * float(single[1].abs().max()))
        assert torch.allclose(got["u"], single[2], rtol=1e-9, atol=1e-12 * float(single[2].abs().max()))


def test_trace_exchange_on_random_slot_tables():
    """TraceExchange needs nothing but the slot table: random tables (random pairing of face dofs of random subdomains,
    random orphans and read-only / write-only slots, as the reference's cross-point quirk produces) must still give a
    partition of the touched slots and mirrored send / receive lists for every world size."""
    rng = np.random.default_rng(2024)
    for trial in range(20):
        n_dom, mx_fdof = int(rng.integers(2, 40)), int(rng.integers(1, 9))
        n_lambda = int(rng.integers(1, n_dom * mx_fdof + 1))
        B = -np.ones((mx_fdof, 2, n_dom), dtype=np.int32, order="F")
        free = [(i, s) for s in range(n_dom) for i in range(mx_fdof)]
        rng.shuffle(free)
        for col in (0, 1):  # every slot gets at most one reader and at most one writer, some get none
            cells = list(free)
            rng.shuffle(cells)
            slots = rng.permutation(n_lambda)[: int(rng.integers(0, min(n_lambda, len(cells)) + 1))]
            for t, (i, s) in zip(slots, cells):
                B[i, col, s] = t
        touched = np.unique(B[B >= 0])
        for world in (1, 2, 3, 7):
            ex = [TraceExchange(B.ravel(order="F"), n_dom, mx_fdof, n_lambda, r, world) for r in range(world)]
            owned = np.concatenate([e.owned_slots for e in ex])
            assert np.array_equal(np.sort(owned), touched)
            for r in range(world):
                for s in range(world):
                    if r != s:
                        assert np.array_equal(ex[r].send_slots.get(s, np.zeros(0, dtype=np.int64)),
                                              ex[s].recv_slots.get(r, np.zeros(0, dtype=np.int64)))
                doms = sorted(d for a, b in ex[r].boundary_ranges + ex[r].interior_ranges for d in range(a, b))
                assert doms == list(range(ex[r].d0, ex[r].d1))
                # a slot is sent exactly when its writer is mine and its owner is not
                sent = np.concatenate(list(ex[r].send_slots.values())) if ex[r].send_slots else np.zeros(0, dtype=np.int64)
                assert not np.intersect1d(sent, ex[r].owned_slots).size


# ------------------------------------------------------------------ partitioned global Helmholtz operator (host logic)
@pytest.mark.parametrize("kind,nb", [("structured", 3), ("structured", 4), ("unstructured", 3)])
def test_helmholtz_partition_replay(kind, nb, unstructured_square):
    """HelmholtzPartition (element partition, local sub-mesh numbering, ownership, the two halo exchanges) replayed in one
    process for 1..5 ranks with the oracle as the local operator: the assembled result must be the oracle's global apply."""
    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import HelmholtzPartition

    if kind == "structured":
        nx = 7
        pm, om = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    else:
        xy, elems = unstructured_square
        pm, om = cd.Mesh2D.from_vertices(xy, elems), oracle.Mesh(xy, elems)
    fem = cd.H1Space(pm, cd.Basis(nb))
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    d = oracle.Discretization(om, nb)
    ofs = oracle.FaceSpaceO(d, list(faces))
    ndof = d.ndof
    rng = np.random.default_rng(5)
    a2, ax = 0.5 + rng.random(ndof), 0.5 + rng.random(ofs.size)
    xg = rng.standard_normal(2 * ndof)
    omega = 4.0
    ref = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xg)

    for world in (1, 2, 3, 5):
        parts = [HelmholtzPartition(cd, pm, fem, fs, r, world) for r in range(world)]
        # ownership is a partition of the dofs; every element belongs to exactly one rank
        owned_g = np.concatenate([p.l2g[p.owned] for p in parts])
        assert np.array_equal(np.sort(owned_g), np.arange(ndof))
        assert np.array_equal(np.sort(np.concatenate([p.my_elems for p in parts])), np.arange(pm.n_elem()))
        assert sum(len(p.faces) for p in parts) == len(faces)
        # local operators (oracle) on the sub-meshes, in the product's local numbering
        local = []
        for p in parts:
            lm = oracle.Mesh(p.mesh.vertices(), p.mesh.elements())
            ld = oracle.Discretization(lm, nb)
            assert np.array_equal(ld.I, p.fem.global_indices())
            lofs = oracle.FaceSpaceO(ld, list(p.faces))
            assert np.array_equal(lofs.proj, p.fs.global_indices())
            local.append((ld, lofs, a2[p.l2g], ax[p.face_l2g]))
        # partitioned vectors: zero at halo entries
        xs = []
        for p in parts:
            z = np.concatenate([xg[:ndof][p.l2g], xg[ndof:][p.l2g]])
            z[p.both_components(p.halo)] = 0.0
            xs.append(z)
        # exchange 1: owners -> halo holders
        for p in parts:
            for s, ids in p.own_to.items():
                dst = parts[s].halo_from[p.rank]
                assert np.array_equal(p.l2g[ids], parts[s].l2g[dst])  # same dofs, same order, on both sides
                xs[s][parts[s].both_components(dst)] = xs[p.rank][p.both_components(ids)]
        ys = []
        for p, (ld, lofs, a2l, axl), z in zip(parts, local, xs):
            assert np.allclose(z[: p.n_loc], xg[:ndof][p.l2g]) and np.allclose(z[p.n_loc:], xg[ndof:][p.l2g])
            ys.append(oracle.helmholtz_apply(ld, oracle.Stiffness(ld), oracle.Mass(ld, a2l), oracle.FaceMass(lofs, axl), lofs, omega, z))
        # exchange 2: halo holders -> owners (partial sums), halo cleared
        incoming = [dict() for _ in parts]
        for p in parts:
            for s, ids in p.halo_from.items():
                incoming[s][p.rank] = ys[p.rank][p.both_components(ids)]
        for p in parts:
            for s, ids in p.own_to.items():
                ys[p.rank][p.both_components(ids)] += incoming[p.rank][s]
            ys[p.rank][p.both_components(p.halo)] = 0.0
        out = np.zeros(2 * ndof)
        for p, y in zip(parts, ys):
            out[p.l2g[p.owned]] = y[p.owned]
            out[ndof + p.l2g[p.owned]] = y[p.n_loc + p.owned]
        assert np.linalg.norm(out - ref) <= 1e-13 * np.linalg.norm(ref)


@pytest.mark.parametrize("nx,nb,world", [(16, 4, 2), (16, 4, 3), (32, 4, 8), (8, 8, 5), (10, 3, 4)])
def test_native_trace_exchange_plan_equals_python(nx, nb, world):
    """cuddh::TraceExchangePlan (the C++ multi-GPU host, csrc/src/multigpu.cpp) and dist.TraceExchange (the Python host)
    derive ownership, send and receive lists from the same slot table: they must be identical, list by list."""
    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import TraceExchange, native_trace_exchange

    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(2 * math.pi * nx / 10, np.ones(fem.size()), fem, nx, nx)
    info = F.info()
    B = F.table("B")
    for rank in range(world):
        ex = TraceExchange(B, info["n_domains"], info["mx_fdof"], info["n_lambda"], rank, world)
        owned, send, recv = native_trace_exchange(B, info["n_domains"], info["mx_fdof"], info["n_lambda"], rank, world)
        assert np.array_equal(owned, ex.owned_slots)
        assert sorted(send) == sorted(ex.send_slots) and sorted(recv) == sorted(ex.recv_slots)
        for p in send:
            assert np.array_equal(send[p], ex.send_slots[p])
        for p in recv:
            assert np.array_equal(recv[p], ex.recv_slots[p])
        # the two launches of the split schedule: the same subdomains in both hosts (boundary padded to whole workgroups)
        from cuddhelmholtz_amd.dist import NeighbourShardedDDH

        class _Engine:  # what NeighbourShardedDDH asks of an engine at construction
            def table(self, name):
                return B

            def info(self):
                return info

        sh = NeighbourShardedDDH(_Engine(), info["n_domains"], rank, world)
        boundary, interior = native_trace_exchange.last_split
        assert np.array_equal(boundary, sh._ids["boundary"]) and np.array_equal(interior, sh._ids["interior"])
        assert boundary.size % 8 == 0 or interior.size == 0 or world == 1


@pytest.mark.parametrize("kind,nb", [("structured", 4), ("unstructured", 3), ("refined", 5)])
def test_native_helmholtz_partition_equals_python(kind, nb, unstructured_square):
    """cuddh::HelmholtzPartition (csrc/src/partition.cpp, what the C++ multi-device host of the global operator apply uses) against
    dist.HelmholtzPartition (validated against the oracle above): same elements, same local -> global map, same ownership, the
    same dofs in the same order in both exchanges, the same boundary faces -- for 1..5 ranks."""
    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import HelmholtzPartition, native_helmholtz_partition

    if kind == "structured":
        pm = cd.Mesh2D.uniform_rect(9, -1.0, 1.0, 6, -1.0, 1.0)
    else:
        xy, elems = unstructured_square
        pm = cd.Mesh2D.from_vertices(xy, elems)
        if kind == "refined":
            pm = pm.refined(1)
    fem = cd.H1Space(pm, cd.Basis(nb))
    fs = cd.FaceSpace(fem, pm.boundary_edges())
    for world in (1, 2, 3, 5):
        for r in range(world):
            py = HelmholtzPartition(cd, pm, fem, fs, r, world)
            cx = native_helmholtz_partition(pm, fem, fs, r, world)
            assert np.array_equal(cx["my_elems"], py.my_elems)
            assert np.array_equal(cx["l2g"], py.l2g)
            assert np.array_equal(cx["owned"], py.owned) and np.array_equal(cx["halo"], py.halo)
            assert np.array_equal(cx["faces"], py.faces) and np.array_equal(cx["face_l2g"], py.face_l2g)
            assert sorted(cx["own_to"]) == sorted(py.own_to) and sorted(cx["halo_from"]) == sorted(py.halo_from)
            for s in py.own_to:
                assert np.array_equal(cx["own_to"][s], py.own_to[s])
            for s in py.halo_from:
                assert np.array_equal(cx["halo_from"][s], py.halo_from[s])
