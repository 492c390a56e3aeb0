"""Size-independent properties at BASELINE.json's full sizes (the oracle cannot reach them in seconds):
1024 x 1024 elements, n_basis 4 (9.4 M dofs, 65,536 subdomains) and 256 x 256 (config 2)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(cuda):
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb = 1024, 4
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    assert fem.size() == (3 * nx + 1) ** 2 == 9443329
    return dict(cd=cd, torch=torch, nx=nx, nb=nb, mesh=mesh, fem=fem, n=fem.size())


def test_config2_properties_256(cuda):
    """BASELINE config 2 (Helmholtz omega = 8 pi, 256 x 256 quads, GMRES(20) on the Stiffness/Mass/FaceMass composite): the
    invariants of the full-size tests on config 2's own mesh for n_basis 4 (Basis(p) reading of p = 3 + 1) and
    n_basis 3, and a GMRES(20) cycle on the fused operator that must reduce the residual exactly like the unfused one."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, omega = 256, 8 * math.pi
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    for nb, ndof in ((4, 591361), (3, 263169)):
        fem = cd.H1Space(mesh, cd.Basis(nb))
        n = fem.size()
        assert n == ndof  # SURVEY 8d sizes
        fs = cd.FaceSpace(fem, mesh.boundary_edges())
        g = torch.Generator(device="cpu").manual_seed(256 + nb)
        a2 = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(cuda)
        ax = (0.5 + torch.rand(fs.size(), generator=g, dtype=torch.float64)).to(cuda)
        A = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
        assert A.fused()
        if nb == 4:
            assert abs(A.bytes_per_apply() - 96.0e6) < 0.2e6  # 96.0 MB per complex apply
        x = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
        z = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
        Ax, Az, Axz, Ax2, U = (torch.empty_like(x) for _ in range(5))
        A.action(x, Ax)
        A.action(z, Az)
        A.action(x, Ax2)
        assert torch.equal(Ax, Ax2)
        A.action(0.25 * x + 3.0 * z, Axz)
        lin = 0.25 * Ax + 3.0 * Az
        assert float(torch.linalg.norm(Axz - lin) / torch.linalg.norm(lin)) < 1e-13
        s1, s2 = float(torch.dot(z, Ax)), float(torch.dot(x, Az))
        assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
        A.action_unfused(x, U)
        assert float(torch.linalg.norm(U - Ax) / torch.linalg.norm(Ax)) < 1e-13
        # two GMRES(20) cycles: fused operator vs the composite of the separate operators through the callback path
        b = Ax
        xs1, xs2 = torch.zeros_like(b), torch.zeros_like(b)
        o1 = cd.gmres(2 * n, xs1, A, b, 20, 3, 1e-12)
        o2 = cd.gmres(2 * n, xs2, lambda p, q: A.action_unfused(p, q), b, 20, 3, 1e-12)
        assert o1.num_matvec == o2.num_matvec == 43
        assert o1.res_norm[-1] < o1.res_norm[0]
        assert abs(o1.res_norm[-1] - o2.res_norm[-1]) <= 1e-9 * o1.res_norm[0]
        assert float(torch.linalg.norm(xs1 - xs2) / torch.linalg.norm(xs1)) < 1e-9


@pytest.mark.parametrize("nb,affine", [(2, "0"), (3, "0"), (2, "1")])
def test_lane_form_chosen_by_size_nb2_nb3(cuda, nb, affine, monkeypatch):
    """768 x 768 elements (589,824 >= the 8192 x 64 of the size rule): the plan must pick helm_lane_kernel with non-temporal
    metric loads by itself for n_basis 2 and 3 (general layout) and for the affine n_basis-2 form; linearity, symmetry,
    determinism, and agreement with the 32-element-patch kernel it replaces (CUDDH_HELM_LANE=0)."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, omega = 768, 24 * math.pi
    monkeypatch.setenv("CUDDH_PLAN_AFFINE", affine)
    for k in ("CUDDH_HELM_LANE", "CUDDH_HELM_PE", "CUDDH_PLAN_STREAMING"):
        monkeypatch.delenv(k, raising=False)
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    n = fem.size()
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    g = torch.Generator(device="cpu").manual_seed(768 + nb)
    a2 = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(cuda)
    ax = (0.5 + torch.rand(fs.size(), generator=g, dtype=torch.float64)).to(cuda)
    A = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    nqS, nqM = nb + 1, 2 + 3 * nb // 2
    # affine n_basis 2 moves ~100 MB per apply (fits the infinity cache: default load policy), the general layouts stream
    assert A.kernel() in (f"helm_lane_kernel<{nb},{nqS},{nqM},NT=1,UG={affine}> pe=64", f"helm_lane_kernel<{nb},{nqS},{nqM},NT=0,UG={affine}> pe=64"), A.kernel()
    if affine == "0":
        assert "NT=1" in A.kernel()
    x = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
    z = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
    Ax, Az, Axz, Ax2 = (torch.empty_like(x) for _ in range(4))
    A.action(x, Ax)
    A.action(z, Az)
    A.action(x, Ax2)
    assert torch.equal(Ax, Ax2)
    A.action(0.5 * x - 2.0 * z, Axz)
    lin = 0.5 * Ax - 2.0 * Az
    assert float(torch.linalg.norm(Axz - lin) / torch.linalg.norm(lin)) < 1e-13
    s1, s2 = float(torch.dot(z, Ax)), float(torch.dot(x, Az))
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    monkeypatch.setenv("CUDDH_HELM_LANE", "0")
    P = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    assert P.kernel().startswith(f"helm_patch_kernel<{nb},"), P.kernel()
    P.action(x, Ax2)
    assert float(torch.linalg.norm(Ax2 - Ax) / torch.linalg.norm(Ax)) < 1e-13
    A.action_unfused(x, Ax2)
    assert float(torch.linalg.norm(Ax2 - Ax) / torch.linalg.norm(Ax)) < 1e-13


def test_operator_invariants_at_full_size(cuda, big):
    cd, torch, fem, n = big["cd"], big["torch"], big["fem"], big["n"]
    one = torch.ones(n, dtype=torch.float64, device=cuda)
    y = torch.empty_like(one)
    cd.StiffnessMatrix(fem).action(one, y)  # constants are in the kernel of the stiffness operator
    assert float(y.abs().max()) < 1e-11
    cd.MassMatrix(fem).action(one, y)  # 1^T M 1 = area of [-1,1]^2
    assert abs(float(y.sum()) - 4.0) < 1e-10
    g = torch.Generator(device="cpu").manual_seed(3)
    coef = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(cuda)
    Mw = cd.MassMatrix(fem, coef)
    x1 = torch.rand(n, generator=g, dtype=torch.float64).to(cuda)
    x2 = torch.rand(n, generator=g, dtype=torch.float64).to(cuda)
    y1, y2 = torch.empty_like(x1), torch.empty_like(x1)
    Mw.action(x1, y1)
    Mw.action(x2, y2)
    # symmetry: x2^T M x1 == x1^T M x2
    a, b = float(torch.dot(x2, y1)), float(torch.dot(x1, y2))
    assert abs(a - b) <= 1e-12 * abs(a)
    # the patch-plan kernels have no atomics: bitwise reproducible; action(c, x, y) accumulates onto what action(x, y) wrote
    S = cd.StiffnessMatrix(fem)
    S.action(x1, y1)
    S.action(x1, y2)
    assert torch.equal(y1, y2)
    scale = float(y1.abs().max())
    S.action(-1.0, x1, y2)
    assert float(y2.abs().max()) <= 1e-13 * scale
    Mw.action(x1, y2)
    Mw.action(2.0, x1, y1)  # y1 = S x1 + 2 M x1
    S.action(-1.0, x1, y1)
    assert float((y1 - 2.0 * y2).abs().max()) <= 1e-12 * scale


def test_fused_helmholtz_linearity_symmetry_determinism_at_full_size(cuda, big, monkeypatch):
    cd, torch, mesh, fem, n, nx = big["cd"], big["torch"], big["mesh"], big["fem"], big["n"], big["nx"]
    monkeypatch.setenv("CUDDH_PLAN_AFFINE", "0")  # the general-geometry layout SURVEY 8d's bytes are stated on
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    assert fs.size() == 4 * 3 * nx
    omega = 32 * math.pi
    a2 = torch.ones(n, dtype=torch.float64, device=cuda)
    ax = torch.ones(fs.size(), dtype=torch.float64, device=cuda)
    A = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    assert A.fused()
    assert A.kernel() == "helm_lane_kernel<4,5,8,NT=1,UG=0> pe=64"  # what bench.py's roofline figure runs
    assert A.bytes_per_apply() == 1535639584  # 1,535 MB at 1024^2, n_basis 4
    g = torch.Generator(device="cpu").manual_seed(12345)
    x = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
    z = (2 * torch.rand(2 * n, generator=g, dtype=torch.float64) - 1).to(cuda)
    Ax, Az, Axz, Ax2 = (torch.empty_like(x) for _ in range(4))
    A.action(x, Ax)
    A.action(z, Az)
    A.action(0.5 * x - 2.0 * z, Axz)
    A.action(x, Ax2)
    assert torch.equal(Ax, Ax2)  # no atomics anywhere in the fused path
    lin = 0.5 * Ax - 2.0 * Az
    assert float(torch.linalg.norm(Axz - lin) / torch.linalg.norm(lin)) < 1e-13
    # "scal(-1, Av) makes the system symmetric" (examples/Helmholtz.hpp:55)
    s1, s2 = float(torch.dot(z, Ax)), float(torch.dot(x, Az))
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    U = torch.empty_like(x)
    A.action_unfused(x, U)  # the composite of the separate operators (examples/Helmholtz.hpp:28-56)
    assert float(torch.linalg.norm(U - Ax) / torch.linalg.norm(Ax)) < 1e-13
    # the affine form of the same operator (uniform mesh: one copy of the stiffness metric instead of 1,048,576)
    # moves 0.91 GB instead of 1.54 GB and agrees to rounding
    monkeypatch.setenv("CUDDH_PLAN_AFFINE", "1")
    B = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    assert B.fused() and B.bytes_per_apply() == 1535639584
    n_elem = nx * nx
    assert B.bytes_affine() == 1535639584 - n_elem * 3 * 25 * 8
    Bx = torch.empty_like(x)
    B.action(x, Bx)
    assert float(torch.linalg.norm(Bx - Ax) / torch.linalg.norm(Ax)) < 1e-13
    # at this size the general-geometry plan above ran helm_lane_kernel (64-element patches, both components per lane);
    # the 32-element-patch kernel it replaced there must give the same vector to rounding
    monkeypatch.setenv("CUDDH_PLAN_AFFINE", "0")
    monkeypatch.setenv("CUDDH_HELM_LANE", "0")
    P = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    P.action(x, Bx)
    assert float(torch.linalg.norm(Bx - Ax) / torch.linalg.norm(Ax)) < 1e-13
    # and the lane form with the whole patch's metric block requested up front (one wavefront per SIMD; same arithmetic in
    # the same order: bitwise the same vector)
    monkeypatch.delenv("CUDDH_HELM_LANE")
    monkeypatch.setenv("CUDDH_HELM_PRE", "1")
    Q = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    assert Q.kernel() == "helm_lane_kernel<4,5,8,NT=1,UG=0,PRE=1> pe=64"
    Q.action(x, Bx)
    assert torch.equal(Bx, Ax)
    # the same plan on vectors in its OWN ordering (pairs, a patch's owned dofs contiguous: what bench.py's roofline figure runs
    # and HelmholtzOperator::gmres iterates on): a permutation in, the same numbers bit for bit out, fewer bytes as laid out
    monkeypatch.delenv("CUDDH_HELM_PRE")
    assert A.has_native()
    zx, zy = torch.empty_like(x), torch.empty_like(x)
    A.to_native(x, zx)
    A.action_native(zx, zy)
    A.from_native(zy, Bx)
    assert torch.equal(Bx, Ax)
    assert A.bytes_native() < A.bytes_per_apply(True) and A.bytes_native() < 1.04 * A.bytes_per_apply()


def test_ddh_properties_at_full_size(cuda, big):
    cd, torch, fem, n, nx = big["cd"], big["torch"], big["fem"], big["n"], big["nx"]
    from cuddhelmholtz_amd.dist import ShardedDDH

    omega = 32 * math.pi
    F = cd.DDH(omega, np.ones(n), fem, nx, nx)
    info = F.info()
    assert (info["n_domains"], info["nt"], info["kernel"]) == (65536, 5120, 5)
    assert F.size() == 4 * 1697280  # 25,792 shared dofs at 128^2 scale as (edges); checked against the closed form below
    n_edges_between_subdomains = 2 * 256 * 255  # 256 x 256 subdomains
    assert F.size() // 4 == n_edges_between_subdomains * 13
    g = torch.Generator(device="cpu").manual_seed(7)
    lam = (2 * torch.rand(F.size(), generator=g) - 1).to(cuda)
    y1, y2, y3 = (torch.zeros_like(lam) for _ in range(3))
    F.action(lam, y1)
    F.action(lam, y2)
    assert torch.equal(y1, y2)  # bitwise reproducible (fixed summation order)
    F.action(2.0 * lam, y3)     # I - T is linear; scaling by 2 is exact in fp32
    assert torch.equal(y3, 2.0 * y1)
    # sharded in three unequal ranges == whole
    upd_whole = torch.zeros_like(lam)
    F.local_traces(0, 65536, None, lam, upd_whole)
    upd_parts = torch.zeros_like(lam)
    for d0, d1 in ((0, 10000), (10000, 40001), (40001, 65536)):
        F.local_traces(d0, d1, None, lam, upd_parts)
    assert torch.equal(upd_whole, upd_parts)
    sh = ShardedDDH(F, 65536)
    y4 = torch.zeros_like(lam)
    sh.action(lam, y4)
    assert torch.equal(y4, y1)
    assert bool(torch.isfinite(y1).all())

    # the 8-rank neighbour exchange replayed in one process: every rank sees only the entries it owns, solves its
    # range, and the messages are delivered by hand; the assembled traces must be bitwise the single-rank ones
    from cuddhelmholtz_amd.dist import NeighbourShardedDDH

    world = 8
    ranks = [NeighbourShardedDDH(F, 65536, r, world, device=cuda, dry_run=True) for r in range(world)]
    owned = torch.cat([r.owned_idx for r in ranks])
    assert owned.numel() == torch.unique(owned).numel()  # ownership is a partition ...
    assert not bool(lam.new_ones(lam.numel()).index_fill_(0, owned, 0).mul(upd_whole).any())  # ... of everything that is written
    outs = []
    for r in ranks:
        lam_r = torch.zeros_like(lam)
        lam_r[r.owned_idx] = lam[r.owned_idx]
        out_r = torch.zeros_like(lam)
        F.local_traces(r.d0, r.d1, None, lam_r, out_r)
        outs.append(out_r)
        assert len(r.send_idx) == (1 if r.rank in (0, world - 1) else 2)  # strips of block rows: at most two neighbours
        # per neighbour: 256 subdomain edges x 13 dofs, minus 2 slots lost to the cross-point quirk at each of the 255
        # interior cross points, times (lambda, mu): 5636 floats = 22 KiB
        assert all(i.numel() == 2 * (256 * 13 - 2 * 255) for i in r.send_idx.values())
    for r in ranks:
        for s_rank, idx in r.send_idx.items():
            dst = ranks[s_rank].recv_idx[r.rank]
            assert torch.equal(dst, idx)
            outs[s_rank][dst] = outs[r.rank][idx]
    for r in ranks:
        for idx in r.send_idx.values():
            outs[r.rank][idx] = 0
    for r, out_r in zip(ranks, outs):
        mask = torch.ones(lam.numel(), dtype=torch.bool, device=cuda)
        mask[r.owned_idx] = False
        assert not bool(out_r[mask].any())  # zero outside the owned entries
    assert torch.equal(torch.stack(outs).sum(0), upd_whole)

    # the same with a 2 x 4 grid of ranks (SURVEY 8e's rectangles: 128 x 64 subdomains each, a rank's subdomains are 64 runs of
    # 128): local solves through the listed one-launch entry point, at most 3 face neighbours here, nothing between diagonals
    from cuddhelmholtz_amd.dist import rank_grid_map

    dom_rank = rank_grid_map(256, 256, 2, 4)
    grid = [NeighbourShardedDDH(F, 65536, r, world, device=cuda, dry_run=True, dom_rank=dom_rank) for r in range(world)]
    outs = []
    for r in grid:
        assert r.ex.d0 is None and r.ex.domains.size == 8192
        lam_r = torch.zeros_like(lam)
        lam_r[r.owned_idx] = lam[r.owned_idx]
        out_r = torch.zeros_like(lam)
        r._solve("all", None, lam_r, out_r)
        outs.append(out_r)
        rx, ry = r.rank % 2, r.rank // 2
        assert sorted(r.send_idx) == sorted(s for s in range(world) if abs(s % 2 - rx) + abs(s // 2 - ry) == 1)
        assert sum(i.numel() for i in r.send_idx.values()) < 2 * 2 * (256 * 13)  # less than a strip sends
    for r in grid:
        for s_rank, idx in r.send_idx.items():
            assert torch.equal(grid[s_rank].recv_idx[r.rank], idx)
            outs[s_rank][idx] = outs[r.rank][idx]
    for r in grid:
        for idx in r.send_idx.values():
            outs[r.rank][idx] = 0
    assert torch.equal(torch.stack(outs).sum(0), upd_whole)
    owned = torch.cat([r.owned_idx for r in grid])
    assert owned.numel() == torch.unique(owned).numel()


@pytest.mark.parametrize("coef", ["one", "disk"])
def test_config3_ddh_properties_at_full_size(cuda, coef):
    """BASELINE config 3 at its own size and frequency: 512 x 512 quads, omega = 16 pi (32 elements per wavelength, nt = 5120),
    n_basis 4, 16,384 reference-size subdomains, with a = 1 and with the example's disk coefficient (examples/DDH.cpp:74-83,
    lumped-projected as in :122-123).  Size-independent properties of the benchmarked fp32 kernel: bitwise repeatability,
    exact homogeneity under scaling by 2, ranges == whole, the listed one-launch form == whole, finiteness; and one Arnoldi
    cycle of the reference's flow (rhs -> gmres(20), one cycle) returning a finite, reduced residual.  (Entry-point parity
    against the oracle in this regime: tests/test_baseline_regime.py.)"""
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb, omega = 512, 4, 16 * math.pi
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    n = fem.size()
    assert n == 2362369
    a = torch.ones(n, dtype=torch.float64, device=cuda)
    if coef == "disk":
        cd.linear_functional(fem, cd.ALPHA_DISK, a)
        cd.DiagInvMassMatrix(fem).action(a, a)
    F = cd.DDH(omega, a.cpu().numpy(), fem, nx, nx)
    info = F.info()
    assert (info["n_domains"], info["nt"], info["kernel"]) == (16384, 5120, 5)
    assert F.size() // 4 == 2 * 128 * 127 * 13
    g = torch.Generator(device="cpu").manual_seed(3)
    lam = (2 * torch.rand(F.size(), generator=g) - 1).to(cuda)
    y1, y2, y3 = (torch.zeros_like(lam) for _ in range(3))
    F.action(lam, y1)
    F.action(lam, y2)
    assert torch.equal(y1, y2)
    F.action(2.0 * lam, y3)
    assert torch.equal(y3, 2.0 * y1)
    assert bool(torch.isfinite(y1).all())
    whole, parts, listed = (torch.zeros_like(lam) for _ in range(3))
    F.local_traces(0, 16384, None, lam, whole)
    for d0, d1 in ((0, 4097), (4097, 9000), (9000, 16384)):
        F.local_traces(d0, d1, None, lam, parts)
    assert torch.equal(whole, parts)
    perm = torch.randperm(16384, generator=g).to(torch.int32).to(cuda)
    F.local_traces_listed(perm[:5001], None, lam, listed)
    F.local_traces_listed(perm[5001:], None, lam, listed)
    assert torch.equal(whole, listed)
    growth = float(whole.double().norm() / lam.double().norm())
    print(f"config 3 (512^2, omega=16pi, a={coef}): |T lambda|/|lambda| = {growth:.3e} for random traces")
    if coef == "one":
        assert growth < 1.5  # non-expansive local solves
    else:
        assert growth > 1e3  # the reference's time stepping is unstable here (tests/test_baseline_regime.py)
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
    b = torch.zeros_like(lam)
    F.rhs(f, b)
    assert bool(torch.isfinite(b).all()) and float(b.norm()) > 0
    x = torch.zeros_like(b)
    out = cd.gmres(F.size(), x, F, b, 20, 2, 1e-4)  # one cycle (the reference's loop runs maxit - 1 cycles)
    assert out.num_matvec == 22 and all(math.isfinite(r) for r in out.res_norm)
    print(f"  one GMRES(20) cycle: relative residual {out.res_norm[-1] / out.res_norm[0]:.3e}")
    assert out.res_norm[-1] < out.res_norm[0]
