"""GPU parity: every hot-path kernel, driven through the C ABI, against the CPU oracle
on the same seeded inputs.  Tolerances (relative l2 unless noted):
  fp64 operator applies          1e-12  (summation order only)
  DDH fp64 mode                  1e-10
  DDH fp32 mode vs fp64 oracle   2e-4   (reference precision; reported, loosely gated)
"""
import math

import numpy as np
import pytest

import oracle
from conftest import load_unstructured_square

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))


def to_dev(torch, a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def meshes(kind, nx=10):
    import cuddhelmholtz_amd as cd

    if kind == "structured":
        return cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    xy, elems = load_unstructured_square()
    return cd.Mesh2D.from_vertices(xy, elems), oracle.Mesh(xy, elems)


# ------------------------------------------------------------------ BLAS-1 (reference tests/linalg.cpp)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n", [0, 1, 5, 1024, 100003])
def test_blas1(cuda, dtype, n):
    import ctypes as C

    import torch

    from cuddhelmholtz_amd import _native as N

    lib = N.lib
    tt = torch.float64 if dtype == "f64" else torch.float32
    npt = np.float64 if dtype == "f64" else np.float32
    rng = np.random.default_rng(n)
    xh, yh = rng.standard_normal(n).astype(npt), rng.standard_normal(n).astype(npt)
    # +1 element offset exercises the unaligned (scalar) path as well
    for off in (0, 1):
        xb = torch.zeros(n + 1, dtype=tt, device=cuda)
        yb = torch.zeros(n + 1, dtype=tt, device=cuda)
        x, y = xb[off:off + n], yb[off:off + n]
        x.copy_(torch.from_numpy(xh))
        y.copy_(torch.from_numpy(yh))
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws = torch.zeros(lib.cuddh_hip_reduce_ws_bytes() // 8, dtype=torch.float64, device=cuda)
        res = torch.zeros(2, dtype=tt, device=cuda)
        a, b = 0.75, -1.25
        N.check(getattr(lib, f"cuddh_hip_axpby_{dtype}")(n, a, p(x), b, p(y), st))
        exp = a * xh.astype(np.float64) + b * yh.astype(np.float64)
        eps = 4 * np.finfo(npt).eps
        assert np.all(np.abs(y.cpu().numpy() - exp) <= eps * (abs(a) * np.abs(xh) + abs(b) * np.abs(yh)))
        yh2 = y.cpu().numpy()
        N.check(getattr(lib, f"cuddh_hip_dot_{dtype}")(n, p(x), p(y), p(res), p(ws), st))
        N.check(getattr(lib, f"cuddh_hip_nrm2_{dtype}")(n, p(x), p(res[1:]), p(ws), st))
        r = res.cpu().numpy()
        tol = 1e-12 if dtype == "f64" else 2e-5
        scale = float(np.sum(np.abs(xh.astype(np.float64) * yh2))) + 1e-300
        assert abs(float(r[0]) - float(np.dot(xh.astype(np.float64), yh2.astype(np.float64)))) <= tol * scale
        assert abs(float(r[1]) - float(np.linalg.norm(xh.astype(np.float64)))) <= tol * (float(np.linalg.norm(xh)) + 1e-300)
        N.check(getattr(lib, f"cuddh_hip_sqdist_{dtype}")(n, p(x), p(y), p(res), p(ws), st))
        d = float(np.sum((xh.astype(np.float64) - yh2) ** 2))
        assert abs(float(res.cpu()[0]) - d) <= tol * (d + 1e-300)
        N.check(getattr(lib, f"cuddh_hip_scal_{dtype}")(n, 3.0, p(y), st))
        assert np.array_equal(y.cpu().numpy(), yh2 * npt(3.0))
        N.check(getattr(lib, f"cuddh_hip_copy_{dtype}")(n, p(x), p(y), st))
        assert np.array_equal(y.cpu().numpy(), xh)
        N.check(getattr(lib, f"cuddh_hip_fill_{dtype}")(n, 2.5, p(y), st))
        assert np.all(y.cpu().numpy() == npt(2.5))
        # guard elements untouched
        assert float(yb[n - off if off == 0 else 0]) == 0.0


def test_python_boundary_rejects_wrong_dtype_and_short_tensors(cuda):
    """A float32 or too-short tensor handed to an fp64 entry point must be a Python error, not an out-of-bounds device access
    behind the C ABI."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb = 8, 4
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    n = fem.size()
    S = cd.StiffnessMatrix(fem)
    x64, y64 = torch.zeros(n, dtype=torch.float64, device=cuda), torch.zeros(n, dtype=torch.float64, device=cuda)
    S.action(x64, y64)
    with pytest.raises(ValueError, match="dtype"):
        S.action(x64.float(), y64)
    with pytest.raises(ValueError, match="at least"):
        S.action(x64[: n - 1], y64)
    with pytest.raises(ValueError, match="CUDA"):
        S.action(x64.cpu(), y64)
    F = cd.DDH(5.0, np.ones(n), fem, nx, nx)
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    b = torch.zeros(F.size(), dtype=torch.float32, device=cuda)
    F.rhs(f, b)
    with pytest.raises(ValueError, match="dtype"):
        F.rhs(f, b.double())  # fp32 DDH: traces are float
    with pytest.raises(ValueError, match="at least"):
        F.rhs(f[:n], b)
    with pytest.raises(ValueError, match="dtype"):
        cd.gmres(F.size(), b.double(), F, b, 5, 2, 1e-3)


# ------------------------------------------------------------------ operators vs oracle
@pytest.mark.parametrize("kind", ["structured", "unstructured"])
@pytest.mark.parametrize("nb", [2, 3, 4, 5, 6, 7, 8])
def test_stiffness_mass_apply(cuda, kind, nb):
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes(kind)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    rng = np.random.default_rng(nb)
    xh = rng.standard_normal(d.ndof)
    coef = 0.5 + rng.random(d.ndof)
    x = to_dev(torch, xh, cuda)

    S = cd.StiffnessMatrix(fem)
    y = torch.full((d.ndof,), 7.0, dtype=torch.float64, device=cuda)  # action(x,y) must overwrite
    S.action(x, y)
    ref = oracle.Stiffness(d).apply(xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    S.action(-0.5, x, y)  # accumulate
    assert rel(y.cpu().numpy(), 0.5 * ref) < 1e-12

    S2 = cd.StiffnessMatrix(fem, nq=nb + 2)
    S2.action(x, y)
    assert rel(y.cpu().numpy(), oracle.Stiffness(d, nq=nb + 2).apply(xh)) < 1e-12

    M = cd.MassMatrix(fem)
    M.action(x, y)
    assert rel(y.cpu().numpy(), oracle.Mass(d).apply(xh)) < 1e-12
    Mw = cd.MassMatrix(fem, to_dev(torch, coef, cuda))
    Mw.action(x, y)
    refw = oracle.Mass(d, coef).apply(xh)
    assert rel(y.cpu().numpy(), refw) < 1e-12
    Mw.action(2.0, x, y)
    assert rel(y.cpu().numpy(), 3.0 * refw) < 1e-12

    Di = cd.DiagInvMassMatrix(fem, to_dev(torch, coef, cuda))
    Di.action(x, y)
    assert rel(y.cpu().numpy(), oracle.diag_inv_mass(d, coef) * xh) < 1e-13
    Di2 = cd.DiagInvMassMatrix(fem)
    z = x.clone()
    Di2.action(z, z)  # in place, as examples/DDH.cpp:123 does
    assert rel(z.cpu().numpy(), oracle.diag_inv_mass(d) * xh) < 1e-13


@pytest.mark.parametrize("nb", [3, 4, 7])
def test_operator_plan_and_generic_kernels_agree(cuda, nb, monkeypatch):
    """StiffnessMatrix / MassMatrix pick the patch-plan kernels (cuddh_hip_operator_plan_*) when one exists for
    (n_basis, n_quad) and the generic batched kernels otherwise; CUDDH_OPERATOR_PLAN=0 forces the latter.  Both must
    match the oracle; the plan path is also bitwise reproducible (no atomics)."""
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes("unstructured")
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    rng = np.random.default_rng(100 + nb)
    xh = rng.standard_normal(d.ndof)
    coef = 0.5 + rng.random(d.ndof)
    x = to_dev(torch, xh, cuda)
    refS = oracle.Stiffness(d).apply(xh)
    refM = oracle.Mass(d, coef).apply(xh)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CUDDH_OPERATOR_PLAN", mode)
        S = cd.StiffnessMatrix(fem)
        M = cd.MassMatrix(fem, to_dev(torch, coef, cuda))
        y = torch.full((d.ndof,), -3.0, dtype=torch.float64, device=cuda)
        S.action(x, y)
        assert rel(y.cpu().numpy(), refS) < 1e-12
        M.action(0.25, x, y)
        assert rel(y.cpu().numpy(), refS + 0.25 * refM) < 1e-12
        out[mode] = y.cpu().numpy().copy()
        if mode == "1":
            y2 = torch.full((d.ndof,), 11.0, dtype=torch.float64, device=cuda)
            S.action(x, y2)
            M.action(0.25, x, y2)
            assert np.array_equal(y2.cpu().numpy(), out["1"])
    assert rel(out["1"], out["0"]) < 1e-13


@pytest.mark.parametrize("kind", ["structured", "unstructured"])
@pytest.mark.parametrize("nb", [3, 4, 6])
def test_facemass_and_facespace(cuda, kind, nb):
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes(kind)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(11)
    xh = rng.standard_normal(ofs.size)
    ch = 0.5 + rng.random(ofs.size)
    x = to_dev(torch, xh, cuda)
    y = torch.full((ofs.size,), -3.0, dtype=torch.float64, device=cuda)
    H = cd.FaceMassMatrix(fs)
    H.action(x, y)
    assert rel(y.cpu().numpy(), oracle.FaceMass(ofs).apply(xh)) < 1e-12
    Hw = cd.FaceMassMatrix(fs, to_dev(torch, ch, cuda))
    Hw.action(x, y)
    ref = oracle.FaceMass(ofs, ch).apply(xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    Hw.action(0.25, x, y)
    assert rel(y.cpu().numpy(), 1.25 * ref) < 1e-12
    Di = cd.DiagInvFaceMassMatrix(fs, to_dev(torch, ch, cuda))
    Di.action(x, y)
    assert rel(y.cpu().numpy(), oracle.diag_inv_facemass(ofs, ch) * xh) < 1e-13

    # restrict / prolong / orth  (source/H1Space.cpp:189-219)
    gh = rng.standard_normal(d.ndof)
    g = to_dev(torch, gh, cuda)
    r = torch.zeros(ofs.size, dtype=torch.float64, device=cuda)
    fs.restrict(g, r)
    assert np.array_equal(r.cpu().numpy(), gh[ofs.proj])
    fs.prolong(x, g)
    exp = gh.copy()
    exp[ofs.proj] += xh
    assert np.array_equal(g.cpu().numpy(), exp)
    fs.orth(g)
    exp[ofs.proj] = 0.0
    assert np.array_equal(g.cpu().numpy(), exp)


@pytest.mark.parametrize("kind", ["structured", "unstructured"])
@pytest.mark.parametrize("p", [3, 5, 8])
def test_reference_mass_test_through_product(cuda, kind, p):
    """tests/mass.cpp:13-83 with the product's own LinearFunctional / MassMatrix / gmres."""
    import torch

    import cuddhelmholtz_amd as cd

    pm, _ = meshes(kind)
    fem = cd.H1Space(pm, cd.Basis(p))
    n = fem.size()
    f = torch.zeros(n, dtype=torch.float64, device=cuda)
    b = torch.zeros_like(f)
    Mf = torch.zeros_like(f)
    u = torch.zeros_like(f)
    cd.nodal_values(fem, cd.MASS_POLY, f)
    cd.linear_functional(fem, cd.MASS_POLY, b, nq=p + 2)
    M = cd.MassMatrix(fem)
    Pc = cd.DiagInvMassMatrix(fem)
    M.action(f, Mf)
    assert float(torch.linalg.norm(Mf - b) / torch.linalg.norm(b)) < 1e-8
    out = cd.gmres(n, u, M, b, 5, 10, 1e-12, Precond=Pc)
    assert out.num_matvec > 0
    assert float(torch.linalg.norm(u - f) / torch.linalg.norm(f)) < 1e-8


@pytest.mark.parametrize("kind", ["structured", "unstructured"])
@pytest.mark.parametrize("p", [6, 8])
def test_reference_stiffness_test_through_product(cuda, kind, p):
    """tests/stiffness.cpp:28-72"""
    import torch

    import cuddhelmholtz_amd as cd

    pm, _ = meshes(kind)
    fem = cd.H1Space(pm, cd.Basis(p))
    n = fem.size()
    f = torch.zeros(n, dtype=torch.float64, device=cuda)
    Lf = torch.zeros_like(f)
    Af = torch.zeros_like(f)
    cd.nodal_values(fem, cd.STIFF_FUNC, f)
    cd.linear_functional(fem, cd.STIFF_NEG_LAPLACIAN, Lf, nq=p + 2)
    cd.StiffnessMatrix(fem, nq=p + 2).action(f, Af)
    assert float(torch.linalg.norm(Af - Lf) / torch.linalg.norm(Lf)) < 1e-6


def test_linear_functionals_vs_oracle(cuda):
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes("unstructured")
    nb = 4
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    F = torch.zeros(d.ndof, dtype=torch.float64, device=cuda)
    omega = 7.0
    cd.linear_functional(fem, cd.GAUSSIANS, F, param=omega)
    assert rel(F.cpu().numpy(), oracle.linear_functional(d, oracle.gaussians(omega))) < 1e-12
    cd.linear_functional(fem, cd.GAUSSIANS, F, param=omega, nq=7)
    assert rel(F.cpu().numpy(), oracle.linear_functional(d, oracle.gaussians(omega), nq=7)) < 1e-12


# ------------------------------------------------------------------ affine meshes
@pytest.mark.parametrize("nb", [2, 3, 4, 5, 7])
def test_affine_and_general_plans_agree(cuda, nb, monkeypatch):
    """On a uniform mesh every element has the same stiffness metric (and the same unweighted mass weights): the plans
    then read ONE copy through scalar loads.  CUDDH_PLAN_AFFINE=0 keeps the general per-element arrays; both forms must
    match the oracle, and each other to rounding."""
    import torch

    import cuddhelmholtz_amd as cd

    nx = 11
    pm, om = meshes("structured", nx)
    d = oracle.Discretization(om, nb)
    rng = np.random.default_rng(40 + nb)
    xh = rng.standard_normal(2 * d.ndof)
    a2 = 0.5 + rng.random(d.ndof)
    faces = pm.boundary_edges()
    ofs = oracle.FaceSpaceO(d, list(faces))
    ax = 0.5 + rng.random(ofs.size)
    omega = 7.0
    refS = oracle.Stiffness(d).apply(xh[: d.ndof])
    refM = oracle.Mass(d).apply(xh[: d.ndof])
    refA = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CUDDH_PLAN_AFFINE", mode)
        fem = cd.H1Space(pm, cd.Basis(nb))
        fs = cd.FaceSpace(fem, faces)
        x = to_dev(torch, xh, cuda)
        yS = torch.full((d.ndof,), 3.0, dtype=torch.float64, device=cuda)
        yM = torch.full((d.ndof,), 3.0, dtype=torch.float64, device=cuda)
        cd.StiffnessMatrix(fem).action(x[: d.ndof], yS)
        cd.MassMatrix(fem).action(x[: d.ndof], yM)
        A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
        y = torch.empty(2 * d.ndof, dtype=torch.float64, device=cuda)
        A.action(x, y)
        assert A.fused() and (A.bytes_affine() > 0) == (mode == "1")
        if mode == "1":  # only the stiffness metric is uniform here (the mass carries the random coefficient a2)
            n_elem = nx * nx
            assert A.bytes_per_apply() - A.bytes_affine() == n_elem * 3 * (nb + 1) ** 2 * 8
        assert rel(yS.cpu().numpy(), refS) < 1e-12 and rel(yM.cpu().numpy(), refM) < 1e-12
        assert rel(y.cpu().numpy(), refA) < 1e-12
        got[mode] = (yS.cpu().numpy(), yM.cpu().numpy(), y.cpu().numpy())
    for a, b in zip(got["1"], got["0"]):
        assert rel(a, b) < 1e-13


# ------------------------------------------------------------------ patch sizes
@pytest.mark.parametrize("kind,nx", [("structured", 13), ("unstructured", 0)])
@pytest.mark.parametrize("nb", [2, 3, 4, 5])
def test_patch_sizes_agree(cuda, kind, nx, nb, monkeypatch):
    """The plan kernels exist for 32- and 64-element patches (real operators: 64 by default; fused apply: 64 for affine
    plans and, through helm_lane_kernel, for large plans on general geometry, 32 otherwise), each with and without the
    non-temporal metric loads large plans use (NT template flag, chosen by plan size; CUDDH_PLAN_STREAMING forces it) and
    with per-element or uniform (UG) metrics.  CUDDH_OP_PE / CUDDH_HELM_PE / CUDDH_HELM_LANE force a form; every
    combination must run the kernel instantiation it is meant to (asserted through kernel()), match the oracle, and agree
    with the others to rounding (the summation order inside a patch differs).  This is how the instantiations the 1024^2
    benchmark uses (lane form + NT) are checked against the oracle on meshes the oracle can handle."""
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes(kind, nx)
    d = oracle.Discretization(om, nb)
    rng = np.random.default_rng(70 + nb)
    xh = rng.standard_normal(2 * d.ndof)
    a2 = 0.5 + rng.random(d.ndof)
    faces = pm.boundary_edges()
    ofs = oracle.FaceSpaceO(d, list(faces))
    ax = 0.5 + rng.random(ofs.size)
    omega = 5.0
    refS = oracle.Stiffness(d).apply(xh[: d.ndof])
    refM = oracle.Mass(d, a2).apply(xh[: d.ndof])
    refA = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    nqS, nqM = nb + 1, 2 + 3 * nb // 2
    got, seen = {}, set()
    # "lane0": helm_lane_kernel (n_basis <= 4; chosen by size otherwise), the slice-by-slice chain large plans run by default;
    # "lane": the same with the whole patch's metric block requested up front (CUDDH_HELM_PRE=1, one wavefront per SIMD)
    # n_basis 3, general layout, helm_patch_kernel: the mass phase takes two slices per round trip (MODE=1) unless
    # CUDDH_HELM_PAIR_MASS=0 ("32m0" / "64m0": the one-slice form)
    for pe in ("32", "64", "lane", "lane0") + (("32m0", "64m0") if nb == 3 else ()):
        pair_mass = not pe.endswith("m0")
        monkeypatch.setenv("CUDDH_HELM_PAIR_MASS", "1" if pair_mass else "0")
        pe = pe[:2] if pe.endswith("m0") else pe
        for affine in ("1", "0"):
            for nt in ("0", "1"):
                if pe.startswith("lane"):
                    monkeypatch.setenv("CUDDH_OP_PE", "64")
                    monkeypatch.delenv("CUDDH_HELM_PE", raising=False)
                    monkeypatch.setenv("CUDDH_HELM_LANE", "1")
                    monkeypatch.setenv("CUDDH_HELM_PRE", {"lane": "1", "lane0": "0"}[pe])
                else:
                    monkeypatch.setenv("CUDDH_OP_PE", pe)
                    monkeypatch.setenv("CUDDH_HELM_PE", pe)
                    monkeypatch.setenv("CUDDH_HELM_LANE", "0")
                monkeypatch.setenv("CUDDH_PLAN_AFFINE", affine)
                monkeypatch.setenv("CUDDH_PLAN_STREAMING", nt)
                fem = cd.H1Space(pm, cd.Basis(nb))
                fs = cd.FaceSpace(fem, faces)
                x = to_dev(torch, xh, cuda)
                a2d = to_dev(torch, a2, cuda)
                yS = torch.full((d.ndof,), -2.0, dtype=torch.float64, device=cuda)
                yM = torch.full((d.ndof,), -2.0, dtype=torch.float64, device=cuda)
                S, M = cd.StiffnessMatrix(fem), cd.MassMatrix(fem, a2d)
                S.action(x[: d.ndof], yS)
                M.action(x[: d.ndof], yM)
                A = cd.HelmholtzOperator(omega, a2d, to_dev(torch, ax, cuda), fem, fs)
                y = torch.empty(2 * d.ndof, dtype=torch.float64, device=cuda)
                A.action(x, y)
                assert A.fused()
                # which instantiation ran
                ug = int(kind == "structured" and affine == "1")  # uniform stiffness metric, read through scalar loads
                lane_like = pe.startswith("lane")
                ope = 64 if lane_like else int(pe)
                assert S.kernel() == f"op_patch_kernel<{nb},{nqS},0,NT={0 if ug else int(nt)},UG={ug},PEK={ope}> pe={ope}", S.kernel()
                assert M.kernel() == f"op_patch_kernel<{nb},{nqM},1,NT={nt},UG=0,PEK={ope}> pe={ope}", M.kernel()  # a2 varies: never uniform
                if lane_like and nb <= 4 and (not ug or nb == 2):
                    want = f"helm_lane_kernel<{nb},{nqS},{nqM},NT={nt},UG={ug}{',PRE=1' if pe != 'lane0' else ''}> pe=64"
                else:
                    # forced sizes apply to n_basis <= 4; affine plans are 64-element ones unless forced; a lane request that
                    # does not apply (affine n_basis 3, 4; n_basis 5) leaves the default size
                    hpe = (int(pe) if not lane_like else (64 if ug else 32)) if nb <= 4 else 32
                    mode = ",MODE=1" if (nb == 3 and not ug and pair_mass) else ""
                    want = f"helm_patch_kernel<{nb},{nqS},{nqM},NT={nt},UG={ug},PEK={hpe}{mode}> pe={hpe}"
                assert A.kernel() == want, (A.kernel(), want)
                seen.add(A.kernel())
                assert rel(yS.cpu().numpy(), refS) < 1e-12 and rel(yM.cpu().numpy(), refM) < 1e-12
                assert rel(y.cpu().numpy(), refA) < 1e-12
                # accumulate form: y <- y + c Op x
                yS2 = to_dev(torch, refS, cuda).clone()
                cd.StiffnessMatrix(fem).action(-1.0, x[: d.ndof], yS2)
                assert float(yS2.abs().max()) < 1e-11 * float(np.abs(refS).max())
                got[(pe, affine, nt, pair_mass)] = (yS.cpu().numpy(), yM.cpu().numpy(), y.cpu().numpy())
    base = got[("32", "0", "0", True)]
    for key, val in got.items():
        for a, b in zip(val, base):
            assert rel(a, b) < 1e-13, key
    if nb <= 4:  # the instantiations large general-geometry plans (the benchmark's) run, and the chain form they replace
        assert f"helm_lane_kernel<{nb},{nqS},{nqM},NT=1,UG=0,PRE=1> pe=64" in seen
        assert f"helm_lane_kernel<{nb},{nqS},{nqM},NT=1,UG=0> pe=64" in seen
    if nb == 2 and kind == "structured":
        assert "helm_lane_kernel<2,3,5,NT=1,UG=1,PRE=1> pe=64" in seen
    if nb == 3:
        assert {"helm_patch_kernel<3,4,6,NT=1,UG=0,PEK=32,MODE=1> pe=32", "helm_patch_kernel<3,4,6,NT=1,UG=0,PEK=32> pe=32"} <= seen


# ------------------------------------------------------------------ fused Helmholtz apply
@pytest.mark.parametrize("kind,nx", [("structured", 10), ("structured", 37), ("unstructured", 0)])
@pytest.mark.parametrize("nb", [2, 3, 4, 5, 6, 7, 8])
def test_fused_helmholtz_apply(cuda, kind, nx, nb):
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes(kind, nx)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(5)
    a2 = 0.5 + rng.random(d.ndof)
    ax = 0.5 + rng.random(ofs.size)
    omega = 9.0
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    assert A.fused()
    xh = rng.standard_normal(2 * d.ndof)
    x = to_dev(torch, xh, cuda)
    y = torch.full((2 * d.ndof,), 123.0, dtype=torch.float64, device=cuda)
    A.action(x, y)
    ref = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    y2 = torch.zeros_like(y)
    A.action_unfused(x, y2)
    assert rel(y2.cpu().numpy(), ref) < 1e-12
    # the fused apply is deterministic: bitwise identical on repetition
    y3 = torch.zeros_like(y)
    A.action(x, y3)
    assert torch.equal(y, y3)
    assert A.bytes_per_apply() > 0 and A.bytes_per_apply(actual=True) > 0


@pytest.mark.parametrize("affine", ["0", "1"])
@pytest.mark.parametrize("kind,nx", [("structured", 10), ("structured", 37), ("unstructured", 0), ("refined", 2)])
@pytest.mark.parametrize("nb,variant", [(2, "lane"), (3, "lane"), (4, "lane"), (2, "patch"), (3, "patch"), (4, "patch"), (5, "patch"), (5, "mfma"), (6, "mfma"),
                                        (7, "mfma"), (8, "mfma")])
def test_fused_helmholtz_apply_native_ordering(cuda, monkeypatch, kind, nx, nb, variant, affine):
    """The lane form (n_basis 2-4) and the matrix-core form (n_basis 5-8) of the fused apply on vectors in the PLAN'S OWN ordering (pairs (u, v); a patch's owned dofs contiguous;
    include/cuddh_hip.h: cuddh_hip_helmholtz_apply_native): against the oracle through the permutation, and bitwise against
    the reference-ordering apply (same arithmetic, same order per element and per dof); gmres() on the native vectors takes the
    same iterations as on the reference ordering.  Every fused kernel: helm_lane_kernel (forced with CUDDH_HELM_LANE=1: its size
    rule would pick helm_patch_kernel for meshes this small), helm_patch_kernel (n_basis 2-5, 32- and 64-element patches),
    helm_mfma_kernel (n_basis 6-8, and 5 on request); general-geometry layout and, on the uniform meshes, the affine form."""
    import torch

    import cuddhelmholtz_amd as cd

    if nb <= 4 and variant != "patch":
        monkeypatch.setenv("CUDDH_HELM_LANE", "1")
    if nb == 5 and variant != "patch":
        monkeypatch.setenv("CUDDH_HELM_NB5_MFMA", "1")  # n_basis 5 in the matrix-core scheme (16-element batches)
    monkeypatch.setenv("CUDDH_PLAN_AFFINE", affine)
    if kind == "refined":
        xy, elems = load_unstructured_square()
        pm = cd.Mesh2D.from_vertices(xy, elems).refined(nx)
        om = oracle.Mesh(pm.vertices(), pm.elements())
    else:
        pm, om = meshes(kind, nx)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    n = d.ndof
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(50 + nb)
    a2 = 0.5 + rng.random(n)
    ax = 0.5 + rng.random(ofs.size)
    omega = 7.0
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    assert A.fused() and A.has_native()
    if affine == "1" and kind == "structured" and nb in (3, 4):
        assert A.kernel().startswith(f"helm_patch_kernel<{nb},")  # affine plans of n_basis 3, 4: two wavefronts per 64-element patch
    elif variant == "patch":
        assert A.kernel().startswith(f"helm_patch_kernel<{nb},")
    else:
        assert A.kernel().startswith(f"helm_lane_kernel<{nb}," if nb <= 4 else f"helm_mfma_kernel<{nb},")
    xh = rng.standard_normal(2 * n)
    x = to_dev(torch, xh, cuda)
    z = torch.full((2 * n,), 7.0, dtype=torch.float64, device=cuda)
    A.to_native(x, z)
    # a permutation: pairs (u_g, v_g), every dof exactly once
    zp = z.cpu().numpy().reshape(n, 2)
    key = lambda a: np.sort(a.view(np.complex128).ravel())  # noqa: E731
    assert np.array_equal(key(np.ascontiguousarray(zp)), key(np.ascontiguousarray(np.stack([xh[:n], xh[n:]], axis=1))))
    back = torch.zeros_like(x)
    A.from_native(z, back)
    assert torch.equal(back, x)
    y = torch.full((2 * n,), 123.0, dtype=torch.float64, device=cuda)
    A.action(x, y)
    zy = torch.full((2 * n,), -5.0, dtype=torch.float64, device=cuda)
    A.action_native(z, zy)
    yn = torch.zeros_like(y)
    A.from_native(zy, yn)
    ref = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    assert rel(yn.cpu().numpy(), ref) < 1e-12
    assert torch.equal(yn, y)
    zy2 = torch.zeros_like(zy)
    A.action_native(z, zy2)
    assert torch.equal(zy, zy2)
    assert 0 < A.bytes_native() < A.bytes_per_apply(actual=True)
    # GMRES(10), two cycles: native iteration vectors vs reference ordering (the inner products sum in another order)
    bh = rng.standard_normal(2 * n)
    b = to_dev(torch, bh, cuda)
    x1, x2 = torch.zeros_like(b), torch.zeros_like(b)
    o1 = cd.gmres(2 * n, x1, A, b, 10, 3, 0.0)
    o2 = A.gmres(x2, b, 10, 3, 0.0)
    assert o1.num_matvec == o2.num_matvec
    assert abs(o1.res_norm[-1] - o2.res_norm[-1]) <= 1e-10 * o1.res_norm[0]
    assert rel(x2.cpu().numpy(), x1.cpu().numpy()) < 1e-9


# ------------------------------------------------------------------ GMRES
def test_gmres_toeplitz_callback(cuda):
    """tests/gmres.cpp:41-77 with a Python-side operator."""
    import torch

    import cuddhelmholtz_amd as cd

    n = 1 << 10
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = torch.rand(n, generator=g, dtype=torch.float64).to(cuda)

    def A(x, y):
        y.copy_(-3.0 * x)
        y[1:] += x[:-1]
        y[:-1] += 1.5 * x[1:]

    b = torch.zeros_like(xs)
    A(xs, b)
    x = torch.zeros_like(xs)
    out = cd.gmres(n, x, A, b, 5, 100, 1e-10)
    assert out.success
    r = torch.zeros_like(xs)
    A(x, r)
    assert float(torch.linalg.norm(r - b) / torch.linalg.norm(b)) < 1e-10
    assert out.num_matvec >= out.num_iter
    assert len(out.res_norm) >= 2 and out.res_norm[-1] < out.res_norm[0]


# ------------------------------------------------------------------ DDH
def ddh_case(nx, nb):
    omega = 2 * math.pi * nx / 10
    om = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    d = oracle.Discretization(om, nb)
    h_a = d.nodal(oracle.alpha_disk)
    f = np.concatenate([oracle.linear_functional(d, oracle.gaussians(omega)), 0.1 * oracle.linear_functional(d, oracle.mass_poly)])
    return omega, d, h_a, f


@pytest.mark.parametrize("real", ["f64", "f32"])
def test_ddh_geometric_factors_table_and_corner_forms(cuda, real):
    """source/DDH.cpp:15-58 on an irregular mesh (every element its own Jacobian; three 'subdomains' of unequal element
    counts): cuddh_hip_ddh_geom_setup_* reading the tabulated Jacobians, cuddh_hip_ddh_geom_from_corners_* evaluating
    the bilinear map itself (the form DDH's constructor uses), and the oracle."""
    import ctypes as C

    import torch

    from cuddhelmholtz_amd import _native as N

    lib = N.lib
    _, om = meshes("unstructured")
    nb = 4
    d = oracle.Discretization(om, nb)
    nel = om.n_elem
    J, _, _ = d.metrics(d.gll_x)
    rng = np.random.default_rng(5)
    counts = np.array([40, 17, 33], dtype=np.int32)
    mx = int(counts.max())
    elems = np.zeros((mx, 3), dtype=np.int32, order="F")
    for s in range(3):
        elems[:counts[s], s] = rng.choice(nel, counts[s], replace=False)
    npt = np.float64 if real == "f64" else np.float32
    Gref = np.zeros((3, nb * nb * mx, 3), dtype=npt, order="F")
    fn = oracle.lib().orc_ddh_geom_f64 if real == "f64" else oracle.lib().orc_ddh_geom_f32
    fn(C.c_int(3), C.c_int(mx), C.c_int(nb), oracle._p(counts), oracle._p(elems), oracle._p(d.gll_w), oracle._p(J), oracle._p(Gref))

    tt = torch.float64 if real == "f64" else torch.float32
    dev = lambda a: to_dev(torch, np.asarray(a).reshape(-1, order="F"), cuda)  # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dc, de, dw, dq = dev(counts), dev(elems), dev(d.gll_w), dev(d.gll_x)
    dcorn = dev(np.ascontiguousarray(d.corners, dtype=np.float64))
    dJ = torch.zeros(4 * nb * nb * nel, dtype=torch.float64, device=cuda)
    assert lib.cuddh_hip_element_metrics(nel, nb, p(dcorn), p(dq), p(dJ), None, None, st) == 0
    assert rel(dJ.cpu().numpy(), J.reshape(-1, order="F")) < 1e-14
    G_tab = torch.full((Gref.size,), 7.0, dtype=tt, device=cuda)
    G_cor = torch.full((Gref.size,), 7.0, dtype=tt, device=cuda)
    sfx = "f64" if real == "f64" else "f32"
    assert getattr(lib, "cuddh_hip_ddh_geom_setup_" + sfx)(3, mx, nel, nb, p(dc), p(de), p(dw), p(dJ), p(G_tab), st) == 0
    assert getattr(lib, "cuddh_hip_ddh_geom_from_corners_" + sfx)(3, mx, nb, p(dc), p(de), p(dw), p(dq), p(dcorn), p(G_cor), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(G_tab, G_cor)  # one definition of the bilinear Jacobian: the two forms round alike
    assert rel(G_cor.cpu().numpy(), Gref.reshape(-1, order="F")) < (1e-13 if real == "f64" else 2e-7)
    pad = G_cor.cpu().numpy().reshape(Gref.shape, order="F")[:, nb * nb * 17:, 1]
    assert not pad.any()  # entries past a subdomain's element count are zero, as in the reference
    assert getattr(lib, "cuddh_hip_ddh_geom_from_corners_" + sfx)(3, mx, nb, p(dc), p(de), p(dw), p(dq), None, p(G_cor), st) != 0


@pytest.mark.parametrize("nx,nb,kernel", [(8, 4, 1), (8, 4, 2), (16, 4, 2), (8, 8, 1), (8, 8, 6), (6, 8, 6), (10, 3, 1), (9, 5, 1), (16, 2, 1)])
def test_ddh_fp64_entry_points(cuda, nx, nb, kernel):
    import torch

    import cuddhelmholtz_amd as cd

    omega, d, h_a, fh = ddh_case(nx, nb)
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, h_a, fem, nx, nx, precision="f64", kernel=kernel)
    O = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    assert F.info()["kernel"] == kernel
    f = to_dev(torch, fh, cuda)
    n = F.size()
    b = torch.zeros(n, dtype=torch.float64, device=cuda)
    F.rhs(f, b)
    b_ref = O.rhs(fh)
    assert rel(b.cpu().numpy(), b_ref) < 1e-10
    rng = np.random.default_rng(2)
    lam_h = rng.standard_normal(n)
    lam_h[np.setdiff1d(np.arange(n), np.concatenate([np.unique(O.t.B[O.t.B >= 0]), np.unique(O.t.B[O.t.B >= 0]) + O.t.n_lambda]))] = 0.0
    lam = to_dev(torch, lam_h, cuda)
    y = torch.full((n,), 5.0, dtype=torch.float64, device=cuda)
    F.action(lam, y)
    # slots no subdomain writes keep their previous content in `update`; compare written slots and lambda - T lambda
    y_ref = O.action(lam_h)
    written = np.unique(O.t.B[:, 1, :][O.t.B[:, 1, :] >= 0])
    written = np.concatenate([written, written + O.t.n_lambda])
    assert rel(y.cpu().numpy()[written], y_ref[written]) < 1e-10
    u = torch.full((2 * d.ndof,), 9.0, dtype=torch.float64, device=cuda)
    F.postprocess(lam, f, u)
    u_ref = O.postprocess(lam_h, fh)
    assert rel(u.cpu().numpy(), u_ref) < 1e-10
    # the solution is assembled in a fixed order (no atomics, unlike source/DDH.cpp:303,306): bitwise reproducible, and the
    # sharded form (two ranges accumulated one after the other) is bitwise the whole
    u2 = torch.full_like(u, -3.0)
    F.postprocess(lam, f, u2)
    assert torch.equal(u, u2)
    nd0 = F.info()["n_domains"]
    u3 = torch.full_like(u, 7.0)
    F.local_solution(0, nd0 // 2, lam, f, u3, True)
    F.local_solution(nd0 // 2, nd0, lam, f, u3, False)
    assert torch.equal(u, u3)
    # sharded entry points (multi-GPU path): two halves reproduce the whole
    nd = F.info()["n_domains"]
    upd = torch.zeros(n, dtype=torch.float64, device=cuda)
    F.local_traces(0, nd // 2, None, lam, upd)
    F.local_traces(nd // 2, nd, None, lam, upd)
    full = torch.zeros(n, dtype=torch.float64, device=cuda)
    F.local_traces(0, nd, None, lam, full)
    assert torch.equal(upd, full)


@pytest.mark.parametrize("nx,nb,kernel", [(8, 4, 1), (8, 4, 2), (8, 4, 3), (8, 4, 4), (8, 4, 5), (16, 4, 2), (16, 4, 3), (16, 4, 4), (16, 4, 5), (8, 8, 1), (8, 8, 6), (8, 8, 7)])
def test_ddh_fp32_reference_precision(cuda, nx, nb, kernel):
    import torch

    import cuddhelmholtz_amd as cd

    omega, d, h_a, fh = ddh_case(nx, nb)
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, h_a, fem, nx, nx, precision="f32", kernel=kernel)
    assert F.info()["kernel"] == kernel
    O64 = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    O32 = oracle.DDH(d, nx, nx, omega, h_a, np.float32)
    f = to_dev(torch, fh, cuda)
    n = F.size()
    b = torch.zeros(n, dtype=torch.float32, device=cuda)
    F.rhs(f, b)
    b64 = O64.rhs(fh)
    e_gpu = rel(b.cpu().numpy(), b64)
    e_cpu = rel(O32.rhs(fh), b64)
    print(f"DDH fp32 rhs: GPU vs fp64 oracle {e_gpu:.2e}; fp32 oracle vs fp64 oracle {e_cpu:.2e}")
    assert e_gpu < 2e-4
    u = torch.zeros(2 * d.ndof, dtype=torch.float64, device=cuda)
    F.postprocess(b, f, u)
    u64 = O64.postprocess(b64, fh)
    assert rel(u.cpu().numpy(), u64) < 2e-4
    # determinism: bitwise identical traces on repetition (the reference's LDS atomics are not)
    b2 = torch.zeros_like(b)
    F.rhs(f, b2)
    assert torch.equal(b, b2)
    # odd subdomain ranges (kernel 6 packs two subdomains per wavefront: the last one of a range may be alone)
    nd = F.info()["n_domains"]
    part = torch.zeros_like(b)
    F.local_traces(0, 3, f, None, part)
    F.local_traces(3, nd, f, None, part)
    assert torch.equal(part, b)
    # listed subdomains in one launch (the split multi-GPU schedule: non-contiguous, any order, odd counts)
    rng = np.random.default_rng(nx + kernel)
    perm = rng.permutation(nd).astype(np.int32)
    cut = max(1, nd // 3) | 1
    listed = torch.zeros_like(b)
    for ids in (perm[:cut], perm[cut:]):
        if len(ids):
            F.local_traces_listed(to_dev(torch, ids, cuda), f, None, listed)
    assert torch.equal(listed, b)
    # the multi-GPU schedule's issue priority (s_setprio) changes when wavefronts are issued, never what they compute
    F.set_wave_priority(True)
    hi = torch.zeros_like(b)
    F.local_traces(0, 3, f, None, hi)
    F.set_wave_priority(False)
    F.local_traces(3, nd, f, None, hi)
    assert torch.equal(hi, b)


@pytest.mark.parametrize("kernel", [1, 2])
def test_ddh_gmres_solve_fp64(cuda, kernel):
    """rhs -> gmres -> postprocess (examples/DDH.cpp:141-144) against the same pipeline on the oracle."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb = 8, 4
    omega, d, h_a, fh = ddh_case(nx, nb)
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, h_a, fem, nx, nx, precision="f64", kernel=kernel)
    O = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    f = to_dev(torch, fh, cuda)
    n = F.size()
    b = torch.zeros(n, dtype=torch.float64, device=cuda)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * d.ndof, dtype=torch.float64, device=cuda)
    F.rhs(f, b)
    out = cd.gmres(n, lam, F, b, 20, 20, 1e-10)
    F.postprocess(lam, f, u)
    b_ref = O.rhs(fh)
    lam_ref, info = oracle.gmres(O.action, b_ref, m=20, maxit=20, tol=1e-10)
    u_ref = O.postprocess(lam_ref, fh)
    assert out.success == info["success"]
    assert abs(out.num_matvec - info["num_matvec"]) <= 1
    assert rel(u.cpu().numpy(), u_ref) < 1e-8


@pytest.mark.parametrize("nx", [16, 32])
def test_ddh_fp32_solve_reference_flow(cuda, nx):
    """The reference's own flow in the reference's own precision (examples/DDH.cpp:141-144: rhs -> float GMRES(20),
    tol 1e-4 -> postprocess) on the benchmarked kernel (5: dense element matrix on the matrix cores), against the same flow
    on the oracle in fp64 (and, at nx = 16, in fp32): same GMRES history to within a step, solutions to fp32 accuracy."""
    import torch

    import cuddhelmholtz_amd as cd

    nb = 4
    omega, d, h_a, fh = ddh_case(nx, nb)
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, h_a, fem, nx, nx, precision="f32", kernel=5)
    assert F.info()["kernel"] == 5
    f = to_dev(torch, fh, cuda)
    n = F.size()
    b = torch.zeros(n, dtype=torch.float32, device=cuda)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * d.ndof, dtype=torch.float64, device=cuda)
    F.rhs(f, b)
    out = cd.gmres(n, lam, F, b, 20, 100, 1e-4)
    F.postprocess(lam, f, u)
    assert out.success
    O64 = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    lam64, info64 = oracle.gmres(O64.action, O64.rhs(fh), m=20, maxit=100, tol=1e-4)
    u64 = O64.postprocess(lam64, fh)
    assert info64["success"]
    e = rel(u.cpu().numpy(), u64)
    msg = f"DDH fp32 kernel 5 solve {nx}x{nx}: {out.num_matvec} matvecs (fp64 oracle {info64['num_matvec']}), u vs fp64 oracle {e:.2e}"
    if nx == 16:
        O32 = oracle.DDH(d, nx, nx, omega, h_a, np.float32)
        lam32, info32 = oracle.gmres(O32.action, O32.rhs(fh), m=20, maxit=100, tol=1e-4, dtype=np.float32)
        u32 = O32.postprocess(lam32, fh)
        msg += f"; fp32 oracle: {info32['num_matvec']} matvecs, u vs fp64 oracle {rel(u32, u64):.2e}, GPU vs fp32 oracle {rel(u.cpu().numpy(), u32):.2e}"
        assert abs(out.num_matvec - info32["num_matvec"]) <= 2
    print(msg)
    assert abs(out.num_matvec - info64["num_matvec"]) <= 2
    # both runs stop at a relative residual of 1e-4, so the solutions agree to a small multiple of that
    assert e < 2e-3


def test_sharded_ddh_single_rank_equals_ddh(cuda):
    """cuddhelmholtz_amd.dist.ShardedDDH with one rank is DDH::action / rhs / postprocess."""
    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import ShardedDDH

    nx, nb = 16, 4
    omega, d, h_a, fh = ddh_case(nx, nb)
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    F = cd.DDH(omega, h_a, fem, nx, nx)
    sh = ShardedDDH(F, F.info()["n_domains"])
    f = to_dev(torch, fh, cuda)
    n = F.size()
    b1 = torch.zeros(n, dtype=torch.float32, device=cuda)
    b2 = torch.zeros_like(b1)
    F.rhs(f, b1)
    sh.rhs(f, b2)
    assert torch.equal(b1, b2)
    y1 = torch.zeros_like(b1)
    y2 = torch.full_like(b1, 3.0)
    F.action(b1, y1)
    sh.action(b1, y2)
    assert torch.equal(y1, y2)
    u1 = torch.zeros(2 * d.ndof, dtype=torch.float64, device=cuda)
    u2 = torch.ones_like(u1)
    F.postprocess(b1, f, u1)
    sh.postprocess(b1, f, u2)
    assert torch.equal(u1, u2)  # fixed-order assembly of the solution


def test_unpreconditioned_gmres_on_helmholtz_operator(cuda):
    """BASELINE config 2 in miniature: GMRES(20) on the fused complex Helmholtz operator, against the oracle's
    GMRES on the oracle's operator (same iteration count, fp64)."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb, omega = 12, 3, 5.0
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    om = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    d = oracle.Discretization(om, nb)
    ofs = oracle.FaceSpaceO(d, om.boundary_edges)
    n = d.ndof
    a2 = d.nodal(oracle.alpha_disk) ** 2
    ax = np.ones(ofs.size)
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    S, M, H = oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax)
    bh = np.concatenate([oracle.linear_functional(d, oracle.gaussians(omega)), np.zeros(n)])
    b = to_dev(torch, bh, cuda)
    x = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    out = cd.gmres(2 * n, x, A, b, 20, 4, 1e-30)
    x_ref, info = oracle.gmres(lambda v: oracle.helmholtz_apply(d, S, M, H, ofs, omega, v), bh, m=20, maxit=4, tol=1e-30)
    assert out.num_matvec == info["num_matvec"]
    assert rel(x.cpu().numpy(), x_ref) < 1e-8
    assert out.res_norm[-1] < out.res_norm[0]


def test_poisson_pipeline(cuda):
    """BASELINE config 1 in miniature: the flow of examples/Poisson.cpp:111-160 (lifting of Dirichlet data through
    FaceLinearFunctional + FaceMassMatrix solve, StiffnessMatrix + orth, GMRES) against the same flow on the oracle."""
    import torch

    import cuddhelmholtz_amd as cd

    nx, nb = 12, 3
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    faces = mesh.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    n, nf = fem.size(), fs.size()
    z = lambda m: torch.zeros(m, dtype=torch.float64, device=cuda)  # noqa: E731
    u, b, G, q, y = z(n), z(n), z(n), z(nf), z(nf)
    S = cd.StiffnessMatrix(fem)

    class Poisson:
        def __call__(self, xin, yout):
            S.action(xin, yout)
            fs.orth(yout)

    cd.linear_functional(fem, cd.CONSTANT, b, param=1.0)
    fs.orth(b)
    cd.face_linear_functional(fs, cd.MASS_POLY, y)
    m, pinv = cd.FaceMassMatrix(fs), cd.DiagInvFaceMassMatrix(fs)
    cd.gmres(nf, q, m, y, 5, 10, 1e-12, Precond=pinv)
    fs.prolong(q, G)
    tmp = z(n)
    Poisson()(G, tmp)
    b -= tmp
    out = cd.gmres(n, u, Poisson(), b, 20, 20, 1e-10)
    u += G

    om = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    d = oracle.Discretization(om, nb)
    ofs = oracle.FaceSpaceO(d, list(faces))
    So = oracle.Stiffness(d)

    def Ao(v):
        w = So.apply(v)
        w[ofs.proj] = 0.0
        return w

    bo = oracle.linear_functional(d, lambda xx, yy: np.ones_like(xx))
    bo[ofs.proj] = 0.0
    yo = oracle.face_linear_functional(ofs, oracle.mass_poly)
    assert rel(y.cpu().numpy(), yo) < 1e-12
    Mo, po = oracle.FaceMass(ofs), oracle.diag_inv_facemass(ofs)
    qo, _ = oracle.gmres(lambda v: po * Mo.apply(v), po * yo, m=5, maxit=10, tol=1e-12)
    assert rel(q.cpu().numpy(), qo) < 1e-9
    Go = np.zeros(d.ndof)
    Go[ofs.proj] += qo
    bo -= Ao(Go)
    uo, info = oracle.gmres(Ao, bo, m=20, maxit=20, tol=1e-10)
    uo += Go
    assert out.success and info["success"]
    assert rel(u.cpu().numpy(), uo) < 1e-7


# ------------------------------------------------------------------ refined unstructured mesh (irregular gather/scatter)
def test_fused_apply_on_refined_unstructured_mesh(cuda):
    """The reference's unstructured fixture refined twice (1,904 quads, every element with its own metric tensor):
    the fused apply and the single-operator plans against the oracle."""
    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.meshtools import refine_quads

    xy, elems = refine_quads(*load_unstructured_square(), times=2)
    assert len(elems) == 119 * 16
    pm, om = cd.Mesh2D.from_vertices(xy, elems), oracle.Mesh(xy, elems)
    nb = 4
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    assert fem.size() == d.ndof
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(77)
    a2, ax = 0.5 + rng.random(d.ndof), 0.5 + rng.random(ofs.size)
    omega = 11.0
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    assert A.fused() and A.bytes_affine() == 0  # no two elements share a metric tensor here
    xh = rng.standard_normal(2 * d.ndof)
    x = to_dev(torch, xh, cuda)
    y = torch.empty_like(x)
    A.action(x, y)
    ref = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    yS = torch.empty(d.ndof, dtype=torch.float64, device=cuda)
    cd.StiffnessMatrix(fem).action(x[: d.ndof], yS)
    assert rel(yS.cpu().numpy(), oracle.Stiffness(d).apply(xh[: d.ndof])) < 1e-12


@pytest.mark.parametrize("nb", [6, 7])
def test_config5_unstructured_high_order_through_the_product(cuda, nb):
    """BASELINE config 5 as far as the reference defines it: meshes/unstructured_square read by the product's
    loader, refined by the product (r = 1: 476 quads >= 256 partitions), "p = 6" (n_basis 6 and, degree reading,
    7) on the matrix-core kernels, omega = 16 pi, 256 subdomains from the product's partitioner: fused Helmholtz apply,
    stiffness and weighted mass against the oracle, EnsembleSpace tables identical to the oracle's."""
    import torch

    import cuddhelmholtz_amd as cd
    from conftest import GOLDEN

    pm = cd.Mesh2D.load(GOLDEN / "unstructured_square").refined(1)
    assert pm.n_elem() == 476
    om = oracle.Mesh(pm.vertices(), pm.elements())
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    assert fem.size() == d.ndof
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(500 + nb)
    a2, ax = 0.5 + rng.random(d.ndof), 0.5 + rng.random(ofs.size)
    omega = 16 * math.pi
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    assert A.fused() and A.kernel().startswith(f"helm_mfma_kernel<{nb},")
    xh = rng.standard_normal(2 * d.ndof)
    x = to_dev(torch, xh, cuda)
    y = torch.empty_like(x)
    A.action(x, y)
    S, M = oracle.Stiffness(d), oracle.Mass(d, a2)
    ref = oracle.helmholtz_apply(d, S, M, oracle.FaceMass(ofs, ax), ofs, omega, xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    yr = torch.empty(d.ndof, dtype=torch.float64, device=cuda)
    Sp, Mp = cd.StiffnessMatrix(fem), cd.MassMatrix(fem, to_dev(torch, a2, cuda))
    Sp.action(x[: d.ndof], yr)
    assert rel(yr.cpu().numpy(), S.apply(xh[: d.ndof])) < 1e-12 and Sp.kernel().startswith(f"op_mfma_kernel<{nb},")
    Mp.action(x[: d.ndof], yr)
    assert rel(yr.cpu().numpy(), M.apply(xh[: d.ndof])) < 1e-12
    # 256 subdomains
    labels = pm.partition(256)
    E = cd.EnsembleSpace(fem, 256, labels)
    ref_e = oracle.ensemble(om, d.I, 256, labels)
    for name, want in [("gI", ref_e.gI), ("sI", ref_e.sI), ("fI", ref_e.fI), ("pI", ref_e.pI), ("cmap", ref_e.cmap), ("faces", ref_e.faces)]:
        assert np.array_equal(E.array(name), want), name


@pytest.mark.parametrize("nx,nb", [(1, 2), (1, 4), (2, 3), (3, 5), (6, 4), (7, 6), (1, 8), (5, 7)])
def test_operators_on_tiny_meshes(cuda, nx, nb):
    """Edge cases of the patch plans: a single element, a single (partly filled) patch, no border dofs at all, an odd
    number of patches (the real-operator kernels take patches in pairs), n_basis 6-8 (matrix-core stiffness plan on 16-element
    batches that are mostly empty here)."""
    import torch

    import cuddhelmholtz_amd as cd

    pm, om = meshes("structured", nx)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    ofs = oracle.FaceSpaceO(d, list(faces))
    rng = np.random.default_rng(1000 * nx + nb)
    a2, ax = 0.5 + rng.random(d.ndof), 0.5 + rng.random(ofs.size)
    xh = rng.standard_normal(2 * d.ndof)
    x = to_dev(torch, xh, cuda)
    omega = 3.0
    A = cd.HelmholtzOperator(omega, to_dev(torch, a2, cuda), to_dev(torch, ax, cuda), fem, fs)
    assert A.fused()  # one-element-per-lane plans up to n_basis 5, matrix-core plans for 6-8
    y = torch.full((2 * d.ndof,), -1.0, dtype=torch.float64, device=cuda)
    A.action(x, y)
    ref = oracle.helmholtz_apply(d, oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(ofs, ax), ofs, omega, xh)
    assert rel(y.cpu().numpy(), ref) < 1e-12
    yS = torch.full((d.ndof,), 2.0, dtype=torch.float64, device=cuda)
    S, M = cd.StiffnessMatrix(fem), cd.MassMatrix(fem, to_dev(torch, a2, cuda))
    S.action(x[: d.ndof], yS)
    refS = oracle.Stiffness(d).apply(xh[: d.ndof])
    assert rel(yS.cpu().numpy(), refS) < 1e-12
    M.action(-2.0, x[: d.ndof], yS)
    assert rel(yS.cpu().numpy(), refS - 2.0 * oracle.Mass(d, a2).apply(xh[: d.ndof])) < 1e-12
