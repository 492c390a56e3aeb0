"""Host-side logic of the product (C++ behind the C handle layer) against the oracle's
line-by-line restatement of the reference: mesh topology, H1 numbering, FaceSpace,
EnsembleSpace maps and the DDH constructor tables.  Integer maps must be identical.  CPU only."""
import math

import numpy as np
import pytest

import cuddhelmholtz_amd as cd
import oracle
from conftest import load_unstructured_square


def product_mesh(kind, nx=10):
    if kind == "structured":
        return cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    xy, elems = load_unstructured_square()
    return cd.Mesh2D.from_vertices(xy, elems), oracle.Mesh(xy, elems)


@pytest.mark.parametrize("kind", ["structured", "unstructured"])
def test_mesh_topology(kind):
    pm, om = product_mesh(kind)
    assert pm.n_elem() == om.n_elem and pm.n_nodes() == om.n_nodes and pm.n_edges() == len(om.edges)
    E = pm.edges()
    for e in om.edges:
        row = E[e.id]
        assert row[0] == int(e.boundary)
        assert (row[1], row[2]) == e.nodes
        assert row[3] == e.elements[0] and row[5] == e.sides[0]
        if not e.boundary:
            assert row[4] == e.elements[1] and row[6] == e.sides[1] and row[7] == e.delta
    assert list(pm.boundary_edges()) == om.boundary_edges
    assert abs(pm.min_h() - om.min_h()) < 1e-15
    if kind == "unstructured":
        assert any(e.delta < 0 for e in om.edges)  # the fixture exercises reversed edges


@pytest.mark.parametrize("kind", ["structured", "unstructured"])
@pytest.mark.parametrize("nb", [2, 3, 4, 5, 8])
def test_h1_numbering_and_facespace(kind, nb):
    pm, om = product_mesh(kind)
    basis = cd.Basis(nb)
    fem = cd.H1Space(pm, basis)
    I, ndof = oracle.h1_numbering(om, nb)
    assert fem.size() == ndof
    if kind == "structured":
        assert ndof == (10 * (nb - 1) + 1) ** 2
    assert np.array_equal(fem.global_indices(), I)
    d = oracle.Discretization(om, nb)
    assert np.allclose(fem.physical_coordinates(), d.coordinates(), atol=1e-14)

    faces = pm.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    fI, proj = oracle.facespace(om, I, list(faces))
    assert fs.size() == len(proj)
    assert np.array_equal(fs.subspace_indices(), fI)
    assert np.array_equal(fs.global_indices(), proj)


@pytest.mark.parametrize("n", list(range(1, 16)))
def test_quadrature_and_basis_tables(n):
    x, w = cd.quadrature(n, "legendre")
    xo, wo = oracle.gauss_legendre(n)
    assert np.allclose(x, xo, atol=2e-16 * 8) and np.allclose(w, wo, rtol=1e-13)
    if n >= 2:
        x, w = cd.quadrature(n, "lobatto")
        xo, wo = oracle.gauss_lobatto(n)
        assert np.allclose(x, xo, atol=2e-15) and np.allclose(w, wo, rtol=1e-13)
        pts = np.concatenate([oracle.gauss_legendre(n + 1)[0], xo])
        P, D = oracle.basis_tables(n, pts)
        b = cd.Basis(n)
        assert np.allclose(b.eval(pts), P, atol=1e-13)
        assert np.allclose(b.deriv(pts), D, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("kind,n_spaces", [("structured", 4), ("structured", 7), ("unstructured", 5)])
def test_ensemble_space_arbitrary_labels(kind, n_spaces):
    pm, om = product_mesh(kind)
    nb = 4
    fem = cd.H1Space(pm, cd.Basis(nb))
    rng = np.random.default_rng(7)
    labels = rng.integers(0, n_spaces, om.n_elem)
    labels[:n_spaces] = np.arange(n_spaces)  # no empty subspace
    ens = cd.EnsembleSpace(fem, n_spaces, labels)
    I, _ = oracle.h1_numbering(om, nb)
    o = oracle.ensemble(om, I, n_spaces, labels)
    assert list(ens.dims) == [n_spaces, o.mx_elems, o.mx_faces, o.mx_ndof, o.mx_fdof, o.cmap.shape[1]]
    for name, ref in [("gI", o.gI), ("sizes", o.s_dof), ("elements", o.elems), ("n_elems", o.s_elems), ("faces", o.faces),
                      ("n_faces", o.s_faces), ("sI", o.sI), ("fI", o.fI), ("pI", o.pI), ("fsizes", o.s_fdof), ("cmap", o.cmap)]:
        assert np.array_equal(ens.array(name), ref), name


@pytest.mark.parametrize("nx,nb", [(8, 4), (16, 4), (8, 8), (10, 3), (9, 5)])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_ddh_constructor_tables(nx, nb, precision):
    omega = 2 * math.pi * nx / 10
    pm = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    om = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(pm, cd.Basis(nb))
    d = oracle.Discretization(om, nb)
    rng = np.random.default_rng(3)
    h_a = 0.5 + rng.random(d.ndof)
    real = np.float32 if precision == "f32" else np.float64
    t = oracle.DDH(d, nx, nx, omega, h_a, real).t
    F = cd.DDH(omega, h_a, fem, nx, nx, precision=precision)
    info = F.info()
    assert F.size() == 2 * t.n_lambda
    assert (info["n_domains"], info["nt"], info["n_lambda"], info["mx_dof"], info["mx_fdof"], info["nel1d"]) == (
        t.n_domains, t.nt, t.n_lambda, t.mx_dof, t.mx_fdof, t.nel1d)
    assert abs(info["dt"] - t.dt) < 1e-18
    for name, ref in [("B", t.B), ("gI", t.gI), ("sI", t.sI)]:
        assert np.array_equal(F.table(name), ref.reshape(-1, order="F")), name
    tol = dict(rtol=2e-7, atol=0) if precision == "f32" else dict(rtol=1e-13, atol=0)
    for name, ref in [("D", t.D), ("m", t.m), ("gmi", t.gmi), ("a", t.a), ("H", t.H), ("filter", t.wh_filter), ("cs", t.cs), ("sn", t.sn)]:
        got = F.table(name)
        assert got.dtype == real
        ref = ref.reshape(-1, order="F")
        scale = np.abs(ref).max()  # D has entries that are zero up to rounding
        assert np.allclose(got, ref, rtol=tol["rtol"], atol=tol["rtol"] * scale), name


def test_ddh_known_answers_large():
    """SURVEY.md 8c for the shipped example size (128^2, n_basis 4): 1024 subdomains, n_shared 25,792,
    DDH::size() 103,168, mx_fdof 48, mx_dof 169, 3,844 orphan trace entries."""
    nx, nb = 128, 4
    pm = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(pm, cd.Basis(nb))
    assert fem.size() == (nx * 3 + 1) ** 2
    F = cd.DDH(2 * math.pi * 12.8, np.ones(fem.size()), fem, nx, nx)
    info = F.info()
    assert (info["n_domains"], info["n_lambda"] // 2, F.size(), info["mx_fdof"], info["mx_dof"]) == (1024, 25792, 103168, 48, 169)
    B = F.table("B")
    used = np.unique(B[B >= 0])
    assert 2 * (info["n_lambda"] - len(used)) == 3844
    assert info["nt"] == 800  # dt = 0.1 h / nb^2, T = 2 pi / omega  (source/DDH.cpp:363-368)


def test_ddh_rejects_bad_block_size():
    pm = cd.Mesh2D.uniform_rect(6, -1.0, 1.0, 6, -1.0, 1.0)
    fem = cd.H1Space(pm, cd.Basis(4))
    with pytest.raises(RuntimeError):
        cd.DDH(5.0, np.ones(fem.size()), fem, 6, 6)  # 6 is not a multiple of 16 / 4


def test_refine_quads_conserves_area_and_conformity(unstructured_square):
    """cuddhelmholtz_amd.meshtools.refine_quads (host utility, not in the reference): every quad -> 4, counter-clockwise,
    conforming (the oracle's mesh builder finds exactly two elements per interior edge), same domain area."""
    from cuddhelmholtz_amd.meshtools import refine_quads

    xy, elems = unstructured_square

    def areas(p, e):
        q = p[e]
        x, y = q[..., 0], q[..., 1]
        return 0.5 * sum(x[:, i] * y[:, (i + 1) % 4] - x[:, (i + 1) % 4] * y[:, i] for i in range(4))

    a0 = areas(xy, elems)
    for times in (1, 2):
        x2, e2 = refine_quads(xy, elems, times)
        assert len(e2) == len(elems) * 4**times
        a2 = areas(x2, e2)
        assert a2.min() > 0 and abs(a2.sum() - a0.sum()) < 1e-13
        # the four children of a straight-sided quad tile it: their areas sum to the parent's
        assert np.allclose(a2.reshape(len(elems), -1).sum(axis=1), a0, rtol=0, atol=1e-14)
        m = oracle.Mesh(x2, e2)
        assert len(m.boundary_edges) == 2**times * 40  # the fixture's boundary has 40 edges
        n_edges = len(m.edges)
        assert 4 * len(e2) == 2 * (n_edges - len(m.boundary_edges)) + len(m.boundary_edges)
        # Euler: V - E + F = 1 for a simply connected planar mesh (outer face not counted)
        assert len(x2) - n_edges + len(e2) == 1
