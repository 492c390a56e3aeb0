"""Pins the CPU oracle to the reference's own known-answer tests.

Each test restates one test of the reference's tests/ directory and runs it on
the oracle.  CPU only.
"""
import math

import numpy as np
import pytest

import oracle
from conftest import load_unstructured_square


def legendre(n, x):
    p0, p1 = np.ones_like(x), x.copy()
    if n == 0:
        return p0
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    return p1


def legendre_deriv(n, x):
    # P_n' by the three-term recurrence P_n' = P_{n-2}' + (2n-1) P_{n-1}
    d = [np.zeros_like(x), np.ones_like(x)]
    for k in range(2, n + 1):
        d.append(d[k - 2] + (2 * k - 1) * legendre(k - 1, x))
    return d[n]


def cheb_combo(x, n):
    """tests/quadrature_rule.cpp:4-16: degree-n polynomial with integral 2 over [-1,1]"""
    Tn = lambda k: np.cos(k * np.arccos(x))  # noqa: E731
    return (1.0 - n * n) * Tn(n) + (1.0 - (n - 1.0) ** 2) * Tn(n - 1)


@pytest.mark.parametrize("n", range(1, 16))
def test_gauss_legendre_exactness(n):
    """tests/quadrature_rule.cpp:24-46, tolerance 1e-10"""
    x, w = oracle.gauss_legendre(n)
    assert abs(np.dot(w, cheb_combo(x, 2 * n - 1)) - 2.0) < 1e-10


@pytest.mark.parametrize("n", range(2, 16))
def test_gauss_lobatto_exactness(n):
    """tests/quadrature_rule.cpp:49-71"""
    x, w = oracle.gauss_lobatto(n)
    assert abs(np.dot(w, cheb_combo(x, 2 * n - 3)) - 2.0) < 1e-10


def test_quadrature_matches_published_values():
    """Values the survey obtained by running the reference's own QuadratureRule/Basis sources."""
    x, w = oracle.gauss_legendre(5)
    assert abs(x[0] - (-0.90617984593866396)) < 1e-15
    x, w = oracle.gauss_legendre(12)
    assert abs(w.sum() - 2.0) < 1e-13
    _, D = oracle.basis_tables(5, oracle.gauss_lobatto(5)[0])
    assert abs(D[0, 0] - (-5.0)) < 1e-13


@pytest.mark.parametrize("n", range(2, 15))
def test_basis_reproduces_legendre(n):
    """tests/basis.cpp:45-113: D and P applied to nodal P_{n-1} at 10 uniform points, 1e-10"""
    nodes, _ = oracle.gauss_lobatto(n)
    y = legendre(n - 1, nodes)
    x = -1.0 + 2.0 * np.arange(10) / 9.0
    P, D = oracle.basis_tables(n, x)
    assert np.max(np.abs(D @ y - legendre_deriv(n - 1, x))) < 1e-10
    assert np.max(np.abs(P @ y - legendre(n - 1, x))) < 1e-10


def _meshes():
    yield "structured", oracle.Mesh.uniform_rect(10, -1.0, 1.0, 10, -1.0, 1.0)
    xy, elems = load_unstructured_square()
    yield "unstructured", oracle.Mesh(xy, elems)


@pytest.mark.parametrize("p", [3, 4, 5, 6, 7, 8])
def test_mass_forward_and_inverse(p):
    """tests/mass.cpp:13-111: M f == (f, phi) and diag-preconditioned GMRES(5) recovers f, 1e-8"""
    for name, mesh in _meshes():
        d = oracle.Discretization(mesh, p)
        f = d.nodal(oracle.mass_poly)
        b = oracle.linear_functional(d, oracle.mass_poly, nq=p + 2)
        M = oracle.Mass(d)
        Mf = M.apply(f)
        assert np.linalg.norm(Mf - b) / np.linalg.norm(b) < 1e-8, name
        pinv = oracle.diag_inv_mass(d)
        u, info = oracle.gmres(lambda v: pinv * M.apply(v), pinv * b, m=5, maxit=10, tol=1e-12)
        assert np.linalg.norm(u - f) / np.linalg.norm(f) < 1e-8, name


@pytest.mark.parametrize("p", [6, 7, 8])
def test_stiffness_against_laplacian(p):
    """tests/stiffness.cpp:28-98: S f == (-Laplace f, phi), 1e-6"""
    for name, mesh in _meshes():
        d = oracle.Discretization(mesh, p)
        f = d.nodal(oracle.stiff_func)
        Lf = oracle.linear_functional(d, oracle.stiff_neg_laplacian, nq=p + 2)
        Af = oracle.Stiffness(d, nq=p + 2).apply(f)
        assert np.linalg.norm(Af - Lf) / np.linalg.norm(Lf) < 1e-6, name


def test_gmres_toeplitz():
    """tests/gmres.cpp:41-77: 1024x1024 nonsymmetric tridiagonal Toeplitz (1, -3, 1.5), GMRES(5), 1e-10"""
    n = 1 << 10
    rng = np.random.default_rng(0)
    xs = rng.random(n)

    def A(x):
        y = -3.0 * x
        y[1:] += 1.0 * x[:-1]
        y[:-1] += 1.5 * x[1:]
        return y

    b = A(xs)
    x, info = oracle.gmres(A, b, m=5, maxit=100, tol=1e-10)
    assert info["success"]
    assert np.linalg.norm(A(x) - b) / np.linalg.norm(b) < 1e-10


@pytest.mark.parametrize("nx,nb,expect", [(8, 4, (4, 52, 208, 48, 4)), (16, 4, (16, 312, 1248, 48, 36)), (8, 8, (16, 360, 1440, 56, 36))])
def test_ddh_structural_known_answers(nx, nb, expect):
    """(subdomains, n_shared, DDH::size(), mx_fdof, orphan slots), derived by replaying
    source/Mesh2D.cpp:60-115, source/EnsembleSpace.cpp:73-286 and source/DDH.cpp:425-440."""
    mesh = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    d = oracle.Discretization(mesh, nb)
    assert all(e.delta == 1 for e in mesh.edges)  # uniform_rect never produces delta < 0
    t = oracle.DDH(d, nx, nx, 2 * math.pi * nx / 10, np.ones(d.ndof)).t
    mx_dof = {4: 169, 8: 225}[nb]
    assert (t.n_domains, t.n_lambda // 2, 2 * t.n_lambda, t.mx_fdof, t.orphan_slots) == expect
    assert t.mx_dof == mx_dof
