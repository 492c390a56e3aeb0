"""CPU oracle for the CuDDHelmholtz hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (see oracle/oracle.c).  numpy glue over liboracle.so (oracle.c) and
over numbering.py; a numpy restatement of GMRES follows source/gmres.cpp.

Pinned by the reference's known-answer tests: quadrature, basis, mass,
stiffness, GMRES.  PARITY UNPINNED by the reference for DDH / EnsembleSpace /
FaceMass / FaceSpace (the reference ships no test or fixture for them).
"""
from __future__ import annotations

import ctypes as C
import math
import subprocess
from pathlib import Path

import numpy as np

from . import numbering  # noqa: F401
from .numbering import Mesh, ddh_tables, ensemble, facespace, h1_numbering  # noqa: F401

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "_build" / "liboracle.so"


def build(force: bool = False) -> Path:
    src_m = max((_DIR / n).stat().st_mtime for n in ("oracle.c", "ddh_body.inc", "Makefile"))
    if force or not _LIB.exists() or _LIB.stat().st_mtime < src_m:
        subprocess.run(["make", "-C", str(_DIR)], check=True, capture_output=True)
    return _LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["F_CONTIGUOUS"] or a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return int(lib().orc_num_threads())


# ------------------------------------------------------------------ 1-D tables
def gauss_legendre(n):
    x, w = np.zeros(n), np.zeros(n)
    lib().orc_gauss_legendre(C.c_int(n), _p(x), _p(w))
    return x, w


def gauss_lobatto(n):
    x, w = np.zeros(n), np.zeros(n)
    lib().orc_gauss_lobatto(C.c_int(n), _p(x), _p(w))
    return x, w


def basis_tables(nb: int, x):
    """P, D of shape (m, nb): values / derivatives of the GLL Lagrange basis at x."""
    nodes, _ = gauss_lobatto(nb)
    x = np.ascontiguousarray(x, dtype=np.float64)
    P = np.zeros((len(x), nb), order="F")
    D = np.zeros((len(x), nb), order="F")
    lib().orc_basis_tables(C.c_int(nb), _p(nodes), C.c_int(len(x)), _p(x), _p(P), _p(D))
    return P, D


# ------------------------------------------------------------------ discretisation bundle
class Discretization:
    """Mesh + H1 numbering + 1-D tables, everything the operator restatements need."""

    def __init__(self, mesh: Mesh, nb: int):
        self.mesh, self.nb = mesh, nb
        self.I, self.ndof = h1_numbering(mesh, nb)
        self.gll_x, self.gll_w = gauss_lobatto(nb)
        self.corners = mesh.corners()
        self._metrics = {}

    def metrics(self, q):
        """J (2,2,n,n,nel), detJ (n,n,nel), x (2,n,n,nel) on the tensor grid of the 1-D points q."""
        key = tuple(np.round(q, 15))
        if key not in self._metrics:
            n, nel = len(q), self.mesh.n_elem
            J = np.zeros((2, 2, n, n, nel), order="F")
            detJ = np.zeros((n, n, nel), order="F")
            xq = np.zeros((2, n, n, nel), order="F")
            qq = np.ascontiguousarray(q, dtype=np.float64)
            lib().orc_element_metrics(C.c_int(nel), _p(self.corners), C.c_int(n), _p(qq), _p(J), _p(detJ), _p(xq))
            self._metrics[key] = (J, detJ, xq)
        return self._metrics[key]

    def coordinates(self) -> np.ndarray:
        """(2, ndof) collocation points, last-writer-wins like source/H1Space.cpp:108-126."""
        _, _, xq = self.metrics(self.gll_x)
        xy = np.zeros((2, self.ndof), order="F")
        If = self.I.reshape(-1, order="F")
        xf = xq.reshape(2, -1, order="F")
        xy[:, If] = xf  # numpy assigns in order, so later duplicates win
        return xy

    def nodal(self, f) -> np.ndarray:
        xy = self.coordinates()
        return f(xy[0], xy[1])


class Stiffness:
    """source/StiffnessMatrix.cpp:40-81,186-211"""

    def __init__(self, disc: Discretization, nq: int = 0):
        self.d = disc
        self.nq = nq or disc.nb + 1
        self.q, self.w = gauss_legendre(self.nq)
        self.P, self.D = basis_tables(disc.nb, self.q)
        J, _, _ = disc.metrics(self.q)
        self.G = np.zeros((3, self.nq, self.nq, disc.mesh.n_elem), order="F")
        lib().orc_stiffness_setup(C.c_int(disc.mesh.n_elem), C.c_int(self.nq), _p(self.w), _p(J), _p(self.G))

    def apply(self, x, y=None, c=1.0):
        d = self.d
        if y is None:
            y = np.zeros(d.ndof)
        lib().orc_stiffness_apply(C.c_int(d.mesh.n_elem), C.c_int(self.nq), C.c_int(d.nb), _p(self.P), _p(self.D), _p(self.G), _p(d.I),
                                  C.c_double(c), _p(np.ascontiguousarray(x)), _p(y))
        return y


class Mass:
    """source/MassMatrix.cpp:69-135,213-239"""

    def __init__(self, disc: Discretization, coef=None):
        self.d = disc
        nb = disc.nb
        self.nq = (nb + 1) if coef is None else (1 + 3 * nb // 2 + 1)
        self.q, self.w = gauss_legendre(self.nq)
        self.P, _ = basis_tables(nb, self.q)
        _, detJ, _ = disc.metrics(self.q)
        self.a = np.zeros((self.nq, self.nq, disc.mesh.n_elem), order="F")
        lib().orc_mass_setup(C.c_int(disc.mesh.n_elem), C.c_int(self.nq), C.c_int(nb), _p(None if coef is None else np.ascontiguousarray(coef)),
                             _p(detJ), _p(self.w), _p(disc.I), _p(self.P), _p(self.a))

    def apply(self, x, y=None, c=1.0):
        d = self.d
        if y is None:
            y = np.zeros(d.ndof)
        lib().orc_mass_apply(C.c_int(d.mesh.n_elem), C.c_int(self.nq), C.c_int(d.nb), _p(d.I), _p(self.P), _p(self.a), C.c_double(c),
                             _p(np.ascontiguousarray(x)), _p(y))
        return y


def diag_inv_mass(disc: Discretization, coef=None) -> np.ndarray:
    _, detJ, _ = disc.metrics(disc.gll_x)
    op = np.zeros(disc.ndof)
    lib().orc_diag_mass(C.c_int(disc.ndof), C.c_int(disc.mesh.n_elem), C.c_int(disc.nb), _p(None if coef is None else np.ascontiguousarray(coef)),
                        _p(detJ), _p(disc.gll_w), _p(disc.I), _p(op))
    return op


class FaceSpaceO:
    def __init__(self, disc: Discretization, faces):
        self.d = disc
        self.faces = list(faces)
        self.fI, self.proj = facespace(disc.mesh, disc.I, self.faces)
        self.size = len(self.proj)

    def edge_measures(self, nq):
        out = np.zeros((nq, len(self.faces)), order="F")
        for f, eid in enumerate(self.faces):
            out[:, f] = self.d.mesh.edges[eid].length / 2  # include/Edge.hpp:112,134-137
        return out


class FaceMass:
    """source/FaceMassMatrix.cpp:51-139,195-223"""

    def __init__(self, fs: FaceSpaceO, coef=None):
        self.fs = fs
        nb = fs.d.nb
        self.nq = (nb + 1) if coef is None else (1 + 3 * nb // 2 + 1)
        self.q, self.w = gauss_legendre(self.nq)
        self.P, _ = basis_tables(nb, self.q)
        detJ = fs.edge_measures(self.nq)
        self.a = np.zeros((self.nq, len(fs.faces)), order="F")
        lib().orc_facemass_setup(C.c_int(len(fs.faces)), C.c_int(nb), C.c_int(self.nq), _p(self.w), _p(self.P), _p(detJ),
                                 _p(None if coef is None else np.ascontiguousarray(coef)), _p(fs.fI), _p(self.a))

    def apply(self, x, y=None, c=1.0):
        fs = self.fs
        if y is None:
            y = np.zeros(fs.size)
        lib().orc_facemass_apply(C.c_int(len(fs.faces)), C.c_int(fs.d.nb), C.c_int(self.nq), _p(self.P), _p(self.a), _p(fs.fI), C.c_double(c),
                                 _p(np.ascontiguousarray(x)), _p(y))
        return y


def diag_inv_facemass(fs: FaceSpaceO, coef=None) -> np.ndarray:
    nb = fs.d.nb
    detJ = fs.edge_measures(nb)
    op = np.zeros(fs.size)
    lib().orc_diag_facemass(C.c_int(fs.size), C.c_int(len(fs.faces)), C.c_int(nb), _p(fs.d.gll_w), _p(detJ),
                            _p(None if coef is None else np.ascontiguousarray(coef)), _p(fs.fI), _p(op))
    return op


def linear_functional(disc: Discretization, f, nq: int = 0, c: float = 1.0) -> np.ndarray:
    """include/LinearFunctional.hpp:45-142: (f, phi_i); nq == 0 is the collocated GLL form."""
    F = np.zeros(disc.ndof)
    nel, nb = disc.mesh.n_elem, disc.nb
    if nq == 0:
        _, detJ, xq = disc.metrics(disc.gll_x)
        fv = np.asfortranarray(f(xq[0], xq[1]))
        lib().orc_lf_collocated(C.c_int(nel), C.c_int(nb), _p(disc.gll_w), _p(detJ), _p(fv), _p(disc.I), C.c_double(c), _p(F))
    else:
        q, w = gauss_legendre(nq)
        P, _ = basis_tables(nb, q)
        _, detJ, xq = disc.metrics(q)
        fv = np.asfortranarray(f(xq[0], xq[1]))
        lib().orc_lf_quadrature(C.c_int(nel), C.c_int(nq), C.c_int(nb), _p(w), _p(P), _p(detJ), _p(fv), _p(disc.I), C.c_double(c), _p(F))
    return F


def face_linear_functional(fs: FaceSpaceO, f, c: float = 1.0) -> np.ndarray:
    """include/FaceLinearFunctional.hpp:111-124 (collocated form): F[I(k,e)] += c w_k detJ f(x_k), x on the edge
    from its first to its second node (include/Edge.hpp:143-149)."""
    d = fs.d
    F = np.zeros(fs.size)
    for e, eid in enumerate(fs.faces):
        edge = d.mesh.edges[eid]
        x0, x1 = d.mesh.xy[edge.nodes[0]], d.mesh.xy[edge.nodes[1]]
        for k in range(d.nb):
            t = 0.5 * (d.gll_x[k] + 1.0)
            xk = x0 + (x1 - x0) * t
            F[fs.fI[k, e]] += c * d.gll_w[k] * (edge.length / 2) * float(f(np.float64(xk[0]), np.float64(xk[1])))
    return F


def helmholtz_apply(disc, S: Stiffness, M: Mass, H: FaceMass, fs: FaceSpaceO, omega: float, x: np.ndarray) -> np.ndarray:
    """examples/Helmholtz.hpp:28-56"""
    n = disc.ndof
    u, v = x[:n], x[n:]
    Au = S.apply(u)
    Av = S.apply(v)
    M.apply(u, Au, -omega * omega)
    M.apply(v, Av, -omega * omega)
    yf = H.apply(v[fs.proj], None, -omega)
    np.add.at(Au, fs.proj, yf)
    yf = H.apply(u[fs.proj], None, omega)
    np.add.at(Av, fs.proj, yf)
    return np.concatenate([Au, -Av])


# ------------------------------------------------------------------ DDH
class DDH:
    """source/DDH.cpp:323-695 in precision `real` (np.float32 = the reference's, np.float64 = parity mode)."""

    def __init__(self, disc: Discretization, nx: int, ny: int, omega: float, h_a: np.ndarray, real=np.float32):
        self.d, self.real = disc, real
        nb = disc.nb
        _, Dm = basis_tables(nb, disc.gll_x)
        J, detJ, _ = disc.metrics(disc.gll_x)
        self.t = ddh_tables(disc.mesh, disc.I, disc.ndof, nx, ny, omega, np.asarray(h_a, dtype=np.float64), disc.gll_x, disc.gll_w, Dm, detJ, real)
        t = self.t
        self.G = np.zeros((3, nb * nb * t.mx_elems, t.n_domains), dtype=real, order="F")
        fn = lib().orc_ddh_geom_f32 if real == np.float32 else lib().orc_ddh_geom_f64
        fn(C.c_int(t.n_domains), C.c_int(t.mx_elems), C.c_int(nb), _p(np.ascontiguousarray(t.s_elems, dtype=np.int32)),
           _p(np.asfortranarray(t.elems, dtype=np.int32)), _p(disc.gll_w), _p(J), _p(self.G))
        self.size = 2 * t.n_lambda

    def solve(self, x=None, want_y=False, lam=None, want_update=True, d0=0, d1=None):
        t, d = self.t, self.d
        real = self.real
        d1 = t.n_domains if d1 is None else d1
        y = np.zeros(2 * d.ndof) if want_y else None
        upd = np.zeros(2 * t.n_lambda, dtype=real) if want_update else None
        fn = lib().orc_ddh_apply_f32 if real == np.float32 else lib().orc_ddh_apply_f64
        fn(C.c_int(d.ndof), C.c_int(t.n_domains), C.c_int(t.n_lambda), C.c_int(t.nb), C.c_int(t.mx_elems), C.c_int(t.mx_dof), C.c_int(t.mx_fdof),
           C.c_int(t.nt), C.c_double(t.omega), C.c_double(t.dt), _p(np.ascontiguousarray(t.s_dof, dtype=np.int32)),
           _p(np.ascontiguousarray(t.s_fdof, dtype=np.int32)), _p(t.B), _p(t.gI), _p(t.sI), _p(t.D), _p(self.G), _p(t.m), _p(t.gmi), _p(t.a), _p(t.H),
           _p(t.wh_filter), _p(t.cs), _p(t.sn), _p(None if x is None else np.ascontiguousarray(x, dtype=np.float64)), _p(y),
           _p(None if lam is None else np.ascontiguousarray(lam, dtype=real)), _p(upd), C.c_int(d0), C.c_int(d1))
        return y, upd

    def rhs(self, f):  # source/DDH.cpp:641-667
        return self.solve(x=f)[1]

    def action(self, lam):  # source/DDH.cpp:611-639
        lam = np.asarray(lam, dtype=self.real)
        upd = self.solve(lam=lam)[1]
        return (self.real(1) * lam + self.real(-1) * upd).astype(self.real)

    def postprocess(self, lam, f):  # source/DDH.cpp:669-695
        return self.solve(x=f, want_y=True, lam=lam, want_update=False)[0]


def ddh_set_wh_iters(n: int = 5):
    """WaveHoltz iterations of every later DDH solve (5 = the reference, source/DDH.cpp:136)."""
    lib().orc_ddh_set_wh_iters(C.c_int(n))


# ------------------------------------------------------------------ GMRES
def gmres(A, b, x0=None, m=20, maxit=100, tol=1e-6, dtype=np.float64, allreduce=None):
    """source/gmres.cpp:91-235 with numpy vectors of `dtype`; A(x) -> A x.  Returns (x, info).
    allreduce (tests of the multi-process host logic only): sums a scalar over the ranks when the vectors are
    partitioned; every inner product then goes through it."""
    T = dtype
    n = len(b)
    b = np.asarray(b, dtype=T)
    x = np.zeros(n, dtype=T) if x0 is None else np.array(x0, dtype=T)
    one = T(1)
    red = (lambda v: v) if allreduce is None else allreduce

    def nrm(v):
        return T(math.sqrt(float(red(np.dot(v, v)))))

    bnrm = nrm(b)
    m1 = m + 1
    V = np.zeros((n, m1), dtype=T, order="F")
    H = np.zeros((m1, m), dtype=T, order="F")
    sn, cs, eta = np.zeros(m, dtype=T), np.zeros(m, dtype=T), np.zeros(m1, dtype=T)
    info = dict(success=False, num_matvec=0, res_norm=[], num_iter=0)

    r = b - np.asarray(A(x), dtype=T)
    info["num_matvec"] += 1
    r_nrm = nrm(r)
    info["res_norm"].append(float(r_nrm))
    if r_nrm < T(tol) * bnrm:
        info["success"] = True
        return x, info
    it = 1
    while it < maxit:
        V[:, 0] = (one / r_nrm) * r
        eta[:] = 0
        eta[0] = r_nrm
        k1 = 0
        for k in range(m):
            k1 = k + 1
            w = np.asarray(A(V[:, k]), dtype=T)
            info["num_matvec"] += 1
            for j in range(k1):
                H[j, k] = T(red(np.dot(w, V[:, j])))
                w = w - H[j, k] * V[:, j]
            H[k1, k] = nrm(w)
            if H[k1, k] == 0:
                break
            V[:, k1] = w * (one / H[k1, k])
            # Givens (source/gmres.cpp:7-23)
            h = H[:, k]
            for i in range(k):
                h1, h2 = h[i], h[i + 1]
                h[i] = cs[i] * h1 + sn[i] * h2
                h[i + 1] = -sn[i] * h1 + cs[i] * h2
            t = T(math.hypot(float(h[k]), float(h[k + 1])))
            cs[k] = h[k] / t
            sn[k] = h[k + 1] / t
            h[k] = cs[k] * h[k] + sn[k] * h[k + 1]
            h[k + 1] = 0
            eta[k1] = -sn[k] * eta[k]
            eta[k] = cs[k] * eta[k]
            if abs(eta[k1]) < T(tol) * bnrm:
                break
        yk = eta[:k1].copy()
        for i in range(k1 - 1, -1, -1):  # ?trsv('U','N','N')
            s = yk[i]
            for j in range(i + 1, k1):
                s -= H[i, j] * yk[j]
            yk[i] = s / H[i, i]
        for k in range(k1):
            x = x + yk[k] * V[:, k]
        r = b - np.asarray(A(x), dtype=T)
        info["num_matvec"] += 1
        r_nrm = nrm(r)
        info["res_norm"].append(float(r_nrm))
        if r_nrm < T(tol) * bnrm:
            info["success"] = True
            break
        it += 1
    info["num_iter"] = it
    return x, info


# ------------------------------------------------------------------ integrands shared with cuddh_capi.h ids
def gaussians(omega):
    s = omega * omega

    def f(x, y):
        return s / math.pi * np.exp(-s * ((x + 0.5) ** 2 + y * y)) + s / math.pi * np.exp(-s * ((x - 0.5) ** 2 + (y + 0.5) ** 2))

    return f


def alpha_disk(x, y):
    return np.where(x * x + y * y < 0.0625, 0.2, 1.0)


def mass_poly(x, y):
    return 3.0 * x * x - 2.0 * x * y + y + 1.0


def stiff_func(x, y):
    return (x**5 - 5.0 * x) * (y**3 - 3.0 * y)


def stiff_neg_laplacian(x, y):
    return -6.0 * y * (x**5 - 5 * x) - 20.0 * x**3 * (y**3 - 3.0 * y)
