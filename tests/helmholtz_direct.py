"""Independent PDE-level references for DDH on uniform_rect meshes (TEST INFRASTRUCTURE).

Nothing here follows source/DDH.cpp's loops or index tables.  The global Helmholtz system, the subdomain
systems and the WaveHoltz local solve are written as dense / sparse linear algebra on a lexicographic node
grid built from the mesh geometry alone, and tied to the product's / oracle's dof numbering only through the
nodes' physical coordinates.  The 1-D ingredients (GLL nodes and weights, differentiation matrix) come from the
oracle's tables, which the reference's own t_quadrature_rule / t_basis pin (tests/test_oracle_pins.py).

What DDH converges to (derived from source/DDH.cpp:201-234,296-319, see DESIGN.md "DDH against the PDE"):
every subdomain s solves, with its own lumped mass m_s and the lumped face mass H_s of ALL its sides,

    S_s U - w^2 a^2 m_s U + i w a H_s U = f|_s + H_s L_s,        U = u + i v,  L = lambda + i mu,

(a Robin problem, d_n U + i a w U = L) and sends  L' = -L + 2 i a w U  to its neighbour.  At the fixed point the
traces agree and the assembled equations are the global collocated Helmholtz system with the load of a dof
multiplied by the number of subdomains holding it: DDH::rhs / postprocess read the GLOBAL load entry
x[gI] in every subdomain (source/DDH.cpp:208-212), not a share of it.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle


class UniformGrid:
    """Lexicographic node grid of uniform_rect(nx, x0, x0 + nx h, ny, y0, y0 + ny h) (square elements of side h) with
    n_basis nb, and its map to a dof numbering (through the nodes' coordinates only)."""

    def __init__(self, nx: int, ny: int, nb: int, coords: np.ndarray, x0=-1.0, y0=-1.0, h=None):
        self.nx, self.ny, self.nb = nx, ny, nb
        self.p = nb - 1
        self.n1x, self.n1y = nx * self.p + 1, ny * self.p + 1
        self.h = 2.0 / nx if h is None else h
        self.x1, self.w1 = oracle.gauss_lobatto(nb)
        _, self.D1 = oracle.basis_tables(nb, self.x1)
        self.gx = self._line(nx, x0)
        self.gy = self._line(ny, y0)
        ix = self._locate(self.gx, coords[0])
        iy = self._locate(self.gy, coords[1])
        self.dof_of = np.full((self.n1x, self.n1y), -1, dtype=np.int64)
        self.dof_of[ix, iy] = np.arange(coords.shape[1])
        self.ndof = self.n1x * self.n1y
        assert (self.dof_of >= 0).all() and coords.shape[1] == self.ndof

    def _line(self, n, x0):
        g = np.zeros(n * self.p + 1)
        for e in range(n):
            g[e * self.p:(e + 1) * self.p + 1] = x0 + self.h * (e + 0.5 * (self.x1 + 1.0))
        return g

    @staticmethod
    def _locate(g, c):
        j = np.searchsorted(g, c - 1e-9)
        assert np.allclose(g[j], c, atol=1e-9)
        return j

    # ---- 1-D assembled matrices on a run of `ne` elements (ne*p+1 nodes)
    def stiff1(self, ne):
        """sum_e D^T W D * (2/h): 1-D stiffness (GLL collocated)."""
        n = ne * self.p + 1
        K = np.zeros((n, n))
        Ke = self.D1.T @ np.diag(self.w1) @ self.D1 * (2.0 / self.h)
        for e in range(ne):
            s = slice(e * self.p, e * self.p + self.nb)
            K[s, s] += Ke
        return K

    def mass1(self, ne):
        """lumped 1-D mass: w * h/2."""
        m = np.zeros(ne * self.p + 1)
        for e in range(ne):
            m[e * self.p:e * self.p + self.nb] += self.w1 * (self.h / 2)
        return m

    def block(self, nex, ney=None, sparse=False):
        """Collocated operators on a block of nex x ney elements, lexicographic (ix fastest):
        S (dense or CSR), lumped mass m, lumped face mass of the four sides Hs[side] (0: y-, 1: x+, 2: y+, 3: x-)."""
        ney = nex if ney is None else ney
        Kx, mx = self.stiff1(nex), self.mass1(nex)
        Ky, my = self.stiff1(ney), self.mass1(ney)
        nxn, nyn = len(mx), len(my)
        # index = ix + nxn*iy  ->  kron(A_y, A_x)
        if sparse:
            S = (sp.kron(sp.diags(my), sp.csr_matrix(Kx)) + sp.kron(sp.csr_matrix(Ky), sp.diags(mx))).tocsr()
        else:
            S = np.kron(np.diag(my), Kx) + np.kron(Ky, np.diag(mx))
        m = np.kron(my, mx)
        H = [np.zeros((nxn, nyn)) for _ in range(4)]  # [ix, iy]
        H[0][:, 0] = mx
        H[1][nxn - 1, :] = my
        H[2][:, nyn - 1] = mx
        H[3][0, :] = my
        return S, m, [h.reshape(-1, order="F") for h in H]

    def whole(self):
        return self.block(self.nx, self.ny, sparse=True)

    def to_grid(self, v):
        """vector in the caller's dof numbering -> lexicographic grid vector"""
        return np.asarray(v)[self.dof_of.reshape(-1, order="F")]

    def from_grid(self, g):
        out = np.zeros(self.ndof, dtype=g.dtype)
        out[self.dof_of.reshape(-1, order="F")] = g
        return out

    def subdomain_nodes(self, bx, by, ne):
        """lexicographic global grid ids of block (bx, by) of ne x ne elements, block-lexicographic order"""
        n = ne * self.p + 1
        ix = bx * ne * self.p + np.arange(n)
        iy = by * ne * self.p + np.arange(n)
        return (ix[:, None] + self.n1x * iy[None, :]).reshape(-1, order="F")

    def blocks(self, ne):
        return [(bx, by) for by in range(self.ny // ne) for bx in range(self.nx // ne)]

    def multiplicity(self, ne):
        """number of ne x ne-element blocks that hold each grid node"""
        mult = np.zeros(self.ndof)
        for bx, by in self.blocks(ne):
            mult[self.subdomain_nodes(bx, by, ne)] += 1
        return mult

    def on_boundary(self):
        b = np.zeros((self.n1x, self.n1y), dtype=bool)
        b[0, :] = b[-1, :] = b[:, 0] = b[:, -1] = True
        return b.reshape(-1, order="F")


def collocated_system(grid: UniformGrid, omega: float, a_grid: np.ndarray, drop_absorbing_at=None):
    """Global collocated (GLL-lumped) Helmholtz matrix  S - w^2 a^2 m + i w a H_boundary  on the grid numbering."""
    S, m, Hs = grid.whole()
    Hb = Hs[0] + Hs[1] + Hs[2] + Hs[3]
    if drop_absorbing_at is not None:
        Hb = np.where(drop_absorbing_at, 0.0, Hb)
    return (S.astype(np.complex128) + sp.diags(-omega**2 * a_grid**2 * m + 1j * omega * a_grid * Hb)).tocsc()


def ddh_fixed_point_system(grid: UniformGrid, ne: int, omega: float, a_grid, f_grid):
    """(K', f') such that K' U = f' is what DDH with EXACT local solves converges to when the decomposition has no
    interior cross point (strips), derived from the transmission conditions alone:
      * the load of a dof held by k subdomains counts k times (every subdomain reads the global load entry);
      * where an interface meets the physical boundary the node's face mass H = H_interface + H_physical multiplies
        both the impedance term and the incoming trace, so the exchange L_s + L_s' = 2 i a w U cancels ALL of it: the
        absorbing term of that node drops out of the assembled equation (the node sees a Neumann condition)."""
    mult = grid.multiplicity(ne)
    junction = (mult > 1) & grid.on_boundary()
    return collocated_system(grid, omega, a_grid, drop_absorbing_at=junction), f_grid * mult


def ddh_fixed_point_solution(grid: UniformGrid, ne: int, omega: float, a_grid, f_grid):
    """What DDH with EXACT local solves converges to on any block decomposition (cross points included), as a vector on
    the grid numbering (complex).  Derivation, from source/DDH.cpp:201-234,296-319 and the slot rule :425-440 only:

    Subdomain s solves  A0_s U_s + i w a H_s U_s = f|_s + H_s L_s  (A0 = S - w^2 a^2 m_s; H_s the lumped mass of ALL its
    sides) and sends  L' = -L + 2 i a w U.  A face node of s has ONE read slot and ONE write slot, both taken from the LAST
    shared pair (s, s') of the connectivity map that contains it.
      * If s and s' name each other (every interface node that is not a cross point): at the fixed point U_s = U_s' and
        L_s + L_s' = 2 i a w U, so adding the two equations cancels the impedance terms with the WHOLE H of the node
        (also its physical-boundary part where an interface ends on the boundary) and leaves  (A0_s + A0_s') U = 2 f.
      * At an interior cross point four subdomains meet and each keeps one pair only.  Pairs are numbered in global edge
        order (elements row by row, sides bottom / right / top / left, source/Mesh2D.cpp:16-17,70-115), so the last pair of
        the SW subdomain is with the NW one, of the SE one with the NE one, and the NW and NE subdomains name each other.
        NW and NE therefore share one value there; SW and SE read slots nobody writes (always 0): each keeps its OWN value
        of the node with the absorbing condition  A0_s U + i w a H_s U = f.
      * Nodes on the physical boundary that are on no interface have no slot: L = 0, plain absorbing condition.
    Every subdomain adds the GLOBAL load entry of its nodes, so a class of k merged copies has the load k f.
    The returned vector is the partition-of-unity average  sum_s (m_s / m_global) U_s  of DDH::postprocess (:298-307)."""
    S, m, Hs = grid.block(ne)
    Hall = Hs[0] + Hs[1] + Hs[2] + Hs[3]
    _, m_glob, _ = grid.whole()
    n = ne * grid.p + 1
    nbx, nby = grid.nx // ne, grid.ny // ne
    li = np.arange(n * n) % n
    lj = np.arange(n * n) // n
    n_unknowns = grid.ndof
    rows, cols, vals = [], [], []
    blocks = []
    for bx, by in grid.blocks(ne):
        ids = grid.subdomain_nodes(bx, by, ne).copy()
        W, E, So, No = bx > 0, bx < nbx - 1, by > 0, by < nby - 1  # which sides are interfaces
        on_iface = (W & (li == 0)) | (E & (li == n - 1)) | (So & (lj == 0)) | (No & (lj == n - 1))
        paired = on_iface.copy()
        # interior cross points where this block is the SW (its NE corner) or the SE (its NW corner) subdomain
        for corner, cond in (((n - 1) + n * (n - 1), E and No), (0 + n * (n - 1), W and No)):
            if cond:
                paired[corner] = False
                ids[corner] = n_unknowns  # its own copy of the node
                n_unknowns += 1
        blocks.append((bx, by, ids, paired))
    f_rhs = np.zeros(n_unknowns, dtype=np.complex128)
    for bx, by, ids, paired in blocks:
        g = grid.subdomain_nodes(bx, by, ne)
        a = a_grid[g]
        A = S.astype(np.complex128)
        A[np.diag_indices(n * n)] += -omega**2 * a * a * m + np.where(paired, 0.0, 1j * omega * a * Hall)
        r, c = np.nonzero(A)
        rows.append(ids[r])
        cols.append(ids[c])
        vals.append(A[r, c])
        np.add.at(f_rhs, ids, f_grid[g])
    K = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n_unknowns, n_unknowns))
    U = solve_complex(K, f_rhs)
    out = np.zeros(grid.ndof, dtype=np.complex128)
    for bx, by, ids, _ in blocks:
        g = grid.subdomain_nodes(bx, by, ne)
        out[g] += (m / m_glob[g]) * U[ids]
    return out


def consistent_system(disc, omega: float, h_a: np.ndarray):
    """The operator of examples/Helmholtz.hpp:28-56 assembled column by column from the reference-pinned oracle
    operators (consistent mass with a^2, face mass with a, Gauss-Legendre rules), in the dof numbering of `disc`."""
    n = disc.ndof
    S, M = oracle.Stiffness(disc), oracle.Mass(disc, h_a**2)
    fs = oracle.FaceSpaceO(disc, disc.mesh.boundary_edges)
    H = oracle.FaceMass(fs, h_a[fs.proj])
    rows, cols, vals = [], [], []
    e = np.zeros(n)
    for j in range(n):
        e[j] = 1.0
        col = (S.apply(e) - omega**2 * M.apply(e)).astype(np.complex128)
        hf = H.apply(e[fs.proj])
        np.add.at(col, fs.proj, 1j * omega * hf)
        e[j] = 0.0
        nz = np.nonzero(col)[0]
        rows.append(nz)
        cols.append(np.full(len(nz), j))
        vals.append(col[nz])
    return sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def solve_complex(K, f_complex):
    if sp.issparse(K):
        return spla.splu(K.tocsc()).solve(f_complex)
    return np.linalg.solve(K, f_complex)


def as_uv(U):
    return np.concatenate([U.real, U.imag])


# ---------------------------------------------------------------- WaveHoltz local solve, continuous in time
def waveholtz_continuous(S, inv_m, Ha, omega, F, G, n_iter=5, rtol=1e-11):
    """The local solve of source/DDH.cpp:237-296 with the time stepping replaced by its ODE limit,

        p' = -q,    q' = inv_m (S p - Ha q - cos(wt) F + sin(wt) G),
        (u, v) <- (2/T) int_0^T (cos(wt) - 1/4) (p, q) dt   from  (p, q)(0) = (u, v),   T = 2 pi / w,

    integrated with DOP853 to `rtol` (the reference: explicit midpoint rule with the trapezoid rule, both O(dt^2)).
    Returns (u, v / w)."""
    from scipy.integrate import solve_ivp

    n = len(F)
    T = 2 * math.pi / omega

    def rhs(t, y):
        p, q = y[:n], y[n:2 * n]
        c, s = math.cos(omega * t), math.sin(omega * t)
        k = (2.0 / T) * (c - 0.25)
        return np.concatenate([-q, inv_m * (S @ p - Ha * q - c * F + s * G), k * p, k * q])

    u, v = np.zeros(n), np.zeros(n)
    for _ in range(n_iter):
        y0 = np.concatenate([u, v, np.zeros(2 * n)])
        sol = solve_ivp(rhs, (0.0, T), y0, method="DOP853", rtol=rtol, atol=1e-14)
        assert sol.success
        u, v = sol.y[2 * n:3 * n, -1], sol.y[3 * n:, -1]
    return u, v / omega


def local_solves_continuous(grid: UniformGrid, ne: int, omega: float, a_grid, f_grid, g_grid, n_iter=5):
    """sum_s R_s^T (m_s / m_global) WaveHoltz_s(f|_s, g|_s) with zero incoming traces: what
    DDH::postprocess(lambda = 0, [f; g]) computes (source/DDH.cpp:298-307), on the grid numbering."""
    S, m, Hs = grid.block(ne)
    Hall = Hs[0] + Hs[1] + Hs[2] + Hs[3]
    _, m_glob, _ = grid.whole()
    u_out, v_out = np.zeros(grid.ndof), np.zeros(grid.ndof)
    for bx, by in grid.blocks(ne):
        ids = grid.subdomain_nodes(bx, by, ne)
        a = a_grid[ids]
        u, v = waveholtz_continuous(S, 1.0 / (a * a * m), Hall * a, omega, f_grid[ids], g_grid[ids], n_iter)
        w = m / m_glob[ids]
        u_out[ids] += w * u
        v_out[ids] += w * v
    return u_out, v_out
