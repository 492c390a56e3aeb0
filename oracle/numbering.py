"""CPU restatement of the reference's host-side index-map construction.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c header).  Pure Python / numpy with
dictionaries where the reference uses unordered_map; written to follow the
reference line by line, not to be fast -- use it on meshes of at most a few
thousand elements.  Citations are relative to arotem3/CuDDHelmholtz.

Pinning: the structural known-answers of SURVEY.md 8c (subdomain counts,
n_shared, DDH::size(), mx_fdof, orphan slots) are asserted in
tests/test_oracle_pins.py.  The reference has no tests of EnsembleSpace / DDH,
so beyond those counts PARITY IS UNPINNED by the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

EMAP1 = (0, 1, 3, 0)  # source/Mesh2D.cpp:16
EMAP2 = (1, 2, 2, 3)  # source/Mesh2D.cpp:17


@dataclass
class Edge:
    id: int
    boundary: bool
    nodes: tuple
    elements: list
    sides: list
    delta: int
    length: float


class Mesh:
    """source/Mesh2D.cpp:11-136 (from_vertices) and :138-171 (uniform_rect)."""

    def __init__(self, xy: np.ndarray, elems: np.ndarray):
        self.xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)  # (n_pts, 2)
        self.elems = np.asarray(elems, dtype=np.int64).reshape(-1, 4)
        self.n_elem = len(self.elems)
        self.n_nodes = len(self.xy)
        self.node_elems = [[] for _ in range(self.n_nodes)]  # (corner, element) per node, in element order
        for el, c in enumerate(self.elems):
            for i in range(4):
                self.node_elems[c[i]].append((i, el))
        self.edges: list[Edge] = []
        edge_map = {}
        for el, c in enumerate(self.elems):
            for s in range(4):
                C0, C1 = int(c[EMAP1[s]]), int(c[EMAP2[s]])
                key = (min(C0, C1), max(C0, C1))
                if key not in edge_map:
                    length = float(np.hypot(*(self.xy[C1] - self.xy[C0])))
                    edge_map[key] = len(self.edges)
                    self.edges.append(Edge(len(self.edges), True, (C0, C1), [el, -1], [s, -1], 1, length))
                else:
                    e = self.edges[edge_map[key]]
                    e0, s0 = e.elements[0], e.sides[0]
                    n1 = int(self.elems[e0][EMAP1[s0]])
                    e.elements[1] = el
                    e.sides[1] = s
                    e.boundary = False
                    e.delta = 1 if C0 == n1 else -1
        self.boundary_edges = [e.id for e in self.edges if e.boundary]
        self.interior_edges = [e.id for e in self.edges if not e.boundary]

    @staticmethod
    def uniform_rect(nx, ax, bx, ny, ay, by) -> "Mesh":
        dx, dy = (bx - ax) / nx, (by - ay) / ny
        xy = np.zeros(((nx + 1) * (ny + 1), 2))
        for j in range(ny + 1):
            for i in range(nx + 1):
                xy[i + (nx + 1) * j] = (ax + dx * i, ay + dy * j)
        elems = np.zeros((nx * ny, 4), dtype=np.int64)
        lid = lambda i, j: i + (nx + 1) * j  # noqa: E731
        for j in range(ny):
            for i in range(nx):
                elems[i + nx * j] = (lid(i, j), lid(i + 1, j), lid(i + 1, j + 1), lid(i, j + 1))
        return Mesh(xy, elems)

    def corners(self) -> np.ndarray:
        """(2, 4, n_elem) F-order corner coordinates."""
        out = np.zeros((2, 4, self.n_elem), order="F")
        for el, c in enumerate(self.elems):
            out[:, :, el] = self.xy[c].T
        return out

    def min_h(self) -> float:
        return min(e.length for e in self.edges)


def e2v(nc, i, f, el):
    """source/H1Space.cpp:27-33"""
    m = i if f in (0, 2) else (nc - 1 if f == 1 else 0)
    n = i if f in (1, 3) else (nc - 1 if f == 2 else 0)
    return m + nc * (n + nc * el)


def n2v(nc, c, el):
    """source/H1Space.cpp:36-42"""
    m = 0 if c in (0, 3) else nc - 1
    n = 0 if c in (0, 1) else nc - 1
    return m + nc * (n + nc * el)


def h1_numbering(mesh: Mesh, nb: int):
    """source/H1Space.cpp:11-106.  Returns (I (nb,nb,n_elem) F-order int32, ndof)."""
    mask = {}
    if nb > 2:
        for eid in mesh.interior_edges:
            e = mesh.edges[eid]
            rev = e.delta < 0
            for i in range(1, nb - 1):
                j = nb - 1 - i if rev else i
                mask[e2v(nb, j, e.sides[1], e.elements[1])] = e2v(nb, i, e.sides[0], e.elements[0])
    for k in range(mesh.n_nodes):
        conn = mesh.node_elems[k]
        if not conn:
            continue
        v0 = n2v(nb, conn[0][0], conn[0][1])
        for c, el in conn[1:]:
            mask[n2v(nb, c, el)] = v0
    Nn = mesh.n_elem * nb * nb
    I = np.zeros(Nn, dtype=np.int32)
    l = 0
    for i in range(Nn):
        if i not in mask:
            I[i] = l
            l += 1
    for v1, v0 in mask.items():
        I[v1] = I[v0]
    return I.reshape((nb, nb, mesh.n_elem), order="F"), l


def facespace(mesh: Mesh, I: np.ndarray, faces):
    """source/H1Space.cpp:129-187.  Returns (fI (nb, nf), proj (fdof))."""
    nb = I.shape[0]
    K = I.reshape(-1, order="F")
    mask, P = {}, []
    fI = np.zeros((nb, len(faces)), dtype=np.int32, order="F")
    for f, eid in enumerate(faces):
        e = mesh.edges[eid]
        for i in range(nb):
            idx = int(K[e2v(nb, i, e.sides[0], e.elements[0])])
            if idx not in mask:
                mask[idx] = len(P)
                P.append(idx)
            fI[i, f] = mask[idx]
    return fI, np.asarray(P, dtype=np.int32)


@dataclass
class Ensemble:
    n_spaces: int
    mx_elems: int
    mx_faces: int
    mx_ndof: int
    mx_fdof: int
    s_elems: np.ndarray
    elems: np.ndarray
    s_faces: np.ndarray
    faces: np.ndarray
    s_dof: np.ndarray
    sI: np.ndarray
    gI: np.ndarray
    s_fdof: np.ndarray
    fI: np.ndarray
    pI: np.ndarray
    cmap: np.ndarray


def ensemble(mesh: Mesh, I: np.ndarray, n_spaces: int, labels) -> Ensemble:
    """source/EnsembleSpace.cpp:11-287."""
    nb = I.shape[0]
    nel = mesh.n_elem
    E = [[] for _ in range(n_spaces)]
    el2s = np.zeros(nel, dtype=np.int64)
    for el in range(nel):
        p = int(labels[el])
        assert 0 <= p < n_spaces
        E[p].append(el)
        el2s[el] = len(E[p]) - 1
    assert min(len(e) for e in E) >= 1
    mx_elems = max(len(e) for e in E)
    s_elems = np.array([len(e) for e in E], dtype=np.int32)
    elems = -np.ones((mx_elems, n_spaces), dtype=np.int32, order="F")
    for p in range(n_spaces):
        elems[: len(E[p]), p] = E[p]

    F = [[] for _ in range(n_spaces)]
    shared_faces = []
    for e in mesh.edges:
        S0 = int(labels[e.elements[0]])
        if e.boundary:
            F[S0].append((e.id, 0))
        else:
            S1 = int(labels[e.elements[1]])
            if S0 != S1:
                F[S0].append((e.id, 0))
                F[S1].append((e.id, 1))
                shared_faces.append((S0, S1, len(F[S0]) - 1, len(F[S1]) - 1))
    mx_faces = max(len(f) for f in F)
    s_faces = np.array([len(f) for f in F], dtype=np.int32)
    faces = -np.ones((mx_faces, n_spaces), dtype=np.int32, order="F")
    face_side = -np.ones((mx_faces, n_spaces), dtype=np.int32, order="F")
    for p in range(n_spaces):
        for i, (f, side) in enumerate(F[p]):
            faces[i, p] = f
            face_side[i, p] = side

    sI = -np.ones((nb, nb, mx_elems, n_spaces), dtype=np.int32, order="F")
    s2g = [[] for _ in range(n_spaces)]
    s_dof = np.zeros(n_spaces, dtype=np.int32)
    for p in range(n_spaces):
        uniq = {}
        for el in range(s_elems[p]):
            g_el = elems[el, p]
            for j in range(nb):
                for i in range(nb):
                    g = int(I[i, j, g_el])
                    if g not in uniq:
                        uniq[g] = len(s2g[p])
                        s2g[p].append(g)
                    sI[i, j, el, p] = uniq[g]
        s_dof[p] = len(uniq)
    mx_ndof = int(s_dof.max())
    gI = -np.ones((mx_ndof, n_spaces), dtype=np.int32, order="F")
    for p in range(n_spaces):
        gI[: s_dof[p], p] = s2g[p]

    fI = -np.ones((nb, mx_faces, n_spaces), dtype=np.int32, order="F")
    f2s = [[] for _ in range(n_spaces)]
    s_fdof = np.zeros(n_spaces, dtype=np.int32)
    for p in range(n_spaces):
        uniq = {}
        for f in range(s_faces[p]):
            e = mesh.edges[faces[f, p]]
            side = face_side[f, p]
            g_el, s = e.elements[side], e.sides[side]
            rev = side == 1 and e.delta < 0
            for i in range(nb):
                j = nb - 1 - i if rev else i
                m = j if s in (0, 2) else (nb - 1 if s == 1 else 0)
                n = j if s in (1, 3) else (nb - 1 if s == 2 else 0)
                idx = int(sI[m, n, el2s[g_el], p])
                if idx not in uniq:
                    uniq[idx] = len(f2s[p])
                    f2s[p].append(idx)
                fI[i, f, p] = uniq[idx]
        s_fdof[p] = len(uniq)
    mx_fdof = int(s_fdof.max())
    pI = -np.ones((mx_fdof, n_spaces), dtype=np.int32, order="F")
    for p in range(n_spaces):
        pI[: s_fdof[p], p] = f2s[p]

    shared_dofs = []
    unique_shared = {}
    for S0, S1, f0, f1 in shared_faces:
        key = (S0, S1) if S0 < S1 else (S1, S0)
        unq = unique_shared.setdefault(key, set())
        for i in range(nb):
            j0, j1 = int(fI[i, f0, S0]), int(fI[i, f1, S1])
            lkey = j0 if S0 < S1 else j1
            if lkey not in unq:
                shared_dofs.append((S0, S1, j0, j1))
                unq.add(lkey)
    cmap = np.array(shared_dofs, dtype=np.int32).reshape(-1, 4).T.copy(order="F")
    return Ensemble(n_spaces, mx_elems, mx_faces, mx_ndof, mx_fdof, s_elems, elems, s_faces, faces, s_dof, sI, gI, s_fdof, fI, pI, cmap)


@dataclass
class DdhTables:
    n_domains: int
    n_lambda: int
    nt: int
    dt: float
    omega: float
    nb: int
    nel1d: int
    mx_dof: int
    mx_fdof: int
    mx_elems: int
    s_dof: np.ndarray
    s_fdof: np.ndarray
    s_elems: np.ndarray
    elems: np.ndarray
    B: np.ndarray
    gI: np.ndarray
    sI: np.ndarray
    D: np.ndarray
    m: np.ndarray
    gmi: np.ndarray
    a: np.ndarray
    H: np.ndarray
    wh_filter: np.ndarray
    cs: np.ndarray
    sn: np.ndarray
    ens: Ensemble
    orphan_slots: int


def ddh_tables(mesh: Mesh, I: np.ndarray, ndof: int, nx: int, ny: int, omega: float, h_a: np.ndarray, gll_x, gll_w, Dmat, detJ, real=np.float32) -> DdhTables:
    """source/DDH.cpp:323-609 (everything except the device-side geometric factors).
    Dmat: (nb, nb) derivative matrix at the GLL nodes; detJ: (nb, nb, n_elem) at the GLL nodes."""
    nb = I.shape[0]
    epd = max(1, 16 // nb)  # DDH_BLOCK_SIZE / n_basis (the reference only allows nb in {4, 8})
    assert nx % epd == 0 and ny % epd == 0
    ndx, ndy = nx // epd, ny // epd
    n_domains = ndx * ndy
    labels = np.zeros(nx * ny, dtype=np.int64)
    for j in range(ny):
        for i in range(nx):
            labels[i + nx * j] = (i // epd) + ndx * (j // epd)
    ens = ensemble(mesh, I, n_domains, labels)

    T = 2 * math.pi / omega
    h = mesh.min_h()
    dt = 0.2 * 0.5 * h / (nb * nb)
    nt = int(math.ceil(T / dt))
    dt = T / nt
    filt = np.zeros(nt + 1, dtype=real)
    for k in range(nt + 1):
        filt[k] = dt * (omega / math.pi) * (math.cos(omega * k * dt) - 0.25)
    filt[0] = real(filt[0] * 0.5)
    filt[nt] = real(filt[nt] * 0.5)
    cs = np.zeros(2 * nt + 1, dtype=real)
    sn = np.zeros(2 * nt + 1, dtype=real)
    for k in range(2 * nt + 1):
        t = 0.5 * k * dt
        cs[k] = -math.cos(omega * t)
        sn[k] = math.sin(omega * t)

    mx_dof, mx_fdof, mx_el = int(ens.s_dof.max()), int(ens.s_fdof.max()), int(ens.s_elems.max())
    n_shared = ens.cmap.shape[1]
    n_lambda = 2 * n_shared
    B = -np.ones((mx_fdof, 2, n_domains), dtype=np.int32, order="F")
    for k in range(n_shared):
        S0, S1, j0, j1 = (int(v) for v in ens.cmap[:, k])
        B[j0, 0, S0] = k
        B[j0, 1, S0] = n_shared + k
        B[j1, 0, S1] = n_shared + k
        B[j1, 1, S1] = k
    used = set(int(v) for v in B.reshape(-1) if v >= 0)
    orphan = 2 * (n_lambda - len(used))  # entries of the size-2*n_lambda trace vector never read nor written

    gI = -np.ones((mx_dof, n_domains), dtype=np.int32, order="F")
    sI = -np.ones((nb, nb, mx_el, n_domains), dtype=np.int32, order="F")
    for s in range(n_domains):
        nd, nf = int(ens.s_dof[s]), int(ens.s_fdof[s])
        perm = -np.ones(nd, dtype=np.int64)
        inv = -np.ones(nd, dtype=np.int64)
        pp = set()
        l = 0
        while l < nf:
            j = int(ens.pI[l, s])
            pp.add(j)
            perm[l] = j
            l += 1
        for i in range(nd):
            if i in pp:
                continue
            perm[l] = i
            l += 1
        for i in range(nd):
            inv[perm[i]] = i
        for i in range(nd):
            gI[i, s] = ens.gI[perm[i], s]
        for el in range(ens.s_elems[s]):
            for ll in range(nb):
                for k in range(nb):
                    sI[k, ll, el, s] = inv[ens.sI[k, ll, el, s]]

    mi = np.zeros(ndof)
    for el in range(mesh.n_elem):
        for j in range(nb):
            for i in range(nb):
                mi[I[i, j, el]] += gll_w[i] * gll_w[j] * detJ[i, j, el]
    mi = 1.0 / mi

    m = np.zeros((mx_dof, n_domains), dtype=real, order="F")
    H = np.zeros((mx_fdof, n_domains), dtype=real, order="F")
    A = np.zeros((mx_dof, n_domains), dtype=real, order="F")
    gmi = np.zeros((mx_dof, n_domains), dtype=real, order="F")
    for s in range(n_domains):
        for el in range(ens.s_elems[s]):
            g_el = ens.elems[el, s]
            for j in range(nb):
                for i in range(nb):
                    l = sI[i, j, el, s]
                    m[l, s] = real(float(m[l, s]) + gll_w[i] * gll_w[j] * detJ[i, j, g_el])
        for i in range(ens.s_dof[s]):
            A[i, s] = h_a[gI[i, s]]
            gmi[i, s] = mi[gI[i, s]]
        for f in range(ens.s_faces[s]):
            e = mesh.edges[ens.faces[f, s]]
            for i in range(nb):
                l = ens.fI[i, f, s]
                H[l, s] = real(float(H[l, s]) + (e.length / 2) * gll_w[i])

    return DdhTables(n_domains, n_lambda, nt, dt, float(omega), nb, epd, mx_dof, mx_fdof, mx_el, ens.s_dof, ens.s_fdof, ens.s_elems, ens.elems,
                     B, gI, sI, np.asarray(Dmat, dtype=real, order="F"), m, gmi, A, H, filt, cs, sn, ens, orphan)
