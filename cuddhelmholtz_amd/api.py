"""Python mirror of the reference's operator interface for the hot path.

Same names and argument meaning as the C++ classes of cuddh.hpp (Mesh2D, Basis,
H1Space, FaceSpace, EnsembleSpace, StiffnessMatrix, MassMatrix, FaceMassMatrix,
DDH, gmres); every method forwards to libcuddh_amd.so.  Device vectors are
torch CUDA tensors (torch is only the allocator / stream provider); host arrays
are numpy.  Column-major (Fortran) shapes throughout, as in the reference.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _native as N

lib = N.lib


def _ptr(t, dtype=None, numel=None, what="tensor"):
    """Device pointer of a torch tensor (None -> NULL).  dtype: "f64" / "f32" / a torch dtype the entry point reads or
    writes; numel: the least number of elements it touches.  A tensor of another type or a shorter one would otherwise become
    an out-of-bounds device access behind the C ABI instead of a Python error."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError(f"{what}: expected a CUDA tensor")
    if not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous tensor")
    if dtype is not None:
        import torch

        want = {"f64": torch.float64, "f32": torch.float32, "i32": torch.int32}.get(dtype, dtype)
        if t.dtype != want:
            raise ValueError(f"{what}: expected dtype {want}, got {t.dtype}")
    if numel is not None and t.numel() < numel:
        raise ValueError(f"{what}: expected at least {numel} elements, got {t.numel()}")
    return C.c_void_p(t.data_ptr())


def _h(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def use_torch_stream() -> None:
    """Make the library launch on torch's current CUDA stream."""
    import torch

    lib.cuddh_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream))


def device_count() -> int:
    return int(lib.cuddh_hip_device_count())


def quadrature(n: int, kind: str = "lobatto"):
    x = np.empty(n)
    w = np.empty(n)
    N.check_capi(lib.cuddh_quadrature(n, 0 if kind == "legendre" else 1, _h(x), _h(w)), "quadrature")
    return x, w


class Basis:
    def __init__(self, n: int):
        self.n = n
        self._h = N.handle(lib.cuddh_basis_create(n), "Basis")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_basis_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def size(self) -> int:
        return self.n

    def eval(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        P = np.empty((len(x), self.n), order="F")
        N.check_capi(lib.cuddh_basis_eval(self._h, len(x), _h(x), _h(P)), "Basis.eval")
        return P

    def deriv(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        D = np.empty((len(x), self.n), order="F")
        N.check_capi(lib.cuddh_basis_deriv(self._h, len(x), _h(x), _h(D)), "Basis.deriv")
        return D


class Mesh2D:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_mesh_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    @staticmethod
    def uniform_rect(nx, ax, bx, ny, ay, by) -> "Mesh2D":
        return Mesh2D(N.handle(lib.cuddh_mesh_uniform_rect(nx, ax, bx, ny, ay, by), "Mesh2D.uniform_rect"))

    @staticmethod
    def from_vertices(xy: np.ndarray, elems: np.ndarray) -> "Mesh2D":
        """xy: (2, n_pts) F-order or (n_pts, 2) C-order; elems: (n_elem, 4) C-order vertex ids."""
        xy = np.ascontiguousarray(xy, dtype=np.float64)
        elems = np.ascontiguousarray(elems, dtype=np.int32)
        return Mesh2D(N.handle(lib.cuddh_mesh_from_vertices(xy.size // 2, _h(xy), elems.size // 4, _h(elems)), "Mesh2D.from_vertices"))

    @staticmethod
    def load(directory) -> "Mesh2D":
        """the reference's text format: info.txt, coordinates.txt, elements.txt (tests/load_unstructured_square.cpp:11-55)"""
        return Mesh2D(N.handle(lib.cuddh_mesh_load(str(directory).encode()), "Mesh2D.load"))

    def refined(self, times: int = 1) -> "Mesh2D":
        """uniform refinement: every quadrilateral -> 4 (edge midpoints + centroid), `times` rounds"""
        return Mesh2D(N.handle(lib.cuddh_mesh_refined(self._h, int(times)), "Mesh2D.refined"))

    def partition(self, n_parts: int) -> np.ndarray:
        """element labels in [0, n_parts) for EnsembleSpace: compact, equally sized sets (centroid Morton order)"""
        out = np.empty(self.n_elem(), dtype=np.int32)
        N.check_capi(lib.cuddh_mesh_partition(self._h, int(n_parts), _h(out)), "Mesh2D.partition")
        return out

    def n_elem(self):
        return lib.cuddh_mesh_n_elem(self._h)

    def n_edges(self):
        return lib.cuddh_mesh_n_edges(self._h)

    def n_nodes(self):
        return lib.cuddh_mesh_n_nodes(self._h)

    def min_h(self):
        return lib.cuddh_mesh_min_h(self._h)

    def boundary_edges(self) -> np.ndarray:
        out = np.empty(lib.cuddh_mesh_n_boundary_edges(self._h), dtype=np.int32)
        N.check_capi(lib.cuddh_mesh_boundary_edges(self._h, _h(out)))
        return out

    def vertices(self) -> np.ndarray:
        """(n_nodes, 2) vertex coordinates"""
        out = np.empty((self.n_nodes(), 2))
        N.check_capi(lib.cuddh_mesh_vertices(self._h, _h(out)))
        return out

    def elements(self) -> np.ndarray:
        """(n_elem, 4) corner vertex ids, counter-clockwise"""
        out = np.empty((self.n_elem(), 4), dtype=np.int32)
        N.check_capi(lib.cuddh_mesh_elements(self._h, _h(out)))
        return out

    def edges(self) -> np.ndarray:
        """(n_edges, 8): type(1=boundary), node0, node1, elem0, elem1, side0, side1, delta"""
        out = np.empty((self.n_edges(), 8), dtype=np.int32)
        N.check_capi(lib.cuddh_mesh_edges(self._h, _h(out)))
        return out


class H1Space:
    def __init__(self, mesh: Mesh2D, basis: Basis):
        self.mesh, self.basis = mesh, basis
        self._h = N.handle(lib.cuddh_h1space_create(mesh._h, basis._h), "H1Space")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_h1space_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def size(self) -> int:
        return lib.cuddh_h1space_size(self._h)

    def global_indices(self) -> np.ndarray:
        nb = self.basis.n
        I = np.empty((nb, nb, self.mesh.n_elem()), dtype=np.int32, order="F")
        N.check_capi(lib.cuddh_h1space_global_indices(self._h, _h(I)))
        return I

    def physical_coordinates(self) -> np.ndarray:
        X = np.empty((2, self.size()), order="F")
        N.check_capi(lib.cuddh_h1space_coordinates(self._h, _h(X)))
        return X


class FaceSpace:
    def __init__(self, fem: H1Space, faces):
        self.fem = fem
        self.faces = np.ascontiguousarray(faces, dtype=np.int32)
        self._h = N.handle(lib.cuddh_facespace_create(fem._h, len(self.faces), _h(self.faces)), "FaceSpace")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_facespace_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def size(self):
        return lib.cuddh_facespace_size(self._h)

    def n_faces(self):
        return len(self.faces)

    def subspace_indices(self) -> np.ndarray:
        I = np.empty((self.fem.basis.n, len(self.faces)), dtype=np.int32, order="F")
        N.check_capi(lib.cuddh_facespace_subspace_indices(self._h, _h(I)))
        return I

    def global_indices(self) -> np.ndarray:
        p = np.empty(self.size(), dtype=np.int32)
        N.check_capi(lib.cuddh_facespace_global_indices(self._h, _h(p)))
        return p

    def restrict(self, x, y):
        N.check_capi(lib.cuddh_facespace_restrict(self._h, _ptr(x, "f64", self.fem.size(), "x"), _ptr(y, "f64", self.size(), "y")), "restrict")

    def prolong(self, x, y):
        N.check_capi(lib.cuddh_facespace_prolong(self._h, _ptr(x, "f64", self.size(), "x"), _ptr(y, "f64", self.fem.size(), "y")), "prolong")

    def orth(self, x):
        N.check_capi(lib.cuddh_facespace_orth(self._h, _ptr(x, "f64", self.fem.size(), "x")), "orth")


class EnsembleSpace:
    _SHAPES = {
        "gI": lambda d, nb: (d[3], d[0]),
        "sizes": lambda d, nb: (d[0],),
        "elements": lambda d, nb: (d[1], d[0]),
        "n_elems": lambda d, nb: (d[0],),
        "faces": lambda d, nb: (d[2], d[0]),
        "n_faces": lambda d, nb: (d[0],),
        "sI": lambda d, nb: (nb, nb, d[1], d[0]),
        "fI": lambda d, nb: (nb, d[2], d[0]),
        "pI": lambda d, nb: (d[4], d[0]),
        "fsizes": lambda d, nb: (d[0],),
        "cmap": lambda d, nb: (4, d[5]),
    }

    def __init__(self, fem: H1Space, n_spaces: int, labels):
        self.fem = fem
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        self._h = N.handle(lib.cuddh_ensemble_create(fem._h, n_spaces, _h(labels)), "EnsembleSpace")
        d = np.empty(6, dtype=np.int32)
        N.check_capi(lib.cuddh_ensemble_dims(self._h, _h(d)))
        self.dims = d

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_ensemble_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def array(self, name: str) -> np.ndarray:
        shape = self._SHAPES[name](self.dims, self.fem.basis.n)
        out = np.empty(shape, dtype=np.int32, order="F")
        if out.size:
            N.check_capi(lib.cuddh_ensemble_array(self._h, name.encode(), _h(out)))
        return out


class _Operator:
    """y = A x (`action(x, y)`) and y += c A x (`action(c, x, y)`) on device vectors."""

    def __init__(self, handle, keepalive=(), n=None):
        self._h = handle
        self._keep = keepalive
        self._n = n  # length of the vectors action() reads and writes (None: not checked)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_operator_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def kernel(self) -> str:
        """kernel instantiation action() launches, e.g. 'helm_lane_kernel<4,5,8,NT=1,UG=0> pe=64' (single operators build
        their plan on the first action(): 'generic' before that and when no specialised kernel exists)"""
        buf = C.create_string_buffer(160)
        N.check_capi(lib.cuddh_operator_kernel_name(self._h, buf, 160), "kernel")
        return buf.value.decode()

    def action(self, *args):
        if len(args) == 2:
            x, y = args
            N.check_capi(lib.cuddh_operator_apply(self._h, _ptr(x, "f64", self._n, "x"), _ptr(y, "f64", self._n, "y")), "action")
        else:
            c, x, y = args
            N.check_capi(lib.cuddh_operator_apply_add(self._h, float(c), _ptr(x, "f64", self._n, "x"), _ptr(y, "f64", self._n, "y")), "action")


class StiffnessMatrix(_Operator):
    def __init__(self, fem: H1Space, nq: int = 0):
        super().__init__(N.handle(lib.cuddh_stiffness_create(fem._h, nq), "StiffnessMatrix"), (fem,), fem.size())


class MassMatrix(_Operator):
    def __init__(self, fem: H1Space, a=None):
        super().__init__(N.handle(lib.cuddh_mass_create(fem._h, _ptr(a, "f64", fem.size(), "a")), "MassMatrix"), (fem, a), fem.size())


class DiagInvMassMatrix(_Operator):
    def __init__(self, fem: H1Space, a=None):
        super().__init__(N.handle(lib.cuddh_diaginv_mass_create(fem._h, _ptr(a, "f64", fem.size(), "a")), "DiagInvMassMatrix"), (fem, a), fem.size())


class FaceMassMatrix(_Operator):
    def __init__(self, fs: FaceSpace, a=None):
        super().__init__(N.handle(lib.cuddh_facemass_create(fs._h, _ptr(a, "f64", fs.size(), "a")), "FaceMassMatrix"), (fs, a), fs.size())


class DiagInvFaceMassMatrix(_Operator):
    def __init__(self, fs: FaceSpace, a=None):
        super().__init__(N.handle(lib.cuddh_diaginv_facemass_create(fs._h, _ptr(a, "f64", fs.size(), "a")), "DiagInvFaceMassMatrix"), (fs, a), fs.size())


class HelmholtzOperator(_Operator):
    """Fused [u;v] -> [Au;Av] of reference examples/Helmholtz.hpp:28-56."""

    def __init__(self, omega: float, a2x, ax, fem: H1Space, fs: FaceSpace):
        super().__init__(N.handle(lib.cuddh_helmholtz_create(float(omega), _ptr(a2x, "f64", fem.size(), "a2x"), _ptr(ax, "f64", fs.size(), "ax"), fem._h, fs._h),
                                  "HelmholtzOperator"), (fem, fs, a2x, ax), 2 * fem.size())

    def action_unfused(self, x, y):
        N.check_capi(lib.cuddh_helmholtz_apply_unfused(self._h, _ptr(x, "f64", self._n, "x"), _ptr(y, "f64", self._n, "y")), "action_unfused")

    def fused(self) -> bool:
        return bool(lib.cuddh_helmholtz_is_fused(self._h))

    def bytes_per_apply(self, actual: bool = False) -> int:
        return int(lib.cuddh_helmholtz_bytes(self._h, 1 if actual else 0))

    # ---- plan-native vector ordering (cuddh_hip.h: cuddh_hip_helmholtz_apply_native)
    def has_native(self) -> bool:
        return bool(lib.cuddh_helmholtz_has_native(self._h))

    def to_native(self, x, z):
        N.check_capi(lib.cuddh_helmholtz_to_native(self._h, _ptr(x, "f64", self._n, "x"), _ptr(z, "f64", self._n, "z")), "HelmholtzOperator.to_native")

    def from_native(self, z, y):
        N.check_capi(lib.cuddh_helmholtz_from_native(self._h, _ptr(z, "f64", self._n, "z"), _ptr(y, "f64", self._n, "y")), "HelmholtzOperator.from_native")

    def action_native(self, z_in, z_out):
        N.check_capi(lib.cuddh_helmholtz_apply_native(self._h, _ptr(z_in, "f64", self._n, "z_in"), _ptr(z_out, "f64", self._n, "z_out")), "HelmholtzOperator.action_native")

    def gmres(self, x, b, m: int, maxit: int, tol: float = 1e-6, verbose: int = 0, max_seconds: float = 6 * 60 * 60) -> "SolverOut":
        """HelmholtzOperator::gmres: x, b in the reference ordering; iteration vectors in plan-native ordering when the plan has one"""
        res = N.SolverResult()
        h_res = np.zeros(maxit + 2)
        h_time = np.zeros(maxit + 2)
        N.check_capi(lib.cuddh_gmres_helmholtz(self._h, _ptr(x, "f64", self._n, "x"), _ptr(b, "f64", self._n, "b"), m, maxit, float(tol), verbose,
                                               float(max_seconds), C.byref(res), _h(h_res), _h(h_time)), "HelmholtzOperator.gmres")
        return _solver_out(res, h_res, h_time)

    def bytes_native(self) -> int:
        return int(lib.cuddh_helmholtz_bytes(self._h, 3))

    def bytes_affine(self) -> int:
        """SURVEY 8d's "affine" figure when the plan reads the stiffness metric from one uniform table (0 otherwise)"""
        return int(lib.cuddh_helmholtz_bytes(self._h, 2))


# integrand ids of cuddh_capi.h
GAUSSIANS, ALPHA_DISK, MASS_POLY, STIFF_NEG_LAPLACIAN, STIFF_FUNC, CONSTANT, ALPHA_DISK_SQ = range(7)


def linear_functional(fem: H1Space, integrand: int, F, param: float = 0.0, nq: int = 0, c: float = 1.0, accumulate: bool = False):
    N.check_capi(lib.cuddh_linear_functional(fem._h, nq, integrand, float(param), float(c), int(accumulate), _ptr(F)), "LinearFunctional")


def face_linear_functional(fs: FaceSpace, integrand: int, F, param: float = 0.0, nq: int = 0, c: float = 1.0, accumulate: bool = False):
    N.check_capi(lib.cuddh_face_linear_functional(fs._h, nq, integrand, float(param), float(c), int(accumulate), _ptr(F)), "FaceLinearFunctional")


def nodal_values(fem: H1Space, integrand: int, out, param: float = 0.0):
    N.check_capi(lib.cuddh_nodal_values(fem._h, integrand, float(param), _ptr(out)), "nodal_values")


@dataclass
class SolverOut:
    success: bool
    num_iter: int
    num_matvec: int
    res_norm: list = field(default_factory=list)
    time: list = field(default_factory=list)


def _solver_out(res: N.SolverResult, h_res, h_time) -> SolverOut:
    n = res.n_res
    return SolverOut(bool(res.success), res.num_iter, res.num_matvec, list(h_res[:n]), list(h_time[:n]))


class DDH:
    """Substructured Helmholtz solver (reference include/DDH.hpp).  precision 'f32' is the
    reference's; 'f64' is the parity mode with double traces."""

    _INT_TABLES = ("B", "gI", "sI")

    def __init__(self, omega: float, h_a: np.ndarray, fem: H1Space, nx: int, ny: int, precision: str = "f32", kernel: int = 0):
        self.fem = fem
        self.f64 = precision == "f64"
        h_a = np.ascontiguousarray(h_a, dtype=np.float64)
        self._h = N.handle(lib.cuddh_ddh_create(float(omega), _h(h_a), fem._h, nx, ny, int(self.f64), kernel), "DDH")
        self.omega = float(omega)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` may already be torn down at interpreter exit
            try:
                lib.cuddh_ddh_destroy(h)
            except Exception:  # noqa: BLE001
                pass

    def size(self) -> int:
        return lib.cuddh_ddh_size(self._h)

    @property
    def trace_dtype(self):
        import torch

        return torch.float64 if self.f64 else torch.float32

    def info(self) -> dict:
        info = np.zeros(8, dtype=np.int32)
        dt = C.c_double()
        N.check_capi(lib.cuddh_ddh_info(self._h, _h(info), C.byref(dt)), "DDH.info")
        keys = ("n_domains", "nt", "n_lambda", "mx_dof", "mx_fdof", "nel1d", "kernel", "is_f64")
        out = dict(zip(keys, map(int, info)))
        out["dt"] = dt.value
        return out

    def set_wh_iters(self, n: int = 5):
        """Verification knob: WaveHoltz iterations per local solve (reference: 5, source/DDH.cpp:136)."""
        N.check_capi(lib.cuddh_ddh_set_wh_iters(self._h, int(n)), "DDH.set_wh_iters")

    def set_wave_priority(self, high: bool):
        """The local solves launched next take issue priority over other resident wavefronts (s_setprio); results unchanged."""
        N.check_capi(lib.cuddh_ddh_set_wave_priority(self._h, 1 if high else 0), "DDH.set_wave_priority")

    def table(self, name: str) -> np.ndarray:
        n = lib.cuddh_ddh_table(self._h, name.encode(), None, 1)
        if n < 0:
            raise RuntimeError(N.last_error())
        dtype = np.int32 if name in self._INT_TABLES else (np.float64 if self.f64 else np.float32)
        out = np.empty(int(n), dtype=dtype)
        if n:
            lib.cuddh_ddh_table(self._h, name.encode(), _h(out), 0)
        return out

    def rhs(self, f, b):
        N.check_capi(lib.cuddh_ddh_rhs(self._h, _ptr(f, "f64", 2 * self.fem.size(), "f"), _ptr(b, self.trace_dtype, self.size(), "b")), "DDH.rhs")

    def postprocess(self, lam, f, u):
        N.check_capi(lib.cuddh_ddh_postprocess(self._h, _ptr(lam, self.trace_dtype, self.size(), "lambda"), _ptr(f, "f64", 2 * self.fem.size(), "f"),
                                               _ptr(u, "f64", 2 * self.fem.size(), "u")), "DDH.postprocess")

    def action(self, x, y):
        N.check_capi(lib.cuddh_ddh_action(self._h, _ptr(x, self.trace_dtype, self.size(), "x"), _ptr(y, self.trace_dtype, self.size(), "y")), "DDH.action")

    def local_traces(self, d0, d1, f, lam, update):
        N.check_capi(lib.cuddh_ddh_local_traces(self._h, d0, d1, _ptr(f, "f64", 2 * self.fem.size(), "f"), _ptr(lam, self.trace_dtype, self.size(), "lambda"),
                                                 _ptr(update, self.trace_dtype, self.size(), "update")), "DDH.local_traces")

    def local_traces_listed(self, domains, f, lam, update):
        """local_traces for the subdomains listed in `domains` (int32 device tensor, distinct ids), one launch"""
        N.check_capi(lib.cuddh_ddh_local_traces_listed(self._h, _ptr(domains, "i32", domains.numel(), "domains"), int(domains.numel()),
                                                        _ptr(f, "f64", 2 * self.fem.size(), "f"), _ptr(lam, self.trace_dtype, self.size(), "lambda"),
                                                        _ptr(update, self.trace_dtype, self.size(), "update")), "DDH.local_traces_listed")

    def local_solution_listed(self, domains, lam, f, u, zero_u=True):
        """local_solution for the subdomains listed in `domains` (int32 device tensor, distinct ids), one launch"""
        N.check_capi(lib.cuddh_ddh_local_solution_listed(self._h, _ptr(domains, "i32", domains.numel(), "domains"), int(domains.numel()),
                                                          _ptr(lam, self.trace_dtype, self.size(), "lambda"), _ptr(f, "f64", 2 * self.fem.size(), "f"),
                                                          _ptr(u, "f64", 2 * self.fem.size(), "u"), int(zero_u)), "DDH.local_solution_listed")

    def local_solution(self, d0, d1, lam, f, u, zero_u=True):
        N.check_capi(lib.cuddh_ddh_local_solution(self._h, d0, d1, _ptr(lam, self.trace_dtype, self.size(), "lambda"), _ptr(f, "f64", 2 * self.fem.size(), "f"),
                                                   _ptr(u, "f64", 2 * self.fem.size(), "u"), int(zero_u)), "DDH.local_solution")


def gmres(n: int, x, A, b, m: int, maxit: int, tol: float = 1e-6, verbose: int = 0, max_seconds: float = 6 * 60 * 60, Precond=None,
          reduce=None) -> SolverOut:
    """Restarted GMRES (reference include/gmres.hpp:33-36).  A: an operator of this module, a DDH,
    or a Python callable `A(x, y)` acting on device tensors of x's dtype.

    reduce (only with a callable A): `reduce(t)` sums the small device tensor `t` over all ranks in place
    (torch.distributed.all_reduce) -- the vectors are then partitioned over the processes, see dist.py."""
    import torch

    if reduce is not None and (isinstance(A, (DDH, _Operator)) or Precond is not None):
        raise ValueError("gmres: reduce= needs a callable operator and no preconditioner")

    res = N.SolverResult()
    h_res = np.zeros(maxit + 2)
    h_time = np.zeros(maxit + 2)
    if isinstance(A, DDH):
        N.check_capi(lib.cuddh_gmres_ddh(n, _ptr(x, A.trace_dtype, n, "x"), A._h, _ptr(b, A.trace_dtype, n, "b"), m, maxit, float(tol), verbose, float(max_seconds), C.byref(res), _h(h_res), _h(h_time)), "gmres")
    elif isinstance(A, _Operator):
        N.check_capi(lib.cuddh_gmres_f64(n, _ptr(x, "f64", n, "x"), A._h, _ptr(b, "f64", n, "b"), Precond._h if Precond is not None else None, m, maxit, float(tol), verbose,
                                         float(max_seconds), C.byref(res), _h(h_res), _h(h_time)), "gmres")
    else:
        dtype = x.dtype
        is64 = dtype == torch.float64
        dev = x.device
        errors = []

        views = {}

        def wrap(ptr, count):
            # view of library-owned device memory as a tensor (no copy); the solver reuses a handful of addresses
            key = (ptr, count)
            t = views.get(key)
            if t is None:
                t = views[key] = torch.as_tensor(_DevArray(ptr, count, is64), device=dev)
            return t

        def cb(ctx, xp, yp):
            try:
                A(wrap(xp, n), wrap(yp, n))
            except Exception as e:  # noqa: BLE001 - must not propagate through C
                errors.append(e)

        def red(ctx, sp, count, f64):
            try:
                reduce(wrap(sp, count))
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        cfun = N.ACTION_CB(cb)
        if reduce is None:
            N.check_capi(lib.cuddh_gmres_callback(n, _ptr(x, dtype, n, "x"), cfun, None, _ptr(b, dtype, n, "b"), int(is64), m, maxit, float(tol), verbose,
                                                  float(max_seconds), C.byref(res), _h(h_res), _h(h_time)), "gmres")
        else:
            rfun = N.REDUCE_CB(red)
            N.check_capi(lib.cuddh_gmres_callback_sharded(n, _ptr(x, dtype, n, "x"), cfun, None, rfun, None, _ptr(b, dtype, n, "b"), int(is64), m, maxit, float(tol),
                                                          verbose, float(max_seconds), C.byref(res), _h(h_res), _h(h_time)), "gmres")
        if errors:
            raise errors[0]
    return _solver_out(res, h_res, h_time)


class _DevArray:
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned device memory."""

    def __init__(self, ptr: int, count: int, is64: bool):
        self.__cuda_array_interface__ = {
            "shape": (count,),
            "typestr": "<f8" if is64 else "<f4",
            "data": (int(ptr), False),
            "version": 2,
        }
