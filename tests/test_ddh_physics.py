"""DDH against the PDE: the independent check the reference does not ship (it has no DDH test, tests/test.hpp:12-18).

tests/helmholtz_direct.py builds, from the mesh geometry and the pinned 1-D tables only, (a) the WaveHoltz local solve as
an ODE integrated to 1e-11, (b) the linear system that DDH's transmission conditions and slot rule imply (what DDH
converges to when its local solves are exact).  Neither follows source/DDH.cpp's loops, index tables or slot maps, so
agreement is evidence for the set-up the restatements share (lumped masses m and gmi, coefficient a, face mass H, slot
table B incl. its cross-point rule, filter, time grid, cs/sn; source/DDH.cpp:363-608) and for the kernels' time stepping --
for the oracle (CPU tests) and for the HIP product (GPU tests) separately.

Measured distances (relative l2; every test prints its own, `pytest -s` shows them):
  local solves (postprocess with zero traces) vs continuous-in-time WaveHoltz ............ 2e-5   (O(dt^2): RK2 + trapezoid)
  converged DDH, 12 WaveHoltz iterations, vs the implied system (strips AND 2-D) ......... 1e-4   (the same O(dt^2))
  converged DDH, the reference's 5 iterations, vs the implied system ..................... 1.3e-2 (4 subdomains) to
      ~1e-1 (1024 subdomains): truncation of the local solves (2e-3 each) amplified by (I - T)^-1
  converged DDH vs the plain Helmholtz solve of examples/Helmholtz.hpp semantics ......... 4e-2 .. 1e-1, of which the
      reference's load / junction / cross-point treatment (helmholtz_direct.ddh_fixed_point_solution) is ~3e-2
The WaveHoltz iteration count is the reference's constant 5 (source/DDH.cpp:136) everywhere except in these tests, which
raise it through the verification knob (oracle.ddh_set_wh_iters / DDH.set_wh_iters) to take the truncation away.
"""
import math

import numpy as np
import pytest

import helmholtz_direct as hd
import oracle


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))


def coefficient(xy):
    """two-valued like examples/DDH.cpp:74-83, placed off-centre so that no symmetry hides an index error"""
    return np.where((xy[0] - 0.1) ** 2 + (xy[1] + 0.05) ** 2 < 0.09, 0.5, 1.0)


def bump(cx, cy, s=40.0):
    return lambda x, y: np.exp(-s * ((x - cx) ** 2 + (y - cy) ** 2))


class Case:
    """uniform_rect(nx, -1, 1, ny, -ny h/2, ny h/2) (square elements), n_basis 4, omega = 2 pi nx / 10 (examples/DDH.cpp:89)"""

    def __init__(self, nx, ny, nb=4):
        self.nx, self.ny, self.nb = nx, ny, nb
        self.h = 2.0 / nx
        self.y0 = -ny * self.h / 2
        self.omega = 2 * math.pi * nx / 10
        self.om = oracle.Mesh.uniform_rect(nx, -1.0, 1.0, ny, self.y0, -self.y0)
        self.d = oracle.Discretization(self.om, nb)
        self.xy = self.d.coordinates()
        self.h_a = coefficient(self.xy)
        self.grid = hd.UniformGrid(nx, ny, nb, self.xy, -1.0, self.y0, self.h)
        self.ne = 16 // nb
        self.fu = oracle.linear_functional(self.d, bump(-0.37, 0.05 if ny < nx else 0.21))
        self.fv = 0.3 * oracle.linear_functional(self.d, bump(0.42, -0.02 if ny < nx else -0.33))
        self.ndof = self.d.ndof

    def grid_uv(self, U):
        return np.concatenate([self.grid.from_grid(U.real.copy()), self.grid.from_grid(U.imag.copy())])

    def implied_solution(self):
        """direct solve of the system DDH with exact local solves converges to (any block decomposition)"""
        g = self.grid
        return self.grid_uv(hd.ddh_fixed_point_solution(g, self.ne, self.omega, g.to_grid(self.h_a), g.to_grid(self.fu) + 1j * g.to_grid(self.fv)))

    def implied_solution_strips(self):
        """the same for strips through the simpler single-valued form (no cross points): cross-checks the general form"""
        g = self.grid
        K, f = hd.ddh_fixed_point_system(g, self.ne, self.omega, g.to_grid(self.h_a), g.to_grid(self.fu) + 1j * g.to_grid(self.fv))
        return self.grid_uv(hd.solve_complex(K, f))

    def continuous_local_solves(self):
        g = self.grid
        u, v = hd.local_solves_continuous(g, self.ne, self.omega, g.to_grid(self.h_a), g.to_grid(self.fu), g.to_grid(self.fv))
        return np.concatenate([g.from_grid(u), g.from_grid(v)])

    def helmholtz_solutions(self):
        """direct solves of the Helmholtz system itself: collocated (DDH's own discretisation) and with the
        reference-pinned consistent operators (examples/Helmholtz.hpp:28-56), with the plain load and with the load DDH
        effectively applies (a dof held by k subdomains counts k times)."""
        g = self.grid
        mult = g.from_grid(g.multiplicity(self.ne))
        Kc = hd.collocated_system(g, self.omega, g.to_grid(self.h_a))
        Kp = hd.consistent_system(self.d, self.omega, self.h_a)
        F = self.fu + 1j * self.fv
        out = {}
        for name, load in (("ddh load", F * mult), ("plain load", F)):
            out["collocated, " + name] = self.grid_uv(hd.solve_complex(Kc, g.to_grid(load)))
            out["consistent, " + name] = hd.as_uv(hd.solve_complex(Kp, load))
        return out

    @property
    def f(self):
        return np.concatenate([self.fu, self.fv])


def oracle_solve(case, wh_iters, tol=1e-10, m=120):
    oracle.ddh_set_wh_iters(wh_iters)
    try:
        O = oracle.DDH(case.d, case.nx, case.ny, case.omega, case.h_a, np.float64)
        b = O.rhs(case.f)
        lam, info = oracle.gmres(O.action, b, m=m, maxit=30, tol=tol)
        assert info["success"], info["res_norm"][-1] / info["res_norm"][0]
        return O.postprocess(lam, case.f), info
    finally:
        oracle.ddh_set_wh_iters(5)


# =============================================================================================== oracle (CPU)
def test_oracle_local_solves_vs_continuous_waveholtz():
    """postprocess with zero traces = partition-of-unity sum of the subdomains' WaveHoltz solves (source/DDH.cpp:237-307)
    against the same thing integrated as an ODE: pins m, gmi, a, H, the sweep's geometry, filter, cs/sn and the time grid."""
    c = Case(8, 8)
    O = oracle.DDH(c.d, c.nx, c.ny, c.omega, c.h_a, np.float64)
    y = O.postprocess(np.zeros(O.size), c.f)
    e = rel(y, c.continuous_local_solves())
    print(f"oracle fp64 local solves vs continuous WaveHoltz (8x8, nt={O.t.nt}): {e:.2e}")
    assert e < 2e-4


def test_oracle_converges_to_implied_system_on_strips():
    """4 subdomains in a row (no cross points): the converged DDH solution against the direct solve of the system its
    transmission conditions imply -- tight once the local solves are run to convergence (12 WaveHoltz iterations), and
    at the truncation level of the reference's 5 iterations otherwise.  Pins the slot table B and the trace update."""
    c = Case(16, 4)
    ref = c.implied_solution()
    assert rel(ref, c.implied_solution_strips()) < 1e-12
    u12, info12 = oracle_solve(c, 12)
    u5, info5 = oracle_solve(c, 5)
    e12, e5 = rel(u12, ref), rel(u5, ref)
    plain = c.grid_uv(hd.solve_complex(hd.collocated_system(c.grid, c.omega, c.grid.to_grid(c.h_a)),
                                        c.grid.to_grid(c.fu) + 1j * c.grid.to_grid(c.fv)))
    print(f"oracle DDH64 strips 16x4: vs implied system {e12:.2e} (12 WaveHoltz iterations, {info12['num_matvec']} matvecs), "
          f"{e5:.2e} (reference's 5, {info5['num_matvec']} matvecs); vs the unmodified Helmholtz system {rel(u12, plain):.2e}")
    assert e12 < 5e-4
    assert e5 < 3e-2
    assert rel(u12, plain) > 10 * e12  # the quirks are real: the unmodified system is NOT what DDH solves


def test_oracle_2d_decomposition():
    """2x2 subdomains, one interior cross point: (i) with exact local solves DDH64 converges to the implied system, cross
    point rule of source/DDH.cpp:425-440 included; (ii) with the reference's 5 iterations it stays within the truncation
    level of it; (iii) reported: the distance to direct solves of the plain Helmholtz system (pinned consistent operators of
    examples/Helmholtz.hpp:28-56 and DDH's own collocated discretisation)."""
    c = Case(8, 8)
    ref = c.implied_solution()
    u12, info12 = oracle_solve(c, 12)
    u5, info5 = oracle_solve(c, 5)
    e12, e5 = rel(u12, ref), rel(u5, ref)
    d = {k: rel(u5, v) for k, v in c.helmholtz_solutions().items()}
    print(f"oracle DDH64 8x8: vs implied system {e12:.2e} (12 WaveHoltz iterations, {info12['num_matvec']} matvecs), {e5:.2e} "
          f"(reference's 5, {info5['num_matvec']} matvecs); vs direct Helmholtz solves: " + "; ".join(f"{k}: {v:.2e}" for k, v in d.items()))
    assert e12 < 5e-4
    assert e5 < 5e-2
    assert d["consistent, ddh load"] < 0.1 and abs(d["consistent, ddh load"] - d["collocated, ddh load"]) < 5e-3


# =============================================================================================== product (GPU)
def product(case, precision, kernel):
    import cuddhelmholtz_amd as cd

    mesh = cd.Mesh2D.uniform_rect(case.nx, -1.0, 1.0, case.ny, case.y0, -case.y0)
    fem = cd.H1Space(mesh, cd.Basis(case.nb))
    assert np.allclose(fem.physical_coordinates(), case.xy, atol=1e-13)  # same numbering as the oracle's (tested elsewhere too)
    F = cd.DDH(case.omega, case.h_a, fem, case.nx, case.ny, precision=precision, kernel=kernel)
    F._keep = (mesh, fem)
    return F


def product_solve(case, F, cuda, wh_iters, tol, m=120):
    import torch

    import cuddhelmholtz_amd as cd

    F.set_wh_iters(wh_iters)
    n = F.size()
    f = torch.from_numpy(case.f).to(cuda)
    b = torch.zeros(n, dtype=F.trace_dtype, device=cuda)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * case.ndof, dtype=torch.float64, device=cuda)
    F.rhs(f, b)
    out = cd.gmres(n, lam, F, b, m, 30, tol)
    assert out.success, out.res_norm[-1] / out.res_norm[0]
    F.postprocess(lam, f, u)
    torch.cuda.synchronize()
    return u.cpu().numpy(), out


@pytest.mark.gpu
@pytest.mark.parametrize("precision,kernel,tol", [("f64", 1, 2e-4), ("f64", 2, 2e-4), ("f32", 3, 5e-4), ("f32", 5, 5e-4)])
def test_product_local_solves_vs_continuous_waveholtz(cuda, precision, kernel, tol):
    import torch

    c = Case(8, 8)
    F = product(c, precision, kernel)
    assert F.info()["kernel"] == kernel
    u = torch.zeros(2 * c.ndof, dtype=torch.float64, device=cuda)
    F.postprocess(torch.zeros(F.size(), dtype=F.trace_dtype, device=cuda), torch.from_numpy(c.f).to(cuda), u)
    e = rel(u.cpu().numpy(), c.continuous_local_solves())
    print(f"product {precision} kernel {kernel} local solves vs continuous WaveHoltz (8x8): {e:.2e}")
    assert e < tol


@pytest.mark.gpu
def test_product_converges_to_implied_system_on_strips(cuda):
    c = Case(32, 4)  # 8 subdomains in a row
    ref = c.implied_solution()
    F64 = product(c, "f64", 0)
    u12, o12 = product_solve(c, F64, cuda, 12, 1e-10)
    u5, o5 = product_solve(c, F64, cuda, 5, 1e-10)
    F32 = product(c, "f32", 0)
    u32, o32 = product_solve(c, F32, cuda, 5, 1e-5)
    e12, e5, e32 = rel(u12, ref), rel(u5, ref), rel(u32, ref)
    print(f"product strips 32x4: DDH64 (kernel {F64.info()['kernel']}) vs implied system {e12:.2e} (12 WaveHoltz iterations, "
          f"{o12.num_matvec} matvecs), {e5:.2e} (5 iterations, {o5.num_matvec} matvecs); fp32 DDH (kernel {F32.info()['kernel']}, "
          f"5 iterations, tol 1e-5, {o32.num_matvec} matvecs) {e32:.2e}; fp32 vs fp64 solution {rel(u32, u5):.2e}")
    assert e12 < 5e-4
    assert e5 < 6e-2 and e32 < 6e-2  # truncated local solves; grows slowly with the number of subdomains
    assert rel(u32, u5) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("nx", [16, 32])
def test_product_2d_decomposition(cuda, nx):
    """4x4 and 8x8 subdomains (9 and 49 interior cross points).  (i) exact local solves: DDH64 on the GPU converges to the
    independently derived implied system; (ii) the reference's 5 iterations, fp64 and fp32 (the benchmarked kernel 5):
    distance to the implied system = truncation level, fp32 solution = fp64 solution to fp32 accuracy; (iii) reported: the
    distance to a direct fp64 solve of the Helmholtz system built from the reference-pinned S / M / FaceMass operators
    (examples/Helmholtz.hpp:28-56) -- the check VERDICT r1 asked for."""
    c = Case(nx, nx)
    ref = c.implied_solution()
    F64 = product(c, "f64", 0)
    u12, o12 = product_solve(c, F64, cuda, 12, 1e-10, m=200)
    u64, o64 = product_solve(c, F64, cuda, 5, 1e-10, m=200)
    F32 = product(c, "f32", 0)
    assert F32.info()["kernel"] == 5
    u32, o32 = product_solve(c, F32, cuda, 5, 1e-5, m=200)
    e12, e5, e32 = rel(u12, ref), rel(u64, ref), rel(u32, ref)
    d64 = {k: rel(u64, v) for k, v in c.helmholtz_solutions().items()}
    print(f"product {nx}x{nx}: DDH64 (kernel {F64.info()['kernel']}) vs implied system {e12:.2e} (12 WaveHoltz iterations, {o12.num_matvec} "
          f"matvecs), {e5:.2e} (5 iterations, {o64.num_matvec} matvecs); fp32 DDH kernel 5 (tol 1e-5, {o32.num_matvec} matvecs) {e32:.2e}, "
          f"fp32 vs fp64 solution {rel(u32, u64):.2e}; DDH64 vs direct Helmholtz solves: " + "; ".join(f"{k}: {v:.2e}" for k, v in d64.items()))
    assert e12 < 5e-4
    assert e5 < 0.15 and e32 < 0.15
    assert rel(u32, u64) < 5e-3
    assert d64["consistent, ddh load"] < 0.2
