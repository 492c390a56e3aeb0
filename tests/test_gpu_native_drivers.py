"""The C++ API end to end on the GPU, without Python in the compute path: this repository's ddh_solve driver and
(when it was built, i.e. where the reference tree was present at build time) the reference's own examples/DDH.cpp
compiled unchanged.  Their solution files must equal the solve driven through the Python mirror of the same API
(same library, deterministic kernels)."""
import math
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
EX = ROOT / "build" / "examples"


def python_solve(cuda, nx, nb, omega, m, maxit, tol):
    import torch

    import cuddhelmholtz_amd as cd

    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    n = fem.size()
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    a = torch.zeros(n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
    F = cd.DDH(omega, a.cpu().numpy(), fem, nx, nx)
    lam = torch.zeros(F.size(), dtype=torch.float32, device=cuda)
    b = torch.zeros_like(lam)
    u = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    F.rhs(f, b)
    out = cd.gmres(F.size(), lam, F, b, m, maxit, tol)
    F.postprocess(lam, f, u)
    return u.cpu().numpy(), fem.physical_coordinates().reshape(-1, order="F"), out


def test_ddh_solve_driver(cuda, tmp_path):
    exe = EX / "ddh_solve"
    if not exe.exists():
        pytest.fail("build/examples/ddh_solve missing: run __graft_entry__.build()")
    nx, nb, w_over_pi = 32, 4, 6.4
    (tmp_path / "sol").mkdir()
    r = subprocess.run([str(exe), str(nx), str(nb), str(w_over_pi), "20", "40", "1e-4", str(tmp_path / "sol")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("ddh_solve")][-1]
    iters = int(re.search(r"num_iter=(\d+)", line).group(1))
    u_cpp = np.fromfile(tmp_path / "sol" / "ddh.0000")
    xy_cpp = np.fromfile(tmp_path / "sol" / "xy.0000")
    u_py, xy_py, out = python_solve(cuda, nx, nb, math.pi * w_over_pi, 20, 40, 1e-4)
    assert out.num_iter == iters
    assert np.array_equal(xy_cpp, xy_py)
    # the solution is assembled in a fixed order (round 1 used fp64 atomics here): the two hosts agree bit for bit
    assert np.array_equal(u_cpp, u_py)
    assert np.isfinite(u_cpp).all() and np.linalg.norm(u_cpp) > 0


def test_ddh_solve_driver_multi_gpu_host(cuda, tmp_path):
    """The same driver through cuddh::ddh_solve_multi_gpu with one device and the communicator forced on: a pure C++ process
    (no torch: RCCL comes from ROCm's own librccl, bound at run time) must reproduce the single-device solve."""
    exe = EX / "ddh_solve"
    if not exe.exists():
        pytest.fail("build/examples/ddh_solve missing: run __graft_entry__.build()")
    nx, nb, w_over_pi = 32, 4, 6.4
    sols = {}
    for tag, extra in (("plain", []), ("multi", ["1", "1"])):
        (tmp_path / tag).mkdir()
        r = subprocess.run([str(exe), str(nx), str(nb), str(w_over_pi), "20", "40", "1e-4", str(tmp_path / tag), *extra], capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("ddh_solve")][-1]
        sols[tag] = (np.fromfile(tmp_path / tag / "ddh.0000"), int(re.search(r"num_matvec=(\d+)", line).group(1)), line)
    assert "devices=1 rccl=1" in sols["multi"][2]
    assert abs(sols["multi"][1] - sols["plain"][1]) <= 2
    err = np.linalg.norm(sols["multi"][0] - sols["plain"][0]) / np.linalg.norm(sols["plain"][0])
    print(f"native driver, multi-GPU host with one rank over RCCL vs plain: {err:.2e}")
    assert err < 1e-3


def test_reference_ddh_example_runs_unchanged(cuda, tmp_path):
    """examples/DDH.cpp of the reference (128^2, degree 3, omega = 2 pi 12.8, GMRES(20), maxit 100, tol 1e-4),
    compiled unchanged against csrc/include/cuddh.hpp."""
    exe = EX / "DDH_reference_driver"
    if not exe.exists():
        pytest.skip("reference example was not built (reference tree absent at build time)")
    (tmp_path / "solution").mkdir()
    r = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "#lambda = 103168" in r.stdout
    u_cpp = np.fromfile(tmp_path / "solution" / "ddh.0000")
    nx = 128
    omega = 2 * math.pi * nx / 10
    u_py, xy_py, out = python_solve(cuda, nx, 4, omega, 20, 100, 1e-4)
    assert u_cpp.size == u_py.size == 2 * (3 * nx + 1) ** 2
    assert np.linalg.norm(u_cpp - u_py) <= 1e-10 * np.linalg.norm(u_py)
    assert np.array_equal(np.fromfile(tmp_path / "solution" / "xy.0000"), xy_py)


def test_reference_test_suite_runs_unchanged(cuda, tmp_path):
    """The reference's own test suite -- tests/test.cpp with quadrature_rule, basis, gmres (fp64 and fp32 Toeplitz solves),
    linalg (every BLAS-1 routine, both precisions), mass (action, inverse through gmres, DiagInvMassMatrix) and stiffness on
    uniform and on the unstructured mesh it loads from text files -- compiled unchanged against csrc/include/cuddh.hpp and
    run on the GPU: every known-answer check the reference holds for this path, evaluated by the reference's own code."""
    import re

    exe = EX / "reference_tests_driver"
    if not exe.exists():
        pytest.skip("reference test suite was not built (reference tree absent at build time)")
    r = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) / (\d+) tests passed", r.stdout)
    assert m, r.stdout[-2000:]
    passed, total = int(m.group(1)), int(m.group(2))
    print(f"reference test suite against the product: {passed} / {total} passed")
    assert total >= 20 and passed == total, r.stdout[-4000:]


def test_reference_poisson_example_runs_unchanged(cuda, tmp_path):
    """BASELINE config 1: examples/Poisson.cpp of the reference (15^2 elements, degree 3, Dirichlet lifting through
    FaceLinearFunctional + preconditioned FaceMassMatrix solve, GMRES(20) to 1e-6), compiled unchanged against
    csrc/include/cuddh.hpp, checked against the same flow on the oracle."""
    import oracle

    exe = EX / "Poisson_reference_driver"
    if not exe.exists():
        pytest.skip("reference example was not built (reference tree absent at build time)")
    (tmp_path / "solution").mkdir()
    r = subprocess.run([str(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "#dof = 2116" in r.stdout and "GMRES successfully converged" in r.stdout
    u = np.fromfile(tmp_path / "solution" / "poisson.0000")
    xy = np.fromfile(tmp_path / "solution" / "xy.0000").reshape(-1, 2).T

    nx, nb = 15, 4
    d = oracle.Discretization(oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), nb)
    assert u.size == d.ndof == 2116 and np.allclose(xy, d.coordinates(), rtol=0, atol=1e-14)
    faces = list(d.mesh.boundary_edges)
    ofs = oracle.FaceSpaceO(d, faces)
    So = oracle.Stiffness(d)

    def g(x, y):  # examples/Poisson.cpp:73-81
        return np.where(np.abs(x - 1.0) < 1e-12, 1.0 - y * y, np.where(np.abs(x + 1.0) < 1e-12, y * (1.0 - y * y), 0.0))

    def A(v):
        w = So.apply(v)
        w[ofs.proj] = 0.0
        return w

    b = oracle.linear_functional(d, lambda x, y: np.ones_like(x))
    b[ofs.proj] = 0.0
    yo = oracle.face_linear_functional(ofs, g)
    Mo, po = oracle.FaceMass(ofs), oracle.diag_inv_facemass(ofs)
    q, _ = oracle.gmres(lambda v: po * Mo.apply(v), po * yo, m=5, maxit=10, tol=1e-12)
    G = np.zeros(d.ndof)
    G[ofs.proj] += q
    b -= A(G)
    uo, info = oracle.gmres(A, b, m=20, maxit=20, tol=1e-6)
    uo += G
    assert info["success"] and f"After {info['num_iter']} iterations" in r.stdout
    assert np.linalg.norm(u - uo) <= 1e-9 * np.linalg.norm(uo)


@pytest.mark.parametrize("ordering", ["native", "reference"])
def test_helmholtz_solve_driver(cuda, tmp_path, ordering):
    """BASELINE config 2 in miniature: the native helmholtz_solve driver (unpreconditioned GMRES(20) on the fused complex
    apply) against the same pipeline driven from Python through the C handle layer; with the iteration vectors in the plan's own
    ordering (HelmholtzOperator::gmres, the driver's default) and in the reference ordering."""
    import torch

    import cuddhelmholtz_amd as cd

    exe = EX / "helmholtz_solve"
    if not exe.exists():
        pytest.fail("build/examples/helmholtz_solve missing: run __graft_entry__.build()")
    nx, nb, w_over_pi, m, maxit = 48, 4, 3.0, 20, 6
    (tmp_path / "sol").mkdir()
    r = subprocess.run([str(exe), str(nx), str(nb), str(w_over_pi), str(m), str(maxit), "0", str(tmp_path / "sol"), ordering], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("helmholtz_solve")][-1]
    assert "fused=1" in line and f"ordering={ordering}" in line
    nmv = int(re.search(r"num_matvec=(\d+)", line).group(1))
    U_cpp = np.fromfile(tmp_path / "sol" / "helmholtz.0000")

    omega = math.pi * w_over_pi
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    n = fem.size()
    a2 = torch.zeros(n, dtype=torch.float64, device=cuda)
    cd.nodal_values(fem, cd.ALPHA_DISK_SQ, a2)
    A = cd.HelmholtzOperator(omega, a2, torch.ones(fs.size(), dtype=torch.float64, device=cuda), fem, fs)
    b = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, b[:n], param=omega)
    x = torch.zeros_like(b)
    out = cd.gmres(2 * n, x, A, b, m, maxit, 0.0)
    assert out.num_matvec == nmv == 1 + (maxit - 1) * (m + 1)
    # (native ordering: the inner products sum in another order, nothing else differs)
    assert np.linalg.norm(U_cpp - x.cpu().numpy()) <= (1e-10 if ordering == "reference" else 1e-8) * np.linalg.norm(U_cpp)
    assert out.res_norm[-1] < out.res_norm[0]


@pytest.mark.parametrize("world,split,grid", [(2, False, None), (3, False, None), (4, False, None), (2, True, None), (4, True, None),
                                              (4, False, (2, 2)), (4, True, (2, 2)), (2, True, (2, 1))])
def test_multi_gpu_host_loopback_ranks(cuda, world, split, grid):
    """cuddh::ddh_solve_multi_gpu with its loopback transport: `world` ranks as host threads sharing the test GPU (one stream
    each), messages as device-to-device copies, reductions summed on the host in rank order.  Everything of the C++ N > 1
    path runs -- TraceExchangePlan, trace pack / unpack, partitioned Krylov vectors, the GMRES reduce hook, the final sum of u
    -- except the RCCL calls themselves (those run with one rank in test_multi_gpu_host_one_rank).  The traces are copied,
    never summed, so the iteration is the single-process one up to the order of the inner-product sums.  split: the north
    star's schedule in the C++ host (boundary subdomains as one listed launch with issue priority on a second stream, exchange
    behind them, interior on the main stream meanwhile).  grid: the ranks own rectangles of the subdomain grid (SURVEY 8e's
    gx x gy; a rank's subdomains are then not one range: listed launches, cross points between four ranks)."""
    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import ddh_solve_multi_gpu

    nx, nb = 32, 4
    omega = 2 * math.pi * nx / 10
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    n = fem.size()
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    a = torch.zeros(n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
    h_a, h_f = a.cpu().numpy(), f.cpu().numpy()
    F = cd.DDH(omega, h_a, fem, nx, nx)
    b = torch.zeros(F.size(), dtype=torch.float32, device=cuda)
    lam = torch.zeros_like(b)
    u = torch.zeros_like(f)
    F.rhs(f, b)
    out = cd.gmres(F.size(), lam, F, b, 20, 100, 1e-4)
    F.postprocess(lam, f, u)
    from cuddhelmholtz_amd import _native as N

    stream_before = N.lib.cuddh_get_stream()
    u_multi, info = ddh_solve_multi_gpu(nx, nb, omega, h_a, h_f, world=world, m=20, maxit=100, tol=1e-4, force_rccl=2, split_schedule=split, rank_grid=grid)
    assert N.lib.cuddh_get_stream() == stream_before  # the call gives the caller's launch stream back (rank 0 ran on this thread)
    assert info["world"] == world and not info["used_rccl"] and info["success"] == int(out.success) == 1
    assert info["bytes_sent_per_action_rank0"] > 0
    assert abs(info["num_matvec"] - out.num_matvec) <= 2
    err = float(np.linalg.norm(u_multi - u.cpu().numpy()) / np.linalg.norm(u.cpu().numpy()))
    print(f"multi-GPU host, {world} loopback ranks, split schedule {split}, grid {grid}: {info['num_matvec']} matvecs (plain {out.num_matvec}), u vs plain solve {err:.2e}")
    assert err < 1e-3


@pytest.mark.parametrize("force_rccl", [False, True])
def test_multi_gpu_host_one_rank(cuda, force_rccl):
    """cuddh::ddh_solve_multi_gpu (one process, one host thread and one stream per device, RCCL) with the single device of
    the test box: with force_rccl the one-rank communicator carries every inner product (ncclAllReduce through the GMRES
    reduce hook) and the final all-reduce of u -- the code path N > 1 runs -- and the result must be the plain DDH solve's."""
    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import ddh_solve_multi_gpu

    nx, nb = 32, 4
    omega = 2 * math.pi * nx / 10
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    n = fem.size()
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    a = torch.zeros(n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
    cd.linear_functional(fem, cd.ALPHA_DISK, a)
    cd.DiagInvMassMatrix(fem).action(a, a)
    h_a, h_f = a.cpu().numpy(), f.cpu().numpy()
    F = cd.DDH(omega, h_a, fem, nx, nx)
    b = torch.zeros(F.size(), dtype=torch.float32, device=cuda)
    lam = torch.zeros_like(b)
    u = torch.zeros_like(f)
    F.rhs(f, b)
    out = cd.gmres(F.size(), lam, F, b, 20, 100, 1e-4)
    F.postprocess(lam, f, u)
    u_multi, info = ddh_solve_multi_gpu(nx, nb, omega, h_a, h_f, world=1, m=20, maxit=100, tol=1e-4, force_rccl=force_rccl)
    assert info["world"] == 1 and bool(info["used_rccl"]) == force_rccl and info["success"] == int(out.success) == 1
    assert abs(info["num_matvec"] - out.num_matvec) <= 2
    err = float(np.linalg.norm(u_multi - u.cpu().numpy()) / np.linalg.norm(u.cpu().numpy()))
    print(f"multi-GPU host, one rank, rccl={force_rccl}: {info['num_matvec']} matvecs (plain {out.num_matvec}), u vs plain solve {err:.2e}, "
          f"t_gmres {info['t_gmres']:.2f} s")
    assert err < 1e-3


@pytest.mark.parametrize("world,fail_rank,split", [(2, 1, False), (3, 0, False), (4, 2, True)])
def test_multi_gpu_host_rank_failure_ends_the_call(cuda, monkeypatch, world, fail_rank, split):
    """One rank throwing (here: injected after its right-hand side through the test hook CUDDH_MULTIGPU_FAIL_RANK) must end the
    call on every rank with that rank's error -- not leave the others blocked in the next barrier / collective.
    Afterwards the caller's stream is intact and a second, healthy call still works."""
    import threading

    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd import _native as N
    from cuddhelmholtz_amd.dist import ddh_solve_multi_gpu

    nx, nb = 16, 4
    omega = 2 * math.pi * nx / 10
    fem = cd.H1Space(cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), cd.Basis(nb))
    n = fem.size()
    f = torch.zeros(2 * n, dtype=torch.float64, device=cuda)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:n], param=omega)
    h_a, h_f = np.ones(n), f.cpu().numpy()
    stream_before = N.lib.cuddh_get_stream()
    monkeypatch.setenv("CUDDH_MULTIGPU_FAIL_RANK", str(fail_rank))
    result = {}

    def call():
        try:
            ddh_solve_multi_gpu(nx, nb, omega, h_a, h_f, world=world, m=20, maxit=50, tol=1e-4, force_rccl=2, split_schedule=split)
            result["error"] = None
        except Exception as e:  # noqa: BLE001
            result["error"] = str(e)

    t = threading.Thread(target=call, daemon=True)
    t.start()
    t.join(120)
    assert not t.is_alive(), "ddh_solve_multi_gpu did not return after one rank failed"
    assert result["error"] and "injected failure" in result["error"], result
    monkeypatch.delenv("CUDDH_MULTIGPU_FAIL_RANK")
    assert N.lib.cuddh_get_stream() == stream_before
    u, info = ddh_solve_multi_gpu(nx, nb, omega, h_a, h_f, world=world, m=20, maxit=50, tol=1e-4, force_rccl=2, split_schedule=split)
    assert info["success"] == 1 and np.isfinite(u).all()


@pytest.mark.parametrize("kind,nb,world", [("structured", 4, 2), ("structured", 4, 3), ("structured", 3, 4), ("refined", 4, 2), ("refined", 5, 3),
                                           ("refined", 6, 4), ("structured", 4, 1)])
def test_multi_gpu_helmholtz_host_loopback_ranks(cuda, kind, nb, world):
    """cuddh::helmholtz_multi_gpu (SURVEY 8e "global operator apply" in the C++ host): the fused Helmholtz operator partitioned
    over `world` loopback ranks (host threads sharing the test GPU, one stream each; element partition, sub-mesh plans, the two
    halo exchanges through cuddh_hip_halo_pack / unpack, partial sums added in rank order) against the single-device operator:
    the apply to 1e-13, and a GMRES solve through the reduce hook against gmres() on the single-device operator."""
    import torch

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import helmholtz_multi_gpu

    if kind == "structured":
        mesh = cd.Mesh2D.uniform_rect(24, -1.0, 1.0, 20, -1.0, 1.0)
    else:
        mesh = cd.Mesh2D.load(GOLDEN / "unstructured_square").refined(1)
    fem = cd.H1Space(mesh, cd.Basis(nb))
    n = fem.size()
    fs = cd.FaceSpace(fem, mesh.boundary_edges())
    rng = np.random.default_rng(100 + nb + world)
    a2, ax = 0.5 + rng.random(n), 0.5 + rng.random(fs.size())
    xh = rng.standard_normal(2 * n)
    omega = 5.0
    A = cd.HelmholtzOperator(omega, torch.from_numpy(a2).to(cuda), torch.from_numpy(ax).to(cuda), fem, fs)
    x = torch.from_numpy(xh).to(cuda)
    y = torch.empty_like(x)
    A.action(x, y)
    ref = y.cpu().numpy()
    from cuddhelmholtz_amd import _native as N

    stream_before = N.lib.cuddh_get_stream()
    got, info = helmholtz_multi_gpu(mesh, nb, omega, a2, ax, xh, world=world, transport=2, reps=2)
    assert N.lib.cuddh_get_stream() == stream_before
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(f"helmholtz_multi_gpu {kind} n_basis {nb}, {world} loopback ranks: apply vs single device {err:.2e}; "
          f"largest share {info['n_loc_max']} dofs ({info['n_halo_max']} halo), {info['halo_bytes_per_apply_max']} B sent per apply")
    assert info["world"] == world and not info["used_rccl"]
    assert err < 1e-13
    if world > 1:
        assert info["n_halo_max"] > 0 and info["halo_bytes_per_apply_max"] > 0 and info["n_loc_max"] < n
    # GMRES(15), 3 cycles on A y = x: partitioned vectors + all-reduced inner products vs the single-device solver
    sol = torch.zeros_like(x)
    out = cd.gmres(2 * n, sol, A, x, 15, 4, 0.0)
    got, info = helmholtz_multi_gpu(mesh, nb, omega, a2, ax, xh, world=world, transport=2, m=15, maxit=4, tol=0.0)
    assert info["num_matvec"] == out.num_matvec
    assert abs(info["res_norm"][-1] - out.res_norm[-1]) <= 1e-9 * out.res_norm[0]
    assert float(np.linalg.norm(got - sol.cpu().numpy()) / np.linalg.norm(sol.cpu().numpy())) < 1e-8
