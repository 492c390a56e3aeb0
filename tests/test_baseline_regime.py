"""DDH parity in BASELINE.json's OWN regime: configs 3 and 4 are 32 elements per wavelength (omega = pi nx / 32: 16 pi at 512^2,
32 pi at 1024^2), where the reference's time grid (source/DDH.cpp:363-368: dt = 0.1 h / n_basis^2, nt = ceil(T / dt)) gives
nt = 5120, i.e. 25,600 RK2 steps per local solve (source/DDH.cpp:237-293) -- six times the example's regime
(omega = 2 pi nx / 10, nt = 800) that the other oracle comparisons run.

The oracle cannot do 65,536 such subdomains in seconds, so these tests cut a WINDOW out of the configuration: 16 x 16 elements
(16 reference-size subdomains, 9 interior cross points, physical-boundary edges all round) with the configuration's mesh width
h = 2 / nx and its omega, placed
  * "one":  anywhere (coefficient a = 1), and
  * "disk": across the rim of the example's two-valued coefficient (examples/DDH.cpp:74-83: a = 0.2 for r < 0.25), so that the
            window holds subdomains inside, outside and cut by the disk.
Every number a subdomain's local solve depends on (h, omega, dt, nt, filter, cs / sn, a, m, H, the slot table) is then exactly
the full problem's; only the position of the physical boundary differs.

What is gated:
  fp64 kernels 1 and 2 vs the fp64 oracle, every entry point (rhs / action / postprocess) ......... <= 1e-10 with a = 1
      (measured 3e-16 .. 1.5e-15).
      With the disk coefficient the local solves are UNSTABLE in this regime (test_*_stability below: |T| ~ 1e7 -- the explicit
      midpoint rule amplifies the fast modes of the a = 0.2 region over 25,600 steps), so rounding seeds of 1e-16 are
      amplified like everything else and NO two evaluation orders of the same arithmetic agree to 1e-10 (measured: rhs 7e-10,
      action 7e-13, postprocess 2e-11).  The gate there is 20 x the ORACLE'S OWN sensitivity: the distance between two fp64
      oracle runs whose forcing / traces differ by a random relative 2^-52 per entry (one ulp) -- computed in the test, printed.
  fp32 kernels 3 and 5 (5 = what bench.py times) vs the fp64 oracle ............................... printed and gated at the
      level the fp32 ORACLE itself reaches against the fp64 oracle (x 4, at least 2e-4): in the reference's precision nothing
      can do better.  With the disk coefficient that level is O(0.1 .. 1) for rhs (fp32 eps x |T|): the reference's fp32
      arithmetic has NO meaningful digits there, the test then only demands finite output of the oracle's magnitude.
"""
import math

import numpy as np
import pytest

import oracle

NB = 4
NW = 16  # window: 16 x 16 elements = 4 x 4 subdomains
CONFIGS = {"config3_512_16pi": (512, 16 * math.pi), "config4_1024_32pi": (1024, 32 * math.pi)}
CENTRES = {"one": (-0.53125, 0.3125), "disk": (0.25, 0.0)}


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / np.linalg.norm(b))


class Window:
    def __init__(self, config, coef, nw=NW):
        nx, self.omega = CONFIGS[config]
        self.h = 2.0 / nx
        cx, cy = CENTRES[coef]
        self.nw = nw
        self.x0, self.y0 = cx - nw * self.h / 2, cy - nw * self.h / 2
        self.x1, self.y1 = self.x0 + nw * self.h, self.y0 + nw * self.h
        self.om = oracle.Mesh.uniform_rect(nw, self.x0, self.x1, nw, self.y0, self.y1)
        self.d = oracle.Discretization(self.om, NB)
        self.ndof = self.d.ndof
        self.h_a = self.d.nodal(oracle.alpha_disk) if coef == "disk" else np.ones(self.ndof)
        s = self.omega**2
        gx, gy = cx - 3 * self.h, cy + 2 * self.h

        def forcing(x, y):  # the example's Gaussian (examples/DDH.cpp:61-72), moved into the window
            return s / math.pi * np.exp(-s * ((x - gx) ** 2 + (y - gy) ** 2))

        def forcing_v(x, y):
            return 0.3 * s / math.pi * np.exp(-s * ((x - cx - 2 * self.h) ** 2 + (y - cy + 4 * self.h) ** 2))

        self.f = np.concatenate([oracle.linear_functional(self.d, forcing), oracle.linear_functional(self.d, forcing_v)])

    def oracle_ddh(self, real):
        return oracle.DDH(self.d, self.nw, self.nw, self.omega, self.h_a, real)

    def product(self, cd, precision, kernel):
        mesh = cd.Mesh2D.uniform_rect(self.nw, self.x0, self.x1, self.nw, self.y0, self.y1)
        fem = cd.H1Space(mesh, cd.Basis(NB))
        F = cd.DDH(self.omega, self.h_a, fem, self.nw, self.nw, precision=precision, kernel=kernel)
        assert F.info()["kernel"] == kernel
        return F, fem

    def traces(self, O, seed=2):
        """random trace vector, zero in the slots nobody reads (they are never initialised by the reference either)"""
        n = O.size
        rng = np.random.default_rng(seed)
        lam = rng.standard_normal(n)
        used = np.unique(O.t.B[O.t.B >= 0])
        keep = np.zeros(n, dtype=bool)
        keep[used] = True
        keep[used + O.t.n_lambda] = True
        lam[~keep] = 0.0
        return lam

    @staticmethod
    def written(O):
        w = np.unique(O.t.B[:, 1, :][O.t.B[:, 1, :] >= 0])
        return np.concatenate([w, w + O.t.n_lambda])


def power_iteration(apply_T, n, steps, seed=1):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    ratios = []
    for _ in range(steps):
        w = np.asarray(apply_T(v), dtype=np.float64)
        r = float(np.linalg.norm(w))
        ratios.append(r)
        v = w / r
    return ratios


# ===================================================================================================== oracle (CPU)
@pytest.mark.parametrize("config", list(CONFIGS))
def test_time_grid_is_baselines(config):
    """nt = 5120 for both configurations, from the window's mesh alone"""
    w = Window(config, "one", nw=4)
    O = w.oracle_ddh(np.float64)
    assert O.t.nt == 5120
    assert abs(O.t.dt * O.t.nt - 2 * math.pi / w.omega) < 1e-15


@pytest.mark.parametrize("config", list(CONFIGS))
def test_oracle_stability_in_baseline_regime(config):
    """Power iteration on T (update = T lambda, source/DDH.cpp:611-639 without the `lambda -` part) on the oracle: with a = 1
    the local solves are non-expansive in this regime, with the example's disk coefficient |T| is ~1e7 -- the explicit
    midpoint rule's growth of the fast modes of the a = 0.2 region (wave speed x 5, time step from the mesh alone) over
    5 nt = 25,600 steps.  This is a property of the reference's algorithm (the CPU restatement shows it without any GPU code);
    the GPU test below shows the product has the same number."""
    out = {}
    for coef in ("one", "disk"):
        w = Window(config, coef, nw=8)
        O = w.oracle_ddh(np.float64)
        assert O.t.nt == 5120
        out[coef] = power_iteration(lambda v: O.solve(lam=v)[1], O.size, 4)
    print(f"oracle fp64 {config} window 8x8: |T^k v|/|T^(k-1) v| a=1: " + " ".join(f"{r:.3e}" for r in out["one"])
          + " | disk: " + " ".join(f"{r:.3e}" for r in out["disk"]))
    assert 0.9 < out["one"][-1] < 1.05
    assert out["disk"][-1] > 1e5


# ===================================================================================================== product (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("coef", ["one", "disk"])
@pytest.mark.parametrize("config", list(CONFIGS))
def test_ddh_window_parity(cuda, config, coef):
    import torch

    import cuddhelmholtz_amd as cd

    w = Window(config, coef)
    O64 = w.oracle_ddh(np.float64)
    O32 = w.oracle_ddh(np.float32)
    assert (O64.t.nt, O64.t.n_domains) == (5120, 16)
    if coef == "disk":
        inside = float((w.h_a < 0.5).mean())
        assert 0.2 < inside < 0.8  # the rim really crosses the window
    lam_h = w.traces(O64)
    written = w.written(O64)
    b64 = O64.rhs(w.f)
    y64 = O64.action(lam_h)
    u64 = O64.postprocess(lam_h, w.f)
    # what the reference's own precision can reach here: fp32 oracle vs fp64 oracle
    lam32 = lam_h.astype(np.float32)
    floor = dict(rhs=rel(O32.rhs(w.f), b64), action=rel(O32.action(lam32)[written], O64.action(lam32.astype(np.float64))[written]),
                 post=rel(O32.postprocess(lam32, w.f), O64.postprocess(lam32.astype(np.float64), w.f)))
    print(f"\n[{config}, a={coef}] nt=5120, 16 subdomains; fp32 oracle vs fp64 oracle: rhs {floor['rhs']:.2e} action {floor['action']:.2e} "
          f"postprocess {floor['post']:.2e}; |rhs|/|f| = {np.linalg.norm(b64) / np.linalg.norm(w.f):.3e}")
    f = torch.from_numpy(w.f).to(cuda)
    # the fp64 oracle's own conditioning: the same entry points with inputs one ulp away (random signs)
    rng = np.random.default_rng(11)
    ulp = 2.0 ** -52
    f_p = w.f * (1 + ulp * rng.choice([-1.0, 1.0], w.f.size))
    lam_p = lam_h * (1 + ulp * rng.choice([-1.0, 1.0], lam_h.size))
    cond = (rel(O64.rhs(f_p), b64), rel(O64.action(lam_p)[written], y64[written]), rel(O64.postprocess(lam_p, f_p), u64))
    gate64 = tuple(max(1e-10, 20 * c) for c in cond)
    print(f"  fp64 oracle vs itself with inputs one ulp away: rhs {cond[0]:.2e} action {cond[1]:.2e} postprocess {cond[2]:.2e}"
          f" -> fp64 gates {gate64[0]:.1e} {gate64[1]:.1e} {gate64[2]:.1e}")
    if coef == "one":
        assert gate64 == (1e-10, 1e-10, 1e-10)  # stable: the plain 1e-10 gate of the north star

    for kernel in (1, 2):
        F, fem = w.product(cd, "f64", kernel)
        assert F.info()["nt"] == 5120 and F.size() == O64.size
        n = F.size()
        b = torch.zeros(n, dtype=torch.float64, device=cuda)
        F.rhs(f, b)
        lam = torch.from_numpy(lam_h).to(cuda)
        y = torch.zeros(n, dtype=torch.float64, device=cuda)
        F.action(lam, y)
        u = torch.zeros(2 * w.ndof, dtype=torch.float64, device=cuda)
        F.postprocess(lam, f, u)
        e = (rel(b.cpu().numpy(), b64), rel(y.cpu().numpy()[written], y64[written]), rel(u.cpu().numpy(), u64))
        print(f"  fp64 kernel {kernel} vs fp64 oracle: rhs {e[0]:.2e} action {e[1]:.2e} postprocess {e[2]:.2e}")
        assert all(x < g for x, g in zip(e, gate64)), (e, gate64)

    for kernel in (3, 5):
        F, fem = w.product(cd, "f32", kernel)
        n = F.size()
        b = torch.zeros(n, dtype=torch.float32, device=cuda)
        F.rhs(f, b)
        lam = torch.from_numpy(lam32).to(cuda)
        y = torch.zeros(n, dtype=torch.float32, device=cuda)
        F.action(lam, y)
        u = torch.zeros(2 * w.ndof, dtype=torch.float64, device=cuda)
        F.postprocess(lam, f, u)
        y64_32 = O64.action(lam32.astype(np.float64))
        u64_32 = O64.postprocess(lam32.astype(np.float64), w.f)
        e = dict(rhs=rel(b.cpu().numpy(), b64), action=rel(y.cpu().numpy()[written], y64_32[written]), post=rel(u.cpu().numpy(), u64_32))
        print(f"  fp32 kernel {kernel} vs fp64 oracle: rhs {e['rhs']:.2e} action {e['action']:.2e} postprocess {e['post']:.2e}")
        assert bool(torch.isfinite(b).all()) and bool(torch.isfinite(y).all()) and bool(torch.isfinite(u).all())
        for k in e:
            # the reference's precision: no fp32 evaluation order does better than the fp32 oracle does; where that is O(1)
            # (unstable local solves) all that can be asked is an output of the oracle's magnitude
            assert e[k] < min(1.5, max(2e-4, 4 * floor[k])), (k, e[k], floor[k])


@pytest.mark.gpu
@pytest.mark.parametrize("config", list(CONFIGS))
def test_product_stability_in_baseline_regime(cuda, config):
    """the same power iteration on the product (fp64 kernel 2 and the benchmarked fp32 kernel 5): the growth factor is the
    oracle's.  With |T| ~ 1e7 and fp32 eps 6e-8 an fp32 action with the disk coefficient carries O(1) relative noise against
    an fp64 run after ONE more application -- which is why bench.py's headline names its coefficient and times a = 1 too."""
    import torch

    import cuddhelmholtz_amd as cd

    for coef in ("one", "disk"):
        w = Window(config, coef, nw=8)
        O = w.oracle_ddh(np.float64)
        ref = power_iteration(lambda v: O.solve(lam=v)[1], O.size, 4)
        for precision, kernel in (("f64", 2), ("f32", 5)):
            F, fem = w.product(cd, precision, kernel)
            nd = F.info()["n_domains"]
            tt = F.trace_dtype

            def T(v):
                lam = torch.from_numpy(np.ascontiguousarray(v)).to(cuda).to(tt)
                out = torch.zeros_like(lam)
                F.local_traces(0, nd, None, lam, out)
                return out.double().cpu().numpy()

            got = power_iteration(T, F.size(), 4)
            print(f"{config} a={coef} {precision} kernel {kernel}: " + " ".join(f"{r:.3e}" for r in got) + " (oracle " + " ".join(f"{r:.3e}" for r in ref) + ")")
            tol = 1e-9 if precision == "f64" else 2e-3
            assert abs(got[-1] - ref[-1]) <= tol * ref[-1]
            if coef == "one":
                assert 0.9 < got[-1] < 1.05
            else:
                assert got[-1] > 1e5
