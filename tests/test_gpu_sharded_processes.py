"""Two processes on the one GPU of the test box run the partitioned DDH (neighbour exchange of trace slots, GMRES
with reduced inner products) with the real HIP engine and must reproduce the single-process solve.  The RCCL
transport itself needs two GPUs; here the same host logic runs over gloo with host-staged payloads."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("precision,overlap,grid", [("f64", False, None), ("f32", True, None), ("f32", False, "2x1"), ("f32", True, "2x1")])
def test_two_processes_neighbour_exchange(cuda, tmp_path, precision, overlap, grid):
    """grid "2x1": the ranks take the left and the right half of the block grid (SURVEY 8e's gx x gy rectangles), so a rank's
    subdomains are half of every block row -- not one range: the local solves go through the listed one-launch entry point."""
    import torch

    sys.path.insert(0, str(ROOT / "tests"))
    import sharded_worker as W

    import cuddhelmholtz_amd as cd

    nx, nb, world = 32, 4, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "sharded_worker.py"), str(nx), str(nb), precision,
                               "1" if overlap else "0", str(tmp_path)] + ([grid] if grid else []), env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)

    dev, fem, F, f = W.problem(nx, nb, precision)
    n = F.size()
    b = torch.zeros(n, dtype=F.trace_dtype, device=dev)
    lam = torch.zeros_like(b)
    u = torch.zeros(2 * fem.size(), dtype=torch.float64, device=dev)
    F.rhs(f, b)
    out = cd.gmres(n, lam, F, b, 10, 3, 0.0)
    F.postprocess(lam, f, u)
    rel = lambda a, r: float((a - r).norm() / r.norm())  # noqa: E731
    for r in range(world):
        got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert got["zero_outside"]
        assert torch.equal(got["b"], b.cpu())  # traces are copied between ranks, never summed: bitwise
        assert got["nmv"] == out.num_matvec
        # inner products are summed in a different order, and the single-process path uses the fused MGS chain
        assert rel(got["lam"].double(), lam.cpu().double()) < (1e-9 if precision == "f64" else 2e-3)
        assert rel(got["u"], u.cpu()) < (1e-9 if precision == "f64" else 2e-3)
        boundary, interior = got["ranges"]
        assert boundary and interior
        if grid:  # 8 x 8 blocks, left / right halves: one boundary column and three interior columns per block row
            assert len(boundary) == 8 and len(interior) == 8


def run_bench_two_ranks(extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--rehearse-gloo", "--nx", "64", "--steps", "3",
           "--warmup", "1", "--no-roofline"] + extra
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("split,grid", [(False, "strips"), (True, "strips"), (True, ""), (False, "1x2")])
def test_bench_two_rank_rehearsal(cuda, split, grid):
    """bench.py's N > 1 code path (partition, neighbour exchange chosen after its start-up cross-check against the
    all-reduce assembly, reduced inner products, max-over-ranks timing, one JSON line from rank 0) with two ranks
    sharing the GPU over gloo; with and without the split schedule (boundary subdomains as one listed launch with issue
    priority on a second stream -- the start-up cross-check then compares THAT assembly bitwise with the all-reduce one).
    grid "" = bench.py's default for the rank count (2x1 for two ranks).  The N > 1 line is self-sufficient:
    exchange kind, fell_back, bytes exchanged, per-rank action time, and the partitioned global apply as `roofline_sharded`
    with its fraction of N x 8 TB/s.  The printed rates are not measurements."""
    import json

    r = run_bench_two_ranks((["--overlap"] if split else []) + (["--rank-grid", grid] if grid else []))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "strong" and out["config"]["finite"]
    assert "partitioned by slot ownership" in out["config"]["sharding"] and "fell back" not in out["config"]["sharding"]
    assert ("split schedule" in out["config"]["sharding"]) == split
    assert out["rank_grid"] == (grid or "2x1")
    assert ("rank grid" in out["config"]["sharding"]) == (grid != "strips")
    assert "cpu_baseline" not in out  # N = 1 only
    assert out["exchange"] == "neighbour" and out["fell_back"] is False
    xb, am = out["exchange_bytes"], out["action_ms"]
    # 64 x 64 elements = 16 x 16 subdomains: the cut between two ranks crosses 16 subdomain edges of 13 trace dofs, minus the
    # slots lost to the cross-point quirk at each of the 15 interior cross points on the cut (2 per cross point on a horizontal cut,
    # 1 on a vertical one: the pair order follows the global edge numbering), lambda and mu halves, 4 bytes
    lost = 1 if out["rank_grid"] == "2x1" else 2
    assert xb["per_action_rank_max"] == xb["per_action_rank_min"] == 2 * (16 * 13 - lost * 15) * 4
    assert xb["per_action_all_ranks"] == 2 * xb["per_action_rank_max"]
    assert 0 < am["min"] <= am["mean"] <= am["max"] and 0 < am["local_solves_only_min"] <= am["local_solves_only_max"]
    rs = out["roofline_sharded"]
    assert "error" not in rs, rs
    assert rs["peak"] == 2 * 8000.0 and rs["n_gpus"] == 2 and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-12 and rs["achieved"] > 0
    sc = out["stable_coefficient"]
    assert sc["finite"] and sc["value"] > 0 and sc["exchange"] == "neighbour"
    assert "disk" in out["config"]["workload"] and out["config"]["coefficient"] == "disk"


def test_bench_neighbour_fallback_is_loud(cuda):
    """When the neighbour exchange fails its start-up cross-check (injected on one rank): with the default `--exchange auto` the
    run continues on the all-reduce assembly and SAYS so at top level (`fell_back`, `exchange`); with an explicit
    `--exchange neighbour` it ends non-zero and prints no result line."""
    import json

    r = run_bench_two_ranks(["--inject-neighbour-failure", "--no-sharded-apply", "--no-stable-coefficient"])
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["fell_back"] is True and out["exchange"] == "allreduce" and "fell back" in out["config"]["sharding"]
    assert out["exchange_bytes"]["per_action_rank_max"] == out["config"]["n_traces"] * 4
    assert "roofline_sharded" not in out and "stable_coefficient" not in out
    r = run_bench_two_ranks(["--inject-neighbour-failure", "--exchange", "neighbour", "--no-sharded-apply", "--no-stable-coefficient"])
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "--exchange neighbour was requested" in r.stderr


def test_partitioned_helmholtz_replay_in_one_process(cuda):
    """Three ranks of the partitioned global Helmholtz operator replayed in one process (real fused kernels on the
    sub-meshes, messages delivered by hand) against the single-GPU fused operator."""
    import torch

    sys.path.insert(0, str(ROOT / "tests"))
    import sharded_worker as W

    import cuddhelmholtz_amd as cd
    from cuddhelmholtz_amd.dist import ShardedHelmholtz

    nx, nb, world = 24, 4, 3
    dev, omega, mesh, fem, fs, a2, ax, b = W.helmholtz_problem(nx, nb)
    to = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    A = cd.HelmholtzOperator(omega, to(a2), to(ax), fem, fs)
    xg = to(np.random.default_rng(3).standard_normal(2 * fem.size()))
    ref = torch.empty_like(xg)
    A.action(xg, ref)
    ranks = [ShardedHelmholtz(cd, omega, a2, ax, mesh, fem, fs, r, world, device=dev) for r in range(world)]
    assert sum(int(r.part.owned.size) for r in ranks) == fem.size()
    xs = [r.scatter(xg) for r in ranks]
    out_x = [r.pack_x(x) for r, x in zip(ranks, xs)]
    for r, x in zip(ranks, xs):
        r.unpack_x(x, {s: out_x[s][r.rank] for s in r.halo_from})
    ys = []
    for r, x in zip(ranks, xs):
        y = torch.empty_like(x)
        r.op.action(x, y)
        ys.append(y)
    out_y = [r.pack_y(y) for r, y in zip(ranks, ys)]
    for r, y in zip(ranks, ys):
        r.unpack_y(y, {s: out_y[s][r.rank] for s in r.own_to})
    got = torch.zeros_like(ref)
    for r, y in zip(ranks, ys):
        world_1 = r.world
        r.world = 1  # gather() without a process group: just the owned entries of this rank
        got += r.gather(y)
        r.world = world_1
    assert float((got - ref).norm() / ref.norm()) < 1e-13


def test_two_processes_partitioned_helmholtz(cuda, tmp_path):
    """Two processes sharing the GPU: halo exchanges over gloo, GMRES with reduced inner products on the partitioned
    vectors, against the single-process operator and solve."""
    import torch

    sys.path.insert(0, str(ROOT / "tests"))
    import sharded_worker as W

    import cuddhelmholtz_amd as cd

    nx, nb, world = 24, 4, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "sharded_worker.py"), "helmholtz", str(nx), str(nb), str(tmp_path)],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)

    dev, omega, mesh, fem, fs, a2, ax, b = W.helmholtz_problem(nx, nb)
    to = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    A = cd.HelmholtzOperator(omega, to(a2), to(ax), fem, fs)
    bg = to(b)
    Ab = torch.empty_like(bg)
    A.action(bg, Ab)
    x = torch.zeros_like(bg)
    out = cd.gmres(bg.numel(), x, A, bg, 10, 3, 0.0)
    rel = lambda a, r: float((a - r).norm() / r.norm())  # noqa: E731
    n_own = n_halo = 0
    for r in range(world):
        got = torch.load(tmp_path / f"rank{r}.pt", weights_only=True)
        assert got["halo_zero"] and got["nmv"] == out.num_matvec
        assert rel(got["Ab"], Ab.cpu()) < 1e-13
        assert rel(got["x"], x.cpu()) < 1e-9
        n_own += got["n_own"]
        n_halo += got["n_loc"] - got["n_own"]
    assert n_own == fem.size() and n_halo > 0  # shared dofs are owned by the lowest rank, the other holds them as halo
