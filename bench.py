#!/usr/bin/env python3
"""Benchmark of the CuDDHelmholtz hot path on MI355X.

Workload (BASELINE.json metric): DDH-GMRES at omega = 32 pi on a 1024 x 1024 structured
square, n_basis = 4 ("p = 4" in the vocabulary of the reference's tests, Basis(p); the only
reading reference DDH supports), fp32 local solves (the reference's precision), GMRES(20).

  step   = one Arnoldi step of GMRES(20) on the substructured operator: one DDH::action
           (65,536 WaveHoltz subdomain solves, sharded over the ranks) + modified Gram-Schmidt
           against the current Krylov basis (k cycles 0..19), issued through the same C-ABI calls
           the product's gmres() makes (fused MGS stages, coefficients on the device).
  value  = 2 * g_ndof * steps / seconds   [DoF.iter/s], whole job, MAX time over ranks.
  gmres_call = one REAL call of the product's gmres() (cuddh_gmres_ddh; with N > 1 the sharded
           native Arnoldi with the all-reduce hook) on the same system, m = 20, one restart cycle,
           tol 0: SURVEY 8d's definition 2 * g_ndof * num_matvec / t_gmres with num_matvec as the
           solver counts it (initial residual + 20 Arnoldi steps + true residual = 22), restart
           bookkeeping included.  Reported beside `value`; the two must agree to a few per cent.
  --gpus N > 1 without a launcher environment: bench.py starts its N ranks itself
           (python -m torch.distributed.run, fresh child processes, before anything touches the GPU)
           and relays rank 0's line; it never prints an N = 1 number for an N > 1 request.
  N > 1  = subdomains split into contiguous ranges, one per rank (total work fixed -> "strong"
           scaling).  Default: trace/Krylov vectors partitioned by slot ownership, the traces a
           rank writes for another rank's subdomains go to that (neighbouring) rank by grouped
           RCCL send/recv, every inner product is all-reduced (cuddhelmholtz_amd/dist.py).  The
           assembly is cross-checked at start-up against the replicated-vector form (one all-reduce
           of the 27 MB trace vector per step, --exchange allreduce), which is also the fallback.

  coefficient: the headline runs the reference example's own two-valued disk coefficient (examples/DDH.cpp:74-83, a = 0.2
           inside r < 0.25).  In BASELINE's regime (32 elements per wavelength, nt = 5120) the reference's time stepping is
           UNSTABLE with it (|T| ~ 1e7: tests/test_baseline_regime.py), so fp32 results carry no meaningful digits there; the time
           per step does not depend on the values.  `stable_coefficient` therefore times the same K steps with a = 1, where the
           local solves are non-expansive and fp32 agrees with the fp64 oracle to 1e-6 (same test file).
  N > 1 fields: `exchange` (neighbour | allreduce), `fell_back`, `exchange_bytes`, `rank_grid`, `action_ms` (per-rank local
           solves + exchange, min / max over ranks), `roofline_sharded` (the partitioned global operator apply, fraction of
           N x 8 TB/s).  An explicit `--exchange neighbour` that falls back ends non-zero.

Extra objects on the same JSON line:
  roofline     = the global operator apply (fused complex Helmholtz apply on the same mesh, in the
                 GENERAL-geometry layout), the HBM-bound kernel of the path: algorithmic bytes /
                 measured launch time; roofline.affine = labelled timing of the uniform-mesh form.
  ddh_kernel   = the dominant kernel of `value` (not HBM-bound): fp32 FLOP rate vs vector peak.
  cpu_baseline = the oracle's restatement of the local solves timed on the host cores on a
                 bounded sample of subdomains (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
FP32_VECTOR_PEAK_TF = 157.3  # MI355X_MICROARCH.md chip-level parameters


def ddh_flops_per_subdomain_step(nb: int, nel: int) -> float:
    """fp32 FLOPs of one RK2 time step of one subdomain as the algorithm is defined
    (source/DDH.cpp:60-109, 247-292): two stiffness sweeps + the update, per element node."""
    nodes = nb * nb * nel * nel
    sweep = nodes * (2 * 2 * nb + 6 + 2 * 2 * nb + 1)  # derivatives, flux, test functions, assembly add
    update = nodes * 26
    return 2.0 * sweep + update


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as fresh child processes through
    torch.distributed.run and relay their output.  Runs before torch or the native library is imported in this process --
    nothing here touches the GPU, and the process is never replaced (no exec)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        try:
            rec = json.loads(ln)
        except ValueError:
            print(ln, file=sys.stderr)
            continue
        if isinstance(rec, dict) and "metric" in rec:
            line = rec
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n_gpus}-rank run failed (exit code {proc.returncode}); no result line", file=sys.stderr)
        return proc.returncode or 1
    if line.get("n_gpus") != n_gpus:
        print(f"bench.py: asked for {n_gpus} GPUs but the run reports n_gpus={line.get('n_gpus')}", file=sys.stderr)
        return 1
    print(json.dumps(line), flush=True)
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=1024, help="elements per side (default: the metric's 1024)")
    ap.add_argument("--nb", type=int, default=4)
    ap.add_argument("--kernel", type=int, default=0, help="DDH kernel: 0 auto, 1 workgroup, 2 wavefront, 3 wavefront with DPP-folded FMAs")
    ap.add_argument("--exchange", choices=("auto", "neighbour", "allreduce"), default="auto",
                    help="N > 1: partitioned trace vectors with neighbour send/recv, or replicated vectors with one all-reduce.  auto "
                         "(default) = neighbour, falling back to allreduce if the start-up cross-check fails (reported as "
                         "fell_back); an explicit `neighbour` that fails its check ends the run non-zero instead")
    ap.add_argument("--coefficient", choices=("disk", "one"), default="disk",
                    help="disk: the example's two-valued coefficient (examples/DDH.cpp:74-83); one: a = 1")
    ap.add_argument("--no-stable-coefficient", action="store_true", help="skip the second timed loop with a = 1")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 host logic on ONE GPU: all ranks share cuda:0, process group on gloo with host-staged "
                         "payloads (RCCL needs one GPU per rank).  Not a measurement.")
    ap.add_argument("--sharded-apply", action=argparse.BooleanOptionalAction, default=True,
                    help="N > 1 (default on): also time the partitioned global Helmholtz apply (element partition + halo exchanges) "
                         "and report its aggregate rate as `roofline_sharded`, fraction of N x 8 TB/s")
    ap.add_argument("--rank-grid", default="auto", metavar="GXxGY|strips|auto",
                    help="N > 1: ranks own rectangles of the subdomain grid (GX x GY = N).  auto (default): SURVEY 8e's grids "
                         "2x1, 2x2, 2x4 for N = 2, 4, 8, strips of block rows otherwise; strips: contiguous subdomain ranges")
    ap.add_argument("--overlap", action="store_true",
                    help="N > 1: split schedule (boundary subdomains first, on a second stream with issue priority, exchange behind "
                         "them, interior meanwhile); default is exchange-after-solve")
    ap.add_argument("--inject-neighbour-failure", action="store_true", help=argparse.SUPPRESS)  # tests: the start-up cross-check "fails"
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-gmres-call", action="store_true", help="skip the real gmres() call timed beside the step loop")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # (multi-process GPU work on this pool needs dmabuf IPC; the image exports it, a bare launcher environment might not)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import cuddhelmholtz_amd as cd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number for a {args.gpus}-GPU request")
    if world > 1 and not args.rehearse_gloo and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible (one rank per GPU; "
                         "--rehearse-gloo rehearses the host logic on one GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    staged = args.rehearse_gloo
    if staged:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ  # under torch.distributed.run, even with one rank
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if staged:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    cd.use_torch_stream()

    def allred(t, op=None):
        """all-reduce of a device tensor (through host memory in the gloo rehearsal)"""
        op = dist.ReduceOp.SUM if op is None else op
        if staged:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    nx, nb = args.nx, args.nb
    omega = math.pi * nx / 32.0  # 32 pi at 1024, 16 pi at 512 (BASELINE.json configs 3 and 4)
    gmres_m = 20

    # ---------------------------------------------------------------- problem set-up (untimed)
    from cuddhelmholtz_amd import _native as N
    from cuddhelmholtz_amd.dist import NeighbourShardedDDH, ShardedDDH, rank_grid_map

    lib = N.lib
    t_setup = time.time()
    mesh = cd.Mesh2D.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0)
    basis = cd.Basis(nb)
    fem = cd.H1Space(mesh, basis)
    ndof = fem.size()
    f = torch.zeros(2 * ndof, dtype=torch.float64, device=dev)
    cd.linear_functional(fem, cd.GAUSSIANS, f[:ndof], param=omega)  # examples/DDH.cpp:120

    def coefficient(name):
        a = torch.ones(ndof, dtype=torch.float64, device=dev)
        if name == "disk":
            cd.linear_functional(fem, cd.ALPHA_DISK, a)              # examples/DDH.cpp:122-123
            cd.DiagInvMassMatrix(fem).action(a, a)
        return a.cpu().numpy()

    F = cd.DDH(omega, coefficient(args.coefficient), fem, nx, nx, precision="f32", kernel=args.kernel)
    info = F.info()
    n = F.size()
    nd = info["n_domains"]
    ndx = nx // info["nel1d"]
    ndy = nd // ndx

    # N > 1: which subdomains a rank owns.  SURVEY 8e: rectangles of the subdomain grid (2x1, 2x2, 2x4 for 2, 4, 8 ranks: at
    # most four face neighbours, shorter messages than strips); strips of block rows (contiguous ranges) otherwise.
    grid = None
    if world > 1 and args.rank_grid != "strips":
        if args.rank_grid == "auto":
            grid = {2: (2, 1), 4: (2, 2), 8: (2, 4)}.get(world)
        else:
            gx, gy = (int(v) for v in args.rank_grid.lower().split("x"))
            if gx * gy != world:
                raise SystemExit(f"bench.py: --rank-grid {args.rank_grid} does not have {world} ranks")
            grid = (gx, gy)
        if grid and (grid[0] > ndx or grid[1] > ndy):
            grid = None
    dom_rank = rank_grid_map(ndx, ndy, *grid) if grid else None
    grid_name = f"{grid[0]}x{grid[1]}" if grid else ("strips" if world > 1 else "1x1")

    def shard(Fx, check: bool):
        """N > 1: the sharded operator on DDH object Fx.  Default: trace vectors partitioned by slot ownership, traces for other
        ranks sent to the neighbouring ranks (grouped RCCL send/recv), inner products all-reduced.  Fallback (--exchange
        allreduce, or --exchange auto when the start-up cross-check fails): replicated vectors, one all-reduce per action.
        Returns (operator, rhs vector, exchange, note)."""
        sh_all = ShardedDDH(Fx, nd, rank, world, always_reduce=dist.is_initialized(), host_staging=staged)
        bx = torch.zeros(n, dtype=torch.float32, device=dev)
        sh_all.rhs(f, bx)
        if world == 1:
            return sh_all, bx, "none", ""
        if args.exchange == "allreduce":
            return sh_all, bx, "allreduce", ""
        ok, sh_nb, b_nb, note = 0.0, None, None, ""
        try:
            sh_nb = NeighbourShardedDDH(Fx, nd, rank, world, device=dev, host_staging=staged, overlap=args.overlap,
                                        set_stream=cd.use_torch_stream, dom_rank=dom_rank)
            b_nb = torch.zeros_like(bx)
            sh_nb.rhs(f, b_nb)
            # traces are copied, not summed: the assembled vector must be bitwise the all-reduce result
            ok = 1.0 if (not check or torch.equal(sh_nb.full(b_nb), bx)) else 0.0
            if args.inject_neighbour_failure and rank == world - 1:
                ok = 0.0
            if ok == 0.0:
                note = "neighbour exchange disagreed with the all-reduce assembly at start-up; fell back"
        except Exception as e:  # noqa: BLE001 - keep the run alive on the proven path and say so
            note = f"neighbour exchange unavailable ({type(e).__name__}: {e}); fell back"
        flag = torch.tensor([ok], device=dev)
        allred(flag, dist.ReduceOp.MIN)  # every rank takes the same branch
        if float(flag.item()) == 1.0:
            return sh_nb, b_nb, "neighbour", ""
        return sh_all, bx, "allreduce", note or "neighbour exchange failed its start-up check on another rank; fell back"

    torch.cuda.synchronize()
    t_constructors = time.time() - t_setup  # mesh, spaces, coefficient projection, DDH constructor + kernel plan
    sh, b, exchange, exchange_note = shard(F, check=True)
    fell_back = bool(exchange_note)
    partitioned = exchange == "neighbour"
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    if fell_back and args.exchange == "neighbour":
        # asked for explicitly: a silent 27 MB all-reduce per action would not be the path the caller wanted measured
        if rank == 0:
            print(f"bench.py: --exchange neighbour was requested but {exchange_note}", file=sys.stderr)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(3)

    # ---------------------------------------------------------------- Arnoldi steps
    ws_half = lib.cuddh_hip_reduce_ws_bytes() // 2  # the two partial-sum buffers of the fused MGS stages (src/krylov.cpp)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    class Stepper:
        """K Arnoldi steps of GMRES(20) on (Fx, shx) from the normalised right-hand side, exactly as the product's gmres() issues
        them (csrc/src/krylov.cpp, reference source/gmres.cpp:160-179)."""

        def __init__(self, Fx, shx, bx, part):
            self.F, self.sh, self.part = Fx, shx, part
            self.V = torch.zeros((gmres_m + 1, n), dtype=torch.float32, device=dev)
            self.hcol = torch.zeros(gmres_m + 2, dtype=torch.float32, device=dev)
            self.ws = torch.zeros(lib.cuddh_hip_reduce_ws_bytes() // 4, dtype=torch.float32, device=dev)
            self.upd = torch.zeros(n, dtype=torch.float32, device=dev)
            self.h_host = torch.zeros(gmres_m + 2, dtype=torch.float32).pin_memory()
            bb = torch.dot(bx, bx).reshape(1)
            if part:
                allred(bb)
            self.V[0].copy_(bx / torch.sqrt(bb))
            self.step_id = 0

        def step(self) -> None:
            k = self.step_id % gmres_m
            self.step_id += 1
            V, hcol, ws, upd = self.V, self.hcol, self.ws, self.upd
            vk, vk1 = V[k], V[k + 1]
            if world == 1:
                self.F.action(vk, vk1)  # DDH::action: local solves, then v_{k+1} = v_k - T v_k
            else:
                # local solves of this rank's subdomains, then the trace exchange / reassembly
                self.sh.traces(None, vk, upd)
                N.check(lib.cuddh_hip_copy_f32(n, p(vk), p(vk1), st))
                N.check(lib.cuddh_hip_axpby_f32(n, -1.0, p(upd), 1.0, p(vk1), st))
            if self.part:
                # vectors partitioned over the ranks: every coefficient is the all-reduced sum of the local dots (MGS parity mode)
                for j in range(k + 1):
                    N.check(lib.cuddh_hip_dot_f32(n, p(vk1), p(V[j]), p(hcol[j:]), p(ws), st))
                    allred(hcol[j:j + 1])
                    N.check(lib.cuddh_hip_axpby_dev_f32(n, -1.0, p(hcol[j:]), p(V[j]), 1.0, p(vk1), st))
                N.check(lib.cuddh_hip_dot_f32(n, p(vk1), p(vk1), p(hcol[k + 1:]), p(ws), st))
                allred(hcol[k + 1:k + 2])
                hcol[k + 1:k + 2].sqrt_()
                N.check(lib.cuddh_hip_scal_inv_dev_f32(n, p(hcol[k + 1:]), p(vk1), st))
            else:
                # fused stages: stage j applies the projection on v_{j-1} and leaves the partial sums of <w, v_j> for stage j+1
                pa, pb = ws.data_ptr(), ws.data_ptr() + ws_half
                N.check(lib.cuddh_hip_mgs_stage_f32(n, p(vk1), None, p(V[0]), C.c_void_p(pa), C.c_void_p(pa), p(hcol), st))
                for j in range(k + 1):
                    vnext = p(V[j + 1]) if j + 1 < k + 1 else None
                    N.check(lib.cuddh_hip_mgs_stage_f32(n, p(vk1), p(V[j]), vnext, C.c_void_p(pa), C.c_void_p(pb), p(hcol[j:]), st))
                    pa, pb = pb, pa
                N.check(lib.cuddh_hip_mgs_finish_f32(n, p(vk1), C.c_void_p(pa), p(hcol[k + 1:]), st))
            self.h_host[: k + 2].copy_(hcol[: k + 2])  # the Hessenberg column reaches the host once per step (Givens on the host)
            if k + 1 == gmres_m:  # restart: continue from the last basis vector
                V[0].copy_(vk1)

        def timed(self, steps, warmup):
            """W untimed + exactly K timed steps between barrier + synchronize on both sides; MAX over ranks"""
            for _ in range(warmup):
                self.step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            barrier()
            elapsed = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
                allred(tmax, dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            return elapsed, bool(torch.isfinite(self.V).all().item())

    stepper = Stepper(F, sh, b, partitioned)
    elapsed, finite = stepper.timed(args.steps, args.warmup)
    upd = stepper.upd

    value = 2.0 * ndof * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # DDH kernel rate (device time of the timed loop is > 99 % local solves)
    flops_step = ddh_flops_per_subdomain_step(nb, info["nel1d"]) * 5 * info["nt"] * nd
    ddh_tf = flops_step * args.steps / elapsed / 1e12 / world
    # FLOPs the chosen kernel actually executes per subdomain per RK2 step (from its ISA: wave-instructions x 64 lanes x 2)
    executed = {3: 244 * 128.0, 5: 8 * 2 * 16 * 16 * 4 + 80 * 128.0}.get(info["kernel"])
    exec_tf = None if executed is None else executed * 5 * info["nt"] * nd * args.steps / elapsed / 1e12 / world

    coef_text = {"disk": "coefficient = the example's two-valued disk (a = 0.2 for r < 0.25, examples/DDH.cpp:74-83, lumped-projected as in "
                         ":122-123; the reference's local solves are UNSTABLE with it at 32 elements per wavelength, see `stable_coefficient`)",
                 "one": "coefficient a = 1"}[args.coefficient]
    send_idx = getattr(sh, "send_idx", {})
    result = {
        "metric": "DDH-GMRES DoF*iter/s",
        "value": value,
        "unit": "DoF*iter/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"DDH-GMRES(20) Arnoldi steps, omega={omega / math.pi:g}pi, {nx}x{nx} quads uniform_rect, n_basis={nb} "
                        f"(reference tests' Basis(p) reading of p={nb}), {nd} reference-size subdomains "
                        f"({info['nel1d']}x{info['nel1d']} elements each), nt={info['nt']} RK2 steps x 5 WaveHoltz iterations per action, "
                        + coef_text,
            "coefficient": args.coefficient,
            "g_ndof": ndof,
            "n_traces": n,
            "ddh_kernel": {1: "workgroup-per-subdomain", 2: "wavefront-per-subdomain", 3: "wavefront-per-subdomain, DPP-folded FMAs", 4: "wavefront-per-subdomain, DPP-folded FMAs + 4x4x1 MFMA", 5: "wavefront-per-subdomain, dense 16x16 element matrix on MFMA (v_mfma_f32_16x16x4_f32)", 6: "wavefront per two subdomains (n_basis 8), DPP-folded FMAs", 7: "wavefront per two subdomains (n_basis 8), separable sweep"}.get(info["kernel"], str(info["kernel"])),
            "sharding": {"none": "single GPU",
                         "allreduce": f"{world} contiguous subdomain ranges, replicated trace vectors, one RCCL all-reduce of the trace vector per step",
                         "neighbour": f"{world} " + (f"rectangles of the subdomain grid ({grid_name})" if grid else "contiguous subdomain ranges")
                                      + ", trace vectors partitioned by slot ownership, grouped RCCL "
                                      f"send/recv of {sum(i.numel() for i in send_idx.values()) * 4 / 1024:.0f} KiB to "
                                      f"{len(send_idx)} neighbour rank(s) per step (rank 0), all-reduce of each inner product"}[exchange]
                        + (", split schedule (boundary subdomains first with issue priority, exchange behind them)" if (args.overlap and exchange == "neighbour") else "")
                        + (f", rank grid {grid_name}" if (grid and exchange == "neighbour") else "")
                        + (f" [{exchange_note}]" if exchange_note else ""),
            "setup_seconds": round(t_constructors, 3),  # Mesh2D + H1Space + load vector / coefficient + DDH constructor + plan
            "rhs_and_exchange_check_seconds": round(t_setup - t_constructors, 3),  # DDH::rhs (one pass of local solves) [+ N > 1 start-up check]
            "finite": finite,
            "replicated_per_rank": "every rank builds all subdomains' tables and keeps global-length Krylov vectors (zero outside the "
                                   "slots it owns); neither is sharded" if world > 1 else None,
        },
        "ddh_kernel": {
            "bound": "fp32-valu/lds (not HBM-bound; SURVEY 8d)",
            "achieved": ddh_tf,
            "peak": FP32_VECTOR_PEAK_TF,
            "unit": "TFLOP/s per GPU",
            "frac": ddh_tf / FP32_VECTOR_PEAK_TF,
            "flops_per_action": flops_step,
            "note": "achieved counts the algorithm's FLOPs (sum-factorised sweeps); executed_tflops counts what the kernel issues "
                    "(kernel 5 applies a dense 16x16 element matrix on v_mfma_f32_16x16x4_f32, which runs on the SIMD's fp32 vector lanes "
                    "on gfx950: same peak, no co-execution with VALU -- profiles/r03/mfma_valu_coexec.txt)",
            "executed_tflops": exec_tf,
            "executed_frac": None if exec_tf is None else exec_tf / FP32_VECTOR_PEAK_TF,
        },
    }

    # ---------------------------------------------------------------- N > 1: what the exchange is and what a rank's action costs
    if world > 1:
        my_bytes = float(sum(i.numel() for i in send_idx.values()) * 4) if partitioned else float(n * 4)
        lam_probe = stepper.V[0]
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = 3
        barrier()
        t_act, t_solve = 0.0, 0.0
        for _ in range(reps):
            e0.record()
            if partitioned:
                sh._solve("all", None, lam_probe, upd)  # local solves only (no exchange)
            else:
                F.local_traces(sh.d0, sh.d1, None, lam_probe, upd)
            e1.record()
            sh.traces(None, lam_probe, upd)             # local solves + exchange / reassembly, as in a step
            e2.record()
            torch.cuda.synchronize()
            t_solve += e0.elapsed_time(e1) / reps
            t_act += e1.elapsed_time(e2) / reps
        stats = torch.tensor([t_act, -t_act, t_solve, -t_solve, my_bytes, -my_bytes], dtype=torch.float64, device=dev)
        allred(stats, dist.ReduceOp.MAX)
        tot = torch.tensor([my_bytes, t_act], dtype=torch.float64, device=dev)
        allred(tot)
        mx_act, mn_act, mx_solve, mn_solve, mx_b, mn_b = (float(v) for v in stats.tolist())
        result["exchange"] = exchange
        result["fell_back"] = fell_back
        result["rank_grid"] = grid_name if partitioned else "strips"
        result["exchange_bytes"] = {"per_action_rank_max": mx_b, "per_action_rank_min": -mn_b, "per_action_all_ranks": float(tot[0].item()),
                                    "what": "packed trace values (fp32) a rank sends to its neighbours per DDH::action" if partitioned
                                            else "the replicated trace vector all-reduced per DDH::action (bytes per rank)"}
        result["action_ms"] = {"max": mx_act, "min": -mn_act, "mean": float(tot[1].item()) / world,
                               "local_solves_only_max": mx_solve, "local_solves_only_min": -mn_solve,
                               "what": "per-rank HIP-event time of one sharded DDH::action (local solves of the rank's subdomains + trace "
                                       "exchange), outside the timed loop; min / max over ranks"}
    else:
        result["exchange"], result["fell_back"], result["rank_grid"] = "none", False, "1x1"

    # ---------------------------------------------------------------- the same K steps with a = 1 (stable local solves)
    if args.coefficient != "one" and not args.no_stable_coefficient:
        F1 = cd.DDH(omega, coefficient("one"), fem, nx, nx, precision="f32", kernel=args.kernel)
        sh1, b1, ex1, note1 = shard(F1, check=False)
        st1 = Stepper(F1, sh1, b1, ex1 == "neighbour")
        el1, fin1 = st1.timed(args.steps, args.warmup)
        result["stable_coefficient"] = {
            "coefficient": "a = 1 (local solves non-expansive at 32 elements per wavelength; fp32 kernel vs fp64 oracle 6e-6 on a window of "
                           "this configuration, tests/test_baseline_regime.py)",
            "value": 2.0 * ndof * args.steps / el1, "unit": "DoF*iter/s", "ms_per_step": 1e3 * el1 / args.steps, "steps": args.steps,
            "warmup": args.warmup, "finite": fin1, "exchange": ex1, "ratio_to_value": (2.0 * ndof * args.steps / el1) / value,
        }
        del st1, sh1, F1

    # ---------------------------------------------------------------- one real gmres() call (SURVEY 8d's definition of the metric)
    if not args.no_gmres_call:
        lam = torch.zeros(n, dtype=torch.float32, device=dev)
        barrier()
        t0 = time.perf_counter()
        if world == 1:
            out = cd.gmres(n, lam, F, b, gmres_m, 2, 0.0)  # cuddh_gmres_ddh: the call of examples/DDH.cpp:143
        else:
            def A_sh(xv, yv):
                sh.traces(None, xv, upd)
                torch.sub(xv, upd, out=yv)

            out = cd.gmres(n, lam, A_sh, b, gmres_m, 2, 0.0, reduce=(lambda t: allred(t)) if partitioned else None)
        barrier()
        t_call = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([t_call], dtype=torch.float64, device=dev)
            allred(tmax, dist.ReduceOp.MAX)
            t_call = float(tmax.item())
        result["gmres_call"] = {
            "call": "cuddh::gmres(n_traces, lambda, &DDH, b, m=20, maxit=2, tol=0)" if world == 1 else
                    "cuddh::gmres (native Arnoldi, callback operator = sharded DDH, ScalarReduce hook) m=20, maxit=2, tol=0",
            "num_matvec": out.num_matvec, "num_iter": out.num_iter, "seconds": t_call,
            "value": 2.0 * ndof * out.num_matvec / t_call, "unit": "DoF*iter/s",
            "rel_residual_after_cycle": out.res_norm[-1] / out.res_norm[0],
            "ratio_to_step_loop": (2.0 * ndof * out.num_matvec / t_call) / value,
        }

    # ---------------------------------------------------------------- roofline: the global operator apply
    if rank == 0 and not args.no_roofline:
        result["roofline"] = helmholtz_roofline(cd, torch, dev, fem, mesh, omega, ndof)

    # ---------------------------------------------------------------- the partitioned global apply over the N ranks
    if world > 1 and args.sharded_apply:
        try:
            result_sh = sharded_apply_rate(cd, torch, dist, dev, fem, mesh, omega, ndof, rank, world, staged, allred)
            err = 0.0
        except Exception as e:  # noqa: BLE001
            result_sh = {"error": f"{type(e).__name__}: {e}"}
            err = 1.0
        flag = torch.tensor([err], device=dev)
        allred(flag, dist.ReduceOp.MAX)
        if float(flag.item()) and "error" not in result_sh:
            result_sh = {"error": "the partitioned apply failed on another rank"}
        if rank == 0:
            result["roofline_sharded"] = result_sh

    # ---------------------------------------------------------------- CPU baseline (oracle), N = 1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(F, info, ndof, nb, omega, args.cpu_seconds)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        assert result["n_gpus"] == args.gpus
        print(json.dumps(result), flush=True)


def sharded_apply_rate(cd, torch, dist, dev, fem, mesh, omega, ndof, rank, world, staged, allred):
    """Aggregate algorithmic GB/s of the fused Helmholtz apply partitioned over the ranks (general-geometry layout, a = 1):
    SURVEY 8d bytes of the whole mesh / MAX over ranks of the time per apply, halo exchanges included."""
    from cuddhelmholtz_amd.dist import ShardedHelmholtz

    prev = os.environ.get("CUDDH_PLAN_AFFINE")
    os.environ["CUDDH_PLAN_AFFINE"] = "0"
    try:
        fs = cd.FaceSpace(fem, mesh.boundary_edges())
        A = ShardedHelmholtz(cd, omega, np.ones(ndof), np.ones(fs.size()), mesh, fem, fs, rank, world, device=dev, host_staging=staged)
    finally:
        if prev is None:
            del os.environ["CUDDH_PLAN_AFFINE"]
        else:
            os.environ["CUDDH_PLAN_AFFINE"] = prev
    x = torch.rand(2 * A.n_loc, dtype=torch.float64, device=dev)
    x.index_fill_(0, A.halo_idx, 0.0)
    y = torch.empty_like(x)
    for _ in range(3):
        A.action(x, y)
    torch.cuda.synchronize()
    dist.barrier()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        A.action(x, y)
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=dev)
    allred(t, dist.ReduceOp.MAX)
    t = float(t.item())
    n_elem, nb = mesh.n_elem(), fem.basis.n
    nqS, nqM, nqF = nb + 1, 2 + 3 * nb // 2, 2 + 3 * nb // 2
    b_alg = n_elem * (3 * nqS * nqS * 8 + nqM * nqM * 8 + nb * nb * 4) + ndof * 32 + fs.n_faces() * (nqF * 8 + nb * 4)
    peak = world * HBM_PEAK_GBS
    return {"label": "fused Helmholtz apply partitioned over the ranks (element partition + two halo exchanges per apply), general-geometry layout",
            "bound": "hbm", "achieved": b_alg / t / 1e9, "peak": peak, "unit": "GB/s", "frac": b_alg / t / 1e9 / peak,
            "seconds_per_apply": t, "algorithmic_bytes": b_alg, "n_gpus": world,
            "local_dofs_rank0": A.n_loc, "halo_dofs_rank0": int(A.part.halo.size),
            "halo_exchange_bytes_rank0": int(A.part.halo.size) * 2 * 8 * 2}


def helmholtz_roofline(cd, torch, dev, fem, mesh, omega, ndof):
    """HBM roofline of the fused complex Helmholtz apply (examples/Helmholtz.hpp:28-56 semantics) on the
    benchmark mesh, coefficient a = 1.  Launch time from HIP events on the stream
    the kernels are launched on."""
    faces = mesh.boundary_edges()
    fs = cd.FaceSpace(fem, faces)
    a2 = torch.ones(ndof, dtype=torch.float64, device=dev)
    ax = torch.ones(fs.size(), dtype=torch.float64, device=dev)
    g = torch.Generator(device="cpu").manual_seed(12345)
    x = (2.0 * torch.rand(2 * ndof, generator=g, dtype=torch.float64) - 1.0).to(dev)
    y = torch.empty_like(x)

    def timed(A, reps=30):
        for _ in range(5):
            A.action(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            A.action(x, y)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    # streaming rates of this box, measured in this run on 1 GiB vectors through the product's own BLAS-1 kernels:
    # a device copy (read + write bytes) and a read-only stream (nrm2).  They are reference points, not ceilings: the
    # operator kernels are free to beat the copy kernel (and the real operators do).
    from cuddhelmholtz_amd import _native as N

    n_copy = 1 << 27  # 1 GiB of doubles
    src = torch.empty(n_copy, dtype=torch.float64, device=dev).fill_(1.0)
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.zeros(N.lib.cuddh_hip_reduce_ws_bytes() // 8, dtype=torch.float64, device=dev)
    res = torch.zeros(2, dtype=torch.float64, device=dev)

    def rate(fn, nbytes, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(reps):
            fn()
        c1.record()
        torch.cuda.synchronize()
        return nbytes * reps / (c0.elapsed_time(c1) * 1e-3) / 1e9

    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    copy_gbs = rate(lambda: N.check(N.lib.cuddh_hip_copy_f64(n_copy, vp(src), vp(dst), st)), 2.0 * 8.0 * n_copy)
    read_gbs = rate(lambda: N.check(N.lib.cuddh_hip_nrm2_f64(n_copy, vp(src), vp(res), vp(ws), st)), 8.0 * n_copy)
    del src, dst

    # the roofline figure is stated on the GENERAL-geometry layout: the plan is told not to exploit that a
    # uniform_rect mesh has one metric tensor for all elements; the affine form is timed separately and labelled
    prev = os.environ.get("CUDDH_PLAN_AFFINE")
    os.environ["CUDDH_PLAN_AFFINE"] = "0"
    A = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    t_ref = timed(A)
    # the same plan on vectors in ITS OWN ordering (pairs (u, v), a patch's owned dofs contiguous: cuddh_hip_helmholtz_apply_native --
    # what HelmholtzOperator::gmres iterates on; same arithmetic, bitwise the same numbers, tests/test_gpu_parity.py)
    t_nat = None
    if A.has_native():
        z, zy = torch.empty_like(x), torch.empty_like(x)
        A.to_native(x, z)

        class _Native:
            def action(self, a, b):
                A.action_native(z, zy)

        t_nat = timed(_Native())
        del z, zy
    t = t_nat if t_nat is not None else t_ref
    # the two real operators the fused apply is made of, alone, same general layout (n_elem (metric + nb^2 4) + ndof 16)
    nb_, ne_ = fem.basis.n, mesh.n_elem()
    xr, yr = x[:ndof], y[:ndof]
    singles = {}
    for name, op, comps, nq in (("StiffnessMatrix::action", cd.StiffnessMatrix(fem), 3, nb_ + 1),
                                ("MassMatrix::action (weighted)", cd.MassMatrix(fem, a2), 1, 1 + 3 * nb_ // 2 + 1)):
        for _ in range(5):
            op.action(xr, yr)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(30):
            op.action(xr, yr)
        s1.record()
        torch.cuda.synchronize()
        ts = s0.elapsed_time(s1) * 1e-3 / 30
        bs = ne_ * (comps * nq * nq * 8 + nb_ * nb_ * 4) + ndof * 16
        singles[name] = {"kernel": op.kernel(), "seconds_per_apply": ts, "algorithmic_bytes": bs, "achieved": bs / ts / 1e9, "unit": "GB/s",
                         "frac": bs / ts / 1e9 / HBM_PEAK_GBS}
        del op
    os.environ["CUDDH_PLAN_AFFINE"] = "1"
    A_aff = cd.HelmholtzOperator(omega, a2, ax, fem, fs)
    affine = None
    if A_aff.fused() and A_aff.bytes_affine() > 0:
        t_aff = timed(A_aff)
        affine = {"label": "affine form (uniform mesh: one copy of the stiffness metric, scalar loads) -- NOT the roofline figure",
                  "seconds_per_apply": t_aff, "affine_bytes": A_aff.bytes_affine(),
                  "achieved_affine_bytes": A_aff.bytes_affine() / t_aff / 1e9, "unit": "GB/s",
                  "frac_of_peak_affine_bytes": A_aff.bytes_affine() / t_aff / 1e9 / HBM_PEAK_GBS}
    if prev is None:
        del os.environ["CUDDH_PLAN_AFFINE"]
    else:
        os.environ["CUDDH_PLAN_AFFINE"] = prev
    b_alg = A.bytes_per_apply(False)
    gbs = b_alg / t / 1e9
    # HBM traffic per apply from the PMC counters: collected in separate rocprofv3 --pmc passes (bench.py cannot run the
    # profiler on itself) by profiles/tools/pmc_traffic.sh and committed under profiles/rNN/helm_pmc_traffic.json.  It is
    # quoted only when that record was taken for this mesh AND this kernel instantiation; `traffic_source` says where it
    # comes from, so a stale record cannot pass for a measurement of this run.
    traffic, traffic_source = None, "no committed PMC record for this mesh, kernel and vector ordering"
    nx_now = int(round(math.sqrt(mesh.n_elem())))
    kernel_now = A.kernel()
    ordering_now = "native" if t_nat is not None else "reference"
    for pmc in sorted((ROOT / "profiles").glob("r*/helm_pmc_traffic.json"), reverse=True):
        rec = json.loads(pmc.read_text())
        if rec.get("nx") == nx_now and rec.get("nb") == fem.basis.n and rec.get("kernel") == kernel_now and rec.get("ordering", "reference") == ordering_now:
            traffic = rec["traffic_bytes_per_apply"]
            traffic_source = {"file": str(pmc.relative_to(ROOT)), "kernel": rec["kernel"], "ordering": ordering_now, "collected_at_commit": rec.get("commit"),
                              "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections as in "
                                     "MI355X_MICROARCH.md; NOT measured in this run"}
            break
    best_stream = max(copy_gbs, read_gbs)
    ref_gbs = b_alg / t_ref / 1e9
    return {
        "bound": "hbm",
        "kernel": (kernel_now + (" on plan-native vectors + helm_border_native_kernel" if t_nat is not None else " + helm_border_kernel")
                   + " (fused complex Helmholtz apply)") if A.fused() else "unfused operator sequence",
        "vector_ordering": ("plan-native: vectors as (u, v) pairs, a patch's owned dofs contiguous (cuddh_hip_helmholtz_apply_native; what "
                            "HelmholtzOperator::gmres iterates on, permuted once at entry and exit); scored against the SAME algorithmic bytes"
                            if t_nat is not None else "reference: [u; v] in H1Space numbering"),
        # the drop-in entry point (Operator::action on reference-ordered vectors), same plan, same run
        "reference_ordering": {"kernel": kernel_now + " + helm_border_kernel", "seconds_per_apply": t_ref, "achieved": ref_gbs, "unit": "GB/s",
                               "frac": ref_gbs / HBM_PEAK_GBS, "layout_bytes": A.bytes_per_apply(True)},
        "achieved": gbs,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": gbs / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "algorithmic_bytes": b_alg,
        "layout_bytes": A.bytes_native() if t_nat is not None else A.bytes_per_apply(True),
        "seconds_per_apply": t,
        "complex_dof_per_s": ndof / t,
        # measured in this run (GB/s): what plain streams reach on this box; reference points, not bounds
        "measured_streams": {"device_copy_f64_1GiB": copy_gbs, "read_only_nrm2_f64_1GiB": read_gbs, "best": best_stream,
                             "frac_of_best": gbs / best_stream},
        "affine": affine,
        "single_operators": singles,  # real vectors, general layout, y = Op x through the same plan machinery
    }


def cpu_baseline(F, info, ndof, nb, omega, budget_s):
    """The oracle's restatement of the DDH local solves (oracle/ddh_body.inc, OpenMP over subdomains)
    on the product's own constructor tables, for a bounded sample of subdomains with the full nt."""
    import oracle

    lib = oracle.lib()
    threads = oracle.num_threads()
    nd, nt = info["n_domains"], info["nt"]
    tabs = {k: F.table(k) for k in ("B", "gI", "sI", "D", "G", "m", "gmi", "a", "H", "filter", "cs", "sn")}
    nel = info["nel1d"]
    s_dof = np.full(nd, info["mx_dof"], dtype=np.int32)
    s_fdof = np.full(nd, info["mx_fdof"], dtype=np.int32)
    lam = np.zeros(2 * info["n_lambda"], dtype=np.float32)
    lam[::7] = 1.0
    upd = np.zeros_like(lam)
    pp = lambda arr: arr.ctypes.data_as(C.c_void_p)  # noqa: E731

    def run(count):
        t0 = time.perf_counter()
        lib.orc_ddh_apply_f32(C.c_int(ndof), C.c_int(nd), C.c_int(info["n_lambda"]), C.c_int(nb), C.c_int(nel * nel), C.c_int(info["mx_dof"]),
                              C.c_int(info["mx_fdof"]), C.c_int(nt), C.c_double(omega), C.c_double(info["dt"]), pp(s_dof), pp(s_fdof), pp(tabs["B"]),
                              pp(tabs["gI"]), pp(tabs["sI"]), pp(tabs["D"]), pp(tabs["G"]), pp(tabs["m"]), pp(tabs["gmi"]), pp(tabs["a"]),
                              pp(tabs["H"]), pp(tabs["filter"]), pp(tabs["cs"]), pp(tabs["sn"]), None, None, pp(lam), pp(upd), C.c_int(0), C.c_int(count))
        return time.perf_counter() - t0

    t1 = run(threads)  # calibration: one subdomain per thread
    count = int(max(threads, min(nd, threads * max(1.0, budget_s / max(t1, 1e-3)))))
    count = (count // threads) * threads
    t = run(count)
    sub_per_s = count / t
    value = 2.0 * ndof * (sub_per_s / nd)
    return {
        "value": value,
        "unit": "DoF*iter/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle fp32 local solves of {count} of {nd} subdomains (full nt={nt}, 5 WaveHoltz iterations) in {t:.1f} s, "
                  f"OpenMP over subdomains; extrapolated to one DDH::action over all subdomains; Krylov BLAS-1 excluded",
        "subdomain_solves_per_s": sub_per_s,
    }


if __name__ == "__main__":
    main()
