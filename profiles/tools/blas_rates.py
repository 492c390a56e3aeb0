"""Streaming rates of the BLAS-1 kernels through the C ABI (bytes moved / time): copy, axpby, dot, and one fused MGS
stage, on vectors of n doubles.  usage: blas_rates.py [n]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import cuddhelmholtz_amd as cd  # noqa: E402
from cuddhelmholtz_amd import _native as N  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
dev = torch.device("cuda:0")
cd.use_torch_stream()
lib = N.lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.rand(n, dtype=torch.float64, device=dev)
z = torch.rand(n, dtype=torch.float64, device=dev)
res = torch.zeros(4, dtype=torch.float64, device=dev)
ws = torch.zeros(lib.cuddh_hip_reduce_ws_bytes() // 8, dtype=torch.float64, device=dev)
half = ws.numel() // 2


def rate(name, f, nbytes, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / reps
    print(f"n={n} {name:28s} {t * 1e6:9.1f} us  {nbytes / t / 1e9:8.1f} GB/s")


rate("copy (1 in, 1 out)", lambda: N.check(lib.cuddh_hip_copy_f64(n, p(x), p(y), st)), 16 * n)
rate("axpby (2 in, 1 out)", lambda: N.check(lib.cuddh_hip_axpby_f64(n, 0.5, p(x), 0.25, p(y), st)), 24 * n)
rate("dot (2 in)", lambda: N.check(lib.cuddh_hip_dot_f64(n, p(x), p(y), p(res), p(ws), st)), 16 * n)
rate("mgs stage (3 in, 1 out)", lambda: N.check(lib.cuddh_hip_mgs_stage_f64(n, p(z), p(x), p(y), p(ws), p(ws[half:]), p(res), st)), 32 * n)
rate("torch copy_ (reference)", lambda: y.copy_(x), 16 * n)
