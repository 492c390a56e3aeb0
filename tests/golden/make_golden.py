"""Generates tests/golden/hotpath_vectors.npz from the CPU oracle (oracle/).

The reference itself cannot run here (CUDA-only kernels), so these
vectors freeze the oracle's outputs, which are pinned to the reference's own
known-answer tests by tests/test_oracle_pins.py.  Inputs are seeded; every array
needed to replay a case is stored.  Run:  python tests/golden/make_golden.py
"""
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import oracle  # noqa: E402
from conftest import load_unstructured_square  # noqa: E402


def helmholtz_case(out, tag, mesh, nb, omega):
    d = oracle.Discretization(mesh, nb)
    rng = np.random.default_rng(2024)
    a2 = 0.5 + rng.random(d.ndof)
    faces = mesh.boundary_edges
    fs = oracle.FaceSpaceO(d, faces)
    ax = 0.5 + rng.random(fs.size)
    x = rng.standard_normal(2 * d.ndof)
    S, M, H = oracle.Stiffness(d), oracle.Mass(d, a2), oracle.FaceMass(fs, ax)
    out[f"{tag}_a2"], out[f"{tag}_ax"], out[f"{tag}_x"] = a2, ax, x
    out[f"{tag}_Sx"] = S.apply(x[: d.ndof])
    out[f"{tag}_Mx"] = M.apply(x[: d.ndof])
    out[f"{tag}_Hx"] = H.apply(x[: fs.size])
    out[f"{tag}_Ax"] = oracle.helmholtz_apply(d, S, M, H, fs, omega, x)


def ddh_case(out, tag, nx, nb):
    omega = 2 * math.pi * nx / 10
    d = oracle.Discretization(oracle.Mesh.uniform_rect(nx, -1.0, 1.0, nx, -1.0, 1.0), nb)
    h_a = d.nodal(oracle.alpha_disk)
    f = np.concatenate([oracle.linear_functional(d, oracle.gaussians(omega)), 0.1 * oracle.linear_functional(d, oracle.mass_poly)])
    O = oracle.DDH(d, nx, nx, omega, h_a, np.float64)
    b = O.rhs(f)
    out[f"{tag}_h_a"], out[f"{tag}_f"], out[f"{tag}_b"] = h_a, f, b
    out[f"{tag}_Tb"] = O.solve(lam=b)[1]
    out[f"{tag}_u"] = O.postprocess(b, f)
    out[f"{tag}_meta"] = np.array([nx, nb, O.t.n_domains, O.t.n_lambda, O.t.nt], dtype=np.int64)
    out[f"{tag}_omega_dt"] = np.array([omega, O.t.dt])


def main():
    out = {}
    xy, elems = load_unstructured_square()
    helmholtz_case(out, "unstructured_nb4", oracle.Mesh(xy, elems), 4, 9.0)
    helmholtz_case(out, "rect12_nb3", oracle.Mesh.uniform_rect(12, -1.0, 1.0, 12, -1.0, 1.0), 3, 5.0)
    ddh_case(out, "ddh_8_4", 8, 4)
    ddh_case(out, "ddh_8_8", 8, 8)
    path = Path(__file__).resolve().parent / "hotpath_vectors.npz"
    np.savez_compressed(path, **out)
    print(path, path.stat().st_size, "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
