"""cuddhelmholtz_amd -- MI355X-native hot path of arotem3/CuDDHelmholtz.

Matrix-free Stiffness/Mass/FaceMass operator apply, the fused complex Helmholtz
apply, the DDH subdomain local solves and the GMRES driver, as hand-written HIP
kernels for gfx950 behind a C ABI (include/cuddh_hip.h), with the host-side
mirror of the reference's C++ API (csrc/include/cuddh.hpp) and this thin Python
mirror on top.  Nothing here computes on the CPU: the first use of any name of
the API loads libcuddh_amd.so and raises if it has not been built
(`python -m cuddhelmholtz_amd.build`); only the `build` submodule works without it.
"""
import importlib

_API = (
    "ALPHA_DISK", "ALPHA_DISK_SQ", "CONSTANT", "DDH", "GAUSSIANS", "MASS_POLY", "STIFF_FUNC", "STIFF_NEG_LAPLACIAN",
    "Basis", "DiagInvFaceMassMatrix", "DiagInvMassMatrix", "EnsembleSpace", "FaceMassMatrix", "FaceSpace", "H1Space",
    "HelmholtzOperator", "MassMatrix", "Mesh2D", "SolverOut", "StiffnessMatrix", "device_count", "face_linear_functional",
    "gmres", "linear_functional", "nodal_values", "quadrature", "use_torch_stream",
)
__all__ = list(_API)


def __getattr__(name):
    if name in _API:
        return getattr(importlib.import_module(".api", __name__), name)
    if name in ("api", "_native", "dist", "build"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
