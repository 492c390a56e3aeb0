"""cuddhelmholtz_amd -- MI355X-native hot path of arotem3/CuDDHelmholtz.

Matrix-free Stiffness/Mass/FaceMass operator apply, the fused complex Helmholtz
apply, the DDH subdomain local solves and the GMRES driver, as hand-written HIP
kernels for gfx950 behind a C ABI (include/cuddh_hip.h), with the host-side
mirror of the reference's C++ API (csrc/include/cuddh.hpp) and this thin Python
mirror on top.  Importing the package loads libcuddh_amd.so and fails loudly if
it has not been built; nothing here computes on the CPU.
"""
from . import _native  # noqa: F401  (loads the shared library or raises)
from .api import (  # noqa: F401
    ALPHA_DISK,
    ALPHA_DISK_SQ,
    CONSTANT,
    DDH,
    GAUSSIANS,
    MASS_POLY,
    STIFF_FUNC,
    STIFF_NEG_LAPLACIAN,
    Basis,
    DiagInvFaceMassMatrix,
    DiagInvMassMatrix,
    EnsembleSpace,
    FaceMassMatrix,
    FaceSpace,
    H1Space,
    HelmholtzOperator,
    MassMatrix,
    Mesh2D,
    SolverOut,
    StiffnessMatrix,
    device_count,
    face_linear_functional,
    gmres,
    linear_functional,
    nodal_values,
    quadrature,
    use_torch_stream,
)
