"""ctypes binding of libcuddh_amd.so (include/cuddh_hip.h + include/cuddh_capi.h).

The library is the product: there is no Python or CPU fallback.  Importing this
module without a built library raises; calling a compute entry point without a
GPU fails inside HIP with its error string.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = Path(__file__).resolve().parent / "lib" / "libcuddh_amd.so"
HEADERS = [ROOT / "include" / "cuddh_hip.h", ROOT / "include" / "cuddh_capi.h"]


class NativeLibraryMissing(RuntimeError):
    pass


def declared_symbols() -> list[str]:
    """Every function name declared in include/*.h (used by the export test)."""
    names: list[str] = []
    for h in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        for m in re.finditer(r"\b(cuddh_[a-z0-9_]+)\s*\(", text):
            n = m.group(1)
            if n not in names and n != "cuddh_action_cb":
                names.append(n)
    return names


def _load() -> C.CDLL:
    # developer knob for same-box A/B runs of two builds of THIS library (profiles/tools/build_variant.py): a file name under
    # cuddhelmholtz_amd/lib/.  Never a fallback: a missing file raises like the default one.
    global LIB_PATH
    import os

    if os.environ.get("CUDDH_AMD_LIBRARY_VARIANT"):
        LIB_PATH = LIB_PATH.parent / os.environ["CUDDH_AMD_LIBRARY_VARIANT"]
    if not LIB_PATH.exists():
        raise NativeLibraryMissing(
            f"{LIB_PATH} is missing: build it with `python -m cuddhelmholtz_amd.build` "
            "(or __graft_entry__.build()). There is no fallback path."
        )
    return C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)


lib = _load()

vp, ci, cd, cf, cs = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t
cp = C.c_char_p


class SolverResult(C.Structure):
    _fields_ = [("success", ci), ("num_iter", ci), ("num_matvec", ci), ("n_res", ci)]


class DdhDesc(C.Structure):
    _fields_ = [
        ("g_ndof", ci), ("n_domains", ci), ("n_lambda", ci), ("nb", ci), ("nel1d", ci), ("mx_dof", ci),
        ("mx_fdof", ci), ("nt", ci), ("omega", cd), ("dt", cd),
        ("s_dof", vp), ("s_fdof", vp), ("B", vp), ("gI", vp), ("sI", vp), ("D", vp), ("G", vp), ("m", vp),
        ("gmi", vp), ("a", vp), ("H", vp), ("wh_filter", vp), ("cs", vp), ("sn", vp),
    ]


ACTION_CB = C.CFUNCTYPE(None, vp, vp, vp)
REDUCE_CB = C.CFUNCTYPE(None, vp, vp, C.c_int, C.c_int)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


# ---- kernel layer (cuddh_hip.h)
_sig("cuddh_hip_malloc_zeroed", ci, C.POINTER(vp), cs)
_sig("cuddh_hip_free", ci, vp)
_sig("cuddh_hip_copy_h2d", ci, vp, vp, cs)
_sig("cuddh_hip_copy_d2h", ci, vp, vp, cs)
_sig("cuddh_hip_copy_h2d_on", ci, vp, vp, cs, vp)
_sig("cuddh_hip_copy_d2h_on", ci, vp, vp, cs, vp)
_sig("cuddh_hip_copy_d2d", ci, vp, vp, cs, vp)
_sig("cuddh_hip_memset_zero", ci, vp, cs, vp)
_sig("cuddh_hip_stream_sync", ci, vp)
_sig("cuddh_hip_device_sync", ci)
_sig("cuddh_hip_device_count", ci)
_sig("cuddh_hip_current_device", ci)
_sig("cuddh_hip_host_alloc", ci, C.POINTER(vp), cs)
_sig("cuddh_hip_host_free", ci, vp)
_sig("cuddh_hip_copy_d2h_async", ci, vp, vp, cs, vp)
_sig("cuddh_hip_event_create", ci, C.POINTER(vp))
_sig("cuddh_hip_event_record", ci, vp, vp)
_sig("cuddh_hip_event_sync", ci, vp)
_sig("cuddh_hip_event_destroy", ci, vp)
_sig("cuddh_hip_error_string", cp, ci)
_sig("cuddh_hip_reduce_ws_bytes", cs)
_sig("cuddh_hip_axpby_f64", ci, ci, cd, vp, cd, vp, vp)
_sig("cuddh_hip_axpby_f32", ci, ci, cf, vp, cf, vp, vp)
_sig("cuddh_hip_axpby_dev_f64", ci, ci, cd, vp, vp, cd, vp, vp)
_sig("cuddh_hip_axpby_dev_f32", ci, ci, cf, vp, vp, cf, vp, vp)
_sig("cuddh_hip_scal_inv_dev_f64", ci, ci, vp, vp, vp)
_sig("cuddh_hip_scal_inv_dev_f32", ci, ci, vp, vp, vp)
for _n in ("dot", "sqdist"):
    _sig(f"cuddh_hip_{_n}_f64", ci, ci, vp, vp, vp, vp, vp)
    _sig(f"cuddh_hip_{_n}_f32", ci, ci, vp, vp, vp, vp, vp)
_sig("cuddh_hip_mgs_stage_f64", ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_mgs_stage_f32", ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_mgs_finish_f64", ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_mgs_finish_f32", ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_nrm2_f64", ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_nrm2_f32", ci, ci, vp, vp, vp, vp)
for _t in ("f64", "f32", "i32"):
    _sig(f"cuddh_hip_copy_{_t}", ci, ci, vp, vp, vp)
_sig("cuddh_hip_scal_f64", ci, ci, cd, vp, vp)
_sig("cuddh_hip_scal_f32", ci, ci, cf, vp, vp)
_sig("cuddh_hip_fill_f64", ci, ci, cd, vp, vp)
_sig("cuddh_hip_fill_f32", ci, ci, cf, vp, vp)
_sig("cuddh_hip_fill_i32", ci, ci, ci, vp, vp)
_sig("cuddh_hip_diag_scale_f64", ci, ci, ci, cd, vp, vp, vp, vp)
_sig("cuddh_hip_reciprocal_f64", ci, ci, vp, vp)
_sig("cuddh_hip_gather_f64", ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_scatter_add_f64", ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_trace_pack_f32", ci, ci, ci, vp, vp, vp, ci, vp)
_sig("cuddh_hip_trace_pack_f64", ci, ci, ci, vp, vp, vp, ci, vp)
_sig("cuddh_hip_trace_unpack_f32", ci, ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_trace_unpack_f64", ci, ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_halo_pack_f64", ci, ci, ci, vp, vp, vp, ci, vp)
_sig("cuddh_hip_halo_unpack_f64", ci, ci, ci, vp, vp, vp, ci, vp)
_sig("cuddh_hip_csr_sum_f64", ci, ci, vp, vp, vp, vp, ci, vp)
_sig("cuddh_hip_zero_indexed_f64", ci, ci, vp, vp, vp)
_sig("cuddh_hip_element_metrics", ci, ci, ci, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_stiffness_setup", ci, ci, ci, vp, vp, vp, vp)
_sig("cuddh_hip_stiffness_apply", ci, ci, ci, ci, vp, vp, vp, vp, cd, vp, vp, vp)
_sig("cuddh_hip_mass_setup", ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_mass_apply", ci, ci, ci, ci, vp, vp, vp, cd, vp, vp, vp)
_sig("cuddh_hip_diag_mass_setup", ci, ci, ci, ci, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_facemass_setup", ci, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_facemass_apply", ci, ci, ci, ci, vp, vp, vp, cd, vp, vp, vp)
_sig("cuddh_hip_diag_facemass_setup", ci, ci, ci, ci, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_helmholtz_plan_create", ci, C.POINTER(vp), ci, ci, ci, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, vp, vp)
_sig("cuddh_hip_helmholtz_plan_destroy", ci, vp)
_sig("cuddh_hip_helmholtz_apply", ci, vp, cd, vp, vp, vp)
_sig("cuddh_hip_helmholtz_plan_bytes", cs, vp, ci)
_sig("cuddh_hip_helmholtz_plan_has_native", ci, vp)
_sig("cuddh_hip_helmholtz_to_native", ci, vp, vp, vp, vp)
_sig("cuddh_hip_helmholtz_from_native", ci, vp, vp, vp, vp)
_sig("cuddh_hip_helmholtz_apply_native", ci, vp, cd, vp, vp, vp)
_sig("cuddh_hip_helmholtz_plan_describe", ci, vp, vp, ci)
_sig("cuddh_hip_helmholtz_plan_read_stamps", ci, vp, vp, ci)
_sig("cuddh_hip_operator_plan_create", ci, C.POINTER(vp), ci, ci, ci, ci, vp, vp, ci, vp, vp, vp)
_sig("cuddh_hip_operator_plan_apply", ci, vp, cd, ci, vp, vp, vp)
_sig("cuddh_hip_ddh_geom_setup_f32", ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_ddh_geom_setup_f64", ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_ddh_geom_from_corners_f32", ci, ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_ddh_geom_from_corners_f64", ci, ci, ci, vp, vp, vp, vp, vp, vp, vp)
_sig("cuddh_hip_ddh_plan_create", ci, C.POINTER(vp), C.POINTER(DdhDesc), ci, ci)
_sig("cuddh_hip_ddh_plan_destroy", ci, vp)
_sig("cuddh_hip_ddh_plan_kernel", ci, vp)
_sig("cuddh_hip_ddh_plan_set_wh_iters", ci, vp, ci)
_sig("cuddh_hip_ddh_plan_set_wave_priority", ci, vp, ci)
_sig("cuddh_hip_ddh_apply_list_f32", ci, vp, vp, ci, vp, vp, ci, vp, vp, vp)
_sig("cuddh_hip_ddh_apply_list_f64", ci, vp, vp, ci, vp, vp, ci, vp, vp, vp)
_sig("cuddh_hip_ddh_plan_set_vector_layout", ci, vp, vp, ci)
_sig("cuddh_hip_ddh_apply_f32", ci, vp, ci, ci, vp, vp, ci, vp, vp, vp)
_sig("cuddh_hip_ddh_apply_f64", ci, vp, ci, ci, vp, vp, ci, vp, vp, vp)

# ---- handle layer (cuddh_capi.h)
_sig("cuddh_last_error", cp)
_sig("cuddh_set_stream", None, vp)
_sig("cuddh_get_stream", vp)
_sig("cuddh_quadrature", ci, ci, ci, vp, vp)
_sig("cuddh_basis_create", vp, ci)
_sig("cuddh_basis_destroy", None, vp)
_sig("cuddh_basis_eval", ci, vp, ci, vp, vp)
_sig("cuddh_basis_deriv", ci, vp, ci, vp, vp)
_sig("cuddh_mesh_uniform_rect", vp, ci, cd, cd, ci, cd, cd)
_sig("cuddh_mesh_from_vertices", vp, ci, vp, ci, vp)
_sig("cuddh_mesh_load", vp, cp)
_sig("cuddh_mesh_refined", vp, vp, ci)
_sig("cuddh_mesh_partition", ci, vp, ci, vp)
_sig("cuddh_mesh_destroy", None, vp)
for _n in ("n_elem", "n_edges", "n_nodes", "n_boundary_edges"):
    _sig(f"cuddh_mesh_{_n}", ci, vp)
_sig("cuddh_mesh_boundary_edges", ci, vp, vp)
_sig("cuddh_mesh_edges", ci, vp, vp)
_sig("cuddh_mesh_vertices", ci, vp, vp)
_sig("cuddh_mesh_elements", ci, vp, vp)
_sig("cuddh_mesh_min_h", cd, vp)
_sig("cuddh_h1space_create", vp, vp, vp)
_sig("cuddh_h1space_destroy", None, vp)
_sig("cuddh_h1space_size", ci, vp)
_sig("cuddh_h1space_global_indices", ci, vp, vp)
_sig("cuddh_h1space_coordinates", ci, vp, vp)
_sig("cuddh_h1space_global_indices_device", vp, vp)
_sig("cuddh_h1space_coordinates_device", vp, vp)
_sig("cuddh_facespace_create", vp, vp, ci, vp)
_sig("cuddh_facespace_destroy", None, vp)
_sig("cuddh_facespace_size", ci, vp)
_sig("cuddh_facespace_subspace_indices", ci, vp, vp)
_sig("cuddh_facespace_global_indices", ci, vp, vp)
_sig("cuddh_facespace_restrict", ci, vp, vp, vp)
_sig("cuddh_facespace_prolong", ci, vp, vp, vp)
_sig("cuddh_facespace_orth", ci, vp, vp)
_sig("cuddh_ensemble_create", vp, vp, ci, vp)
_sig("cuddh_ensemble_destroy", None, vp)
_sig("cuddh_ensemble_dims", ci, vp, vp)
_sig("cuddh_ensemble_array", ci, vp, cp, vp)
_sig("cuddh_stiffness_create", vp, vp, ci)
_sig("cuddh_mass_create", vp, vp, vp)
_sig("cuddh_diaginv_mass_create", vp, vp, vp)
_sig("cuddh_facemass_create", vp, vp, vp)
_sig("cuddh_diaginv_facemass_create", vp, vp, vp)
_sig("cuddh_helmholtz_create", vp, cd, vp, vp, vp, vp)
_sig("cuddh_operator_destroy", None, vp)
_sig("cuddh_operator_apply", ci, vp, vp, vp)
_sig("cuddh_operator_apply_add", ci, vp, cd, vp, vp)
_sig("cuddh_helmholtz_apply_unfused", ci, vp, vp, vp)
_sig("cuddh_helmholtz_is_fused", ci, vp)
_sig("cuddh_helmholtz_read_stamps", ci, vp, vp, ci)
_sig("cuddh_operator_kernel_name", ci, vp, vp, ci)
_sig("cuddh_helmholtz_bytes", cs, vp, ci)
_sig("cuddh_helmholtz_has_native", ci, vp)
_sig("cuddh_helmholtz_to_native", ci, vp, vp, vp)
_sig("cuddh_helmholtz_from_native", ci, vp, vp, vp)
_sig("cuddh_helmholtz_apply_native", ci, vp, vp, vp)
_sig("cuddh_linear_functional", ci, vp, ci, ci, cd, cd, ci, vp)
_sig("cuddh_face_linear_functional", ci, vp, ci, ci, cd, cd, ci, vp)
_sig("cuddh_nodal_values", ci, vp, ci, cd, vp)
_sig("cuddh_ddh_create", vp, cd, vp, vp, ci, ci, ci, ci)
_sig("cuddh_ddh_destroy", None, vp)
_sig("cuddh_ddh_size", ci, vp)
_sig("cuddh_ddh_info", ci, vp, vp, vp)
_sig("cuddh_ddh_set_wh_iters", ci, vp, ci)
_sig("cuddh_ddh_set_wave_priority", ci, vp, ci)


class MultiGpuResult(C.Structure):
    _fields_ = [("success", ci), ("num_iter", ci), ("num_matvec", ci), ("n_res", ci), ("world", ci), ("used_rccl", ci),
                ("t_setup", cd), ("t_rhs", cd), ("t_gmres", cd), ("t_postprocess", cd), ("bytes_sent_per_action_rank0", C.c_longlong)]


class HelmholtzMultiGpuResult(C.Structure):
    _fields_ = [("success", ci), ("num_iter", ci), ("num_matvec", ci), ("n_res", ci), ("world", ci), ("used_rccl", ci),
                ("t_setup", cd), ("t_apply", cd), ("t_gmres", cd), ("n_loc_max", C.c_longlong), ("n_halo_max", C.c_longlong),
                ("halo_bytes_per_apply_max", C.c_longlong)]


_sig("cuddh_helmholtz_multi_gpu", ci, vp, ci, cd, vp, vp, vp, vp, ci, ci, ci, ci, ci, cd, C.POINTER(HelmholtzMultiGpuResult), vp)
_sig("cuddh_helmholtz_partition_query", ci, vp, vp, vp, ci, ci, ci, ci, vp)
_sig("cuddh_ddh_solve_multi_gpu", ci, ci, ci, cd, vp, vp, vp, ci, ci, ci, cd, ci, C.POINTER(MultiGpuResult), vp)
_sig("cuddh_trace_exchange_query", ci, vp, ci, ci, ci, ci, ci, ci, ci, vp)
_sig("cuddh_ddh_rhs", ci, vp, vp, vp)
_sig("cuddh_ddh_postprocess", ci, vp, vp, vp, vp)
_sig("cuddh_ddh_action", ci, vp, vp, vp)
_sig("cuddh_ddh_local_traces", ci, vp, ci, ci, vp, vp, vp)
_sig("cuddh_ddh_local_traces_listed", ci, vp, vp, ci, vp, vp, vp)
_sig("cuddh_ddh_local_solution_listed", ci, vp, vp, ci, vp, vp, vp, ci)
_sig("cuddh_ddh_local_solution", ci, vp, ci, ci, vp, vp, vp, ci)
_sig("cuddh_ddh_table", C.c_longlong, vp, cp, vp, ci)
_sig("cuddh_gmres_f64", ci, ci, vp, vp, vp, vp, ci, ci, cd, ci, cd, C.POINTER(SolverResult), vp, vp)
_sig("cuddh_gmres_helmholtz", ci, vp, vp, vp, ci, ci, cd, ci, cd, C.POINTER(SolverResult), vp, vp)
_sig("cuddh_gmres_ddh", ci, ci, vp, vp, vp, ci, ci, cd, ci, cd, C.POINTER(SolverResult), vp, vp)
_sig("cuddh_gmres_callback", ci, ci, vp, ACTION_CB, vp, vp, ci, ci, ci, cd, ci, cd, C.POINTER(SolverResult), vp, vp)
_sig("cuddh_gmres_callback_sharded", ci, ci, vp, ACTION_CB, vp, REDUCE_CB, vp, vp, ci, ci, ci, cd, ci, cd, C.POINTER(SolverResult), vp, vp)


def last_error() -> str:
    return lib.cuddh_last_error().decode()


def check(err: int, what: str = "") -> None:
    """Raise on a non-zero return of a kernel-layer (hip error code) call."""
    if err:
        raise RuntimeError(f"{what or 'cuddh_hip call'} failed: {lib.cuddh_hip_error_string(err).decode()} ({err})")


def check_capi(err: int, what: str = "") -> None:
    if err:
        raise RuntimeError(f"{what or 'cuddh call'} failed: {last_error()}")


def handle(h, what: str):
    if not h:
        raise RuntimeError(f"{what} failed: {last_error()}")
    return h
