/*
 * cuddh_capi.h -- C handle layer over the host-side C++ mirror of the reference
 * API (cuddh::Mesh2D, Basis, H1Space, FaceSpace, EnsembleSpace, the operator
 * classes, DDH, gmres).  It exists so that non-C++ hosts (the Python package,
 * tests, bench.py) drive exactly the objects a C++ user of cuddh.hpp drives.
 * Plain pointers and sizes only.  Unless a comment says HOST, vector arguments
 * are DEVICE pointers.  Functions returning `int` return 0 on success; functions
 * returning a handle return NULL on failure; cuddh_last_error() then holds the
 * message.  Handles are freed with the matching *_destroy.
 */
#ifndef CUDDH_CAPI_H
#define CUDDH_CAPI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *cuddh_last_error(void);
/* stream used by every subsequent launch of the library (NULL = null stream) */
void cuddh_set_stream(void *stream);
/* the calling thread's current launch stream (NULL = the null stream) */
void *cuddh_get_stream(void);

/* ---- quadrature / basis (reference include/QuadratureRule.hpp, include/Basis.hpp); HOST arrays */
/* type 0 = Gauss-Legendre, 1 = Gauss-Lobatto */
int cuddh_quadrature(int n, int type, double *h_x, double *h_w);
void *cuddh_basis_create(int n);
void cuddh_basis_destroy(void *basis);
int cuddh_basis_eval(void *basis, int m, const double *h_x, double *h_P);  /* P (m, n) */
int cuddh_basis_deriv(void *basis, int m, const double *h_x, double *h_D); /* D (m, n) */

/* ---- mesh (reference include/Mesh2D.hpp) */
void *cuddh_mesh_uniform_rect(int nx, double ax, double bx, int ny, double ay, double by);
void *cuddh_mesh_from_vertices(int n_pts, const double *h_xy, int n_elem, const int *h_elems);
/* mesh ingestion (csrc/include/cuddh/meshio.hpp): the reference's text format (tests/load_unstructured_square.cpp:11-55),
 * `times` rounds of uniform refinement of an existing mesh, and labels for EnsembleSpace (n_parts compact element sets) */
void *cuddh_mesh_load(const char *dir);
void *cuddh_mesh_refined(void *mesh, int times);
int cuddh_mesh_partition(void *mesh, int n_parts, int *h_labels); /* (n_elem) */
void cuddh_mesh_destroy(void *mesh);
int cuddh_mesh_n_elem(void *mesh);
int cuddh_mesh_n_edges(void *mesh);
int cuddh_mesh_n_nodes(void *mesh);
int cuddh_mesh_n_boundary_edges(void *mesh);
int cuddh_mesh_boundary_edges(void *mesh, int *h_out);
/* h_out[8*e..] = {type (0 interior, 1 boundary), node0, node1, elem0, elem1, side0, side1, delta} for every edge */
int cuddh_mesh_edges(void *mesh, int *h_out);
double cuddh_mesh_min_h(void *mesh);
int cuddh_mesh_vertices(void *mesh, double *h_xy);  /* (n_nodes, 2) */
int cuddh_mesh_elements(void *mesh, int *h_elems);  /* (n_elem, 4) corner node ids, counter-clockwise */

/* ---- spaces (reference include/H1Space.hpp) */
void *cuddh_h1space_create(void *mesh, void *basis);
void cuddh_h1space_destroy(void *fem);
int cuddh_h1space_size(void *fem);
int cuddh_h1space_global_indices(void *fem, int *h_I);  /* (nb, nb, n_elem) */
int cuddh_h1space_coordinates(void *fem, double *h_xy); /* (2, ndof) */
const int *cuddh_h1space_global_indices_device(void *fem);
const double *cuddh_h1space_coordinates_device(void *fem);

void *cuddh_facespace_create(void *fem, int n_faces, const int *h_faces);
void cuddh_facespace_destroy(void *fs);
int cuddh_facespace_size(void *fs);
int cuddh_facespace_subspace_indices(void *fs, int *h_I); /* (nb, n_faces) */
int cuddh_facespace_global_indices(void *fs, int *h_proj); /* (fdof) */
int cuddh_facespace_restrict(void *fs, const double *x, double *y);
int cuddh_facespace_prolong(void *fs, const double *x, double *y);
int cuddh_facespace_orth(void *fs, double *x);

/* ---- EnsembleSpace (reference include/EnsembleSpace.hpp) */
void *cuddh_ensemble_create(void *fem, int n_spaces, const int *h_labels);
void cuddh_ensemble_destroy(void *ens);
/* h_dims = {n_spaces, mx_elems, mx_faces, mx_ndof, mx_fdof, n_shared} */
int cuddh_ensemble_dims(void *ens, int *h_dims);
/* name in {"gI","sizes","elements","n_elems","faces","n_faces","sI","fI","pI","fsizes","cmap"}; copies the whole HOST array */
int cuddh_ensemble_array(void *ens, const char *name, int *h_out);

/* ---- operators (reference include/StiffnessMatrix.hpp, MassMatrix.hpp, FaceMassMatrix.hpp) */
void *cuddh_stiffness_create(void *fem, int nq /* 0: default rule */);
void *cuddh_mass_create(void *fem, const double *coef /* DEVICE nodal coefficient or NULL */);
void *cuddh_diaginv_mass_create(void *fem, const double *coef);
void *cuddh_facemass_create(void *fs, const double *coef /* DEVICE FaceSpace vector or NULL */);
void *cuddh_diaginv_facemass_create(void *fs, const double *coef);
/* fused complex Helmholtz operator (examples/Helmholtz.hpp semantics); a2x H1 nodal, ax FaceSpace values, DEVICE */
void *cuddh_helmholtz_create(double omega, const double *a2x, const double *ax, void *fem, void *fs);
void cuddh_operator_destroy(void *op);
int cuddh_operator_apply(void *op, const double *x, double *y);               /* y = A x */
int cuddh_operator_apply_add(void *op, double c, const double *x, double *y); /* y += c A x */
int cuddh_helmholtz_apply_unfused(void *op, const double *x, double *y);
int cuddh_helmholtz_is_fused(void *op);
/* kernel instantiation the operator's action() launches (Helmholtz, Stiffness, Mass; "generic"/"unfused" otherwise) */
int cuddh_operator_kernel_name(void *op, char *buf, int cap);
/* diagnostic: phase time stamps of the last fused apply, see cuddh_hip_helmholtz_plan_read_stamps */
int cuddh_helmholtz_read_stamps(void *op, unsigned long long *h_out, int n_patches);
size_t cuddh_helmholtz_bytes(void *op, int actual); /* actual: 0 / 1 / 2 / 3 as cuddh_hip_helmholtz_plan_bytes */
/* plan-native vector ordering of the fused operator (cuddh_hip.h: cuddh_hip_helmholtz_apply_native); vectors of 2 * ndof doubles */
int cuddh_helmholtz_has_native(void *op);
int cuddh_helmholtz_to_native(void *op, const double *x, double *z);
int cuddh_helmholtz_from_native(void *op, const double *z, double *y);
int cuddh_helmholtz_apply_native(void *op, const double *z_in, double *z_out);

/* ---- load vectors with built-in integrands (device lambdas cannot cross a C ABI).
 * integrand: 0 two Gaussians of examples/DDH.cpp:61-72 (param = omega)
 *            1 disk coefficient of examples/DDH.cpp:74-83
 *            2 tests/mass.cpp:3-7 polynomial     3 tests/stiffness.cpp:16-22 (-Laplacian)
 *            4 tests/stiffness.cpp:6-11 function  5 constant param    6 square of integrand 1
 * nq = 0: collocated Gauss-Lobatto rule of the basis; otherwise Gauss-Legendre with nq points. */
int cuddh_linear_functional(void *fem, int nq, int integrand, double param, double c, int accumulate, double *F);
int cuddh_face_linear_functional(void *fs, int nq, int integrand, double param, double c, int accumulate, double *F);
/* out[i] = integrand(x_i) at the collocation point of every H1 dof */
int cuddh_nodal_values(void *fem, int integrand, double param, double *out);

/* ---- DDH (reference include/DDH.hpp) */
/* h_a HOST nodal coefficient; f64 != 0 selects the fp64 parity variant (double traces);
 * kernel: 0 auto, 1 workgroup-per-subdomain, 2 wavefront-per-subdomain, 3 wavefront with DPP-folded FMAs */
void *cuddh_ddh_create(double omega, const double *h_a, void *fem, int nx, int ny, int f64, int kernel);
void cuddh_ddh_destroy(void *ddh);
int cuddh_ddh_size(void *ddh);
/* h_info = {n_domains, nt, n_lambda, mx_dof, mx_fdof, nel1d, kernel (needs a GPU; -1 if none), is_f64}; *h_dt = time step */
int cuddh_ddh_info(void *ddh, int *h_info, double *h_dt);
/* ---- DDH over the GPUs of one node from one process (csrc/include/cuddh/multigpu.hpp; RCCL send/recv + all-reduce) */
typedef struct cuddh_multi_gpu_result
{
    int success, num_iter, num_matvec, n_res, world, used_rccl;
    double t_setup, t_rhs, t_gmres, t_postprocess;
    long long bytes_sent_per_action_rank0;
} cuddh_multi_gpu_result;
/* rhs -> gmres -> postprocess of examples/DDH.cpp:141-144 on uniform_rect(nx), Basis(nb), fp32 DDH, `world` devices.
 * h_a (ndof), h_f (2 ndof), h_u (2 ndof) HOST in the global numbering; h_res (maxit + 2) receives the residual history.
 * force_rccl: bits 0-1 = transport (0 RCCL for world > 1, 1 RCCL also for one rank, 2 loopback: ranks are threads sharing
 * device 0, a test transport), bit 2 (value 4) = split schedule (boundary subdomains first on a second stream with issue
 * priority, exchange behind them, interior meanwhile), bits 8-15 / 16-23 = gx / gy of a rank grid (gx gy = world: ranks own
 * rectangles of the subdomain grid; 0 = strips of block rows); csrc/include/cuddh/multigpu.hpp. */
int cuddh_ddh_solve_multi_gpu(int nx, int nb, double omega, const double *h_a, const double *h_f, double *h_u, int world, int m,
                              int maxit, double tol, int force_rccl, cuddh_multi_gpu_result *out, double *h_res);
/* ---- the fused Helmholtz operator (examples/Helmholtz.hpp:28-56) partitioned over the GPUs of one node from one process
 * (csrc/include/cuddh/multigpu.hpp: helmholtz_multi_gpu; partition.hpp).  mesh: a Mesh2D handle; h_a2x (ndof), h_ax (FaceSpace of
 * ALL boundary edges, in increasing edge id), h_x / h_y ([u; v], 2 ndof) HOST, global numbering.  maxit == 0: h_y = A h_x, then
 * `reps` timed applies; maxit > 0: GMRES(m) solve of A y = h_x.  transport: 0 RCCL, 1 RCCL also for one rank, 2 loopback ranks
 * (threads sharing device 0; a test transport). */
typedef struct cuddh_helmholtz_multi_gpu_result
{
    int success, num_iter, num_matvec, n_res, world, used_rccl;
    double t_setup, t_apply, t_gmres;
    long long n_loc_max, n_halo_max, halo_bytes_per_apply_max;
} cuddh_helmholtz_multi_gpu_result;
int cuddh_helmholtz_multi_gpu(void *mesh, int nb, double omega, const double *h_a2x, const double *h_ax, const double *h_x, double *h_y,
                              int world, int transport, int reps, int m, int maxit, double tol, cuddh_helmholtz_multi_gpu_result *out,
                              double *h_res /* maxit + 2 or NULL */);
/* lists of HelmholtzPartition::build for `rank` of `world` (host only): which = 0 my_elems, 1 l2g, 2 owned, 3 halo, 4 own_to[peer],
 * 5 halo_from[peer], 6 face_l2g, 7 faces.  Returns the count (-1 on error); h_out may be NULL to ask for the count only. */
int cuddh_helmholtz_partition_query(void *mesh, void *fem, void *fs, int rank, int world, int which, int peer, int *h_out);
/* ownership / send / receive lists of the trace exchange for `rank` of `world` (host only; what both the C++ and the
 * Python multi-GPU hosts use).  which: 0 owned slots, 1 slots sent to `peer`, 2 slots received from `peer`, 3 / 4 the subdomains
 * of the boundary / interior launch of the split schedule.  Returns the
 * count (-1 on error); h_out may be NULL to ask for the count only.  h_B: (mx_fdof, 2, n_domains). */
int cuddh_trace_exchange_query(const int *h_B, int n_domains, int mx_fdof, int n_lambda, int rank, int world, int which, int peer,
                               int *h_out);
/* verification knob: WaveHoltz iterations per local solve (reference: 5, source/DDH.cpp:136; 0 restores it) */
int cuddh_ddh_set_wh_iters(void *ddh, int n);
/* the local solves launched next take issue priority on the device (cuddh_hip_ddh_plan_set_wave_priority); 0 restores */
int cuddh_ddh_set_wave_priority(void *ddh, int high);
/* traces are float for f64 == 0 and double otherwise */
int cuddh_ddh_rhs(void *ddh, const double *f, void *b);
int cuddh_ddh_postprocess(void *ddh, const void *lambda, const double *f, double *u);
int cuddh_ddh_action(void *ddh, const void *x, void *y);
int cuddh_ddh_local_traces(void *ddh, int d0, int d1, const double *f, const void *lambda, void *update);
/* the same for n listed subdomains (d_domains: DEVICE ints, distinct, in range) in one launch */
int cuddh_ddh_local_traces_listed(void *ddh, const int *d_domains, int n, const double *f, const void *lambda, void *update);
int cuddh_ddh_local_solution_listed(void *ddh, const int *d_domains, int n, const void *lambda, const double *f, double *u, int zero_u);
int cuddh_ddh_local_solution(void *ddh, int d0, int d1, const void *lambda, const double *f, double *u, int zero_u);
/* HOST copies of the constructor's tables: name in {"B","gI","sI"} (int) or
 * {"D","G","m","gmi","a","H","filter","cs","sn"} (float / double by f64).  count_only != 0: just return the length. */
long long cuddh_ddh_table(void *ddh, const char *name, void *h_out, int count_only);

/* ---- GMRES (reference include/gmres.hpp).  h_res / h_time: HOST arrays of maxit + 1 entries (may be NULL). */
typedef struct cuddh_solver_result
{
    int success;
    int num_iter;
    int num_matvec;
    int n_res; /* entries written to h_res / h_time */
} cuddh_solver_result;

int cuddh_gmres_f64(int n, double *x, void *op, const double *b, void *precond /* or NULL */, int m, int maxit, double tol,
                    int verbose, double max_seconds, cuddh_solver_result *out, double *h_res, double *h_time);
/* HelmholtzOperator::gmres: x, b in the reference ordering [u; v]; the iteration runs on plan-native vectors when the plan has them */
int cuddh_gmres_helmholtz(void *op, double *x, const double *b, int m, int maxit, double tol, int verbose, double max_seconds,
                          cuddh_solver_result *out, double *h_res, double *h_time);
/* op: a DDH handle (f64 == 0: float vectors, f64 != 0: double vectors) */
int cuddh_gmres_ddh(int n, void *x, void *ddh, const void *b, int m, int maxit, double tol, int verbose, double max_seconds,
                    cuddh_solver_result *out, double *h_res, double *h_time);
/* operator supplied by the caller: cb(ctx, x, y) must compute y = A x on DEVICE vectors on the library stream */
typedef void (*cuddh_action_cb)(void *ctx, const void *x, void *y);
int cuddh_gmres_callback(int n, void *x, cuddh_action_cb cb, void *ctx, const void *b, int is_f64, int m, int maxit, double tol,
                         int verbose, double max_seconds, cuddh_solver_result *out, double *h_res, double *h_time);
/* the same with vectors partitioned over processes (one per GPU): reduce(ctx, d_scalars, count, is_f64) must sum
 * `count` DEVICE scalars over all ranks in place, ordered on the library stream (an RCCL all-reduce) */
typedef void (*cuddh_reduce_cb)(void *ctx, void *d_scalars, int count, int is_f64);
int cuddh_gmres_callback_sharded(int n, void *x, cuddh_action_cb cb, void *ctx, cuddh_reduce_cb reduce, void *reduce_ctx,
                                 const void *b, int is_f64, int m, int maxit, double tol, int verbose, double max_seconds,
                                 cuddh_solver_result *out, double *h_res, double *h_time);

#ifdef __cplusplus
}
#endif
#endif /* CUDDH_CAPI_H */
