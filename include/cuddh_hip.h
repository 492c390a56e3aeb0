/*
 * cuddh_hip.h -- C ABI of the MI355X (gfx950) kernel layer.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an
 * int (0 = success, otherwise the hipError_t value) and takes the HIP stream as
 * an opaque `void*` last argument (NULL = the null stream).  All array
 * arguments are DEVICE pointers unless the name starts with `h_`.  Layouts are
 * column major (first index fastest), exactly the layouts the reference's
 * operator classes already hold in their HostDeviceArray members, so that a
 * maintainer can replace the body of each reference `action()` by one call
 * (see INTEGRATION.md).  Citations are relative to the reference repository
 * arotem3/CuDDHelmholtz.
 *
 * None of these functions allocates, frees or synchronises unless its comment
 * says so (safe to capture in a hipGraph).
 */
#ifndef CUDDH_HIP_H
#define CUDDH_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ runtime */

/* Replaces the cudaMalloc/cudaMemset/cudaMemcpy/cudaFree calls of
 * include/HostDeviceArray.hpp:143,189,259-262,269,296-299.  Blocking. */
int cuddh_hip_malloc_zeroed(void **ptr, size_t bytes);
int cuddh_hip_free(void *ptr);
int cuddh_hip_copy_h2d(void *dst, const void *h_src, size_t bytes);
int cuddh_hip_copy_d2h(void *h_dst, const void *src, size_t bytes);
/* the same, queued on `stream` behind the work already there and complete on return (what HostDeviceArray mirrors and the
 * solver's per-step coefficient copies use: the launch stream may be a non-blocking one) */
int cuddh_hip_copy_h2d_on(void *dst, const void *h_src, size_t bytes, void *stream);
int cuddh_hip_copy_d2h_on(void *h_dst, const void *src, size_t bytes, void *stream);
int cuddh_hip_copy_d2d(void *dst, const void *src, size_t bytes, void *stream);
/* include/linalg.hpp:46-48 (zeros -> cudaMemset) */
int cuddh_hip_memset_zero(void *ptr, size_t bytes, void *stream);
int cuddh_hip_stream_sync(void *stream);
int cuddh_hip_device_sync(void);
/* number of visible devices; 0 if there is no GPU (never fails) */
int cuddh_hip_device_count(void);
/* Pinned host memory, asynchronous device -> host copies and events: what the Arnoldi loop needs to fetch a Hessenberg column
 * while the next matrix-vector product is already queued (csrc/src/krylov.cpp; the reference copies with blocking cudaMemcpy,
 * source/linalg.cpp:67-83).  h_dst of copy_d2h_async must come from host_alloc. */
int cuddh_hip_host_alloc(void **ptr, size_t bytes);
int cuddh_hip_host_free(void *ptr);
int cuddh_hip_copy_d2h_async(void *h_dst, const void *src, size_t bytes, void *stream);
int cuddh_hip_event_create(void **ev);
int cuddh_hip_event_record(void *ev, void *stream);
int cuddh_hip_event_sync(void *ev);
int cuddh_hip_event_destroy(void *ev);
/* the calling thread's current device (hipGetDevice), -1 if there is none; the BLAS-1 wrappers key their reduction scratch by it */
int cuddh_hip_current_device(void);
const char *cuddh_hip_error_string(int err);

/* ------------------------------------------------------------------ BLAS-1
 * source/linalg.cpp:51-201.  `ws` is a caller-owned device workspace of at
 * least cuddh_hip_reduce_ws_bytes() bytes used for the two-stage reduction;
 * `result` is a DEVICE scalar (no host sync, no allocation -- the reference
 * allocates, memsets and D2H-copies a scalar per call, source/linalg.cpp:67-83).
 */
size_t cuddh_hip_reduce_ws_bytes(void);
int cuddh_hip_axpby_f64(int n, double a, const double *x, double b, double *y, void *stream);
int cuddh_hip_axpby_f32(int n, float a, const float *x, float b, float *y, void *stream);
/* y <- (sa * *a_dev) * x + b * y with the coefficient read from device memory
 * (lets the modified Gram-Schmidt loop of source/gmres.cpp:167-172 run without
 * a host round trip per dot product). */
int cuddh_hip_axpby_dev_f64(int n, double sa, const double *a_dev, const double *x, double b, double *y, void *stream);
int cuddh_hip_axpby_dev_f32(int n, float sa, const float *a_dev, const float *x, float b, float *y, void *stream);
/* x <- x / *a_dev   (source/gmres.cpp:179 with H(k+1,k) left on the device) */
int cuddh_hip_scal_inv_dev_f64(int n, const double *a_dev, double *x, void *stream);
int cuddh_hip_scal_inv_dev_f32(int n, const float *a_dev, float *x, void *stream);
int cuddh_hip_dot_f64(int n, const double *x, const double *y, double *result, void *ws, void *stream);
int cuddh_hip_dot_f32(int n, const float *x, const float *y, float *result, void *ws, void *stream);
/* result <- sqrt(sum x[i]^2) */
int cuddh_hip_nrm2_f64(int n, const double *x, double *result, void *ws, void *stream);
int cuddh_hip_nrm2_f32(int n, const float *x, float *result, void *ws, void *stream);
/* result <- sum (x[i]-y[i])^2  (square root taken by the caller; source/linalg.cpp:103-137) */
int cuddh_hip_sqdist_f64(int n, const double *x, const double *y, double *result, void *ws, void *stream);
int cuddh_hip_sqdist_f32(int n, const float *x, const float *y, float *result, void *ws, void *stream);
/* One stage of the modified Gram-Schmidt loop of source/gmres.cpp:167-172 in a single launch:
 *   h = sum of the partial sums `pin` left by the previous stage (= <w, vprev>), stored to *hout;
 *   w <- w - h * vprev;   pout <- per-workgroup partial sums of <w, vnext>  (vnext == NULL: <w, w>).
 * vprev == NULL starts a chain (no update, pin/hout ignored).  pin and pout are the two halves of a workspace of
 * cuddh_hip_reduce_ws_bytes() bytes, used alternately.  `finish` turns the <w,w> partials into the norm
 * (*hout = ||w||) and normalises w (source/gmres.cpp:174-179).  Fixed summation order. */
int cuddh_hip_mgs_stage_f64(int n, double *w, const double *vprev, const double *vnext, const double *pin, double *pout, double *hout, void *stream);
int cuddh_hip_mgs_stage_f32(int n, float *w, const float *vprev, const float *vnext, const float *pin, float *pout, float *hout, void *stream);
int cuddh_hip_mgs_finish_f64(int n, double *w, const double *pin, double *hout, void *stream);
int cuddh_hip_mgs_finish_f32(int n, float *w, const float *pin, float *hout, void *stream);
int cuddh_hip_copy_f64(int n, const double *x, double *y, void *stream);
int cuddh_hip_copy_f32(int n, const float *x, float *y, void *stream);
int cuddh_hip_copy_i32(int n, const int *x, int *y, void *stream);
int cuddh_hip_scal_f64(int n, double a, double *x, void *stream);
int cuddh_hip_scal_f32(int n, float a, float *x, void *stream);
int cuddh_hip_fill_f64(int n, double a, double *x, void *stream);
int cuddh_hip_fill_f32(int n, float a, float *x, void *stream);
int cuddh_hip_fill_i32(int n, int a, int *x, void *stream);
/* y[i] (+)= c * p[i] * x[i]; accumulate != 0 selects +=.  x may alias y.
 * source/MassMatrix.cpp:316-334, source/FaceMassMatrix.cpp:303-321 */
int cuddh_hip_diag_scale_f64(int n, int accumulate, double c, const double *p, const double *x, double *y, void *stream);
/* x[i] <- 1 / x[i]   (source/MassMatrix.cpp:276-279) */
int cuddh_hip_reciprocal_f64(int n, double *x, void *stream);

/* ------------------------------------------------------------------ index maps
 * source/H1Space.cpp:189-219 (FaceSpace::restrict / prolong / orth) */
int cuddh_hip_gather_f64(int n, const int *proj, const double *x, double *y, void *stream);      /* y[i]  = x[proj[i]] */
int cuddh_hip_scatter_add_f64(int n, const int *proj, const double *x, double *y, void *stream); /* y[proj[i]] += x[i] */
int cuddh_hip_zero_indexed_f64(int n, const int *proj, double *x, void *stream);                 /* x[proj[i]] = 0 */
/* y[r] = (accumulate ? y[r] : 0) + sum of x[src[k]], k in [off[r], off[r+1]), added in list order (fixed-order replacement of
 * the atomic adds of source/DDH.cpp:303,306) */
int cuddh_hip_csr_sum_f64(int n_rows, const int *off, const int *src, const double *x, double *y, int accumulate, void *stream);
/* Trace exchange of the multi-GPU DDH path (new: the reference is single-GPU).  A slot t stands for entries t and n_half + t
 * of a trace vector v (lambda and mu halves).  pack: buf[i] = v[slot[i]], buf[n + i] = v[n_half + slot[i]] and, with
 * clear != 0, those entries of v are zeroed (they belong to the receiving rank); unpack: the inverse copy. */
int cuddh_hip_trace_pack_f32(int n, int n_half, const int *slot, float *v, float *buf, int clear, void *stream);
int cuddh_hip_trace_pack_f64(int n, int n_half, const int *slot, double *v, double *buf, int clear, void *stream);
int cuddh_hip_trace_unpack_f32(int n, int n_half, const int *slot, const float *buf, float *v, void *stream);
int cuddh_hip_trace_unpack_f64(int n, int n_half, const int *slot, const double *buf, double *v, void *stream);
/* Halo exchange of the partitioned global operator apply (new: the reference is single-GPU; it partitions the operator of
 * examples/Helmholtz.hpp:28-56).  Entries ids[i] and n_half + ids[i] of a local [u; v] vector v travel as the pair
 * (buf[2 i], buf[2 i + 1]): the messages for several neighbours are contiguous pieces of ONE buffer and one launch packs them all.
 * pack: buf <- v at ids and, with clear != 0, those entries of v are zeroed (partial sums handed to the owner).
 * unpack: v at ids <- buf (add == 0: halo values of x arriving from their owners) or v at ids += buf (add != 0: partial sums of y
 * arriving at the owner; ids of one call are distinct, messages from different senders are added in rank order -> reproducible). */
int cuddh_hip_halo_pack_f64(int n, int n_half, const int *ids, double *v, double *buf, int clear, void *stream);
int cuddh_hip_halo_unpack_f64(int n, int n_half, const int *ids, const double *buf, double *v, int add, void *stream);

/* ------------------------------------------------------------------ element operators (fp64)
 * Shapes: P,D (nq,nb); I (nb,nb,n_elem); J (2,2,nq,nq,n_elem); detJ (nq,nq,n_elem);
 * G (3,nq,nq,n_elem); a (nq,nq,n_elem); w (nq). */

/* source/Mesh2D.cpp:173-227 + source/Element.cpp:5-36: the metric arrays of Mesh2D::ElementMetricCollection on the tensor
 * grid of a 1-D rule q (n points), computed on the device from the elements' corners (2, 4, n_elem; counter-clockwise):
 * J (2,2,n,n,n_elem), detJ (n,n,n_elem), x (2,n,n,n_elem); any output may be NULL. */
int cuddh_hip_element_metrics(int n_elem, int n, const double *corners, const double *q, double *J, double *detJ, double *x, void *stream);
/* source/StiffnessMatrix.cpp:5-38  (setup_geometric_factors) */
int cuddh_hip_stiffness_setup(int n_elem, int nq, const double *w, const double *J, double *G, void *stream);
/* source/StiffnessMatrix.cpp:83-205  y[I] += c * S x */
int cuddh_hip_stiffness_apply(int n_elem, int nq, int nb, const double *P, const double *D, const double *G,
                              const int *I, double c, const double *x, double *y, void *stream);
/* source/MassMatrix.cpp:5-67  (init_mass_matrix); coef may be NULL (=1) */
int cuddh_hip_mass_setup(int n_elem, int nq, int nb, const double *coef, const double *detJ, const double *w,
                         const int *I, const double *P, double *a, void *stream);
/* source/MassMatrix.cpp:137-239  y[I] += c * M x */
int cuddh_hip_mass_apply(int n_elem, int nq, int nb, const int *I, const double *P, const double *a, double c,
                         const double *x, double *y, void *stream);
/* source/MassMatrix.cpp:241-280 (init_diag_mass): op <- 1 / lumped mass.  Zero-fills op first. */
int cuddh_hip_diag_mass_setup(int ndof, int n_elem, int nb, const double *coef, const double *detJ, const double *w,
                              const int *I, double *op, void *stream);
/* source/FaceMassMatrix.cpp:5-49; I (nb,n_faces); detJ (nq,n_faces); a (nq,n_faces) */
int cuddh_hip_facemass_setup(int n_faces, int nb, int nq, const double *w, const double *P, const double *detJ,
                             const double *coef, const int *I, double *a, void *stream);
/* source/FaceMassMatrix.cpp:141-223 */
int cuddh_hip_facemass_apply(int n_faces, int nb, int nq, const double *P, const double *a, const int *I, double c,
                             const double *x, double *y, void *stream);
/* source/FaceMassMatrix.cpp:225-255 (init_diag); op must be zero on entry (the reference relies on that too) */
int cuddh_hip_diag_facemass_setup(int ndof, int n_faces, int nb, const double *w, const double *detJ,
                                  const double *coef, const int *I, double *op, void *stream);

/* ------------------------------------------------------------------ fused complex Helmholtz apply
 * One launch pair for  [u;v] -> [Au;Av]  of examples/Helmholtz.hpp:28-56
 * (2 memsets + 11 launches there).  The plan owns patch-ordered copies of the
 * index map and of the metric arrays (layout in DESIGN.md).  create/destroy
 * allocate, copy and synchronise; apply does not.
 *
 * h_I (nb,nb,n_elem) HOST; h_xy optional HOST element centroids (2,n_elem) used
 * only to order elements into compact patches (NULL = keep element order);
 * G_S (3,nqS,nqS,n_elem), a_M (nqM,nqM,n_elem) DEVICE arrays as produced by
 * cuddh_hip_stiffness_setup / cuddh_hip_mass_setup; boundary-face data:
 * h_fI (nb,n_faces) HOST face -> H1 global index (already composed with the
 * FaceSpace projection), h_face_elem (n_faces) HOST element each face belongs to
 * (Edge::elements[0]), a_F (nqF,n_faces) DEVICE from cuddh_hip_facemass_setup.
 * Returns hipErrorNotSupported (801) for (nb,nqS,nqM) combinations without a
 * specialised kernel; callers then use the separate operators.
 */
typedef struct cuddh_helmholtz_plan cuddh_helmholtz_plan;
int cuddh_hip_helmholtz_plan_create(cuddh_helmholtz_plan **plan, int ndof, int n_elem, int nb, const int *h_I,
                                    const double *h_xy, int nqS, const double *h_PS, const double *h_DS,
                                    const double *G_S, int nqM, const double *h_PM, const double *a_M,
                                    int n_faces, const int *h_fI, const int *h_face_elem, int nqF, const double *h_PF,
                                    const double *a_F);
int cuddh_hip_helmholtz_plan_destroy(cuddh_helmholtz_plan *plan);
/* y = [ S u - w^2 M u - w H v ;  -(S v - w^2 M v + w H u) ],  x = [u;v], y = [Au;Av], each of length ndof */
int cuddh_hip_helmholtz_apply(const cuddh_helmholtz_plan *plan, double omega, const double *x, double *y, void *stream);
/* bytes the plan's apply reads+writes per call: actual == 0 the SURVEY 8d formula of the general-geometry layout,
 * 1 as actually laid out, 2 the "affine" figure when the plan found a metric array identical in every element (uniform
 * meshes: one copy read through scalar loads instead of one per element; 0 otherwise).  CUDDH_PLAN_AFFINE=0 in the
 * environment at plan creation keeps the general form (used to measure the general-geometry roofline on a uniform mesh). */
size_t cuddh_hip_helmholtz_plan_bytes(const cuddh_helmholtz_plan *plan, int actual); /* 3: as laid out for the native apply below */
/* Plan-native vector ordering (not in the reference: its vectors are always [u; v] in H1Space numbering).  For solvers that keep
 * their iteration vectors in the plan's own order -- gmres() only needs an operator and inner products, both invariant under a
 * permutation -- a vector z is an array of ndof (u, v) PAIRS ordered [patch 0's owned dofs | patch 1's ... | patch-border dofs],
 * so the apply reads and writes a patch's owned dofs with contiguous 16-byte accesses and no index list.  has_native: 1 when the
 * plan runs the lane form (helm_lane_kernel: n_basis 2-4, large plans); to_native / from_native are the permutation and its
 * inverse (x, y: [u; v] of length 2 ndof; z: 2 ndof doubles); apply_native is cuddh_hip_helmholtz_apply on native vectors
 * (z_in != z_out).  Same arithmetic in the same order per element and per dof: from_native(apply_native(to_native(x))) is
 * bitwise cuddh_hip_helmholtz_apply(x). */
int cuddh_hip_helmholtz_plan_has_native(const cuddh_helmholtz_plan *plan);
int cuddh_hip_helmholtz_to_native(const cuddh_helmholtz_plan *plan, const double *x, double *z, void *stream);
int cuddh_hip_helmholtz_from_native(const cuddh_helmholtz_plan *plan, const double *z, double *y, void *stream);
int cuddh_hip_helmholtz_apply_native(const cuddh_helmholtz_plan *plan, double omega, const double *z_in, double *z_out, void *stream);
/* which kernel instantiation the plan's apply launches, e.g. "helm_lane_kernel<4,5,8,NT=1,UG=0> pe=64" (tests assert
 * the form they mean to exercise; bench.py reports it instead of re-deriving the size rules) */
int cuddh_hip_helmholtz_plan_describe(const cuddh_helmholtz_plan *plan, char *buf, int cap);
/* diagnostic: with CUDDH_HELM_STAMPS=1 in the environment at plan creation the lane-form kernels record 8 phase time stamps
 * (100 MHz constant clock) per patch; this copies those of the last apply to h_out ([n_patches][8]). */
int cuddh_hip_helmholtz_plan_read_stamps(const cuddh_helmholtz_plan *plan, unsigned long long *h_out, int n_patches);

/* The same plan machinery for ONE real operator -- the bandwidth path of StiffnessMatrix::action
 * (source/StiffnessMatrix.cpp:186-205, kind 0, metric = G (3,nq,nq,n_elem)) and MassMatrix::action
 * (source/MassMatrix.cpp:213-239, kind 1, metric = a (nq,nq,n_elem)):  y = [y +] c * Op x  without atomics
 * and without a zero-fill (accumulate != 0 keeps the reference's action(c,x,y) meaning, 0 is action(x,y)).
 * h_D is ignored for kind 1.  Returns hipErrorNotSupported (801) unless 2 <= nb <= 5 and nq is the rule the
 * reference's constructors pick (nb+1; or 1+3nb/2+1 for the weighted mass); callers then use
 * cuddh_hip_stiffness_apply / cuddh_hip_mass_apply.  destroy/bytes: the helmholtz_plan functions. */
int cuddh_hip_operator_plan_create(cuddh_helmholtz_plan **plan, int kind, int ndof, int n_elem, int nb, const int *h_I,
                                   const double *h_xy, int nq, const double *h_P, const double *h_D, const double *metric);
int cuddh_hip_operator_plan_apply(const cuddh_helmholtz_plan *plan, double c, int accumulate, const double *x, double *y,
                                  void *stream);

/* ------------------------------------------------------------------ DDH local solves
 * source/DDH.cpp:15-58 (init_geom_factors): G (3, nb*nb*mx_elems, n_domains) as
 * Real triples (the reference stores float3), from J (2,2,nb,nb,g_elem). */
int cuddh_hip_ddh_geom_setup_f32(int n_domains, int mx_elems, int g_elem, int nb, const int *n_elems, const int *elems,
                                 const double *w, const double *J, float *G, void *stream);
int cuddh_hip_ddh_geom_setup_f64(int n_domains, int mx_elems, int g_elem, int nb, const int *n_elems, const int *elems,
                                 const double *w, const double *J, double *G, void *stream);

/* The same factors straight from the elements' corners ((2, 4, g_elem), counter-clockwise) and the rule's nodes `points`
 * (nb): the bilinear Jacobian of source/Element.cpp:5-36 is evaluated in the kernel (with the arithmetic of
 * cuddh_hip_element_metrics), so the (2,2,nb,nb,g_elem) table of source/Mesh2D.cpp:173-227 is never built. */
int cuddh_hip_ddh_geom_from_corners_f32(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w,
                                        const double *points, const double *corners, float *G, void *stream);
int cuddh_hip_ddh_geom_from_corners_f64(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w,
                                        const double *points, const double *corners, double *G, void *stream);

/* The arrays DDH holds after its constructor (include/DDH.hpp:55-83,
 * source/DDH.cpp:425-608), in the reference's own layouts.  `real` is float for
 * *_f32 and double for *_f64.  All DEVICE pointers. */
typedef struct cuddh_ddh_desc
{
    int g_ndof;      /* global H1 dofs */
    int n_domains;   /* subdomains (one LDS/register-resident local solve each) */
    int n_lambda;    /* 2 * n_shared */
    int nb;          /* n_basis */
    int nel1d;       /* elements per subdomain edge (NEL); mx_elems = nel1d^2 */
    int mx_dof;      /* leading dimension of gI, m, gmi, a */
    int mx_fdof;     /* leading dimension of B, H */
    int nt;          /* time steps per period */
    double omega;
    double dt;
    const int *s_dof;      /* (n_domains) subdomain sizes          EnsembleSpace::sizes  */
    const int *s_fdof;     /* (n_domains) face-space sizes         EnsembleSpace::fsizes */
    const int *B;          /* (mx_fdof, 2, n_domains) read/write lambda slots, -1 = none  source/DDH.cpp:425-440 */
    const int *gI;         /* (mx_dof, n_domains) permuted subdomain dof -> global dof     source/DDH.cpp:486-497 */
    const int *sI;         /* (nb, nb, mx_elems, n_domains) element node -> permuted subdomain dof  :489-509 */
    const void *D;         /* real (nb, nb) GLL differentiation matrix  :523-528 */
    const void *G;         /* real (3, nb*nb*mx_elems, n_domains)       :530-537 */
    const void *m;         /* real (mx_dof, n_domains) lumped subdomain mass  :541-584 */
    const void *gmi;       /* real (mx_dof, n_domains) global inverse lumped mass :559-591 */
    const void *a;         /* real (mx_dof, n_domains) coefficient      :589 */
    const void *H;         /* real (mx_fdof, n_domains) lumped face mass :593-607 */
    const void *wh_filter; /* real (nt+1)   :370-375 */
    const void *cs;        /* real (2nt+1)  :377-386 */
    const void *sn;        /* real (2nt+1) */
} cuddh_ddh_desc;

typedef struct cuddh_ddh_plan cuddh_ddh_plan;
/* Builds the device-side tables the kernels need on top of the descriptor
 * (deterministic owner-gather lists replacing the LDS atomics of
 * source/DDH.cpp:108; structure check for the wave-per-subdomain kernel).
 * Allocates and synchronises.  is_f64 selects the arithmetic type of `desc`.
 * kernel: 0 = auto, 1 = generic (one workgroup per subdomain, LDS),
 *         2 = wave (one wavefront per subdomain, registers + DPP; nb == 4 only),
 *         3 = wave with the DPP reads folded into the FMAs by hand (fp32; what auto picks when it applies),
 *         4 = 3 with the in-lane contractions on the matrix pipe (v_mfma_f32_4x4x1_16b_f32, fp32),
 *         5 = one 16x16 element matrix applied to the 16 elements of a subdomain per sweep with four
 *             v_mfma_f32_16x16x4_f32 (fp32; needs the same metric tensor in every element, which
 *             plan_create verifies on the device; what auto picks when it applies),
 *         6 = n_basis == 8 (2x2 elements, the reference's other supported shape): one wavefront per TWO
 *             subdomains, registers + DPP over the eight lanes of a column octet,
 *         7 = 6 in separable form (fp32): on the rectangles of a uniform mesh the metric is diagonal and a product of
 *             1-D factors, so a sweep needs ONE 8x8 contraction per direction (D^T diag D precomputed) instead of two
 *             (plan_create verifies the geometry; what auto picks for nb == 8 when it applies). */
int cuddh_hip_ddh_plan_create(cuddh_ddh_plan **plan, const cuddh_ddh_desc *desc, int is_f64, int kernel);
int cuddh_hip_ddh_plan_destroy(cuddh_ddh_plan *plan);
/* which kernel the plan resolved to (1..7) */
int cuddh_hip_ddh_plan_kernel(const cuddh_ddh_plan *plan);
/* Numbering of the forcing x and the solution y of the NEXT apply calls: d_gI (mx_dof, n_domains) DEVICE replaces desc.gI and
 * g_ndof replaces desc.g_ndof for x and y (NULL restores the descriptor's).  With the identity numbering
 * gI(l, s) = l + mx_dof s every subdomain dof has its own entry, so the partition-of-unity contributions that source/DDH.cpp:298-307
 * adds with atomics land in distinct places and can be summed per global dof in a FIXED order afterwards
 * (cuddh_hip_csr_sum_f64): DDH::postprocess is then bitwise reproducible. */
int cuddh_hip_ddh_plan_set_vector_layout(cuddh_ddh_plan *plan, const int *d_gI, int g_ndof);
/* Issue priority of the wavefronts of the NEXT apply calls (s_setprio 3 when high != 0; 0 restores the default).  For the
 * multi-GPU schedule that solves the subdomains feeding other ranks on a second stream: with equal priority they finish
 * together with everything else resident on the device (profiles/r02/overlap_timeline.txt).  Results are unaffected. */
int cuddh_hip_ddh_plan_set_wave_priority(cuddh_ddh_plan *plan, int high);
/* WaveHoltz iterations per local solve.  The reference hard-wires 5 (`constexpr int wh_maxit = 5`,
 * source/DDH.cpp:136) and that is the default; 0 restores it.  A verification knob: with more iterations the local
 * solves become exact and DDH converges to the Helmholtz system its transmission conditions imply
 * (tests/test_ddh_physics.py); production callers never touch it. */
int cuddh_hip_ddh_plan_set_wh_iters(cuddh_ddh_plan *plan, int wh_iters);

/* source/DDH.cpp:111-321 (ddh_action + stiffness).  x: forcing [F;G] (2*g_ndof
 * doubles) or NULL; y: solution output [u;v] (2*g_ndof doubles, zero-filled by
 * the call) or NULL; lambda: traces (2*n_lambda) or NULL; update: outgoing
 * traces (2*n_lambda) or NULL.  Only subdomains [dom_begin, dom_end) are solved
 * (the whole range for a single GPU; a rank's shard for the multi-GPU path).
 * If y != NULL and zero_y != 0 the call zero-fills y first (source/DDH.cpp:144).
 */
int cuddh_hip_ddh_apply_f32(const cuddh_ddh_plan *plan, int dom_begin, int dom_end, const double *x, double *y,
                            int zero_y, const float *lambda, float *update, void *stream);
int cuddh_hip_ddh_apply_f64(const cuddh_ddh_plan *plan, int dom_begin, int dom_end, const double *x, double *y,
                            int zero_y, const double *lambda, double *update, void *stream);
/* The same for the n subdomains listed in d_domains (DEVICE, each in [0, n_domains), no duplicates: the caller's contract) in
 * ONE launch -- the subdomains of a rank whose traces other ranks wait for are not a contiguous range (two block rows of a
 * strip, the rim of a 2-D rank grid), and launching them piecewise leaves pieces running alone on an almost idle device. */
int cuddh_hip_ddh_apply_list_f32(const cuddh_ddh_plan *plan, const int *d_domains, int n, const double *x, double *y, int zero_y,
                                 const float *lambda, float *update, void *stream);
int cuddh_hip_ddh_apply_list_f64(const cuddh_ddh_plan *plan, const int *d_domains, int n, const double *x, double *y, int zero_y,
                                 const double *lambda, double *update, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CUDDH_HIP_H */
