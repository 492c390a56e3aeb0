// Matrix-free element operators in the reference's own data layouts (fp64).
//
// These are the general-purpose kernels behind StiffnessMatrix / MassMatrix /
// FaceMassMatrix::action: any n_basis, any quadrature size, any mesh.  A
// workgroup of 256 threads processes a batch of elements at once (as many as
// fit its LDS budget); each sum-factorisation stage is a flat loop over
// (element-in-batch, output index) so every wavefront is full regardless of
// nq*nq, unlike the one-tiny-block-per-element launch of the reference
// (source/StiffnessMatrix.cpp:102).  Results are scattered with hardware fp64
// atomics.  The bandwidth-critical fused path is helmholtz_fused.hip.
#include <algorithm>

#include "common.hpp"

using namespace cuddh_k;

namespace
{
    constexpr int BLOCK = 256;
    constexpr int LDS_BUDGET = 48 * 1024; // bytes of dynamic LDS per workgroup (3 workgroups per CU)

    // number of elements a workgroup handles at once
    // rounds: how many passes of the 256 threads the largest stage should take (measured: 1 is best for the stiffness
    // kernel, 2 for the mass kernels -- less idle tail per round vs more LDS per workgroup)
    inline int batch_size(int doubles_per_elem, int shared_doubles, int work_per_elem, int rounds = 1)
    {
        int e = (LDS_BUDGET / 8 - shared_doubles) / doubles_per_elem;
        const int want = rounds * ((BLOCK + work_per_elem - 1) / work_per_elem);
        if (e > want)
            e = want;
        return e < 1 ? 1 : e;
    }

    // ------------------------------------------------------------ element metrics on the tensor grid of a 1-D rule
    // The bilinear map of source/Element.cpp:5-36 evaluated at (q[i], q[j]) of every element: Jacobian [x_xi, y_xi, x_eta,
    // y_eta], its determinant, the physical point.  The reference tabulates these on the host and mirrors them
    // (source/Mesh2D.cpp:173-227); at 1024^2 elements that is 0.5 GB of host work and upload per rule.
    __global__ void __launch_bounds__(BLOCK) element_metrics_kernel(long long n_pts, int n, const double *__restrict__ corners, const double *__restrict__ q,
                                                                    double *__restrict__ J, double *__restrict__ detJ, double *__restrict__ x)
    {
        for (long long t = blockIdx.x * (long long)BLOCK + threadIdx.x; t < n_pts; t += (long long)gridDim.x * BLOCK)
        {
            const long long el = t / (n * n);
            const int loc = static_cast<int>(t - el * (n * n));
            const double s = q[loc % n], e = q[loc / n];
            const double *c = corners + 8 * el; // (2, 4): corners counter-clockwise
            double j4[4];
            bilinear_jacobian(c, s, e, j4);
            if (J)
            {
                J[4 * t] = j4[0];
                J[4 * t + 1] = j4[1];
                J[4 * t + 2] = j4[2];
                J[4 * t + 3] = j4[3];
            }
            if (detJ)
                detJ[t] = j4[0] * j4[3] - j4[1] * j4[2];
            if (x)
            {
                const double N0 = 0.25 * (1 - s) * (1 - e), N1 = 0.25 * (1 + s) * (1 - e), N2 = 0.25 * (1 + s) * (1 + e), N3 = 0.25 * (1 - s) * (1 + e);
                x[2 * t] = ((c[0] * N0 + c[2] * N1) + c[4] * N2) + c[6] * N3;
                x[2 * t + 1] = ((c[1] * N0 + c[3] * N1) + c[5] * N2) + c[7] * N3;
            }
        }
    }

    // ------------------------------------------------------------ geometric factors (K2)
    __global__ void __launch_bounds__(BLOCK) stiffness_setup_kernel(long long n_pts, int nq, const double *__restrict__ w, const double *__restrict__ J,
                                                                    double *__restrict__ G)
    {
        for (long long t = blockIdx.x * (long long)BLOCK + threadIdx.x; t < n_pts; t += (long long)gridDim.x * BLOCK)
        {
            const int loc = static_cast<int>(t % (nq * nq));
            const double W = w[loc % nq] * w[loc / nq];
            const double x_xi = J[4 * t], y_xi = J[4 * t + 1], x_eta = J[4 * t + 2], y_eta = J[4 * t + 3];
            const double det = x_xi * y_eta - x_eta * y_xi;
            G[3 * t] = W * (y_eta * y_eta + x_eta * x_eta) / det;
            G[3 * t + 1] = -W * (y_xi * y_eta + x_xi * x_eta) / det;
            G[3 * t + 2] = W * (y_xi * y_xi + x_xi * x_xi) / det;
        }
    }

    // ------------------------------------------------------------ stiffness apply (K1)
    // LDS per workgroup: P, D (nq*nb each) + per element: u (nb*nb), Pu, Du (nq*nb each), F0, F1 (nq*nq each),
    // T0, T1 (nb*nq each)
    // NBT, NQT: compile-time n_basis / n_quad (loops unroll, index arithmetic divides by constants); 0 = take the arguments
    template <int NBT, int NQT>
    __global__ void __launch_bounds__(BLOCK) stiffness_apply_kernel(int n_elem, int nq_arg, int nb_arg, int E, const double *__restrict__ gP,
                                                                    const double *__restrict__ gD, const double *__restrict__ G,
                                                                    const int *__restrict__ I, double c, const double *__restrict__ x,
                                                                    double *__restrict__ y)
    {
        extern __shared__ double lds[];
        const int nb = NBT ? NBT : nb_arg, nq = NQT ? NQT : nq_arg;
        const int nbb = nb * nb, nqb = nq * nb, nqq = nq * nq;
        double *P = lds, *D = P + nqb;
        double *u = D + nqb;      // [E][nb*nb]   u(k,l) at k + nb*l
        double *Pu = u + E * nbb; // [E][nq*nb]   (q,l) at q + nq*l
        double *Du = Pu + E * nqb;
        double *F0 = Du + E * nqb; // [E][nq*nq]  (q,r) at q + nq*r
        double *F1 = F0 + E * nqq;
        double *T0 = F1 + E * nqq; // [E][nb*nq]  (k,r) at k + nb*r
        double *T1 = T0 + E * nqb;
        const int tid = threadIdx.x;

        for (int i = tid; i < nqb; i += BLOCK)
        {
            P[i] = gP[i];
            D[i] = gD[i];
        }

        for (int base = blockIdx.x * E; base < n_elem; base += gridDim.x * E)
        {
            const int ne = min(E, n_elem - base);
            __syncthreads();
            for (int t = tid; t < ne * nbb; t += BLOCK)
                u[t] = x[I[(size_t)base * nbb + t]];
            __syncthreads();

            // xi pass: interpolate and differentiate along the first index
            for (int t = tid; t < ne * nqb; t += BLOCK)
            {
                const int e = t / nqb, r = t % nqb, q = r % nq, l = r / nq;
                const double *ue = u + e * nbb + nb * l;
                double p = 0.0, d = 0.0;
                for (int k = 0; k < nb; ++k)
                {
                    p += P[q + nq * k] * ue[k];
                    d += D[q + nq * k] * ue[k];
                }
                Pu[t] = p;
                Du[t] = d;
            }
            __syncthreads();

            // eta pass to the quadrature points, metric tensor applied there
            for (int t = tid; t < ne * nqq; t += BLOCK)
            {
                const int e = t / nqq, loc = t % nqq, q = loc % nq, r = loc / nq;
                const double *pu = Pu + e * nqb + q, *du = Du + e * nqb + q;
                double dx = 0.0, dy = 0.0;
                for (int l = 0; l < nb; ++l)
                {
                    dx += P[r + nq * l] * du[nq * l];
                    dy += D[r + nq * l] * pu[nq * l];
                }
                const double *g = G + 3 * ((size_t)(base + e) * nqq + loc);
                F0[t] = g[0] * dx + g[1] * dy;
                F1[t] = g[1] * dx + g[2] * dy;
            }
            __syncthreads();

            // test functions: xi direction
            for (int t = tid; t < ne * nqb; t += BLOCK)
            {
                const int e = t / nqb, rem = t % nqb, k = rem % nb, r = rem / nb;
                const double *f0 = F0 + e * nqq + nq * r, *f1 = F1 + e * nqq + nq * r;
                double a = 0.0, b = 0.0;
                for (int q = 0; q < nq; ++q)
                {
                    a += D[q + nq * k] * f0[q];
                    b += P[q + nq * k] * f1[q];
                }
                T0[t] = a;
                T1[t] = b;
            }
            __syncthreads();

            // test functions: eta direction, then scatter
            for (int t = tid; t < ne * nbb; t += BLOCK)
            {
                const int e = t / nbb, rem = t % nbb, k = rem % nb, l = rem / nb;
                const double *t0 = T0 + e * nqb + k, *t1 = T1 + e * nqb + k;
                double s = 0.0;
                for (int r = 0; r < nq; ++r)
                    s += P[r + nq * l] * t0[nb * r] + D[r + nq * l] * t1[nb * r];
                atomic_add(y + I[(size_t)base * nbb + t], c * s);
            }
        }
    }

    // ------------------------------------------------------------ mass set-up (K4) and apply (K3)
    // LDS: P (nq*nb) + per element: u (nb*nb) , A (nq*nb), Q (nq*nq), Bq (nq*nb)
    template <bool SETUP, int NBT, int NQT>
    __global__ void __launch_bounds__(BLOCK) mass_kernel(int n_elem, int nq_arg, int nb_arg, int E, const double *__restrict__ gP,
                                                         const int *__restrict__ I,
                                                         const double *__restrict__ coef,  // SETUP: nodal coefficient or null
                                                         const double *__restrict__ detJ,  // SETUP
                                                         const double *__restrict__ w,     // SETUP
                                                         const double *__restrict__ a,     // APPLY: weights at quadrature points
                                                         double c, const double *__restrict__ x, double *__restrict__ out)
    {
        extern __shared__ double lds[];
        const int nb = NBT ? NBT : nb_arg, nq = NQT ? NQT : nq_arg;
        const int nbb = nb * nb, nqb = nq * nb, nqq = nq * nq;
        double *P = lds;
        double *u = P + nqb;      // [E][nb*nb]
        double *A = u + E * nbb;  // [E][nq*nb]  (q,l) at q + nq*l
        double *Q = A + E * nqb;  // [E][nq*nq]  (q,r) at q + nq*r
        double *Bq = Q + E * nqq; // [E][nb*nq]  (k,r) at k + nb*r
        const int tid = threadIdx.x;

        for (int i = tid; i < nqb; i += BLOCK)
            P[i] = gP[i];

        for (int base = blockIdx.x * E; base < n_elem; base += gridDim.x * E)
        {
            const int ne = min(E, n_elem - base);
            __syncthreads();
            for (int t = tid; t < ne * nbb; t += BLOCK)
            {
                if (SETUP)
                    u[t] = coef ? coef[I[(size_t)base * nbb + t]] : 1.0;
                else
                    u[t] = x[I[(size_t)base * nbb + t]];
            }
            __syncthreads();

            for (int t = tid; t < ne * nqb; t += BLOCK)
            {
                const int e = t / nqb, r = t % nqb, q = r % nq, l = r / nq;
                const double *ue = u + e * nbb + nb * l;
                double p = 0.0;
                for (int k = 0; k < nb; ++k)
                    p += P[q + nq * k] * ue[k];
                A[t] = p;
            }
            __syncthreads();

            for (int t = tid; t < ne * nqq; t += BLOCK)
            {
                const int e = t / nqq, loc = t % nqq, q = loc % nq, r = loc / nq;
                const double *ae = A + e * nqb + q;
                double v = 0.0;
                for (int l = 0; l < nb; ++l)
                    v += P[r + nq * l] * ae[nq * l];
                const size_t gq = (size_t)(base + e) * nqq + loc;
                if (SETUP)
                    out[gq] = v * (w[q] * w[r] * detJ[gq]);
                else
                    Q[t] = a[gq] * v;
            }
            if (SETUP)
                continue;
            __syncthreads();

            for (int t = tid; t < ne * nqb; t += BLOCK)
            {
                const int e = t / nqb, rem = t % nqb, k = rem % nb, r = rem / nb;
                const double *qe = Q + e * nqq + nq * r;
                double s = 0.0;
                for (int q = 0; q < nq; ++q)
                    s += P[q + nq * k] * qe[q];
                Bq[t] = s;
            }
            __syncthreads();

            for (int t = tid; t < ne * nbb; t += BLOCK)
            {
                const int e = t / nbb, rem = t % nbb, k = rem % nb, l = rem / nb;
                const double *be = Bq + e * nqb + k;
                double s = 0.0;
                for (int r = 0; r < nq; ++r)
                    s += P[r + nq * l] * be[nb * r];
                atomic_add(out + I[(size_t)base * nbb + t], c * s);
            }
        }
    }

    // ------------------------------------------------------------ lumped (diagonal) masses (K5, K9)
    __global__ void __launch_bounds__(BLOCK) lumped_mass_kernel(long long n_nodes, int nb, int dim, const double *__restrict__ coef,
                                                                const double *__restrict__ detJ, const double *__restrict__ w,
                                                                const int *__restrict__ I, double *__restrict__ op)
    {
        // dim == 2: element nodes (nb*nb per element); dim == 1: face nodes (nb per face)
        for (long long t = blockIdx.x * (long long)BLOCK + threadIdx.x; t < n_nodes; t += (long long)gridDim.x * BLOCK)
        {
            const int loc = static_cast<int>(t % (dim == 2 ? nb * nb : nb));
            const int idx = I[t];
            double m = (dim == 2 ? w[loc % nb] * w[loc / nb] : w[loc]) * detJ[t];
            if (coef)
                m *= coef[idx];
            atomic_add(op + idx, m);
        }
    }

    // ------------------------------------------------------------ face mass (K8, K7): one thread per face node / quadrature point
    template <bool SETUP>
    __global__ void __launch_bounds__(BLOCK) facemass_kernel(int n_faces, int nb, int nq, int FPB, const double *__restrict__ gP,
                                                             const int *__restrict__ I, const double *__restrict__ coef,
                                                             const double *__restrict__ detJ, const double *__restrict__ w,
                                                             const double *__restrict__ a, double c, const double *__restrict__ x,
                                                             double *__restrict__ out)
    {
        extern __shared__ double lds[];
        double *P = lds;          // nq*nb
        double *u = P + nq * nb;  // [FPB][nb]
        double *Pu = u + FPB * nb; // [FPB][nq]
        const int tid = threadIdx.x;
        for (int i = tid; i < nq * nb; i += BLOCK)
            P[i] = gP[i];

        for (int base = blockIdx.x * FPB; base < n_faces; base += gridDim.x * FPB)
        {
            const int nf = min(FPB, n_faces - base);
            __syncthreads();
            for (int t = tid; t < nf * nb; t += BLOCK)
            {
                const int idx = I[(size_t)base * nb + t];
                if (SETUP)
                    u[t] = coef ? coef[idx] : 1.0;
                else
                    u[t] = x[idx];
            }
            __syncthreads();
            for (int t = tid; t < nf * nq; t += BLOCK)
            {
                const int f = t / nq, q = t % nq;
                double v = 0.0;
                for (int l = 0; l < nb; ++l)
                    v += P[q + nq * l] * u[f * nb + l];
                const size_t gq = (size_t)base * nq + t;
                if (SETUP)
                    out[gq] = v * (w[q] * detJ[gq]);
                else
                    Pu[t] = v * a[gq];
            }
            if (SETUP)
                continue;
            __syncthreads();
            for (int t = tid; t < nf * nb; t += BLOCK)
            {
                const int f = t / nb, k = t % nb;
                double s = 0.0;
                for (int q = 0; q < nq; ++q)
                    s += P[q + nq * k] * Pu[f * nq + q];
                atomic_add(out + I[(size_t)base * nb + t], c * s);
            }
        }
    }

    inline int grid_for_batches(int n_items, int per_block)
    {
        long long g = ((long long)n_items + per_block - 1) / per_block;
        const long long cap = 256LL * 16;
        if (g > cap)
            g = cap;
        return static_cast<int>(g < 1 ? 1 : g);
    }
} // namespace

extern "C"
{
    int cuddh_hip_element_metrics(int n_elem, int n, const double *corners, const double *q, double *J, double *detJ, double *x, void *stream)
    {
        const long long n_pts = (long long)n_elem * n * n;
        if (n_pts <= 0)
            return 0;
        hipLaunchKernelGGL(element_metrics_kernel, dim3(stream_grid(n_pts, BLOCK)), dim3(BLOCK), 0, as_stream(stream), n_pts, n, corners, q, J, detJ, x);
        return launch_status();
    }

    int cuddh_hip_stiffness_setup(int n_elem, int nq, const double *w, const double *J, double *G, void *stream)
    {
        const long long n_pts = (long long)n_elem * nq * nq;
        if (n_pts <= 0)
            return 0;
        hipLaunchKernelGGL(stiffness_setup_kernel, dim3(stream_grid(n_pts, BLOCK)), dim3(BLOCK), 0, as_stream(stream), n_pts, nq, w, J, G);
        return launch_status();
    }

    int cuddh_hip_stiffness_apply(int n_elem, int nq, int nb, const double *P, const double *D, const double *G, const int *I,
                                  double c, const double *x, double *y, void *stream)
    {
        if (n_elem <= 0)
            return 0;
        const int per_elem = nb * nb + 4 * nq * nb + 2 * nq * nq;
        const int shared = 2 * nq * nb;
        if ((per_elem + shared) * 8 > 64 * 1024)
            return static_cast<int>(hipErrorInvalidValue);
        const int E = batch_size(per_elem, shared, nq * nq);
        const size_t lds = (size_t)(shared + E * per_elem) * sizeof(double);
        const dim3 grid(grid_for_batches(n_elem, E)), block(BLOCK);
        hipStream_t st = as_stream(stream);
        // (compile-time sizes were measured for this kernel too: no gain at n_basis 8, slower at 7 -- it is bound by its LDS
        // passes and barriers, not by index arithmetic; the mass kernel below does gain)
        hipLaunchKernelGGL((stiffness_apply_kernel<0, 0>), grid, block, lds, st, n_elem, nq, nb, E, P, D, G, I, c, x, y);
        return launch_status();
    }

    int cuddh_hip_mass_setup(int n_elem, int nq, int nb, const double *coef, const double *detJ, const double *w, const int *I,
                             const double *P, double *a, void *stream)
    {
        if (n_elem <= 0)
            return 0;
        const int per_elem = nb * nb + 2 * nq * nb + nq * nq;
        const int shared = nq * nb;
        if ((per_elem + shared) * 8 > 64 * 1024)
            return static_cast<int>(hipErrorInvalidValue);
        const int E = batch_size(per_elem, shared, nq * nq);
        const size_t lds = (size_t)(shared + E * per_elem) * sizeof(double);
        hipLaunchKernelGGL((mass_kernel<true, 0, 0>), dim3(grid_for_batches(n_elem, E)), dim3(BLOCK), lds, as_stream(stream), n_elem, nq, nb, E,
                           P, I, coef, detJ, w, static_cast<const double *>(nullptr), 0.0, static_cast<const double *>(nullptr), a);
        return launch_status();
    }

    int cuddh_hip_mass_apply(int n_elem, int nq, int nb, const int *I, const double *P, const double *a, double c, const double *x,
                             double *y, void *stream)
    {
        if (n_elem <= 0)
            return 0;
        const int per_elem = nb * nb + 2 * nq * nb + nq * nq;
        const int shared = nq * nb;
        if ((per_elem + shared) * 8 > 64 * 1024)
            return static_cast<int>(hipErrorInvalidValue);
        const int E = batch_size(per_elem, shared, nq * nq, 2);
        const size_t lds = (size_t)(shared + E * per_elem) * sizeof(double);
        const dim3 grid(grid_for_batches(n_elem, E)), block(BLOCK);
        hipStream_t st = as_stream(stream);
        const double *none = nullptr;
#define CUDDH_M_CASE(NB_, NQ_)                                                                                                      \
    if (nb == NB_ && nq == NQ_)                                                                                                     \
    {                                                                                                                               \
        hipLaunchKernelGGL((mass_kernel<false, NB_, NQ_>), grid, block, lds, st, n_elem, nq, nb, E, P, I, none, none, none, a, c, x, y); \
        return launch_status();                                                                                                     \
    }
        // n_basis + 1 (a == 1) and 1 + 3 n_basis / 2 + 1 (weighted), reference source/MassMatrix.cpp:74,108
        CUDDH_M_CASE(6, 7)
        CUDDH_M_CASE(7, 8)
        CUDDH_M_CASE(8, 9)
        CUDDH_M_CASE(6, 11)
        CUDDH_M_CASE(7, 12)
        CUDDH_M_CASE(8, 14)
#undef CUDDH_M_CASE
        hipLaunchKernelGGL((mass_kernel<false, 0, 0>), grid, block, lds, st, n_elem, nq, nb, E, P, I, none, none, none, a, c, x, y);
        return launch_status();
    }

    int cuddh_hip_diag_mass_setup(int ndof, int n_elem, int nb, const double *coef, const double *detJ, const double *w, const int *I,
                                  double *op, void *stream)
    {
        if (ndof <= 0)
            return 0;
        hipError_t e = hipMemsetAsync(op, 0, (size_t)ndof * sizeof(double), as_stream(stream));
        if (e != hipSuccess)
            return static_cast<int>(e);
        const long long n_nodes = (long long)n_elem * nb * nb;
        hipLaunchKernelGGL(lumped_mass_kernel, dim3(stream_grid(n_nodes, BLOCK)), dim3(BLOCK), 0, as_stream(stream), n_nodes, nb, 2, coef,
                           detJ, w, I, op);
        int err = launch_status();
        if (err)
            return err;
        return cuddh_hip_reciprocal_f64(ndof, op, stream);
    }

    int cuddh_hip_facemass_setup(int n_faces, int nb, int nq, const double *w, const double *P, const double *detJ, const double *coef,
                                 const int *I, double *a, void *stream)
    {
        if (n_faces <= 0)
            return 0;
        const int FPB = std::max(1, BLOCK / nq);
        const size_t lds = (size_t)(nq * nb + FPB * (nb + nq)) * sizeof(double);
        hipLaunchKernelGGL((facemass_kernel<true>), dim3(grid_for_batches(n_faces, FPB)), dim3(BLOCK), lds, as_stream(stream), n_faces, nb,
                           nq, FPB, P, I, coef, detJ, w, static_cast<const double *>(nullptr), 0.0,
                           static_cast<const double *>(nullptr), a);
        return launch_status();
    }

    int cuddh_hip_facemass_apply(int n_faces, int nb, int nq, const double *P, const double *a, const int *I, double c,
                                 const double *x, double *y, void *stream)
    {
        if (n_faces <= 0)
            return 0;
        const int FPB = std::max(1, BLOCK / nq);
        const size_t lds = (size_t)(nq * nb + FPB * (nb + nq)) * sizeof(double);
        hipLaunchKernelGGL((facemass_kernel<false>), dim3(grid_for_batches(n_faces, FPB)), dim3(BLOCK), lds, as_stream(stream), n_faces, nb,
                           nq, FPB, P, I, static_cast<const double *>(nullptr), static_cast<const double *>(nullptr),
                           static_cast<const double *>(nullptr), a, c, x, y);
        return launch_status();
    }

    int cuddh_hip_diag_facemass_setup(int ndof, int n_faces, int nb, const double *w, const double *detJ, const double *coef,
                                      const int *I, double *op, void *stream)
    {
        if (ndof <= 0)
            return 0;
        const long long n_nodes = (long long)n_faces * nb;
        hipLaunchKernelGGL(lumped_mass_kernel, dim3(stream_grid(n_nodes, BLOCK)), dim3(BLOCK), 0, as_stream(stream), n_nodes, nb, 1, coef,
                           detJ, w, I, op);
        int err = launch_status();
        if (err)
            return err;
        return cuddh_hip_reciprocal_f64(ndof, op, stream);
    }
}
