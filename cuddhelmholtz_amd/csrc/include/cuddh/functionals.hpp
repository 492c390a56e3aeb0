// Load vectors (f, phi_i) for a user functor evaluated on the device.
// Contracts: reference include/LinearFunctional.hpp:11-43,145-181 and
// include/FaceLinearFunctional.hpp:11-44,127-164.  These stay header templates
// compiled by hipcc in user code because the functor is a device lambda
// (reference examples/DDH.cpp:120-122); they are set-up code, not the hot path.
//
// Without a quadrature argument the Gauss-Lobatto nodes of the basis are used
// (collocation: F[I] += c w_i w_j detJ f(x_ij)); with one, f is sampled at the
// rule's points into a scratch array and contracted with P^T (x) P^T.
#ifndef CUDDH_AMD_FUNCTIONALS_HPP
#define CUDDH_AMD_FUNCTIONALS_HPP

#include <hip/hip_runtime.h>

#include "blas1.hpp"
#include "launch.hpp"
#include "memory.hpp"
#include "spaces.hpp"

namespace cuddh
{
    class LinearFunctional
    {
    public:
        explicit LinearFunctional(const H1Space &fem);
        LinearFunctional(const H1Space &fem, const QuadratureRule &quad);

        /// F[i] += c (f, phi_i);  f(const double x[2]) -> double, callable on the device
        template <typename Func>
        void action(double c, Func &&f, double *F) const;

        /// F[i] = (f, phi_i)
        template <typename Func>
        void action(Func &&f, double *F) const;

    private:
        const H1Space &fem;
        const int ndof, n_elem, n_basis, n_quad;
        const bool collocated;
        const Mesh2D::ElementMetricCollection &metrics;
        host_device_dvec _w, _P;
        mutable host_device_dvec samples; // (n_quad, n_quad, n_elem) scratch
    };

    class FaceLinearFunctional
    {
    public:
        explicit FaceLinearFunctional(const FaceSpace &fs);
        FaceLinearFunctional(const FaceSpace &fs, const QuadratureRule &quad);

        template <typename Func>
        void action(double c, Func &&f, double *F) const;

        template <typename Func>
        void action(Func &&f, double *F) const;

    private:
        const FaceSpace &fs;
        const Mesh2D::EdgeMetricCollection &metrics;
        const int fdof, n_faces, n_basis, n_quad;
        const bool collocated;
        host_device_dvec _w, _P;
        mutable host_device_dvec samples; // (n_quad, n_faces) scratch
    };

    // ---------------------------------------------------------------- volume

    template <typename Func>
    void LinearFunctional::action(double c, Func &&f, double *F) const
    {
        const double *w = _w.device_read();
        const double *detJ = metrics.measures(MemorySpace::DEVICE);
        const double *X = metrics.physical_coordinates(MemorySpace::DEVICE);
        const int *I = fem.global_indices(MemorySpace::DEVICE);
        const int nq = n_quad, nb = n_basis;
        const int pts = nq * nq;

        if (collocated)
        {
            // quadrature points are the nodes: one thread per element node
            forall(n_elem * pts, [=] __device__(int t) -> void
            {
                const int loc = t % pts;
                double xy[2] = {X[2 * t], X[2 * t + 1]};
                const double v = c * w[loc % nq] * w[loc / nq] * detJ[t] * f(xy);
                atomicAdd(F + I[t], v);
            });
            return;
        }

        if (samples.size() != n_elem * pts)
            samples.resize(n_elem * pts);
        double *g = samples.device_write();
        const double *P = _P.device_read();

        forall(n_elem * pts, [=] __device__(int t) -> void
        {
            const int loc = t % pts;
            double xy[2] = {X[2 * t], X[2 * t + 1]};
            g[t] = w[loc % nq] * w[loc / nq] * detJ[t] * f(xy);
        });

        forall(n_elem * nb * nb, [=] __device__(int t) -> void
        {
            const int el = t / (nb * nb);
            const int i = t % nb, j = (t / nb) % nb;
            const double *ge = g + el * pts;
            double acc = 0.0;
            for (int r = 0; r < nq; ++r)
            {
                double row = 0.0;
                for (int q = 0; q < nq; ++q)
                    row += P[q + nq * i] * ge[q + nq * r];
                acc += P[r + nq * j] * row;
            }
            atomicAdd(F + I[t], c * acc);
        });
    }

    template <typename Func>
    void LinearFunctional::action(Func &&f, double *F) const
    {
        zeros(ndof, F);
        action(1.0, f, F);
    }

    // ---------------------------------------------------------------- faces

    template <typename Func>
    void FaceLinearFunctional::action(double c, Func &&f, double *F) const
    {
        const double *w = _w.device_read();
        const double *detJ = metrics.measures(MemorySpace::DEVICE);
        const double *X = metrics.physical_coordinates(MemorySpace::DEVICE);
        const int *I = fs.subspace_indices(MemorySpace::DEVICE);
        const int nq = n_quad, nb = n_basis;

        if (collocated)
        {
            forall(n_faces * nq, [=] __device__(int t) -> void
            {
                double xy[2] = {X[2 * t], X[2 * t + 1]};
                atomicAdd(F + I[t], c * w[t % nq] * detJ[t] * f(xy));
            });
            return;
        }

        if (samples.size() != n_faces * nq)
            samples.resize(n_faces * nq);
        double *g = samples.device_write();
        const double *P = _P.device_read();

        forall(n_faces * nq, [=] __device__(int t) -> void
        {
            double xy[2] = {X[2 * t], X[2 * t + 1]};
            g[t] = w[t % nq] * detJ[t] * f(xy);
        });

        forall(n_faces * nb, [=] __device__(int t) -> void
        {
            const int e = t / nb, j = t % nb;
            double acc = 0.0;
            for (int q = 0; q < nq; ++q)
                acc += P[q + nq * j] * g[q + nq * e];
            atomicAdd(F + I[t], c * acc);
        });
    }

    template <typename Func>
    void FaceLinearFunctional::action(Func &&f, double *F) const
    {
        zeros(fdof, F);
        action(1.0, f, F);
    }
} // namespace cuddh

#endif
