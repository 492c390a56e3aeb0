// Restarted GMRES.  Iteration semantics follow reference source/gmres.cpp:91-235
// (see krylov.hpp); the arithmetic is scheduled differently: the k+1 projection
// coefficients of an Arnoldi step stay on the device (each axpy reads its
// coefficient from device memory) and reach the host in one copy per step.
#include "cuddh/krylov.hpp"

#include <algorithm>
#include <cmath>
#include <string>

#include "cuddh_hip.h"

namespace cuddh
{
    namespace
    {
        // ---- scalar-type dispatch onto the C ABI
        inline int k_dot(int n, const double *x, const double *y, double *r, void *ws) { return cuddh_hip_dot_f64(n, x, y, r, ws, stream()); }
        inline int k_dot(int n, const float *x, const float *y, float *r, void *ws) { return cuddh_hip_dot_f32(n, x, y, r, ws, stream()); }
        inline int k_axpby_dev(int n, double sa, const double *a, const double *x, double b, double *y) { return cuddh_hip_axpby_dev_f64(n, sa, a, x, b, y, stream()); }
        inline int k_axpby_dev(int n, float sa, const float *a, const float *x, float b, float *y) { return cuddh_hip_axpby_dev_f32(n, sa, a, x, b, y, stream()); }
        inline int k_mgs_stage(int n, double *w, const double *vp, const double *vn, const double *pi, double *po, double *h) { return cuddh_hip_mgs_stage_f64(n, w, vp, vn, pi, po, h, stream()); }
        inline int k_mgs_stage(int n, float *w, const float *vp, const float *vn, const float *pi, float *po, float *h) { return cuddh_hip_mgs_stage_f32(n, w, vp, vn, pi, po, h, stream()); }
        inline int k_mgs_finish(int n, double *w, const double *pi, double *h) { return cuddh_hip_mgs_finish_f64(n, w, pi, h, stream()); }
        inline int k_mgs_finish(int n, float *w, const float *pi, float *h) { return cuddh_hip_mgs_finish_f32(n, w, pi, h, stream()); }

        // apply the k previous rotations to column h, then build rotation k that zeroes h[k+1]
        template <typename scalar>
        void rotate_column(scalar *h, scalar *cs, scalar *sn, int k)
        {
            for (int i = 0; i < k; ++i)
            {
                const scalar a = h[i], b = h[i + 1];
                h[i] = cs[i] * a + sn[i] * b;
                h[i + 1] = cs[i] * b - sn[i] * a;
            }
            const scalar r = std::hypot(h[k], h[k + 1]);
            cs[k] = h[k] / r;
            sn[k] = h[k + 1] / r;
            h[k] = cs[k] * h[k] + sn[k] * h[k + 1];
            h[k + 1] = 0;
        }

        // back substitution R y = g for the leading kk x kk block of the (ld x m) column-major array R
        // (the reference calls LAPACK ?trsv('U','N','N'), source/gmres.cpp:26-41)
        template <typename scalar>
        void back_substitute(int kk, const scalar *R, int ld, scalar *g)
        {
            for (int i = kk - 1; i >= 0; --i)
            {
                scalar s = g[i];
                for (int j = i + 1; j < kk; ++j)
                    s -= R[i + ld * j] * g[j];
                g[i] = s / R[i + ld * i];
            }
        }

        class LeftPreconditioned : public Operator
        {
        public:
            LeftPreconditioned(int n, const Operator *A_, const Operator *P_) : tmp(n), A(A_), P(P_) {}

            void action(double, const double *, double *) const override
            {
                cuddh_error("gmres: the preconditioned system only supports action(x, y).");
            }

            void action(const double *x, double *y) const override
            {
                double *q = tmp.device_write();
                A->action(x, q);
                P->action(q, y);
            }

        private:
            mutable host_device_dvec tmp;
            const Operator *A;
            const Operator *P;
        };

        template <typename scalar, typename Op>
        solver_out arnoldi_restarted(int n, scalar *x, const Op *A, const scalar *b, int m, int maxit, scalar tol, int verbose,
                                     double max_seconds, const ScalarReduce *red = nullptr)
        {
            using clock = std::chrono::high_resolution_clock;
            const scalar one = 1, zero = 0;
            const int m1 = m + 1;
            constexpr int is_f64 = sizeof(scalar) == 8;

            HostDeviceArray<scalar> r_store(n), V_store(n * m1), col_store(m1 + 1);
            scalar *r = r_store.device_write();
            scalar *V = V_store.device_write();
            scalar *dcol = col_store.device_write(); // Hessenberg column under construction, on the device

            void *ws = nullptr;
            detail::check_hip(cuddh_hip_malloc_zeroed(&ws, cuddh_hip_reduce_ws_bytes()), "gmres workspace");
            struct Guard
            {
                void *p;
                ~Guard() { cuddh_hip_free(p); }
            } guard{ws};

            // Operators that only queue device work (operator.hpp: QueuesDeviceWorkOnly) are driven one Arnoldi step ahead of the host:
            // two pinned Hessenberg columns and two events
            const bool ahead = !red && dynamic_cast<const QueuesDeviceWorkOnly *>(A) != nullptr;
            struct AheadBuffers
            {
                scalar *pin[2] = {nullptr, nullptr};
                void *ev[2] = {nullptr, nullptr};
                ~AheadBuffers()
                {
                    for (int i = 0; i < 2; ++i)
                    {
                        (void)cuddh_hip_host_free(pin[i]);
                        (void)cuddh_hip_event_destroy(ev[i]);
                    }
                }
            } ab;
            if (ahead)
                for (int i = 0; i < 2; ++i)
                {
                    detail::check_hip(cuddh_hip_host_alloc(reinterpret_cast<void **>(&ab.pin[i]), sizeof(scalar) * (m1 + 1)), "gmres pinned column");
                    detail::check_hip(cuddh_hip_event_create(&ab.ev[i]), "gmres event");
                }
            scalar *const *pin = ab.pin;
            void *const *ev = ab.ev;

            // 2-norm; with partitioned vectors the sum of squares is reduced over the ranks first
            auto norm_of = [&](const scalar *v) -> scalar
            {
                if (!red)
                    return norm(n, v);
                scalar ss = 0;
                detail::check_hip(k_dot(n, v, v, dcol, ws), "gmres norm");
                red->fn(red->user, dcol, 1, is_f64);
                detail::check_hip(cuddh_hip_stream_sync(stream()), "gmres sync");
                detail::check_hip(cuddh_hip_copy_d2h_on(&ss, dcol, sizeof(scalar), stream()), "gmres norm copy");
                return std::sqrt(ss);
            };

            const scalar bnrm = norm_of(b);

            std::vector<scalar> H(static_cast<std::size_t>(m1) * m, 0), cs(m, 0), sn(m, 0), eta(m1, 0);

            solver_out out;
            out.success = false;
            out.num_iter = 0;
            out.num_matvec = 0;
            out.res_norm.reserve(maxit + 1);
            out.time.reserve(maxit + 1);

            A->action(x, r);
            out.num_matvec++;
            axpby(n, one, b, -one, r); // r = b - A x
            scalar r_nrm = norm_of(r);

            out.res_norm.push_back(static_cast<double>(r_nrm));
            out.time.push_back(0.0);
            const auto t0 = clock::now();

            if (r_nrm < tol * bnrm)
            {
                out.success = true;
                if (verbose)
                    std::cout << "After 0 iterations, GMRES achieved rel. residual of " << out.res_norm.back() / bnrm
                              << "\nGMRES successfully converged within desired tolerance." << std::endl;
                return out;
            }

            if (verbose)
                std::cout << std::setprecision(5) << std::scientific;

            int it = 1;
            for (; it < maxit; ++it)
            {
                axpby(n, one / r_nrm, r, zero, V); // v0 = r / ||r||
                std::fill(eta.begin(), eta.end(), zero);
                eta[0] = r_nrm;

                // One Arnoldi step queued on the stream: w = A v_k, modified Gram-Schmidt against v_0..v_k as a chain of fused stages
                // (stage j applies the projection on v_{j-1} and leaves the partial sums of <w, v_j> -- the last one <w, w> -- for
                // the next stage, so one launch per basis vector does what dot + reduce + axpy did; coefficients stay on the device),
                // normalisation.  On breakdown (norm == 0) v_{k+1} becomes non-finite but is never used.
                auto queue_step = [&](int k)
                {
                    scalar *vk = V + static_cast<std::size_t>(k) * n;
                    scalar *vk1 = vk + n;
                    A->action(vk, vk1);
                    scalar *pa = static_cast<scalar *>(ws), *pb = pa + cuddh_hip_reduce_ws_bytes() / (2 * sizeof(double));
                    detail::check_hip(k_mgs_stage(n, vk1, static_cast<const scalar *>(nullptr), V, pa, pa, dcol), "gmres mgs");
                    for (int j = 0; j <= k; ++j)
                    {
                        const scalar *vj = V + static_cast<std::size_t>(j) * n;
                        const scalar *vnext = (j < k) ? vj + n : nullptr;
                        detail::check_hip(k_mgs_stage(n, vk1, vj, vnext, pa, pb, dcol + j), "gmres mgs");
                        std::swap(pa, pb);
                    }
                    detail::check_hip(k_mgs_finish(n, vk1, pa, dcol + k + 1), "gmres normalise");
                    if (ahead)
                    {
                        // the Hessenberg column leaves for pinned host memory behind the step; the next step may be queued behind it
                        detail::check_hip(cuddh_hip_copy_d2h_async(pin[k & 1], dcol, sizeof(scalar) * (k + 2), stream()), "gmres column copy");
                        detail::check_hip(cuddh_hip_event_record(ev[k & 1], stream()), "gmres event");
                    }
                };

                int k1 = 0;
                if (ahead)
                    queue_step(0);
                for (int k = 0; k < m; ++k)
                {
                    k1 = k + 1;
                    scalar *vk = V + static_cast<std::size_t>(k) * n;
                    scalar *vk1 = vk + n;
                    scalar *h = H.data() + static_cast<std::size_t>(m1) * k;

                    if (red)
                    {
                        A->action(vk, vk1);
                        // partitioned vectors: every coefficient is summed over the ranks before it is applied
                        for (int j = 0; j < k1; ++j)
                        {
                            const scalar *vj = V + static_cast<std::size_t>(j) * n;
                            detail::check_hip(k_dot(n, vk1, vj, dcol + j, ws), "gmres dot");
                            red->fn(red->user, dcol + j, 1, is_f64);
                            detail::check_hip(k_axpby_dev(n, -one, dcol + j, vj, one, vk1), "gmres projection");
                        }
                        detail::check_hip(k_dot(n, vk1, vk1, dcol + k1, ws), "gmres dot");
                        red->fn(red->user, dcol + k1, 1, is_f64);
                        detail::check_hip(cuddh_hip_stream_sync(stream()), "gmres sync");
                        detail::check_hip(cuddh_hip_copy_d2h_on(h, dcol, sizeof(scalar) * (k1 + 1), stream()), "gmres column copy");
                        h[k1] = std::sqrt(h[k1]); // the reduced sum of squares
                        if (h[k1] != zero)
                            scal(n, one / h[k1], vk1);
                    }
                    else if (ahead)
                    {
                        // one step ahead: step k + 1 is queued (its operator apply needs nothing from the host) before the host waits
                        // for step k's column, so the device never idles while the host rotates; if the iteration stops at column k
                        // the extra step is discarded (it wrote v_{k+2} and the device column, neither is read afterwards)
                        if (k + 1 < m)
                            queue_step(k + 1);
                        detail::check_hip(cuddh_hip_event_sync(ev[k & 1]), "gmres event wait");
                        std::copy(pin[k & 1], pin[k & 1] + k1 + 1, h);
                    }
                    else
                    {
                        queue_step(k);
                        detail::check_hip(cuddh_hip_stream_sync(stream()), "gmres sync");
                        detail::check_hip(cuddh_hip_copy_d2h_on(h, dcol, sizeof(scalar) * (k1 + 1), stream()), "gmres column copy");
                    }
                    out.num_matvec++;

                    if (h[k1] == zero)
                        break;

                    rotate_column(h, cs.data(), sn.data(), k);
                    eta[k1] = -sn[k] * eta[k];
                    eta[k] = cs[k] * eta[k];

                    if (std::abs(eta[k1]) < tol * bnrm)
                        break;
                }

                back_substitute(k1, H.data(), m1, eta.data());
                for (int k = 0; k < k1; ++k)
                    axpby(n, eta[k], V + static_cast<std::size_t>(k) * n, one, x);

                A->action(x, r);
                out.num_matvec++;
                axpby(n, one, b, -one, r);
                r_nrm = norm_of(r);

                out.res_norm.push_back(static_cast<double>(r_nrm));
                const double elapsed = std::chrono::duration<double>(clock::now() - t0).count();
                out.time.push_back(elapsed);
                bool out_of_time = elapsed > max_seconds;
                if (red)
                {
                    // each rank reads its own clock: the decision to stop must be the same on all of them, or one rank leaves the
                    // sequence of collectives while the others enter the next one -- reduce the flag like every other scalar
                    const scalar flag = out_of_time ? one : zero;
                    scalar total = zero;
                    detail::check_hip(cuddh_hip_copy_h2d_on(dcol, &flag, sizeof(scalar), stream()), "gmres stop flag");
                    red->fn(red->user, dcol, 1, is_f64);
                    detail::check_hip(cuddh_hip_stream_sync(stream()), "gmres sync");
                    detail::check_hip(cuddh_hip_copy_d2h_on(&total, dcol, sizeof(scalar), stream()), "gmres stop flag");
                    out_of_time = total > zero;
                }
                if (out_of_time)
                    break;

                if (verbose == 1)
                {
                    const int width = 30;
                    const int filled = std::min(width, width * it / std::max(1, maxit - 1));
                    std::cout << "[" << std::string(filled, '#') << std::string(width - filled, ' ') << "] || iteration "
                              << std::setw(10) << it + 1 << " / " << maxit << " || rel. res. = " << std::setw(10)
                              << r_nrm / bnrm << "\r" << std::flush;
                }
                else if (verbose >= 2)
                {
                    std::cout << "iteration " << std::setw(10) << it + 1 << " / " << maxit
                              << " || rel. res. = " << std::setw(10) << r_nrm / bnrm << std::endl;
                }

                if (r_nrm < tol * bnrm)
                {
                    out.success = true;
                    break;
                }
            }

            if (verbose == 1)
                std::cout << std::endl;
            if (verbose)
                std::cout << "After " << it << " iterations, GMRES achieved rel. residual of " << out.res_norm.back() / bnrm
                          << (out.success ? "\nGMRES successfully converged within desired tolerance."
                                          : "\nGMRES failed to converge within desired tolerance.")
                          << std::endl;

            out.num_iter = it;
            return out;
        }
    } // namespace

    solver_out gmres(int n, double *x, const Operator *A, const double *b, int m, int maxit, double tol, int verbose,
                     double max_seconds)
    {
        return arnoldi_restarted<double>(n, x, A, b, m, maxit, tol, verbose, max_seconds);
    }

    solver_out gmres(int n, double *x, const Operator *A, const double *b, const Operator *Precond, int m, int maxit,
                     double tol, int verbose, double max_seconds)
    {
        LeftPreconditioned PA(n, A, Precond);
        host_device_dvec Pb(n);
        double *d_Pb = Pb.device_write();
        Precond->action(b, d_Pb);
        return arnoldi_restarted<double>(n, x, &PA, d_Pb, m, maxit, tol, verbose, max_seconds);
    }

    solver_out gmres(int n, float *x, const SinglePrecisionOperator *A, const float *b, int m, int maxit, float tol,
                     int verbose, double max_seconds)
    {
        return arnoldi_restarted<float>(n, x, A, b, m, maxit, tol, verbose, max_seconds);
    }

    solver_out gmres(int n, double *x, const Operator *A, const double *b, int m, int maxit, double tol, int verbose,
                     double max_seconds, const ScalarReduce &reduce)
    {
        return arnoldi_restarted<double>(n, x, A, b, m, maxit, tol, verbose, max_seconds, reduce.fn ? &reduce : nullptr);
    }

    solver_out gmres(int n, float *x, const SinglePrecisionOperator *A, const float *b, int m, int maxit, float tol, int verbose,
                     double max_seconds, const ScalarReduce &reduce)
    {
        return arnoldi_restarted<float>(n, x, A, b, m, maxit, tol, verbose, max_seconds, reduce.fn ? &reduce : nullptr);
    }
} // namespace cuddh
