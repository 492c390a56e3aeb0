// Substructured domain-decomposition Helmholtz solver (WaveHoltz local solves).
// Contract: reference include/DDH.hpp:21-84.  Call sequence
// (reference examples/DDH.cpp:141-144):  F.rhs(f, b); gmres(F.size(), lambda, &F, b, ...);
// F.postprocess(lambda, f, u).  `action` applies I - T to the interface traces.
//
// DDH works in fp32 like the reference.  DDH64 is the same algorithm in fp64
// with double traces; it exists so parity against the fp64 oracle can be gated
// at 1e-10 (the reference's float LDS atomics make its own results
// order-dependent, source/DDH.cpp:108).
#ifndef CUDDH_AMD_DDH_HPP
#define CUDDH_AMD_DDH_HPP

#include <memory>

#include "blas1.hpp"
#include "ensemble.hpp"
#include "krylov.hpp"
#include "memory.hpp"
#include "operator.hpp"
#include "operators.hpp"

struct cuddh_ddh_plan;

namespace cuddh
{
    namespace detail
    {
        /// setup shared by DDH and DDH64 (scalar = float or double)
        template <typename Real>
        class DDHCore
        {
        public:
            DDHCore(double omega, const double *h_a, const H1Space &fem, int nx, int ny, int kernel);
            ~DDHCore();

            int n_traces() const { return 2 * n_lambda; }
            int num_domains() const { return n_domains; }
            int num_steps() const { return nt; }
            double time_step() const { return dt; }
            int kernel_kind() const;
            /// WaveHoltz iterations per local solve; the reference hard-wires 5 (source/DDH.cpp:136), the default.
            /// Verification knob (tests/test_ddh_physics.py), see cuddh_hip_ddh_plan_set_wh_iters.
            void set_waveholtz_iterations(int n) const;
            /// The wavefronts of the following solve() calls take issue priority over other resident work (s_setprio): for the
            /// multi-GPU schedule that runs the subdomains other ranks wait for beside the rest (cuddh_hip_ddh_plan_set_wave_priority).
            void set_wave_priority(bool high) const;
            const EnsembleSpace &ensemble() const { return *efem; }

            /// runs the local solves of subdomains [dom_begin, dom_end)
            void solve(int dom_begin, int dom_end, const double *x, double *y, bool zero_y, const Real *lambda,
                       Real *update) const;
            /// traces only, for the n subdomains listed in d_domains (DEVICE), one launch (cuddh_hip_ddh_apply_list_*)
            void solve_listed(const int *d_domains, int n, const double *x, const Real *lambda, Real *update) const;
            /// the general form for listed subdomains (solution output y as in solve())
            void solve_listed(const int *d_domains, int n, const double *x, double *y, bool zero_y, const Real *lambda, Real *update) const;

            // host copies of the constructor's tables (tests compare them with the oracle)
            const host_device_ivec &table_B() const { return _Bf; }
            const host_device_ivec &table_gI() const { return _gI; }
            const host_device_ivec &table_sI() const { return _sI; }
            const HostDeviceArray<Real> &table_D() const { return _D; }
            const HostDeviceArray<Real> &table_G() const
            {
                ensure_plan();
                return _g_tensor;
            }
            const HostDeviceArray<Real> &table_m() const { return _m; }
            const HostDeviceArray<Real> &table_gmi() const { return _gmi; }
            const HostDeviceArray<Real> &table_a() const { return _a; }
            const HostDeviceArray<Real> &table_H() const { return _H; }
            const HostDeviceArray<Real> &table_filter() const { return _wh_filter; }
            const HostDeviceArray<Real> &table_cs() const { return _cs; }
            const HostDeviceArray<Real> &table_sn() const { return _sn; }
            int max_dof() const { return mx_dof; }
            int max_fdof() const { return mx_fdof; }
            int elems_per_side() const { return nel1d; }

        private:
            void solve_impl(const int *d_list, int d0, int d1, const double *x, double *y, bool zero_y, const Real *lambda, Real *update) const;

            /// device-side part of the set-up (geometric factors, kernel plan); deferred to first use so
            /// that the host tables can be built and inspected without a GPU
            void ensure_plan() const;
            /// tables of the fixed-order assembly of y (first call with y != nullptr)
            void ensure_assembly() const;

            int g_ndof, g_elem, n_basis, n_domains, n_lambda, nt, mx_dof, mx_fdof, mx_elem_per_dom, nel1d;
            double omega, dt;
            const Mesh2D *fem_mesh;
            const Basis *fem_basis;
            int requested_kernel = 0;

            host_device_ivec _Bf, _gI, _sI;
            HostDeviceArray<Real> _D, _m, _gmi, _H, _wh_filter, _cs, _sn, _a;
            mutable HostDeviceArray<Real> _g_tensor;
            std::unique_ptr<EnsembleSpace> efem;
            mutable cuddh_ddh_plan *plan = nullptr;
            // y = sum over subdomains of weighted local solutions, assembled in a fixed order instead of with atomics:
            // identity numbering of the subdomain dofs, their forcing / solution in that numbering, and per global dof the
            // list of subdomain dofs that are copies of it (increasing subdomain, like the reference's serial meaning)
            mutable host_device_ivec _gI_local, _csr_off, _csr_src;
            mutable host_device_dvec _x_local, _y_local;
        };
    } // namespace detail

    class DDH : public SinglePrecisionOperator
    {
    public:
        /// @param h_a HOST nodal coefficient a(x); @param fem space on a Mesh2D::uniform_rect(nx, ..., ny, ...) mesh
        DDH(double omega, const double *h_a, const H1Space &fem, int nx, int ny);
        /// extension: pick the local-solve kernel (0 auto, 1 generic workgroup, 2 wavefront-per-subdomain)
        DDH(double omega, const double *h_a, const H1Space &fem, int nx, int ny, int kernel);
        ~DDH() = default;

        /// dimension of the substructured problem
        int size() const { return core.n_traces(); }

        void rhs(const double *f, float *b) const;
        void postprocess(const float *lambda, const double *f, double *u) const;
        void action(const float *x, float *y) const override;

        /// extension for the multi-GPU path: update <- T_local(lambda) over subdomains [d0, d1) only
        /// (slots written by other subdomains are left untouched); no `lambda - update` step.
        void local_traces(int d0, int d1, const double *f, const float *lambda, float *update) const;
        void local_solution(int d0, int d1, const float *lambda, const double *f, double *u, bool zero_u) const;

        const detail::DDHCore<float> &internals() const { return core; }

    private:
        detail::DDHCore<float> core;
    };

    class DDH64 : public Operator
    {
    public:
        DDH64(double omega, const double *h_a, const H1Space &fem, int nx, int ny, int kernel = 0);

        int size() const { return core.n_traces(); }

        void rhs(const double *f, double *b) const;
        void postprocess(const double *lambda, const double *f, double *u) const;
        void action(const double *x, double *y) const override;
        /// not defined for the substructured operator
        void action(double c, const double *x, double *y) const override;

        void local_traces(int d0, int d1, const double *f, const double *lambda, double *update) const;
        void local_solution(int d0, int d1, const double *lambda, const double *f, double *u, bool zero_u) const;

        const detail::DDHCore<double> &internals() const { return core; }

    private:
        detail::DDHCore<double> core;
    };
} // namespace cuddh

#endif
