// Matrix-free finite element operators on an H1Space / FaceSpace (fp64).
// Contracts: reference include/StiffnessMatrix.hpp:14-25, include/MassMatrix.hpp:19-61,
// include/FaceMassMatrix.hpp:16-54.  `action(x, y)` overwrites y, `action(c, x, y)`
// accumulates; all vectors are DEVICE pointers; x and y must not alias except for
// the diagonal operators.  The operators keep references to their spaces.
#ifndef CUDDH_AMD_OPERATORS_HPP
#define CUDDH_AMD_OPERATORS_HPP

#include <cstddef>
#include <string>

#include "blas1.hpp"
#include "memory.hpp"
#include "operator.hpp"
#include "spaces.hpp"

struct cuddh_helmholtz_plan;

namespace cuddh
{
    namespace detail
    {
        /// Patch plan of one element operator (include/cuddh_hip.h, cuddh_hip_operator_plan_*): built on the first
        /// action(), null when no specialised kernel exists for (n_basis, n_quad) or CUDDH_OPERATOR_PLAN=0.
        class OperatorPlan
        {
        public:
            OperatorPlan() = default;
            ~OperatorPlan();
            OperatorPlan(const OperatorPlan &) = delete;
            OperatorPlan &operator=(const OperatorPlan &) = delete;

            /// kind 0: stiffness (metric = G), 1: mass (metric = a); returns the plan or nullptr
            const cuddh_helmholtz_plan *get(int kind, const H1Space &fem, int n_quad, const double *h_P, const double *h_D,
                                            const double *d_metric) const;
            std::size_t bytes(bool actual) const;
            std::size_t bytes_affine() const;
            /// kernel instantiation the plan launches ("generic" when there is no plan (yet))
            std::string kernel_name() const;

        private:
            mutable cuddh_helmholtz_plan *plan = nullptr;
            mutable bool tried = false;
        };
    } // namespace detail

    /// (grad u, grad phi)
    class StiffnessMatrix : public Operator, public QueuesDeviceWorkOnly
    {
    public:
        /// Gauss-Legendre rule with n_basis + 1 points
        explicit StiffnessMatrix(const H1Space &fem);
        StiffnessMatrix(const H1Space &fem, const QuadratureRule &quad);

        void action(double c, const double *x, double *y) const override;
        void action(const double *x, double *y) const override;

        // raw tables, used by the fused Helmholtz operator
        int quad_size() const { return n_quad; }
        const host_device_dvec &P() const { return _P; }
        const host_device_dvec &D() const { return _D; }
        const host_device_dvec &G() const { return _G; }
        /// bytes one action() moves through the patch plan (0 when the generic kernel is in use)
        std::size_t bytes_per_apply(bool actual = false) const { return plan.bytes(actual); }
        /// kernel instantiation action() launches (after the first action(); "generic" = element_ops.hip)
        std::string kernel_name() const { return plan.kernel_name(); }

    private:
        void setup(const QuadratureRule &quad);

        const H1Space &fem;
        const int ndof, n_elem, n_basis, n_quad;
        host_device_dvec _P, _D, _G;
        detail::OperatorPlan plan;
    };

    /// (a u, phi)
    class MassMatrix : public Operator, public QueuesDeviceWorkOnly
    {
    public:
        /// a == 1; Gauss-Legendre rule with n_basis + 1 points
        explicit MassMatrix(const H1Space &fem);
        /// a: DEVICE nodal coefficient; rule with 1 + 3*n_basis/2 + 1 points
        MassMatrix(const double *a, const H1Space &fem);

        void action(double c, const double *x, double *y) const override;
        void action(const double *x, double *y) const override;

        int quad_size() const { return n_quad; }
        const host_device_dvec &P() const { return _P; }
        const host_device_dvec &weights() const { return _a; }
        std::size_t bytes_per_apply(bool actual = false) const { return plan.bytes(actual); }
        std::string kernel_name() const { return plan.kernel_name(); }

    private:
        void setup(const double *a);

        const H1Space &fem;
        const int ndof, n_elem, n_basis, n_quad;
        host_device_dvec _P, _a;
        detail::OperatorPlan plan;
    };

    /// inverse of the Gauss-Lobatto lumped mass matrix
    class DiagInvMassMatrix : public Operator
    {
    public:
        explicit DiagInvMassMatrix(const H1Space &fem);
        DiagInvMassMatrix(const double *a, const H1Space &fem);

        void action(double c, const double *x, double *y) const override;
        void action(const double *x, double *y) const override;

    private:
        void setup(const double *a);

        const H1Space &fem;
        const int ndof;
        host_device_dvec _p;
    };

    /// <a u, phi> on the faces of a FaceSpace; vectors are FaceSpace vectors
    class FaceMassMatrix : public Operator
    {
    public:
        explicit FaceMassMatrix(const FaceSpace &fs);
        /// a: DEVICE FaceSpace vector
        FaceMassMatrix(const double *a, const FaceSpace &fs);

        void action(double c, const double *x, double *y) const override;
        void action(const double *x, double *y) const override;

        int quad_size() const { return n_quad; }
        const host_device_dvec &P() const { return _P; }
        const host_device_dvec &weights() const { return _a; }

    private:
        void setup(const double *a);

        const FaceSpace &fs;
        const int ndof, n_faces, n_basis, n_quad;
        host_device_dvec _a, _P;
    };

    class DiagInvFaceMassMatrix : public Operator
    {
    public:
        explicit DiagInvFaceMassMatrix(const FaceSpace &fs);
        DiagInvFaceMassMatrix(const double *a, const FaceSpace &fs);

        void action(double c, const double *x, double *y) const override;
        void action(const double *x, double *y) const override;

    private:
        void setup(const double *a, const FaceSpace &fs);

        const int ndof;
        host_device_dvec inv_m;
    };
} // namespace cuddh

#endif
