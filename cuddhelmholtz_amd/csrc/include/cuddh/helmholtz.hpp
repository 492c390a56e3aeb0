// Fused complex Helmholtz operator  [u; v] -> [S u - w^2 M u - w H v ; -(S v - w^2 M v + w H u)].
// Same mathematics and the same result vector as the example-level composite of
// reference examples/Helmholtz.hpp:10-81 (2 memsets + 11 launches per apply there);
// here one plan-based launch pair reads the metrics, the index map and x once.
#ifndef CUDDH_AMD_HELMHOLTZ_HPP
#define CUDDH_AMD_HELMHOLTZ_HPP

#include <cstddef>
#include <string>

#include "krylov.hpp"
#include "operator.hpp"
#include "operators.hpp"
#include "spaces.hpp"

struct cuddh_helmholtz_plan;

namespace cuddh
{
    class HelmholtzOperator : public Operator, public QueuesDeviceWorkOnly
    {
    public:
        /// a2x: DEVICE H1 nodal values of a^2(x); ax: DEVICE FaceSpace values of a(x)
        HelmholtzOperator(double omega, const double *a2x, const double *ax, const H1Space &fem, const FaceSpace &fs);
        ~HelmholtzOperator();

        HelmholtzOperator(const HelmholtzOperator &) = delete;
        HelmholtzOperator &operator=(const HelmholtzOperator &) = delete;

        /// y <- A x with x = [u; v], y = [Au; Av] (each of length fem.size())
        void action(const double *x, double *y) const override;
        /// not implemented (neither is it in the reference example)
        void action(double c, const double *x, double *y) const override;

        /// the same result through the separate operators, launch by launch like the reference example
        void action_unfused(const double *x, double *y) const;

        /// true when the plan-based kernel is in use (false: fell back to the separate operators)
        bool fused() const { return plan != nullptr; }
        /// the kernel plan (diagnostics: cuddh_hip_helmholtz_plan_read_stamps)
        const cuddh_helmholtz_plan *kernel_plan() const { return plan; }
        /// kernel instantiation of the fused apply ("unfused" when it fell back to the separate operators)
        std::string kernel_name() const;

        /// Plan-native vector ordering (cuddh_hip.h: cuddh_hip_helmholtz_apply_native).  A Krylov solver only needs the operator and
        /// inner products, so it can keep its vectors in the order the kernel likes best -- pairs (u, v), a patch's owned dofs
        /// contiguous -- and permute once at entry and exit.  has_native(): the plan offers it (lane form); to_native / from_native:
        /// [u; v] <-> native (2 * size doubles each); action_native: action() on native vectors, bitwise the same numbers.
        bool has_native() const;
        void to_native(const double *x, double *z) const;
        void from_native(const double *z, double *y) const;
        void action_native(const double *z_in, double *z_out) const;
        /// gmres(2 * fem.size(), x, this, b, m, maxit, tol, ...) with the iteration vectors in native ordering when the plan has one
        /// (x and b stay in the reference ordering: they are permuted at entry and exit); the plain gmres() otherwise
        solver_out gmres(double *x, const double *b, int m, int maxit, double tol = 1e-6, int verbose = 0, double max_seconds = 6 * 60 * 60) const;

        /// bytes per apply: algorithmic (SURVEY 8d formula) or as laid out by the plan
        std::size_t bytes_per_apply(bool actual) const;
        /// bytes the native apply moves as laid out (0 without a native ordering)
        std::size_t bytes_native() const;
        /// bytes of the "affine" form when the plan found the stiffness metric identical in every element
        /// (uniform meshes: it is then read from one small table instead of n_elem copies); 0 otherwise
        std::size_t bytes_affine() const;

    private:
        const double omega;
        const int ndof, fdof;
        const H1Space &fem;
        const FaceSpace &fs;
        StiffnessMatrix S;
        MassMatrix M;
        FaceMassMatrix H;
        mutable host_device_dvec xf, yf;
        cuddh_helmholtz_plan *plan = nullptr;
    };
} // namespace cuddh

#endif
