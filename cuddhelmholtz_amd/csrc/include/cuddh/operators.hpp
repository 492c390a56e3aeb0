// Matrix-free finite element operators on an H1Space / FaceSpace (fp64).
// Contracts: reference include/StiffnessMatrix.hpp:14-25, include/MassMatrix.hpp:19-61,
// include/FaceMassMatrix.hpp:16-54.  `action(x, y)` overwrites y, `action(c, x, y)`
// accumulates; all vectors are DEVICE pointers; x and y must not alias except for
// the diagonal operators.  The operators keep references to their spaces.
#ifndef CUDDH_AMD_OPERATORS_HPP
#define CUDDH_AMD_OPERATORS_HPP

#include "blas1.hpp"
#include "memory.hpp"
#include "operator.hpp"
#include "spaces.hpp"

namespace cuddh
{
    /// (grad u, grad phi)
    class StiffnessMatrix : public Operator
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

    private:
        void setup(const QuadratureRule &quad);

        const H1Space &fem;
        const int ndof, n_elem, n_basis, n_quad;
        host_device_dvec _P, _D, _G;
    };

    /// (a u, phi)
    class MassMatrix : public Operator
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

    private:
        void setup(const double *a);

        const H1Space &fem;
        const int ndof, n_elem, n_basis, n_quad;
        host_device_dvec _P, _a;
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
