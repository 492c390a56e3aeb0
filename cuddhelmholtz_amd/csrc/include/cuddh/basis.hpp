// 1-D nodal Lagrange basis on Gauss-Lobatto nodes.
// Contract: reference include/Basis.hpp:12-63; eval/deriv fill column-major
// (m, n_basis) matrices: P(i,j) = phi_j(x_i), D(i,j) = phi_j'(x_i).
#ifndef CUDDH_AMD_BASIS_HPP
#define CUDDH_AMD_BASIS_HPP

#include "quadrature.hpp"
#include "tensor.hpp"

namespace cuddh
{
    class Basis
    {
    public:
        /// n = number of Gauss-Lobatto nodes (= polynomial degree + 1)
        explicit Basis(int n);

        int size() const { return n; }

        void eval(int m, const double *x, double *P) const;
        void deriv(int m, const double *x, double *D) const;

        const_dmat_wrapper mass_matrix() const { return const_dmat_wrapper(M.data(), n, n); }
        const_dmat_wrapper derivative_matrix() const { return const_dmat_wrapper(Dn.data(), n, n); }
        const QuadratureRule &quadrature() const { return q; }

    private:
        int n;
        QuadratureRule q;
        dvec bw; // barycentric weights of the nodes
        dmat M;  // (phi_i, phi_j) with the n-point Gauss-Legendre rule
        dmat Dn; // collocation derivative matrix
    };
} // namespace cuddh

#endif
