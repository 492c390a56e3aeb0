// Gauss-Legendre and Gauss-Lobatto rules on [-1, 1].
// Contract: reference include/QuadratureRule.hpp:15-78 (same accessors, same
// `name()` cache key "legendre%05d" / "lobatto%05d").  Nodes come from Newton
// iteration on the Legendre recurrence for every n (no tabulated nodes, no
// LAPACK eigen-solver as in source/QuadratureRule.cpp:64-202); they agree with
// the reference's to rounding.
#ifndef CUDDH_AMD_QUADRATURE_HPP
#define CUDDH_AMD_QUADRATURE_HPP

#include <string>

#include "tensor.hpp"

namespace cuddh
{
    class QuadratureRule
    {
    public:
        enum QuadratureType
        {
            GaussLegendre,
            GaussLobatto
        };

        QuadratureRule();
        QuadratureRule(int n, QuadratureType type = GaussLobatto);

        QuadratureRule(const QuadratureRule &) = default;
        QuadratureRule(QuadratureRule &&) = default;
        QuadratureRule &operator=(const QuadratureRule &) = default;
        QuadratureRule &operator=(QuadratureRule &&) = default;

        int size() const { return _n; }
        QuadratureType type() const { return _type; }
        std::string name() const;

        const_dvec_wrapper x() const { return const_dvec_wrapper(_x.data(), _n); }
        double x(int i) const { return _x[i]; }
        const_dvec_wrapper w() const { return const_dvec_wrapper(_w.data(), _n); }
        double w(int i) const { return _w[i]; }

    private:
        int _n;
        QuadratureType _type;
        dvec _x;
        dvec _w;
    };
} // namespace cuddh

#endif
