// Lagrange basis on Gauss-Lobatto nodes in barycentric form
// (evaluation semantics of reference source/Basis.cpp:140-170).
#include "cuddh/basis.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace cuddh
{
    namespace
    {
        // index of the node x0 coincides with (to one ulp of 1), or -1
        int coincident_node(double x0, const_dvec_wrapper nodes)
        {
            constexpr double eps = std::numeric_limits<double>::epsilon();
            for (int k = 0; k < nodes.size(); ++k)
                if (std::abs(x0 - nodes[k]) <= eps)
                    return k;
            return -1;
        }
    } // namespace

    Basis::Basis(int n_) : n(n_), q(n_, QuadratureRule::GaussLobatto), bw(n_), M(n_, n_), Dn(n_, n_)
    {
        // barycentric weights 1 / prod_{j != i} (x_i - x_j), scaled by their range
        for (int i = 0; i < n; ++i)
        {
            double prod = 1.0;
            for (int j = 0; j < n; ++j)
                if (j != i)
                    prod *= q.x(i) - q.x(j);
            bw[i] = 1.0 / prod;
        }
        const auto mm = std::minmax_element(bw.begin(), bw.end());
        const double range = *mm.second - *mm.first;
        for (int i = 0; i < n; ++i)
            bw[i] /= range;

        QuadratureRule gl(n, QuadratureRule::GaussLegendre);
        dmat P(n, n);
        eval(n, gl.x(), P);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j)
            {
                double s = 0.0;
                for (int k = 0; k < n; ++k)
                    s += gl.w(k) * P(k, i) * P(k, j);
                M(i, j) = s;
                M(j, i) = s;
            }

        deriv(n, q.x(), Dn);
    }

    void Basis::eval(int m, const double *x, double *P_) const
    {
        auto P = reshape(P_, m, n);
        for (int r = 0; r < m; ++r)
        {
            const int hit = coincident_node(x[r], q.x());
            if (hit >= 0)
            {
                for (int j = 0; j < n; ++j)
                    P(r, j) = (j == hit) ? 1.0 : 0.0;
                continue;
            }
            double denom = 0.0;
            for (int k = 0; k < n; ++k)
                denom += bw[k] / (x[r] - q.x(k));
            for (int j = 0; j < n; ++j)
                P(r, j) = (bw[j] / (x[r] - q.x(j))) / denom;
        }
    }

    void Basis::deriv(int m, const double *x, double *D_) const
    {
        auto D = reshape(D_, m, n);
        for (int r = 0; r < m; ++r)
        {
            const int hit = coincident_node(x[r], q.x());
            if (hit >= 0)
            {
                // phi_j'(x_i) = (w_j / w_i) / (x_i - x_j), rows sum to zero
                double diag = 0.0;
                for (int j = 0; j < n; ++j)
                {
                    if (j == hit)
                        continue;
                    const double v = (bw[j] / bw[hit]) / (q.x(hit) - q.x(j));
                    D(r, j) = v;
                    diag -= v;
                }
                D(r, hit) = diag;
                continue;
            }
            // phi_j = (w_j / (x - x_j)) / s,  s = sum_k w_k / (x - x_k)
            // phi_j' = phi_j * ( -1/(x - x_j) - s'/s ),  s' = -sum_k w_k / (x - x_k)^2
            double s = 0.0, ds = 0.0;
            for (int k = 0; k < n; ++k)
            {
                const double inv = 1.0 / (x[r] - q.x(k));
                s += bw[k] * inv;
                ds -= bw[k] * inv * inv;
            }
            for (int j = 0; j < n; ++j)
            {
                const double inv = 1.0 / (x[r] - q.x(j));
                D(r, j) = (bw[j] * inv / s) * (-inv - ds / s);
            }
        }
    }
} // namespace cuddh
