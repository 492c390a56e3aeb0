// Gauss-Legendre / Gauss-Lobatto nodes and weights by Newton iteration on the
// three-term Legendre recurrence.  Weight formulas as in the reference
// (source/QuadratureRule.cpp:130-131: w = 2 / ((1-x^2) P_n'(x)^2);
//  :201: w = 2 / (n (n-1) P_{n-1}(x)^2)).
#include "cuddh/quadrature.hpp"

#include <cmath>
#include <iomanip>
#include <sstream>

namespace
{
    // P_n(x) and P_n'(x)
    void legendre(int n, double x, double &p, double &dp)
    {
        double p0 = 1.0, p1 = x;
        if (n == 0)
        {
            p = 1.0;
            dp = 0.0;
            return;
        }
        for (int k = 2; k <= n; ++k)
        {
            const double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
            p0 = p1;
            p1 = pk;
        }
        p = p1;
        // (1 - x^2) P_n' = n (P_{n-1} - x P_n); at the end points use the closed form
        if (std::abs(x) == 1.0)
            dp = 0.5 * n * (n + 1.0) * ((x < 0 && n % 2 == 0) ? -1.0 : 1.0); // P_n'(+-1) = (+-1)^(n+1) n(n+1)/2
        else
            dp = n * (p0 - x * p1) / (1.0 - x * x);
    }

    void gauss_legendre(int n, double *x, double *w)
    {
        if (n < 1)
            cuddh::cuddh_error("QuadratureRule error: Gauss-Legendre rules require n >= 1.");
        for (int i = 0; i < n / 2; ++i)
        {
            double t = -std::cos(M_PI * (i + 0.75) / (n + 0.5));
            for (int it = 0; it < 100; ++it)
            {
                double p, dp;
                legendre(n, t, p, dp);
                const double step = p / dp;
                t -= step;
                if (std::abs(step) <= 4e-16 * std::abs(t))
                    break;
            }
            x[i] = t;
            x[n - 1 - i] = -t;
        }
        if (n & 1)
            x[n / 2] = 0.0;
        for (int i = 0; i < n; ++i)
        {
            double p, dp;
            legendre(n, x[i], p, dp);
            w[i] = 2.0 / ((1.0 - x[i] * x[i]) * dp * dp);
        }
    }

    void gauss_lobatto(int n, double *x, double *w)
    {
        if (n < 2)
            cuddh::cuddh_error("QuadratureRule error: Gauss-Lobatto rules require n >= 2.");
        const int N = n - 1; // interior nodes are the roots of P_N'
        x[0] = -1.0;
        x[n - 1] = 1.0;
        for (int i = 1; i < n / 2; ++i)
        {
            double t = -std::cos(M_PI * i / N);
            for (int it = 0; it < 100; ++it)
            {
                double p, dp;
                legendre(N, t, p, dp);
                // q = P_N', q' = P_N'' = (2 x P_N' - N (N+1) P_N) / (1 - x^2)
                const double ddp = (2.0 * t * dp - N * (N + 1.0) * p) / (1.0 - t * t);
                const double step = dp / ddp;
                t -= step;
                if (std::abs(step) <= 4e-16 * std::abs(t))
                    break;
            }
            x[i] = t;
            x[n - 1 - i] = -t;
        }
        if (n & 1)
            x[n / 2] = 0.0;
        for (int i = 0; i < n; ++i)
        {
            double p, dp;
            legendre(N, x[i], p, dp);
            w[i] = 2.0 / (n * (n - 1.0) * p * p);
        }
    }
} // namespace

namespace cuddh
{
    QuadratureRule::QuadratureRule() : _n(0), _type(GaussLobatto) {}

    QuadratureRule::QuadratureRule(int n, QuadratureType type) : _n(n), _type(type), _x(n), _w(n)
    {
        if (type == GaussLegendre)
            gauss_legendre(n, _x, _w);
        else
            gauss_lobatto(n, _x, _w);
    }

    std::string QuadratureRule::name() const
    {
        std::ostringstream s;
        s << (_type == GaussLegendre ? "legendre" : "lobatto") << std::setw(5) << std::setfill('0') << _n;
        return s.str();
    }
} // namespace cuddh
