// Bilinear quadrilateral map (conventions of reference source/Element.cpp:5-36).
#include "cuddh/geometry.hpp"

namespace cuddh
{
    void QuadElement::physical_coordinates(const double *xi, double *x) const
    {
        const double s = xi[0], t = xi[1];
        const double N[4] = {0.25 * (1 - s) * (1 - t), 0.25 * (1 + s) * (1 - t), 0.25 * (1 + s) * (1 + t), 0.25 * (1 - s) * (1 + t)};
        x[0] = x[1] = 0.0;
        for (int c = 0; c < 4; ++c)
        {
            x[0] += xc[c][0] * N[c];
            x[1] += xc[c][1] * N[c];
        }
    }

    void QuadElement::jacobian(const double *xi, double *J) const
    {
        const double s = xi[0], t = xi[1];
        for (int a = 0; a < 2; ++a)
        {
            // d/dxi along the bottom (c0->c1) and top (c3->c2) edges, d/deta along left (c0->c3) and right (c1->c2)
            J[a] = 0.25 * ((1 - t) * (xc[1][a] - xc[0][a]) + (1 + t) * (xc[2][a] - xc[3][a]));
            J[2 + a] = 0.25 * ((1 - s) * (xc[3][a] - xc[0][a]) + (1 + s) * (xc[2][a] - xc[1][a]));
        }
    }

    double QuadElement::area() const
    {
        const double centre[2] = {0.0, 0.0};
        return 4.0 * measure(centre); // det J is bilinear: the midpoint rule is exact
    }
} // namespace cuddh
