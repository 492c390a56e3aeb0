/*
 * oracle.c -- CPU restatement of the reference's hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cuddhelmholtz_amd/ may include, link
 * or call this file; it is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / reported baseline.
 *
 * Every function restates one reference kernel in plain C, with the reference's
 * data layouts (column major, first index fastest) and loop structure, and cites
 * the reference lines it follows (paths relative to arotem3/CuDDHelmholtz).
 * Accumulation is serial per output in a fixed order (the reference uses
 * atomics, whose order is not defined).
 *
 * Pinning: quadrature, basis, mass, stiffness and GMRES are pinned by the
 * reference's own known-answer tests restated in tests/test_oracle_pins.py
 * (tests/quadrature_rule.cpp, tests/basis.cpp, tests/mass.cpp,
 * tests/stiffness.cpp, tests/gmres.cpp).  The reference has no test of DDH,
 * EnsembleSpace, FaceMassMatrix or FaceSpace: for those functions
 * PARITY IS UNPINNED by the reference.  Independent evidence for
 * the DDH restatement: tests/test_ddh_physics.py compares its local solves with a
 * continuous-in-time WaveHoltz model and its converged solution with a direct
 * solve of the Helmholtz system DDH's transmission conditions imply
 * (tests/helmholtz_direct.py; neither follows source/DDH.cpp's loops or tables).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ quadrature
 * Nodes: roots of P_n (Gauss-Legendre) / of P'_{n-1} plus the end points
 * (Gauss-Lobatto), found by Newton iteration; the reference tabulates them for
 * n <= 10 / 9 and otherwise polishes LAPACK eigenvalues with three Newton steps
 * (source/QuadratureRule.cpp:64-127,134-197).  Weights exactly as
 * source/QuadratureRule.cpp:130-131 and :200-201. */
static void legendre(int n, double x, double *p, double *pm1)
{
    double a = 1.0, b = x;
    if (n == 0) { *p = 1.0; *pm1 = 0.0; return; }
    for (int k = 2; k <= n; ++k)
    {
        const double c = ((2.0 * k - 1.0) * x * b - (k - 1.0) * a) / k;
        a = b;
        b = c;
    }
    *p = b;
    *pm1 = a;
}

void orc_gauss_legendre(int n, double *x, double *w)
{
    for (int i = 0; i < n / 2; ++i)
    {
        double t = -cos(M_PI * (i + 0.75) / (n + 0.5));
        for (int it = 0; it < 200; ++it)
        {
            double p, pm1;
            legendre(n, t, &p, &pm1);
            const double dp = n * (pm1 - t * p) / (1.0 - t * t);
            const double dt = p / dp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        x[i] = t;
        x[n - 1 - i] = -t;
    }
    if (n & 1) x[n / 2] = 0.0;
    for (int i = 0; i < n; ++i)
    {
        double p, pm1;
        legendre(n, x[i], &p, &pm1);
        const double dp = n * (pm1 - x[i] * p) / (1.0 - x[i] * x[i]);
        w[i] = 2.0 / (1.0 - x[i] * x[i]) / (dp * dp);
    }
}

void orc_gauss_lobatto(int n, double *x, double *w)
{
    const int N = n - 1;
    x[0] = -1.0;
    x[n - 1] = 1.0;
    for (int i = 1; i < n / 2; ++i)
    {
        double t = -cos(M_PI * i / N);
        for (int it = 0; it < 200; ++it)
        {
            double p, pm1;
            legendre(N, t, &p, &pm1);
            const double dp = N * (pm1 - t * p) / (1.0 - t * t);
            const double ddp = (2.0 * t * dp - N * (N + 1.0) * p) / (1.0 - t * t);
            const double dt = dp / ddp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        x[i] = t;
        x[n - 1 - i] = -t;
    }
    if (n & 1) x[n / 2] = 0.0;
    for (int i = 0; i < n; ++i)
    {
        double p, pm1;
        legendre(N, x[i], &p, &pm1);
        w[i] = 2.0 / (n * (n - 1.0) * p * p);
    }
}

/* ------------------------------------------------------------------ basis
 * source/Basis.cpp:3-24 (barycentric weights), :32-53 (interpolation),
 * :61-105 (derivative), :140-170 (eval / deriv tables, shape (m, n)). */
static void bary_weights(int n, const double *x, double *w)
{
    double lo = INFINITY, hi = -INFINITY;
    for (int i = 0; i < n; ++i)
    {
        w[i] = 1.0;
        for (int j = 0; j < n; ++j)
            if (i != j) w[i] *= x[i] - x[j];
        w[i] = 1.0 / w[i];
        if (w[i] < lo) lo = w[i];
        if (w[i] > hi) hi = w[i];
    }
    for (int i = 0; i < n; ++i) w[i] /= (hi - lo);
}

static double lagrange_value(int n, double x0, const double *x, const double *w, const double *y)
{
    double A = 0.0, B = 0.0;
    for (int i = 0; i < n; ++i)
    {
        const double d = x0 - x[i];
        if (x0 == x[i] || fabs(d) <= 2.220446049250313e-16) return y[i];
        const double c = w[i] / d;
        A += c * y[i];
        B += c;
    }
    return A / B;
}

static double lagrange_slope(int n, double x0, const double *x, const double *w, const double *y)
{
    int at = -1;
    double A = 0.0, B = 0.0;
    const double p = lagrange_value(n, x0, x, w, y);
    for (int j = 0; j < n; ++j)
        if (x0 == x[j] || fabs(x0 - x[j]) <= 2.220446049250313e-16) { at = j; B = -w[j]; }
    if (at >= 0)
    {
        for (int j = 0; j < n; ++j)
            if (j != at) A += w[j] * (p - y[j]) / (x0 - x[j]);
    }
    else
    {
        for (int j = 0; j < n; ++j)
        {
            const double t = w[j] / (x0 - x[j]);
            A += t * (p - y[j]) / (x0 - x[j]);
            B += t;
        }
    }
    return A / B;
}

/* nodes: the n Gauss-Lobatto nodes; P, D: (m, n) column major */
void orc_basis_tables(int n, const double *nodes, int m, const double *x, double *P, double *D)
{
    double *w = (double *)malloc(sizeof(double) * n), *y = (double *)calloc(n, sizeof(double));
    bary_weights(n, nodes, w);
    for (int i = 0; i < n; ++i)
    {
        y[i] = 1.0;
        for (int j = 0; j < m; ++j)
        {
            if (P) P[j + m * i] = lagrange_value(n, x[j], nodes, w, y);
            if (D) D[j + m * i] = lagrange_slope(n, x[j], nodes, w, y);
        }
        y[i] = 0.0;
    }
    free(w);
    free(y);
}

/* ------------------------------------------------------------------ geometry
 * source/Element.cpp:5-27: bilinear map of a quad with corners X (2,4), CCW. */
void orc_quad_map(const double *X, double s, double t, double *xy, double *J)
{
    if (xy)
    {
        const double b[4] = {0.25 * (1 - s) * (1 - t), 0.25 * (1 + s) * (1 - t), 0.25 * (1 + s) * (1 + t), 0.25 * (1 - s) * (1 + t)};
        xy[0] = xy[1] = 0.0;
        for (int i = 0; i < 4; ++i) { xy[0] += X[2 * i] * b[i]; xy[1] += X[2 * i + 1] * b[i]; }
    }
    if (J)
    {
        J[0] = 0.25 * ((1 - t) * (X[2] - X[0]) + (1 + t) * (X[4] - X[6]));
        J[1] = 0.25 * ((1 - t) * (X[3] - X[1]) + (1 + t) * (X[5] - X[7]));
        J[2] = 0.25 * ((1 - s) * (X[6] - X[0]) + (1 + s) * (X[4] - X[2]));
        J[3] = 0.25 * ((1 - s) * (X[7] - X[1]) + (1 + s) * (X[5] - X[3]));
    }
}

/* element metrics on the tensor grid of a rule (source/Mesh2D.cpp:173-227):
 * J (2,2,n,n,nel), detJ (n,n,nel), xq (2,n,n,nel); corners (2,4,nel) */
void orc_element_metrics(int nel, const double *corners, int n, const double *q, double *J, double *detJ, double *xq)
{
    for (int el = 0; el < nel; ++el)
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
            {
                double jj[4], xy[2];
                orc_quad_map(corners + 8 * (size_t)el, q[i], q[j], xy, jj);
                const size_t t = i + (size_t)n * (j + (size_t)n * el);
                if (J) memcpy(J + 4 * t, jj, sizeof jj);
                if (detJ) detJ[t] = jj[0] * jj[3] - jj[1] * jj[2];
                if (xq) { xq[2 * t] = xy[0]; xq[2 * t + 1] = xy[1]; }
            }
}

/* ------------------------------------------------------------------ stiffness
 * source/StiffnessMatrix.cpp:5-38 */
void orc_stiffness_setup(int n_elem, int nq, const double *w, const double *J, double *G)
{
    for (size_t el = 0; el < (size_t)n_elem; ++el)
        for (int j = 0; j < nq; ++j)
            for (int i = 0; i < nq; ++i)
            {
                const size_t t = i + (size_t)nq * (j + nq * el);
                const double W = w[i] * w[j];
                const double X_xi = J[4 * t], Y_xi = J[4 * t + 1], X_eta = J[4 * t + 2], Y_eta = J[4 * t + 3];
                const double det = X_xi * Y_eta - X_eta * Y_xi;
                G[3 * t] = W * (Y_eta * Y_eta + X_eta * X_eta) / det;
                G[3 * t + 1] = -W * (Y_xi * Y_eta + X_xi * X_eta) / det;
                G[3 * t + 2] = W * (Y_xi * Y_xi + X_xi * X_xi) / det;
            }
}

#define MAXQ 64

/* source/StiffnessMatrix.cpp:83-184: y[I] += c S x; P, D (nq, nb); G (3,nq,nq,nel); I (nb,nb,nel) */
void orc_stiffness_apply(int n_elem, int nq, int nb, const double *P, const double *D, const double *G, const int *I, double c,
                         const double *x, double *y)
{
    static double u[MAXQ][MAXQ], Pu[MAXQ][MAXQ], Du[MAXQ][MAXQ], F[MAXQ][MAXQ][2];
    for (int el = 0; el < n_elem; ++el)
    {
        const int *Ie = I + (size_t)nb * nb * el;
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx) u[tx][ty] = x[Ie[tx + nb * ty]];
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double pxu = 0.0, dxu = 0.0;
                for (int k = 0; k < nb; ++k)
                {
                    pxu += P[tx + nq * k] * u[k][ty];
                    dxu += D[tx + nq * k] * u[k][ty];
                }
                Pu[tx][ty] = pxu;
                Du[tx][ty] = dxu;
            }
        for (int ty = 0; ty < nq; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                const double *g = G + 3 * (tx + (size_t)nq * (ty + (size_t)nq * el));
                double Dx = 0.0, Dy = 0.0;
                for (int l = 0; l < nb; ++l)
                {
                    Dx += P[ty + nq * l] * Du[tx][l];
                    Dy += D[ty + nq * l] * Pu[tx][l];
                }
                F[tx][ty][0] = g[0] * Dx + g[1] * Dy;
                F[tx][ty][1] = g[1] * Dx + g[2] * Dy;
            }
        for (int ty = 0; ty < nq; ++ty)
            for (int tx = 0; tx < nb; ++tx)
            {
                double df = 0.0, pg = 0.0;
                for (int i = 0; i < nq; ++i)
                {
                    df += D[i + nq * tx] * F[i][ty][0];
                    pg += P[i + nq * tx] * F[i][ty][1];
                }
                Du[tx][ty] = df;
                Pu[tx][ty] = pg;
            }
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx)
            {
                double Su = 0.0;
                for (int j = 0; j < nq; ++j) Su += P[j + nq * ty] * Du[tx][j] + D[j + nq * ty] * Pu[tx][j];
                y[Ie[tx + nb * ty]] += c * Su;
            }
    }
}

/* ------------------------------------------------------------------ mass
 * source/MassMatrix.cpp:5-67: a(q,r,el) = (sum P P coef) w_q w_r detJ; coef may be NULL */
void orc_mass_setup(int n_elem, int nq, int nb, const double *coef, const double *detJ, const double *w, const int *I, const double *P,
                    double *a)
{
    static double Q[MAXQ][MAXQ], z[MAXQ][MAXQ];
    for (int el = 0; el < n_elem; ++el)
    {
        const int *Ie = I + (size_t)nb * nb * el;
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx) Q[tx][ty] = coef ? coef[Ie[tx + nb * ty]] : 1.0;
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double px = 0.0;
                for (int k = 0; k < nb; ++k) px += P[tx + nq * k] * Q[k][ty];
                z[tx][ty] = px;
            }
        for (int ty = 0; ty < nq; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double ppx = 0.0;
                for (int l = 0; l < nb; ++l) ppx += P[ty + nq * l] * z[tx][l];
                const size_t t = tx + (size_t)nq * (ty + (size_t)nq * el);
                a[t] = ppx * (w[tx] * w[ty] * detJ[t]);
            }
    }
}

/* source/MassMatrix.cpp:137-211 */
void orc_mass_apply(int n_elem, int nq, int nb, const int *I, const double *P, const double *a, double c, const double *x, double *y)
{
    static double u[MAXQ][MAXQ], Pu[MAXQ][MAXQ];
    for (int el = 0; el < n_elem; ++el)
    {
        const int *Ie = I + (size_t)nb * nb * el;
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx) u[tx][ty] = x[Ie[tx + nb * ty]];
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double pu = 0.0;
                for (int k = 0; k < nb; ++k) pu += P[tx + nq * k] * u[k][ty];
                Pu[tx][ty] = pu;
            }
        for (int ty = 0; ty < nq; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double ppu = 0.0;
                for (int l = 0; l < nb; ++l) ppu += P[ty + nq * l] * Pu[tx][l];
                u[tx][ty] = a[tx + (size_t)nq * (ty + (size_t)nq * el)] * ppu;
            }
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double qu = 0.0;
                for (int j = 0; j < nq; ++j) qu += P[j + nq * ty] * u[tx][j];
                Pu[tx][ty] = qu;
            }
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx)
            {
                double qqu = 0.0;
                for (int i = 0; i < nq; ++i) qqu += P[i + nq * tx] * Pu[i][ty];
                y[Ie[tx + nb * ty]] += c * qqu;
            }
    }
}

/* source/MassMatrix.cpp:241-280: op = 1 / lumped mass */
void orc_diag_mass(int ndof, int n_elem, int nb, const double *coef, const double *detJ, const double *w, const int *I, double *op)
{
    memset(op, 0, sizeof(double) * ndof);
    for (size_t t = 0; t < (size_t)n_elem * nb * nb; ++t)
    {
        const int loc = (int)(t % (nb * nb));
        double m = w[loc % nb] * w[loc / nb] * detJ[t];
        if (coef) m *= coef[I[t]];
        op[I[t]] += m;
    }
    for (int i = 0; i < ndof; ++i) op[i] = 1.0 / op[i];
}

/* ------------------------------------------------------------------ face mass
 * source/FaceMassMatrix.cpp:5-49 and :141-193; I (nb, nf); detJ, a (nq, nf) */
void orc_facemass_setup(int n_faces, int nb, int nq, const double *w, const double *P, const double *detJ, const double *coef, const int *I,
                        double *a)
{
    for (int e = 0; e < n_faces; ++e)
        for (int k = 0; k < nq; ++k)
        {
            double pa = 0.0;
            for (int l = 0; l < nb; ++l) pa += P[k + nq * l] * (coef ? coef[I[l + nb * e]] : 1.0);
            a[k + (size_t)nq * e] = pa * (w[k] * detJ[k + (size_t)nq * e]);
        }
}

void orc_facemass_apply(int n_faces, int nb, int nq, const double *P, const double *a, const int *I, double c, const double *x, double *y)
{
    double Pu[MAXQ * 2];
    for (int f = 0; f < n_faces; ++f)
    {
        for (int k = 0; k < nq; ++k)
        {
            double pu = 0.0;
            for (int l = 0; l < nb; ++l) pu += P[k + nq * l] * x[I[l + nb * f]];
            Pu[k] = pu * a[k + (size_t)nq * f];
        }
        for (int k = 0; k < nb; ++k)
        {
            double Mu = 0.0;
            for (int i = 0; i < nq; ++i) Mu += P[i + nq * k] * Pu[i];
            y[I[k + nb * f]] += c * Mu;
        }
    }
}

/* source/FaceMassMatrix.cpp:225-255 */
void orc_diag_facemass(int ndof, int n_faces, int nb, const double *w, const double *detJ, const double *coef, const int *I, double *op)
{
    memset(op, 0, sizeof(double) * ndof);
    for (int f = 0; f < n_faces; ++f)
        for (int k = 0; k < nb; ++k)
        {
            const int idx = I[k + nb * f];
            op[idx] += (coef ? coef[idx] : 1.0) * (w[k] * detJ[k + (size_t)nb * f]);
        }
    for (int i = 0; i < ndof; ++i) op[i] = 1.0 / op[i];
}

/* collocated load vector, include/LinearFunctional.hpp:110-142: F[I] += c w_i w_j detJ f; fvals = f at the GLL points (nb,nb,nel) */
void orc_lf_collocated(int n_elem, int nb, const double *w, const double *detJ, const double *fvals, const int *I, double c, double *F)
{
    for (size_t t = 0; t < (size_t)n_elem * nb * nb; ++t)
    {
        const int loc = (int)(t % (nb * nb));
        F[I[t]] += fvals[t] * (c * w[loc % nb] * w[loc / nb] * detJ[t]);
    }
}

/* full-quadrature load vector, include/LinearFunctional.hpp:45-108; fvals (nq,nq,nel), P (nq,nb) */
void orc_lf_quadrature(int n_elem, int nq, int nb, const double *w, const double *P, const double *detJ, const double *fvals, const int *I,
                       double c, double *F)
{
    static double g[MAXQ][MAXQ], Pg[MAXQ][MAXQ];
    for (int el = 0; el < n_elem; ++el)
    {
        for (int ty = 0; ty < nq; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                const size_t t = tx + (size_t)nq * (ty + (size_t)nq * el);
                g[tx][ty] = w[tx] * w[ty] * detJ[t] * fvals[t];
            }
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nq; ++tx)
            {
                double qu = 0.0;
                for (int j = 0; j < nq; ++j) qu += P[j + nq * ty] * g[tx][j];
                Pg[tx][ty] = qu;
            }
        for (int ty = 0; ty < nb; ++ty)
            for (int tx = 0; tx < nb; ++tx)
            {
                double qqu = 0.0;
                for (int i = 0; i < nq; ++i) qqu += P[i + nq * tx] * Pg[i][ty];
                F[I[tx + nb * (ty + (size_t)nb * el)]] += c * qqu;
            }
    }
}

/* ------------------------------------------------------------------ DDH local solves, fp32 and fp64 */
/* WaveHoltz iterations per local solve: 5 in the reference (`constexpr int wh_maxit = 5`, source/DDH.cpp:136).
 * tests/test_ddh_physics.py raises it to show that DDH converges to the independently derived fixed point once the
 * truncation of the local solves is taken away; everything else always runs with 5. */
static int orc_wh_iters = 5;
void orc_ddh_set_wh_iters(int n) { orc_wh_iters = n > 0 ? n : 5; }

#define REAL float
#define FN(name) name##_f32
#include "ddh_body.inc"
#undef REAL
#undef FN
#define REAL double
#define FN(name) name##_f64
#include "ddh_body.inc"
#undef REAL
#undef FN
