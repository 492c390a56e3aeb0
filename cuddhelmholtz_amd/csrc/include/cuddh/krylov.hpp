// Restarted GMRES(m) on DEVICE vectors.
// Contract: reference include/gmres.hpp:14-36 and the iteration of
// source/gmres.cpp:91-235: r = b - A x; at most maxit-1 restart cycles
// (`for (it = 1; it < maxit; ++it)`); modified Gram-Schmidt; Givens rotations;
// inner exit when |eta_{k+1}| < tol*||b||; true residual after every cycle;
// num_matvec counts every operator application; the preconditioned overload
// applies LEFT preconditioning (solves P A x = P b).
// The Hessenberg column is accumulated on the device and fetched once per
// Arnoldi step instead of one blocking copy per dot product.
#ifndef CUDDH_AMD_KRYLOV_HPP
#define CUDDH_AMD_KRYLOV_HPP

#include <chrono>
#include <iomanip>
#include <iostream>
#include <vector>

#include "blas1.hpp"
#include "operator.hpp"
#include "tensor.hpp"

namespace cuddh
{
    struct solver_out
    {
        bool success;
        int num_iter;
        int num_matvec;
        std::vector<double> res_norm;
        std::vector<double> time;
    };

    solver_out gmres(int n, double *x, const Operator *A, const double *b, const Operator *Precond, int m, int maxit,
                     double tol = 1e-6, int verbose = 0, double max_seconds = 6 * 60 * 60);
    solver_out gmres(int n, double *x, const Operator *A, const double *b, int m, int maxit, double tol = 1e-6,
                     int verbose = 0, double max_seconds = 6 * 60 * 60);
    solver_out gmres(int n, float *x, const SinglePrecisionOperator *A, const float *b, int m, int maxit,
                     float tol = 1e-4, int verbose = 0, double max_seconds = 6 * 60 * 60);

    /// Vectors partitioned over processes (one per GPU): each rank passes its part of x and b (or a copy that is
    /// zero outside the part it owns) and `fn` must sum `count` DEVICE scalars over all ranks in place, ordered on
    /// stream() (an RCCL all-reduce).  Same iteration; every inner product is reduced before it is used, so all
    /// ranks take identical decisions.  Not in the reference (single GPU).
    struct ScalarReduce
    {
        void (*fn)(void *user, void *d_scalars, int count, int is_f64);
        void *user;
    };
    solver_out gmres(int n, double *x, const Operator *A, const double *b, int m, int maxit, double tol, int verbose,
                     double max_seconds, const ScalarReduce &reduce);
    solver_out gmres(int n, float *x, const SinglePrecisionOperator *A, const float *b, int m, int maxit, float tol, int verbose,
                     double max_seconds, const ScalarReduce &reduce);
} // namespace cuddh

#endif
