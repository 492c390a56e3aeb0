// Level-1 vector operations on DEVICE arrays.
// Contract: reference include/linalg.hpp:16-53.  `dot`, `norm`, `dist` return the
// value on the host (they synchronise); everything else is asynchronous on the
// library stream.  Unlike the reference (source/linalg.cpp:67-83) no call
// allocates: reductions use a persistent workspace and a fixed summation order,
// so results are bitwise reproducible run to run.
// Deviation: ones(int*) really writes 1 (the reference's cudaMemset(x, 1, ...)
// byte-fill writes 0x01010101, include/linalg.hpp:53).
#ifndef CUDDH_AMD_BLAS1_HPP
#define CUDDH_AMD_BLAS1_HPP

#include <cmath>

#include "launch.hpp"
#include "memory.hpp"

namespace cuddh
{
    /// y <- a x + b y
    void axpby(int n, double a, const double *x, double b, double *y);
    void axpby(int n, float a, const float *x, float b, float *y);

    double dot(int n, const double *x, const double *y);
    float dot(int n, const float *x, const float *y);

    double norm(int n, const double *x);
    float norm(int n, const float *x);

    /// ||x - y||
    double dist(int n, const double *x, const double *y);
    float dist(int n, const float *x, const float *y);

    void copy(int n, const double *x, double *y);
    void copy(int n, const float *x, float *y);
    void copy(int n, const int *x, int *y);

    void scal(int n, double a, double *x);
    void scal(int n, float a, float *x);

    void fill(int n, double a, double *x);
    void fill(int n, float a, float *x);
    void fill(int n, int a, int *x);

    void zeros(int n, double *x);
    void zeros(int n, float *x);
    void zeros(int n, int *x);

    inline void ones(int n, double *x) { fill(n, 1.0, x); }
    inline void ones(int n, float *x) { fill(n, 1.0f, x); }
    inline void ones(int n, int *x) { fill(n, 1, x); }
} // namespace cuddh

#endif
