// Level-1 operations: thin host wrappers over the C-ABI kernels.
#include "cuddh/blas1.hpp"

#include <cmath>

#include "cuddh_hip.h"

namespace cuddh
{
    namespace
    {
        // persistent reduction workspace + one device scalar, created on first use -- one per (host thread, device): the launch
        // stream is per thread (runtime.cpp), so two threads reducing on their own streams or devices must not share partial sums
        struct ReduceScratch
        {
            void *ws = nullptr;
            void *scalar = nullptr;

            void ensure()
            {
                if (ws)
                    return;
                detail::check_hip(cuddh_hip_malloc_zeroed(&ws, cuddh_hip_reduce_ws_bytes()), "reduction workspace");
                detail::check_hip(cuddh_hip_malloc_zeroed(&scalar, 16), "reduction result");
            }
        };

        ReduceScratch &scratch()
        {
            constexpr int max_devices = 64;
            thread_local ReduceScratch per_device[max_devices]; // intentionally never freed: outlives every stream
            int dev = cuddh_hip_current_device();
            if (dev < 0 || dev >= max_devices)
                dev = 0;
            ReduceScratch &s = per_device[dev];
            s.ensure();
            return s;
        }

        template <typename T>
        T fetch(const void *dev_scalar)
        {
            detail::check_hip(cuddh_hip_stream_sync(stream()), "stream sync");
            T v;
            detail::check_hip(cuddh_hip_copy_d2h_on(&v, dev_scalar, sizeof(T), stream()), "scalar copy");
            return v;
        }
    } // namespace

    void axpby(int n, double a, const double *x, double b, double *y)
    {
        detail::check_hip(cuddh_hip_axpby_f64(n, a, x, b, y, stream()), "axpby");
    }

    void axpby(int n, float a, const float *x, float b, float *y)
    {
        detail::check_hip(cuddh_hip_axpby_f32(n, a, x, b, y, stream()), "axpby");
    }

    double dot(int n, const double *x, const double *y)
    {
        auto &s = scratch();
        detail::check_hip(cuddh_hip_dot_f64(n, x, y, static_cast<double *>(s.scalar), s.ws, stream()), "dot");
        return fetch<double>(s.scalar);
    }

    float dot(int n, const float *x, const float *y)
    {
        auto &s = scratch();
        detail::check_hip(cuddh_hip_dot_f32(n, x, y, static_cast<float *>(s.scalar), s.ws, stream()), "dot");
        return fetch<float>(s.scalar);
    }

    double norm(int n, const double *x) { return std::sqrt(dot(n, x, x)); }
    float norm(int n, const float *x) { return std::sqrt(dot(n, x, x)); }

    double dist(int n, const double *x, const double *y)
    {
        auto &s = scratch();
        detail::check_hip(cuddh_hip_sqdist_f64(n, x, y, static_cast<double *>(s.scalar), s.ws, stream()), "dist");
        return std::sqrt(fetch<double>(s.scalar));
    }

    float dist(int n, const float *x, const float *y)
    {
        auto &s = scratch();
        detail::check_hip(cuddh_hip_sqdist_f32(n, x, y, static_cast<float *>(s.scalar), s.ws, stream()), "dist");
        return std::sqrt(fetch<float>(s.scalar));
    }

    void copy(int n, const double *x, double *y) { detail::check_hip(cuddh_hip_copy_f64(n, x, y, stream()), "copy"); }
    void copy(int n, const float *x, float *y) { detail::check_hip(cuddh_hip_copy_f32(n, x, y, stream()), "copy"); }
    void copy(int n, const int *x, int *y) { detail::check_hip(cuddh_hip_copy_i32(n, x, y, stream()), "copy"); }

    void scal(int n, double a, double *x) { detail::check_hip(cuddh_hip_scal_f64(n, a, x, stream()), "scal"); }
    void scal(int n, float a, float *x) { detail::check_hip(cuddh_hip_scal_f32(n, a, x, stream()), "scal"); }

    void fill(int n, double a, double *x) { detail::check_hip(cuddh_hip_fill_f64(n, a, x, stream()), "fill"); }
    void fill(int n, float a, float *x) { detail::check_hip(cuddh_hip_fill_f32(n, a, x, stream()), "fill"); }
    void fill(int n, int a, int *x) { detail::check_hip(cuddh_hip_fill_i32(n, a, x, stream()), "fill"); }

    void zeros(int n, double *x)
    {
        detail::check_hip(cuddh_hip_memset_zero(x, static_cast<std::size_t>(n) * sizeof(double), stream()), "zeros");
    }
    void zeros(int n, float *x)
    {
        detail::check_hip(cuddh_hip_memset_zero(x, static_cast<std::size_t>(n) * sizeof(float), stream()), "zeros");
    }
    void zeros(int n, int *x)
    {
        detail::check_hip(cuddh_hip_memset_zero(x, static_cast<std::size_t>(n) * sizeof(int), stream()), "zeros");
    }
} // namespace cuddh
