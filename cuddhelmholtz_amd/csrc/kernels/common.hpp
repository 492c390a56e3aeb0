// Shared helpers of the HIP kernel layer (gfx950 only).
#ifndef CUDDH_KERNELS_COMMON_HPP
#define CUDDH_KERNELS_COMMON_HPP

#include <hip/hip_runtime.h>

#include "cuddh_hip.h"

namespace cuddh_k
{
    constexpr int WAVE = 64;

    inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

    /// error of the launch just issued (also clears the sticky launch error)
    inline int launch_status() { return static_cast<int>(hipGetLastError()); }

    /// grid size for a memory-bound grid-stride kernel: enough workgroups to fill
    /// 256 CUs several times over, bounded so tiny problems launch tiny grids
    inline int stream_grid(long long items, int block, int items_per_thread = 1)
    {
        const long long per_block = static_cast<long long>(block) * items_per_thread;
        long long g = (items + per_block - 1) / per_block;
        const long long cap = 256LL * 8;
        if (g > cap)
            g = cap;
        if (g < 1)
            g = 1;
        return static_cast<int>(g);
    }

    /// sum over the 64 lanes of a wavefront; every lane gets the result of lane 0's tree
    template <typename T>
    __device__ inline T wave_sum(T v)
    {
#pragma unroll
        for (int off = WAVE / 2; off > 0; off >>= 1)
            v += __shfl_down(v, off, WAVE);
        return v;
    }

    /// Jacobian [x_xi, y_xi, x_eta, y_eta] of the bilinear map of source/Element.cpp:5-36 at (s, e); c = (2, 4) corners,
    /// counter-clockwise.  One definition for every kernel that needs it, so that all of them round alike.
    __device__ inline void bilinear_jacobian(const double *__restrict__ c, double s, double e, double j4[4])
    {
#pragma unroll
        for (int a = 0; a < 2; ++a)
        {
            j4[a] = 0.25 * ((1 - e) * (c[2 + a] - c[0 + a]) + (1 + e) * (c[4 + a] - c[6 + a]));
            j4[2 + a] = 0.25 * ((1 - s) * (c[6 + a] - c[0 + a]) + (1 + s) * (c[4 + a] - c[2 + a]));
        }
    }

    /// hardware floating point atomic add (global_atomic_add_f64 / _f32), no CAS loop
    __device__ inline void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }
    __device__ inline void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
} // namespace cuddh_k

#endif
