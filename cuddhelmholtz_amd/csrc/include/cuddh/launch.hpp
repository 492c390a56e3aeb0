// Kernel launchers for user-supplied device lambdas.
// Contract (reference include/forall.hpp:21-64): `forall(n, f)` calls f(k) for
// k in [0,n); `forall_{1,2,3}d(bx[,by[,bz]], n, f)` runs one workgroup of the
// given shape per item and calls f(item) from every thread of it (the body reads
// threadIdx itself).  Launches are asynchronous on the library stream.
#ifndef CUDDH_AMD_LAUNCH_HPP
#define CUDDH_AMD_LAUNCH_HPP

#include <hip/hip_runtime.h>

#ifndef CUDDH_FORALL_BLOCK_SIZE
#define CUDDH_FORALL_BLOCK_SIZE 256
#endif

namespace cuddh
{
    /// stream every library launch goes to (null stream unless set_stream was called)
    hipStream_t stream();
    void set_stream(hipStream_t s);

    namespace detail
    {
        template <typename Body>
        __global__ void flat_range_kernel(int n, Body body)
        {
            // grid-stride so that the grid can stay bounded for very long ranges
            for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
                body(k);
        }

        template <typename Body>
        __global__ void per_item_kernel(int n, Body body)
        {
            for (int item = blockIdx.x; item < n; item += gridDim.x)
            {
                body(item);
                __syncthreads(); // bodies may keep state in __shared__ arrays
            }
        }

        inline int bounded_grid(long long want)
        {
            // 2^20 workgroups is far above what fills 256 CUs and far below the grid limit
            const long long cap = 1 << 20;
            return static_cast<int>(want < cap ? want : cap);
        }
    } // namespace detail

    template <typename LAMBDA>
    inline void forall(int n, LAMBDA &&fun)
    {
        if (n <= 0)
            return;
        const long long blocks = (static_cast<long long>(n) + CUDDH_FORALL_BLOCK_SIZE - 1) / CUDDH_FORALL_BLOCK_SIZE;
        hipLaunchKernelGGL(detail::flat_range_kernel, dim3(detail::bounded_grid(blocks)), dim3(CUDDH_FORALL_BLOCK_SIZE), 0,
                           stream(), n, fun);
    }

    template <typename LAMBDA>
    inline void forall_1d(int bx, int n, LAMBDA &&fun)
    {
        if (n <= 0)
            return;
        hipLaunchKernelGGL(detail::per_item_kernel, dim3(detail::bounded_grid(n)), dim3(bx), 0, stream(), n, fun);
    }

    template <typename LAMBDA>
    inline void forall_2d(int bx, int by, int n, LAMBDA &&fun)
    {
        if (n <= 0)
            return;
        hipLaunchKernelGGL(detail::per_item_kernel, dim3(detail::bounded_grid(n)), dim3(bx, by), 0, stream(), n, fun);
    }

    template <typename LAMBDA>
    inline void forall_3d(int bx, int by, int bz, int n, LAMBDA &&fun)
    {
        if (n <= 0)
            return;
        hipLaunchKernelGGL(detail::per_item_kernel, dim3(detail::bounded_grid(n)), dim3(bx, by, bz), 0, stream(), n, fun);
    }
} // namespace cuddh

#endif
