// Issue-rate / latency microbenchmark for gfx950, fp64: v_fma_f64 (independent and one dependent chain), v_mfma_f64_4x4x4
// and v_mfma_f64_16x16x4 (independent accumulators and one dependent chain), at 1 - 4 waves per SIMD.  Wall-clock (s_memrealtime,
// 100 MHz) per instruction converted to nominal 2.4 GHz cycles.  build: hipcc -O3 --offload-arch=gfx950 f64_rates.hip -o f64_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, int iters, unsigned long long *ticks)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12;
    d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (MODE == 0) // 8 independent fma chains
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                             "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        else if constexpr (MODE == 1) // one dependent fma chain
        {
#pragma unroll
            for (int r = 0; r < 64; ++r)
                asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(a0) : "v"(b));
        }
        else if constexpr (MODE == 2) // 8 independent 4x4x4 (64 per iteration)
        {
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a7, 0, 0, 0);
            }
        }
        else if constexpr (MODE == 3) // one dependent 4x4x4 chain
        {
#pragma unroll
            for (int r = 0; r < 64; ++r)
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a0, 0, 0, 0);
        }
        else if constexpr (MODE == 4) // 4 independent 16x16x4 (64 per iteration)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, m1, 0, 0, 0);
                m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, m2, 0, 0, 0);
                m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, m3, 0, 0, 0);
            }
        }
        else if constexpr (MODE == 5) // one dependent 16x16x4 chain
        {
#pragma unroll
            for (int r = 0; r < 64; ++r)
                m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, m0, 0, 0, 0);
        }
        else if constexpr (MODE == 6) // the slice pattern: mfma 4x4x4 -> fma on its result -> mfma on that (dependent chain of 3 kinds)
        {
#pragma unroll
            for (int r = 0; r < 16; ++r)
            {
                a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a1, 0.0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, a1, a0, 0, 0, 0);
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a2) : "v"(a1), "v"(b));
                asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a2) : "v"(a0), "v"(c));
                a1 = a2;
            }
        }
    }
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3];
    if (threadIdx.x == 0)
        ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter)
{
    const int iters = 2000;
    double *out;
    unsigned long long *ticks;
    hipMalloc(&out, 4096 * 64 * sizeof(double));
    hipMalloc(&ticks, 4096 * sizeof(unsigned long long));
    std::printf("%-46s", name);
    for (int wps : {1, 2, 3, 4})
    {
        const int blocks = 256 * 4 * wps; // one workgroup of one wavefront each; the dispatcher spreads them over CUs and SIMDs
        k<MODE><<<blocks, 64>>>(out, 10, ticks);
        hipDeviceSynchronize();
        k<MODE><<<blocks, 64>>>(out, iters, ticks);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto t : h)
            mean += t;
        mean /= blocks;
        const double cyc = mean * 24.0 / ((double)iters * per_iter); // 100 MHz ticks -> 2.4 GHz cycles, per instruction of ONE wave
        std::printf("  %dw: %6.1f (per SIMD %5.1f)", wps, cyc, cyc / wps);
    }
    std::printf("\n");
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    std::printf("nominal 2.4 GHz cycles per instruction as seen by one wavefront (and divided by the wavefronts sharing the SIMD)\n");
    run<0>("v_fmac_f64, 8 independent chains", 64);
    run<1>("v_fmac_f64, one dependent chain", 64);
    run<2>("v_mfma_f64_4x4x4, 8 independent accumulators", 64);
    run<3>("v_mfma_f64_4x4x4, one dependent chain", 64);
    run<4>("v_mfma_f64_16x16x4, 4 independent accumulators", 64);
    run<5>("v_mfma_f64_16x16x4, one dependent chain", 64);
    run<6>("mfma4 -> mfma4 -> mul -> fmac dependent (per 4)", 16);
    return 0;
}
