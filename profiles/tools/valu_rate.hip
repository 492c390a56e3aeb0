// Issue-rate microbenchmark for gfx950: cycles per wave64 instruction of plain v_fmac_f32, v_fmac_f32 with a DPP
// quad_perm / row_shr / row_half_mirror operand, v_mov_b32_dpp and v_pk_fma_f32, at 1, 2, 3, 4 and 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INSTR)                                                                                          \
    for (int it = 0; it < iters; ++it)                                                                       \
    {                                                                                                        \
        asm volatile(REP8(REP8(INSTR)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    }

// eight independent accumulators, 64 instructions per iteration
#define ONE(I) I
#define GROUP(OP, TAIL)                                                                                      \
    OP " %0, %8, %9" TAIL "\n\t" OP " %1, %8, %9" TAIL "\n\t" OP " %2, %8, %9" TAIL "\n\t" OP " %3, %8, %9" TAIL "\n\t" \
    OP " %4, %8, %9" TAIL "\n\t" OP " %5, %8, %9" TAIL "\n\t" OP " %6, %8, %9" TAIL "\n\t" OP " %7, %8, %9" TAIL "\n\t"

template <int MODE>
__global__ void __launch_bounds__(64) rate_kernel(float *out, int iters, long long *cycles)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0f + 1e-7f * threadIdx.x, c = 1e-9f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (MODE == 0)
            asm volatile(REP8(GROUP("v_fmac_f32", "")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if constexpr (MODE == 1)
            asm volatile(REP8(GROUP("v_fmac_f32_dpp", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if constexpr (MODE == 2)
            asm volatile(REP8(GROUP("v_fmac_f32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if constexpr (MODE == 3)
            asm volatile(REP8(GROUP("v_fmac_f32_dpp", " row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if constexpr (MODE == 4) // v_mov_b32_dpp dst, src
            asm volatile(REP8("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if constexpr (MODE == 5) // scalar coefficient: v_fmac_f32 dst, s, v
            asm volatile(REP8(GROUP("v_fmac_f32", "")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(iters * 1e-9f), "v"(c));
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

// packed: 4 accumulator pairs
__global__ void __launch_bounds__(64) pk_kernel(float *out, int iters, long long *cycles)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {1.f * threadIdx.x, 1}, a1 = {2, 3}, a2 = {4, 5}, a3 = {6, 7}, a4 = {1, 1}, a5 = {2, 2}, a6 = {3, 3}, a7 = {4, 4};
    f2 b = {1.0f, 1.0f + 1e-7f}, c = {1e-9f, 2e-9f};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
        asm volatile(REP8("v_pk_fma_f32 %0, %8, %9, %0\n\tv_pk_fma_f32 %1, %8, %9, %1\n\tv_pk_fma_f32 %2, %8, %9, %2\n\tv_pk_fma_f32 %3, %8, %9, %3\n\t"
                          "v_pk_fma_f32 %4, %8, %9, %4\n\tv_pk_fma_f32 %5, %8, %9, %5\n\tv_pk_fma_f32 %6, %8, %9, %6\n\tv_pk_fma_f32 %7, %8, %9, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

// fp64: v_fma_f64 and v_mfma_f64_16x16x4_f64 (eight independent accumulators / four independent 4-register accumulators)
__global__ void __launch_bounds__(64) fma64_kernel(float *out, int iters, long long *cycles)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
        asm volatile(REP8("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                          "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) mfma64_kernel(float *out, int iters, long long *cycles)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-6;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 16; ++u) // 64 MFMAs per iteration, like the 64 VALU instructions of the other kernels
        {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(c0.x + c1.y + c2.z + c3.w);
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

__global__ void __launch_bounds__(64) mfma64_4x4_kernel(float *out, int iters, long long *cycles)
{
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-6;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int u = 0; u < 8; ++u) // 64 MFMAs per iteration
        {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = static_cast<float>(c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7);
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char *name, K kernel, int waves_per_simd)
{
    const int n_cu = 256, blocks = n_cu * 4 * waves_per_simd, iters = 2000;
    float *out;
    long long *cyc;
    hipMalloc(&out, blocks * 64 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, out, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h)
        mean += v;
    mean /= blocks;
    const double instr = 64.0 * iters;
    // clock64 ticks at 100 MHz on this part: report wall-derived cycles at the nominal 2.4 GHz as well
    std::printf("%-28s waves/SIMD %d: %8.3f ms, %.2f cycles(2.4GHz)/instr/wave -> %.2f cycles/instr per SIMD\n", name, waves_per_simd, ms,
                ms * 1e-3 * 2.4e9 / instr, ms * 1e-3 * 2.4e9 / instr / waves_per_simd);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 3, 4, 8})
    {
        run("v_fmac_f32", rate_kernel<0>, w);
        run("v_fmac_f32 (sgpr coef)", rate_kernel<5>, w);
        run("v_fmac_f32_dpp quad_perm", rate_kernel<1>, w);
        run("v_fmac_f32_dpp row_shr", rate_kernel<2>, w);
        run("v_fmac_f32_dpp half_mirror", rate_kernel<3>, w);
        run("v_mov_b32_dpp quad_perm", rate_kernel<4>, w);
        run("v_pk_fma_f32", pk_kernel, w);
        run("v_fma_f64", fma64_kernel, w);
        run("v_mfma_f64_16x16x4_f64", mfma64_kernel, w);
        run("v_mfma_f64_4x4x4_4b_f64", mfma64_4x4_kernel, w);
    }
    return 0;
}
