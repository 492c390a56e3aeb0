// Do fp32 matrix-core instructions (v_mfma_f32_16x16x4_f32) and fp32 vector instructions (v_fmac_f32) of DIFFERENT wavefronts on
// the same SIMD execute at the same time on gfx950?  (ddh_mfma_kernel: 8 MFMA + ~72 VALU per RK2 step and wavefront.)
// 512-thread workgroups, one per CU: wavefronts w and w + 4 share a SIMD.  Instruction counts are chosen so that a wavefront
// of either kind alone needs the same time T; then
//   all eight MFMA  -> 2 T    all eight VALU -> 2 T
//   four MFMA + four VALU (split by wave >= 4, or by parity) -> T if the pipes co-execute, 2 T if they exclude each other.
// A fourth form interleaves the two kinds in ONE wavefront (independent operands): T if a wavefront's own VALU work can issue
// under its own MFMA.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_coexec.hip -o mfma_valu_coexec
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

#define REP4(x) x x x x
#define REP8(x) x x x x x x x x
#define VGROUP                                                                                                             \
    "v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t"                  \
    "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9\n\t"

// one "unit" of MFMA work = 4 independent v_mfma_f32_16x16x4_f32 (4 x 32 cycles); one unit of VALU work = 64 v_fmac_f32 (64 x 2)
__device__ __forceinline__ void mfma_unit(f4 &c0, f4 &c1, f4 &c2, f4 &c3, float a, float b)
{
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// the same unit on the XDL matrix pipe: 8 independent v_mfma_f32_16x16x32_bf16 (8 x 16 cycles if the instruction takes 4 passes)
__device__ __forceinline__ void xdl_unit(f4 &c0, f4 &c1, f4 &c2, f4 &c3, bf8 a, bf8 b)
{
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
}

// MODE 6: every wave bf16 MFMA; 7: waves 0-3 bf16 MFMA, waves 4-7 v_fmac_f32; 8: both interleaved in every wave (half each);
// 9: bf16 MFMA interleaved in-wave with the FULL count of v_fmac_f32 in every wave (2T of VALU per SIMD + T of MFMA: 2T if hidden)
template <int MODE>
__global__ void __launch_bounds__(512) xdl_kernel(float *out, int units)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float a = 1.0f + 1e-7f * threadIdx.x, c = 1e-9f;
    bf8 ba, bb;
    for (int i = 0; i < 8; ++i)
    {
        ba[i] = (__bf16)(1.0f + 0.01f * (threadIdx.x & 7));
        bb[i] = (__bf16)(1e-3f * i);
    }
    if (MODE == 8 || MODE == 9)
    {
        const int n = MODE == 8 ? units / 2 : units;
        for (int u = 0; u < n; ++u)
        {
            if (MODE == 8 || (u & 1))
                xdl_unit(c0, c1, c2, c3, ba, bb);
            asm volatile(REP8(VGROUP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
        }
    }
    else if (MODE == 7 && wave >= 4)
    {
        for (int u = 0; u < units; ++u)
            asm volatile(REP8(VGROUP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
    }
    else
    {
        for (int u = 0; u < units; ++u)
            xdl_unit(c0, c1, c2, c3, ba, bb);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

#define VGROUP2                                                                                                            \
    "v_fmac_f32 %0, %9, %8\n\tv_fmac_f32 %1, %9, %8\n\tv_fmac_f32 %2, %9, %8\n\tv_fmac_f32 %3, %9, %8\n\t"                  \
    "v_fmac_f32 %4, %9, %8\n\tv_fmac_f32 %5, %9, %8\n\tv_fmac_f32 %6, %9, %8\n\tv_fmac_f32 %7, %9, %8\n\t"
// controls and fine-grained in-wave interleaving.  MODE 10: waves 0-3 and waves 4-7 run two COPIES of the v_fmac_f32 loop (different
// code addresses, same pipe: isolates instruction-fetch effects of two programs per SIMD); 11: every wave, per unit 8 x (1 bf16 MFMA +
// 8 v_fmac_f32) -- the full VALU count with the matrix work spread through it; 12: per unit 4 x (1 fp32 MFMA + 16 v_fmac_f32)
template <int MODE>
__global__ void __launch_bounds__(512) fine_kernel(float *out, int units)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float a = 1.0f + 1e-7f * threadIdx.x, c = 1e-9f, b = 1e-6f;
    bf8 ba, bb;
    for (int i = 0; i < 8; ++i)
    {
        ba[i] = (__bf16)(1.0f + 0.01f * (threadIdx.x & 7));
        bb[i] = (__bf16)(1e-3f * i);
    }
    if (MODE == 10)
    {
        if (wave >= 4)
            for (int u = 0; u < units; ++u)
                asm volatile(REP8(VGROUP2) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
        else
            for (int u = 0; u < units; ++u)
                asm volatile(REP8(VGROUP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
    }
    else if (MODE == 11)
    {
        for (int u = 0; u < units; ++u)
        {
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                f4 &cc = (j & 3) == 0 ? c0 : (j & 3) == 1 ? c1 : (j & 3) == 2 ? c2 : c3;
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, cc, 0, 0, 0);
                asm volatile(VGROUP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
            }
        }
    }
    else
    {
        for (int u = 0; u < units; ++u)
        {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                f4 &cc = j == 0 ? c0 : j == 1 ? c1 : j == 2 ? c2 : c3;
                cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, cc, 0, 0, 0);
                asm volatile(VGROUP VGROUP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// MODE 0: every wave MFMA; 1: every wave VALU; 2: waves >= 4 VALU, others MFMA; 3: odd waves VALU; 4: both kinds interleaved in every
// wave (half the units of each, so that a wave alone needs T); 5: like 2 but the VALU waves run DPP-modified fmacs (half rate)
template <int MODE>
__global__ void __launch_bounds__(512) coexec_kernel(float *out, int units)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform role: a scalar branch, not an exec mask
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float a = 1.0f + 1e-7f * threadIdx.x, b = 1e-6f, c = 1e-9f;
    bool valu_role = false;
    if (MODE == 1)
        valu_role = true;
    if (MODE == 2 || MODE == 5)
        valu_role = wave >= 4;
    if (MODE == 3)
        valu_role = wave & 1;
    if (MODE == 4)
    {
        for (int u = 0; u < units / 2; ++u)
        {
            mfma_unit(c0, c1, c2, c3, a, b);
            asm volatile(REP8(VGROUP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
        }
    }
    else if (valu_role)
    {
        if (MODE == 5)
            for (int u = 0; u < units / 2; ++u)
                asm volatile(REP8("v_fmac_f32_dpp %0, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %1, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %2, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %3, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %4, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %5, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %6, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                                  "v_fmac_f32_dpp %7, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
        else
            for (int u = 0; u < units; ++u)
                asm volatile(REP8(VGROUP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a), "v"(c));
    }
    else
    {
        for (int u = 0; u < units; ++u)
            mfma_unit(c0, c1, c2, c3, a, b);
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0.x + c1.y + c2.z + c3.w + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename K>
float run(const char *name, K kernel, int units)
{
    const int blocks = 256;
    float *out;
    (void)hipMalloc(&out, blocks * 512 * sizeof(float));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), 0, 0, out, 8);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), 0, 0, out, units);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    std::printf("%-72s %8.3f ms\n", name, best);
    (void)hipFree(out);
    return best;
}

int main()
{
    const int units = 20000; // per wave: 80,000 MFMA (x 32 cycles) or 1,280,000 v_fmac_f32 (x 2 cycles) = 2.56e6 cycles = T ~ 1.07 ms at 2.4 GHz
    std::printf("gfx950, 256 workgroups x 8 wavefronts (2 per SIMD); T = one wavefront's work alone, 2T = two such wavefronts serialised\n");
    const float m = run("all eight wavefronts v_mfma_f32_16x16x4_f32          (expect 2T)", coexec_kernel<0>, units);
    const float v = run("all eight wavefronts v_fmac_f32                      (expect 2T)", coexec_kernel<1>, units);
    const float x = run("waves 0-3 MFMA, waves 4-7 v_fmac_f32                 (T if co-executing)", coexec_kernel<2>, units);
    const float y = run("even waves MFMA, odd waves v_fmac_f32                (T if co-executing)", coexec_kernel<3>, units);
    const float s = run("every wave: MFMA and v_fmac_f32 interleaved, half each (T if overlapped in-wave)", coexec_kernel<4>, units);
    const float d = run("waves 0-3 MFMA, waves 4-7 v_fmac_f32_dpp (half count) (T if co-executing)", coexec_kernel<5>, units);
    std::printf("co-execution factor, split by wave >= 4: %.2f  by parity: %.2f  in-wave: %.2f  with DPP: %.2f   (1.0 = fully exclusive, 2.0 = fully overlapped)\n",
                0.5f * (m + v) / x, 0.5f * (m + v) / y, 0.5f * (m + v) / s, 0.5f * (m + v) / d);
    std::printf("-- the same with the XDL matrix pipe: v_mfma_f32_16x16x32_bf16, 8 per unit --\n");
    const float xm = run("all eight wavefronts v_mfma_f32_16x16x32_bf16", xdl_kernel<6>, units);
    const float xx = run("waves 0-3 bf16 MFMA, waves 4-7 v_fmac_f32", xdl_kernel<7>, units);
    const float xs = run("every wave: bf16 MFMA and v_fmac_f32 interleaved, half each", xdl_kernel<8>, units);
    const float xf = run("every wave: full v_fmac_f32 count + bf16 MFMA every second unit", xdl_kernel<9>, units);
    std::printf("bf16: all-MFMA %.3f ms (%.1f cycles per instruction and SIMD at 2.4 GHz); split by wave: %.3f ms (max(T_m, T_v) = %.3f if co-executing, sum = %.3f if exclusive);\n"
                "      in-wave half each: %.3f ms (exclusive: %.3f); in-wave full VALU + half MFMA: %.3f ms (VALU alone: %.3f)\n",
                xm, xm * 1e-3 * 2.4e9 / (2.0 * units * 8), xx, 0.5f * (xm > v ? xm : v), 0.5f * (xm + v), xs, 0.5f * (xm + v), xf, v);
    std::printf("-- controls and fine-grained in-wave interleaving (full v_fmac_f32 count in every wave: %.3f ms if the matrix work hides) --\n", v);
    run("control: waves 0-3 / 4-7 run two copies of the v_fmac_f32 loop", fine_kernel<10>, units);
    run("every wave: 8 x (1 bf16 MFMA + 8 v_fmac_f32) per unit", fine_kernel<11>, units);
    run("every wave: 4 x (1 fp32 MFMA + 16 v_fmac_f32) per unit", fine_kernel<12>, units);
    return 0;
}
