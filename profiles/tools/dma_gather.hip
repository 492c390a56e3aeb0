// Feasibility of hiding a patch's x gather behind the previous patch's arithmetic ("next" item 1-(0)):
// persistent wavefronts, each walking a sequence of patches; per patch 640 dof indices -> 2 x 640 doubles of x ([u; v]) into
// LDS, then W dependent FMAs per lane ("slices"), then one store.
//   reg    : indices -> x values through registers -> ds_write, then the arithmetic            (what helm_lane_kernel does)
//   dma    : the NEXT patch's indices are requested before the arithmetic, its x values are gathered by
//            global_load_lds_dword (per-lane global address, LDS destination = base + 4 lane, no destination registers)
//            into the other half of a double-buffered LDS area while the arithmetic of the current patch runs
// build: hipcc -O3 --offload-arch=gfx950 dma_gather.hip -o dma_gather;  run: ./dma_gather [patches per wave] [W]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                      \
    do                                                                                                \
    {                                                                                                 \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess)                                                                         \
        {                                                                                             \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                                \
            std::exit(1);                                                                             \
        }                                                                                             \
    } while (0)

constexpr int ROWS = 10, NL = 64 * ROWS; // 640 local dofs per patch

__device__ inline double work(double a, int W)
{
    double acc = a;
    for (int i = 0; i < W; ++i)
        acc = acc * 1.0000001 + 1e-9; // dependent chain: the arithmetic of a wavefront between two gathers
    return acc;
}

__global__ void __launch_bounds__(64, 2) reg_kernel(const int *__restrict__ dofs, const double *__restrict__ x, int ndof, double *__restrict__ out,
                                                    int n_waves, int per_wave, int W)
{
    extern __shared__ double lds[]; // [2][NL]
    const int wave = blockIdx.x, lane = threadIdx.x;
    if (wave >= n_waves)
        return;
    for (int p = 0; p < per_wave; ++p)
    {
        const int *d = dofs + ((size_t)wave * per_wave + p) * NL;
        int gi[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            gi[j] = d[64 * j + lane];
        double xu[ROWS], xv[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
        {
            xu[j] = x[gi[j]];
            xv[j] = x[ndof + gi[j]];
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
        {
            lds[64 * j + lane] = xu[j];
            lds[NL + 64 * j + lane] = xv[j];
        }
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            s += lds[64 * j + (lane ^ 1)] - lds[NL + 64 * j + (lane ^ 3)];
        out[((size_t)wave * per_wave + p) * 64 + lane] = work(s, W);
        __syncthreads();
    }
}

// With the builtin the compiler knows that the instruction writes LDS and waits for it (vmcnt(0)) before the next LDS read it cannot
// prove disjoint -- which serialises the gather and the arithmetic again.  ASM = 1 issues it as inline assembly (M0 = LDS byte
// address of the 256-byte destination row); the kernel then waits for it explicitly at the top of the next patch.
#ifndef ASM
#define ASM 1
#endif
#ifndef ORDER
#define ORDER 1
#endif
__device__ inline void dma_dword(const void *g, unsigned lds_byte)
{
#if ASM
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds_byte) : "memory");
#else
    __builtin_amdgcn_global_load_lds(g, reinterpret_cast<__attribute__((address_space(3))) void *>(lds_byte), 4, 0, 0);
#endif
}

// LDS per buffer: 4 planes of NL dwords: u.lo, u.hi, v.lo, v.hi
__global__ void __launch_bounds__(64, 2) dma_kernel(const int *__restrict__ dofs, const double *__restrict__ x, int ndof, double *__restrict__ out,
                                                    int n_waves, int per_wave, int W)
{
    extern __shared__ unsigned ldsw[]; // [2 buffers][4 planes][NL]
    const int wave = blockIdx.x, lane = threadIdx.x;
    if (wave >= n_waves)
        return;
    const unsigned base = static_cast<unsigned>(reinterpret_cast<size_t>(ldsw));
    auto gather = [&](int p, int buf)
    {
        const int *d = dofs + ((size_t)wave * per_wave + p) * NL;
        int gi[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            gi[j] = d[64 * j + lane];
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
        {
            const char *pu = reinterpret_cast<const char *>(x + gi[j]), *pv = reinterpret_cast<const char *>(x + ndof + gi[j]);
            const unsigned dst = base + 4u * (unsigned)(buf * 4 * NL + 64 * j);
            dma_dword(pu, dst);
            dma_dword(pu + 4, dst + 4u * NL);
            dma_dword(pv, dst + 8u * NL);
            dma_dword(pv + 4, dst + 12u * NL);
        }
    };
    gather(0, 0);
    for (int p = 0; p < per_wave; ++p)
    {
        const int buf = p & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this patch's values are in LDS
        __syncthreads();
        const unsigned *b = ldsw + buf * 4 * NL;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
        {
            const int iu = 64 * j + (lane ^ 1), iv = 64 * j + (lane ^ 3);
            const double u = __hiloint2double((int)b[NL + iu], (int)b[iu]);
            const double v = __hiloint2double((int)b[3 * NL + iv], (int)b[2 * NL + iv]);
            s += u - v;
        }
        // LDS reads of this patch first (ORDER = 1): DS operations issued after an LDS-DMA of the same wavefront wait for it
#if ORDER
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s)::"memory");
#endif
        if (p + 1 < per_wave)
            gather(p + 1, buf ^ 1); // in flight during the arithmetic below
        out[((size_t)wave * per_wave + p) * 64 + lane] = work(s, W);
    }
}

int main(int argc, char **argv)
{
    const int per_wave = argc > 1 ? std::atoi(argv[1]) : 8;
    const int W = argc > 2 ? std::atoi(argv[2]) : 4000;
    const int n_waves = 2048, ndof = 9443329;
    const size_t n_patches = (size_t)n_waves * per_wave;
    std::vector<int> hd(n_patches * NL);
    unsigned long long s = 12345;
    for (size_t p = 0; p < n_patches; ++p)
    {
        // runs of 4 consecutive dofs starting at random places near a patch-dependent base (the element-wise numbering of the reference)
        const int basep = (int)((p * 577ull) % (ndof - 70000));
        for (int i = 0; i < NL; i += 4)
        {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const int start = basep + (int)((s >> 33) % 65000);
            for (int k = 0; k < 4; ++k)
                hd[p * NL + i + k] = start + k;
        }
    }
    std::vector<double> hx(2 * (size_t)ndof);
    for (size_t i = 0; i < hx.size(); ++i)
        hx[i] = 1e-3 * (double)(i % 1013) - 0.5;
    int *dofs;
    double *x, *out;
    CHECK(hipMalloc(&dofs, hd.size() * 4));
    CHECK(hipMalloc(&x, hx.size() * 8));
    CHECK(hipMalloc(&out, n_patches * 64 * 8));
    CHECK(hipMemcpy(dofs, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch)
    {
        for (int i = 0; i < 2; ++i)
            launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i)
            launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-44s %9.1f us per launch, %6.2f us per patch and wavefront\n", name, 1e3 * ms / reps, 1e3 * ms / reps / per_wave);
    };
    std::printf("%d wavefronts (8 per CU), %d patches each, %d dependent FMAs per patch\n", n_waves, per_wave, W);
    const size_t lds_reg = 2 * NL * 8 + 10000, lds_dma = 2 * 4 * NL * 4 + 0; // both leave 8 wavefronts per CU
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma));
    time("gather through registers, then arithmetic", [&] { hipLaunchKernelGGL(reg_kernel, dim3(n_waves), dim3(64), lds_reg, 0, dofs, x, ndof, out, n_waves, per_wave, W); });
    std::vector<double> ref(n_patches * 64), got(ref.size());
    CHECK(hipMemcpy(ref.data(), out, ref.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemset(out, 0, ref.size() * 8));
    time("next gather by LDS-DMA during the arithmetic", [&] { hipLaunchKernelGGL(dma_kernel, dim3(n_waves), dim3(64), lds_dma, 0, dofs, x, ndof, out, n_waves, per_wave, W); });
    CHECK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i)
        bad += got[i] != ref[i];
    std::printf("results %s (%zu of %zu differ)\n", bad ? "DIFFER" : "identical", bad, got.size());
    return 0;
}
