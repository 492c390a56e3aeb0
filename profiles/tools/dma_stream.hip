// Feasibility microbenchmark for the LDS-DMA metric ring of the fused apply (round 2).
// Streams n_patches blocks of ROWS x 64 doubles (the per-patch metric block of the plan, 512 B per row) and does W fp64
// FMAs per row and lane on them, three ways:
//   reg : the shape of helm_lane_kernel -- slices of 15 rows loaded into registers, consumed, next slice (dependent
//         chain of round trips), 2 wavefronts per SIMD;
//   dma : one wavefront per workgroup, the block copied by global_load_lds_dwordx4 into an LDS ring of RS KiB that runs
//         ahead of the consumer (no VGPR cost), rows read back with ds_read_b64; workgroups per CU limited by LDS;
//   copy: plain streaming read of the same bytes (16 B per lane), the box's reference rate.
// hipcc -O3 --offload-arch=gfx950 dma_stream.hip -o dma_stream && ./dma_stream [n_patches] [rows] [ring KiB] [extra LDS bytes]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                     \
    do                                                                               \
    {                                                                                \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess)                                                        \
        {                                                                            \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

constexpr int W = 24; // FMAs per row and lane (3300 per patch of 139 rows in the real kernel)

// W FMAs on four independent chains (the real kernel's slices have plenty of instruction-level parallelism)
__device__ inline void work(double (&acc)[4], double g)
{
#pragma unroll
    for (int i = 0; i < W; ++i)
        acc[i & 3] = __builtin_fma(acc[i & 3], 0.999, g);
}

__global__ void __launch_bounds__(64, 2) reg_kernel(const double *__restrict__ M, double *__restrict__ out, int n_patches, int rows, int chunk)
{
    const int patch = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (patch >= n_patches)
        return;
    const double *p = M + (size_t)patch * rows * 64 + threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = 0; r0 < rows; r0 += 15)
    {
        double g[15];
#pragma unroll
        for (int r = 0; r < 15; ++r)
            g[r] = r0 + r < rows ? __builtin_nontemporal_load(&p[(size_t)(r0 + r) * 64]) : 0.0;
#pragma unroll
        for (int r = 0; r < 15; ++r)
            if (r0 + r < rows)
                work(acc, g[r]);
    }
    out[(size_t)patch * 64 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// the slice chain with 16-byte loads: layout [slice][8 pairs][64 lanes][2] instead of [slice][15 rows][64 lanes]; a slice is 8 load
// instructions of 1 KiB (16 values per lane) instead of 15 of 512 B
typedef double dbl2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64, 2) reg16_kernel(const double *__restrict__ M, double *__restrict__ out, int n_patches, int rows, int chunk)
{
    const int patch = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (patch >= n_patches)
        return;
    const dbl2v *p = reinterpret_cast<const dbl2v *>(M + (size_t)patch * rows * 64) + threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r0 = 0; r0 < rows; r0 += 16)
    {
        dbl2v g[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            g[r] = (r0 + 2 * r < rows) ? __builtin_nontemporal_load(&p[(size_t)(r0 / 2 + r) * 64]) : dbl2v{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r0 + 2 * r < rows)
            {
                work(acc, g[r].x);
                work(acc, g[r].y);
            }
    }
    out[(size_t)patch * 64 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// whole block requested up front (ROWS x 2 registers per lane, the register file of a SIMD that holds ONE wavefront has
// 512 per lane: 256 VGPR + 256 AGPR), consumed in order: the loads run as far ahead as the memory system allows
template <int ROWS>
__global__ void __launch_bounds__(64, 1) regall_kernel(const double *__restrict__ M, double *__restrict__ out, int n_patches, int chunk)
{
    const int patch = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (patch >= n_patches)
        return;
    const double *p = M + (size_t)patch * ROWS * 64 + threadIdx.x;
    double g[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        g[r] = __builtin_nontemporal_load(&p[(size_t)r * 64]);
    __builtin_amdgcn_sched_barrier(0); // every load is issued before the first value is consumed
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
        work(acc, g[r]);
    out[(size_t)patch * 64 + threadIdx.x] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int N>
__device__ inline void wait_vm_imm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wave-uniform n: wait until at most n vector-memory operations are outstanding
__device__ inline void wait_vm(int n)
{
    switch (n)
    {
#define C(k)              \
    case k:               \
        wait_vm_imm<k>(); \
        break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20) C(21) C(22) C(23)
        C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36) C(37) C(38) C(39) C(40) C(41) C(42) C(43) C(44) C(45)
        C(46) C(47) C(48) C(49) C(50) C(51) C(52) C(53) C(54) C(55) C(56) C(57) C(58) C(59) C(60) C(61) C(62)
#undef C
    default:
        break; // more than 62 may stay outstanding: nothing to wait for (the counter saturates at 63)
    }
}

// one LDS-DMA piece: 64 lanes x 16 B = 1 KiB from gsrc (+ 16 B per lane) to the wave-uniform LDS byte address lds_dst
__device__ inline void dma_piece(const char *gsrc_lane, unsigned lds_dst)
{
    __builtin_amdgcn_global_load_lds(gsrc_lane, reinterpret_cast<__attribute__((address_space(3))) void *>(lds_dst), 16, 0, 0);
}

// rows: rows of the block (even); RS: ring slots of 1 KiB (2 rows each); STEP: rows consumed per step
__global__ void __launch_bounds__(64, 1) dma_kernel(const double *__restrict__ M, double *__restrict__ out, int n_patches, int rows, int RS, int chunk,
                                                    int STEP)
{
    extern __shared__ double lds[];
    const int patch = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (patch >= n_patches)
        return;
    const int lane = threadIdx.x;
    const char *src = reinterpret_cast<const char *>(M + (size_t)patch * rows * 64) + lane * 16;
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<size_t>(lds)); // LDS byte address of the ring
    const int T = rows / 2; // pieces
    int issued = 0;
    auto issue_until = [&](int k)
    {
        k = k < T ? k : T;
        for (; issued < k; ++issued)
            dma_piece(src + (size_t)issued * 1024, lds_base + (unsigned)(issued % RS) * 1024u);
    };
    issue_until(RS);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    volatile double *ring = lds;
    for (int r0 = 0; r0 < rows; r0 += 15)
    {
        const int r1 = r0 + 15 < rows ? r0 + 15 : rows;
        const int last_piece = (r1 - 1) / 2;
        wait_vm(issued - last_piece - 1);
        asm volatile("" ::: "memory");
        double g[15];
#pragma unroll
        for (int r = 0; r < 15; ++r)
            g[r] = r0 + r < rows ? ring[((r0 + r) % (2 * RS)) * 64 + lane] : 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the reads are done before their slots are refilled
        if (STEP == 0)
            issue_until(r1 / 2 + RS); // refill before the slice's arithmetic
#pragma unroll
        for (int r = 0; r < 15; ++r)
            if (r0 + r < rows)
                work(acc, g[r]);
        if (STEP != 0)
            issue_until(r1 / 2 + RS);
    }
    out[(size_t)patch * 64 + lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

typedef double dbl2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) copy_kernel(const dbl2 *__restrict__ M, double *__restrict__ out, size_t n2)
{
    double acc = 0.0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
    {
        const dbl2 v = __builtin_nontemporal_load(&M[i]);
        acc += v.x + v.y;
    }
    if (acc == 12345.678)
        out[0] = acc;
}

int main(int argc, char **argv)
{
    const int n_patches = argc > 1 ? std::atoi(argv[1]) : 16384;
    const int rows = argc > 2 ? std::atoi(argv[2]) : 140;
    const int RS = argc > 3 ? std::atoi(argv[3]) : 30;
    const int extra = argc > 4 ? std::atoi(argv[4]) : 10000; // the xy arrays of the real kernel
    const size_t n = (size_t)n_patches * rows * 64;
    double *M, *out;
    CHECK(hipMalloc(&M, n * 8));
    CHECK(hipMalloc(&out, (size_t)n_patches * 64 * 8));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = 1e-3 * (double)(i % 977);
    for (size_t o = 0; o < n; o += h.size())
        CHECK(hipMemcpy(M + o, h.data(), std::min(h.size(), n - o) * 8, hipMemcpyHostToDevice));
    const int chunk = (n_patches + 7) / 8;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double gb = n * 8.0 / 1e9;
    auto time = [&](const char *name, auto launch)
    {
        for (int i = 0; i < 3; ++i)
            launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i)
            launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-34s %8.1f us  %7.1f GB/s\n", name, 1e3 * ms / reps, gb / (1e-3 * ms / reps));
    };
    std::printf("n_patches %d rows %d (%.1f KiB per patch, %.3f GB), W = %d FMA per row\n", n_patches, rows, rows * 0.5, gb, W);
    time("copy (nontemporal 16 B/lane)", [&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const dbl2 *>(M), out, n / 2); });
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int wpc : {8, 12, 16, 32})
    {
        char name[96];
        std::snprintf(name, sizeof name, "reg chain, %d waves/CU (LDS-limited)", wpc);
        time(name, [&] { hipLaunchKernelGGL(reg_kernel, dim3(8 * chunk), dim3(64), 163840 / wpc - 256, 0, M, out, n_patches, rows, chunk); });
    }
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(reg16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int wpc : {8, 12, 16})
    {
        char name[96];
        std::snprintf(name, sizeof name, "reg chain 16 B/lane, %d waves/CU", wpc);
        time(name, [&] { hipLaunchKernelGGL(reg16_kernel, dim3(8 * chunk), dim3(64), 163840 / wpc - 256, 0, M, out, n_patches, rows, chunk); });
    }
    if (rows == 140)
    {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(regall_kernel<140>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        for (int wpc : {4, 8})
        {
            char name[96];
            std::snprintf(name, sizeof name, "all 140 rows up front, %d waves/CU", wpc);
            time(name, [&] { hipLaunchKernelGGL(regall_kernel<140>, dim3(8 * chunk), dim3(64), 163840 / wpc - 256, 0, M, out, n_patches, chunk); });
        }
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(regall_kernel<76>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    }
    std::vector<double> ref((size_t)n_patches * 64), got(ref.size());
    CHECK(hipMemcpy(ref.data(), out, ref.size() * 8, hipMemcpyDeviceToHost));
    for (int rs : {RS, 16, 22, 38, 60})
        for (int step : {1, 0})
        {
            const size_t ldsb = (size_t)rs * 1024 + extra;
            CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            char name[96];
            std::snprintf(name, sizeof name, "dma ring %2d KiB refill %s (%d wg/CU)", rs, step ? "after" : "before", (int)(163840 / ldsb));
            time(name, [&] { hipLaunchKernelGGL(dma_kernel, dim3(8 * chunk), dim3(64), ldsb, 0, M, out, n_patches, rows, rs, chunk, step); });
            CHECK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < got.size(); ++i)
                bad += got[i] != ref[i];
            if (bad)
                std::printf("   MISMATCH vs reg kernel in %zu of %zu values\n", bad, got.size());
        }
    return 0;
}
