// Streaming-rate variants for the BLAS-1 layer (copy = read + write bytes; read = read-only stream) on 1 GiB of doubles.
// build: hipcc -O3 --offload-arch=gfx950 stream_copy.hip -o stream_copy
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// grid-stride, U independent 16-byte accesses in flight per thread; NT: non-temporal loads and stores
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy_stride(const dbl2 *__restrict__ x, dbl2 *__restrict__ y, long long n2)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = blockIdx.x * 256LL + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride)
    {
        dbl2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NTL ? __builtin_nontemporal_load(&x[i + u * stride]) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            if (NTS)
                __builtin_nontemporal_store(v[u], &y[i + u * stride]);
            else
                y[i + u * stride] = v[u];
        }
    }
    for (; i < n2; i += stride)
        y[i] = x[i];
}

// one contiguous chunk per workgroup (U x 256 x 16 B tiles), no grid-stride loop: grid = n2 / (256 U)
template <int U, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) copy_tile(const dbl2 *__restrict__ x, dbl2 *__restrict__ y, long long n2)
{
    const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
    dbl2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n2)
            v[u] = NTL ? __builtin_nontemporal_load(&x[base + u * 256]) : x[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + u * 256 < n2)
        {
            if (NTS)
                __builtin_nontemporal_store(v[u], &y[base + u * 256]);
            else
                y[base + u * 256] = v[u];
        }
}

template <int U, bool NTL>
__global__ void __launch_bounds__(256) read_stride(const dbl2 *__restrict__ x, double *__restrict__ out, long long n2)
{
    const long long stride = (long long)gridDim.x * 256;
    double acc = 0.0;
    long long i = blockIdx.x * 256LL + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride)
    {
        dbl2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = NTL ? __builtin_nontemporal_load(&x[i + u * stride]) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += v[u].x * v[u].x + v[u].y * v[u].y;
    }
    for (; i < n2; i += stride)
        acc += x[i].x * x[i].x + x[i].y * x[i].y;
    if (acc == 12345.678)
        out[0] = acc;
}

template <typename F>
void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-64s %8.1f us  %7.1f GB/s\n", name, 1e3 * ms / reps, bytes * reps / (1e-3 * ms) / 1e9);
}

int main()
{
    const long long n = 1LL << 27, n2 = n / 2;
    double *x, *y;
    CHECK(hipMalloc(&x, n * 8));
    CHECK(hipMalloc(&y, n * 8));
    CHECK(hipMemset(x, 0, n * 8));
    CHECK(hipMemset(y, 0, n * 8));
    const dbl2 *X = reinterpret_cast<const dbl2 *>(x);
    dbl2 *Y = reinterpret_cast<dbl2 *>(y);
    const double cb = 2.0 * 8.0 * n, rb = 8.0 * n;
    for (int g : {2048, 4096, 8192, 16384})
    {
        char nm[128];
        std::snprintf(nm, sizeof nm, "copy grid-stride U=4 plain, %d workgroups", g);
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy_stride<4, false, false>), dim3(g), dim3(256), 0, 0, X, Y, n2); });
        std::snprintf(nm, sizeof nm, "copy grid-stride U=4 nt load + nt store, %d workgroups", g);
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy_stride<4, true, true>), dim3(g), dim3(256), 0, 0, X, Y, n2); });
        std::snprintf(nm, sizeof nm, "copy grid-stride U=4 nt load only, %d workgroups", g);
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy_stride<4, true, false>), dim3(g), dim3(256), 0, 0, X, Y, n2); });
        std::snprintf(nm, sizeof nm, "copy grid-stride U=8 nt load + nt store, %d workgroups", g);
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy_stride<8, true, true>), dim3(g), dim3(256), 0, 0, X, Y, n2); });
        std::snprintf(nm, sizeof nm, "copy grid-stride U=1 nt load + nt store, %d workgroups", g);
        timeit(nm, cb, [&] { hipLaunchKernelGGL((copy_stride<1, true, true>), dim3(g), dim3(256), 0, 0, X, Y, n2); });
        std::snprintf(nm, sizeof nm, "read grid-stride U=4 plain, %d workgroups", g);
        timeit(nm, rb, [&] { hipLaunchKernelGGL((read_stride<4, false>), dim3(g), dim3(256), 0, 0, X, y, n2); });
        std::snprintf(nm, sizeof nm, "read grid-stride U=4 nt, %d workgroups", g);
        timeit(nm, rb, [&] { hipLaunchKernelGGL((read_stride<4, true>), dim3(g), dim3(256), 0, 0, X, y, n2); });
        std::snprintf(nm, sizeof nm, "read grid-stride U=1 nt, %d workgroups", g);
        timeit(nm, rb, [&] { hipLaunchKernelGGL((read_stride<1, true>), dim3(g), dim3(256), 0, 0, X, y, n2); });
    }
    timeit("copy one tile per workgroup U=4 nt/nt", cb, [&] { hipLaunchKernelGGL((copy_tile<4, true, true>), dim3((n2 + 1023) / 1024), dim3(256), 0, 0, X, Y, n2); });
    timeit("copy one tile per workgroup U=8 nt/nt", cb, [&] { hipLaunchKernelGGL((copy_tile<8, true, true>), dim3((n2 + 2047) / 2048), dim3(256), 0, 0, X, Y, n2); });
    timeit("copy one tile per workgroup U=4 plain", cb, [&] { hipLaunchKernelGGL((copy_tile<4, false, false>), dim3((n2 + 1023) / 1024), dim3(256), 0, 0, X, Y, n2); });
    timeit("hipMemcpyDtoD", cb, [&] { (void)hipMemcpyAsync(y, x, n * 8, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
