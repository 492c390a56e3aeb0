// Level-1 kernels: streaming maps with 16-byte accesses and a deterministic
// two-stage reduction (wavefront shuffles, then LDS across the four waves of a
// workgroup, then one wavefront over the per-workgroup partials).
#include "common.hpp"

using namespace cuddh_k;

namespace
{
    constexpr int BLOCK = 256;
    constexpr int MAX_PARTIALS = 1024;

    template <typename T>
    struct Pack; // 16-byte vector of T
    template <>
    struct Pack<double>
    {
        using type = double2;
        static constexpr int N = 2;
    };
    template <>
    struct Pack<float>
    {
        using type = float4;
        static constexpr int N = 4;
    };
    template <>
    struct Pack<int>
    {
        using type = int4;
        static constexpr int N = 4;
    };

    inline bool aligned16(const void *p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

    // Access pattern of every streaming kernel here (profiles/r03/stream_copy_variants.txt, 1 GiB of doubles on MI355X): a
    // workgroup works on contiguous TILES of UNROLL x 256 sixteen-byte vectors, a thread's UNROLL accesses 256 vectors (4 KiB)
    // apart, and workgroups take tiles in order -- at any moment the chip reads one contiguous window.  A copy that way runs at
    // 6.2 TB/s (read + write); the former grid-stride form, whose four accesses per thread were a whole grid apart, at 4.4.
    // Vectors too large to stay in the 256 MB infinity cache anyway are read and written with the non-temporal hint.
    constexpr int UNROLL = 4;
    constexpr int TILE = UNROLL * BLOCK; // 16-byte vectors per tile
    constexpr long long NT_BYTES = 64LL << 20;

    template <bool NT, typename V>
    __device__ inline V stream_load(const V *p)
    {
        if constexpr (NT)
            return __builtin_nontemporal_load(p);
        else
            return *p;
    }
    template <bool NT, typename V>
    __device__ inline void stream_store(V *p, const V &v)
    {
        if constexpr (NT)
            __builtin_nontemporal_store(v, p);
        else
            *p = v;
    }
    inline int tile_grid(long long n_vectors, int cap)
    {
        long long g = (n_vectors + TILE - 1) / TILE;
        if (g > cap)
            g = cap;
        return static_cast<int>(g < 1 ? 1 : g);
    }

    // ---------------- y = f(x, y) element-wise; F is a functor T(T x, T y)
    template <typename T, typename F, bool READ_X, bool READ_Y, bool NT>
    __global__ void __launch_bounds__(BLOCK) map_kernel(int n, const T *__restrict__ x, T *__restrict__ y, F f, int vectorised)
    {
        using V = typename Pack<T>::type;
        using VE = T __attribute__((ext_vector_type(Pack<T>::N))); // (clang vector: what the non-temporal builtins take)
        constexpr int N = Pack<T>::N;
        static_assert(sizeof(V) == sizeof(VE), "16-byte vectors");
        int done = 0;
        if (vectorised)
        {
            const int nv = n / N;
            const int n_tiles = (nv + TILE - 1) / TILE;
            const VE *xv = reinterpret_cast<const VE *>(x);
            VE *yv = reinterpret_cast<VE *>(y);
            for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
            {
                const int base = tile * TILE + threadIdx.x;
                VE a[UNROLL], b[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                {
                    // clamped, not guarded: a request under `if (i < nv)` is a basic block of its own and the tile's requests
                    // would go out one group at a time (see mgs_stage_kernel)
                    const int i = min(base + u * BLOCK, nv - 1);
                    if (READ_X)
                        a[u] = stream_load<NT>(xv + i);
                    if (READ_Y)
                        b[u] = stream_load<NT>(yv + i);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                {
                    const int i = base + u * BLOCK;
                    if (i < nv)
                    {
                        VE r;
#pragma unroll
                        for (int c = 0; c < N; ++c)
                            r[c] = f(READ_X ? a[u][c] : T(0), READ_Y ? b[u][c] : T(0));
                        stream_store<NT>(yv + i, r);
                    }
                }
            }
            done = nv * N;
        }
        const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
        for (int i = done + tid; i < n; i += stride)
            y[i] = f(READ_X ? x[i] : T(0), READ_Y ? y[i] : T(0));
    }

    template <typename T, bool RX, bool RY, typename F>
    int launch_map(int n, const T *x, T *y, F f, void *stream)
    {
        if (n <= 0)
            return 0;
        const int vec = aligned16(y) && (!RX || aligned16(x));
        const int g = tile_grid(n / Pack<T>::N, 1 << 20);
        if (static_cast<long long>(n) * sizeof(T) >= NT_BYTES)
            hipLaunchKernelGGL((map_kernel<T, F, RX, RY, true>), dim3(g), dim3(BLOCK), 0, as_stream(stream), n, x, y, f, vec);
        else
            hipLaunchKernelGGL((map_kernel<T, F, RX, RY, false>), dim3(g), dim3(BLOCK), 0, as_stream(stream), n, x, y, f, vec);
        return launch_status();
    }

    // functors
    template <typename T>
    struct Axpby
    {
        T a, b;
        __device__ T operator()(T x, T y) const { return a * x + b * y; }
    };
    template <typename T>
    struct Ax
    {
        T a;
        __device__ T operator()(T x, T) const { return a * x; }
    };
    template <typename T>
    struct AxpbyDev
    {
        T sa;
        const T *a;
        T b;
        __device__ T operator()(T x, T y) const { return (sa * *a) * x + b * y; }
    };
    template <typename T>
    struct ScaleInvDev
    {
        const T *a;
        __device__ T operator()(T, T y) const { return y / *a; }
    };
    template <typename T>
    struct Scale
    {
        T a;
        __device__ T operator()(T, T y) const { return y * a; }
    };
    template <typename T>
    struct Const
    {
        T a;
        __device__ T operator()(T, T) const { return a; }
    };
    template <typename T>
    struct Ident
    {
        __device__ T operator()(T x, T) const { return x; }
    };
    struct Recip
    {
        __device__ double operator()(double, double y) const { return 1.0 / y; }
    };

    // ---------------- reductions
    template <typename T>
    __device__ inline T block_sum(T v)
    {
        __shared__ T part[BLOCK / WAVE];
        v = wave_sum(v);
        const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
        if (lane == 0)
            part[w] = v;
        __syncthreads();
        T s = T(0);
        if (threadIdx.x == 0)
        {
#pragma unroll
            for (int i = 0; i < BLOCK / WAVE; ++i)
                s += part[i];
        }
        return s; // valid in thread 0
    }

    // MODE 0: sum x*y   1: sum (x-y)^2   2: sum x*x (one stream: y is not read)
    template <typename T, int MODE, bool NT>
    __global__ void __launch_bounds__(BLOCK) reduce_stage1(int n, const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ partial,
                                                           int vectorised)
    {
        using VE = T __attribute__((ext_vector_type(Pack<T>::N)));
        constexpr int N = Pack<T>::N;
        T acc = T(0);
        int done = 0;
        auto term = [&](T a, T b)
        {
            if (MODE == 0)
                acc += a * b;
            else if (MODE == 1)
            {
                const T d = a - b;
                acc += d * d;
            }
            else
                acc += a * a;
        };
        if (vectorised)
        {
            const int nv = n / N;
            const int n_tiles = (nv + TILE - 1) / TILE;
            const VE *xv = reinterpret_cast<const VE *>(x), *yv = reinterpret_cast<const VE *>(y);
            for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) // contiguous tiles, taken in order by the workgroups
            {
                const int base = tile * TILE + threadIdx.x;
                VE a[UNROLL], b[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                {
                    const int i = min(base + u * BLOCK, nv - 1); // (clamped, not guarded: see map_kernel)
                    a[u] = stream_load<NT>(xv + i);
                    if (MODE != 2)
                        b[u] = stream_load<NT>(yv + i);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    if (base + u * BLOCK < nv)
#pragma unroll
                        for (int c = 0; c < N; ++c)
                            term(a[u][c], MODE != 2 ? b[u][c] : T(0));
            }
            done = nv * N;
        }
        const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
        for (int i = done + tid; i < n; i += stride)
            term(x[i], MODE != 2 ? y[i] : T(0));
        const T s = block_sum(acc);
        if (threadIdx.x == 0)
            partial[blockIdx.x] = s;
    }

    template <typename T, bool SQRT>
    __global__ void __launch_bounds__(BLOCK) reduce_stage2(int n_partial, const T *__restrict__ partial, T *__restrict__ result)
    {
        T acc = T(0);
        for (int i = threadIdx.x; i < n_partial; i += BLOCK)
            acc += partial[i];
        const T s = block_sum(acc);
        if (threadIdx.x == 0)
            *result = SQRT ? sqrt(s) : s;
    }

    template <typename T, int MODE, bool SQRT>
    int launch_reduce(int n, const T *x, const T *y, T *result, void *ws, void *stream)
    {
        if constexpr (MODE == 0)
            if (x == y) // <x, x>: one stream instead of two (the same products, bit for bit)
                return launch_reduce<T, 2, SQRT>(n, x, y, result, ws, stream);
        T *partial = static_cast<T *>(ws);
        const int g = tile_grid(n / Pack<T>::N, MAX_PARTIALS);
        const int vec = aligned16(x) && aligned16(y);
        if (static_cast<long long>(n) * sizeof(T) >= NT_BYTES)
            hipLaunchKernelGGL((reduce_stage1<T, MODE, true>), dim3(g), dim3(BLOCK), 0, as_stream(stream), n, x, y, partial, vec);
        else
            hipLaunchKernelGGL((reduce_stage1<T, MODE, false>), dim3(g), dim3(BLOCK), 0, as_stream(stream), n, x, y, partial, vec);
        hipLaunchKernelGGL((reduce_stage2<T, SQRT>), dim3(1), dim3(BLOCK), 0, as_stream(stream), g, partial, result);
        return launch_status();
    }

    // explicit fused multiply-adds in the Gram-Schmidt kernels: left to the compiler, the fp32 stage kernel got packed multiplies + adds
    // (two roundings) where the same expressions in another kernel became FMAs -- the results must not depend on that choice
    __device__ inline double fmadd(double a, double b, double c) { return __builtin_fma(a, b, c); }
    __device__ inline float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

    // ---------------- fused modified Gram-Schmidt stage (one launch per projection instead of dot + reduce + axpy)
    // stage:  h = sum(pin)            (the coefficient <w, v_prev> whose partial sums the previous stage left behind)
    //         w <- w - h * v_prev     (skipped when v_prev == nullptr: first stage)
    //         pout[block] = partial sum of <w, v_next>   (v_next == nullptr: <w, w>, for the norm)
    // Every workgroup sums the same <= MAX_PARTIALS partials in the same order, so all of them use the same h.
    // PREV / NEXT: v_prev / v_next given (compile-time, so that no request sits under a run-time condition).  A workgroup's FIRST tile
    // is requested before the coefficient is summed -- the requests do not depend on it -- with clamped addresses and no branches,
    // and the <= 4 partial sums per thread go out with them: for vectors of a few MB (config 2: 9.5 MB) a stage is a chain of
    // dependent trips, not a bandwidth problem, and the former form (requests under `if (i < nv)` inside the tile loop, whose header
    // waits for everything outstanding; the partial sums in a loop of their own) made seven of them in a row
    // (profiles/r03/mgs_stage_small_vectors.txt).
    template <typename T, bool PREV, bool NEXT>
    __global__ void __launch_bounds__(BLOCK) mgs_stage_kernel(int n, T *__restrict__ w, const T *__restrict__ vprev, const T *__restrict__ vnext,
                                                              const T *__restrict__ pin, int npin, T *__restrict__ pout, T *__restrict__ hout,
                                                              int vectorised)
    {
        __shared__ T h_sh;
        using VE = T __attribute__((ext_vector_type(Pack<T>::N)));
        constexpr int N = Pack<T>::N;
        const int nv = vectorised ? n / N : 0;
        const int n_tiles = (nv + TILE - 1) / TILE;
        VE *wv = reinterpret_cast<VE *>(w);
        const VE *pv = reinterpret_cast<const VE *>(vprev), *nvv = reinterpret_cast<const VE *>(vnext);
        int tile = blockIdx.x;
        const bool first = tile < n_tiles; // (workgroup-uniform)
        VE a0[UNROLL], b0[UNROLL], c0[UNROLL];
        if (first)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const int i = min(tile * TILE + (int)threadIdx.x + u * BLOCK, nv - 1);
                a0[u] = wv[i];
                if constexpr (PREV)
                    b0[u] = pv[i];
                if constexpr (NEXT)
                    c0[u] = nvv[i];
            }
        }
        T h = T(0);
        if constexpr (PREV)
        {
            static_assert(MAX_PARTIALS <= 4 * BLOCK, "four partial sums per thread");
            T q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = pin[min((int)threadIdx.x + k * BLOCK, npin - 1)];
            T a = T(0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                a = (int)threadIdx.x + k * BLOCK < npin ? a + q[k] : a; // (the order of the former loop)
            const T s = block_sum(a);
            if (threadIdx.x == 0)
            {
                h_sh = s;
                if (blockIdx.x == 0)
                    *hout = s;
            }
            __syncthreads();
            h = h_sh;
        }
        T acc = T(0);
        const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
        if (first)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const int i = tile * TILE + (int)threadIdx.x + u * BLOCK;
                if (i < nv)
                {
#pragma unroll
                    for (int e = 0; e < N; ++e)
                    {
                        if constexpr (PREV)
                            a0[u][e] = fmadd(-h, b0[u][e], a0[u][e]);
                        acc = fmadd(a0[u][e], NEXT ? c0[u][e] : a0[u][e], acc);
                    }
                    if constexpr (PREV)
                        wv[i] = a0[u];
                }
            }
            tile += gridDim.x;
        }
        // further tiles (vectors of more than MAX_PARTIALS tiles, 16 MB: enough workgroups in flight to hide the trips)
        for (; tile < n_tiles; tile += gridDim.x)
        {
            const int base = tile * TILE + threadIdx.x;
            VE a[UNROLL], b[UNROLL], c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const int i = min(base + u * BLOCK, nv - 1); // (clamped, not guarded: all requests of a tile go out together)
                a[u] = wv[i];
                if constexpr (PREV)
                    b[u] = pv[i];
                if constexpr (NEXT)
                    c[u] = nvv[i];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const int i = base + u * BLOCK;
                if (i < nv)
                {
#pragma unroll
                    for (int e = 0; e < N; ++e)
                    {
                        if constexpr (PREV)
                            a[u][e] = fmadd(-h, b[u][e], a[u][e]);
                        acc = fmadd(a[u][e], NEXT ? c[u][e] : a[u][e], acc);
                    }
                    if constexpr (PREV)
                        wv[i] = a[u];
                }
            }
        }
        for (int i = nv * N + tid; i < n; i += stride)
        {
            T wi = w[i];
            if constexpr (PREV)
            {
                wi = fmadd(-h, vprev[i], wi);
                w[i] = wi;
            }
            acc = fmadd(wi, NEXT ? vnext[i] : wi, acc);
        }
        __syncthreads(); // block_sum reuses its LDS scratch
        const T s = block_sum(acc);
        if (threadIdx.x == 0)
            pout[blockIdx.x] = s;
    }

    // nrm = sqrt(sum(pin));  *hout = nrm;  w <- w / nrm
    template <typename T>
    __global__ void __launch_bounds__(BLOCK) mgs_finish_kernel(int n, T *__restrict__ w, const T *__restrict__ pin, int npin, T *__restrict__ hout,
                                                               int vectorised)
    {
        __shared__ T nrm_sh;
        using VE = T __attribute__((ext_vector_type(Pack<T>::N)));
        constexpr int N = Pack<T>::N;
        const int nv = vectorised ? n / N : 0;
        const int n_tiles = (nv + TILE - 1) / TILE;
        VE *wv = reinterpret_cast<VE *>(w);
        int tile = blockIdx.x;
        const bool first = tile < n_tiles; // the first tile is requested before the norm is summed (see mgs_stage_kernel)
        VE a0[UNROLL];
        if (first)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                a0[u] = wv[min(tile * TILE + (int)threadIdx.x + u * BLOCK, nv - 1)];
        }
        T q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            q[k] = pin[min((int)threadIdx.x + k * BLOCK, npin - 1)];
        T a = T(0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            a = (int)threadIdx.x + k * BLOCK < npin ? a + q[k] : a;
        const T s = block_sum(a);
        if (threadIdx.x == 0)
        {
            nrm_sh = sqrt(s);
            if (blockIdx.x == 0)
                *hout = nrm_sh;
        }
        __syncthreads();
        const T nrm = nrm_sh;
        const int tid = blockIdx.x * BLOCK + threadIdx.x, stride = gridDim.x * BLOCK;
        if (first)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const int i = tile * TILE + (int)threadIdx.x + u * BLOCK;
                if (i < nv)
                {
#pragma unroll
                    for (int c = 0; c < N; ++c)
                        a0[u][c] = a0[u][c] / nrm;
                    wv[i] = a0[u];
                }
            }
            tile += gridDim.x;
        }
        for (; tile < n_tiles; tile += gridDim.x)
        {
            const int base = tile * TILE + threadIdx.x;
            VE av[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                av[u] = wv[min(base + u * BLOCK, nv - 1)];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                if (base + u * BLOCK < nv)
                {
#pragma unroll
                    for (int c = 0; c < N; ++c)
                        av[u][c] = av[u][c] / nrm;
                    wv[base + u * BLOCK] = av[u];
                }
        }
        for (int i = nv * N + tid; i < n; i += stride)
            w[i] = w[i] / nrm;
    }

    inline int mgs_grid(int n) { return tile_grid(n / 2, MAX_PARTIALS); } // (n / 2: at least as many tiles as either precision has)

    template <typename T>
    int launch_mgs_stage(int n, T *w, const T *vprev, const T *vnext, const T *pin, T *pout, T *hout, void *stream)
    {
        const int g = mgs_grid(n);
        const int vec = aligned16(w) && (!vprev || aligned16(vprev)) && (!vnext || aligned16(vnext));
        const dim3 grid(g), block(BLOCK);
        hipStream_t st = as_stream(stream);
        if (vprev && vnext)
            hipLaunchKernelGGL((mgs_stage_kernel<T, true, true>), grid, block, 0, st, n, w, vprev, vnext, pin, g, pout, hout, vec);
        else if (vprev)
            hipLaunchKernelGGL((mgs_stage_kernel<T, true, false>), grid, block, 0, st, n, w, vprev, vnext, pin, g, pout, hout, vec);
        else if (vnext)
            hipLaunchKernelGGL((mgs_stage_kernel<T, false, true>), grid, block, 0, st, n, w, vprev, vnext, pin, g, pout, hout, vec);
        else
            hipLaunchKernelGGL((mgs_stage_kernel<T, false, false>), grid, block, 0, st, n, w, vprev, vnext, pin, g, pout, hout, vec);
        return launch_status();
    }

    template <typename T>
    int launch_mgs_finish(int n, T *w, const T *pin, T *hout, void *stream)
    {
        const int g = mgs_grid(n);
        hipLaunchKernelGGL((mgs_finish_kernel<T>), dim3(g), dim3(BLOCK), 0, as_stream(stream), n, w, pin, g, hout, aligned16(w) ? 1 : 0);
        return launch_status();
    }

    // ---------------- indexed maps
    __global__ void __launch_bounds__(BLOCK) gather_kernel(int n, const int *__restrict__ proj, const double *__restrict__ x, double *__restrict__ y)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
            y[i] = x[proj[i]];
    }

    __global__ void __launch_bounds__(BLOCK) scatter_add_kernel(int n, const int *__restrict__ proj, const double *__restrict__ x, double *__restrict__ y)
    {
        // proj is injective (FaceSpace dof -> distinct H1 dof), so a plain read-modify-write is race free
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
            y[proj[i]] += x[i];
    }

    __global__ void __launch_bounds__(BLOCK) zero_indexed_kernel(int n, const int *__restrict__ proj, double *__restrict__ x)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
            x[proj[i]] = 0.0;
    }

    // trace exchange of the multi-GPU DDH path (slot t stands for entries t and n_half + t of a trace vector):
    // pack: buf[i] = v[slot[i]], buf[n + i] = v[n_half + slot[i]] and, with clear != 0, v <- 0 there (the slots belong to the
    // receiving rank); unpack: v[slot[i]] = buf[i], v[n_half + slot[i]] = buf[n + i]
    template <typename T>
    __global__ void __launch_bounds__(BLOCK) trace_pack_kernel(int n, int n_half, const int *__restrict__ slot, T *__restrict__ v, T *__restrict__ buf, int clear)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const int t = slot[i];
            buf[i] = v[t];
            buf[n + i] = v[n_half + t];
            if (clear)
            {
                v[t] = T(0);
                v[n_half + t] = T(0);
            }
        }
    }

    template <typename T>
    __global__ void __launch_bounds__(BLOCK) trace_unpack_kernel(int n, int n_half, const int *__restrict__ slot, const T *__restrict__ buf, T *__restrict__ v)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const int t = slot[i];
            v[t] = buf[i];
            v[n_half + t] = buf[n + i];
        }
    }

    // halo exchange of the partitioned operator apply: entries t and n_half + t of a local [u; v] vector travel as the PAIR
    // (buf[2 i], buf[2 i + 1]), so the messages for several neighbours are contiguous pieces of one buffer packed by one launch
    __global__ void __launch_bounds__(BLOCK) halo_pack_kernel(int n, int n_half, const int *__restrict__ ids, double *__restrict__ v, double2 *__restrict__ buf,
                                                              int clear)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const int t = ids[i];
            buf[i] = make_double2(v[t], v[n_half + t]);
            if (clear)
            {
                v[t] = 0.0;
                v[n_half + t] = 0.0;
            }
        }
    }

    __global__ void __launch_bounds__(BLOCK) halo_unpack_kernel(int n, int n_half, const int *__restrict__ ids, const double2 *__restrict__ buf,
                                                                double *__restrict__ v, int add)
    {
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const int t = ids[i]; // the ids of one unpack are distinct: a dof has one owner and, per sender, appears once
            const double2 b = buf[i];
            v[t] = add ? v[t] + b.x : b.x;
            v[n_half + t] = add ? v[n_half + t] + b.y : b.y;
        }
    }

    // y[r] = (accumulate ? y[r] : 0) + sum_{k in [off[r], off[r+1])} x[src[k]], the terms added in the order they are listed
    __global__ void __launch_bounds__(BLOCK) csr_sum_kernel(int n_rows, const int *__restrict__ off, const int *__restrict__ src, const double *__restrict__ x,
                                                            double *__restrict__ y, int accumulate)
    {
        for (int r = blockIdx.x * BLOCK + threadIdx.x; r < n_rows; r += gridDim.x * BLOCK)
        {
            double s = accumulate ? y[r] : 0.0;
            for (int k = off[r]; k < off[r + 1]; ++k)
                s += x[src[k]];
            y[r] = s;
        }
    }

    __global__ void __launch_bounds__(BLOCK) diag_scale_kernel(int n, int accumulate, double c, const double *__restrict__ p, const double *x, double *y)
    {
        // x may alias y (DiagInvMassMatrix is applied in place by the DDH example)
        for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const double v = c * p[i] * x[i];
            y[i] = accumulate ? y[i] + v : v;
        }
    }
} // namespace

extern "C"
{
    size_t cuddh_hip_reduce_ws_bytes(void) { return 2 * MAX_PARTIALS * sizeof(double); }

    int cuddh_hip_mgs_stage_f64(int n, double *w, const double *vprev, const double *vnext, const double *pin, double *pout, double *hout, void *s)
    {
        return launch_mgs_stage<double>(n, w, vprev, vnext, pin, pout, hout, s);
    }
    int cuddh_hip_mgs_stage_f32(int n, float *w, const float *vprev, const float *vnext, const float *pin, float *pout, float *hout, void *s)
    {
        return launch_mgs_stage<float>(n, w, vprev, vnext, pin, pout, hout, s);
    }
    int cuddh_hip_mgs_finish_f64(int n, double *w, const double *pin, double *hout, void *s) { return launch_mgs_finish<double>(n, w, pin, hout, s); }
    int cuddh_hip_mgs_finish_f32(int n, float *w, const float *pin, float *hout, void *s) { return launch_mgs_finish<float>(n, w, pin, hout, s); }

    int cuddh_hip_axpby_f64(int n, double a, const double *x, double b, double *y, void *s)
    {
        if (b == 0.0)
            return launch_map<double, true, false>(n, x, y, Ax<double>{a}, s);
        return launch_map<double, true, true>(n, x, y, Axpby<double>{a, b}, s);
    }
    int cuddh_hip_axpby_f32(int n, float a, const float *x, float b, float *y, void *s)
    {
        if (b == 0.0f)
            return launch_map<float, true, false>(n, x, y, Ax<float>{a}, s);
        return launch_map<float, true, true>(n, x, y, Axpby<float>{a, b}, s);
    }
    int cuddh_hip_axpby_dev_f64(int n, double sa, const double *a, const double *x, double b, double *y, void *s)
    {
        return launch_map<double, true, true>(n, x, y, AxpbyDev<double>{sa, a, b}, s);
    }
    int cuddh_hip_axpby_dev_f32(int n, float sa, const float *a, const float *x, float b, float *y, void *s)
    {
        return launch_map<float, true, true>(n, x, y, AxpbyDev<float>{sa, a, b}, s);
    }
    int cuddh_hip_scal_inv_dev_f64(int n, const double *a, double *x, void *s)
    {
        return launch_map<double, false, true>(n, static_cast<const double *>(nullptr), x, ScaleInvDev<double>{a}, s);
    }
    int cuddh_hip_scal_inv_dev_f32(int n, const float *a, float *x, void *s)
    {
        return launch_map<float, false, true>(n, static_cast<const float *>(nullptr), x, ScaleInvDev<float>{a}, s);
    }

    int cuddh_hip_dot_f64(int n, const double *x, const double *y, double *r, void *ws, void *s) { return launch_reduce<double, 0, false>(n, x, y, r, ws, s); }
    int cuddh_hip_dot_f32(int n, const float *x, const float *y, float *r, void *ws, void *s) { return launch_reduce<float, 0, false>(n, x, y, r, ws, s); }
    int cuddh_hip_nrm2_f64(int n, const double *x, double *r, void *ws, void *s) { return launch_reduce<double, 2, true>(n, x, x, r, ws, s); }
    int cuddh_hip_nrm2_f32(int n, const float *x, float *r, void *ws, void *s) { return launch_reduce<float, 2, true>(n, x, x, r, ws, s); }
    int cuddh_hip_sqdist_f64(int n, const double *x, const double *y, double *r, void *ws, void *s) { return launch_reduce<double, 1, false>(n, x, y, r, ws, s); }
    int cuddh_hip_sqdist_f32(int n, const float *x, const float *y, float *r, void *ws, void *s) { return launch_reduce<float, 1, false>(n, x, y, r, ws, s); }

    int cuddh_hip_copy_f64(int n, const double *x, double *y, void *s) { return launch_map<double, true, false>(n, x, y, Ident<double>{}, s); }
    int cuddh_hip_copy_f32(int n, const float *x, float *y, void *s) { return launch_map<float, true, false>(n, x, y, Ident<float>{}, s); }
    int cuddh_hip_copy_i32(int n, const int *x, int *y, void *s) { return launch_map<int, true, false>(n, x, y, Ident<int>{}, s); }

    int cuddh_hip_scal_f64(int n, double a, double *x, void *s) { return launch_map<double, false, true>(n, static_cast<const double *>(nullptr), x, Scale<double>{a}, s); }
    int cuddh_hip_scal_f32(int n, float a, float *x, void *s) { return launch_map<float, false, true>(n, static_cast<const float *>(nullptr), x, Scale<float>{a}, s); }

    int cuddh_hip_fill_f64(int n, double a, double *x, void *s) { return launch_map<double, false, false>(n, static_cast<const double *>(nullptr), x, Const<double>{a}, s); }
    int cuddh_hip_fill_f32(int n, float a, float *x, void *s) { return launch_map<float, false, false>(n, static_cast<const float *>(nullptr), x, Const<float>{a}, s); }
    int cuddh_hip_fill_i32(int n, int a, int *x, void *s) { return launch_map<int, false, false>(n, static_cast<const int *>(nullptr), x, Const<int>{a}, s); }

    int cuddh_hip_reciprocal_f64(int n, double *x, void *s) { return launch_map<double, false, true>(n, static_cast<const double *>(nullptr), x, Recip{}, s); }

    int cuddh_hip_diag_scale_f64(int n, int accumulate, double c, const double *p, const double *x, double *y, void *s)
    {
        if (n <= 0)
            return 0;
        hipLaunchKernelGGL(diag_scale_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, accumulate, c, p, x, y);
        return launch_status();
    }

    int cuddh_hip_gather_f64(int n, const int *proj, const double *x, double *y, void *s)
    {
        if (n <= 0)
            return 0;
        hipLaunchKernelGGL(gather_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, proj, x, y);
        return launch_status();
    }
    int cuddh_hip_scatter_add_f64(int n, const int *proj, const double *x, double *y, void *s)
    {
        if (n <= 0)
            return 0;
        hipLaunchKernelGGL(scatter_add_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, proj, x, y);
        return launch_status();
    }
    int cuddh_hip_trace_pack_f32(int n, int n_half, const int *slot, float *v, float *buf, int clear, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL((trace_pack_kernel<float>), dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, slot, v, buf, clear);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_trace_pack_f64(int n, int n_half, const int *slot, double *v, double *buf, int clear, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL((trace_pack_kernel<double>), dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, slot, v, buf, clear);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_trace_unpack_f32(int n, int n_half, const int *slot, const float *buf, float *v, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL((trace_unpack_kernel<float>), dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, slot, buf, v);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_trace_unpack_f64(int n, int n_half, const int *slot, const double *buf, double *v, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL((trace_unpack_kernel<double>), dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, slot, buf, v);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_halo_pack_f64(int n, int n_half, const int *ids, double *v, double *buf, int clear, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL(halo_pack_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, ids, v, reinterpret_cast<double2 *>(buf), clear);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_halo_unpack_f64(int n, int n_half, const int *ids, const double *buf, double *v, int add, void *s)
    {
        if (n > 0)
            hipLaunchKernelGGL(halo_unpack_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, n_half, ids, reinterpret_cast<const double2 *>(buf), v, add);
        return n > 0 ? launch_status() : 0;
    }
    int cuddh_hip_csr_sum_f64(int n_rows, const int *off, const int *src, const double *x, double *y, int accumulate, void *s)
    {
        if (n_rows > 0)
            hipLaunchKernelGGL(csr_sum_kernel, dim3(stream_grid(n_rows, BLOCK)), dim3(BLOCK), 0, as_stream(s), n_rows, off, src, x, y, accumulate);
        return n_rows > 0 ? launch_status() : 0;
    }
    int cuddh_hip_zero_indexed_f64(int n, const int *proj, double *x, void *s)
    {
        if (n <= 0)
            return 0;
        hipLaunchKernelGGL(zero_indexed_kernel, dim3(stream_grid(n, BLOCK)), dim3(BLOCK), 0, as_stream(s), n, proj, x);
        return launch_status();
    }
}
