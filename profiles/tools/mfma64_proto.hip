// Feasibility prototype for the high-order (n_basis 8) stiffness apply on v_mfma_f64_16x16x4_f64 (DESIGN.md section 8,
// next step 1).  NOT part of the product: it runs only the inner loop of the proposed kernel -- 16 elements per wavefront,
// lane = element + 16 g, per eta-quadrature slice r: in-lane eta contraction, 4 forward MFMAs (xi contraction to the
// quadrature rows in C layout), flux with streamed metric values, 8 backward MFMAs, in-lane accumulation -- with the
// element values taken from registers instead of an x gather and the result reduced to one number per lane.  It answers
// one question: how fast can that loop stream the metric array?
// build: hipcc -O3 --offload-arch=gfx950 mfma64_proto.hip -o mfma64_proto ; run: ./mfma64_proto [n_elem]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NB = 8, NQ = 9;
typedef double d4 __attribute__((ext_vector_type(4)));

// G layout: [batch of 16 elements][r][c][q (NQ)][16 e]
__global__ void __launch_bounds__(64, 2) proto_kernel(int n_batches, const double *__restrict__ G, const double *__restrict__ P,
                                                      const double *__restrict__ D, double *__restrict__ out)
{
    const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
    for (int batch = blockIdx.x; batch < n_batches; batch += gridDim.x)
    {
        // u(k, l) for k in {g, g + 4}: synthetic values
        double U[2][NB], OUT[4][NB];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int l = 0; l < NB; ++l)
                U[s][l] = 1e-3 * (1 + lane + 64 * (s + 2 * l)) + batch * 1e-9;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = 0; l < NB; ++l)
                OUT[j][l] = 0.0;
        // A operands: forward  A[i = q][k' = g + 4 s] = D(q, k') / P(q, k'),  backward A'[i = k][kappa = (s', g)] = D(q = 4 g + s', k)
        double AfD[2], AfP[2], AbD[4], AbP[4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
        {
            const int q = e, kp = g + 4 * s; // this lane supplies row i = lane % 16 = e of the A operand
            AfD[s] = q < NQ ? D[q + NQ * kp] : 0.0;
            AfP[s] = q < NQ ? P[q + NQ * kp] : 0.0;
        }
#pragma unroll
        for (int sp = 0; sp < 4; ++sp)
        {
            const int k = e, q = 4 * g + sp;
            AbD[sp] = (k < NB && q < NQ) ? D[q + NQ * k] : 0.0;
            AbP[sp] = (k < NB && q < NQ) ? P[q + NQ * k] : 0.0;
        }
        const double *Gb = G + (size_t)batch * NQ * 3 * NQ * 16;
#pragma unroll 1
        for (int r = 0; r < NQ; ++r)
        {
            // metric values of my 4 quadrature rows q = 4 g + j at (r, e)
            double ga[4], gb[4], gc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                const int q = 4 * g + j;
                const bool ok = q < NQ;
                const size_t o = (((size_t)r * 3) * NQ + (ok ? q : 0)) * 16 + e;
                ga[j] = ok ? __builtin_nontemporal_load(&Gb[o]) : 0.0;
                gb[j] = ok ? __builtin_nontemporal_load(&Gb[o + (size_t)NQ * 16]) : 0.0;
                gc[j] = ok ? __builtin_nontemporal_load(&Gb[o + (size_t)2 * NQ * 16]) : 0.0;
            }
            // in-lane eta contraction
            double pl[2], dl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    a += P[r + NQ * l] * U[s][l];
                    b += D[r + NQ * l] * U[s][l];
                }
                pl[s] = a;
                dl[s] = b;
            }
            d4 dx = {0, 0, 0, 0}, dy = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
                dx = __builtin_amdgcn_mfma_f64_16x16x4f64(AfD[s], pl[s], dx, 0, 0, 0);
                dy = __builtin_amdgcn_mfma_f64_16x16x4f64(AfP[s], dl[s], dy, 0, 0, 0);
            }
            double F0[4], F1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {
                F0[j] = ga[j] * dx[j] + gb[j] * dy[j];
                F1[j] = gb[j] * dx[j] + gc[j] * dy[j];
            }
            d4 W0 = {0, 0, 0, 0}, W1 = {0, 0, 0, 0};
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
            {
                W0 = __builtin_amdgcn_mfma_f64_16x16x4f64(AbD[sp], F0[sp], W0, 0, 0, 0);
                W1 = __builtin_amdgcn_mfma_f64_16x16x4f64(AbP[sp], F1[sp], W1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    OUT[j][l] += P[r + NQ * l] * W0[j] + D[r + NQ * l] * W1[j];
        }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int l = 0; l < NB; ++l)
                s += OUT[j][l];
        out[(size_t)batch * 64 + lane] = s;
    }
}

int main(int argc, char **argv)
{
    const int n_elem = argc > 1 ? std::atoi(argv[1]) : 589824; // 768^2
    const int n_batches = n_elem / 16;
    const size_t nG = (size_t)n_batches * NQ * 3 * NQ * 16;
    double *G, *P, *D, *out;
    hipMalloc(&G, nG * sizeof(double));
    hipMalloc(&P, NQ * NB * sizeof(double));
    hipMalloc(&D, NQ * NB * sizeof(double));
    hipMalloc(&out, (size_t)n_batches * 64 * sizeof(double));
    std::vector<double> h(NQ * NB);
    for (int i = 0; i < NQ * NB; ++i)
        h[i] = 0.01 * (i % 7) - 0.02;
    hipMemcpy(P, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(D, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(G, 0, nG * sizeof(double));
    for (int occ : {1, 2})
    {
        const int blocks = 256 * 4 * occ * 4; // several waves' worth of batches per resident wave
        hipLaunchKernelGGL(proto_kernel, dim3(blocks), dim3(64), 0, 0, n_batches, G, P, D, out);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i)
            hipLaunchKernelGGL(proto_kernel, dim3(blocks), dim3(64), 0, 0, n_batches, G, P, D, out);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double t = ms * 1e-3 / 10;
        std::printf("n_elem %d grid %d: %.1f us per pass, metric stream %.1f GB/s (%.1f MB), %.0f cycles per element and SIMD\n", n_elem, blocks,
                    t * 1e6, nG * 8.0 / t / 1e9, nG * 8.0 / 1e6, t * 2.4e9 * 1024 / n_elem);
    }
    return 0;
}
