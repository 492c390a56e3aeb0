// DDH local solves (WaveHoltz time stepping of every subdomain) for gfx950.
//
// What is computed is the algorithm of reference source/DDH.cpp:111-321; how it
// is computed is different:
//   * the stiffness assembly is an owner gather in a fixed order, not float
//     atomics into LDS (source/DDH.cpp:108), so results are bitwise reproducible;
//   * `ddh_wave_kernel` (n_basis == 4, 4x4 elements) runs one subdomain per
//     WAVEFRONT with the whole state in registers: lane = (element, xi-node),
//     each lane owns the four eta-nodes of its column.  eta-direction
//     contractions are in-lane FMAs with scalar-register coefficients,
//     xi-direction contractions read the other three lanes of the quad through
//     DPP quad_perm operands, element-to-element assembly uses DPP row shifts
//     (xi neighbours) and one ds_bpermute pair (eta neighbours).  No LDS
//     storage, no barriers, 25,600 time steps per launch;
//   * `ddh_block_kernel` (any n_basis <= 10) runs one subdomain per workgroup
//     with the field staged in LDS, 3 barriers per stiffness sweep.
#include <algorithm>
#include <cmath>

#include "common.hpp"

using namespace cuddh_k;

struct cuddh_ddh_plan
{
    cuddh_ddh_desc d;
    int is_f64;
    int kernel; // 1 block, 2 wave, 3 wave with hand-folded DPP FMAs (fp32), 4 = 3 + MFMA for the in-lane contractions,
                // 5 dense element matrix on the matrix cores (fp32, uniform geometry)
    int nodes;  // nb*nb*nel1d*nel1d
    int wh_iters = 5; // WaveHoltz iterations per local solve (source/DDH.cpp:136); WH_ITERS_REFERENCE
    const int *gI_override = nullptr; // cuddh_hip_ddh_plan_set_vector_layout: x and y in another numbering than d.gI
    int g_ndof_override = 0;
    float *Aop = nullptr; // kernel 5: element stiffness matrix as MFMA A operands, [4 k-steps][64 lanes]
    float *Sep = nullptr; // kernel 7: [Ax | Ay | beta | gamma] of the separable nb = 8 sweep
    int wave_priority = 0; // cuddh_hip_ddh_plan_set_wave_priority
};

namespace
{
    constexpr int WH_ITERS_REFERENCE = 5; // WaveHoltz iterations of the reference (source/DDH.cpp:136)

    template <typename Real>
    struct DdhArgs
    {
        int g_ndof, n_lambda, nt, mx_dof, mx_fdof, nodes, dom_begin, dom_end;
        int wh_iters; // 5 unless a verification run changed it (cuddh_hip_ddh_plan_set_wh_iters)
        int prio;     // != 0: these wavefronts take issue priority over others on their SIMD (cuddh_hip_ddh_plan_set_wave_priority)
        Real omega, dt;
        const int *s_dof, *s_fdof, *B, *gI, *sI;
        const Real *G, *m, *gmi, *a, *H;
        const double *x;
        double *y;
        const Real *lambda;
        Real *update;
        const int *dom_list; // null: positions [dom_begin, dom_end) ARE the subdomains; else subdomain = dom_list[position]
    };

    template <typename Real>
    __device__ inline int domain_at(const DdhArgs<Real> &A, int position)
    {
        return A.dom_list ? A.dom_list[position] : position;
    }

    // Issue priority of the calling wavefront (s_setprio).  All wavefronts of one rank's local solves are resident at once and
    // advance at the same rate, so a launch of the few subdomains whose traces other ranks wait for finishes no earlier than
    // the launch of all the others it shares the SIMDs with (profiles/r02/overlap_timeline.txt) -- unless its wavefronts are
    // issued first whenever they are ready.
    __device__ inline void raise_priority(int prio)
    {
        if (prio)
            __builtin_amdgcn_s_setprio(3);
    }

    // ---------------------------------------------------------------- DPP helpers
    template <int CTRL>
    __device__ inline float dpp_read(float v)
    {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
    }

    template <int CTRL>
    __device__ inline double dpp_read(double v)
    {
        const long long bits = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), CTRL, 0xf, 0xf, true);
        return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
    }

    constexpr int QUAD_BCAST(int i) { return i * 0x55; } // quad_perm:[i,i,i,i]
    constexpr int ROW_SHL1 = 0x101;                      // lane j reads lane j+1 of its 16-lane row
    constexpr int ROW_SHR1 = 0x111;                      // lane j reads lane j-1

    // ---------------------------------------------------------------- wavefront-per-subdomain kernel (NB = 4, 4x4 elements)

    // out[l] = sum_i c[i] * (value of in[l] in lane i of my quad), l = 0..3.
    // Generic form: the compiler emits v_mov_b32_dpp + FMA pairs.
    template <typename Real>
    __device__ inline void quad_contract(const Real (&in)[4], const Real (&c)[4], Real (&out)[4])
    {
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            Real s = c[0] * dpp_read<QUAD_BCAST(0)>(in[l]);
            s += c[1] * dpp_read<QUAD_BCAST(1)>(in[l]);
            s += c[2] * dpp_read<QUAD_BCAST(2)>(in[l]);
            s += c[3] * dpp_read<QUAD_BCAST(3)>(in[l]);
            out[l] = s;
        }
    }

    // fp32 form with the cross-lane read folded into the FMA (v_fmac_f32_dpp): 16 VALU instructions
    // instead of 32.  hipcc does not fold a DPP move into an accumulating FMA by itself.  The leading
    // s_nop covers the "VALU write -> DPP read" hazard (2 wait states) for the inputs; inside the block
    // DPP operands are only the (unmodified) inputs.
#define CUDDH_QP(i) " quad_perm:[" #i "," #i "," #i "," #i "] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    __device__ inline void quad_contract_asm(const float (&in)[4], const float (&c)[4], float (&out)[4])
    {
        float o0, o1, o2, o3;
        asm volatile("s_nop 1\n\t"
                     "v_mul_f32_dpp %0, %4, %8" CUDDH_QP(0)
                     "v_mul_f32_dpp %1, %5, %8" CUDDH_QP(0)
                     "v_mul_f32_dpp %2, %6, %8" CUDDH_QP(0)
                     "v_mul_f32_dpp %3, %7, %8" CUDDH_QP(0)
                     "v_fmac_f32_dpp %0, %4, %9" CUDDH_QP(1)
                     "v_fmac_f32_dpp %1, %5, %9" CUDDH_QP(1)
                     "v_fmac_f32_dpp %2, %6, %9" CUDDH_QP(1)
                     "v_fmac_f32_dpp %3, %7, %9" CUDDH_QP(1)
                     "v_fmac_f32_dpp %0, %4, %10" CUDDH_QP(2)
                     "v_fmac_f32_dpp %1, %5, %10" CUDDH_QP(2)
                     "v_fmac_f32_dpp %2, %6, %10" CUDDH_QP(2)
                     "v_fmac_f32_dpp %3, %7, %10" CUDDH_QP(2)
                     "v_fmac_f32_dpp %0, %4, %11" CUDDH_QP(3)
                     "v_fmac_f32_dpp %1, %5, %11" CUDDH_QP(3)
                     "v_fmac_f32_dpp %2, %6, %11" CUDDH_QP(3)
                     "v_fmac_f32_dpp %3, %7, %11" CUDDH_QP(3)
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                     : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
        out[0] = o0;
        out[1] = o1;
        out[2] = o2;
        out[3] = o3;
    }

    // r[l] = mR * (z[l] of lane+1) + mL * (z[l] of lane-1) within the 16-lane row, fp32, 8 instructions
    __device__ inline void row_neighbours_asm(const float (&z)[4], float mR, float mL, float (&r)[4])
    {
        float o0, o1, o2, o3;
        asm volatile("s_nop 1\n\t"
                     "v_mul_f32_dpp %0, %4, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_mul_f32_dpp %1, %5, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_mul_f32_dpp %2, %6, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_mul_f32_dpp %3, %7, %8 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %0, %4, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %5, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %2, %6, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %3, %7, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                     : "v"(z[0]), "v"(z[1]), "v"(z[2]), "v"(z[3]), "v"(mR), "v"(mL));
        r[0] = o0;
        r[1] = o1;
        r[2] = o2;
        r[3] = o3;
    }
#undef CUDDH_QP

    // VAR 0: plain HIP; 1: DPP reads folded into the FMAs (fp32); 2: as 1, and the in-lane (eta) contractions run on the
    // matrix pipe as v_mfma_f32_4x4x1_16b_f32 (one 4x4 block per element, exact fp32 FMA chains), beside the VALU
    template <int VAR, typename Real>
    __device__ inline void wave_stiffness(const Real (&w)[4], Real (&z)[4], const Real (&gx)[4], const Real (&gy)[4], const Real (&gz)[4],
                                          const Real (&Dk)[4], const Real (&DTk)[4], const Real *__restrict__ Dm, Real mR, Real mL,
                                          Real mU, Real mD, int lane)
    {
        constexpr bool ASM = VAR >= 1;
        Real ux[4], uy[4];
        // eta derivative uy(k,l) = sum_i D(l,i) u(k,i): in-lane
        if constexpr (VAR == 2)
        {
            // block = element, output row = l (register), column = k (lane): A_b[l] = D(l,i) is lane k==l's Dk[i],
            // B_b[k] = u(k,i) is my w[i]
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Dk[i], w[i], acc, 0, 0, 0);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                uy[l] = acc[l];
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 4; ++l)
            {
                Real s = Dm[l] * w[0];
#pragma unroll
                for (int i = 1; i < 4; ++i)
                    s += Dm[l + 4 * i] * w[i];
                uy[l] = s;
            }
        }
        // xi derivative: u(i, l) sits in lane i of my quad
        if constexpr (ASM)
            quad_contract_asm(w, Dk, ux);
        else
            quad_contract(w, Dk, ux);
        Real f1[4], f2[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            f1[l] = gx[l] * ux[l] + gy[l] * uy[l];
            f2[l] = gy[l] * ux[l] + gz[l] * uy[l];
        }
        // test functions: sum_i D(i,k) f1(i,l)  +  sum_i D(i,l) f2(k,i)
        Real zz[4];
        if constexpr (ASM)
            quad_contract_asm(f1, DTk, zz);
        else
            quad_contract(f1, DTk, zz);
        if constexpr (VAR == 2)
        {
            // A_b[l] = D(i,l) is lane k==l's DTk[i], B_b[k] = f2(k,i) is my f2[i]; accumulate onto the xi part
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc = {zz[0], zz[1], zz[2], zz[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(DTk[i], f2[i], acc, 0, 0, 0);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                zz[l] = acc[l];
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 4; ++l)
            {
                Real s = zz[l];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    s += Dm[i + 4 * l] * f2[i];
                zz[l] = s;
            }
        }
        // assembly across elements.  xi neighbours: my k==3 column meets the k==0 column of lane+1 (and vice versa)
        if constexpr (ASM)
        {
            Real r[4];
            row_neighbours_asm(zz, mR, mL, r);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                z[l] = zz[l] + r[l];
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 4; ++l)
            {
                const Real from_right = dpp_read<ROW_SHL1>(zz[l]);
                const Real from_left = dpp_read<ROW_SHR1>(zz[l]);
                z[l] = zz[l] + (mR * from_right + mL * from_left);
            }
        }
        // eta neighbours: my l==3 node meets the l==0 node of lane+16 (and vice versa)
        {
            const Real top = z[3], bottom = z[0];
            const Real from_above = __shfl(bottom, (lane + 16) & 63, 64);
            const Real from_below = __shfl(top, (lane + 48) & 63, 64);
            z[3] = top + mU * from_above;
            z[0] = bottom + mD * from_below;
        }
    }

    template <typename Real, int VAR>
    __global__ void __launch_bounds__(256) ddh_wave_kernel(DdhArgs<Real> A, const Real *__restrict__ Dmat, const Real *__restrict__ filt,
                                                          const Real *__restrict__ cs, const Real *__restrict__ sn)
    {
        const int lane = threadIdx.x & 63;
        const int position = A.dom_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
        if (position >= A.dom_end)
            return; // wave-uniform: the kernel has no barriers
        const int s = domain_at(A, position);
        raise_priority(A.prio);

        const int k = lane & 3, el = lane >> 2, ex = el & 3, ey = el >> 2;
        const int fdof = A.s_fdof[s];
        const int *sI = A.sI + 256 * (size_t)s;
        const size_t dbase = (size_t)A.mx_dof * s, fbase = (size_t)A.mx_fdof * s;

        Real gx[4], gy[4], gz[4], invm[4], Hi[4], F[4], Gf[4];
        Real p[4], q[4], u[4], v[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            const int node = k + 4 * (l + 4 * el);
            const int d = sI[node];
            const Real *g = A.G + 3 * ((size_t)node + 256 * (size_t)s);
            gx[l] = g[0];
            gy[l] = g[1];
            gz[l] = g[2];
            const Real ai = A.a[dbase + d], mi = A.m[dbase + d];
            invm[l] = Real(1) / (ai * ai * mi);
            Real f = 0, gg = 0, h = 0;
            if (A.x)
            {
                const int gidx = A.gI[dbase + d];
                f = static_cast<Real>(A.x[gidx]);
                gg = static_cast<Real>(A.x[A.g_ndof + gidx]);
            }
            if (d < fdof)
            {
                h = A.H[fbase + d];
                if (A.lambda)
                {
                    const int slot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                    if (slot >= 0)
                    {
                        f += h * A.lambda[slot];
                        gg += h * A.lambda[A.n_lambda + slot];
                    }
                }
                h *= ai;
            }
            F[l] = f;
            Gf[l] = gg;
            Hi[l] = h;
            p[l] = q[l] = u[l] = v[l] = 0;
        }

        Real Dk[4], DTk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            Dk[i] = Dmat[k + 4 * i];  // D(k, i)
            DTk[i] = Dmat[i + 4 * k]; // D(i, k)
        }
        const Real mR = (k == 3 && ex < 3) ? Real(1) : Real(0);
        const Real mL = (k == 0 && ex > 0) ? Real(1) : Real(0);
        const Real mU = (ey < 3) ? Real(1) : Real(0);
        const Real mD = (ey > 0) ? Real(1) : Real(0);

        const Real dt = A.dt, half_dt = Real(0.5) * A.dt;
        const int nt = A.nt;

        for (int whit = 0; whit < A.wh_iters; ++whit)
        {
            {
                const Real k0 = filt[0];
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    p[l] = u[l];
                    q[l] = v[l];
                    u[l] *= k0;
                    v[l] *= k0;
                }
            }
            for (int it = 1; it <= nt; ++it)
            {
                const Real c0 = cs[2 * it - 2], s0 = sn[2 * it - 2];
                const Real c1 = cs[2 * it - 1], s1 = sn[2 * it - 1];
                const Real kw = filt[it];
                Real z[4], ph[4], qh[4];

                wave_stiffness<VAR>(p, z, gx, gy, gz, Dk, DTk, Dmat, mR, mL, mU, mD, lane);
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    const Real dq = ((z[l] - Hi[l] * q[l]) + c0 * F[l] + s0 * Gf[l]) * invm[l];
                    ph[l] = p[l] - half_dt * q[l];
                    qh[l] = q[l] + half_dt * dq;
                    p[l] -= dt * qh[l];
                }
                wave_stiffness<VAR>(ph, z, gx, gy, gz, Dk, DTk, Dmat, mR, mL, mU, mD, lane);
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    const Real dq = ((z[l] - Hi[l] * qh[l]) + c1 * F[l] + s1 * Gf[l]) * invm[l];
                    q[l] += dt * dq;
                    u[l] += kw * p[l];
                    v[l] += kw * q[l];
                }
            }
        }

        const Real rw = Real(1) / A.omega;
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            v[l] *= rw;
            // every shared node is held by 2 or 4 (lane, l) pairs with identical values: the copy with
            // the smallest element-node index writes
            const bool owner = !(k == 0 && ex > 0) && !(l == 0 && ey > 0);
            if (!owner)
                continue;
            const int d = sI[k + 4 * (l + 4 * el)];
            if (A.y)
            {
                const int gidx = A.gI[dbase + d];
                const Real M = A.m[dbase + d] * A.gmi[dbase + d];
                atomic_add(A.y + gidx, static_cast<double>(M * u[l]));
                atomic_add(A.y + A.g_ndof + gidx, static_cast<double>(M * v[l]));
            }
            if (A.update && d < fdof)
            {
                const int wslot = A.B[d + (size_t)A.mx_fdof * (1 + 2 * (size_t)s)];
                if (wslot >= 0)
                {
                    Real lam = 0, mu = 0;
                    if (A.lambda)
                    {
                        const int rslot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                        if (rslot >= 0)
                        {
                            lam = A.lambda[rslot];
                            mu = A.lambda[A.n_lambda + rslot];
                        }
                    }
                    const Real S = Real(2) * A.a[dbase + d] * A.omega;
                    A.update[wslot] = -lam - S * v[l];
                    A.update[A.n_lambda + wslot] = -mu + S * u[l];
                }
            }
        }
    }

    // ---------------------------------------------------------------- wavefront per TWO subdomains (NB = 8, 2x2 elements)
    // The reference's other supported shape (source/DDH.cpp:631-636).  lane = k + 8 * element + 32 * (subdomain of the
    // pair); every lane owns the eight eta-nodes of its column, so eta contractions are 8x8 in-lane FMAs with scalar
    // coefficients.  The xi contraction runs over the eight lanes of an octet: partner k^x for x = 1,2,3 is a DPP
    // quad_perm operand, k^7 is row_half_mirror, and k^4, k^5, k^6 are quad_perms of the half-mirrored copy; the
    // coefficient of partner x is the per-lane register D(k, k^x).  xi neighbours (ex 0|1) sit in adjacent lanes 7|8 of a
    // 16-lane row (DPP row shifts), eta neighbours (ey 0|1) 16 lanes apart (one ds_bpermute pair per sweep).
    constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E, QP_XOR3 = 0x1B, ROW_HALF_MIRROR = 0x141;

    // out[l] = sum_x c[x] * (in[l] of lane k^x of my octet)
    template <typename Real>
    __device__ inline void octet_contract(const Real (&in)[8], const Real (&c)[8], Real (&out)[8])
    {
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            const Real t = dpp_read<ROW_HALF_MIRROR>(in[l]);
            Real s = c[0] * in[l];
            s += c[1] * dpp_read<QP_XOR1>(in[l]);
            s += c[2] * dpp_read<QP_XOR2>(in[l]);
            s += c[3] * dpp_read<QP_XOR3>(in[l]);
            s += c[7] * t;
            s += c[6] * dpp_read<QP_XOR1>(t);
            s += c[5] * dpp_read<QP_XOR2>(t);
            s += c[4] * dpp_read<QP_XOR3>(t);
            out[l] = s;
        }
    }

    // fp32 form with the cross-lane reads folded into the FMAs: 36 VALU instructions per four nodes.  The half-mirrored
    // copies are written first and read through DPP at least four instructions later (the hazard needs two).
#define CUDDH_DPP_TAIL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CUDDH_X1 " quad_perm:[1,0,3,2]" CUDDH_DPP_TAIL
#define CUDDH_X2 " quad_perm:[2,3,0,1]" CUDDH_DPP_TAIL
#define CUDDH_X3 " quad_perm:[3,2,1,0]" CUDDH_DPP_TAIL
    __device__ inline void octet_contract4_asm(const float *in, const float (&c)[8], float *out)
    {
        float o0, o1, o2, o3, t0, t1, t2, t3;
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %4, %8 row_half_mirror" CUDDH_DPP_TAIL
                     "v_mov_b32_dpp %5, %9 row_half_mirror" CUDDH_DPP_TAIL
                     "v_mov_b32_dpp %6, %10 row_half_mirror" CUDDH_DPP_TAIL
                     "v_mov_b32_dpp %7, %11 row_half_mirror" CUDDH_DPP_TAIL
                     "v_mul_f32 %0, %8, %12\n\t"
                     "v_mul_f32 %1, %9, %12\n\t"
                     "v_mul_f32 %2, %10, %12\n\t"
                     "v_mul_f32 %3, %11, %12\n\t"
                     "v_fmac_f32_dpp %0, %8, %13" CUDDH_X1
                     "v_fmac_f32_dpp %1, %9, %13" CUDDH_X1
                     "v_fmac_f32_dpp %2, %10, %13" CUDDH_X1
                     "v_fmac_f32_dpp %3, %11, %13" CUDDH_X1
                     "v_fmac_f32_dpp %0, %8, %14" CUDDH_X2
                     "v_fmac_f32_dpp %1, %9, %14" CUDDH_X2
                     "v_fmac_f32_dpp %2, %10, %14" CUDDH_X2
                     "v_fmac_f32_dpp %3, %11, %14" CUDDH_X2
                     "v_fmac_f32_dpp %0, %8, %15" CUDDH_X3
                     "v_fmac_f32_dpp %1, %9, %15" CUDDH_X3
                     "v_fmac_f32_dpp %2, %10, %15" CUDDH_X3
                     "v_fmac_f32_dpp %3, %11, %15" CUDDH_X3
                     "v_fmac_f32 %0, %4, %19\n\t"
                     "v_fmac_f32 %1, %5, %19\n\t"
                     "v_fmac_f32 %2, %6, %19\n\t"
                     "v_fmac_f32 %3, %7, %19\n\t"
                     "v_fmac_f32_dpp %0, %4, %18" CUDDH_X1
                     "v_fmac_f32_dpp %1, %5, %18" CUDDH_X1
                     "v_fmac_f32_dpp %2, %6, %18" CUDDH_X1
                     "v_fmac_f32_dpp %3, %7, %18" CUDDH_X1
                     "v_fmac_f32_dpp %0, %4, %17" CUDDH_X2
                     "v_fmac_f32_dpp %1, %5, %17" CUDDH_X2
                     "v_fmac_f32_dpp %2, %6, %17" CUDDH_X2
                     "v_fmac_f32_dpp %3, %7, %17" CUDDH_X2
                     "v_fmac_f32_dpp %0, %4, %16" CUDDH_X3
                     "v_fmac_f32_dpp %1, %5, %16" CUDDH_X3
                     "v_fmac_f32_dpp %2, %6, %16" CUDDH_X3
                     "v_fmac_f32_dpp %3, %7, %16" CUDDH_X3
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                     : "v"(in[0]), "v"(in[1]), "v"(in[2]), "v"(in[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]),
                       "v"(c[6]), "v"(c[7]));
        out[0] = o0;
        out[1] = o1;
        out[2] = o2;
        out[3] = o3;
    }
#undef CUDDH_X1
#undef CUDDH_X2
#undef CUDDH_X3
#undef CUDDH_DPP_TAIL

    template <bool ASM, typename Real>
    __device__ inline void octet_contract_any(const Real (&in)[8], const Real (&c)[8], Real (&out)[8])
    {
        if constexpr (ASM && sizeof(Real) == 4)
        {
            octet_contract4_asm(&in[0], c, &out[0]);
            octet_contract4_asm(&in[4], c, &out[4]);
        }
        else
            octet_contract(in, c, out);
    }

    template <bool ASM, typename Real>
    __device__ inline void wave8_stiffness(const Real (&w)[8], Real (&z)[8], const Real (&gx)[8], const Real (&gy)[8], const Real (&gz)[8],
                                           const Real (&Dk)[8], const Real (&DTk)[8], const Real *__restrict__ Dm, Real mR, Real mL,
                                           Real mU, Real mD, int lane)
    {
        Real ux[8], uy[8];
        // eta derivative uy(k,l) = sum_i D(l,i) u(k,i): in-lane, scalar coefficients
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            Real s = Dm[l] * w[0];
#pragma unroll
            for (int i = 1; i < 8; ++i)
                s += Dm[l + 8 * i] * w[i];
            uy[l] = s;
        }
        // xi derivative ux(k,l) = sum_i D(k,i) u(i,l): over the octet
        octet_contract_any<ASM>(w, Dk, ux);
        Real f1[8], f2[8];
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            f1[l] = gx[l] * ux[l] + gy[l] * uy[l];
            f2[l] = gy[l] * ux[l] + gz[l] * uy[l];
        }
        // test functions: sum_i D(i,k) f1(i,l)  +  sum_i D(i,l) f2(k,i)
        Real zz[8];
        octet_contract_any<ASM>(f1, DTk, zz);
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            Real s = zz[l];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                s += Dm[i + 8 * l] * f2[i];
            zz[l] = s;
        }
        // assembly across elements.  xi neighbours: column k==7 of ex==0 meets column k==0 of ex==1 (lanes 7|8 of a row)
        if constexpr (ASM && sizeof(Real) == 4)
        {
            const float(&za)[4] = reinterpret_cast<const float(&)[4]>(zz[0]);
            const float(&zb)[4] = reinterpret_cast<const float(&)[4]>(zz[4]);
            float ra[4], rb[4];
            row_neighbours_asm(za, mR, mL, ra);
            row_neighbours_asm(zb, mR, mL, rb);
#pragma unroll
            for (int l = 0; l < 4; ++l)
            {
                z[l] = zz[l] + ra[l];
                z[4 + l] = zz[4 + l] + rb[l];
            }
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 8; ++l)
            {
                const Real from_right = dpp_read<ROW_SHL1>(zz[l]);
                const Real from_left = dpp_read<ROW_SHR1>(zz[l]);
                z[l] = zz[l] + (mR * from_right + mL * from_left);
            }
        }
        // eta neighbours: node l==7 of ey==0 meets node l==0 of ey==1, 16 lanes up
        {
            const Real top = z[7], bottom = z[0];
            const Real from_above = __shfl(bottom, (lane + 16) & 63, 64);
            const Real from_below = __shfl(top, (lane + 48) & 63, 64);
            z[7] = top + mU * from_above;
            z[0] = bottom + mD * from_below;
        }
    }

    // Separable form (kernel 7).  On the rectangles of a uniform_rect mesh the metric is diagonal and a product of 1-D
    // factors, gx(k,l) = alpha_k beta_l, gz(k,l) = gamma_k delta_l, gy = 0, so the sweep  D^T (G (D u))  collapses to
    //     z(k,l) = beta_l * sum_j Ax(k,j) u(j,l)  +  gamma_k * sum_j Ay(l,j) u(k,j),
    // Ax = D^T diag(alpha) D, Ay = D^T diag(delta) D (8 x 8, built in double when the plan is created): ONE contraction
    // per direction instead of two.  Sep = [Ax (k + 8 j): 64 | Ay, packed upper triangle: 36 (+ 28 unused) | beta: 8 | gamma: 8].
    __host__ __device__ constexpr int sym_index(int a, int b) { return a <= b ? a + b * (b + 1) / 2 : b + a * (a + 1) / 2; }

    template <bool ASM, typename Real>
    __device__ inline void wave8_stiffness_sep(const Real (&w)[8], Real (&z)[8], const Real (&AxK)[8], const Real *__restrict__ Sep,
                                               Real gamma, Real mR, Real mL, Real mU, Real mD, int lane)
    {
        // Ay is symmetric: its 36 distinct entries are stored packed (upper triangle, column by column) so that they and
        // beta can stay in scalar registers for the whole time loop
        const Real *Ay = Sep + 64, *beta = Sep + 128;
        Real X[8], zz[8];
        octet_contract_any<ASM>(w, AxK, X);
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            Real s = Ay[sym_index(l, 0)] * w[0];
#pragma unroll
            for (int j = 1; j < 8; ++j)
                s += Ay[sym_index(l, j)] * w[j];
            zz[l] = gamma * s + beta[l] * X[l];
        }
        if constexpr (ASM && sizeof(Real) == 4)
        {
            const float(&za)[4] = reinterpret_cast<const float(&)[4]>(zz[0]);
            const float(&zb)[4] = reinterpret_cast<const float(&)[4]>(zz[4]);
            float ra[4], rb[4];
            row_neighbours_asm(za, mR, mL, ra);
            row_neighbours_asm(zb, mR, mL, rb);
#pragma unroll
            for (int l = 0; l < 4; ++l)
            {
                z[l] = zz[l] + ra[l];
                z[4 + l] = zz[4 + l] + rb[l];
            }
        }
        else
        {
#pragma unroll
            for (int l = 0; l < 8; ++l)
            {
                const Real from_right = dpp_read<ROW_SHL1>(zz[l]);
                const Real from_left = dpp_read<ROW_SHR1>(zz[l]);
                z[l] = zz[l] + (mR * from_right + mL * from_left);
            }
        }
        {
            const Real top = z[7], bottom = z[0];
            const Real from_above = __shfl(bottom, (lane + 16) & 63, 64);
            const Real from_below = __shfl(top, (lane + 48) & 63, 64);
            z[7] = top + mU * from_above;
            z[0] = bottom + mD * from_below;
        }
    }

    template <typename Real, bool ASM, bool SEP>
    __global__ void __launch_bounds__(256) ddh_wave8_kernel(DdhArgs<Real> A, const Real *__restrict__ Dmat, const Real *__restrict__ filt,
                                                           const Real *__restrict__ cs, const Real *__restrict__ sn,
                                                           const Real *__restrict__ Sep)
    {
        raise_priority(A.prio);
        const int lane = threadIdx.x & 63;
        const int s_first = A.dom_begin + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        if (s_first >= A.dom_end)
            return; // wave-uniform: the kernel has no barriers
        const int sub = lane >> 5;
        const bool valid = s_first + sub < A.dom_end;
        const int s = domain_at(A, valid ? s_first + sub : s_first); // a missing second subdomain recomputes the first, writes nothing

        const int k = lane & 7, el = (lane >> 3) & 3, ex = el & 1, ey = el >> 1;
        const int fdof = A.s_fdof[s];
        const int *sI = A.sI + 256 * (size_t)s;
        const size_t dbase = (size_t)A.mx_dof * s, fbase = (size_t)A.mx_fdof * s;

        Real gx[8], gy[8], gz[8], invm[8], Hi[8], F[8], Gf[8];
        Real p[8], q[8], u[8], v[8];
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            const int node = k + 8 * (l + 8 * el);
            const int d = sI[node];
            if constexpr (!SEP)
            {
                const Real *g = A.G + 3 * ((size_t)node + 256 * (size_t)s);
                gx[l] = g[0];
                gy[l] = g[1];
                gz[l] = g[2];
            }
            else
                gx[l] = gy[l] = gz[l] = 0;
            const Real ai = A.a[dbase + d], mi = A.m[dbase + d];
            invm[l] = Real(1) / (ai * ai * mi);
            Real f = 0, gg = 0, h = 0;
            if (A.x)
            {
                const int gidx = A.gI[dbase + d];
                f = static_cast<Real>(A.x[gidx]);
                gg = static_cast<Real>(A.x[A.g_ndof + gidx]);
            }
            if (d < fdof)
            {
                h = A.H[fbase + d];
                if (A.lambda)
                {
                    const int slot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                    if (slot >= 0)
                    {
                        f += h * A.lambda[slot];
                        gg += h * A.lambda[A.n_lambda + slot];
                    }
                }
                h *= ai;
            }
            F[l] = f;
            Gf[l] = gg;
            Hi[l] = h;
            p[l] = q[l] = u[l] = v[l] = 0;
        }

        Real Dk[8], DTk[8];
        Real gamma = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x)
        {
            if constexpr (SEP)
            {
                Dk[x] = Sep[k + 8 * (k ^ x)]; // Ax(k, k^x)
                DTk[x] = 0;
            }
            else
            {
                Dk[x] = Dmat[k + 8 * (k ^ x)];  // D(k, k^x)
                DTk[x] = Dmat[(k ^ x) + 8 * k]; // D(k^x, k)
            }
        }
        if constexpr (SEP)
            gamma = Sep[136 + k];
        const Real mR = (k == 7 && ex == 0) ? Real(1) : Real(0);
        const Real mL = (k == 0 && ex == 1) ? Real(1) : Real(0);
        const Real mU = (ey == 0) ? Real(1) : Real(0);
        const Real mD = (ey == 1) ? Real(1) : Real(0);

        const Real dt = A.dt, half_dt = Real(0.5) * A.dt;
        const int nt = A.nt;

        for (int whit = 0; whit < A.wh_iters; ++whit)
        {
            {
                const Real k0 = filt[0];
#pragma unroll
                for (int l = 0; l < 8; ++l)
                {
                    p[l] = u[l];
                    q[l] = v[l];
                    u[l] *= k0;
                    v[l] *= k0;
                }
            }
            for (int it = 1; it <= nt; ++it)
            {
                const Real c0 = cs[2 * it - 2], s0 = sn[2 * it - 2];
                const Real c1 = cs[2 * it - 1], s1 = sn[2 * it - 1];
                const Real kw = filt[it];
                Real z[8], ph[8], qh[8];

                if constexpr (SEP)
                    wave8_stiffness_sep<ASM>(p, z, Dk, Sep, gamma, mR, mL, mU, mD, lane);
                else
                    wave8_stiffness<ASM>(p, z, gx, gy, gz, Dk, DTk, Dmat, mR, mL, mU, mD, lane);
#pragma unroll
                for (int l = 0; l < 8; ++l)
                {
                    const Real dq = ((z[l] - Hi[l] * q[l]) + c0 * F[l] + s0 * Gf[l]) * invm[l];
                    ph[l] = p[l] - half_dt * q[l];
                    qh[l] = q[l] + half_dt * dq;
                    p[l] -= dt * qh[l];
                }
                if constexpr (SEP)
                    wave8_stiffness_sep<ASM>(ph, z, Dk, Sep, gamma, mR, mL, mU, mD, lane);
                else
                    wave8_stiffness<ASM>(ph, z, gx, gy, gz, Dk, DTk, Dmat, mR, mL, mU, mD, lane);
#pragma unroll
                for (int l = 0; l < 8; ++l)
                {
                    const Real dq = ((z[l] - Hi[l] * qh[l]) + c1 * F[l] + s1 * Gf[l]) * invm[l];
                    q[l] += dt * dq;
                    u[l] += kw * p[l];
                    v[l] += kw * q[l];
                }
            }
        }

        if (!valid)
            return;
        const Real rw = Real(1) / A.omega;
#pragma unroll
        for (int l = 0; l < 8; ++l)
        {
            v[l] *= rw;
            // every shared node is held by 2 or 4 (lane, l) pairs with identical values: the copy with
            // the smallest element-node index writes
            const bool owner = !(k == 0 && ex > 0) && !(l == 0 && ey > 0);
            if (!owner)
                continue;
            const int d = sI[k + 8 * (l + 8 * el)];
            if (A.y)
            {
                const int gidx = A.gI[dbase + d];
                const Real M = A.m[dbase + d] * A.gmi[dbase + d];
                atomic_add(A.y + gidx, static_cast<double>(M * u[l]));
                atomic_add(A.y + A.g_ndof + gidx, static_cast<double>(M * v[l]));
            }
            if (A.update && d < fdof)
            {
                const int wslot = A.B[d + (size_t)A.mx_fdof * (1 + 2 * (size_t)s)];
                if (wslot >= 0)
                {
                    Real lam = 0, mu = 0;
                    if (A.lambda)
                    {
                        const int rslot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                        if (rslot >= 0)
                        {
                            lam = A.lambda[rslot];
                            mu = A.lambda[A.n_lambda + rslot];
                        }
                    }
                    const Real S = Real(2) * A.a[dbase + d] * A.omega;
                    A.update[wslot] = -lam - S * v[l];
                    A.update[A.n_lambda + wslot] = -mu + S * u[l];
                }
            }
        }
    }

    // ---------------------------------------------------------------- dense element matrix on the matrix cores (NB = 4, uniform geometry)
    // When every element of every subdomain has the same metric tensor (always the case on the uniform_rect
    // meshes DDH supports) the element-local part of a stiffness sweep is Z = K U with one 16x16 element matrix K
    // and U = (16 nodes) x (16 elements of the subdomain): exactly one 16x16x16 product, i.e. four
    // v_mfma_f32_16x16x4_f32 per sweep on the matrix pipe, leaving the VALU only the assembly and the RK2 update.
    // Lane = element + 16 * h, register index = l (eta node): the B-operand map B[kk][j] with kk = 4 s + (lane >> 4),
    // j = lane & 15 makes register s of a lane the node (h, l = s); the rows of K are ordered m = 4 h + l so that the
    // C/D map (row = 4 (lane >> 4) + reg) returns the result in the same layout (h = k, the xi node).
    // On gfx950 the f32 MFMA runs at the vector rate and does not overlap with VALU work, so what counts is the
    // number of issue cycles: 4 MFMAs (128 cycles) replace the ~80 VALU instructions (160 cycles + DPP hazards) of the
    // sum-factorised sweep, and the xi-neighbour exchange goes through ds_bpermute (LDS pipe, no VALU slots); a
    // variant with odd elements stored mirrored in xi so that every exchange is a DPP row shift measured 24 % slower
    // (96 instead of 72 VALU instructions per time step).
    __global__ void __launch_bounds__(256) ddh_mfma_kernel(DdhArgs<float> A, const float *__restrict__ Aop, const float *__restrict__ filt,
                                                          const float *__restrict__ cs, const float *__restrict__ sn)
    {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const int lane = threadIdx.x & 63;
        const int position = A.dom_begin + blockIdx.x * 4 + (threadIdx.x >> 6);
        if (position >= A.dom_end)
            return; // wave-uniform, no barriers in this kernel
        const int s = domain_at(A, position);
        raise_priority(A.prio);

        const int k = lane >> 4, el = lane & 15, ex = el & 3, ey = el >> 2;
        const int fdof = A.s_fdof[s];
        const int *sI = A.sI + 256 * (size_t)s;
        const size_t dbase = (size_t)A.mx_dof * s, fbase = (size_t)A.mx_fdof * s;

        float invm[4], Hi[4], F[4], Gf[4], p[4], q[4], u[4], v[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            const int d = sI[k + 4 * (l + 4 * el)];
            const float ai = A.a[dbase + d], mi = A.m[dbase + d];
            invm[l] = 1.0f / (ai * ai * mi);
            float f = 0, gg = 0, h = 0;
            if (A.x)
            {
                const int gidx = A.gI[dbase + d];
                f = static_cast<float>(A.x[gidx]);
                gg = static_cast<float>(A.x[A.g_ndof + gidx]);
            }
            if (d < fdof)
            {
                h = A.H[fbase + d];
                if (A.lambda)
                {
                    const int slot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                    if (slot >= 0)
                    {
                        f += h * A.lambda[slot];
                        gg += h * A.lambda[A.n_lambda + slot];
                    }
                }
                h *= ai;
            }
            F[l] = f;
            Gf[l] = gg;
            Hi[l] = h;
            p[l] = q[l] = u[l] = v[l] = 0;
        }
        float Ka[4];
#pragma unroll
        for (int st = 0; st < 4; ++st)
            Ka[st] = Aop[64 * st + lane];

        // xi neighbours live in another 16-lane row (k = 3 of element ex meets k = 0 of element ex + 1): ds_bpermute;
        // eta neighbours are 4 lanes away in the same row: DPP row_shl/shr:4
        const bool hasR = (k == 3 && ex < 3), hasL = (k == 0 && ex > 0);
        const int partner = hasR ? (el + 1) : (hasL ? (el - 1 + 48) : lane);
        const float mX = (hasR || hasL) ? 1.0f : 0.0f;
        const float mU = (ey < 3) ? 1.0f : 0.0f, mD = (ey > 0) ? 1.0f : 0.0f;

        auto sweep = [&](const float (&w)[4], float (&z)[4])
        {
            f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            // The four matrix instructions of a sweep are issued with priority over the other resident wavefronts' vector work: the
            // fp32 matrix instructions run on the SIMD's own fp32 lanes (no co-execution with VALU: SQ_VALU_MFMA_COEXEC_CYCLES = 0,
            // profiles/r03/pmc_ddh_kernel5_coexec.txt, mfma_valu_coexec.txt) and a vector instruction issued behind a matrix
            // instruction waits for its passes to drain, so grouping the wavefronts' matrix instructions saves issue cycles
            // (same-box A/B, 16,384 subdomains: 85.2 -> 83.4 ms per action; order of issue only, results bitwise unchanged).
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int st = 0; st < 4; ++st)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ka[st], w[st], acc, 0, 0, 0);
            if (!A.prio) // a launch that holds issue priority as a whole (multi-GPU boundary subdomains) keeps it
                __builtin_amdgcn_s_setprio(0);
            float fx[4];
#pragma unroll
            for (int l = 0; l < 4; ++l)
                fx[l] = __shfl(acc[l], partner, 64);
#pragma unroll
            for (int l = 0; l < 4; ++l)
                z[l] = acc[l] + mX * fx[l];
            const float from_above = dpp_read<0x104>(z[0]); // row_shl:4 : lane + 4 = element above
            const float from_below = dpp_read<0x114>(z[3]); // row_shr:4
            z[3] += mU * from_above;
            z[0] += mD * from_below;
        };

        const float dt = A.dt, half_dt = 0.5f * A.dt;
        const int nt = A.nt;
        for (int whit = 0; whit < A.wh_iters; ++whit)
        {
            {
                const float k0 = filt[0];
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    p[l] = u[l];
                    q[l] = v[l];
                    u[l] *= k0;
                    v[l] *= k0;
                }
            }
            for (int it = 1; it <= nt; ++it)
            {
                const float c0 = cs[2 * it - 2], s0 = sn[2 * it - 2];
                const float c1 = cs[2 * it - 1], s1 = sn[2 * it - 1];
                const float kw = filt[it];
                float z[4], ph[4], qh[4];
                sweep(p, z);
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    const float dq = ((z[l] - Hi[l] * q[l]) + c0 * F[l] + s0 * Gf[l]) * invm[l];
                    ph[l] = p[l] - half_dt * q[l];
                    qh[l] = q[l] + half_dt * dq;
                    p[l] -= dt * qh[l];
                }
                sweep(ph, z);
#pragma unroll
                for (int l = 0; l < 4; ++l)
                {
                    const float dq = ((z[l] - Hi[l] * qh[l]) + c1 * F[l] + s1 * Gf[l]) * invm[l];
                    q[l] += dt * dq;
                    u[l] += kw * p[l];
                    v[l] += kw * q[l];
                }
            }
        }

        const float rw = 1.0f / A.omega;
#pragma unroll
        for (int l = 0; l < 4; ++l)
        {
            v[l] *= rw;
            const bool owner = !(k == 0 && ex > 0) && !(l == 0 && ey > 0);
            if (!owner)
                continue;
            const int d = sI[k + 4 * (l + 4 * el)];
            if (A.y)
            {
                const int gidx = A.gI[dbase + d];
                const float M = A.m[dbase + d] * A.gmi[dbase + d];
                atomic_add(A.y + gidx, static_cast<double>(M * u[l]));
                atomic_add(A.y + A.g_ndof + gidx, static_cast<double>(M * v[l]));
            }
            if (A.update && d < fdof)
            {
                const int wslot = A.B[d + (size_t)A.mx_fdof * (1 + 2 * (size_t)s)];
                if (wslot >= 0)
                {
                    float lam = 0, mu = 0;
                    if (A.lambda)
                    {
                        const int rslot = A.B[d + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                        if (rslot >= 0)
                        {
                            lam = A.lambda[rslot];
                            mu = A.lambda[A.n_lambda + rslot];
                        }
                    }
                    const float S = 2.0f * A.a[dbase + d] * A.omega;
                    A.update[wslot] = -lam - S * v[l];
                    A.update[A.n_lambda + wslot] = -mu + S * u[l];
                }
            }
        }
    }

    // is the metric tensor of every element of every subdomain identical to that of (subdomain 0, element 0)?
    __global__ void __launch_bounds__(256) ddh_uniform_check_kernel(long long n_nodes_total, int nodes_per_elem, const float *__restrict__ G,
                                                                    int *__restrict__ bad)
    {
        for (long long t = blockIdx.x * 256LL + threadIdx.x; t < n_nodes_total; t += gridDim.x * 256LL)
        {
            const int node = static_cast<int>(t % nodes_per_elem); // node within its element
            const float *g = G + 3 * t, *g0 = G + 3 * node;
            if (g[0] != g0[0] || g[1] != g0[1] || g[2] != g0[2])
                atomicExch(bad, 1);
        }
    }

    // structure check for the wave kernels (NB nodes per direction, NEL x NEL elements, NB*NEL == 16): the expected
    // number of dofs, and xi/eta neighbours share exactly the expected nodes
    template <int NB, int NEL>
    __global__ void __launch_bounds__(256) ddh_wave_check_kernel(int n_domains, const int *__restrict__ s_dof, const int *__restrict__ sI,
                                                                 int *__restrict__ bad)
    {
        constexpr int SIDE = NB * NEL - (NEL - 1), NDOF = SIDE * SIDE, NN = NB * NB;
        const int s = blockIdx.x, tid = threadIdx.x;
        if (s >= n_domains)
            return;
        const int k = tid % NB, l = (tid / NB) % NB, el = tid / NN, ex = el % NEL, ey = el / NEL;
        const int *I = sI + 256 * (size_t)s;
        bool ok = I[tid] >= 0 && I[tid] < NDOF;
        if (tid == 0)
            ok = ok && s_dof[s] == NDOF;
        if (k == NB - 1 && ex < NEL - 1)
            ok = ok && I[tid] == I[0 + NB * (l + NB * (el + 1))];
        if (l == NB - 1 && ey < NEL - 1)
            ok = ok && I[tid] == I[k + NB * (0 + NB * (el + NEL))];
        if (!ok)
            atomicExch(bad, 1);
    }

    // ---------------------------------------------------------------- workgroup-per-subdomain kernel (generic)
    template <typename Real, int NB>
    __global__ void __launch_bounds__(256) ddh_block_kernel(DdhArgs<Real> A, const Real *__restrict__ Dmat, const Real *__restrict__ filt,
                                                           const Real *__restrict__ cs, const Real *__restrict__ sn)
    {
        raise_priority(A.prio);
        const int T = A.nodes; // == blockDim.x
        const int tid = threadIdx.x;
        const int s = domain_at(A, A.dom_begin + blockIdx.x);

        extern __shared__ double lds_raw[];
        Real *s_p = reinterpret_cast<Real *>(lds_raw); // [T] field, indexed by subdomain dof
        Real *s_f1 = s_p + T;                           // [T] fluxes, indexed by element node
        Real *s_f2 = s_f1 + T;
        Real *s_su = s_f2 + T;                          // [T] element contributions
        int *s_cnt = reinterpret_cast<int *>(s_su + T); // [T]
        int *s_lst = s_cnt + T;                         // [T][4]

        const int ndof = A.s_dof[s], fdof = A.s_fdof[s];
        const int k = tid % NB, l = (tid / NB) % NB, el = tid / (NB * NB);
        const int *sI = A.sI + (size_t)T * s;
        const int mydof = sI[tid];
        const bool active = mydof >= 0;

        int row[NB], col[NB];
        Real Dk[NB], Dl[NB], DTk[NB], DTl[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i)
        {
            row[i] = active ? sI[i + NB * (l + NB * el)] : 0; // dof of node (i, l)
            col[i] = active ? sI[k + NB * (i + NB * el)] : 0; // dof of node (k, i)
            Dk[i] = Dmat[k + NB * i];
            Dl[i] = Dmat[l + NB * i];
            DTk[i] = Dmat[i + NB * k];
            DTl[i] = Dmat[i + NB * l];
        }
        const Real *g = A.G + 3 * ((size_t)tid + (size_t)T * s);
        const Real g0 = active ? g[0] : Real(0), g1 = active ? g[1] : Real(0), g2 = active ? g[2] : Real(0);

        // who contributes to dof `tid`: element nodes in increasing order (fixed summation order)
        s_cnt[tid] = 0;
        __syncthreads();
        if (active)
        {
            const int slot = atomicAdd(&s_cnt[mydof], 1);
            if (slot < 4)
                s_lst[4 * mydof + slot] = tid;
        }
        __syncthreads();
        int nc = 0, c[4] = {0, 0, 0, 0};
        if (tid < ndof)
        {
            nc = s_cnt[tid];
            if (nc > 4)
                __builtin_trap(); // more than four elements meet at a node: not a structured subdomain
            for (int j = 0; j < nc; ++j)
                c[j] = s_lst[4 * tid + j];
            for (int a = 1; a < nc; ++a) // insertion sort of at most 4 entries
                for (int b = a; b > 0 && c[b - 1] > c[b]; --b)
                {
                    const int t = c[b];
                    c[b] = c[b - 1];
                    c[b - 1] = t;
                }
        }

        // per-dof data
        int g_idx = -1;
        Real ai = 0, mi = 0, inv_mi = 0, Hi = 0, F = 0, G = 0, lam = 0, mu = 0;
        Real u = 0, v = 0, p = 0, q = 0;
        const size_t dbase = (size_t)A.mx_dof * s, fbase = (size_t)A.mx_fdof * s;
        if (tid < ndof)
        {
            g_idx = A.gI[dbase + tid];
            ai = A.a[dbase + tid];
            mi = A.m[dbase + tid];
            inv_mi = Real(1) / (ai * ai * mi);
            if (A.x)
            {
                F = static_cast<Real>(A.x[g_idx]);
                G = static_cast<Real>(A.x[A.g_ndof + g_idx]);
            }
        }
        if (tid < fdof)
        {
            Hi = A.H[fbase + tid];
            if (A.lambda)
            {
                const int slot = A.B[tid + (size_t)A.mx_fdof * (0 + 2 * (size_t)s)];
                if (slot >= 0)
                {
                    lam = A.lambda[slot];
                    mu = A.lambda[A.n_lambda + slot];
                    F += Hi * lam;
                    G += Hi * mu;
                }
            }
            Hi *= ai;
        }

        // one stiffness sweep: field in s_p (already published), returns (S field)[tid] for tid < ndof
        auto sweep = [&]() -> Real
        {
            Real Ux = 0, Uy = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
            {
                Ux += Dk[i] * s_p[row[i]];
                Uy += Dl[i] * s_p[col[i]];
            }
            s_f1[tid] = g0 * Ux + g1 * Uy;
            s_f2[tid] = g1 * Ux + g2 * Uy;
            __syncthreads();
            Real Su = 0;
#pragma unroll
            for (int i = 0; i < NB; ++i)
            {
                Su += DTk[i] * s_f1[i + NB * (l + NB * el)];
                Su += DTl[i] * s_f2[k + NB * (i + NB * el)];
            }
            s_su[tid] = Su;
            __syncthreads();
            Real z = 0;
            for (int j = 0; j < nc; ++j)
                z += s_su[c[j]];
            return z;
        };

        const Real dt = A.dt, half_dt = Real(0.5) * A.dt;
        const int nt = A.nt;
        for (int whit = 0; whit < A.wh_iters; ++whit)
        {
            const Real k0 = filt[0];
            p = u;
            q = v;
            u *= k0;
            v *= k0;
            for (int it = 1; it <= nt; ++it)
            {
                s_p[tid] = p;
                __syncthreads();
                Real z = sweep() - Hi * q;
                Real dq = (z + cs[2 * it - 2] * F + sn[2 * it - 2] * G) * inv_mi;
                const Real ph = p - half_dt * q;
                const Real qh = q + half_dt * dq;
                p -= dt * qh;

                s_p[tid] = ph;
                __syncthreads();
                z = sweep() - Hi * qh;
                dq = (z + cs[2 * it - 1] * F + sn[2 * it - 1] * G) * inv_mi;
                q += dt * dq;

                const Real kw = filt[it];
                u += kw * p;
                v += kw * q;
            }
        }

        v *= Real(1) / A.omega;

        if (A.y && tid < ndof)
        {
            const Real M = mi * A.gmi[dbase + tid];
            atomic_add(A.y + g_idx, static_cast<double>(M * u));
            atomic_add(A.y + A.g_ndof + g_idx, static_cast<double>(M * v));
        }
        if (A.update && tid < fdof)
        {
            const int wslot = A.B[tid + (size_t)A.mx_fdof * (1 + 2 * (size_t)s)];
            if (wslot >= 0)
            {
                const Real S = Real(2) * ai * A.omega;
                A.update[wslot] = -lam - S * v;
                A.update[A.n_lambda + wslot] = -mu + S * u;
            }
        }
    }

    // ---------------------------------------------------------------- geometric factors (K13)
    template <typename Real>
    __global__ void __launch_bounds__(256) ddh_geom_kernel(long long total, int nodes, int nb, int mx_elems, const int *__restrict__ n_elems,
                                                          const int *__restrict__ elems, const double *__restrict__ w,
                                                          const double *__restrict__ J, const double *__restrict__ corners,
                                                          const double *__restrict__ pts, Real *__restrict__ G)
    {
        for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL)
        {
            const int s = static_cast<int>(t / nodes), node = static_cast<int>(t % nodes);
            const int i = node % nb, j = (node / nb) % nb, el = node / (nb * nb);
            Real *g = G + 3 * t;
            if (el >= n_elems[s])
            {
                g[0] = g[1] = g[2] = 0;
                continue;
            }
            const int g_el = elems[el + (size_t)mx_elems * s];
            const double W = w[i] * w[j];
            double j4[4];
            if (J)
            {
                const double *Jp = J + 4 * ((size_t)i + nb * ((size_t)j + nb * (size_t)g_el));
                j4[0] = Jp[0], j4[1] = Jp[1], j4[2] = Jp[2], j4[3] = Jp[3];
            }
            else // straight from the element's corners: no (2, 2, nb, nb, g_elem) table
                bilinear_jacobian(corners + 8 * (size_t)g_el, pts[i], pts[j], j4);
            const double x_xi = j4[0], y_xi = j4[1], x_eta = j4[2], y_eta = j4[3];
            const double det = x_xi * y_eta - x_eta * y_xi;
            g[0] = static_cast<Real>(W * (y_eta * y_eta + x_eta * x_eta) / det);
            g[1] = static_cast<Real>(-W * (y_xi * y_eta + x_xi * x_eta) / det);
            g[2] = static_cast<Real>(W * (y_xi * y_xi + x_xi * x_xi) / det);
        }
    }

    template <typename Real>
    int geom_setup(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w, const double *J,
                   const double *corners, const double *pts, Real *G, void *stream)
    {
        const int nodes = nb * nb * mx_elems;
        const long long total = (long long)nodes * n_domains;
        if (total <= 0)
            return 0;
        hipLaunchKernelGGL((ddh_geom_kernel<Real>), dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(stream), total, nodes, nb,
                           mx_elems, n_elems, elems, w, J, corners, pts, G);
        return launch_status();
    }

    template <typename Real, int NB>
    void launch_block(const DdhArgs<Real> &A, const cuddh_ddh_desc &d, int n_local, hipStream_t st)
    {
        const int T = A.nodes;
        const size_t lds = (size_t)T * (4 * sizeof(Real) + 5 * sizeof(int));
        hipLaunchKernelGGL((ddh_block_kernel<Real, NB>), dim3(n_local), dim3(T), lds, st, A, static_cast<const Real *>(d.D),
                           static_cast<const Real *>(d.wh_filter), static_cast<const Real *>(d.cs), static_cast<const Real *>(d.sn));
    }

    // Builds plan->Aop for kernel 5.  Returns 0 on success, -1 if the geometry is not uniform, > 0 on a HIP error.
    int build_dense_element_matrix(cuddh_ddh_plan *p)
    {
        const cuddh_ddh_desc &d = p->d;
        const float *G = static_cast<const float *>(d.G);
        int *flag = nullptr;
        hipError_t e = hipMalloc(&flag, sizeof(int));
        if (e != hipSuccess)
            return static_cast<int>(e);
        (void)hipMemset(flag, 0, sizeof(int));
        const long long n_nodes = 256LL * d.n_domains;
        hipLaunchKernelGGL(ddh_uniform_check_kernel, dim3(stream_grid(n_nodes, 256)), dim3(256), 0, nullptr, n_nodes, 16, G, flag);
        int bad = 1;
        e = hipMemcpy(&bad, flag, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipFree(flag);
        if (e != hipSuccess)
            return static_cast<int>(e);
        if (bad)
            return -1;

        float hD[16], hG[48];
        e = hipMemcpy(hD, d.D, sizeof hD, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(hG, G, sizeof hG, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return static_cast<int>(e);

        // K[n_out][n_in], n = k + 4 l: one stiffness sweep (source/DDH.cpp:60-109) applied to unit vectors, in double
        auto Dm = [&](int a, int b) { return static_cast<double>(hD[a + 4 * b]); }; // D(a,b)
        double K[16][16];
        for (int nin = 0; nin < 16; ++nin)
        {
            double U[16] = {0};
            U[nin] = 1.0;
            double f1[16], f2[16];
            for (int l = 0; l < 4; ++l)
                for (int k = 0; k < 4; ++k)
                {
                    double ux = 0, uy = 0;
                    for (int i = 0; i < 4; ++i)
                    {
                        ux += Dm(k, i) * U[i + 4 * l];
                        uy += Dm(l, i) * U[k + 4 * i];
                    }
                    const float *g = hG + 3 * (k + 4 * l);
                    f1[k + 4 * l] = g[0] * ux + g[1] * uy;
                    f2[k + 4 * l] = g[1] * ux + g[2] * uy;
                }
            for (int l = 0; l < 4; ++l)
                for (int k = 0; k < 4; ++k)
                {
                    double su = 0;
                    for (int i = 0; i < 4; ++i)
                        su += Dm(i, k) * f1[i + 4 * l] + Dm(i, l) * f2[k + 4 * i];
                    K[k + 4 * l][nin] = su;
                }
        }
        // A operand of step st, lane ln:  A[i = ln & 15][kk = 4 st + (ln >> 4)] = K'[m = i][nu = kk],
        // m = 4 k_out + l_out, nu = 4 l_in + k_in
        float hA[256];
        for (int st = 0; st < 4; ++st)
            for (int ln = 0; ln < 64; ++ln)
            {
                const int m = ln & 15, nu = 4 * st + (ln >> 4);
                const int k_out = m >> 2, l_out = m & 3, l_in = nu >> 2, k_in = nu & 3;
                hA[64 * st + ln] = static_cast<float>(K[k_out + 4 * l_out][k_in + 4 * l_in]);
            }
        e = hipMalloc(reinterpret_cast<void **>(&p->Aop), sizeof hA);
        if (e == hipSuccess)
            e = hipMemcpy(p->Aop, hA, sizeof hA, hipMemcpyHostToDevice);
        return static_cast<int>(e);
    }

    // kernel 7 tables: needs one metric tensor for all elements, diagonal (gy == 0) and a product of 1-D factors
    // (rectangles).  Returns 0 on success, -1 when the geometry does not qualify, > 0 on a HIP error.
    int build_separable_tables(cuddh_ddh_plan *p)
    {
        const cuddh_ddh_desc &d = p->d;
        const float *G = static_cast<const float *>(d.G);
        int *flag = nullptr;
        hipError_t e = hipMalloc(&flag, sizeof(int));
        if (e != hipSuccess)
            return static_cast<int>(e);
        (void)hipMemset(flag, 0, sizeof(int));
        const long long n_nodes = 256LL * d.n_domains;
        hipLaunchKernelGGL(ddh_uniform_check_kernel, dim3(stream_grid(n_nodes, 256)), dim3(256), 0, nullptr, n_nodes, 64, G, flag);
        int bad = 1;
        e = hipMemcpy(&bad, flag, sizeof(int), hipMemcpyDeviceToHost);
        (void)hipFree(flag);
        if (e != hipSuccess)
            return static_cast<int>(e);
        if (bad)
            return -1;
        float hD[64], hG[192];
        e = hipMemcpy(hD, d.D, sizeof hD, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            e = hipMemcpy(hG, G, sizeof hG, hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return static_cast<int>(e);
        auto Dm = [&](int a, int b) { return static_cast<double>(hD[a + 8 * b]); }; // D(a,b)
        auto g = [&](int c, int k, int l) { return static_cast<double>(hG[3 * (k + 8 * l) + c]); };
        double scale = 0.0;
        for (int n = 0; n < 64; ++n)
            scale = std::max(scale, std::fabs(static_cast<double>(hG[3 * n])) + std::fabs(static_cast<double>(hG[3 * n + 2])));
        double alpha[8], beta[8], gamma[8], delta[8];
        for (int i = 0; i < 8; ++i)
        {
            alpha[i] = g(0, i, 0);
            beta[i] = g(0, 0, i) / g(0, 0, 0);
            gamma[i] = g(2, i, 0) / g(2, 0, 0);
            delta[i] = g(2, 0, i);
        }
        for (int l = 0; l < 8; ++l)
            for (int k = 0; k < 8; ++k)
                if (std::fabs(g(1, k, l)) > 1e-6 * scale || std::fabs(g(0, k, l) - alpha[k] * beta[l]) > 1e-6 * scale ||
                    std::fabs(g(2, k, l) - gamma[k] * delta[l]) > 1e-6 * scale)
                    return -1;
        float hS[144] = {0};
        for (int a = 0; a < 8; ++a)
            for (int b = 0; b < 8; ++b)
            {
                double ax = 0.0, ay = 0.0;
                for (int i = 0; i < 8; ++i)
                {
                    ax += Dm(i, a) * alpha[i] * Dm(i, b);
                    ay += Dm(i, a) * delta[i] * Dm(i, b);
                }
                hS[a + 8 * b] = static_cast<float>(ax);
                if (a <= b)
                    hS[64 + sym_index(a, b)] = static_cast<float>(ay); // Ay(a,b) == Ay(b,a)
            }
        for (int i = 0; i < 8; ++i)
        {
            hS[128 + i] = static_cast<float>(beta[i]);
            hS[136 + i] = static_cast<float>(gamma[i]);
        }
        e = hipMalloc(reinterpret_cast<void **>(&p->Sep), sizeof hS);
        if (e == hipSuccess)
            e = hipMemcpy(p->Sep, hS, sizeof hS, hipMemcpyHostToDevice);
        return static_cast<int>(e);
    }

    template <typename Real>
    int apply(const cuddh_ddh_plan *plan, const int *dom_list, int dom_begin, int dom_end, const double *x, double *y, int zero_y, const Real *lambda,
              Real *update, void *stream)
    {
        if (!plan || plan->is_f64 != (sizeof(Real) == 8 ? 1 : 0))
            return static_cast<int>(hipErrorInvalidValue);
        const cuddh_ddh_desc &d = plan->d;
        if (dom_begin < 0 || (!dom_list && dom_end > d.n_domains) || dom_begin > dom_end)
            return static_cast<int>(hipErrorInvalidValue);
        hipStream_t st = as_stream(stream);
        const int g_ndof = plan->gI_override ? plan->g_ndof_override : d.g_ndof;
        if (y && zero_y)
        {
            hipError_t e = hipMemsetAsync(y, 0, (size_t)2 * g_ndof * sizeof(double), st);
            if (e != hipSuccess)
                return static_cast<int>(e);
        }
        const int n_local = dom_end - dom_begin;
        if (n_local == 0)
            return 0;

        DdhArgs<Real> A;
        A.g_ndof = g_ndof;
        A.n_lambda = d.n_lambda;
        A.nt = d.nt;
        A.wh_iters = plan->wh_iters;
        A.prio = plan->wave_priority;
        A.mx_dof = d.mx_dof;
        A.mx_fdof = d.mx_fdof;
        A.nodes = plan->nodes;
        A.dom_begin = dom_begin;
        A.dom_end = dom_end;
        A.dom_list = dom_list;
        A.omega = static_cast<Real>(d.omega);
        A.dt = static_cast<Real>(d.dt);
        A.s_dof = d.s_dof;
        A.s_fdof = d.s_fdof;
        A.B = d.B;
        A.gI = plan->gI_override ? plan->gI_override : d.gI;
        A.sI = d.sI;
        A.G = static_cast<const Real *>(d.G);
        A.m = static_cast<const Real *>(d.m);
        A.gmi = static_cast<const Real *>(d.gmi);
        A.a = static_cast<const Real *>(d.a);
        A.H = static_cast<const Real *>(d.H);
        A.x = x;
        A.y = y;
        A.lambda = lambda;
        A.update = update;

        if (plan->kernel == 5)
        {
            if constexpr (sizeof(Real) == 4)
            {
                hipLaunchKernelGGL(ddh_mfma_kernel, dim3((n_local + 3) / 4), dim3(256), 0, st, A, plan->Aop, static_cast<const float *>(d.wh_filter),
                                   static_cast<const float *>(d.cs), static_cast<const float *>(d.sn));
                return launch_status();
            }
        }
        if (plan->kernel == 6 || plan->kernel == 7)
        {
            const dim3 grid((n_local + 7) / 8), block(256); // four wavefronts per workgroup, two subdomains per wavefront
            const Real *D = static_cast<const Real *>(d.D), *fl = static_cast<const Real *>(d.wh_filter);
            const Real *cs = static_cast<const Real *>(d.cs), *sn = static_cast<const Real *>(d.sn);
            if constexpr (sizeof(Real) == 4)
            {
                if (plan->kernel == 7)
                    hipLaunchKernelGGL((ddh_wave8_kernel<float, true, true>), grid, block, 0, st, A, D, fl, cs, sn, plan->Sep);
                else
                    hipLaunchKernelGGL((ddh_wave8_kernel<float, true, false>), grid, block, 0, st, A, D, fl, cs, sn, plan->Sep);
            }
            else
                hipLaunchKernelGGL((ddh_wave8_kernel<double, false, false>), grid, block, 0, st, A, D, fl, cs, sn,
                                   static_cast<const double *>(nullptr));
            return launch_status();
        }
        if (plan->kernel >= 2)
        {
            // kernels 3 and 4 exist in fp32 only; fp64 always takes the plain form
            constexpr int v3 = sizeof(Real) == 4 ? 1 : 0, v4 = sizeof(Real) == 4 ? 2 : 0;
            const dim3 grid((n_local + 3) / 4), block(256);
            const Real *D = static_cast<const Real *>(d.D), *fl = static_cast<const Real *>(d.wh_filter);
            const Real *cs = static_cast<const Real *>(d.cs), *sn = static_cast<const Real *>(d.sn);
            if (plan->kernel == 4)
                hipLaunchKernelGGL((ddh_wave_kernel<Real, v4>), grid, block, 0, st, A, D, fl, cs, sn);
            else if (plan->kernel == 3 || plan->kernel == 5)
                hipLaunchKernelGGL((ddh_wave_kernel<Real, v3>), grid, block, 0, st, A, D, fl, cs, sn);
            else
                hipLaunchKernelGGL((ddh_wave_kernel<Real, 0>), grid, block, 0, st, A, D, fl, cs, sn);
            return launch_status();
        }

        switch (d.nb)
        {
        case 2: launch_block<Real, 2>(A, d, n_local, st); break;
        case 3: launch_block<Real, 3>(A, d, n_local, st); break;
        case 4: launch_block<Real, 4>(A, d, n_local, st); break;
        case 5: launch_block<Real, 5>(A, d, n_local, st); break;
        case 6: launch_block<Real, 6>(A, d, n_local, st); break;
        case 7: launch_block<Real, 7>(A, d, n_local, st); break;
        case 8: launch_block<Real, 8>(A, d, n_local, st); break;
        case 9: launch_block<Real, 9>(A, d, n_local, st); break;
        case 10: launch_block<Real, 10>(A, d, n_local, st); break;
        default: return static_cast<int>(hipErrorInvalidValue);
        }
        return launch_status();
    }
    // do all subdomains have the structured NEL x NEL element layout the wave kernels assume?  *bad = 0 if so
    template <int NB, int NEL>
    int run_structure_check(const cuddh_ddh_desc *desc, int *bad)
    {
        int *flag = nullptr;
        hipError_t e = hipMalloc(&flag, sizeof(int));
        if (e == hipSuccess)
            e = hipMemset(flag, 0, sizeof(int));
        *bad = 1;
        if (e == hipSuccess)
        {
            hipLaunchKernelGGL((ddh_wave_check_kernel<NB, NEL>), dim3(desc->n_domains), dim3(256), 0, nullptr, desc->n_domains,
                               desc->s_dof, desc->sI, flag);
            e = hipMemcpy(bad, flag, sizeof(int), hipMemcpyDeviceToHost);
        }
        if (flag)
            (void)hipFree(flag);
        return static_cast<int>(e);
    }
} // namespace

extern "C"
{
    int cuddh_hip_ddh_geom_setup_f32(int n_domains, int mx_elems, int g_elem, int nb, const int *n_elems, const int *elems,
                                     const double *w, const double *J, float *G, void *stream)
    {
        (void)g_elem;
        return geom_setup<float>(n_domains, mx_elems, nb, n_elems, elems, w, J, nullptr, nullptr, G, stream);
    }

    int cuddh_hip_ddh_geom_setup_f64(int n_domains, int mx_elems, int g_elem, int nb, const int *n_elems, const int *elems,
                                     const double *w, const double *J, double *G, void *stream)
    {
        (void)g_elem;
        return geom_setup<double>(n_domains, mx_elems, nb, n_elems, elems, w, J, nullptr, nullptr, G, stream);
    }

    int cuddh_hip_ddh_geom_from_corners_f32(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w,
                                            const double *points, const double *corners, float *G, void *stream)
    {
        if (!corners || !points)
            return static_cast<int>(hipErrorInvalidValue);
        return geom_setup<float>(n_domains, mx_elems, nb, n_elems, elems, w, nullptr, corners, points, G, stream);
    }

    int cuddh_hip_ddh_geom_from_corners_f64(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w,
                                            const double *points, const double *corners, double *G, void *stream)
    {
        if (!corners || !points)
            return static_cast<int>(hipErrorInvalidValue);
        return geom_setup<double>(n_domains, mx_elems, nb, n_elems, elems, w, nullptr, corners, points, G, stream);
    }

    int cuddh_hip_ddh_plan_create(cuddh_ddh_plan **out, const cuddh_ddh_desc *desc, int is_f64, int kernel)
    {
        *out = nullptr;
        if (!desc || desc->nb < 2 || desc->nb > 10 || desc->nel1d < 1 || kernel < 0 || kernel > 7)
            return static_cast<int>(hipErrorInvalidValue);
        const int nodes = desc->nb * desc->nb * desc->nel1d * desc->nel1d;
        if (nodes > 256)
            return static_cast<int>(hipErrorInvalidValue);

        cuddh_ddh_plan *p = new cuddh_ddh_plan;
        p->d = *desc;
        p->is_f64 = is_f64 ? 1 : 0;
        p->nodes = nodes;
        p->kernel = 1;

        const bool wave_shape = (desc->nb == 4 && desc->nel1d == 4);
        const bool wave8_shape = (desc->nb == 8 && desc->nel1d == 2);
        if ((kernel >= 2 && kernel <= 5 && !wave_shape) || (kernel >= 6 && !wave8_shape) || (kernel == 7 && is_f64))
        {
            delete p;
            return static_cast<int>(hipErrorInvalidValue);
        }
        if (wave8_shape && kernel != 1)
        {
            int bad = 1;
            const int e = run_structure_check<8, 2>(desc, &bad);
            if (e)
            {
                delete p;
                return e;
            }
            if (!bad)
                p->kernel = 6;
            else if (kernel >= 6)
            {
                delete p;
                return static_cast<int>(hipErrorInvalidValue);
            }
            // kernel 7 (one contraction per direction) needs fp32 and the rectangles of a uniform mesh
            if (!bad && !is_f64 && (kernel == 0 || kernel == 7))
            {
                const int err7 = build_separable_tables(p);
                if (err7 == 0)
                    p->kernel = 7;
                else if (kernel == 7)
                {
                    delete p;
                    return err7 > 0 ? err7 : static_cast<int>(hipErrorInvalidValue);
                }
            }
        }
        if (wave_shape && kernel != 1)
        {
            int bad = 1;
            const int e = run_structure_check<4, 4>(desc, &bad);
            if (e)
            {
                delete p;
                return e;
            }
            if (!bad)
                p->kernel = (kernel >= 2) ? kernel : 3; // auto prefers the folded-DPP form (fp64 runs the plain form either way)
            else if (kernel >= 2)
            {
                delete p;
                return static_cast<int>(hipErrorInvalidValue);
            }
            // kernel 5 (dense element matrix on the matrix cores) needs fp32 and one metric tensor for all elements
            if (!bad && !is_f64 && (kernel == 0 || kernel == 5))
            {
                int err5 = build_dense_element_matrix(p);
                if (err5 == 0)
                    p->kernel = 5;
                else if (kernel == 5)
                {
                    delete p;
                    return err5 > 0 ? err5 : static_cast<int>(hipErrorInvalidValue);
                }
            }
            else if (kernel == 5)
            {
                delete p;
                return static_cast<int>(hipErrorInvalidValue);
            }
        }
        *out = p;
        return 0;
    }

    int cuddh_hip_ddh_plan_destroy(cuddh_ddh_plan *plan)
    {
        if (plan && plan->Aop)
            (void)hipFree(plan->Aop);
        if (plan && plan->Sep)
            (void)hipFree(plan->Sep);
        delete plan;
        return 0;
    }

    int cuddh_hip_ddh_plan_kernel(const cuddh_ddh_plan *plan) { return plan ? plan->kernel : 0; }

    int cuddh_hip_ddh_plan_set_vector_layout(cuddh_ddh_plan *plan, const int *d_gI, int g_ndof)
    {
        if (!plan || (d_gI && g_ndof <= 0))
            return static_cast<int>(hipErrorInvalidValue);
        plan->gI_override = d_gI;
        plan->g_ndof_override = d_gI ? g_ndof : 0;
        return 0;
    }

    int cuddh_hip_ddh_plan_set_wave_priority(cuddh_ddh_plan *plan, int high)
    {
        if (!plan)
            return static_cast<int>(hipErrorInvalidValue);
        plan->wave_priority = high ? 1 : 0;
        return 0;
    }

    int cuddh_hip_ddh_plan_set_wh_iters(cuddh_ddh_plan *plan, int wh_iters)
    {
        if (!plan || wh_iters < 0)
            return static_cast<int>(hipErrorInvalidValue);
        plan->wh_iters = wh_iters == 0 ? WH_ITERS_REFERENCE : wh_iters;
        return 0;
    }

    int cuddh_hip_ddh_apply_f32(const cuddh_ddh_plan *plan, int dom_begin, int dom_end, const double *x, double *y, int zero_y,
                                const float *lambda, float *update, void *stream)
    {
        return apply<float>(plan, nullptr, dom_begin, dom_end, x, y, zero_y, lambda, update, stream);
    }

    int cuddh_hip_ddh_apply_list_f32(const cuddh_ddh_plan *plan, const int *d_domains, int n, const double *x, double *y, int zero_y,
                                     const float *lambda, float *update, void *stream)
    {
        if (!d_domains && n > 0)
            return static_cast<int>(hipErrorInvalidValue);
        return apply<float>(plan, d_domains, 0, n, x, y, zero_y, lambda, update, stream);
    }

    int cuddh_hip_ddh_apply_list_f64(const cuddh_ddh_plan *plan, const int *d_domains, int n, const double *x, double *y, int zero_y,
                                     const double *lambda, double *update, void *stream)
    {
        if (!d_domains && n > 0)
            return static_cast<int>(hipErrorInvalidValue);
        return apply<double>(plan, d_domains, 0, n, x, y, zero_y, lambda, update, stream);
    }

    int cuddh_hip_ddh_apply_f64(const cuddh_ddh_plan *plan, int dom_begin, int dom_end, const double *x, double *y, int zero_y,
                                const double *lambda, double *update, void *stream)
    {
        return apply<double>(plan, nullptr, dom_begin, dom_end, x, y, zero_y, lambda, update, stream);
    }
}
