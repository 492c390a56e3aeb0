// Plan-based matrix-free operator applies: the fused complex Helmholtz apply
//     [u;v] -> [S u - w^2 M u - w H v ; -(S v - w^2 M v + w H u)]
// and the single real operators  y = [y +] c S x,  y = [y +] c M x  behind the same plan.
//
// Kernels in this file
//   helm_patch_kernel   n_basis 2-5, complex: one element per lane, lanes 0-31 u / 32-63 v; patches of 32 elements (one
//                       wavefront) or, for affine plans, 64 (two wavefronts sharing the LDS copy)
//   helm_lane_kernel    n_basis 2-4, complex, general geometry, large plans: lane = element, both components per lane,
//                       64-element patches, one wavefront each
//   op_patch_kernel     n_basis 2-5, real   : one element per lane, one patch of 64 elements per wavefront
//   helm_mfma_kernel    n_basis 6-8, complex: batches of 16 elements, 1-D contractions on v_mfma_f64_4x4x4 / 16x16x4; n_basis 6, 7: the
//                       batch's metric data in chunks requested one ahead and parked in LDS (no trip to memory inside a slice)
//   op_mfma_kernel      n_basis 6-8, real   : the same for one operator
//   helm_border_kernel / op_border_kernel: sums of the per-patch contributions at dofs shared by several patches
//   repack_*, uniform_metric_kernel: plan construction
//
// Common design (MI355X): elements are grouped into patches (Morton order of their centroids: 8x8, 4x8 or 4x4 blocks on a
// structured mesh); one wavefront owns one patch, or two wavefronts share one.
//   * x of the patch's dofs is gathered once into LDS; results are accumulated in LDS in colour phases (elements of one
//     colour share no dof), i.e. without atomics and in a fixed order;
//   * the sum factorisation of an element runs in registers; the 1-D interpolation / differentiation matrices are loads
//     from a uniform pointer (scalar registers) or MFMA A operands;
//   * metric arrays are stored patch-major, structure-of-arrays, one contiguous block per quadrature slice, so every load
//     instruction reads 256-512 contiguous bytes and DRAM pages are read whole; on affine meshes one copy serves all;
//   * the complex kernels also take x / y in the plan's own ordering (pairs (u, v), a patch's owned dofs contiguous: 16-byte
//     accesses at addresses known at kernel entry, lists for the border dofs only) -- what HelmholtzOperator::gmres iterates on;
//   * dofs owned by one patch are stored straight to y; dofs on patch borders go to per-patch slots that a second small
//     kernel sums in a fixed order.
// Every apply is therefore bitwise reproducible and needs no zero-fill of y.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.hpp"

using namespace cuddh_k;

struct cuddh_helmholtz_plan
{
    int ndof = 0, n_elem = 0, nb = 0, nqS = 0, nqM = 0, nqF = 0, n_faces = 0;
    int n_patches = 0, max_loc = 0, ncol = 0, nfcol = 0, n_shared = 0, n_slots = 0;
    // per patch
    int *dof_off = nullptr;   // [n_patches + 1] offsets into dof_list / slot_of
    int *dof_list = nullptr;  // global dof of every patch-local dof
    int *slot_of = nullptr;   // where a patch-local dof's result goes: its global dof (>= 0) if the patch owns it, else -(slot in `part`) - 1
    int dof_stride = 0;       // != 0: dof_list / slot_of are stored with this fixed stride per patch (= max_loc), padded with the last entry
    int *own_count = nullptr; // [n_patches] local dofs [0, own_count) are owned (slot_of == dof_list there), the border dofs come last
    int *patch_nel = nullptr; // elements in the patch (32 except possibly the last)
    uint32_t *lidx = nullptr; // [n_patches][ceil(nb*nb/2)][32]: element nodes 2j, 2j+1 -> patch-local dofs, packed lo | hi << 16
    uint8_t *colour = nullptr; // [n_patches][32]
    double *Gp = nullptr;     // [n_patches][q: nqS][3][r: nqS][32]  (point (q,r) = xi index q, eta index r)
    double *aMp = nullptr;    // [n_patches][q: nqM][r: nqM][32]
    // affine meshes (every element has the same metric array, e.g. uniform_rect): one copy, read through scalar loads
    double *Gu = nullptr; // [q][3][r]  (then Gp is not allocated)
    double *au = nullptr; // [q][r]     (then aMp is not allocated)
    // high-order single operators on the fp64 matrix cores: batches of 16 elements, metric as [batch][r][3][q][16]
    double *Gm = nullptr, *Am = nullptr; // stiffness metric, mass weights ([batch][r][q][16])
    long long gm_stride = 0, am_stride = 0; // doubles between the blocks of consecutive batches; 0: one block for all (affine mesh)
    int pe = 32; // elements per patch (16 for the matrix-core plans)
    // faces, grouped by patch
    int *face_off = nullptr;       // [n_patches + 1]
    uint16_t *face_lidx = nullptr; // [n_faces_total][nb]
    int *face_id = nullptr;        // original face index (column of aF)
    uint8_t *face_col = nullptr;
    const double *aF = nullptr; // borrowed: (nqF, n_faces)
    // basis tables
    double *PS = nullptr, *DS = nullptr, *PM = nullptr, *PF = nullptr;
    // patch-border dofs
    int *shared_dof = nullptr, *shared_off = nullptr; // border dof j sums the contiguous slots shared_off[j] .. shared_off[j+1]
    double *part = nullptr; // [2][n_slots]
    size_t bytes_alg = 0, bytes_actual = 0;
    int streaming = 0; // metric loads carry the non-temporal hint (plans larger than the infinity cache)
    int lane_form = 0; // fused apply through helm_lane_kernel (one element per lane, both components)
    int prefetch = 0;  // lane form with the whole patch's metric block requested up front (one wavefront per SIMD)
    int pair_mass = 0; // helm_patch_kernel, general layout, n_basis 3: two mass slices per round trip (CUDDH_HELM_PAIR_MASS=0 for A/B)
    int pair_layout = 0; // metric slices stored as [pairs of values][64 lanes][2] (+ one single row): 16-byte loads (lane form)
    unsigned long long *stamps = nullptr; // CUDDH_HELM_STAMPS=1: [n_patches][8] phase time stamps (100 MHz) of the lane form
    size_t bytes_affine = 0; // algorithmic bytes of the affine form (0 when neither metric array is uniform)
    // Plan-native vector ordering (lane-form plans; cuddh_hip_helmholtz_apply_native): a vector is an array of (u, v) PAIRS in the
    // order [patch 0's owned dofs | patch 1's owned dofs | ... | patch-border dofs in shared_dof order], so a patch reads and writes
    // its owned dofs with contiguous 16-byte accesses and needs index lists for its border dofs only.
    int *own_off = nullptr;          // [n_patches + 1] native position of a patch's first owned dof; own_off[n_patches] = all owned dofs
    int *bpos = nullptr;             // [n_patches][bstride] native position of the patch's border dofs (padded with the last entry)
    int *bslot = nullptr;            // [n_patches][bstride] slot in `part` of the patch's border dofs
    int bstride = 0;
    int *global_of_native = nullptr; // [ndof] reference dof id at every native position (the permutation, for to / from native)
    int n_owned = 0;
    size_t bytes_native = 0; // bytes the native apply moves (layout figure, like bytes_actual)
};

namespace
{
    constexpr int PE = 32; // elements per patch

    // metric arrays are read exactly once per apply: when the plan does not fit the 256 MB infinity cache anyway they
    // are loaded with the streaming hint so that they do not evict x, y and the dof lists (NT); small plans keep the
    // default policy and stay cache resident from one apply to the next
    template <bool NT>
    __device__ inline double metric_load(const double *p)
    {
        if constexpr (NT)
            return __builtin_nontemporal_load(p);
        else
            return *p;
    }

    // 16-byte metric loads.  A slice of NV values per element is stored as NV/2 pairs, [pair][64 lanes][2 doubles], followed
    // by one [64 lanes] row when NV is odd: a wavefront's load instruction then moves 1 KiB instead of 512 B.  On the metric
    // stream alone (profiles/tools/dma_stream.hip, 8 wavefronts per CU, one slice per round trip) that is 5.7 against 4.6 TB/s.
    typedef double dbl2_t __attribute__((ext_vector_type(2)));
    template <int NV, bool NT>
    __device__ inline void load_pairs(const double *__restrict__ slice, int lane, double (&v)[NV])
    {
        constexpr int VP = NV / 2;
        const dbl2_t *p2 = reinterpret_cast<const dbl2_t *>(slice) + lane;
#pragma unroll
        for (int k = 0; k < VP; ++k)
        {
            dbl2_t t;
            if constexpr (NT)
                t = __builtin_nontemporal_load(p2 + k * 64);
            else
                t = p2[k * 64];
            v[2 * k] = t.x;
            v[2 * k + 1] = t.y;
        }
        if constexpr (NV % 2 == 1)
            v[NV - 1] = metric_load<NT>(slice + 2 * VP * 64 + lane);
    }

    struct HelmArgs
    {
        int ndof, max_loc, ncol, nfcol, nqF, n_slots, n_patches, xcd_chunk;
        int dof_stride; // != 0: patch p's dof_list / slot_of segment starts at p * dof_stride (no offset load), padded to dof_stride entries
        double omega;
        const int *dof_off, *dof_list, *slot_of, *patch_nel, *face_off, *face_id;
        const int *own_count;
        const uint32_t *lidx;
        const uint16_t *face_lidx;
        const uint8_t *colour, *face_col;
        const double *Gp, *aMp, *aF;
        const double *x;
        double *y, *part;
        unsigned long long *stamps; // diagnostic: phase time stamps per patch, or null
        int pair_mass;              // helm_patch_kernel: two mass slices per round trip
        // plan-native vector ordering (helm_lane_kernel<..., NATIVE>): x, y and part are then arrays of (u, v) pairs
        const int *own_off, *bpos, *bslot;
        int bstride;
    };

    // Variants measured and dropped: software-pipelined slice loads, slices split between the half-waves
    // and exchanged with ds_bpermute, three role-specialised wavefronts per patch, touch-prefetch of the metric block.
    // UG: the stiffness metric is the same in every element and comes from the uniform table GU (scalar loads)
    // PEK elements per patch: 32 = one wavefront, 64 = two wavefronts sharing the LDS copy of a larger patch (fewer border dofs)
    // MODE 0: one slice per dependent round trip; 1: the mass phase takes two slices per round trip (n_basis 3: 256^2 19.5 ->
    // 18.0 us, 512^2 unchanged; at n_basis 5 it spills 35 registers at 3 wavefronts per SIMD: 421 -> 549 us, not used there.  A
    // double-buffered chain at 2 wavefronts per SIMD for n_basis 5 -- next slice requested before the current one is computed
    // -- compiled to 256 registers + 13 spilled with the requests sunk below the arithmetic again: 431 -> 450 us, removed)
#ifndef HELM_PATCH_RELOAD_MAP
#define HELM_PATCH_RELOAD_MAP 1
#endif
    // NATIVE: x and y in the plan's own vector ordering (pairs (u, v), a patch's owned dofs contiguous; see helm_lane_kernel)
    template <int NB, int NQS, int NQM, bool NT, bool UG, int PEK, int MODE = 0, bool NATIVE = false>
    __global__ void __launch_bounds__(2 * PEK, (NB >= 5 ? (UG ? 2 : 3) : ((NB == 4 && !UG) ? 3 : 4))) helm_patch_kernel(HelmArgs A, const double *__restrict__ PS, const double *__restrict__ DS,
                                                           const double *__restrict__ PM, const double *__restrict__ PF,
                                                           const double *__restrict__ GU)
    {
        constexpr int NN = NB * NB;
        extern __shared__ double lds[];
        // workgroups are dealt round-robin to the 8 XCDs: give each XCD one contiguous (Morton-ordered) range of patches
        // so that the dofs shared by neighbouring patches are served by the same L2
        const int patch = (blockIdx.x & 7) * A.xcd_chunk + (blockIdx.x >> 3);
        if (patch >= A.n_patches)
            return; // whole workgroup
        constexpr int NTH = 2 * PEK;
        const int lane = threadIdx.x;
        // every wavefront holds 32 elements, u in lanes 0-31 and v in lanes 32-63, so that a metric value is loaded once
        const int comp = (lane >> 5) & 1, le = (lane & 31) + 32 * (lane >> 6);
        const int ML = A.max_loc;
        // PEK = 32: xs = [2][ML] copy of x, ys = [2][ML] accumulator.  PEK = 64: every wavefront accumulates its 32 elements
        // in its OWN [2][ML] array, so that the two waves meet at three barriers only (after the gather, after the register
        // fill, before the write-out) instead of at every colour phase: wave 1 uses ys, wave 0 re-uses xs once both waves
        // have filled their registers from it; the write-out adds the two.
        constexpr bool TWO = PEK == 64;
        double *xs = lds;          // [2][ML]
        double *ys = lds + 2 * ML; // [2][ML]
        double *yw = (TWO && (lane >> 6) == 0) ? xs : ys; // this wavefront's accumulator

        // (fixed-stride lists: the first index requests do not wait for an offset load; `cap` = highest valid list index)
        const int nloc = A.dof_off[patch + 1] - A.dof_off[patch];
        const int off = A.dof_stride ? patch * A.dof_stride : A.dof_off[patch];
        const int cap = A.dof_stride ? A.dof_stride - 1 : nloc - 1;
        const int *dofs = A.dof_list + off;

        const double *Gp = A.Gp + (size_t)patch * 3 * NQS * NQS * PEK + le;
        const double *ap = A.aMp + (size_t)patch * NQM * NQM * PEK + le;

        // A wavefront's life is a chain of dependent memory round trips (about 2 us each under load); everything that does
        // not depend on the LDS copy of x is therefore requested up front: the element -> local-dof map, the colours, the
        // first metric slice, and the dof indices of the whole patch (384 per pass) before any x value.
        const bool active = le < A.patch_nel[patch];
        constexpr int NP = (NN + 1) / 2;
        const uint32_t *li = A.lidx + ((size_t)patch * NP) * PEK + le;
        // element node -> patch-local dof, two 16-bit indices per register, kept for the gather here and the
        // scatter below (lidx is padded to 32 lanes per patch, so inactive lanes read valid zeros)
        uint32_t lpk[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j)
            lpk[j] = li[j * PEK];
        const int mycol = active ? A.colour[patch * PEK + le] : -1;
        // n_basis 5, general layout: 3 waves/SIMD with 26 spilled registers measured 1-14 % faster than 2 waves without
        // (same-box A/B, 256^2 ... 1024^2 and the irregular mesh); the affine form loses 30 % that way and stays at 2.
        // n_basis 4, general layout: 153 VGPRs at 3 waves/SIMD (no spills, first slice prefetched) measured faster than
        // 128 VGPRs with 8 spilled at 4 waves/SIMD (1024^2: 402 vs 420 us; irregular 0.49 M quads: 187 vs 222 us)
        constexpr bool PAIRM = MODE == 1;
        constexpr bool PRE = !UG && NB <= 4;
        double g_first[3 * NQS];
        if constexpr (PRE)
        {
#pragma unroll
            for (int r = 0; r < NQS; ++r)
            {
                g_first[3 * r + 0] = metric_load<NT>(&Gp[((0 * 3 + 0) * NQS + r) * PEK]);
                g_first[3 * r + 1] = metric_load<NT>(&Gp[((0 * 3 + 1) * NQS + r) * PEK]);
                g_first[3 * r + 2] = metric_load<NT>(&Gp[((0 * 3 + 2) * NQS + r) * PEK]);
            }
        }

        constexpr int ROWS = PEK == 32 ? 6 : 5; // rows of NTH dofs per pass: 384 cover a 4x8-element patch of n_basis 4 (325), 640 an 8x8 one (625)
        // where the results of the first pass of the write-out go: requested here, with everything else that is independent
        // of the element phase, when the registers allow (EARLY), otherwise just before the colour phases
        constexpr bool EARLY = !UG && NB == 4;
        const int *slot = A.slot_of + off;
        // rows of NTH local dofs below own_rows hold owned dofs only (the plan numbers those first): their destination is the
        // gather index, so the destination list is read for the tail rows only (see op_patch_kernel)
        const int own_rows = A.own_count[patch] / NTH;
        int dest0[ROWS];
        if constexpr (EARLY && !NATIVE)
        {
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                if (j >= own_rows)
                    dest0[j] = slot[min(NTH * j + lane, cap)];
        }
        // native ordering: local dofs [0, nown) are this patch's owned dofs at native positions own0 + i; the border dofs behind
        // them take their native position (gather) and their slot (write-out) from the patch's two short lists
        int own0 = 0, nown = 0;
        const int *bp = nullptr, *bs = nullptr;
        const int bcap = A.bstride - 1;
        const dbl2_t *X2 = reinterpret_cast<const dbl2_t *>(A.x);
        if constexpr (NATIVE)
        {
            own0 = A.own_off[patch];
            nown = A.own_off[patch + 1] - own0;
            bp = A.bpos + (size_t)patch * A.bstride;
            bs = A.bslot + (size_t)patch * A.bstride;
            for (int base = 0; base == 0 || base < nloc; base += NTH * ROWS)
            {
                int pos[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + NTH * j + lane;
                    pos[j] = own0 + i;
                    if (base + NTH * j + NTH - 1 >= nown) // (workgroup-uniform) a row that holds border dofs
                        pos[j] = bp[max(0, min(i - nown, bcap))];
                }
                dbl2_t xv2[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + NTH * j + lane;
                    xv2[j] = X2[i < nown ? own0 + i : pos[j]];
                }
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + NTH * j + lane;
                    if (i < nloc)
                    {
                        xs[i] = xv2[j].x;
                        xs[ML + i] = xv2[j].y;
                        ys[i] = 0.0;
                        ys[ML + i] = 0.0;
                    }
                }
            }
        }
        else
        for (int base = 0; base == 0 || base < nloc; base += NTH * ROWS) // (the first pass does not wait for nloc)
        {
            int gi[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                gi[j] = dofs[min(base + NTH * j + lane, cap)];
            if constexpr (EARLY)
                if (base == 0)
#pragma unroll
                    for (int j = 0; j < ROWS; ++j)
                        if (j < own_rows)
                            dest0[j] = gi[j];
            double xu[ROWS], xv[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                xu[j] = A.x[gi[j]];
                xv[j] = A.x[A.ndof + gi[j]];
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + NTH * j + lane;
                if (i < nloc)
                {
                    xs[i] = xu[j];
                    xs[ML + i] = xv[j];
                    ys[i] = 0.0;
                    ys[ML + i] = 0.0;
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ element phase
        const double *xc = xs + comp * ML;
        auto lix_of = [&](int n) -> int { return (n & 1) ? static_cast<int>(lpk[n >> 1] >> 16) : static_cast<int>(lpk[n >> 1] & 0xFFFFu); };
        const double keep = active ? 1.0 : 0.0;
        // (n_basis 5, 3 wavefronts per SIMD: the 25-27 spilled registers are stored once before the slices and re-loaded once
        // after them -- values the element phase does not touch.  Streaming the element's values from the LDS copy of x in every
        // slice instead of holding them in registers removes 6 of them for 350 more LDS reads per lane: 768^2 a wash, irregular mesh
        // 5 % slower; measured and removed, profiles/r03/nb5_ab.txt.)
        double u[NN], out[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n)
        {
            u[n] = keep * xc[lix_of(n)];
            out[n] = 0.0;
        }

        if constexpr (TWO)
        {
            __syncthreads(); // both waves hold their u: xs becomes wave 0's accumulator
            if ((lane >> 6) == 0)
                for (int i = lane; i < nloc; i += 64)
                {
                    xs[i] = 0.0;
                    xs[ML + i] = 0.0;
                }
        }

        // One quadrature "slice" = all points with the same xi index q.
        // stiffness: out(k,l) += sum_q [ D(q,k) sum_r P(r,l) F0(q,r) + P(q,k) sum_r D(r,l) F1(q,r) ]
        auto load_stiff = [&](int q, double (&g)[3 * NQS])
        {
#pragma unroll
            for (int r = 0; r < NQS; ++r)
            {
                if constexpr (UG)
                {
                    g[3 * r + 0] = GU[(q * 3 + 0) * NQS + r];
                    g[3 * r + 1] = GU[(q * 3 + 1) * NQS + r];
                    g[3 * r + 2] = GU[(q * 3 + 2) * NQS + r];
                }
                else
                {
                    g[3 * r + 0] = metric_load<NT>(&Gp[((q * 3 + 0) * NQS + r) * PEK]);
                    g[3 * r + 1] = metric_load<NT>(&Gp[((q * 3 + 1) * NQS + r) * PEK]);
                    g[3 * r + 2] = metric_load<NT>(&Gp[((q * 3 + 2) * NQS + r) * PEK]);
                }
            }
        };
        auto stiff_slice = [&](int q, const double (&g)[3 * NQS])
        {
            double pu[NB], du[NB], t0[NB], t1[NB];
#pragma unroll
            for (int l = 0; l < NB; ++l)
            {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    a += PS[q + NQS * k] * u[k + NB * l];
                    b += DS[q + NQS * k] * u[k + NB * l];
                }
                pu[l] = a;
                du[l] = b;
                t0[l] = 0.0;
                t1[l] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < NQS; ++r)
            {
                const double ga = g[3 * r + 0], gb = g[3 * r + 1], gc = g[3 * r + 2];
                double dx = 0.0, dy = 0.0;
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    dx += PS[r + NQS * l] * du[l];
                    dy += DS[r + NQS * l] * pu[l];
                }
                const double f0 = ga * dx + gb * dy;
                const double f1 = gb * dx + gc * dy;
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    t0[l] += PS[r + NQS * l] * f0;
                    t1[l] += DS[r + NQS * l] * f1;
                }
            }
#pragma unroll
            for (int l = 0; l < NB; ++l)
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    out[k + NB * l] = fma(DS[q + NQS * k], t0[l], fma(PS[q + NQS * k], t1[l], out[k + NB * l]));
        };

        // mass: out(k,l) += -w^2 sum_q P(q,k) sum_r P(r,l) a(q,r) (sum_{k'l'} P(q,k') P(r,l') u(k',l'))
        const double w2 = -A.omega * A.omega;
        auto load_mass = [&](int q, double (&a)[NQM])
        {
#pragma unroll
            for (int r = 0; r < NQM; ++r)
                a[r] = metric_load<NT>(&ap[(q * NQM + r) * PEK]);
        };
        auto mass_slice = [&](int q, const double (&am)[NQM])
        {
            double pu[NB], t[NB];
#pragma unroll
            for (int l = 0; l < NB; ++l)
            {
                double a = 0.0;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    a += PM[q + NQM * k] * u[k + NB * l];
                pu[l] = a;
                t[l] = 0.0;
            }
#pragma unroll
            for (int r = 0; r < NQM; ++r)
            {
                double val = 0.0;
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    val += PM[r + NQM * l] * pu[l];
                val *= am[r] * w2;
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    t[l] += PM[r + NQM * l] * val;
            }
#pragma unroll
            for (int l = 0; l < NB; ++l)
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    out[k + NB * l] += PM[q + NQM * k] * t[l];
        };

        {
            // q stays a real loop: unrolling it lets the scheduler hoist every load of the element and exhausts the VGPRs
            if constexpr (PRE)
                stiff_slice(0, g_first);
#pragma unroll 1
            for (int q = PRE ? 1 : 0; q < NQS; ++q)
            {
                double g[3 * NQS];
                load_stiff(q, g);
                stiff_slice(q, g);
            }
            // two mass slices per round trip where their 2 NQM weights need no more registers than the 3 NQS metric values of a
            // stiffness slice, dead by now (n_basis 5: 18 = 18; n_basis 3: 12 = 12): NQS + ceil(NQM / 2) dependent round trips
            // instead of NQS + NQM
            if constexpr (PAIRM)
            {
#pragma unroll 1
                for (int q = 0; q + 1 < NQM; q += 2)
                {
                    double am0[NQM], am1[NQM];
                    load_mass(q, am0);
                    load_mass(q + 1, am1);
                    mass_slice(q, am0);
                    mass_slice(q + 1, am1);
                }
                if constexpr (NQM % 2 == 1)
                {
                    double am[NQM];
                    load_mass(NQM - 1, am);
                    mass_slice(NQM - 1, am);
                }
            }
            else
            {
#pragma unroll 1
                for (int q = 0; q < NQM; ++q)
                {
                    double am[NQM];
                    load_mass(q, am);
                    mass_slice(q, am);
                }
            }
        }

        if constexpr (!EARLY && !NATIVE)
        {
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest0[j] = (j < own_rows ? dofs : slot)[min(NTH * j + lane, nloc - 1)];
        }

        // accumulate: elements of one colour touch disjoint dofs
        if constexpr (NB == 5 && !UG && HELM_PATCH_RELOAD_MAP)
        {
            // n_basis 5 at 3 wavefronts per SIMD: the element -> local dof map (13 registers) is not used between the register fill
            // and here, and the compiler spills it across the slices (a scratch store and a scratch load per register and lane).
            // Re-reading it (cache-hot, 1.6 KB per patch) costs one load instead.
            const uint32_t *li2 = li;
            asm volatile("" : "+v"(li2)); // (opaque: a new load, not the values from kernel entry)
#pragma unroll
            for (int j = 0; j < NP; ++j)
                lpk[j] = li2[j * PEK];
        }
        {
            const double sgn = comp ? -1.0 : 1.0; // the v row is negated (symmetrised system)
            double *yc = yw + comp * ML;
            for (int c = 0; c < A.ncol; ++c)
            {
                if (mycol == c)
                {
                    // loads first, then adds and stores: see helm_lane_kernel's colour phases
                    double acc[NN];
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        acc[n] = yc[lix_of(n)];
                    asm volatile("" ::: "memory"); // all loads before all stores, also in the generated code (the scheduler
                    __builtin_amdgcn_sched_barrier(0); // otherwise re-serialises them to save registers)
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        yc[lix_of(n)] = acc[n] + sgn * out[n];
                }
                if constexpr (TWO)
                    __builtin_amdgcn_wave_barrier(); // the accumulator is private to the wave: LDS operations of one wave are in order
                else
                    __syncthreads();
            }
        }

        // ------------------------------------------------------------ boundary faces:  Au -= w H v,  Av -= w H u
        {
            const int f_begin = A.face_off[patch], nf = A.face_off[patch + 1] - f_begin;
            const double *xo = xs + (1 - comp) * ML; // the other component (PEK = 32)
            const double *xg = A.x + (size_t)(1 - comp) * A.ndof; // PEK = 64: xs is gone, the few face values come from global memory
            double *yc = yw + comp * ML;
            const int nqF = A.nqF;
            for (int f0 = 0; f0 < nf; f0 += PEK)
            {
                const int f = f0 + le;
                const bool fa = f < nf;
                double res[NB];
                int fl[NB];
                int fc = -1;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    res[k] = 0.0;
                    fl[k] = 0;
                }
                if (fa)
                {
                    const uint16_t *fli = A.face_lidx + (size_t)(f_begin + f) * NB;
                    const double *af = A.aF + (size_t)nqF * A.face_id[f_begin + f];
                    fc = A.face_col[f_begin + f];
                    double w[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                    {
                        fl[k] = fli[k];
                        if constexpr (NATIVE && TWO)
                        {
                            const dbl2_t t = X2[fl[k] < nown ? own0 + fl[k] : bp[fl[k] - nown]];
                            w[k] = comp ? t.x : t.y; // the other component
                        }
                        else
                            w[k] = TWO ? xg[dofs[fl[k]]] : xo[fl[k]];
                    }
                    for (int q = 0; q < nqF; ++q)
                    {
                        double pv = 0.0;
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            pv += PF[q + nqF * k] * w[k];
                        pv *= af[q];
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            res[k] += PF[q + nqF * k] * pv;
                    }
                }
                for (int c = 0; c < A.nfcol; ++c)
                {
                    if (fc == c)
                    {
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            yc[fl[k]] -= A.omega * res[k];
                    }
                    if constexpr (TWO)
                        __builtin_amdgcn_wave_barrier();
                    else
                        __syncthreads();
                }
            }
        }

        // ------------------------------------------------------------ write out
        if constexpr (TWO)
            __syncthreads(); // both accumulators complete
        auto result = [&](int i) -> double { return TWO ? xs[i] + ys[i] : ys[i]; };
        if constexpr (NATIVE)
        {
            dbl2_t *Y2 = reinterpret_cast<dbl2_t *>(A.y), *P2 = reinterpret_cast<dbl2_t *>(A.part);
            for (int i = lane; i < nloc; i += NTH)
            {
                dbl2_t r;
                r.x = result(i);
                r.y = result(ML + i);
                if (i < nown)
                    Y2[own0 + i] = r; // 16 bytes per lane, contiguous
                else
                    P2[bs[min(i - nown, bcap)]] = r;
            }
            return;
        }
        for (int base = 0; base < nloc; base += NTH * ROWS)
        {
            int dest[ROWS]; // one index per dof: global dof (owned) or -(slot) - 1 (border)
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest[j] = base == 0 ? dest0[j] : slot[min(base + NTH * j + lane, nloc - 1)];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + NTH * j + lane;
                if (i >= nloc)
                    continue;
                if (dest[j] >= 0)
                {
                    A.y[dest[j]] = result(i);
                    A.y[A.ndof + dest[j]] = result(ML + i);
                }
                else
                {
                    const int sl = -dest[j] - 1;
                    dbl2_t r; // a slot is the pair (u, v): one 16-byte store, one 16-byte load in the border kernel
                    r.x = result(i);
                    r.y = result(ML + i);
                    reinterpret_cast<dbl2_t *>(A.part)[sl] = r;
                }
            }
        }
    }

    // ---------------------------------------------------------------- fused complex apply, one element per lane, BOTH components
    // Patches of 64 elements, one wavefront per patch, lane = element: the lane applies the operators to u and then to v of
    // its element with the same metric values.  Against helm_patch_kernel (32 elements x 2 components per wavefront) every
    // metric load instruction fetches 512 distinct bytes instead of 256, so a wavefront keeps twice the bytes in flight per
    // round trip of its chain; the price is twice the element state per lane (2 waves per SIMD instead of 3).  The LDS copy of
    // x is consumed when the registers are filled and the same array then accumulates y (as in helm_mfma_kernel); the
    // boundary-face term re-reads its few x values from global memory.
    // UG: the stiffness metric is the same in every element and comes from the uniform table GU (scalar loads)
    // PRE ("whole patch in the register file"): one wavefront per SIMD owns all 512 registers per lane (256 VGPR + 256 AGPR,
    // loads can target either), so EVERY metric value of the patch -- 3 NQS^2 + NQM^2 doubles per lane, 139 for n_basis 4 = 278
    // registers, 71 KB per wavefront -- is requested before the gather and the slices then run out of registers without a single
    // further round trip.  The chain of a wavefront shrinks from 2 + NQS + NQM/2 dependent round trips to 3, and a CU keeps
    // 4 x 63 loads x 512 B = 129 KB in flight instead of 8 x 7.7 KB.  Measured on the metric stream alone
    // (profiles/tools/dma_stream.hip, profiles/r02/metric_stream_microbench.txt): slice-by-slice at 8 wavefronts per CU
    // 4.4 TB/s, whole block up front at 4 wavefronts per CU 6.46 TB/s (a read-only stream on that box: 6.9); the same block
    // through an LDS-DMA ring: 1.8-4.0 TB/s (one wavefront's LDS-DMA pieces are served one at a time, ~1.5 B/clk).
    // In the real kernel it LOSES (1024^2: 434-442 us against 373-376 us for the chain; 357 us even with the slice arithmetic
    // removed): with 4 x 32 KB per CU in flight (33 MB on the chip) every dependent round trip takes ~7 us, and a lone
    // wavefront per SIMD has nothing to overlap its light round trips (dof indices -> x values -> destination slots) with:
    // ~22 us per patch of exposed latency.  What limits these kernels is the NUMBER of dependent round trips a wavefront makes
    // times the loaded latency (= bytes in flight / bandwidth), so every round trip must carry a full share of the bytes.
    // Kept as a tested option (CUDDH_HELM_PRE=1); the default lane form is the chain with its light round trips merged.
    // Two further one-wavefront-per-SIMD forms were built on it, measured and removed again (profiles/r02/one_wave_per_simd.txt):
    // two patches per wavefront with the second patch's requests issued behind the first one's arithmetic (500 registers, no
    // spills once the slices' FMAs were pinned before the next stage's loads -- the compiler otherwise sinks them to the first
    // use of the result and keeps every array live: 400 spills) ran 427-463 us, and n_basis 5 in lane form with staged
    // prefetch 662 us against 428 us for helm_patch_kernel.  A lone wavefront issues one vector instruction every 4 cycles
    // instead of every 2: the arithmetic of a patch (8 us at n_basis 4) takes twice as long and nothing hides it.  A register
    // ring of three / four slices in helm_mfma_kernel (n_basis 6-8) also lost (spills at 168 registers: 190 -> 335 us).
    // NATIVE: x and y are in the plan's own vector ordering (pairs (u, v); a patch's owned dofs contiguous, cuddh_helmholtz_plan):
    // the gather of the owned dofs is one 16-byte load per dof at an address known at kernel entry -- no index list, no dependent
    // index -> value round trip -- and their write-out one 16-byte store; only the border dofs (15 % of an 8 x 8-element patch
    // at n_basis 4) go through lists (native position for the gather, slot for the write-out).
#ifndef HELM_LANE_FAST_GATHER
#define HELM_LANE_FAST_GATHER 1 // (measurement: 0 = the native gather as a loop, scalars through vector loads)
#endif
    template <int NB, int NQS, int NQM, bool NT, bool UG, bool PRE = false, bool NATIVE = false>
    __global__ void __launch_bounds__(64, PRE ? 1 : (NB == 2 ? 5 : (NB == 3 ? 3 : 2))) helm_lane_kernel(HelmArgs A, const double *__restrict__ PS, const double *__restrict__ DS,
                                                              const double *__restrict__ PM, const double *__restrict__ PF,
                                                              const double *__restrict__ GU)
    {
        constexpr int NN = NB * NB, NP = (NN + 1) / 2, PEK = 64;
        extern __shared__ double lds[];
        const int patch = (blockIdx.x & 7) * A.xcd_chunk + (blockIdx.x >> 3);
        if (patch >= A.n_patches)
            return; // whole workgroup
        const int lane = threadIdx.x;
        const int ML = A.max_loc;
        double *xy = lds; // [2][ML]: the gathered x, then the accumulated y
        // diagnostic phase stamps (wave-uniform branch; the 100 MHz constant clock; every stamp drains the wavefront's memory
        // operations first and is a compiler barrier, so a phase's time includes the completion of what it issued)
        auto stamp = [&](int k)
        {
            if (A.stamps)
            {
                unsigned long long t;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                if (lane == 0)
                    A.stamps[(size_t)patch * 8 + k] = t;
            }
        };
        stamp(0);

        // (fixed-stride lists: the first index requests do not wait for an offset load; `cap` = highest valid list index)
        // FASTL (native ordering): the patch's scalars through the scalar cache and the gather written straight-line (see below)
        constexpr bool FASTL = NATIVE && !PRE && HELM_LANE_FAST_GATHER;
        int nloc, off, nel_s = 0, own0 = 0, nown = 0;
        if constexpr (FASTL)
        {
            unsigned long long d2, o2;
            asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(d2), "=&s"(o2), "=&s"(nel_s)
                         : "s"(A.dof_off + patch), "s"(A.own_off + patch), "s"(A.patch_nel + patch));
            nloc = static_cast<int>(d2 >> 32) - static_cast<int>(d2 & 0xFFFFFFFFull);
            off = A.dof_stride ? patch * A.dof_stride : static_cast<int>(d2 & 0xFFFFFFFFull);
            own0 = static_cast<int>(o2 & 0xFFFFFFFFull);
            nown = static_cast<int>(o2 >> 32) - own0;
        }
        else
        {
            nloc = A.dof_off[patch + 1] - A.dof_off[patch];
            off = A.dof_stride ? patch * A.dof_stride : A.dof_off[patch];
        }
        const int cap = A.dof_stride ? A.dof_stride - 1 : nloc - 1;
        const int *dofs = A.dof_list + off;
        const double *Gp = A.Gp + (size_t)patch * 3 * NQS * NQS * PEK; // pair layout, see load_pairs
        const double *ap = A.aMp + (size_t)patch * NQM * NQM * PEK;

        constexpr int ROWS = 10; // 640 dofs per pass: an 8x8-element patch of n_basis 4 (625) in one
        const dbl2_t *X2 = reinterpret_cast<const dbl2_t *>(A.x);
        const int *bp = nullptr, *bs = nullptr;
        const int bcap = A.bstride - 1;
        // FASTL, oldest requests first (the counter is in order): the native positions of the first 128 border dofs, then the owned
        // rows of x -- before anything else is asked for, and not inside a loop (its header would wait for whatever is outstanding)
        int pbL[FASTL ? 2 : 1];
        dbl2_t xoL[FASTL ? ROWS : 1];
        if constexpr (FASTL)
        {
            bp = A.bpos + (size_t)patch * A.bstride;
            bs = A.bslot + (size_t)patch * A.bstride;
            pbL[0] = bp[min(lane, bcap)];
            pbL[1] = bp[min(64 + lane, bcap)];
            const int last = own0 + max(nown - 1, 0);
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                xoL[j] = X2[min(own0 + 64 * j + lane, last)];
        }
        bool active;
        if constexpr (FASTL)
            active = lane < nel_s;
        else
            active = lane < A.patch_nel[patch];
        const uint32_t *li = A.lidx + ((size_t)patch * NP) * PEK + lane;
        uint32_t lpk[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j)
            lpk[j] = li[j * PEK];
        int mycol;
        if constexpr (FASTL)
        {
            const int col_raw = A.colour[patch * PEK + lane]; // (padded to 64 lanes per patch)
            mycol = active ? col_raw : -1;
        }
        else
            mycol = active ? A.colour[patch * PEK + lane] : -1;
        auto load_stiff = [&](int q, double (&g)[3 * NQS])
        {
            if constexpr (UG)
            {
#pragma unroll
                for (int r = 0; r < NQS; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        g[3 * r + c] = GU[(q * 3 + c) * NQS + r];
            }
            else
            {
                double v[3 * NQS]; // memory order of a slice: v = c * NQS + r
                load_pairs<3 * NQS, NT>(Gp + (size_t)q * 3 * NQS * PEK, lane, v);
#pragma unroll
                for (int r = 0; r < NQS; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        g[3 * r + c] = v[c * NQS + r];
            }
        };
        auto load_mass = [&](int q, double (&am)[NQM]) { load_pairs<NQM, NT>(ap + (size_t)q * NQM * PEK, lane, am); };
        // The chain form requests the first TWO stiffness slices before the gather: the element state (u, out: 8 NB^2 registers)
        // is not live yet, so the registers are there, and the two light round trips of the gather (dof indices, then x
        // values) each carry a slice instead of one of them carrying nothing.  The write-out's destination indices ride
        // along too (below), so the wavefront makes two dependent round trips fewer than it used to.
        // (n_basis 4 only: at n_basis 2 and 3 the launch bounds hold 5 and 3 wavefronts per SIMD and the extra live registers spill)
        constexpr bool TWO_AHEAD = !UG && !PRE && NB >= 4;
        constexpr bool EARLY_DEST = PRE || NB >= 4;
        double g_first[3 * NQS], g_second[3 * NQS];
        // NATIVE (n_basis 4): the x values are requested FIRST and the slices behind them (native_gather below): loads return in
        // order, and with the addresses of the owned dofs known at kernel entry nothing has to arrive before x is asked for -- the
        // LDS copy of x is complete when 10 KB have arrived, not 28 (profiles/r03/lane_stamps_native.txt: 10 of a wavefront's 31 us
        // passed before that), and the slices land while the registers are being filled
        // (same-box A/B, profiles/r03/native_gather_ab.txt: structured 1024^2 within 1 % of the slices-first order -- the kernel moves
        // its bytes at the rate the memory system gives 2,048 resident wavefronts either way -- irregular 487k-quad mesh 143 -> 140 us)
        constexpr bool X_FIRST = NATIVE && !UG && !PRE && NB >= 4;
        if constexpr (!UG && !PRE && !X_FIRST)
            load_stiff(0, g_first);
        if constexpr (TWO_AHEAD && !X_FIRST)
            load_stiff(1, g_second);
        // PRE: the whole metric block of the patch, requested in this order: dof indices (above), stiffness metric, x values
        // of the first gather pass, mass weights -- so that what the chain needs first is oldest (loads return in order)
        double gS[PRE ? NQS : 1][3 * NQS];
        double aW[PRE ? NQM : 1][NQM];

        static_assert(!(NATIVE && PRE), "the native ordering is implemented for the chain form");
        // native ordering: local dofs [0, nown) are this patch's owned dofs at native positions own0 + i; the border dofs behind
        // them take their native position (gather) and their slot (write-out) from the patch's two short lists
        if constexpr (NATIVE && !FASTL)
        {
            own0 = A.own_off[patch];
            nown = A.own_off[patch + 1] - own0;
            bp = A.bpos + (size_t)patch * A.bstride;
            bs = A.bslot + (size_t)patch * A.bstride;
        }
        // Where the write-out sends its results (slot_of: the global dof of an owned local dof, -(slot) - 1 for a border dof).
        // The plan numbers a patch's owned dofs first, so for the rows of 64 local dofs below j_own = own_count / 64 the
        // destination IS the gather index and slot_of is not read at all: only its tail is (the rows holding border dofs),
        // requested with the other indices at kernel entry when the registers allow (EARLY_DEST).
        const int *slot = A.slot_of + off;
        const int j_own = NATIVE ? (nown >> 6) : (A.own_count[patch] >> 6); // wave-uniform
        int dest0[ROWS];
        if constexpr (EARLY_DEST && !NATIVE)
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                if (j >= j_own)
                    dest0[j] = slot[min(64 * j + lane, cap)];
        // Native gather, written WITHOUT wave-uniform branches (a conditional load is a basic block of its own to the compiler, which
        // then waits for every outstanding load at its end): the owned dofs are rows of 64 contiguous pairs at own0 + i (addresses
        // clamped, writes predicated), the border dofs a short list of native positions handled on its own.
        const int nbord = nloc - nown;
        int bslot_early[2] = {0, 0}; // slots of the first 128 border dofs (an 8 x 8-element patch of n_basis 4 has 96-100), requested early
        if constexpr (NATIVE)
        {
            bslot_early[0] = bs[min(lane, bcap)];
            bslot_early[1] = bs[min(64 + lane, bcap)];
        }
        auto native_gather = [&]()
        {
            const int last = max(nown - 1, 0);
            for (int base = 0; base == 0 || base < nown; base += 64 * ROWS)
            {
                dbl2_t xv2[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                    xv2[j] = X2[own0 + min(base + 64 * j + lane, last)];
                if (base == 0)
                {
                    // the border dofs: native position from the list, then the value (a dependent pair of light round trips that
                    // starts with the owned rows' requests already under way)
                    for (int t0 = 0; t0 < nbord; t0 += 64)
                    {
                        const int t = t0 + lane;
                        const dbl2_t xb = X2[bp[min(t, bcap)]];
                        if (t < nbord)
                        {
                            xy[nown + t] = xb.x;
                            xy[ML + nown + t] = xb.y;
                        }
                    }
                    if constexpr (X_FIRST)
                    {
                        // behind the x requests: the first two stiffness slices
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        load_stiff(0, g_first);
                        if constexpr (TWO_AHEAD)
                            load_stiff(1, g_second);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + 64 * j + lane;
                    if (i < nown)
                    {
                        xy[i] = xv2[j].x;
                        xy[ML + i] = xv2[j].y;
                    }
                }
            }
        };
        auto gather_pass = [&](int base, auto first)
        {
            constexpr bool WITH_METRIC = PRE && decltype(first)::value;
            int gi[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                gi[j] = dofs[min(base + 64 * j + lane, cap)];
            if constexpr (EARLY_DEST && decltype(first)::value)
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                    if (j < j_own)
                        dest0[j] = gi[j];
            if constexpr (WITH_METRIC)
            {
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!UG)
                {
#pragma unroll
                    for (int q = 0; q < NQS; ++q)
                        load_stiff(q, gS[q]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            double xu[ROWS], xv[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                xu[j] = A.x[gi[j]];
                xv[j] = A.x[A.ndof + gi[j]];
            }
            if constexpr (WITH_METRIC)
            {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQM; ++q)
                    load_mass(q, aW[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + 64 * j + lane;
                if (i < nloc)
                {
                    xy[i] = xu[j];
                    xy[ML + i] = xv[j];
                }
            }
        };
        if constexpr (FASTL)
        {
            // straight-line: border values behind their positions, the first two slices behind those, the values pinned before
            // the guarded LDS writes (the compiler would sink a row's request into its guard), then whatever a larger patch has left
            dbl2_t xb[2];
            xb[0] = X2[pbL[0]];
            xb[1] = X2[pbL[1]];
            if constexpr (X_FIRST)
            {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                load_stiff(0, g_first);
                if constexpr (TWO_AHEAD)
                    load_stiff(1, g_second);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                asm volatile("" : "+v"(xoL[j].x), "+v"(xoL[j].y));
            asm volatile("" : "+v"(xb[0].x), "+v"(xb[0].y), "+v"(xb[1].x), "+v"(xb[1].y));
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = 64 * j + lane;
                if (i < nown)
                {
                    xy[i] = xoL[j].x;
                    xy[ML + i] = xoL[j].y;
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
            {
                const int t = 64 * r + lane;
                if (t < nbord)
                {
                    xy[nown + t] = xb[r].x;
                    xy[ML + nown + t] = xb[r].y;
                }
            }
            for (int i = 64 * ROWS + lane; i < nown; i += 64) // a patch larger than the regular block (irregular meshes)
            {
                const dbl2_t t2 = X2[own0 + i];
                xy[i] = t2.x;
                xy[ML + i] = t2.y;
            }
            for (int t = 128 + lane; t < nbord; t += 64)
            {
                const dbl2_t t2 = X2[bp[min(t, bcap)]];
                xy[nown + t] = t2.x;
                xy[ML + nown + t] = t2.y;
            }
        }
        else if constexpr (NATIVE)
            native_gather();
        else
        {
            gather_pass(0, std::true_type{});
            for (int base = 64 * ROWS; base < nloc; base += 64 * ROWS)
                gather_pass(base, std::false_type{});
        }
        __syncthreads();
        stamp(1); // x is in LDS

        auto lix_of = [&](int n) -> int { return (n & 1) ? static_cast<int>(lpk[n >> 1] >> 16) : static_cast<int>(lpk[n >> 1] & 0xFFFFu); };
        const double keep = active ? 1.0 : 0.0;
        double u[2][NN], out[2][NN];
#pragma unroll
        for (int n = 0; n < NN; ++n)
        {
            const int l = lix_of(n);
            u[0][n] = keep * xy[l];
            u[1][n] = keep * xy[ML + l];
            out[0][n] = 0.0;
            out[1][n] = 0.0;
        }
        __syncthreads();
        for (int i = lane; i < nloc; i += 64)
        {
            xy[i] = 0.0;
            xy[ML + i] = 0.0;
        }
        __syncthreads();
        stamp(2); // element values in registers

        auto stiff_slice = [&](int q, const double (&g)[3 * NQS])
        {
#pragma unroll
            for (int c = 0; c < 2; ++c)
            {
                double pu[NB], du[NB], t0[NB], t1[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                    {
                        a += PS[q + NQS * k] * u[c][k + NB * l];
                        b += DS[q + NQS * k] * u[c][k + NB * l];
                    }
                    pu[l] = a;
                    du[l] = b;
                    t0[l] = 0.0;
                    t1[l] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < NQS; ++r)
                {
                    const double ga = g[3 * r + 0], gb = g[3 * r + 1], gc = g[3 * r + 2];
                    double dx = 0.0, dy = 0.0;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                    {
                        dx += PS[r + NQS * l] * du[l];
                        dy += DS[r + NQS * l] * pu[l];
                    }
                    const double f0 = ga * dx + gb * dy;
                    const double f1 = gb * dx + gc * dy;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                    {
                        t0[l] += PS[r + NQS * l] * f0;
                        t1[l] += DS[r + NQS * l] * f1;
                    }
                }
#pragma unroll
                for (int l = 0; l < NB; ++l)
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        out[c][k + NB * l] = fma(DS[q + NQS * k], t0[l], fma(PS[q + NQS * k], t1[l], out[c][k + NB * l]));
            }
        };
        const double w2 = -A.omega * A.omega;
        auto mass_slice = [&](int q, const double (&am)[NQM])
        {
#pragma unroll
            for (int c = 0; c < 2; ++c)
            {
                double pu[NB], t[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        a += PM[q + NQM * k] * u[c][k + NB * l];
                    pu[l] = a;
                    t[l] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < NQM; ++r)
                {
                    double val = 0.0;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        val += PM[r + NQM * l] * pu[l];
                    val *= am[r] * w2;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        t[l] += PM[r + NQM * l] * val;
                }
#pragma unroll
                for (int l = 0; l < NB; ++l)
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        out[c][k + NB * l] += PM[q + NQM * k] * t[l];
            }
        };

        if constexpr (PRE)
        {
            // every metric value is already on its way (or here): the slices run straight through, fully unrolled so that the
            // arrays stay in registers
#pragma unroll
            for (int q = 0; q < NQS; ++q)
            {
                if constexpr (UG)
                {
                    double g[3 * NQS];
                    load_stiff(q, g);
                    stiff_slice(q, g);
                }
                else
                    stiff_slice(q, gS[q]);
            }
#pragma unroll
            for (int q = 0; q < NQM; ++q)
                mass_slice(q, aW[q]);
        }
        if constexpr (!UG && !PRE)
            stiff_slice(0, g_first);
        if constexpr (TWO_AHEAD)
            stiff_slice(1, g_second);
#pragma unroll 1
        for (int q = PRE ? NQS : (UG ? 0 : (TWO_AHEAD ? 2 : 1)); q < NQS; ++q)
        {
            double g[3 * NQS];
            load_stiff(q, g);
            stiff_slice(q, g);
        }
        // two mass slices per round trip: 2 x NQM values are no more registers than the 3 x NQS of a stiffness slice, which
        // are dead by now, and the chain of dependent round trips shrinks from NQS + NQM to NQS + NQM / 2
#pragma unroll 1
        for (int q = PRE ? NQM : 0; q + 1 < NQM; q += 2)
        {
            double am0[NQM], am1[NQM];
            load_mass(q, am0);
            load_mass(q + 1, am1);
            mass_slice(q, am0);
            mass_slice(q + 1, am1);
        }
        if constexpr (NQM % 2 == 1 && !PRE)
        {
            double am[NQM];
            load_mass(NQM - 1, am);
            mass_slice(NQM - 1, am);
        }

        if constexpr (!EARLY_DEST && !NATIVE) // (the owned rows re-read their gather indices: lines the gather brought on chip)
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest0[j] = (j >= j_own ? slot : dofs)[min(64 * j + lane, nloc - 1)];

        if (A.stamps)
        {
#pragma unroll
            for (int n = 0; n < NN; ++n)
                asm volatile("" : "+v"(out[0][n]), "+v"(out[1][n])::"memory"); // the slices' arithmetic is finished
        }
        stamp(3); // slices done
        // accumulate in colour phases; the v row is negated (symmetrised system)
        for (int c = 0; c < A.ncol; ++c)
        {
            if (mycol == c)
            {
                // Read every value first, then add and store.  The nodes of an element are distinct dofs, which the compiler
                // cannot know: written as `xy[l] += ...` per node it waits for each store before the next load -- 32 LDS
                // round trips in a row, 5.2 us of a wavefront's 36 us (phase stamps, profiles/r02/lane_stamps_before.txt;
                // 2.2 us afterwards, same-box A/B of the whole apply 389 -> 380 us).  The scheduler re-serialises the batch to
                // save registers unless fenced.  (n_basis 2 and 3 keep the serial form: their launch bounds leave no registers.)
                if constexpr (NB >= 4 || PRE)
                {
                    double au[NN], av[NN];
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                    {
                        const int l = lix_of(n);
                        au[n] = xy[l];
                        av[n] = xy[ML + l];
                    }
                    asm volatile("" ::: "memory"); // all loads before all stores, also in the generated code (the scheduler
                    __builtin_amdgcn_sched_barrier(0); // otherwise re-serialises them to save registers)
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                    {
                        const int l = lix_of(n);
                        xy[l] = au[n] + out[0][n];
                        xy[ML + l] = av[n] - out[1][n];
                    }
                }
                else
                {
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                    {
                        const int l = lix_of(n);
                        xy[l] += out[0][n];
                        xy[ML + l] -= out[1][n];
                    }
                }
            }
            __syncthreads();
        }

        stamp(4); // colour phases done
        // boundary faces:  Au -= w H v,  Av -= w H u  (lane = face, both rows)
        {
            const int f_begin = A.face_off[patch], nf = A.face_off[patch + 1] - f_begin;
            const int nqF = A.nqF;
            for (int f0 = 0; f0 < nf; f0 += 64)
            {
                const int f = f0 + lane;
                const bool fa = f < nf;
                double ru[NB], rv[NB];
                int fl[NB];
                int fc = -1;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    ru[k] = 0.0;
                    rv[k] = 0.0;
                    fl[k] = 0;
                }
                if (fa)
                {
                    const uint16_t *fli = A.face_lidx + (size_t)(f_begin + f) * NB;
                    const double *af = A.aF + (size_t)nqF * A.face_id[f_begin + f];
                    fc = A.face_col[f_begin + f];
                    double wu[NB], wv[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                    {
                        fl[k] = fli[k];
                        if constexpr (NATIVE)
                        {
                            const dbl2_t t = X2[fl[k] < nown ? own0 + fl[k] : bp[fl[k] - nown]];
                            wu[k] = t.x;
                            wv[k] = t.y;
                        }
                        else
                        {
                            const int gd = dofs[fl[k]];
                            wu[k] = A.x[gd];
                            wv[k] = A.x[A.ndof + gd];
                        }
                    }
                    for (int q = 0; q < nqF; ++q)
                    {
                        double pu = 0.0, pv = 0.0;
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                        {
                            pu += PF[q + nqF * k] * wu[k];
                            pv += PF[q + nqF * k] * wv[k];
                        }
                        pu *= af[q];
                        pv *= af[q];
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                        {
                            ru[k] += PF[q + nqF * k] * pv; // the u row takes H v
                            rv[k] += PF[q + nqF * k] * pu; // the v row takes H u
                        }
                    }
                }
                for (int c = 0; c < A.nfcol; ++c)
                {
                    if (fc == c)
                    {
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                        {
                            xy[fl[k]] -= A.omega * ru[k];
                            xy[ML + fl[k]] -= A.omega * rv[k];
                        }
                    }
                    __syncthreads();
                }
            }
        }

        stamp(5); // faces done
        // write out
        if constexpr (NATIVE)
        {
            dbl2_t *Y2 = reinterpret_cast<dbl2_t *>(A.y), *P2 = reinterpret_cast<dbl2_t *>(A.part);
            for (int t0 = 0; t0 < nbord; t0 += 64) // border dofs to their slots
            {
                const int t = t0 + lane;
                const int sl = t0 == 0 ? bslot_early[0] : (t0 == 64 ? bslot_early[1] : bs[min(t, bcap)]);
                if (t < nbord)
                {
                    dbl2_t r;
                    r.x = xy[nown + t];
                    r.y = xy[ML + nown + t];
                    P2[sl] = r;
                }
            }
            for (int base = 0; base < nown; base += 64 * ROWS)
            {
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + 64 * j + lane;
                    if (i < nown)
                    {
                        dbl2_t r;
                        r.x = xy[i];
                        r.y = xy[ML + i];
                        Y2[own0 + i] = r; // 16 bytes per lane, 1 KiB contiguous per instruction
                    }
                }
            }
            stamp(6);
            return;
        }
        for (int base = 0; base < nloc; base += 64 * ROWS)
        {
            int dest[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest[j] = base == 0 ? dest0[j] : slot[min(base + 64 * j + lane, nloc - 1)];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + 64 * j + lane;
                if (i >= nloc)
                    continue;
                if (dest[j] >= 0)
                {
                    A.y[dest[j]] = xy[i];
                    A.y[A.ndof + dest[j]] = xy[ML + i];
                }
                else
                {
                    const int sl = -dest[j] - 1;
                    dbl2_t r; // a slot is the pair (u, v)
                    r.x = xy[i];
                    r.y = xy[ML + i];
                    reinterpret_cast<dbl2_t *>(A.part)[sl] = r;
                }
            }
        }
        stamp(6); // write-out done (stores complete: the stamp drains them)
    }

    __global__ void __launch_bounds__(256) helm_border_kernel(int n_shared, int ndof, int n_slots, const int *__restrict__ shared_dof,
                                                             const int *__restrict__ shared_off,
                                                             const double *__restrict__ part, double *__restrict__ y)
    {
        const dbl2_t *p2 = reinterpret_cast<const dbl2_t *>(part); // a slot is the pair (u, v)
        for (int j = blockIdx.x * 256 + threadIdx.x; j < n_shared; j += gridDim.x * 256)
        {
            // the slots of one dof are contiguous, in patch order; the first four are requested together (clamped addresses) and
            // added in that order under a select -- one trip to memory instead of one per slot (a dof has 2-4 slots almost always)
            const int t0 = shared_off[j], t1 = shared_off[j + 1], last = max(t1 - 1, t0);
            dbl2_t q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = p2[min(t0 + k, last)];
            double su = 0.0, sv = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                su = t0 + k < t1 ? su + q[k].x : su;
                sv = t0 + k < t1 ? sv + q[k].y : sv;
            }
            for (int t = t0 + 4; t < t1; ++t)
            {
                const dbl2_t s = p2[t];
                su += s.x;
                sv += s.y;
            }
            const int g = shared_dof[j];
            y[g] = su;
            y[ndof + g] = sv;
        }
    }

    // native ordering: the border dofs are the tail of the vector, in shared_dof order -- contiguous 16-byte stores
    __global__ void __launch_bounds__(256) helm_border_native_kernel(int n_shared, int n_owned, const int *__restrict__ shared_off,
                                                                    const dbl2_t *__restrict__ part, dbl2_t *__restrict__ y)
    {
        for (int j = blockIdx.x * 256 + threadIdx.x; j < n_shared; j += gridDim.x * 256)
        {
            // (as helm_border_kernel: same slots, same order, the first four requested together)
            const int t0 = shared_off[j], t1 = shared_off[j + 1], last = max(t1 - 1, t0);
            dbl2_t q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = part[min(t0 + k, last)];
            dbl2_t s = {0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                s.x = t0 + k < t1 ? s.x + q[k].x : s.x;
                s.y = t0 + k < t1 ? s.y + q[k].y : s.y;
            }
            for (int t = t0 + 4; t < t1; ++t)
            {
                s.x += part[t].x;
                s.y += part[t].y;
            }
            y[n_owned + j] = s;
        }
    }

    // reference ordering [u; v] <-> plan-native ordering (pairs), through the permutation global_of_native
    __global__ void __launch_bounds__(256) to_native_kernel(int ndof, const int *__restrict__ global_of_native, const double *__restrict__ x,
                                                           dbl2_t *__restrict__ z)
    {
        for (int n = blockIdx.x * 256 + threadIdx.x; n < ndof; n += gridDim.x * 256)
        {
            const int g = global_of_native[n];
            dbl2_t t;
            t.x = x[g];
            t.y = x[ndof + g];
            z[n] = t;
        }
    }

    __global__ void __launch_bounds__(256) from_native_kernel(int ndof, const int *__restrict__ global_of_native, const dbl2_t *__restrict__ z,
                                                             double *__restrict__ y)
    {
        for (int n = blockIdx.x * 256 + threadIdx.x; n < ndof; n += gridDim.x * 256)
        {
            const int g = global_of_native[n];
            const dbl2_t t = z[n];
            y[g] = t.x;
            y[ndof + g] = t.y;
        }
    }

    template <typename T>
    int upload(T **dst, const std::vector<T> &v)
    {
        *dst = nullptr;
        if (v.empty())
            return 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(dst), v.size() * sizeof(T));
        if (e != hipSuccess)
            return static_cast<int>(e);
        return static_cast<int>(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    }

    // does every element's block of `per_elem` values equal element 0's, to the absolute tolerance tol?
    __global__ void __launch_bounds__(256) uniform_metric_kernel(long long total, int per_elem, const double *__restrict__ src, double tol,
                                                                int *__restrict__ differs)
    {
        for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL)
            if (fabs(src[t] - src[t % per_elem]) > tol)
                atomicExch(differs, 1);
    }

    // If the reference-layout metric array (comps, nq, nq, n_elem) is the same for every element (to 1e-13 of its largest
    // entry), upload one copy as [q][c][r] and return it in *table; otherwise leave *table null.
    int uniform_table(double **table, int comps, int nq, int n_elem, const double *d_src)
    {
        *table = nullptr;
        const int per_elem = comps * nq * nq;
        std::vector<double> block(per_elem);
        hipError_t e = hipMemcpy(block.data(), d_src, per_elem * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess)
            return static_cast<int>(e);
        double scale = 0.0;
        for (double v : block)
            scale = std::max(scale, std::fabs(v));
        int *flag = nullptr, differs = 1;
        e = hipMalloc(reinterpret_cast<void **>(&flag), sizeof(int));
        if (e == hipSuccess)
            e = hipMemset(flag, 0, sizeof(int));
        if (e == hipSuccess)
        {
            const long long total = (long long)per_elem * n_elem;
            hipLaunchKernelGGL(uniform_metric_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, nullptr, total, per_elem, d_src,
                               1e-13 * scale, flag);
            e = hipMemcpy(&differs, flag, sizeof(int), hipMemcpyDeviceToHost);
        }
        if (flag)
            (void)hipFree(flag);
        if (e != hipSuccess)
            return static_cast<int>(e);
        if (differs)
            return 0;
        std::vector<double> t(per_elem);
        for (int q = 0; q < nq; ++q)
            for (int c = 0; c < comps; ++c)
                for (int r = 0; r < nq; ++r)
                    t[(q * comps + c) * nq + r] = block[c + comps * (q + nq * r)];
        return upload(table, t);
    }

    // reference layout (c, q, r, el) -> [batch][r][c][q][16]  (matrix-core kernels: slices over the eta index r)
    __global__ void __launch_bounds__(256) repack_mfma_kernel(long long total, int comps, int nq, const int *__restrict__ perm,
                                                             const double *__restrict__ src, double *__restrict__ dst)
    {
        for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL)
        {
            const int le = static_cast<int>(t % 16);
            long long rest = t / 16;
            const int q = static_cast<int>(rest % nq);
            rest /= nq;
            const int c = static_cast<int>(rest % comps);
            rest /= comps;
            const int r = static_cast<int>(rest % nq);
            const long long batch = rest / nq;
            const int el = perm[batch * 16 + le];
            dst[t] = el >= 0 ? src[c + (size_t)comps * ((q + (size_t)nq * r) + (size_t)nq * nq * el)] : 0.0;
        }
    }

    // reference layout (c, q, r, el) -> [patch][q][c][r][32]
    // pair != 0: within a slice (patch, q) the comps*nq values v = c*nq + r are stored as pairs, [v/2][pe lanes][2], the last one
    // of an odd count as a [pe lanes] row behind them (load_pairs)
    __global__ void __launch_bounds__(256) repack_kernel(long long total, int comps, int nq, int pe, const int *__restrict__ perm,
                                                        const double *__restrict__ src, double *__restrict__ dst, int pair)
    {
        for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += gridDim.x * 256LL)
        {
            const int le = static_cast<int>(t % pe);
            long long rest = t / pe;
            const int r = static_cast<int>(rest % nq);
            rest /= nq;
            const int c = static_cast<int>(rest % comps);
            rest /= comps;
            const int q = static_cast<int>(rest % nq);
            const long long patch = rest / nq;
            const int el = perm[patch * pe + le];
            const double val = el >= 0 ? src[c + (size_t)comps * ((q + (size_t)nq * r) + (size_t)nq * nq * el)] : 0.0;
            if (!pair)
                dst[t] = val;
            else
            {
                const int nv = comps * nq, v = c * nq + r, vp = nv / 2;
                const long long slice = (patch * nq + q) * (long long)nv * pe;
                dst[v < 2 * vp ? slice + ((long long)(v / 2) * pe + le) * 2 + (v & 1) : slice + (long long)2 * vp * pe + le] = val;
            }
        }
    }

    int upload_raw(double **dst, const double *src, size_t n)
    {
        std::vector<double> v(src, src + n);
        return upload(dst, v);
    }

    inline uint32_t spread_bits(uint32_t v)
    {
        v &= 0xFFFF;
        v = (v | (v << 8)) & 0x00FF00FF;
        v = (v | (v << 4)) & 0x0F0F0F0F;
        v = (v | (v << 2)) & 0x33333333;
        v = (v | (v << 1)) & 0x55555555;
        return v;
    }

    template <int NB, int NQS, int NQM, int PEK>
    void launch_patch_pe(const cuddh_helmholtz_plan *p, const HelmArgs &A, hipStream_t st, bool native = false)
    {
        const size_t lds = (size_t)4 * p->max_loc * sizeof(double);
        const dim3 grid(8 * A.xcd_chunk), block(2 * PEK);
        if (native) // the same MODE as the reference-ordering launch below, so that the two orderings give bitwise the same numbers
        {
            constexpr bool CAN_PAIR_N = 2 * NQM <= 3 * NQS && NB <= 4;
            if constexpr (CAN_PAIR_N)
                if (!p->Gu && A.pair_mass)
                {
                    if (p->streaming)
                        hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, false, PEK, 1, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else
                        hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, false, PEK, 1, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    return;
                }
            if (p->Gu && p->streaming)
                hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, true, PEK, 0, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
            else if (p->Gu)
                hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, true, PEK, 0, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
            else if (p->streaming)
                hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, false, PEK, 0, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
            else
                hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, false, PEK, 0, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
            return;
        }
        // general layout, 2 NQM <= 3 NQS (n_basis 3 and 5): the mass phase takes two slices per round trip (PM)
        constexpr bool CAN_PAIR = 2 * NQM <= 3 * NQS && NB <= 4;
        if (p->Gu && p->streaming)
            hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, true, PEK>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
        else if (p->Gu)
            hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, true, PEK>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
        else if (CAN_PAIR && A.pair_mass)
        {
            if constexpr (CAN_PAIR)
            {
                if (p->streaming)
                    hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, false, PEK, 1>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                else
                    hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, false, PEK, 1>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
            }
        }
        else if (p->streaming)
            hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, true, false, PEK>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
        else
            hipLaunchKernelGGL((helm_patch_kernel<NB, NQS, NQM, false, false, PEK>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
    }

    template <int NB, int NQS, int NQM>
    void launch_patch(const cuddh_helmholtz_plan *p, const HelmArgs &A, hipStream_t st, bool native = false)
    {
        if constexpr (NB <= 4)
            if (p->pe == 64 && p->lane_form)
            {
                const size_t lds = (size_t)2 * p->max_loc * sizeof(double);
                const dim3 grid(8 * A.xcd_chunk), block(64);
                if (native)
                {
                    if (p->Gu && p->streaming)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, true, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else if (p->Gu)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, true, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else if (p->streaming)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, false, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, false, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    return;
                }
                if (p->prefetch)
                {
                    // whole-patch prefetch: one wavefront per SIMD, the patch's metric block requested up front
                    if (p->Gu && p->streaming)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, true, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else if (p->Gu)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, true, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else if (p->streaming)
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    else
                        hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                    return;
                }
                if (p->Gu && p->streaming)
                    hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                else if (p->Gu)
                    hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                else if (p->streaming)
                    hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, true, false>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                else
                    hipLaunchKernelGGL((helm_lane_kernel<NB, NQS, NQM, false, false>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gu);
                return;
            }
        if (p->pe == 64)
            launch_patch_pe<NB, NQS, NQM, 64>(p, A, st, native);
        else
            launch_patch_pe<NB, NQS, NQM, 32>(p, A, st, native);
    }

    // fused complex apply on the fp64 matrix cores (16-element batches)
    bool helm_mfma(int nb, int nqS, int nqM) { return nb >= 6 && nb <= 8 && nqS == nb + 1 && nqM == 2 + 3 * nb / 2; }

    bool supported(int nb, int nqS, int nqM)
    {
        return (nb == 3 && nqS == 4 && nqM == 6) || (nb == 4 && nqS == 5 && nqM == 8) || (nb == 5 && nqS == 6 && nqM == 9) ||
               (nb == 2 && nqS == 3 && nqM == 5) || helm_mfma(nb, nqS, nqM);
    }

    HelmArgs plan_args(const cuddh_helmholtz_plan *p, const double *x, double *y)
    {
        HelmArgs A;
        A.ndof = p->ndof;
        A.max_loc = p->max_loc;
        A.ncol = p->ncol;
        A.nfcol = p->nfcol;
        A.nqF = p->nqF;
        A.n_slots = p->n_slots;
        A.n_patches = p->n_patches;
        A.xcd_chunk = (p->n_patches + 7) / 8;
        A.omega = 0.0;
        A.dof_off = p->dof_off;
        A.dof_list = p->dof_list;
        A.slot_of = p->slot_of;
        A.own_count = p->own_count;
        A.dof_stride = p->dof_stride;
        A.patch_nel = p->patch_nel;
        A.face_off = p->face_off;
        A.face_id = p->face_id;
        A.lidx = p->lidx;
        A.face_lidx = p->face_lidx;
        A.colour = p->colour;
        A.face_col = p->face_col;
        A.Gp = p->Gp;
        A.aMp = p->aMp;
        A.aF = p->aF;
        A.x = x;
        A.y = y;
        A.part = p->part;
        A.stamps = p->stamps;
        A.pair_mass = p->pair_mass;
        A.own_off = p->own_off;
        A.bpos = p->bpos;
        A.bslot = p->bslot;
        A.bstride = p->bstride;
        return A;
    }

    // ---------------------------------------------------------------- one real operator through the same plan
    // y = [y +] c * S x  (KIND 0)  or  y = [y +] c * M x  (KIND 1) on a real vector.  Same layouts and colour phases as the
    // complex kernel; a wavefront owns one 64-element patch (PEK = 64, the default) or two 32-element patches, one per
    // half-wave (PEK = 32, kept for comparison): every metric load instruction fetches 512 contiguous bytes, no lane idles.
    // UM: the metric array of this operator is the same in every element and comes from the uniform table MU
    // PEK = 32: two patches of 32 elements per wavefront; PEK = 64: one patch of 64 (a third fewer border dofs and slots)
    template <int NB, int NQ, int KIND, bool NT, bool UM, int PEK>
    __global__ void __launch_bounds__(64, (NB >= 5 ? 2 : ((NB == 4 && KIND == 0 && !UM) ? 3 : 4))) op_patch_kernel(HelmArgs A, int accumulate, const double *__restrict__ P,
                                                                             const double *__restrict__ D, const double *__restrict__ MU)
    {
        constexpr int NN = NB * NB, NP = (NN + 1) / 2;
        constexpr int NM = (KIND == 0 ? 3 : 1) * NQ; // metric values of one slice
        extern __shared__ double lds[];
        constexpr bool ONE = PEK == 64;
        const int pair = (blockIdx.x & 7) * A.xcd_chunk + (blockIdx.x >> 3);
        const int first = ONE ? pair : 2 * pair; // first (or only) patch of this wavefront
        if (first >= A.n_patches)
            return; // whole workgroup
        const int lane = threadIdx.x;
        const int half = ONE ? 0 : lane >> 5, le = ONE ? lane : lane & 31;
        const int ML = A.max_loc;
        double *xs = lds;                      // [patches of this wavefront][ML]
        double *ys = lds + (ONE ? 1 : 2) * ML; // the same
        const int n_here = ONE ? 1 : min(2, A.n_patches - first);

        // the dof lists of the two patches are adjacent: one combined list, split at n0
        // (one patch per wavefront: fixed-stride lists, the first index requests do not wait for an offset load)
        const int n0 = A.dof_off[first + 1] - A.dof_off[first];
        const int ntot = A.dof_off[first + n_here] - A.dof_off[first];
        const int off = (ONE && A.dof_stride) ? first * A.dof_stride : A.dof_off[first];
        const int cap = (ONE && A.dof_stride) ? A.dof_stride - 1 : ntot - 1;
        const int *dofs = A.dof_list + off;

        // requests that do not depend on the LDS copy of x go out first (see helm_patch_kernel)
        const bool have = half < n_here;
        const int patch = have ? first + half : first; // a missing second patch re-reads the first (results dropped)
        const bool active = have && le < A.patch_nel[patch];
        const uint32_t *li = A.lidx + ((size_t)patch * NP) * PEK + le;
        uint32_t lpk[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j)
            lpk[j] = li[j * PEK];
        const int mycol = active ? A.colour[patch * PEK + le] : -1;
        const double *Mp = (KIND == 0 ? A.Gp + (size_t)patch * 3 * NQ * NQ * PEK : A.aMp + (size_t)patch * NQ * NQ * PEK) + le;
        auto load_slice = [&](int q, double (&g)[NM])
        {
#pragma unroll
            for (int t = 0; t < NM; ++t) // KIND 0: t = c * NQ + r (component c of point (q, r)); KIND 1: t = r
                g[t] = UM ? MU[q * NM + t] : metric_load<NT>(&Mp[(q * NM + t) * PEK]);
        };
        double g_first[NM];
        if constexpr (!UM)
            load_slice(0, g_first);

        constexpr int ROWS = 12; // 768 dofs per pass: both patches of n_basis 4 (2 x 325) in one
        constexpr bool EARLY = false; // measured: holding them across the element phase costs this kernel more than the round trip
        const int *slot = A.slot_of + off;
        int dest0[ROWS];
        if constexpr (EARLY)
        {
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest0[j] = slot[min(64 * j + lane, ntot - 1)];
        }
        for (int base = 0; base == 0 || base < ntot; base += 64 * ROWS) // (the first pass does not wait for ntot)
        {
            int gi[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                gi[j] = dofs[min(base + 64 * j + lane, cap)];
            double xv[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                xv[j] = A.x[gi[j]];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int t = base + 64 * j + lane;
                if (t < ntot)
                {
                    const int pos = t < n0 ? t : ML + (t - n0);
                    xs[pos] = xv[j];
                    ys[pos] = 0.0;
                }
            }
        }
        __syncthreads();

        // ------------------------------------------------------------ element phase
        const double *xc = xs + half * ML;
        auto lix_of = [&](int n) -> int { return (n & 1) ? static_cast<int>(lpk[n >> 1] >> 16) : static_cast<int>(lpk[n >> 1] & 0xFFFFu); };
        const double keep = active ? 1.0 : 0.0;
        double u[NN], out[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n)
        {
            u[n] = keep * xc[lix_of(n)];
            out[n] = 0.0;
        }

        auto slice = [&](int q, const double (&g)[NM])
        {
            if constexpr (KIND == 0)
            {
                double pu[NB], du[NB], t0[NB], t1[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                    {
                        a += P[q + NQ * k] * u[k + NB * l];
                        b += D[q + NQ * k] * u[k + NB * l];
                    }
                    pu[l] = a;
                    du[l] = b;
                    t0[l] = 0.0;
                    t1[l] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < NQ; ++r)
                {
                    double dx = 0.0, dy = 0.0;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                    {
                        dx += P[r + NQ * l] * du[l];
                        dy += D[r + NQ * l] * pu[l];
                    }
                    const double f0 = g[0 * NQ + r] * dx + g[1 * NQ + r] * dy;
                    const double f1 = g[1 * NQ + r] * dx + g[2 * NQ + r] * dy;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                    {
                        t0[l] += P[r + NQ * l] * f0;
                        t1[l] += D[r + NQ * l] * f1;
                    }
                }
#pragma unroll
                for (int l = 0; l < NB; ++l)
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        out[k + NB * l] = fma(D[q + NQ * k], t0[l], fma(P[q + NQ * k], t1[l], out[k + NB * l]));
            }
            else
            {
                double pu[NB], t[NB];
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        a += P[q + NQ * k] * u[k + NB * l];
                    pu[l] = a;
                    t[l] = 0.0;
                }
#pragma unroll
                for (int r = 0; r < NQ; ++r)
                {
                    double val = 0.0;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        val += P[r + NQ * l] * pu[l];
                    val *= g[r];
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        t[l] += P[r + NQ * l] * val;
                }
#pragma unroll
                for (int l = 0; l < NB; ++l)
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        out[k + NB * l] += P[q + NQ * k] * t[l];
            }
        };
        if constexpr (!UM)
            slice(0, g_first);
        // (two slices per round trip were measured slower: more registers, same memory queues)
#pragma unroll 1
        for (int q = UM ? 0 : 1; q < NQ; ++q)
        {
            double g[NM];
            load_slice(q, g);
            slice(q, g);
        }

        if constexpr (!EARLY)
        {
            // rows of 64 local dofs that hold owned dofs only (the plan numbers those first): the destination is the gather
            // index, re-read from the lines the gather brought on chip instead of the destination list
            const int own_rows = ONE ? (A.own_count[first] >> 6) : 0;
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest0[j] = (j < own_rows ? dofs : slot)[min(64 * j + lane, ntot - 1)];
        }

        // accumulate in colour phases (both patches at once: they use different halves of ys)
        {
            double *yc = ys + half * ML;
            for (int c = 0; c < A.ncol; ++c)
            {
                if (mycol == c)
                {
                    double acc[NN]; // loads first, then adds and stores (see helm_lane_kernel)
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        acc[n] = yc[lix_of(n)];
                    asm volatile("" ::: "memory"); // all loads before all stores, also in the generated code (the scheduler
                    __builtin_amdgcn_sched_barrier(0); // otherwise re-serialises them to save registers)
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        yc[lix_of(n)] = acc[n] + out[n];
                }
                __syncthreads();
            }
        }

        // ------------------------------------------------------------ write out
        const double c = A.omega; // the scale factor travels in the omega field
        for (int base = 0; base < ntot; base += 64 * ROWS)
        {
            int dest[ROWS]; // one index per dof: global dof (owned) or -(slot) - 1 (border)
            double y0[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest[j] = base == 0 ? dest0[j] : slot[min(base + 64 * j + lane, ntot - 1)];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                y0[j] = (accumulate && dest[j] >= 0) ? A.y[dest[j]] : 0.0;
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int t = base + 64 * j + lane;
                if (t >= ntot)
                    continue;
                const double val = c * ys[t < n0 ? t : ML + (t - n0)];
                if (dest[j] >= 0)
                    A.y[dest[j]] = y0[j] + val;
                else
                    A.part[-dest[j] - 1] = val;
            }
        }
    }

    // ---------------------------------------------------------------- high-order stiffness on the fp64 matrix cores
    // y = [y +] c * S x for 6 <= n_basis <= 8 (one element per lane needs more registers than a lane has there).  A
    // wavefront takes a batch of 16 elements: lane = element + 16 g; lane group g holds the element's values u(k, l) for the
    // xi-indices k = g and g + 4 and all eta-indices l.  Per eta-quadrature index r: the in-lane eta contraction gives the B
    // operand of v_mfma_f64_16x16x4_f64 directly (B[k = lane / 16][n = lane % 16], k-step s <-> xi-index 4 s + g), A = the 1-D
    // matrix D or P padded to 16 rows, and the product comes back in the C layout of the fp64 instruction -- register j of
    // lane group g is row 4 j + g -- i.e. every lane holds the xi-quadrature rows q = g, g + 4, g + 8 of its element for the
    // pointwise metric product.  That is again a B operand (k-step j <-> q = 4 j + g) for the contraction back over q, whose
    // result lands in the layout u started in (k = 4 j + g): no shuffle anywhere.  10 MFMAs + ~110 fp64 VALU per slice; the
    // metric array is streamed as [batch][r][3][q][16].  Colour phases, border slots, write-out: as op_patch_kernel.
    typedef double mfma_d4 __attribute__((ext_vector_type(4)));

    // KIND 1: the mass operator in the same scheme -- per slice one forward product v = P pl, the weight a(q, r), one backward
    // product; metric [batch][r][q][16]
    template <int NB, int NQ, int KIND>
    __global__ void __launch_bounds__(64, 2) op_mfma_kernel(HelmArgs A, int accumulate, const double *__restrict__ P, const double *__restrict__ D,
                                                            const double *__restrict__ Gm, long long gm_stride)
    {
        static_assert(NB >= 5 && NB <= 8 && NQ <= 16, "xi-indices k = g + 4 s with s < 2");
        constexpr int NN = NB * NB, NP = (NN + 1) / 2, PEM = 16;
        constexpr int JQ = (NQ + 3) / 4; // registers / k-steps that carry quadrature rows
        extern __shared__ double lds[];
        const int patch = (blockIdx.x & 7) * A.xcd_chunk + (blockIdx.x >> 3);
        if (patch >= A.n_patches)
            return; // whole workgroup
        const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
        const int ML = A.max_loc;
        double *xs = lds, *ys = lds + ML;
        const int off = A.dof_off[patch];
        const int nloc = A.dof_off[patch + 1] - off;
        const int *dofs = A.dof_list + off;
        const bool active = e < A.patch_nel[patch];
        const int mycol = active ? A.colour[patch * PEM + e] : -1;
        const uint32_t *li = A.lidx + (size_t)patch * NP * PEM + e;
        // patch-local dofs of my nodes (k = g + 4 s, l): the same for the input and for the result
        int id[2][NB];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int l = 0; l < NB; ++l)
            {
                const int n = g + 4 * s + NB * l;
                const uint32_t w = (g + 4 * s < NB) ? li[(n >> 1) * PEM] : 0u;
                id[s][l] = (n & 1) ? static_cast<int>(w >> 16) : static_cast<int>(w & 0xFFFFu);
            }

        constexpr int ROWS = 14; // 896 dofs per pass: a 4x4-element batch of n_basis 8 (841) in one
        for (int base = 0; base < nloc; base += 64 * ROWS)
        {
            int gi[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                gi[j] = dofs[min(base + 64 * j + lane, nloc - 1)];
            double xv[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                xv[j] = A.x[gi[j]];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + 64 * j + lane;
                if (i < nloc)
                {
                    xs[i] = xv[j];
                    ys[i] = 0.0;
                }
            }
        }
        __syncthreads();

        double U[2][NB], OUT[2][NB];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int l = 0; l < NB; ++l)
            {
                U[s][l] = (active && g + 4 * s < NB) ? xs[id[s][l]] : 0.0;
                OUT[s][l] = 0.0;
            }

        // A operands: row i = lane % 16, column = 4 * step + lane / 16.  forward (rows q, columns k'): D(q, k'), P(q, k');
        // backward (rows k, columns q): D(q, k), P(q, k)
        // backward products run on v_mfma_f64_4x4x4_f64 (four independent 4 x 4 x 4 blocks, one per group of four elements):
        // same B / C lane layout as the 16 x 16 x 4 form (k-index and output row = lane / 16, column = lane % 16) but only the
        // KB = ceil(NB / 4) row blocks that exist are computed -- no padding to 16 rows.  A[rb][j]: row 4 rb + (lane & 3),
        // column 4 j + (lane >> 4).
        constexpr int KB = (NB + 3) / 4;
        double AfD[2], AfP[2], AbD[KB][JQ], AbP[KB][JQ];
#pragma unroll
        for (int s = 0; s < 2; ++s)
        {
            const int q = e, kp = 4 * s + g;
            const bool ok = q < NQ && kp < NB;
            AfD[s] = (KIND == 0 && ok) ? D[q + NQ * kp] : 0.0;
            AfP[s] = ok ? P[q + NQ * kp] : 0.0;
        }
#pragma unroll
        for (int rb = 0; rb < KB; ++rb)
#pragma unroll
            for (int sp = 0; sp < JQ; ++sp)
            {
                const int k = 4 * rb + (lane & 3), q = 4 * sp + g;
                const bool ok = k < NB && q < NQ;
                AbD[rb][sp] = (KIND == 0 && ok) ? D[q + NQ * k] : 0.0;
                AbP[rb][sp] = ok ? P[q + NQ * k] : 0.0;
            }

        if constexpr (KIND == 0)
        {
            const double *Gb = Gm + (size_t)patch * gm_stride + e; // gm_stride == 0: affine mesh, one block for every batch
#pragma unroll 1
            for (int r = 0; r < NQ; ++r)
            {
                // metric at my quadrature rows q = 4 j + g
                double ga[JQ], gb[JQ], gc[JQ];
#pragma unroll
                for (int j = 0; j < JQ; ++j)
                {
                    const int q = 4 * j + g;
                    const bool ok = q < NQ;
                    const size_t o = (((size_t)r * 3) * NQ + (ok ? q : 0)) * PEM;
                    ga[j] = ok ? __builtin_nontemporal_load(&Gb[o]) : 0.0;
                    gb[j] = ok ? __builtin_nontemporal_load(&Gb[o + (size_t)NQ * PEM]) : 0.0;
                    gc[j] = ok ? __builtin_nontemporal_load(&Gb[o + (size_t)2 * NQ * PEM]) : 0.0;
                }
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
                // dx(q, r) = sum_k' D(q, k') pl_k'(r),  dy(q, r) = sum_k' P(q, k') dl_k'(r)
                mfma_d4 dx = {0, 0, 0, 0}, dy = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 2; ++s)
                {
                    dx = __builtin_amdgcn_mfma_f64_16x16x4f64(AfD[s], pl[s], dx, 0, 0, 0);
                    dy = __builtin_amdgcn_mfma_f64_16x16x4f64(AfP[s], dl[s], dy, 0, 0, 0);
                }
                // W0(k) = sum_q D(q, k) F0(q, r),  W1(k) = sum_q P(q, k) F1(q, r)
                double W0[KB], W1[KB];
#pragma unroll
                for (int rb = 0; rb < KB; ++rb)
                    W0[rb] = W1[rb] = 0.0;
#pragma unroll
                for (int j = 0; j < JQ; ++j)
                {
                    const double f0 = ga[j] * dx[j] + gb[j] * dy[j];
                    const double f1 = gb[j] * dx[j] + gc[j] * dy[j];
#pragma unroll
                    for (int rb = 0; rb < KB; ++rb)
                    {
                        W0[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbD[rb][j], f0, W0[rb], 0, 0, 0);
                        W1[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbP[rb][j], f1, W1[rb], 0, 0, 0);
                    }
                }
                // out(k, l) += P(r, l) W0(k) + D(r, l) W1(k),  k = 4 s + g
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        OUT[s][l] = fma(P[r + NQ * l], W0[s], fma(D[r + NQ * l], W1[s], OUT[s][l]));
            }
        }
        else
        {
            const double *ab = Gm + (size_t)patch * gm_stride + e;
#pragma unroll 1
            for (int r = 0; r < NQ; ++r)
            {
                double am[JQ];
#pragma unroll
                for (int j = 0; j < JQ; ++j)
                {
                    const int q = 4 * j + g;
                    am[j] = q < NQ ? __builtin_nontemporal_load(&ab[((size_t)r * NQ + q) * PEM]) : 0.0;
                }
                double pl[2];
#pragma unroll
                for (int s = 0; s < 2; ++s)
                {
                    double a = 0.0;
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        a += P[r + NQ * l] * U[s][l];
                    pl[s] = a;
                }
                // v(q, r) = sum_k' P(q, k') pl_k'(r);  W(k) = sum_q P(q, k) a(q, r) v(q, r)
                mfma_d4 v = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    v = __builtin_amdgcn_mfma_f64_16x16x4f64(AfP[s], pl[s], v, 0, 0, 0);
                double W[KB];
#pragma unroll
                for (int rb = 0; rb < KB; ++rb)
                    W[rb] = 0.0;
#pragma unroll
                for (int j = 0; j < JQ; ++j)
#pragma unroll
                    for (int rb = 0; rb < KB; ++rb)
                        W[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbP[rb][j], am[j] * v[j], W[rb], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int l = 0; l < NB; ++l)
                        OUT[s][l] += P[r + NQ * l] * W[s];
            }
        }

        // accumulate in colour phases: the four lane groups of an element hold different nodes of it
        for (int c = 0; c < A.ncol; ++c)
        {
            if (mycol == c)
            {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (g + 4 * s < NB)
                    {
                        double acc[NB]; // loads first, then adds and stores (see helm_lane_kernel)
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            acc[l] = ys[id[s][l]];
                        asm volatile("" ::: "memory"); // all loads before all stores, also in the generated code (the scheduler
                        __builtin_amdgcn_sched_barrier(0); // otherwise re-serialises them to save registers)
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            ys[id[s][l]] = acc[l] + OUT[s][l];
                    }
            }
            __syncthreads();
        }

        // write out (as op_patch_kernel)
        const double cs = A.omega; // the scale factor travels in the omega field
        const int *slot = A.slot_of + off;
        for (int base = 0; base < nloc; base += 64 * ROWS)
        {
            int dest[ROWS];
            double y0[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                dest[j] = slot[min(base + 64 * j + lane, nloc - 1)];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                y0[j] = (accumulate && dest[j] >= 0) ? A.y[dest[j]] : 0.0;
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + 64 * j + lane;
                if (i >= nloc)
                    continue;
                const double val = cs * ys[i];
                if (dest[j] >= 0)
                    A.y[dest[j]] = y0[j] + val;
                else
                    A.part[-dest[j] - 1] = val;
            }
        }
    }

    // ---------------------------------------------------------------- fused complex Helmholtz apply on the fp64 matrix cores
    // [u; v] -> [S u - w^2 M u - w H v ; -(S v - w^2 M v + w H u)] for n_basis 6-8 in the scheme of op_mfma_kernel: a batch
    // of 16 elements per WORKGROUP of two wavefronts, wave 0 applying the operators to u and wave 1 to v (stiffness slices,
    // then mass slices).  Each wave needs the registers of a single-operator kernel only (3-4 waves per SIMD; with both
    // components in one wave it was 2) and a batch is finished in about half the time: 263 -> 184 us (n_basis 7) and
    // 364 -> 256 us (n_basis 8) at 147k elements.  Both waves read the same metric slices, so those use the default cache
    // policy (the second read hits).  The LDS copy of x is consumed when the registers are filled, so the same LDS serves as
    // the accumulator y (2 x max_loc doubles per batch); the boundary-face term re-reads its few x values from global memory.
    // NATIVE: x and y in the plan's own vector ordering (pairs (u, v), a batch's owned dofs contiguous; see helm_lane_kernel): the
    // 128 threads gather pairs with 16-byte loads at addresses known at kernel entry and write pairs back the same way -- no
    // dof list and no destination list for owned dofs (irregular 121,856-quad mesh: n_basis 6 145.6 -> 126.6 us, 7 163.1 -> 149.5 us).
    // Measured and removed (profiles/r03/config5_ab.txt): the batch's mass weights copied into LDS through the DMA path
    // (global_load_lds_dwordx4, no destination registers, both wavefronts sharing the copy; NQM dependent round trips fewer) --
    // 10-18 % SLOWER at n_basis 6-8 (native ordering, same mesh: n_basis 6 126.6 -> 139.8 us, 7 149.5 -> 168.9 us): a
    // wavefront's LDS-DMA pieces are served one at a time, and the LDS they need costs a quarter of the resident workgroups.
    template <int NB, int NQS, int NQM, bool NATIVE = false, int NCS = 0, int NCM = 0, int OCC = 0>
    __global__ void __launch_bounds__(128, (OCC ? OCC : NB <= 6 ? 4 : 3)) helm_mfma_kernel(HelmArgs A, const double *__restrict__ PS, const double *__restrict__ DS,
                                                              const double *__restrict__ PM, const double *__restrict__ PF,
                                                              const double *__restrict__ Gm, long long gm_stride,
                                                              const double *__restrict__ Am, long long am_stride)
    {
        static_assert(NB >= 5 && NB <= 8 && NQS <= 16 && NQM <= 16, "xi-indices k = g + 4 s with s < 2");
        constexpr int NN = NB * NB, NP = (NN + 1) / 2, PEM = 16;
        constexpr int JS = (NQS + 3) / 4, JM = (NQM + 3) / 4;
        // stiffness / mass slices per dependent round trip.  Same-box A/B of four builds (profiles/r02/mfma_grouping_ab.txt): at
        // n_basis 7 two / four per trip fit the 168 registers of 3 wavefronts per SIMD (161) and gain 6-8 % (irregular
        // 121,856-quad mesh 190 -> 179 us, 384^2 230 -> 214 us); n_basis 6 (128-register cap, 21 spilled; 3 wavefronts per SIMD
        // without spills: no better) and n_basis 8 (18 spilled) lose 5 %, four / six per trip at 2 wavefronts per SIMD loses 15 %
        // (round 3, native ordering, same-box A/B of build variants at n_basis 6, profiles/r03/config5_ab.txt: 3 wavefronts per SIMD
        // 126.5 -> 142 us, with two / four slices per trip 137 us, two mass slices per trip at 4 wavefronts 131 us, the element ->
        // local dof map re-read before the colour phases instead of held in registers 128 us -- none kept)
        constexpr int GS = (NB == 7 && NCS == 0) ? 2 : 1, GM = (NB == 7 && NCS == 0) ? 4 : 1; // (staged: slices come from LDS, one at a time)
        // forward products (rows = quadrature points) on v_mfma_f64_4x4x4 like the backward ones: JS row blocks x KB xi blocks
        // instead of 16 padded rows per k-step (n_basis 6: 7 of 16 rows were real).  Same-box A/B: n_basis 6 +2 %, 7 +2 %;
        // n_basis 8 (three row blocks, spills) -9 %, so it keeps the 16x16x4 form
        constexpr bool FWD4 = NB <= 7;
        extern __shared__ double lds[];
        const int patch = (blockIdx.x & 7) * A.xcd_chunk + (blockIdx.x >> 3);
        if (patch >= A.n_patches)
            return; // whole workgroup
        const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
        const int cmp = threadIdx.x >> 6; // component of this wavefront
        const int ML = A.max_loc;
        double *xy = lds; // [2][ML]: first the gathered x, then the accumulated y
        auto stamp = [&](int k) // diagnostic, as in helm_lane_kernel (wavefront 0 of the batch)
        {
            if (A.stamps)
            {
                unsigned long long t;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                if (threadIdx.x == 0)
                    A.stamps[(size_t)patch * 8 + k] = t;
            }
        };
        stamp(0);
        // STAGE (NCS > 0): the batch's metric data (3 NQS^2 + NQM^2 doubles per element, contiguous per batch) travels in NCS + NCM
        // chunks of whole slices through one LDS buffer behind xy: the 128 threads request chunk k + 1 (16 bytes per thread and piece,
        // held in registers) before the slices of chunk k are worked from LDS, and park it when those are done.  A slice then costs
        // no dependent trip to memory, the two wavefronts share one copy, and the requests are whole lines.
        constexpr bool STAGE = NCS > 0;
        constexpr int HS = STAGE ? (NQS + NCS - 1) / NCS : NQS, HM = STAGE ? (NQM + NCM - 1) / NCM : NQM; // slices per chunk
        constexpr int CS2 = HS * 3 * NQS * PEM / 2, CM2 = HM * NQM * PEM / 2;                                // dbl2 per (full) chunk
        constexpr int CH2 = CS2 > CM2 ? CS2 : CM2, RC = (CH2 + 127) / 128;
        double *mC = lds + 2 * ((ML + 1) & ~1);
        dbl2_t nx[STAGE ? RC : 1];
        const dbl2_t *gsrc = reinterpret_cast<const dbl2_t *>(Gm + (size_t)patch * gm_stride);
        const dbl2_t *asrc = reinterpret_cast<const dbl2_t *>(Am + (size_t)patch * am_stride);
        // chunk c of the sequence [stiffness 0 .. NCS-1, mass 0 .. NCM-1]: request (clamped addresses, no branches) / park
        auto chunk_request = [&](int c)
        {
            const bool st = c < NCS;
            const int h = st ? c : c - NCS;
            const int first = st ? h * HS : h * HM, cnt = st ? min(HS, NQS - first) : min(HM, NQM - first);
            const dbl2_t *src = st ? gsrc + first * (3 * NQS * PEM / 2) : asrc + first * (NQM * PEM / 2);
            const int n2 = cnt * (st ? 3 * NQS * PEM / 2 : NQM * PEM / 2);
#pragma unroll
            for (int j = 0; j < RC; ++j)
                nx[j] = __builtin_nontemporal_load(src + min(128 * j + (int)threadIdx.x, n2 - 1));
            return n2;
        };
        auto chunk_park = [&](int n2)
        {
#pragma unroll
            for (int j = 0; j < RC; ++j)
                if (128 * j + (int)threadIdx.x < n2)
                    reinterpret_cast<dbl2_t *>(mC)[128 * j + threadIdx.x] = nx[j];
        };
        // FASTG (staged native form): everything whose address is known at kernel entry is requested in one burst and without
        // branches (a load under a condition becomes its own basic block and drains the queue first).  The batch's few scalars come
        // through the scalar cache; then, oldest first because the counter is in order: border positions, the owned rows of x,
        // chunk 0, colours and the element -> local dof map.  What remains dependent is border position -> x for the border rows
        // (phase stamps before: four trips in a row, 4.7 of a batch's 20 us)
        constexpr bool FASTG = NATIVE && STAGE;
        constexpr int GR = NB == 5 ? 3 : NB == 6 ? 4 : NB == 7 ? 5 : 7; // rows of 128 that hold a 4x4-element batch
        int off, nloc, nel, own0 = 0, nown = 0;
        const int *bp = nullptr, *bs = nullptr;
        const int bcap = A.bstride - 1;
        const dbl2_t *X2 = reinterpret_cast<const dbl2_t *>(A.x);
        int pb[FASTG ? GR : 1];
        dbl2_t xo[FASTG ? GR : 1];
        if constexpr (FASTG)
        {
            unsigned long long d2, o2;
            asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(d2), "=&s"(o2), "=&s"(nel)
                         : "s"(A.dof_off + patch), "s"(A.own_off + patch), "s"(A.patch_nel + patch));
            off = static_cast<int>(d2 & 0xFFFFFFFFull);
            nloc = static_cast<int>(d2 >> 32) - off;
            own0 = static_cast<int>(o2 & 0xFFFFFFFFull);
            nown = static_cast<int>(o2 >> 32) - own0;
            bp = A.bpos + (size_t)patch * A.bstride;
            bs = A.bslot + (size_t)patch * A.bstride;
            const int last_own = own0 + max(nown - 1, 0);
#pragma unroll
            for (int j = 0; j < GR; ++j)
                pb[j] = bp[max(0, min(128 * j + (int)threadIdx.x - nown, bcap))];
#pragma unroll
            for (int j = 0; j < GR; ++j)
                xo[j] = X2[min(own0 + 128 * j + (int)threadIdx.x, last_own)]; // owned rows: no dependence on the list
        }
        int n2_next = 0;
        if constexpr (STAGE)
            n2_next = chunk_request(0);
        const uint32_t *li = A.lidx + (size_t)patch * NP * PEM + e;
        // the batch's element -> local dof map (NP x 16 words) and colours are parked in LDS too (behind the chunk buffer): twelve
        // registers fewer through the slices, and whole-line requests instead of twelve scattered ones per lane
        constexpr int LW = NP * PEM, LWR = (LW + 127) / 128;
        uint32_t *lw = reinterpret_cast<uint32_t *>(mC + 2 * CH2);
        int *colL = reinterpret_cast<int *>(lw + LW);
        uint32_t lwr[FASTG ? LWR : 1];
        int col_raw = -1;
        if constexpr (FASTG)
        {
            col_raw = A.colour[patch * PEM + e];
#pragma unroll
            for (int j = 0; j < LWR; ++j)
                lwr[j] = A.lidx[(size_t)patch * LW + min(128 * j + (int)threadIdx.x, LW - 1)];
        }
        else
        {
            off = A.dof_off[patch];
            nloc = A.dof_off[patch + 1] - off;
            nel = A.patch_nel[patch];
        }
        const int *dofs = A.dof_list + off;
        const bool active = e < nel;
        int mycol = -1;
        if constexpr (!FASTG)
            mycol = active ? A.colour[patch * PEM + e] : -1;
        int id[2][NB];
        if constexpr (!FASTG)
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    const int n = g + 4 * s + NB * l;
                    const uint32_t w = (g + 4 * s < NB) ? li[(n >> 1) * PEM] : 0u;
                    id[s][l] = (n & 1) ? static_cast<int>(w >> 16) : static_cast<int>(w & 0xFFFFu);
                }
        }
        auto local_id = [&](int s, int l) -> int // node g + 4 s + NB l of this lane's element
        {
            if constexpr (FASTG)
            {
                const int n = g + 4 * s + NB * l; // (g + 4 s >= NB: some other node's id, a valid LDS index; the caller discards the value)
                const uint32_t w = lw[min(n >> 1, NP - 1) * PEM + e];
                return (n & 1) ? static_cast<int>(w >> 16) : static_cast<int>(w & 0xFFFFu);
            }
            else
                return id[s][l];
        };

        constexpr int ROWS = 7; // 128 threads x 7 = 896 dofs per pass: a 4x4-element batch of n_basis 8 (841) in one
        if constexpr (FASTG)
        {
            {
                // straight-line (a loop's header would wait for the burst above before the first request goes out); the values are
                // pinned before the guarded LDS writes, or the compiler sinks a row's loads into its guard, one drained trip each
                dbl2_t xb[GR];
#pragma unroll
                for (int j = 0; j < GR; ++j)
                    xb[j] = X2[pb[j]];
#pragma unroll
                for (int j = 0; j < GR; ++j)
                    asm volatile("" : "+v"(xo[j].x), "+v"(xo[j].y), "+v"(xb[j].x), "+v"(xb[j].y));
#pragma unroll
                for (int j = 0; j < GR; ++j)
                {
                    const int i = 128 * j + (int)threadIdx.x;
                    if (i < nloc)
                    {
                        xy[i] = i < nown ? xo[j].x : xb[j].x;
                        xy[ML + i] = i < nown ? xo[j].y : xb[j].y;
                    }
                }
            }
            for (int i = 128 * GR + (int)threadIdx.x; i < nloc; i += 128) // a batch that is not a 4x4 block (irregular meshes)
            {
                const dbl2_t t = X2[i < nown ? own0 + i : bp[min(i - nown, bcap)]];
                xy[i] = t.x;
                xy[ML + i] = t.y;
            }
        }
        else if constexpr (NATIVE)
        {
            own0 = A.own_off[patch];
            nown = A.own_off[patch + 1] - own0;
            bp = A.bpos + (size_t)patch * A.bstride;
            bs = A.bslot + (size_t)patch * A.bstride;
            for (int base = 0; base < nloc; base += 128 * ROWS)
            {
                int pos[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + 128 * j + (int)threadIdx.x;
                    pos[j] = own0 + i;
                    if (base + 128 * j + 127 >= nown) // (workgroup-uniform) a row that holds border dofs
                        pos[j] = bp[max(0, min(i - nown, bcap))];
                }
                dbl2_t xv2[ROWS];
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + 128 * j + (int)threadIdx.x;
                    xv2[j] = X2[i < nown ? own0 + i : pos[j]];
                }
#pragma unroll
                for (int j = 0; j < ROWS; ++j)
                {
                    const int i = base + 128 * j + (int)threadIdx.x;
                    if (i < nloc)
                    {
                        xy[i] = xv2[j].x;
                        xy[ML + i] = xv2[j].y;
                    }
                }
            }
        }
        else
        for (int base = 0; base < nloc; base += 128 * ROWS)
        {
            int gi[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                gi[j] = dofs[min(base + 128 * j + (int)threadIdx.x, nloc - 1)];
            double xu[ROWS], xv[ROWS];
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                xu[j] = A.x[gi[j]];
                xv[j] = A.x[A.ndof + gi[j]];
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
            {
                const int i = base + 128 * j + (int)threadIdx.x;
                if (i < nloc)
                {
                    xy[i] = xu[j];
                    xy[ML + i] = xv[j];
                }
            }
        }
        if constexpr (STAGE)
            chunk_park(n2_next);
        if constexpr (FASTG)
        {
#pragma unroll
            for (int j = 0; j < LWR; ++j)
                if (128 * j + (int)threadIdx.x < LW)
                    lw[128 * j + threadIdx.x] = lwr[j];
            if (threadIdx.x < PEM)
                colL[threadIdx.x] = col_raw;
        }
        __syncthreads();
        stamp(1); // x (and the staged metric data) in LDS
        double U[1][2][NB], OUT[1][2][NB]; // [this wave's component][s][l]
        if constexpr (FASTG)
        {
            // without branches: all ids, then all values (lanes without an element or a node read valid LDS and discard)
            int il[2][NB];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    il[s][l] = local_id(s, l);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                {
                    const double t = xy[cmp * ML + il[s][l]];
                    U[0][s][l] = (active && g + 4 * s < NB) ? t : 0.0;
                    OUT[0][s][l] = 0.0;
                }
        }
        else
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int l = 0; l < NB; ++l)
            {
                U[0][s][l] = (active && g + 4 * s < NB) ? xy[cmp * ML + local_id(s, l)] : 0.0;
                OUT[0][s][l] = 0.0;
            }
        // xy becomes the accumulator y: cleared once every thread has its values.  Staged form: between the two barriers at which the
        // first chunk is parked (nobody touches xy during the slices), so the clear costs no barrier of its own.
        if constexpr (!STAGE)
        {
            __syncthreads();
            for (int i = threadIdx.x; i < nloc; i += 128)
            {
                xy[i] = 0.0;
                xy[ML + i] = 0.0;
            }
            __syncthreads();
        }

        stamp(2); // element values in registers (unstaged form: and LDS cleared)
        // ------------------------------------------------------------ stiffness slices
        {
            constexpr int KB = (NB + 3) / 4; // backward products on v_mfma_f64_4x4x4_f64, as in op_mfma_kernel
            double AfD[FWD4 ? 1 : 2], AfP[FWD4 ? 1 : 2], AbD[KB][JS], AbP[KB][JS];
            double AfD4[FWD4 ? JS : 1][KB], AfP4[FWD4 ? JS : 1][KB]; // A[i][k] <-> lane i + 4 b + 16 k: row q = 4 j + (lane & 3), column k = 4 s + g
            if constexpr (FWD4)
            {
#pragma unroll
                for (int j = 0; j < JS; ++j)
#pragma unroll
                    for (int s = 0; s < KB; ++s)
                    {
                        const int q = 4 * j + (lane & 3), kp = 4 * s + g;
                        const bool ok = q < NQS && kp < NB;
                        AfD4[j][s] = ok ? DS[q + NQS * kp] : 0.0;
                        AfP4[j][s] = ok ? PS[q + NQS * kp] : 0.0;
                    }
            }
            else
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
                const int q = e, kp = 4 * s + g;
                const bool ok = q < NQS && kp < NB;
                AfD[s] = ok ? DS[q + NQS * kp] : 0.0;
                AfP[s] = ok ? PS[q + NQS * kp] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < KB; ++rb)
#pragma unroll
                for (int sp = 0; sp < JS; ++sp)
                {
                    const int k = 4 * rb + (lane & 3), q = 4 * sp + g;
                    const bool ok = k < NB && q < NQS;
                    AbD[rb][sp] = ok ? DS[q + NQS * k] : 0.0;
                    AbP[rb][sp] = ok ? PS[q + NQS * k] : 0.0;
                }
            const double *Gb = Gm + (size_t)patch * gm_stride + e;
            // GS slices travel together (a slice is 3 JS doubles per lane, 3 KB per wavefront); the fence keeps the group's loads
            // ahead of its arithmetic in the generated code
            if constexpr (STAGE) // the coefficient tables above have landed: the loop's waits must not count the chunk requests it issues
                __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
#pragma unroll 1
            for (int r0 = 0; r0 < NQS; r0 += GS)
            {
                if constexpr (STAGE)
                    if (r0 % HS == 0) // a chunk begins: its successor is requested now and parked when this chunk's slices are done
                        n2_next = chunk_request(r0 / HS + 1);
                double gaG[GS][JS], gbG[GS][JS], gcG[GS][JS];
#pragma unroll
                for (int gi = 0; gi < GS; ++gi)
#pragma unroll
                    for (int j = 0; j < JS; ++j)
                    {
                        const int q = 4 * j + g, rr = min(r0 + gi, NQS - 1);
                        const bool ok = q < NQS;
                        const size_t o = (((size_t)rr * 3) * NQS + (ok ? q : 0)) * PEM;
                        if constexpr (STAGE)
                        {
                            const int ol = (((rr % HS) * 3) * NQS + (ok ? q : 0)) * PEM + e;
                            gaG[gi][j] = ok ? mC[ol] : 0.0;
                            gbG[gi][j] = ok ? mC[ol + NQS * PEM] : 0.0;
                            gcG[gi][j] = ok ? mC[ol + 2 * NQS * PEM] : 0.0;
                        }
                        else
                        {
                            gaG[gi][j] = ok ? Gb[o] : 0.0;
                            gbG[gi][j] = ok ? Gb[o + (size_t)NQS * PEM] : 0.0;
                            gcG[gi][j] = ok ? Gb[o + (size_t)2 * NQS * PEM] : 0.0;
                        }
                    }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int gi = 0; gi < GS; ++gi)
                {
                    const int r = r0 + gi;
                    if (r >= NQS)
                        break;
                    const double(&ga)[JS] = gaG[gi], (&gb)[JS] = gbG[gi], (&gc)[JS] = gcG[gi];
                    constexpr int c = 0;
                    {
                    double pl[2], dl[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                    {
                        double a = 0.0, b = 0.0;
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                        {
                            a += PS[r + NQS * l] * U[c][s][l];
                            b += DS[r + NQS * l] * U[c][s][l];
                        }
                        pl[s] = a;
                        dl[s] = b;
                    }
                    double dx[JS], dy[JS];
                    if constexpr (FWD4)
                    {
                        // forward products on the 4x4x4 form too: JS row blocks x KB xi blocks instead of 16 padded rows per k-step
#pragma unroll
                        for (int j = 0; j < JS; ++j)
                        {
                            dx[j] = dy[j] = 0.0;
#pragma unroll
                            for (int s = 0; s < KB; ++s)
                            {
                                dx[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(AfD4[j][s], pl[s], dx[j], 0, 0, 0);
                                dy[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(AfP4[j][s], dl[s], dy[j], 0, 0, 0);
                            }
                        }
                    }
                    else
                    {
                        mfma_d4 dx4 = {0, 0, 0, 0}, dy4 = {0, 0, 0, 0};
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                        {
                            dx4 = __builtin_amdgcn_mfma_f64_16x16x4f64(AfD[s], pl[s], dx4, 0, 0, 0);
                            dy4 = __builtin_amdgcn_mfma_f64_16x16x4f64(AfP[s], dl[s], dy4, 0, 0, 0);
                        }
#pragma unroll
                        for (int j = 0; j < JS; ++j)
                        {
                            dx[j] = dx4[j];
                            dy[j] = dy4[j];
                        }
                    }
                    double W0[KB], W1[KB];
#pragma unroll
                    for (int rb = 0; rb < KB; ++rb)
                        W0[rb] = W1[rb] = 0.0;
#pragma unroll
                    for (int j = 0; j < JS; ++j)
                    {
                        const double f0 = ga[j] * dx[j] + gb[j] * dy[j];
                        const double f1 = gb[j] * dx[j] + gc[j] * dy[j];
#pragma unroll
                        for (int rb = 0; rb < KB; ++rb)
                        {
                            W0[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbD[rb][j], f0, W0[rb], 0, 0, 0);
                            W1[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbP[rb][j], f1, W1[rb], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            OUT[c][s][l] = fma(PS[r + NQS * l], W0[s], fma(DS[r + NQS * l], W1[s], OUT[c][s][l])); // two FMAs (the sum form compiles to mul + fma + add)
                    }
                }
                if constexpr (STAGE)
                    if ((r0 + 1) % HS == 0 || r0 + 1 == NQS) // the chunk's last slice: both wavefronts are done with the buffer
                    {
                        __syncthreads();
                        chunk_park(n2_next);
                        if (r0 + 1 == (HS < NQS ? HS : NQS)) // first park: everybody has read its values from xy (see above)
                            for (int i = threadIdx.x; i < nloc; i += 128)
                            {
                                xy[i] = 0.0;
                                xy[ML + i] = 0.0;
                            }
                        __syncthreads();
                    }
            }
        }

        stamp(3); // stiffness slices done
        // ------------------------------------------------------------ mass slices: out -= w^2 M u
        {
            const double w2 = -A.omega * A.omega;
            constexpr int KB = (NB + 3) / 4;
            double AfP[FWD4 ? 1 : 2], AbP[KB][JM];
            double AfP4[FWD4 ? JM : 1][KB];
            if constexpr (FWD4)
            {
#pragma unroll
                for (int j = 0; j < JM; ++j)
#pragma unroll
                    for (int s = 0; s < KB; ++s)
                    {
                        const int q = 4 * j + (lane & 3), kp = 4 * s + g;
                        AfP4[j][s] = (q < NQM && kp < NB) ? PM[q + NQM * kp] : 0.0;
                    }
            }
            else
#pragma unroll
            for (int s = 0; s < 2; ++s)
            {
                const int q = e, kp = 4 * s + g;
                AfP[s] = (q < NQM && kp < NB) ? PM[q + NQM * kp] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < KB; ++rb)
#pragma unroll
                for (int sp = 0; sp < JM; ++sp)
                {
                    const int k = 4 * rb + (lane & 3), q = 4 * sp + g;
                    AbP[rb][sp] = (k < NB && q < NQM) ? PM[q + NQM * k] : 0.0;
                }
            const double *ab = Am + (size_t)patch * am_stride + e;
            if constexpr (STAGE)
                __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), as above
#pragma unroll 1
            for (int r0 = 0; r0 < NQM; r0 += GM)
            {
                if constexpr (STAGE)
                    if (r0 % HM == 0 && r0 / HM + 1 < NCM)
                        n2_next = chunk_request(NCS + r0 / HM + 1);
                double amG[GM][JM];
#pragma unroll
                for (int gi = 0; gi < GM; ++gi)
#pragma unroll
                    for (int j = 0; j < JM; ++j)
                    {
                        const int q = 4 * j + g, rr = min(r0 + gi, NQM - 1);
                        if constexpr (STAGE)
                            amG[gi][j] = q < NQM ? mC[((rr % HM) * NQM + q) * PEM + e] : 0.0;
                        else
                            amG[gi][j] = q < NQM ? ab[((size_t)rr * NQM + q) * PEM] : 0.0;
                    }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int gi = 0; gi < GM; ++gi)
                {
                    const int r = r0 + gi;
                    if (r >= NQM)
                        break;
                    double am[JM];
#pragma unroll
                    for (int j = 0; j < JM; ++j)
                        am[j] = w2 * amG[gi][j];
                    constexpr int c = 0;
                    {
                    double pl[2];
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                    {
                        double a = 0.0;
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            a += PM[r + NQM * l] * U[c][s][l];
                        pl[s] = a;
                    }
                    double v[JM];
                    if constexpr (FWD4)
                    {
#pragma unroll
                        for (int j = 0; j < JM; ++j)
                        {
                            v[j] = 0.0;
#pragma unroll
                            for (int s = 0; s < KB; ++s)
                                v[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(AfP4[j][s], pl[s], v[j], 0, 0, 0);
                        }
                    }
                    else
                    {
                        mfma_d4 v4 = {0, 0, 0, 0};
#pragma unroll
                        for (int s = 0; s < 2; ++s)
                            v4 = __builtin_amdgcn_mfma_f64_16x16x4f64(AfP[s], pl[s], v4, 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < JM; ++j)
                            v[j] = v4[j];
                    }
                    double W[KB];
#pragma unroll
                    for (int rb = 0; rb < KB; ++rb)
                        W[rb] = 0.0;
#pragma unroll
                    for (int j = 0; j < JM; ++j)
#pragma unroll
                        for (int rb = 0; rb < KB; ++rb)
                            W[rb] = __builtin_amdgcn_mfma_f64_4x4x4f64(AbP[rb][j], am[j] * v[j], W[rb], 0, 0, 0);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            OUT[c][s][l] += PM[r + NQM * l] * W[s];
                    }
                }
                if constexpr (STAGE)
                    if ((r0 + 1) % HM == 0 && r0 + 1 < NQM)
                    {
                        __syncthreads();
                        chunk_park(n2_next);
                        __syncthreads();
                    }
            }
        }

        stamp(4); // mass slices done
        // ------------------------------------------------------------ accumulate in colour phases; the v row is negated
        int ilc[2][NB];
        if constexpr (FASTG)
        {
            mycol = active ? colL[e] : -1;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    ilc[s][l] = local_id(s, l);
        }
        else
        {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int l = 0; l < NB; ++l)
                    ilc[s][l] = id[s][l];
        }
        for (int c = 0; c < A.ncol; ++c)
        {
            if (mycol == c)
            {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (g + 4 * s < NB)
                    {
                        double acc[NB]; // loads first, then adds and stores (see helm_lane_kernel)
                        const int(&il)[NB] = ilc[s];
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            acc[l] = xy[cmp * ML + il[l]];
                        asm volatile("" ::: "memory"); // all loads before all stores, also in the generated code (the scheduler
                        __builtin_amdgcn_sched_barrier(0); // otherwise re-serialises them to save registers)
#pragma unroll
                        for (int l = 0; l < NB; ++l)
                            xy[cmp * ML + il[l]] = acc[l] + (cmp ? -OUT[0][s][l] : OUT[0][s][l]);
                    }
            }
            __syncthreads();
        }

        stamp(5); // colour phases done
        // ------------------------------------------------------------ boundary faces:  Au -= w H v,  Av -= w H u
        {
            const int f_begin = A.face_off[patch], nf = A.face_off[patch + 1] - f_begin;
            const int comp = cmp, le = lane;
            const double *xo = A.x + (size_t)(1 - comp) * A.ndof; // the other component, from global memory (few values)
            double *yc = xy + comp * ML;
            const int nqF = A.nqF;
            for (int f0 = 0; f0 < nf; f0 += 64)
            {
                const int f = f0 + le;
                const bool fa = f < nf;
                double res[NB];
                int fl[NB];
                int fc = -1;
#pragma unroll
                for (int k = 0; k < NB; ++k)
                {
                    res[k] = 0.0;
                    fl[k] = 0;
                }
                if (fa)
                {
                    const uint16_t *fli = A.face_lidx + (size_t)(f_begin + f) * NB;
                    const double *af = A.aF + (size_t)nqF * A.face_id[f_begin + f];
                    fc = A.face_col[f_begin + f];
                    double w[NB];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                    {
                        fl[k] = fli[k];
                        if constexpr (NATIVE)
                        {
                            const dbl2_t t = X2[fl[k] < nown ? own0 + fl[k] : bp[fl[k] - nown]];
                            w[k] = comp ? t.x : t.y; // the other component
                        }
                        else
                            w[k] = xo[dofs[fl[k]]];
                    }
                    for (int q = 0; q < nqF; ++q)
                    {
                        double pv = 0.0;
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            pv += PF[q + nqF * k] * w[k];
                        pv *= af[q];
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            res[k] += PF[q + nqF * k] * pv;
                    }
                }
                for (int c = 0; c < A.nfcol; ++c)
                {
                    if (fc == c)
                    {
#pragma unroll
                        for (int k = 0; k < NB; ++k)
                            yc[fl[k]] -= A.omega * res[k];
                    }
                    __syncthreads();
                }
            }
        }

        stamp(6); // faces done
        // ------------------------------------------------------------ write out
        if constexpr (NATIVE)
        {
            // both components of every local dof are in LDS: all 128 threads write pairs (the colour / face phases ended on a barrier)
            dbl2_t *Y2 = reinterpret_cast<dbl2_t *>(A.y), *P2 = reinterpret_cast<dbl2_t *>(A.part);
            for (int i = threadIdx.x; i < nloc; i += 128)
            {
                dbl2_t r;
                r.x = xy[i];
                r.y = xy[ML + i];
                if (i < nown)
                    Y2[own0 + i] = r;
                else
                    P2[bs[min(i - nown, bcap)]] = r;
            }
            stamp(7);
            return;
        }
        const int *slot = A.slot_of + off;
        constexpr int WROWS = 14;
        for (int base = 0; base < nloc; base += 64 * WROWS)
        {
            int dest[WROWS];
#pragma unroll
            for (int j = 0; j < WROWS; ++j)
                dest[j] = slot[min(base + 64 * j + lane, nloc - 1)];
#pragma unroll
            for (int j = 0; j < WROWS; ++j)
            {
                const int i = base + 64 * j + lane;
                if (i >= nloc)
                    continue;
                if (dest[j] >= 0)
                    A.y[(size_t)cmp * A.ndof + dest[j]] = xy[cmp * ML + i];
                else
                    A.part[2 * (size_t)(-dest[j] - 1) + cmp] = xy[cmp * ML + i]; // a slot is the pair (u, v); this wavefront holds one component
            }
        }
    }

    // How helm_mfma_kernel gets its metric data: 0 = one dependent trip to memory per slice; 100 w + 10 cs + cm = in cs + cm chunks
    // requested one ahead and parked in LDS, w wavefronts per SIMD.  Same-box A/B (profiles/r03/mfma_stage_ab.txt): n_basis 6 and 7
    // gain 6-9 % with 2 + 2 chunks; n_basis 8 (register spills at 3 wavefronts per SIMD, 2.5 by LDS) and n_basis 5 on the matrix cores
    // do not.  CUDDH_HELM_MFMA_STAGE overrides (measurement knob; 0 = off).
    int mfma_stage(int nb)
    {
        static const int knob = [] { const char *e = std::getenv("CUDDH_HELM_MFMA_STAGE"); return e ? std::atoi(e) : -1; }();
        if (knob >= 0)
            return knob;
        return (nb == 6 || nb == 7) ? 322 : 0;
    }

    template <int NB, int NQS, int NQM>
    void launch_helm_mfma(const cuddh_helmholtz_plan *p, const HelmArgs &A, hipStream_t st, bool native)
    {
        const dim3 grid(8 * A.xcd_chunk), block(128); // one wavefront per component
        const int stage = mfma_stage(NB);
        if (stage)
        {
            auto go = [&](auto nat, auto ncs, auto ncm, auto occ)
            {
                constexpr bool NAT = decltype(nat)::value;
                constexpr int NCS = decltype(ncs)::value, NCM = decltype(ncm)::value, OCC = decltype(occ)::value;
                constexpr int HS = (NQS + NCS - 1) / NCS, HM = (NQM + NCM - 1) / NCM;
                constexpr int CH = (HS * 3 * NQS > HM * NQM ? HS * 3 * NQS : HM * NQM) * 16;
                const size_t lds_s = ((size_t)2 * ((p->max_loc + 1) & ~1) + CH) * sizeof(double) + ((NB * NB + 1) / 2 * 16 + 16) * sizeof(int);
                hipLaunchKernelGGL((helm_mfma_kernel<NB, NQS, NQM, NAT, NCS, NCM, OCC>), grid, block, lds_s, st, A, p->PS, p->DS, p->PM, p->PF, p->Gm, p->gm_stride, p->Am,
                                   p->am_stride);
            };
            using std::integral_constant;
            using std::true_type;
            if (!native) // reference ordering: the default form only
            {
                go(std::false_type{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 3>{});
                return;
            }
#define CUDDH_STAGE_CASE(o, cs, cm)                                                                                  \
    if (stage == 100 * o + 10 * cs + cm)                                                                             \
    {                                                                                                                \
        go(true_type{}, integral_constant<int, cs>{}, integral_constant<int, cm>{}, integral_constant<int, o>{});   \
        return;                                                                                                      \
    }
            CUDDH_STAGE_CASE(3, 2, 2)
            CUDDH_STAGE_CASE(3, 2, 1)
            CUDDH_STAGE_CASE(3, 3, 2)
            CUDDH_STAGE_CASE(2, 2, 2)
#undef CUDDH_STAGE_CASE
            std::fprintf(stderr, "CUDDH_HELM_MFMA_STAGE=%d is not a built variant (322, 321, 332, 222)\n", stage);
            std::abort();
        }
        const size_t lds = (size_t)2 * p->max_loc * sizeof(double);
        if (native)
            hipLaunchKernelGGL((helm_mfma_kernel<NB, NQS, NQM, true>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gm, p->gm_stride, p->Am, p->am_stride);
        else
            hipLaunchKernelGGL((helm_mfma_kernel<NB, NQS, NQM, false>), grid, block, lds, st, A, p->PS, p->DS, p->PM, p->PF, p->Gm, p->gm_stride, p->Am, p->am_stride);
    }

    void launch_helm_mfma_any(const cuddh_helmholtz_plan *p, const HelmArgs &A, hipStream_t st, bool native)
    {
        if (p->nb == 5)
            launch_helm_mfma<5, 6, 9>(p, A, st, native);
        else if (p->nb == 6)
            launch_helm_mfma<6, 7, 11>(p, A, st, native);
        else if (p->nb == 7)
            launch_helm_mfma<7, 8, 12>(p, A, st, native);
        else
            launch_helm_mfma<8, 9, 14>(p, A, st, native);
    }

    __global__ void __launch_bounds__(256) op_border_kernel(int n_shared, int accumulate, const int *__restrict__ shared_dof,
                                                           const int *__restrict__ shared_off,
                                                           const double *__restrict__ part, double *__restrict__ y)
    {
        for (int j = blockIdx.x * 256 + threadIdx.x; j < n_shared; j += gridDim.x * 256)
        {
            const int g = shared_dof[j];
            // contiguous slots, patch order; the first four requested together (see helm_border_kernel)
            const int t0 = shared_off[j], t1 = shared_off[j + 1], last = max(t1 - 1, t0);
            double q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                q[k] = part[min(t0 + k, last)];
            double s = accumulate ? y[g] : 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s = t0 + k < t1 ? s + q[k] : s;
            for (int t = t0 + 4; t < t1; ++t)
                s += part[t];
            y[g] = s;
        }
    }

    // kind 0: stiffness with nq = nb + 1; kind 1: mass with nq = nb + 1 (a == 1) or 1 + 3 nb / 2 + 1 (weighted)
    // stiffness for 6 <= n_basis <= 8 runs on the fp64 matrix cores (op_mfma_kernel, 16-element batches)
    bool op_mfma(int kind, int nb, int nq)
    {
        if (nb < 6 || nb > 8)
            return false;
        return (kind == 0 && nq == nb + 1) || (kind == 1 && (nq == nb + 1 || nq == 2 + 3 * nb / 2));
    }

    bool op_supported(int kind, int nb, int nq)
    {
        if (op_mfma(kind, nb, nq))
            return true;
        if (nb < 2 || nb > 5) // one element per lane: n_basis 6 measured slower than the generic kernels (256 VGPRs + spills)
            return false;
        if (kind == 0)
            return nq == nb + 1;
        return kind == 1 && (nq == nb + 1 || nq == 2 + 3 * nb / 2);
    }

    template <int NB, int NQ, int KIND, int PEK>
    void launch_op_pe(const cuddh_helmholtz_plan *p, const HelmArgs &A, int accumulate, hipStream_t st)
    {
        const size_t lds = (size_t)(PEK == 64 ? 2 : 4) * p->max_loc * sizeof(double);
        const dim3 grid(8 * A.xcd_chunk), block(64);
        const double *P = KIND == 0 ? p->PS : p->PM, *MU = KIND == 0 ? p->Gu : p->au;
        if (MU)
            hipLaunchKernelGGL((op_patch_kernel<NB, NQ, KIND, false, true, PEK>), grid, block, lds, st, A, accumulate, P, p->DS, MU);
        else if (p->streaming)
            hipLaunchKernelGGL((op_patch_kernel<NB, NQ, KIND, true, false, PEK>), grid, block, lds, st, A, accumulate, P, p->DS, MU);
        else
            hipLaunchKernelGGL((op_patch_kernel<NB, NQ, KIND, false, false, PEK>), grid, block, lds, st, A, accumulate, P, p->DS, MU);
    }

    template <int NB, int NQ, int KIND>
    void launch_op_one(const cuddh_helmholtz_plan *p, const HelmArgs &A, int accumulate, hipStream_t st)
    {
        if (p->pe == 64)
            launch_op_pe<NB, NQ, KIND, 64>(p, A, accumulate, st);
        else
            launch_op_pe<NB, NQ, KIND, 32>(p, A, accumulate, st);
    }

    bool launch_op(const cuddh_helmholtz_plan *p, const HelmArgs &A, int accumulate, hipStream_t st)
    {
        const int kind = p->nqS > 0 ? 0 : 1, nq = kind == 0 ? p->nqS : p->nqM;
#define CUDDH_OP_CASE(NB_, NQ_, K_)                      \
    if (p->nb == NB_ && nq == NQ_ && kind == K_)         \
    {                                                    \
        launch_op_one<NB_, NQ_, K_>(p, A, accumulate, st); \
        return true;                                     \
    }
        CUDDH_OP_CASE(2, 3, 0)
        CUDDH_OP_CASE(3, 4, 0)
        CUDDH_OP_CASE(4, 5, 0)
        CUDDH_OP_CASE(5, 6, 0)
        CUDDH_OP_CASE(2, 3, 1)
        CUDDH_OP_CASE(3, 4, 1)
        CUDDH_OP_CASE(4, 5, 1)
        CUDDH_OP_CASE(5, 6, 1)
        CUDDH_OP_CASE(2, 5, 1)
        CUDDH_OP_CASE(3, 6, 1)
        CUDDH_OP_CASE(4, 8, 1)
        CUDDH_OP_CASE(5, 9, 1)
#undef CUDDH_OP_CASE
        return false;
    }
} // namespace

extern "C"
{
    int cuddh_hip_helmholtz_plan_destroy(cuddh_helmholtz_plan *p)
    {
        if (!p)
            return 0;
        void *ptrs[] = {p->own_count, p->dof_off, p->dof_list, p->slot_of, p->patch_nel, p->lidx, p->colour, p->Gp, p->aMp, p->Gu, p->au, p->Gm, p->Am, p->face_off,
                        p->face_lidx, p->face_id, p->face_col, p->PS, p->DS, p->PM, p->PF, p->shared_dof, p->shared_off,
                        p->part, p->stamps, p->own_off, p->bpos, p->bslot, p->global_of_native};
        for (void *q : ptrs)
            if (q)
                (void)hipFree(q);
        delete p;
        return 0;
    }

    // nqS == 0 or nqM == 0: a plan for a single real operator (the other tables are left empty)
    static int build_plan(cuddh_helmholtz_plan **out, int ndof, int n_elem, int nb, const int *h_I, const double *h_xy, int nqS,
                          const double *h_PS, const double *h_DS, const double *G_S, int nqM, const double *h_PM, const double *a_M,
                          int n_faces, const int *h_fI, const int *h_face_elem, int nqF, const double *h_PF, const double *a_F,
                          int pe = PE, bool want_pairs = false)
    {
        *out = nullptr;
        const bool mfma = pe == 16; // 16-element batches: the matrix-core kernels

        cuddh_helmholtz_plan *p = new cuddh_helmholtz_plan;
        p->ndof = ndof;
        p->n_elem = n_elem;
        p->nb = nb;
        p->nqS = nqS;
        p->nqM = nqM;
        p->nqF = nqF;
        p->n_faces = n_faces;
        p->aF = a_F;
        const int nn = nb * nb;

        // ---- element order: Morton curve over the centroids
        std::vector<int> perm(n_elem);
        for (int e = 0; e < n_elem; ++e)
            perm[e] = e;
        if (h_xy)
        {
            double lo[2] = {h_xy[0], h_xy[1]}, hi[2] = {h_xy[0], h_xy[1]};
            for (int e = 0; e < n_elem; ++e)
                for (int a = 0; a < 2; ++a)
                {
                    lo[a] = std::min(lo[a], h_xy[2 * e + a]);
                    hi[a] = std::max(hi[a], h_xy[2 * e + a]);
                }
            std::vector<uint64_t> key(n_elem);
            for (int e = 0; e < n_elem; ++e)
            {
                uint32_t c[2];
                for (int a = 0; a < 2; ++a)
                {
                    const double span = hi[a] - lo[a];
                    const double t = span > 0 ? (h_xy[2 * e + a] - lo[a]) / span : 0.0;
                    c[a] = static_cast<uint32_t>(std::min(65535.0, std::max(0.0, t * 65535.0 + 0.5)));
                }
                key[e] = (static_cast<uint64_t>(spread_bits(c[0]) | (spread_bits(c[1]) << 1)) << 32) | static_cast<uint32_t>(e);
            }
            std::sort(key.begin(), key.end());
            for (int e = 0; e < n_elem; ++e)
                perm[e] = static_cast<int>(key[e] & 0xFFFFFFFFu);
        }

        const int n_patches = (n_elem + pe - 1) / pe;
        p->n_patches = n_patches;
        p->pe = pe;
        std::vector<int> padded_perm((size_t)n_patches * pe, -1);
        std::copy(perm.begin(), perm.end(), padded_perm.begin());
        std::vector<int> patch_of_elem(n_elem);
        for (int pos = 0; pos < n_elem; ++pos)
            patch_of_elem[perm[pos]] = pos / pe;

        // ---- faces bucketed by the patch of their element
        std::vector<int> face_off(n_patches + 1, 0);
        for (int f = 0; f < n_faces; ++f)
            face_off[patch_of_elem[h_face_elem[f]] + 1]++;
        for (int q = 0; q < n_patches; ++q)
            face_off[q + 1] += face_off[q];
        std::vector<int> face_id(n_faces);
        {
            std::vector<int> cursor(face_off.begin(), face_off.end() - 1);
            for (int f = 0; f < n_faces; ++f)
                face_id[cursor[patch_of_elem[h_face_elem[f]]]++] = f;
        }

        // ---- patch-local numbering, colours
        std::vector<int> dof_off(n_patches + 1, 0), dof_list, patch_nel(n_patches);
        const int np2 = (nn + 1) / 2;
        std::vector<uint32_t> lidx((size_t)n_patches * np2 * pe, 0);
        std::vector<uint16_t> face_lidx((size_t)n_faces * nb, 0);
        std::vector<uint8_t> colour((size_t)n_patches * pe, 0), face_col(n_faces, 0);
        std::vector<int> stamp(ndof, -1), loc(ndof, 0), touches(ndof, 0);
        std::vector<uint32_t> used, usedF;
        int max_loc = 0, ncol = 1, nfcol = 1;
        dof_list.reserve((size_t)n_elem * nn / 2);
        // how many patches touch a dof (a dof touched by one patch is OWNED by it: its result goes straight to y)
        for (int q = 0; q < n_patches; ++q)
        {
            const int nel = std::min(pe, n_elem - q * pe);
            for (int le = 0; le < nel; ++le)
            {
                const int *gi = h_I + (size_t)nn * perm[q * pe + le];
                for (int n = 0; n < nn; ++n)
                    if (stamp[gi[n]] != q)
                    {
                        stamp[gi[n]] = q;
                        touches[gi[n]]++;
                    }
            }
        }
        std::fill(stamp.begin(), stamp.end(), -1);
        // Patch-local numbering: the owned dofs first, the border dofs (shared with other patches) last, each group in the
        // order its dofs are first met.  The write-out of an owned dof then needs no destination entry -- it is the gather index
        // -- so a kernel that knows own_count reads only the tail of the patch's slot_of segment (helm_lane_kernel does).
        std::vector<int> own_count(n_patches), border_first;
        for (int q = 0; q < n_patches; ++q)
        {
            const int first = static_cast<int>(dof_list.size());
            dof_off[q] = first;
            const int nel = std::min(pe, n_elem - q * pe);
            patch_nel[q] = nel;
            border_first.clear();
            for (int le = 0; le < nel; ++le)
            {
                const int *gi = h_I + (size_t)nn * perm[q * pe + le];
                for (int n = 0; n < nn; ++n)
                {
                    const int g = gi[n];
                    if (stamp[g] == q)
                        continue;
                    stamp[g] = q;
                    if (touches[g] > 1)
                        border_first.push_back(g);
                    else
                    {
                        loc[g] = static_cast<int>(dof_list.size()) - first;
                        dof_list.push_back(g);
                    }
                }
            }
            own_count[q] = static_cast<int>(dof_list.size()) - first;
            for (const int g : border_first)
            {
                loc[g] = static_cast<int>(dof_list.size()) - first;
                dof_list.push_back(g);
            }
            used.assign(dof_list.size() - first, 0);
            for (int le = 0; le < nel; ++le)
            {
                const int *gi = h_I + (size_t)nn * perm[q * pe + le];
                uint32_t taken = 0;
                for (int n = 0; n < nn; ++n)
                {
                    const int g = gi[n];
                    lidx[((size_t)q * np2 + n / 2) * pe + le] |= static_cast<uint32_t>(loc[g]) << (16 * (n & 1));
                    taken |= used[loc[g]];
                }
                int c = 0;
                while (c < 31 && (taken >> c & 1u))
                    ++c;
                colour[(size_t)q * pe + le] = static_cast<uint8_t>(c);
                ncol = std::max(ncol, c + 1);
                for (int n = 0; n < nn; ++n)
                    used[loc[gi[n]]] |= 1u << c;
            }
            const int nloc = static_cast<int>(dof_list.size()) - first;
            if (nloc > 65535)
            {
                cuddh_hip_helmholtz_plan_destroy(p);
                return static_cast<int>(hipErrorInvalidValue);
            }
            max_loc = std::max(max_loc, nloc);

            usedF.assign(nloc, 0);
            for (int t = face_off[q]; t < face_off[q + 1]; ++t)
            {
                const int *fg = h_fI + (size_t)nb * face_id[t];
                uint32_t taken = 0;
                for (int k = 0; k < nb; ++k)
                {
                    if (stamp[fg[k]] != q)
                    {
                        cuddh_hip_helmholtz_plan_destroy(p);
                        return static_cast<int>(hipErrorInvalidValue); // face dof not in its element's patch
                    }
                    face_lidx[(size_t)t * nb + k] = static_cast<uint16_t>(loc[fg[k]]);
                    taken |= usedF[loc[fg[k]]];
                }
                int c = 0;
                while (c < 31 && (taken >> c & 1u))
                    ++c;
                face_col[t] = static_cast<uint8_t>(c);
                nfcol = std::max(nfcol, c + 1);
                for (int k = 0; k < nb; ++k)
                    usedF[loc[fg[k]]] |= 1u << c;
            }
        }
        dof_off[n_patches] = static_cast<int>(dof_list.size());
        p->max_loc = max_loc;
        p->ncol = ncol;
        p->nfcol = n_faces > 0 ? nfcol : 0;

        // ---- dofs touched by more than one patch get one slot per touching patch
        std::vector<int> shared_index(ndof, -1), shared_dof, shared_off(1, 0);
        for (int g = 0; g < ndof; ++g)
            if (touches[g] > 1)
            {
                shared_index[g] = static_cast<int>(shared_dof.size());
                shared_dof.push_back(g);
                shared_off.push_back(shared_off.back() + touches[g]);
            }
        const int n_shared = static_cast<int>(shared_dof.size());
        const int n_slots = shared_off.back();
        std::vector<int> slot_of(dof_list.size()), fill(shared_off.begin(), shared_off.end() - 1);
        for (size_t i = 0; i < dof_list.size(); ++i)
        {
            const int j = shared_index[dof_list[i]];
            // owned: the global dof itself; border: -(slot) - 1, the slots of one dof being contiguous and ordered by patch
            slot_of[i] = j >= 0 ? -(fill[j]++) - 1 : dof_list[i];
        }
        p->n_shared = n_shared;
        p->n_slots = n_slots;
        // ---- upload
        int err = 0;
        auto ok = [&](int e)
        {
            if (e && !err)
                err = e;
        };
        // ---- plan-native vector ordering (lane-form plans): owned dofs patch by patch, then the border dofs in shared_dof order
        size_t native_list_entries = 0;
        if (nqS > 0 && nqM > 0) // every fused plan; which kernels take native vectors: cuddh_hip_helmholtz_plan_has_native
        {
            std::vector<int> own_off(n_patches + 1, 0);
            for (int q = 0; q < n_patches; ++q)
                own_off[q + 1] = own_off[q] + own_count[q];
            const int n_owned = own_off[n_patches];
            int bstride = 1;
            for (int q = 0; q < n_patches; ++q)
                bstride = std::max(bstride, dof_off[q + 1] - dof_off[q] - own_count[q]);
            std::vector<int> bpos((size_t)n_patches * bstride, 0), bslot((size_t)n_patches * bstride, 0), g_of_n(ndof, -1);
            for (int q = 0; q < n_patches; ++q)
            {
                const int nloc = dof_off[q + 1] - dof_off[q], nb_q = nloc - own_count[q];
                for (int i = 0; i < own_count[q]; ++i)
                    g_of_n[own_off[q] + i] = dof_list[dof_off[q] + i];
                for (int t = 0; t < bstride; ++t)
                {
                    const int i = own_count[q] + std::min(t, std::max(nb_q - 1, 0));
                    if (nb_q == 0)
                        continue; // no border dofs: the padding is never read (clamped index 0, value 0 = a valid position)
                    bpos[(size_t)q * bstride + t] = n_owned + shared_index[dof_list[dof_off[q] + i]];
                    bslot[(size_t)q * bstride + t] = -slot_of[dof_off[q] + i] - 1;
                }
                native_list_entries += nb_q;
            }
            for (int j = 0; j < n_shared; ++j)
                g_of_n[n_owned + j] = shared_dof[j];
            bool perm_ok = n_owned + n_shared == ndof;
            for (int n = 0; n < ndof && perm_ok; ++n)
                perm_ok = g_of_n[n] >= 0;
            if (perm_ok) // (a dof no element touches would break the permutation: such a space keeps the reference ordering only)
            {
                p->n_owned = n_owned;
                p->bstride = bstride;
                ok(upload(&p->own_off, own_off));
                ok(upload(&p->bpos, bpos));
                ok(upload(&p->bslot, bslot));
                ok(upload(&p->global_of_native, g_of_n));
            }
        }
        ok(upload(&p->dof_off, dof_off));
        // Fixed stride for the per-patch lists (helm_lane_kernel, helm_patch_kernel, op_patch_kernel with one patch per wavefront):
        // segment p starts at p * max_loc and is padded with its last entry, so a kernel can request its first indices without
        // waiting for dof_off -- one dependent (scalar) round trip less at the head of every wavefront's chain.  The matrix-core
        // kernels and op_patch_kernel with two patches per wavefront keep the packed lists.
        const bool fixed_stride = !mfma && (pe == 64 || (nqS > 0 && nqM > 0));
        if (fixed_stride)
        {
            std::vector<int> dl((size_t)n_patches * max_loc), so((size_t)n_patches * max_loc);
            for (int q = 0; q < n_patches; ++q)
            {
                const int n = dof_off[q + 1] - dof_off[q];
                for (int i = 0; i < max_loc; ++i)
                {
                    dl[(size_t)q * max_loc + i] = dof_list[dof_off[q] + std::min(i, n - 1)];
                    so[(size_t)q * max_loc + i] = slot_of[dof_off[q] + std::min(i, n - 1)];
                }
            }
            p->dof_stride = max_loc;
            ok(upload(&p->dof_list, dl));
            ok(upload(&p->slot_of, so));
        }
        else
        {
            ok(upload(&p->dof_list, dof_list));
            ok(upload(&p->slot_of, slot_of));
        }
        ok(upload(&p->own_count, own_count));
        ok(upload(&p->patch_nel, patch_nel));
        ok(upload(&p->lidx, lidx));
        ok(upload(&p->colour, colour));
        ok(upload(&p->face_off, face_off));
        ok(upload(&p->face_lidx, face_lidx));
        ok(upload(&p->face_id, face_id));
        ok(upload(&p->face_col, face_col));
        ok(upload(&p->shared_dof, shared_dof));
        ok(upload(&p->shared_off, shared_off));
        if (nqS > 0)
        {
            ok(upload_raw(&p->PS, h_PS, (size_t)nqS * nb));
            ok(upload_raw(&p->DS, h_DS, (size_t)nqS * nb));
        }
        if (nqM > 0)
            ok(upload_raw(&p->PM, h_PM, (size_t)nqM * nb));
        if (n_faces > 0)
            ok(upload_raw(&p->PF, h_PF, (size_t)nqF * nb));
        if (n_slots > 0)
            ok(static_cast<int>(hipMalloc(reinterpret_cast<void **>(&p->part), (size_t)2 * n_slots * sizeof(double))));

        int *d_perm = nullptr;
        ok(upload(&d_perm, padded_perm));
        // affine meshes: one copy of a metric array instead of one per element (CUDDH_PLAN_AFFINE=0 keeps the general form)
        bool try_affine = true;
        if (const char *e = std::getenv("CUDDH_PLAN_AFFINE"))
            try_affine = std::atoi(e) != 0;
        bool uniform_G = false, uniform_a = false; // matrix-core plans: a uniform array becomes ONE 16-element block (stride 0)
        if (mfma && try_affine)
        {
            double *probe = nullptr;
            if (nqS > 0)
            {
                ok(uniform_table(&probe, 3, nqS, n_elem, G_S));
                uniform_G = probe != nullptr;
                if (probe)
                    (void)hipFree(probe);
            }
            if (nqM > 0)
            {
                probe = nullptr;
                ok(uniform_table(&probe, 1, nqM, n_elem, a_M));
                uniform_a = probe != nullptr;
                if (probe)
                    (void)hipFree(probe);
            }
        }
        if (try_affine && !mfma && nqS > 0)
            ok(uniform_table(&p->Gu, 3, nqS, n_elem, G_S));
        if (try_affine && !mfma && nqM > 0 && nqS == 0) // the fused complex kernel always reads per-element mass weights (they carry a(x)^2)
            ok(uniform_table(&p->au, 1, nqM, n_elem, a_M));
        int *d_zero = nullptr; // "element 0 in all 16 lanes": the one block of a uniform array
        if (uniform_G || uniform_a)
        {
            ok(static_cast<int>(hipMalloc(reinterpret_cast<void **>(&d_zero), (size_t)pe * sizeof(int))));
            if (d_zero)
                ok(static_cast<int>(hipMemset(d_zero, 0, (size_t)pe * sizeof(int))));
        }
        // the lane form of the fused apply (64-element patches, general geometry -- or the affine n_basis-2 form, whose mass
        // weights are still per element) reads its slices with 16-byte loads
        p->pair_layout = (want_pairs && !mfma && pe == 64 && nqS > 0 && nqM > 0 && (!p->Gu || nb == 2)) ? 1 : 0;
        const long long nG = p->Gu ? 0 : (long long)(uniform_G ? 1 : n_patches) * 3 * nqS * nqS * pe;
        const long long nA = p->au ? 0 : (long long)(uniform_a ? 1 : n_patches) * nqM * nqM * pe;
        if (nG > 0)
            ok(static_cast<int>(hipMalloc(reinterpret_cast<void **>(mfma ? &p->Gm : &p->Gp), nG * sizeof(double))));
        if (nA > 0)
            ok(static_cast<int>(hipMalloc(reinterpret_cast<void **>(mfma ? &p->Am : &p->aMp), nA * sizeof(double))));
        if (!err)
        {
            if (nG > 0 && mfma)
                hipLaunchKernelGGL(repack_mfma_kernel, dim3(stream_grid(nG, 256)), dim3(256), 0, nullptr, nG, 3, nqS, uniform_G ? d_zero : d_perm, G_S,
                                   p->Gm);
            else if (nG > 0)
                hipLaunchKernelGGL(repack_kernel, dim3(stream_grid(nG, 256)), dim3(256), 0, nullptr, nG, 3, nqS, pe, d_perm, G_S, p->Gp, p->pair_layout);
            if (nA > 0 && mfma)
                hipLaunchKernelGGL(repack_mfma_kernel, dim3(stream_grid(nA, 256)), dim3(256), 0, nullptr, nA, 1, nqM, uniform_a ? d_zero : d_perm, a_M,
                                   p->Am);
            else if (nA > 0)
                hipLaunchKernelGGL(repack_kernel, dim3(stream_grid(nA, 256)), dim3(256), 0, nullptr, nA, 1, nqM, pe, d_perm, a_M, p->aMp, p->pair_layout);
            ok(launch_status());
            ok(static_cast<int>(hipDeviceSynchronize()));
        }
        if (d_zero)
            (void)hipFree(d_zero);
        if (d_perm)
            (void)hipFree(d_perm);
        if (err)
        {
            cuddh_hip_helmholtz_plan_destroy(p);
            return err;
        }

        const bool single = nqS == 0 || nqM == 0; // one real vector in, one out
        p->bytes_alg = (size_t)n_elem * ((size_t)3 * nqS * nqS * 8 + (size_t)nqM * nqM * 8 + (size_t)nn * 4) +
                       (size_t)ndof * (single ? 16 : 32) + (size_t)n_faces * ((size_t)nqF * 8 + (size_t)nb * 4);
        size_t exclusive = 0;
        for (int s : slot_of)
            exclusive += s >= 0;
        // destination list of the write-out (slot_of): the plan kernels read it only for the rows of local dofs that hold border
        // dofs -- rows of 64 (helm_lane_kernel, op_patch_kernel with one patch per wavefront) or of 2 pe (helm_patch_kernel);
        // the matrix-core kernels and op_patch_kernel with two patches per wavefront read all of it
        size_t dest_entries = dof_list.size();
        const bool fused_plan = nqS > 0 && nqM > 0;
        const int dest_row = mfma ? 0 : (fused_plan ? (p->pair_layout ? 64 : 2 * pe) : (pe == 64 ? 64 : 0));
        if (dest_row > 0)
        {
            dest_entries = n_patches; // own_count
            for (int q = 0; q < n_patches; ++q)
                dest_entries += (dof_off[q + 1] - dof_off[q]) - (own_count[q] / dest_row) * dest_row;
        }
        const size_t dest_bytes = dest_entries * 4;
        p->bytes_actual = (size_t)nG * 8 + (size_t)nA * 8 + lidx.size() * 4 + colour.size() + dof_list.size() * (4 + 16) + dest_bytes + // dof (gather), x
                          exclusive * 16 + (size_t)n_slots * (16 + 16) + (size_t)n_shared * (16 + 8) +
                          (size_t)n_faces * ((size_t)nqF * 8 + (size_t)nb * 2 + 5);
        if (p->own_off) // the native apply: no dof list for owned dofs, two short lists for the border dofs, x gathered once per touching patch
            p->bytes_native = (size_t)nG * 8 + (size_t)nA * 8 + lidx.size() * 4 + colour.size() + (size_t)(n_patches + 1) * 4 + native_list_entries * 8 +
                              dof_list.size() * 16 + exclusive * 16 + (size_t)n_slots * (16 + 16) + (size_t)n_shared * (16 + 4) +
                              (size_t)n_faces * ((size_t)nqF * 8 + (size_t)nb * 2 + 5);
        if (p->Gu || p->au) // SURVEY 8d's "affine" figure: the uniform metric arrays are not traffic
            p->bytes_affine = p->bytes_alg - (size_t)n_elem * ((p->Gu ? (size_t)3 * nqS * nqS * 8 : 0) + (p->au ? (size_t)nqM * nqM * 8 : 0));
        if (mfma)
        {
            p->gm_stride = uniform_G ? 0 : (long long)3 * nqS * nqS * pe;
            p->am_stride = uniform_a ? 0 : (long long)nqM * nqM * pe;
            if (uniform_G || uniform_a)
                p->bytes_affine = p->bytes_alg - (size_t)n_elem * ((uniform_G ? (size_t)3 * nqS * nqS * 8 : 0) + (uniform_a ? (size_t)nqM * nqM * 8 : 0));
        }
        p->streaming = p->bytes_actual > (size_t)256 << 20; // the infinity cache
        if (const char *e = std::getenv("CUDDH_PLAN_STREAMING")) // measurement knob: 0 / 1 overrides the size rule
            p->streaming = std::atoi(e) != 0;
        *out = p;
        return 0;
    }

    int cuddh_hip_helmholtz_plan_create(cuddh_helmholtz_plan **out, int ndof, int n_elem, int nb, const int *h_I, const double *h_xy,
                                        int nqS, const double *h_PS, const double *h_DS, const double *G_S, int nqM,
                                        const double *h_PM, const double *a_M, int n_faces, const int *h_fI, const int *h_face_elem,
                                        int nqF, const double *h_PF, const double *a_F)
    {
        *out = nullptr;
        if (!supported(nb, nqS, nqM) || n_elem <= 0)
            return static_cast<int>(hipErrorNotSupported);
        int pe = helm_mfma(nb, nqS, nqM) ? 16 : PE;
        // n_basis 5 sits between the two schemes: one element per lane needs more registers than 3 wavefronts per SIMD have
        // (26 spilled), the matrix-core scheme pads 5 xi-indices to 8.  Measured (profiles/r03/config5_ab.txt, 768^2): one element
        // per lane 385 us, matrix cores 479 us (reference ordering) / 425 us (native ordering) -- the default stays;
        // CUDDH_HELM_NB5_MFMA=1 selects the matrix-core scheme (tests keep it correct)
        if (nb == 5 && nqS == 6 && nqM == 9)
            if (const char *e = std::getenv("CUDDH_HELM_NB5_MFMA"))
                if (std::atoi(e) != 0)
                    pe = 16;
        if (pe == PE && nb <= 4)
        {
            // Affine plans (uniform stiffness metric, read through scalar loads) use 64-element patches, two wavefronts sharing
            // one LDS copy: a third fewer border dofs and slots.  Measured at 1024^2: n_basis 4 331 -> 316 us, n_basis 3
            // 137 -> 130 us; with per-element metrics the larger patch is a wash on structured meshes and 6 % slower on
            // the irregular one (the two waves wait for each other at every colour phase), n_basis 5 does not change.
            const char *a = std::getenv("CUDDH_PLAN_AFFINE");
            double *probe = nullptr;
            if (!(a && std::atoi(a) == 0) && uniform_table(&probe, 3, nqS, n_elem, G_S) == 0 && probe)
                pe = 64;
            if (probe)
                (void)hipFree(probe);
            if (const char *e = std::getenv("CUDDH_HELM_PE")) // measurement knob
                pe = std::atoi(e) == 64 ? 64 : PE;
        }
        // General geometry, n_basis 4, at least two full rounds of wavefronts (4096 patches of 64 elements = 512^2 elements):
        // helm_lane_kernel.  Same-box A/B against helm_patch_kernel: 1024^2 403 -> 382 us, 512^2 102.5 -> 91.6 us, irregular
        // 1.95 M quads 817 -> 764 us; at 256^2 (half a round) it is 10 % slower, hence the size rule.  CUDDH_HELM_LANE=0/1 overrides.
        // n_basis 3 (151 VGPRs, 3 waves/SIMD): 1024^2 213 -> 185 us (4.75 TB/s), a wash at 512^2 where the plan fits the
        // infinity cache: from 8192 patches on.
        // n_basis 2 (92 VGPRs, 5 waves/SIMD): 1024^2 124 -> 103 us, 2048^2 510 -> 418-438 us.
        // Affine plans: only n_basis 2 gains from the lane form (1024^2: 80 -> 68 us); n_basis 3 loses (133 -> 154 us) and
        // n_basis 4 loses a lot (311 -> 365 us: with no metric traffic the kernel lives on occupancy).
        const bool affine_plan = pe == 64; // decided above
        const bool affine_lane = affine_plan && nb == 2;
        bool lane_form = (pe == PE && ((nb == 4 && n_elem >= 4096 * 64) || (nb <= 3 && n_elem >= 8192 * 64))) || (affine_lane && n_elem >= 8192 * 64);
        if (const char *e = std::getenv("CUDDH_HELM_LANE"))
            lane_form = nb <= 4 && (pe == PE || affine_lane) && std::atoi(e) == 1;
        if (lane_form)
            pe = 64;
        const int err = build_plan(out, ndof, n_elem, nb, h_I, h_xy, nqS, h_PS, h_DS, G_S, nqM, h_PM, a_M, n_faces, h_fI, h_face_elem, nqF,
                                   h_PF, a_F, pe, lane_form);
        if (!err && *out)
        {
            (*out)->lane_form = (*out)->pair_layout; // = lane_form && (general geometry || affine n_basis 2): decided in build_plan
            // the lane form with the whole metric block in the register file (PRE, one wavefront per SIMD): measured SLOWER
            // than the slice-by-slice chain at two wavefronts per SIMD (1024^2, n_basis 4: 434-442 vs 373-376 us), although
            // the metric stream alone runs at 6.46 TB/s that way -- see the comment at the kernel.  CUDDH_HELM_PRE=1 selects it
            // for A/B runs; tests keep it correct.
            (*out)->pair_mass = nb == 3 && !(*out)->Gu && !(*out)->lane_form;
            if (const char *e = std::getenv("CUDDH_HELM_PAIR_MASS"))
                (*out)->pair_mass = (*out)->pair_mass && std::atoi(e) != 0;
            (*out)->prefetch = 0;
            if (const char *e = std::getenv("CUDDH_HELM_PRE"))
                (*out)->prefetch = (*out)->lane_form && std::atoi(e) != 0;
            if (std::getenv("CUDDH_HELM_STAMPS") && ((*out)->lane_form || (*out)->pe == 16)) // diagnostic, see cuddh_hip_helmholtz_plan_read_stamps
                if (hipMalloc(reinterpret_cast<void **>(&(*out)->stamps), (size_t)(*out)->n_patches * 8 * sizeof(unsigned long long)) == hipSuccess)
                    (void)hipMemset((*out)->stamps, 0, (size_t)(*out)->n_patches * 8 * sizeof(unsigned long long));
        }
        return err;
    }

    int cuddh_hip_operator_plan_create(cuddh_helmholtz_plan **out, int kind, int ndof, int n_elem, int nb, const int *h_I,
                                       const double *h_xy, int nq, const double *h_P, const double *h_D, const double *metric)
    {
        *out = nullptr;
        if (n_elem <= 0 || !op_supported(kind, nb, nq))
            return static_cast<int>(hipErrorNotSupported);
        // n_basis 2-5: one 64-element patch per wavefront (8x8 elements on a structured mesh).  Two 32-element patches per
        // wavefront (CUDDH_OP_PE=32) have a third more border dofs and slots: 1024^2, n_basis 4, general layout: stiffness
        // 174-187 -> 162-171 us, mass 109 -> 97 us, weighted mass 168 -> 153 us; affine 113 / 87 / 98 -> 98 / 75 / 82 us.
        int pe = op_mfma(kind, nb, nq) ? 16 : 64;
        if (const char *e = std::getenv("CUDDH_OP_PE")) // measurement knob
            pe = (pe != 16 && std::atoi(e) == 32) ? PE : pe;
        if (kind == 0)
            return build_plan(out, ndof, n_elem, nb, h_I, h_xy, nq, h_P, h_D, metric, 0, nullptr, nullptr, 0, nullptr, nullptr, 0,
                              nullptr, nullptr, pe);
        return build_plan(out, ndof, n_elem, nb, h_I, h_xy, 0, nullptr, nullptr, nullptr, nq, h_P, metric, 0, nullptr, nullptr, 0,
                          nullptr, nullptr, pe);
    }

    int cuddh_hip_operator_plan_apply(const cuddh_helmholtz_plan *p, double c, int accumulate, const double *x, double *y, void *stream)
    {
        if (!p || (p->nqS > 0) == (p->nqM > 0))
            return static_cast<int>(hipErrorInvalidValue);
        hipStream_t st = as_stream(stream);
        HelmArgs A = plan_args(p, x, y);
        A.omega = c;
        if (p->Gm || p->Am) // 16-element batches on the fp64 matrix cores, one batch per wavefront
        {
            const size_t lds = (size_t)2 * p->max_loc * sizeof(double);
            const dim3 grid(8 * A.xcd_chunk), block(64);
            const int kind = p->nqS > 0 ? 0 : 1, nq = kind == 0 ? p->nqS : p->nqM;
            const double *P = kind == 0 ? p->PS : p->PM;
            bool launched = false;
#define CUDDH_MFMA_CASE(NB_, NQ_, K_)                                                                              \
    if (!launched && p->nb == NB_ && nq == NQ_ && kind == K_)                                                      \
    {                                                                                                              \
        hipLaunchKernelGGL((op_mfma_kernel<NB_, NQ_, K_>), grid, block, lds, st, A, accumulate, P, p->DS, K_ == 0 ? p->Gm : p->Am,  \
                           K_ == 0 ? p->gm_stride : p->am_stride);  \
        launched = true;                                                                                           \
    }
            CUDDH_MFMA_CASE(6, 7, 0)
            CUDDH_MFMA_CASE(7, 8, 0)
            CUDDH_MFMA_CASE(8, 9, 0)
            CUDDH_MFMA_CASE(6, 7, 1)
            CUDDH_MFMA_CASE(7, 8, 1)
            CUDDH_MFMA_CASE(8, 9, 1)
            CUDDH_MFMA_CASE(6, 11, 1)
            CUDDH_MFMA_CASE(7, 12, 1)
            CUDDH_MFMA_CASE(8, 14, 1)
#undef CUDDH_MFMA_CASE
            if (!launched)
                return static_cast<int>(hipErrorNotSupported);
        }
        else
        {
            const int n_waves = p->pe == 64 ? p->n_patches : (p->n_patches + 1) / 2; // two 32-element patches per wavefront
            A.xcd_chunk = (n_waves + 7) / 8;
            if (!launch_op(p, A, accumulate, st))
                return static_cast<int>(hipErrorNotSupported);
        }
        int err = launch_status();
        if (err)
            return err;
        if (p->n_shared > 0)
        {
            hipLaunchKernelGGL(op_border_kernel, dim3(stream_grid(p->n_shared, 256)), dim3(256), 0, st, p->n_shared, accumulate,
                               p->shared_dof, p->shared_off, p->part, y);
            err = launch_status();
        }
        return err;
    }

    int cuddh_hip_helmholtz_apply(const cuddh_helmholtz_plan *p, double omega, const double *x, double *y, void *stream)
    {
        if (!p)
            return static_cast<int>(hipErrorInvalidValue);
        hipStream_t st = as_stream(stream);
        HelmArgs A = plan_args(p, x, y);
        A.omega = omega;

        if (p->Gm && p->Am) // n_basis 6-8 (5 on request): fp64 matrix cores, one 16-element batch per workgroup
            launch_helm_mfma_any(p, A, st, false);
        else if (p->nb == 4)
            launch_patch<4, 5, 8>(p, A, st);
        else if (p->nb == 3)
            launch_patch<3, 4, 6>(p, A, st);
        else if (p->nb == 5)
            launch_patch<5, 6, 9>(p, A, st);
        else
            launch_patch<2, 3, 5>(p, A, st);
        int err = launch_status();
        if (err)
            return err;
        if (p->n_shared > 0)
        {
            hipLaunchKernelGGL(helm_border_kernel, dim3(stream_grid(p->n_shared, 256)), dim3(256), 0, st, p->n_shared, p->ndof, p->n_slots,
                               p->shared_dof, p->shared_off, p->part, y);
            err = launch_status();
        }
        return err;
    }

    // ---- plan-native vector ordering (see cuddh_helmholtz_plan): for solvers that keep their vectors in the plan's order
    int cuddh_hip_helmholtz_plan_has_native(const cuddh_helmholtz_plan *p)
    {
        if (!p || !p->own_off)
            return 0;
        return 1; // every fused kernel takes native vectors: helm_lane_kernel, helm_patch_kernel, helm_mfma_kernel
    }

    int cuddh_hip_helmholtz_to_native(const cuddh_helmholtz_plan *p, const double *x, double *z, void *stream)
    {
        if (!cuddh_hip_helmholtz_plan_has_native(p))
            return static_cast<int>(hipErrorNotSupported);
        hipLaunchKernelGGL(to_native_kernel, dim3(stream_grid(p->ndof, 256)), dim3(256), 0, as_stream(stream), p->ndof, p->global_of_native, x,
                           reinterpret_cast<dbl2_t *>(z));
        return launch_status();
    }

    int cuddh_hip_helmholtz_from_native(const cuddh_helmholtz_plan *p, const double *z, double *y, void *stream)
    {
        if (!cuddh_hip_helmholtz_plan_has_native(p))
            return static_cast<int>(hipErrorNotSupported);
        hipLaunchKernelGGL(from_native_kernel, dim3(stream_grid(p->ndof, 256)), dim3(256), 0, as_stream(stream), p->ndof, p->global_of_native,
                           reinterpret_cast<const dbl2_t *>(z), y);
        return launch_status();
    }

    int cuddh_hip_helmholtz_apply_native(const cuddh_helmholtz_plan *p, double omega, const double *z_in, double *z_out, void *stream)
    {
        if (!cuddh_hip_helmholtz_plan_has_native(p))
            return static_cast<int>(hipErrorNotSupported);
        if (z_in == z_out)
            return static_cast<int>(hipErrorInvalidValue);
        hipStream_t st = as_stream(stream);
        HelmArgs A = plan_args(p, z_in, z_out);
        A.omega = omega;
        if (p->Gm && p->Am)
            launch_helm_mfma_any(p, A, st, true);
        else if (p->nb == 4)
            launch_patch<4, 5, 8>(p, A, st, true);
        else if (p->nb == 3)
            launch_patch<3, 4, 6>(p, A, st, true);
        else if (p->nb == 5)
            launch_patch<5, 6, 9>(p, A, st, true);
        else
            launch_patch<2, 3, 5>(p, A, st, true);
        int err = launch_status();
        if (err)
            return err;
        if (p->n_shared > 0)
        {
            hipLaunchKernelGGL(helm_border_native_kernel, dim3(stream_grid(p->n_shared, 256)), dim3(256), 0, st, p->n_shared, p->n_owned, p->shared_off,
                               reinterpret_cast<const dbl2_t *>(p->part), reinterpret_cast<dbl2_t *>(z_out));
            err = launch_status();
        }
        return err;
    }

    // diagnostic (CUDDH_HELM_STAMPS=1 at plan creation): copies the [n_patches][8] phase stamps of the last lane-form apply
    int cuddh_hip_helmholtz_plan_read_stamps(const cuddh_helmholtz_plan *p, unsigned long long *h_out, int n_patches)
    {
        if (!p || !p->stamps || n_patches > p->n_patches)
            return static_cast<int>(hipErrorInvalidValue);
        return static_cast<int>(hipMemcpy(h_out, p->stamps, (size_t)n_patches * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }

    int cuddh_hip_helmholtz_plan_describe(const cuddh_helmholtz_plan *p, char *buf, int cap)
    {
        if (!p || !buf || cap <= 0)
            return static_cast<int>(hipErrorInvalidValue);
        // mirrors the dispatch of cuddh_hip_helmholtz_apply / cuddh_hip_operator_plan_apply above
        const bool fused = p->nqS > 0 && p->nqM > 0;
        const int kind = p->nqS > 0 ? 0 : 1, nq = kind == 0 ? p->nqS : p->nqM;
        const bool nt = p->streaming != 0;
        if (fused && p->Gm && p->Am)
            {
                const int stg = mfma_stage(p->nb);
                if (stg)
                    std::snprintf(buf, cap, "helm_mfma_kernel<%d,%d,%d,chunks=%d+%d> pe=16 affine=%d", p->nb, p->nqS, p->nqM, (stg / 10) % 10, stg % 10, p->gm_stride == 0 ? 1 : 0);
                else
                    std::snprintf(buf, cap, "helm_mfma_kernel<%d,%d,%d> pe=16 affine=%d", p->nb, p->nqS, p->nqM, p->gm_stride == 0 ? 1 : 0);
            }
        else if (fused && p->nb <= 4 && p->pe == 64 && p->lane_form)
            std::snprintf(buf, cap, "helm_lane_kernel<%d,%d,%d,NT=%d,UG=%d%s> pe=64", p->nb, p->nqS, p->nqM, nt, p->Gu ? 1 : 0, p->prefetch ? ",PRE=1" : "");
        else if (fused)
            std::snprintf(buf, cap, "helm_patch_kernel<%d,%d,%d,NT=%d,UG=%d,PEK=%d%s> pe=%d", p->nb, p->nqS, p->nqM, nt, p->Gu ? 1 : 0, p->pe,
                          p->pair_mass ? ",MODE=1" : "", p->pe);
        else if (p->Gm || p->Am)
            std::snprintf(buf, cap, "op_mfma_kernel<%d,%d,%d> pe=16 affine=%d", p->nb, nq, kind, (kind == 0 ? p->gm_stride : p->am_stride) == 0 ? 1 : 0);
        else
        {
            const bool mu = kind == 0 ? p->Gu != nullptr : p->au != nullptr;
            std::snprintf(buf, cap, "op_patch_kernel<%d,%d,%d,NT=%d,UG=%d,PEK=%d> pe=%d", p->nb, nq, kind, (!mu && nt) ? 1 : 0, mu ? 1 : 0, p->pe, p->pe);
        }
        return 0;
    }

    size_t cuddh_hip_helmholtz_plan_bytes(const cuddh_helmholtz_plan *p, int actual)
    {
        if (!p)
            return 0;
        return actual == 3 ? p->bytes_native : (actual == 2 ? p->bytes_affine : (actual ? p->bytes_actual : p->bytes_alg));
    }
}
