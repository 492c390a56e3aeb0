// DDH set-up (host) and the three entry points rhs / action / postprocess.
// Set-up rules restated from reference source/DDH.cpp:323-609; the local solves
// themselves are the HIP kernels behind cuddh_hip_ddh_apply_*.
#include "cuddh/ddh.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>

#include "cuddh/parallel.hpp"
#include "cuddh_hip.h"

namespace cuddh
{
    namespace detail
    {
        namespace
        {
            template <typename Real>
            void geom_setup(int n_domains, int mx_elems, int nb, const int *n_elems, const int *elems, const double *w,
                            const double *points, const double *corners, Real *G)
            {
                int err;
                if constexpr (std::is_same_v<Real, float>)
                    err = cuddh_hip_ddh_geom_from_corners_f32(n_domains, mx_elems, nb, n_elems, elems, w, points, corners, G, stream());
                else
                    err = cuddh_hip_ddh_geom_from_corners_f64(n_domains, mx_elems, nb, n_elems, elems, w, points, corners, G, stream());
                check_hip(err, "DDH geometric factors");
            }
        } // namespace

        template <typename Real>
        DDHCore<Real>::DDHCore(double omega_, const double *h_a, const H1Space &fem, int nx, int ny, int kernel)
            : g_ndof(fem.size()), g_elem(fem.mesh().n_elem()), n_basis(fem.basis().size()), omega(omega_), fem_mesh(&fem.mesh()),
              fem_basis(&fem.basis())
        {
            const int nb = n_basis;
            if (nb < 2 || nb > 16)
                cuddh_error("DDH error: n_basis must be in [2, 16].");

            // ---- blocks of nel1d x nel1d elements (16 / nb per side: <= 256 nodes per subdomain)
            nel1d = std::max(1, 16 / nb);
            if (nx % nel1d != 0 || ny % nel1d != 0)
                cuddh_error("DDH error: nx and ny must be multiples of 16 / n_basis.");
            if (nx * ny != g_elem)
                cuddh_error("DDH error: nx * ny does not match the mesh.");

            PhaseTimer timer;
            const int ndx = nx / nel1d, ndy = ny / nel1d;
            n_domains = ndx * ndy;
            std::vector<int> labels(static_cast<std::size_t>(nx) * ny);
            parallel_for(static_cast<std::size_t>(ny), [&](std::size_t j0, std::size_t j1, int)
            {
                for (int j = static_cast<int>(j0); j < static_cast<int>(j1); ++j)
                    for (int i = 0; i < nx; ++i)
                        labels[i + static_cast<std::size_t>(nx) * j] = (i / nel1d) + ndx * (j / nel1d);
            }, 8);

            efem.reset(new EnsembleSpace(fem, n_domains, labels.data()));
            timer.lap("EnsembleSpace");

            // ---- WaveHoltz time grid: dt = 0.1 h / nb^2 shrunk so that nt dt is one period
            const double T = 2.0 * M_PI / omega;
            const double h = fem.mesh().min_h();
            dt = 0.2 * 0.5 * h / (nb * nb);
            nt = static_cast<int>(std::ceil(T / dt));
            dt = T / nt;

            _wh_filter.resize(nt + 1);
            Real *filt = _wh_filter.host_write();
            for (int k = 0; k <= nt; ++k)
                filt[k] = static_cast<Real>(dt * (omega / M_PI) * (std::cos(omega * k * dt) - 0.25));
            filt[0] = static_cast<Real>(filt[0] * 0.5); // trapezoid rule end points
            filt[nt] = static_cast<Real>(filt[nt] * 0.5);

            _cs.resize(2 * nt + 1);
            _sn.resize(2 * nt + 1);
            Real *cs = _cs.host_write();
            Real *sn = _sn.host_write();
            for (int k = 0; k <= 2 * nt; ++k)
            {
                const double t = 0.5 * k * dt;
                cs[k] = static_cast<Real>(-std::cos(omega * t));
                sn[k] = static_cast<Real>(std::sin(omega * t));
            }

            timer.lap("time grid tables");
            // ---- extents
            auto sizes = efem->sizes(MemorySpace::HOST);
            auto fsizes = efem->fsizes(MemorySpace::HOST);
            auto s_nel = efem->n_elems(MemorySpace::HOST);
            mx_dof = mx_fdof = mx_elem_per_dom = 0;
            for (int s = 0; s < n_domains; ++s)
            {
                mx_dof = std::max(mx_dof, sizes(s));
                mx_fdof = std::max(mx_fdof, fsizes(s));
                mx_elem_per_dom = std::max(mx_elem_per_dom, s_nel(s));
            }

            // ---- trace slots.  lambda = (l0, l1, m0, m1): pair k of cmap owns slot k on its first
            // subdomain's side and slot n_shared + k on the second's; each side reads its own slot and
            // writes the other's.  Later pairs overwrite earlier ones at cross points (reference quirk:
            // source/DDH.cpp:436-439), which leaves a few slots unused.
            auto cmap = efem->connectivity_map(MemorySpace::HOST);
            const int n_shared = cmap.shape(1);
            n_lambda = 2 * n_shared;

            _Bf.resize(2 * mx_fdof * n_domains);
            auto B = reshape(_Bf.host_write(), mx_fdof, 2, n_domains);
            std::fill(B.begin(), B.end(), -1);
            for (int k = 0; k < n_shared; ++k)
            {
                const int S0 = cmap(0, k), S1 = cmap(1, k), j0 = cmap(2, k), j1 = cmap(3, k);
                B(j0, 0, S0) = k;
                B(j0, 1, S0) = n_shared + k;
                B(j1, 0, S1) = n_shared + k;
                B(j1, 1, S1) = k;
            }

            timer.lap("trace slots");
            // ---- renumber each subdomain so that its face dofs come first (in face-space order)
            auto faceproj = efem->face_proj(MemorySpace::HOST);
            auto g_inds = efem->global_indices(MemorySpace::HOST);
            auto s_inds = efem->subspace_indices(MemorySpace::HOST);
            auto s_elems = efem->elements(MemorySpace::HOST);

            _gI.resize(mx_dof * n_domains);
            auto gI = reshape(_gI.host_write(), mx_dof, n_domains);
            _sI.resize(nb * nb * mx_elem_per_dom * n_domains);
            auto sI = reshape(_sI.host_write(), nb, nb, mx_elem_per_dom, n_domains);

            parallel_for(static_cast<std::size_t>(n_domains), [&](std::size_t s0, std::size_t s1, int)
            {
                std::vector<int> new_of_old(mx_dof);
                for (int s = static_cast<int>(s0); s < static_cast<int>(s1); ++s)
                {
                    const int nd = sizes(s), nf = fsizes(s);
                    std::fill(new_of_old.begin(), new_of_old.end(), -1);
                    int next = 0;
                    for (; next < nf; ++next)
                    {
                        const int old = faceproj(next, s);
                        new_of_old[old] = next;
                        gI(next, s) = g_inds(old, s);
                    }
                    for (int old = 0; old < nd; ++old)
                        if (new_of_old[old] < 0)
                        {
                            new_of_old[old] = next;
                            gI(next, s) = g_inds(old, s);
                            ++next;
                        }
                    for (int el = 0; el < s_nel(s); ++el)
                        for (int l = 0; l < nb; ++l)
                            for (int k = 0; k < nb; ++k)
                                sI(k, l, el, s) = new_of_old[s_inds(k, l, el, s)];
                }
            }, 16);

            timer.lap("face-first renumbering");
            // ---- local operators
            const Basis &basis = fem.basis();
            const QuadratureRule &q = basis.quadrature();

            _D.resize(nb * nb);
            {
                dmat Dd(nb, nb);
                basis.deriv(nb, q.x(), Dd);
                Real *hD = _D.host_write();
                for (int i = 0; i < nb * nb; ++i)
                    hD[i] = static_cast<Real>(Dd[i]);
            }

            // det J at the GLL points, evaluated where it is used (the reference reads the tabulated array,
            // source/DDH.cpp:551-552; tabulating 16 n_elem doubles on the host costs more than the two loops that read them)
            const Mesh2D &the_mesh = fem.mesh();
            auto detJ = [&](int i, int j, int el)
            {
                const double xi[2] = {q.x(i), q.x(j)};
                return the_mesh.element(el)->measure(xi);
            };
            auto fem_gi = fem.global_indices(MemorySpace::HOST);

            timer.lap("element metrics");
            // global lumped mass and its inverse.  The reference adds the element contributions of a dof in element order
            // (source/DDH.cpp:559-566); to keep that order (and with it every bit of the sum) with several threads, a dof
            // touched by ONE range of elements is summed by that range, the few touched by several ranges afterwards, serially.
            std::vector<double> inv_mass(g_ndof, 0.0);
            {
                const int nn = nb * nb;
                std::vector<std::atomic<int>> toucher(g_ndof); // -1 none, c one range, -2 several
                parallel_for(static_cast<std::size_t>(g_ndof), [&](std::size_t g0, std::size_t g1, int)
                {
                    for (std::size_t g = g0; g < g1; ++g)
                        toucher[g].store(-1, std::memory_order_relaxed);
                });
                parallel_for(static_cast<std::size_t>(g_elem), [&](std::size_t e0, std::size_t e1, int c)
                {
                    for (std::size_t v = e0 * nn; v < e1 * nn; ++v)
                    {
                        std::atomic<int> &t = toucher[fem_gi[v]];
                        int cur = t.load(std::memory_order_relaxed);
                        while (cur != c && cur != -2 && !t.compare_exchange_weak(cur, cur == -1 ? c : -2, std::memory_order_relaxed))
                        {
                        }
                    }
                }, 64);
                const int C = chunk_count(static_cast<std::size_t>(g_elem), 64);
                std::vector<std::vector<int>> seam(C); // elements with a dof that several ranges touch, per range, in order
                parallel_for(static_cast<std::size_t>(g_elem), [&](std::size_t e0, std::size_t e1, int c)
                {
                    for (int el = static_cast<int>(e0); el < static_cast<int>(e1); ++el)
                    {
                        bool on_seam = false;
                        for (int j = 0; j < nb; ++j)
                            for (int i = 0; i < nb; ++i)
                            {
                                const int g = fem_gi(i, j, el);
                                if (toucher[g].load(std::memory_order_relaxed) == -2)
                                    on_seam = true;
                                else
                                    inv_mass[g] += q.w(i) * q.w(j) * detJ(i, j, el);
                            }
                        if (on_seam)
                            seam[c].push_back(el);
                    }
                }, 64);
                for (int c = 0; c < C; ++c)
                    for (const int el : seam[c])
                        for (int j = 0; j < nb; ++j)
                            for (int i = 0; i < nb; ++i)
                            {
                                const int g = fem_gi(i, j, el);
                                if (toucher[g].load(std::memory_order_relaxed) == -2)
                                    inv_mass[g] += q.w(i) * q.w(j) * detJ(i, j, el);
                            }
                parallel_for(static_cast<std::size_t>(g_ndof), [&](std::size_t g0, std::size_t g1, int)
                {
                    for (std::size_t g = g0; g < g1; ++g)
                        inv_mass[g] = 1.0 / inv_mass[g];
                });
            }

            _m.resize(mx_dof * n_domains);
            _H.resize(mx_fdof * n_domains);
            _a.resize(mx_dof * n_domains);
            _gmi.resize(mx_dof * n_domains);
            auto m = reshape(_m.host_write(), mx_dof, n_domains);
            auto H = reshape(_H.host_write(), mx_fdof, n_domains);
            auto A = reshape(_a.host_write(), mx_dof, n_domains);
            auto gmi = reshape(_gmi.host_write(), mx_dof, n_domains);

            auto faces = efem->faces(MemorySpace::HOST);
            auto n_faces = efem->n_faces(MemorySpace::HOST);
            auto f_inds = efem->face_indices(MemorySpace::HOST);

            parallel_for(static_cast<std::size_t>(n_domains), [&](std::size_t s0, std::size_t s1, int)
            {
                for (int s = static_cast<int>(s0); s < static_cast<int>(s1); ++s)
                {
                    // the storage type accumulates, as in the reference (float += double)
                    for (int el = 0; el < s_nel(s); ++el)
                    {
                        const int g_el = s_elems(el, s);
                        for (int j = 0; j < nb; ++j)
                            for (int i = 0; i < nb; ++i)
                            {
                                Real &acc = m(sI(i, j, el, s), s);
                                acc = static_cast<Real>(acc + q.w(i) * q.w(j) * detJ(i, j, g_el));
                            }
                    }
                    for (int i = 0; i < sizes(s); ++i)
                    {
                        A(i, s) = static_cast<Real>(h_a[gI(i, s)]);
                        gmi(i, s) = static_cast<Real>(inv_mass[gI(i, s)]);
                    }
                    for (int f = 0; f < n_faces(s); ++f)
                    {
                        const Edge *edge = fem.mesh().edge(faces(f, s));
                        for (int i = 0; i < nb; ++i)
                        {
                            Real &acc = H(f_inds(i, f, s), s);
                            acc = static_cast<Real>(acc + edge->measure(q.x(i)) * q.w(i));
                        }
                    }
                }
            }, 16);

            timer.lap("lumped masses, H, a");
            requested_kernel = kernel;
        }

        template <typename Real>
        DDHCore<Real>::~DDHCore()
        {
            if (plan)
                cuddh_hip_ddh_plan_destroy(plan);
        }

        template <typename Real>
        void DDHCore<Real>::ensure_plan() const
        {
            if (plan)
                return;
            PhaseTimer timer;
            const int nb = n_basis;
            const QuadratureRule &q = fem_basis->quadrature();

            // geometric factors G (3, nb*nb*mx_elems, n_domains) from the Jacobians at the GLL points
            host_device_dvec w(nb);
            double *hw = w.host_write();
            for (int i = 0; i < nb; ++i)
                hw[i] = q.w(i);
            // (the reference reads the tabulated Jacobians, source/DDH.cpp:530-537; here the kernel evaluates the bilinear map from
            // the corners: the 0.5 GB table at 1024^2 is neither built nor kept)
            const auto &metrics = fem_mesh->element_metrics(q);
            const double *d_corners = metrics.corner_coordinates_device();
            check_hip(cuddh_hip_stream_sync(stream()), "DDH element corners");
            timer.lap("plan: element corners to the device");
            _g_tensor.resize(3 * nb * nb * mx_elem_per_dom * n_domains);
            geom_setup<Real>(n_domains, mx_elem_per_dom, nb, efem->n_elems(MemorySpace::DEVICE), efem->elements(MemorySpace::DEVICE),
                             w.device_read(), metrics.rule_nodes_device(), d_corners, _g_tensor.device_write());
            check_hip(cuddh_hip_stream_sync(stream()), "DDH geometric factors");
            timer.lap("plan: geometric factors");

            cuddh_ddh_desc d;
            d.g_ndof = g_ndof;
            d.n_domains = n_domains;
            d.n_lambda = n_lambda;
            d.nb = nb;
            d.nel1d = nel1d;
            d.mx_dof = mx_dof;
            d.mx_fdof = mx_fdof;
            d.nt = nt;
            d.omega = omega;
            d.dt = dt;
            d.s_dof = efem->sizes(MemorySpace::DEVICE);
            d.s_fdof = efem->fsizes(MemorySpace::DEVICE);
            d.B = _Bf.device_read();
            d.gI = _gI.device_read();
            d.sI = _sI.device_read();
            d.D = _D.device_read();
            d.G = _g_tensor.device_read();
            d.m = _m.device_read();
            d.gmi = _gmi.device_read();
            d.a = _a.device_read();
            d.H = _H.device_read();
            d.wh_filter = _wh_filter.device_read();
            d.cs = _cs.device_read();
            d.sn = _sn.device_read();
            check_hip(cuddh_hip_stream_sync(stream()), "DDH table upload");
            timer.lap("plan: table uploads");
            check_hip(cuddh_hip_ddh_plan_create(&plan, &d, std::is_same_v<Real, double> ? 1 : 0, requested_kernel),
                      "DDH plan");
            timer.lap("plan: structure check + kernel tables");
        }

        template <typename Real>
        int DDHCore<Real>::kernel_kind() const
        {
            ensure_plan();
            return cuddh_hip_ddh_plan_kernel(plan);
        }

        template <typename Real>
        void DDHCore<Real>::set_waveholtz_iterations(int n) const
        {
            ensure_plan();
            check_hip(cuddh_hip_ddh_plan_set_wh_iters(plan, n), "DDH WaveHoltz iterations");
        }

        template <typename Real>
        void DDHCore<Real>::set_wave_priority(bool high) const
        {
            ensure_plan();
            check_hip(cuddh_hip_ddh_plan_set_wave_priority(plan, high ? 1 : 0), "DDH wave priority");
        }

        template <typename Real>
        void DDHCore<Real>::ensure_assembly() const
        {
            if (_csr_off.size() > 0)
                return;
            const int N = mx_dof * n_domains;
            const int *gi = _gI.host_read();
            auto sizes = efem->sizes(MemorySpace::HOST);
            _gI_local.resize(N);
            int *loc = _gI_local.host_write();
            parallel_for(static_cast<std::size_t>(N), [&](std::size_t i0, std::size_t i1, int)
            {
                for (std::size_t i = i0; i < i1; ++i)
                    loc[i] = static_cast<int>(i);
            });
            _csr_off.resize(g_ndof + 1);
            int *off = _csr_off.host_write();
            std::fill(off, off + g_ndof + 1, 0);
            for (int s = 0; s < n_domains; ++s)
                for (int l = 0; l < sizes(s); ++l)
                    ++off[gi[l + static_cast<std::size_t>(mx_dof) * s] + 1];
            for (int g = 0; g < g_ndof; ++g)
                off[g + 1] += off[g];
            _csr_src.resize(off[g_ndof]);
            int *src = _csr_src.host_write();
            std::vector<int> fill(off, off + g_ndof);
            for (int s = 0; s < n_domains; ++s) // increasing subdomain: the order of a serial loop over the subdomains
                for (int l = 0; l < sizes(s); ++l)
                    src[fill[gi[l + static_cast<std::size_t>(mx_dof) * s]]++] = l + mx_dof * s;
            _x_local.resize(2 * N);
            _y_local.resize(2 * N);
        }

        template <typename Real>
        void DDHCore<Real>::solve(int d0, int d1, const double *x, double *y, bool zero_y, const Real *lambda, Real *update) const
        {
            solve_impl(nullptr, d0, d1, x, y, zero_y, lambda, update);
        }

        template <typename Real>
        void DDHCore<Real>::solve_listed(const int *d_domains, int n, const double *x, double *y, bool zero_y, const Real *lambda,
                                         Real *update) const
        {
            solve_impl(d_domains, 0, n, x, y, zero_y, lambda, update);
        }

        template <typename Real>
        void DDHCore<Real>::solve_impl(const int *d_list, int d0, int d1, const double *x, double *y, bool zero_y, const Real *lambda,
                                       Real *update) const
        {
            ensure_plan();
            auto run = [&](const double *xx, double *yy, bool zy)
            {
                int err;
                if constexpr (std::is_same_v<Real, float>)
                    err = d_list ? cuddh_hip_ddh_apply_list_f32(plan, d_list, d1, xx, yy, zy ? 1 : 0, lambda, update, stream())
                                 : cuddh_hip_ddh_apply_f32(plan, d0, d1, xx, yy, zy ? 1 : 0, lambda, update, stream());
                else
                    err = d_list ? cuddh_hip_ddh_apply_list_f64(plan, d_list, d1, xx, yy, zy ? 1 : 0, lambda, update, stream())
                                 : cuddh_hip_ddh_apply_f64(plan, d0, d1, xx, yy, zy ? 1 : 0, lambda, update, stream());
                check_hip(err, "DDH local solves");
            };
            if (!y)
            {
                run(x, nullptr, false);
                return;
            }
            // With a solution output the reference adds every subdomain's weighted values into y with atomics
            // (source/DDH.cpp:298-307), in whatever order the blocks finish.  Here the subdomains write into their own entries
            // (identity numbering) and a second kernel sums the copies of each global dof in increasing subdomain order:
            // postprocess is bitwise reproducible and equals a serial loop over the subdomains.
            // (footprint of this fixed-order assembly: two arrays of 2 * mx_dof * n_domains doubles, 0.54 GB at 1024^2 with n_basis 4,
            // kept for the lifetime of the object once a solution output was asked for; index arithmetic is 32-bit)
            if (2LL * mx_dof * n_domains > 2147483647LL)
                cuddh_error("DDH error: solution assembly needs 2 * mx_dof * n_domains < 2^31 (about 8192^2 elements at n_basis 4).");
            ensure_assembly();
            const int N = mx_dof * n_domains;
            const int *d_loc = _gI_local.device_read();
            double *xl = nullptr;
            if (x)
            {
                xl = _x_local.device_write();
                check_hip(cuddh_hip_gather_f64(N, _gI.device_read(), x, xl, stream()), "DDH forcing gather");
                check_hip(cuddh_hip_gather_f64(N, _gI.device_read(), x + g_ndof, xl + N, stream()), "DDH forcing gather");
            }
            double *yl = _y_local.device_write();
            check_hip(cuddh_hip_ddh_plan_set_vector_layout(plan, d_loc, N), "DDH local layout");
            try
            {
                run(xl, yl, true);
            }
            catch (...)
            {
                cuddh_hip_ddh_plan_set_vector_layout(plan, nullptr, 0);
                throw;
            }
            check_hip(cuddh_hip_ddh_plan_set_vector_layout(plan, nullptr, 0), "DDH layout reset");
            const int *off = _csr_off.device_read(), *src = _csr_src.device_read();
            check_hip(cuddh_hip_csr_sum_f64(g_ndof, off, src, yl, y, zero_y ? 0 : 1, stream()), "DDH solution assembly");
            check_hip(cuddh_hip_csr_sum_f64(g_ndof, off, src, yl + N, y + g_ndof, zero_y ? 0 : 1, stream()), "DDH solution assembly");
        }

        template <typename Real>
        void DDHCore<Real>::solve_listed(const int *d_domains, int n, const double *x, const Real *lambda, Real *update) const
        {
            solve_impl(d_domains, 0, n, x, nullptr, false, lambda, update);
        }

        template class DDHCore<float>;
        template class DDHCore<double>;
    } // namespace detail

    // ------------------------------------------------------------ DDH (fp32, reference precision)

    DDH::DDH(double omega, const double *h_a, const H1Space &fem, int nx, int ny) : core(omega, h_a, fem, nx, ny, 0) {}
    DDH::DDH(double omega, const double *h_a, const H1Space &fem, int nx, int ny, int kernel)
        : core(omega, h_a, fem, nx, ny, kernel)
    {
    }

    void DDH::action(const float *x, float *y) const
    {
        core.solve(0, core.num_domains(), nullptr, nullptr, false, x, y);
        axpby(size(), 1.0f, x, -1.0f, y); // y = lambda - T lambda
    }

    void DDH::rhs(const double *f, float *b) const { core.solve(0, core.num_domains(), f, nullptr, false, nullptr, b); }

    void DDH::postprocess(const float *lambda, const double *f, double *u) const
    {
        core.solve(0, core.num_domains(), f, u, true, lambda, nullptr);
    }

    void DDH::local_traces(int d0, int d1, const double *f, const float *lambda, float *update) const
    {
        core.solve(d0, d1, f, nullptr, false, lambda, update);
    }

    void DDH::local_solution(int d0, int d1, const float *lambda, const double *f, double *u, bool zero_u) const
    {
        core.solve(d0, d1, f, u, zero_u, lambda, nullptr);
    }

    // ------------------------------------------------------------ DDH64 (fp64 parity mode)

    DDH64::DDH64(double omega, const double *h_a, const H1Space &fem, int nx, int ny, int kernel)
        : core(omega, h_a, fem, nx, ny, kernel)
    {
    }

    void DDH64::action(const double *x, double *y) const
    {
        core.solve(0, core.num_domains(), nullptr, nullptr, false, x, y);
        axpby(size(), 1.0, x, -1.0, y);
    }

    void DDH64::action(double, const double *, double *) const
    {
        cuddh_error("DDH64::action(c, x, y) is not defined for the substructured operator.");
    }

    void DDH64::rhs(const double *f, double *b) const { core.solve(0, core.num_domains(), f, nullptr, false, nullptr, b); }

    void DDH64::postprocess(const double *lambda, const double *f, double *u) const
    {
        core.solve(0, core.num_domains(), f, u, true, lambda, nullptr);
    }

    void DDH64::local_traces(int d0, int d1, const double *f, const double *lambda, double *update) const
    {
        core.solve(d0, d1, f, nullptr, false, lambda, update);
    }

    void DDH64::local_solution(int d0, int d1, const double *lambda, const double *f, double *u, bool zero_u) const
    {
        core.solve(d0, d1, f, u, zero_u, lambda, nullptr);
    }
} // namespace cuddh
