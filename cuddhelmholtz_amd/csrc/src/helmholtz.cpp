// Complex Helmholtz operator: fused plan-based apply and the launch-by-launch composite.
#include "cuddh/helmholtz.hpp"

#include <vector>

#include "cuddh_hip.h"

namespace cuddh
{
    HelmholtzOperator::HelmholtzOperator(double omega_, const double *a2x, const double *ax, const H1Space &fem_, const FaceSpace &fs_)
        : omega(omega_), ndof(fem_.size()), fdof(fs_.size()), fem(fem_), fs(fs_), S(fem_), M(a2x, fem_), H(ax, fs_), xf(fs_.size()),
          yf(fs_.size())
    {
        const int nb = fem.basis().size();
        const int n_elem = fem.mesh().n_elem();
        const int n_faces = fs.n_faces();

        // element centroids: only used to group elements into compact patches
        std::vector<double> centroid(static_cast<std::size_t>(2) * n_elem);
        const double mid[2] = {0.0, 0.0};
        for (int el = 0; el < n_elem; ++el)
            fem.mesh().element(el)->physical_coordinates(mid, centroid.data() + 2 * el);

        // face node -> H1 dof
        std::vector<int> face_to_h1(static_cast<std::size_t>(nb) * n_faces);
        const int *fI = fs.subspace_indices(MemorySpace::HOST);
        const int *proj = fs.global_indices(MemorySpace::HOST);
        for (std::size_t t = 0; t < face_to_h1.size(); ++t)
            face_to_h1[t] = proj[fI[t]];
        std::vector<int> face_elem(n_faces);
        const int *face_ids = fs.faces(MemorySpace::HOST);
        for (int f = 0; f < n_faces; ++f)
            face_elem[f] = fem.mesh().edge(face_ids[f])->elements[0];

        const int err = cuddh_hip_helmholtz_plan_create(&plan, ndof, n_elem, nb, fem.global_indices(MemorySpace::HOST),
                                                        centroid.data(), S.quad_size(), S.P().host_read(), S.D().host_read(),
                                                        S.G().device_read(), M.quad_size(), M.P().host_read(),
                                                        M.weights().device_read(), n_faces, face_to_h1.data(), face_elem.data(),
                                                        H.quad_size(), H.P().host_read(), H.weights().device_read());
        if (err == 801) // hipErrorNotSupported: no specialised kernel for this (nb, nq) -- use the separate operators
            plan = nullptr;
        else
            detail::check_hip(err, "HelmholtzOperator plan");
    }

    HelmholtzOperator::~HelmholtzOperator()
    {
        if (plan)
            cuddh_hip_helmholtz_plan_destroy(plan);
    }

    void HelmholtzOperator::action(const double *x, double *y) const
    {
        if (!plan)
        {
            action_unfused(x, y);
            return;
        }
        detail::check_hip(cuddh_hip_helmholtz_apply(plan, omega, x, y, stream()), "HelmholtzOperator::action");
    }

    void HelmholtzOperator::action(double, const double *, double *) const
    {
        cuddh_error("HelmholtzOperator::action(c, x, y) not implemented");
    }

    void HelmholtzOperator::action_unfused(const double *x, double *y) const
    {
        const double *u = x, *v = x + ndof;
        double *Au = y, *Av = y + ndof;
        double *d_xf = xf.device_write();
        double *d_yf = yf.device_write();

        S.action(u, Au);
        S.action(v, Av);
        M.action(-omega * omega, u, Au);
        M.action(-omega * omega, v, Av);

        zeros(fdof, d_yf);
        fs.restrict(v, d_xf);
        H.action(-omega, d_xf, d_yf);
        fs.prolong(d_yf, Au);

        zeros(fdof, d_yf);
        fs.restrict(u, d_xf);
        H.action(omega, d_xf, d_yf);
        fs.prolong(d_yf, Av);

        scal(ndof, -1.0, Av);
    }

    std::string HelmholtzOperator::kernel_name() const
    {
        char buf[128] = "unfused";
        if (plan)
            cuddh_hip_helmholtz_plan_describe(plan, buf, sizeof buf);
        return buf;
    }

    bool HelmholtzOperator::has_native() const { return plan && cuddh_hip_helmholtz_plan_has_native(plan); }

    void HelmholtzOperator::to_native(const double *x, double *z) const
    {
        detail::check_hip(cuddh_hip_helmholtz_to_native(plan, x, z, stream()), "HelmholtzOperator::to_native");
    }

    void HelmholtzOperator::from_native(const double *z, double *y) const
    {
        detail::check_hip(cuddh_hip_helmholtz_from_native(plan, z, y, stream()), "HelmholtzOperator::from_native");
    }

    void HelmholtzOperator::action_native(const double *z_in, double *z_out) const
    {
        detail::check_hip(cuddh_hip_helmholtz_apply_native(plan, omega, z_in, z_out, stream()), "HelmholtzOperator::action_native");
    }

    namespace
    {
        class NativeView : public Operator, public QueuesDeviceWorkOnly
        {
        public:
            explicit NativeView(const HelmholtzOperator &A_) : A(A_) {}
            void action(const double *x, double *y) const override { A.action_native(x, y); }
            void action(double, const double *, double *) const override { cuddh_error("HelmholtzOperator (native view): action(c, x, y) not implemented"); }

        private:
            const HelmholtzOperator &A;
        };
    } // namespace

    solver_out HelmholtzOperator::gmres(double *x, const double *b, int m, int maxit, double tol, int verbose, double max_seconds) const
    {
        const int n = 2 * ndof;
        if (!has_native())
            return cuddh::gmres(n, x, this, b, m, maxit, tol, verbose, max_seconds);
        host_device_dvec zx(n), zb(n);
        double *d_zx = zx.device_write(), *d_zb = zb.device_write();
        to_native(x, d_zx);
        to_native(b, d_zb);
        NativeView V(*this);
        solver_out out = cuddh::gmres(n, d_zx, &V, d_zb, m, maxit, tol, verbose, max_seconds);
        from_native(d_zx, x);
        return out;
    }

    std::size_t HelmholtzOperator::bytes_native() const { return cuddh_hip_helmholtz_plan_bytes(plan, 3); }

    std::size_t HelmholtzOperator::bytes_affine() const { return cuddh_hip_helmholtz_plan_bytes(plan, 2); }

    std::size_t HelmholtzOperator::bytes_per_apply(bool actual) const
    {
        return cuddh_hip_helmholtz_plan_bytes(plan, actual ? 1 : 0);
    }
} // namespace cuddh
