// Operator classes: table set-up on the host, kernels through the C ABI.
#include "cuddh/operators.hpp"

#include <cstdlib>
#include <vector>

#include "cuddh_hip.h"

namespace cuddh
{
    namespace detail
    {
        OperatorPlan::~OperatorPlan()
        {
            if (plan)
                cuddh_hip_helmholtz_plan_destroy(plan);
        }

        const cuddh_helmholtz_plan *OperatorPlan::get(int kind, const H1Space &fem, int n_quad, const double *h_P, const double *h_D,
                                                      const double *d_metric) const
        {
            if (tried)
                return plan;
            tried = true;
            if (const char *e = std::getenv("CUDDH_OPERATOR_PLAN"))
                if (std::atoi(e) == 0)
                    return nullptr;
            const int n_elem = fem.mesh().n_elem();
            // element centroids: only used to group elements into compact patches
            std::vector<double> centroid(static_cast<std::size_t>(2) * n_elem);
            const double mid[2] = {0.0, 0.0};
            for (int el = 0; el < n_elem; ++el)
                fem.mesh().element(el)->physical_coordinates(mid, centroid.data() + 2 * el);
            const int err = cuddh_hip_operator_plan_create(&plan, kind, fem.size(), n_elem, fem.basis().size(),
                                                           fem.global_indices(MemorySpace::HOST), centroid.data(), n_quad, h_P, h_D,
                                                           d_metric);
            if (err == 801) // hipErrorNotSupported: the generic kernels cover every (n_basis, n_quad)
                plan = nullptr;
            else
                check_hip(err, "operator plan");
            return plan;
        }

        std::size_t OperatorPlan::bytes(bool actual) const { return cuddh_hip_helmholtz_plan_bytes(plan, actual ? 1 : 0); }
        std::size_t OperatorPlan::bytes_affine() const { return cuddh_hip_helmholtz_plan_bytes(plan, 2); }
        std::string OperatorPlan::kernel_name() const
        {
            char buf[128] = "generic";
            if (plan)
                cuddh_hip_helmholtz_plan_describe(plan, buf, sizeof buf);
            return buf;
        }
    } // namespace detail

    namespace
    {
        host_device_dvec rule_weights(const QuadratureRule &quad)
        {
            host_device_dvec w(quad.size());
            double *h = w.host_write();
            for (int i = 0; i < quad.size(); ++i)
                h[i] = quad.w(i);
            return w;
        }
    } // namespace

    // ------------------------------------------------------------ StiffnessMatrix

    StiffnessMatrix::StiffnessMatrix(const H1Space &fem_)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()),
          n_quad(fem_.mesh().max_element_order() + fem_.basis().size()), _P(n_quad * n_basis), _D(n_quad * n_basis),
          _G(3 * n_quad * n_quad * n_elem)
    {
        setup(QuadratureRule(n_quad, QuadratureRule::GaussLegendre));
    }

    StiffnessMatrix::StiffnessMatrix(const H1Space &fem_, const QuadratureRule &quad)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()), n_quad(quad.size()),
          _P(n_quad * n_basis), _D(n_quad * n_basis), _G(3 * n_quad * n_quad * n_elem)
    {
        setup(quad);
    }

    void StiffnessMatrix::setup(const QuadratureRule &quad)
    {
        fem.basis().eval(n_quad, quad.x(), _P.host_write());
        fem.basis().deriv(n_quad, quad.x(), _D.host_write());
        const double *J = fem.mesh().element_metrics(quad).jacobians(MemorySpace::DEVICE);
        host_device_dvec w = rule_weights(quad);
        detail::check_hip(cuddh_hip_stiffness_setup(n_elem, n_quad, w.device_read(), J, _G.device_write(), stream()),
                          "StiffnessMatrix setup");
        detail::check_hip(cuddh_hip_stream_sync(stream()), "StiffnessMatrix setup"); // w dies here
    }

    void StiffnessMatrix::action(double c, const double *x, double *y) const
    {
        if (const cuddh_helmholtz_plan *pl = plan.get(0, fem, n_quad, _P.host_read(), _D.host_read(), _G.device_read()))
        {
            detail::check_hip(cuddh_hip_operator_plan_apply(pl, c, 1, x, y, stream()), "StiffnessMatrix::action");
            return;
        }
        detail::check_hip(cuddh_hip_stiffness_apply(n_elem, n_quad, n_basis, _P.device_read(), _D.device_read(),
                                                    _G.device_read(), fem.global_indices(MemorySpace::DEVICE), c, x, y,
                                                    stream()),
                          "StiffnessMatrix::action");
    }

    void StiffnessMatrix::action(const double *x, double *y) const
    {
        if (const cuddh_helmholtz_plan *pl = plan.get(0, fem, n_quad, _P.host_read(), _D.host_read(), _G.device_read()))
        {
            detail::check_hip(cuddh_hip_operator_plan_apply(pl, 1.0, 0, x, y, stream()), "StiffnessMatrix::action");
            return;
        }
        zeros(ndof, y);
        action(1.0, x, y);
    }

    // ------------------------------------------------------------ MassMatrix

    MassMatrix::MassMatrix(const H1Space &fem_)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()),
          n_quad(fem_.basis().size() + fem_.mesh().max_element_order()), _P(n_quad * n_basis), _a(n_quad * n_quad * n_elem)
    {
        setup(nullptr);
    }

    MassMatrix::MassMatrix(const double *a, const H1Space &fem_)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()),
          n_quad(1 + 3 * fem_.basis().size() / 2 + fem_.mesh().max_element_order()), _P(n_quad * n_basis),
          _a(n_quad * n_quad * n_elem)
    {
        setup(a);
    }

    void MassMatrix::setup(const double *a)
    {
        QuadratureRule quad(n_quad, QuadratureRule::GaussLegendre);
        fem.basis().eval(n_quad, quad.x(), _P.host_write());
        const double *detJ = fem.mesh().element_metrics(quad).measures(MemorySpace::DEVICE);
        host_device_dvec w = rule_weights(quad);
        detail::check_hip(cuddh_hip_mass_setup(n_elem, n_quad, n_basis, a, detJ, w.device_read(),
                                               fem.global_indices(MemorySpace::DEVICE), _P.device_read(), _a.device_write(),
                                               stream()),
                          "MassMatrix setup");
        detail::check_hip(cuddh_hip_stream_sync(stream()), "MassMatrix setup");
    }

    void MassMatrix::action(double c, const double *x, double *y) const
    {
        if (const cuddh_helmholtz_plan *pl = plan.get(1, fem, n_quad, _P.host_read(), nullptr, _a.device_read()))
        {
            detail::check_hip(cuddh_hip_operator_plan_apply(pl, c, 1, x, y, stream()), "MassMatrix::action");
            return;
        }
        detail::check_hip(cuddh_hip_mass_apply(n_elem, n_quad, n_basis, fem.global_indices(MemorySpace::DEVICE),
                                               _P.device_read(), _a.device_read(), c, x, y, stream()),
                          "MassMatrix::action");
    }

    void MassMatrix::action(const double *x, double *y) const
    {
        if (const cuddh_helmholtz_plan *pl = plan.get(1, fem, n_quad, _P.host_read(), nullptr, _a.device_read()))
        {
            detail::check_hip(cuddh_hip_operator_plan_apply(pl, 1.0, 0, x, y, stream()), "MassMatrix::action");
            return;
        }
        zeros(ndof, y);
        action(1.0, x, y);
    }

    // ------------------------------------------------------------ DiagInvMassMatrix

    DiagInvMassMatrix::DiagInvMassMatrix(const H1Space &fem_) : fem(fem_), ndof(fem_.size()), _p(fem_.size()) { setup(nullptr); }

    DiagInvMassMatrix::DiagInvMassMatrix(const double *a, const H1Space &fem_) : fem(fem_), ndof(fem_.size()), _p(fem_.size())
    {
        setup(a);
    }

    void DiagInvMassMatrix::setup(const double *a)
    {
        const QuadratureRule &gll = fem.basis().quadrature();
        const double *detJ = fem.mesh().element_metrics(gll).measures(MemorySpace::DEVICE);
        host_device_dvec w = rule_weights(gll);
        detail::check_hip(cuddh_hip_diag_mass_setup(ndof, fem.mesh().n_elem(), fem.basis().size(), a, detJ, w.device_read(),
                                                    fem.global_indices(MemorySpace::DEVICE), _p.device_write(), stream()),
                          "DiagInvMassMatrix setup");
        detail::check_hip(cuddh_hip_stream_sync(stream()), "DiagInvMassMatrix setup");
    }

    void DiagInvMassMatrix::action(double c, const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_diag_scale_f64(ndof, 1, c, _p.device_read(), x, y, stream()), "DiagInvMassMatrix::action");
    }

    void DiagInvMassMatrix::action(const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_diag_scale_f64(ndof, 0, 1.0, _p.device_read(), x, y, stream()), "DiagInvMassMatrix::action");
    }

    // ------------------------------------------------------------ FaceMassMatrix

    FaceMassMatrix::FaceMassMatrix(const FaceSpace &fs_)
        : fs(fs_), ndof(fs_.size()), n_faces(fs_.n_faces()), n_basis(fs_.h1_space().basis().size()),
          n_quad(fs_.h1_space().mesh().max_element_order() + n_basis), _a(n_quad * n_faces), _P(n_quad * n_basis)
    {
        setup(nullptr);
    }

    FaceMassMatrix::FaceMassMatrix(const double *a, const FaceSpace &fs_)
        : fs(fs_), ndof(fs_.size()), n_faces(fs_.n_faces()), n_basis(fs_.h1_space().basis().size()),
          n_quad(fs_.h1_space().mesh().max_element_order() + 3 * n_basis / 2 + 1), _a(n_quad * n_faces), _P(n_quad * n_basis)
    {
        setup(a);
    }

    void FaceMassMatrix::setup(const double *a)
    {
        QuadratureRule quad(n_quad, QuadratureRule::GaussLegendre);
        fs.h1_space().basis().eval(n_quad, quad.x(), _P.host_write());
        const double *detJ = fs.metrics(quad).measures(MemorySpace::DEVICE);
        host_device_dvec w = rule_weights(quad);
        detail::check_hip(cuddh_hip_facemass_setup(n_faces, n_basis, n_quad, w.device_read(), _P.device_read(), detJ, a,
                                                   fs.subspace_indices(MemorySpace::DEVICE), _a.device_write(), stream()),
                          "FaceMassMatrix setup");
        detail::check_hip(cuddh_hip_stream_sync(stream()), "FaceMassMatrix setup");
    }

    void FaceMassMatrix::action(double c, const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_facemass_apply(n_faces, n_basis, n_quad, _P.device_read(), _a.device_read(),
                                                   fs.subspace_indices(MemorySpace::DEVICE), c, x, y, stream()),
                          "FaceMassMatrix::action");
    }

    void FaceMassMatrix::action(const double *x, double *y) const
    {
        zeros(ndof, y);
        action(1.0, x, y);
    }

    // ------------------------------------------------------------ DiagInvFaceMassMatrix

    DiagInvFaceMassMatrix::DiagInvFaceMassMatrix(const FaceSpace &fs) : ndof(fs.size()), inv_m(fs.size()) { setup(nullptr, fs); }

    DiagInvFaceMassMatrix::DiagInvFaceMassMatrix(const double *a, const FaceSpace &fs) : ndof(fs.size()), inv_m(fs.size())
    {
        setup(a, fs);
    }

    void DiagInvFaceMassMatrix::setup(const double *a, const FaceSpace &fs)
    {
        const QuadratureRule &gll = fs.h1_space().basis().quadrature();
        host_device_dvec w = rule_weights(gll);
        const double *detJ = fs.metrics(gll).measures(MemorySpace::DEVICE);
        // device_write() hands back a zero-filled fresh allocation, which the accumulation needs
        detail::check_hip(cuddh_hip_diag_facemass_setup(ndof, fs.n_faces(), gll.size(), w.device_read(), detJ, a,
                                                        fs.subspace_indices(MemorySpace::DEVICE), inv_m.device_write(),
                                                        stream()),
                          "DiagInvFaceMassMatrix setup");
        detail::check_hip(cuddh_hip_stream_sync(stream()), "DiagInvFaceMassMatrix setup");
    }

    void DiagInvFaceMassMatrix::action(double c, const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_diag_scale_f64(ndof, 1, c, inv_m.device_read(), x, y, stream()),
                          "DiagInvFaceMassMatrix::action");
    }

    void DiagInvFaceMassMatrix::action(const double *x, double *y) const
    {
        detail::check_hip(cuddh_hip_diag_scale_f64(ndof, 0, 1.0, inv_m.device_read(), x, y, stream()),
                          "DiagInvFaceMassMatrix::action");
    }
} // namespace cuddh
