// C handle layer (include/cuddh_capi.h).  Compiled as HIP: the built-in integrands
// are device functors fed to the LinearFunctional header templates.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

#include "cuddh.hpp"
#include "cuddh_capi.h"
#include "cuddh_hip.h"

using namespace cuddh;

namespace
{
    thread_local std::string g_error;

    template <typename F>
    int guarded(F &&f)
    {
        try
        {
            f();
            return 0;
        }
        catch (const std::exception &e)
        {
            g_error = e.what();
        }
        catch (...)
        {
            g_error = "unknown error";
        }
        return 1;
    }

    template <typename T, typename F>
    T *guarded_new(F &&f)
    {
        T *out = nullptr;
        guarded([&] { out = f(); });
        return out;
    }

    struct Integrand
    {
        int id;
        double param;

        __host__ __device__ double operator()(const double X[2]) const
        {
            const double x = X[0], y = X[1];
            switch (id)
            {
            case 0:
            {
                const double s = param * param;
                double r = (x + 0.5) * (x + 0.5) + y * y;
                double F = s / M_PI * exp(-s * r);
                r = (x - 0.5) * (x - 0.5) + (y + 0.5) * (y + 0.5);
                return F + s / M_PI * exp(-s * r);
            }
            case 1: return (x * x + y * y < 0.0625) ? 0.2 : 1.0;
            case 2: return 3.0 * x * x - 2.0 * x * y + y + 1.0;
            case 3:
            {
                const double x3 = x * x * x, x5 = x3 * x * x, y3 = y * y * y;
                return -6.0 * y * (x5 - 5.0 * x) - 20.0 * x3 * (y3 - 3.0 * y);
            }
            case 4:
            {
                const double x5 = x * x * x * x * x, y3 = y * y * y;
                return (x5 - 5.0 * x) * (y3 - 3.0 * y);
            }
            case 5: return param;
            case 6:
            {
                const double a = (x * x + y * y < 0.0625) ? 0.2 : 1.0;
                return a * a;
            }
            default: return 0.0;
            }
        }
    };

    struct OpHandle
    {
        std::unique_ptr<Operator> op;
        HelmholtzOperator *helm = nullptr; // non-owning view when op is a HelmholtzOperator
    };

    struct DdhHandle
    {
        std::unique_ptr<DDH> f32;
        std::unique_ptr<DDH64> f64;
        bool is64() const { return static_cast<bool>(f64); }
    };

    void fill_result(const solver_out &o, cuddh_solver_result *out, double *h_res, double *h_time)
    {
        if (out)
        {
            out->success = o.success ? 1 : 0;
            out->num_iter = o.num_iter;
            out->num_matvec = o.num_matvec;
            out->n_res = static_cast<int>(o.res_norm.size());
        }
        for (std::size_t i = 0; i < o.res_norm.size(); ++i)
        {
            if (h_res)
                h_res[i] = o.res_norm[i];
            if (h_time)
                h_time[i] = o.time[i];
        }
    }

    class CallbackOp64 : public Operator
    {
    public:
        CallbackOp64(cuddh_action_cb cb_, void *ctx_) : cb(cb_), ctx(ctx_) {}
        void action(const double *x, double *y) const override { cb(ctx, x, y); }
        void action(double, const double *, double *) const override { cuddh_error("callback operator: action(c,x,y) unsupported"); }

    private:
        cuddh_action_cb cb;
        void *ctx;
    };

    class CallbackOp32 : public SinglePrecisionOperator
    {
    public:
        CallbackOp32(cuddh_action_cb cb_, void *ctx_) : cb(cb_), ctx(ctx_) {}
        void action(const float *x, float *y) const override { cb(ctx, x, y); }

    private:
        cuddh_action_cb cb;
        void *ctx;
    };

    template <typename T>
    long long copy_table(const HostDeviceArray<T> &t, void *h_out, int count_only)
    {
        if (!count_only && t.size() > 0)
            std::memcpy(h_out, t.host_read(), sizeof(T) * static_cast<std::size_t>(t.size()));
        return t.size();
    }

    template <typename Real>
    long long ddh_table(const detail::DDHCore<Real> &c, const std::string &name, void *h_out, int count_only)
    {
        if (name == "B") return copy_table(c.table_B(), h_out, count_only);
        if (name == "gI") return copy_table(c.table_gI(), h_out, count_only);
        if (name == "sI") return copy_table(c.table_sI(), h_out, count_only);
        if (name == "D") return copy_table(c.table_D(), h_out, count_only);
        if (name == "G") return copy_table(c.table_G(), h_out, count_only);
        if (name == "m") return copy_table(c.table_m(), h_out, count_only);
        if (name == "gmi") return copy_table(c.table_gmi(), h_out, count_only);
        if (name == "a") return copy_table(c.table_a(), h_out, count_only);
        if (name == "H") return copy_table(c.table_H(), h_out, count_only);
        if (name == "filter") return copy_table(c.table_filter(), h_out, count_only);
        if (name == "cs") return copy_table(c.table_cs(), h_out, count_only);
        if (name == "sn") return copy_table(c.table_sn(), h_out, count_only);
        throw std::runtime_error("unknown DDH table: " + name);
    }
} // namespace

extern "C"
{
    const char *cuddh_last_error(void) { return g_error.c_str(); }

    void cuddh_set_stream(void *s) { set_stream(static_cast<hipStream_t>(s)); }
    void *cuddh_get_stream(void) { return stream(); }

    // ------------------------------------------------------------ quadrature / basis
    int cuddh_quadrature(int n, int type, double *h_x, double *h_w)
    {
        return guarded([&]
        {
            QuadratureRule q(n, type == 0 ? QuadratureRule::GaussLegendre : QuadratureRule::GaussLobatto);
            for (int i = 0; i < n; ++i)
            {
                h_x[i] = q.x(i);
                h_w[i] = q.w(i);
            }
        });
    }

    void *cuddh_basis_create(int n) { return guarded_new<Basis>([&] { return new Basis(n); }); }
    void cuddh_basis_destroy(void *b) { delete static_cast<Basis *>(b); }
    int cuddh_basis_eval(void *b, int m, const double *h_x, double *h_P)
    {
        return guarded([&] { static_cast<Basis *>(b)->eval(m, h_x, h_P); });
    }
    int cuddh_basis_deriv(void *b, int m, const double *h_x, double *h_D)
    {
        return guarded([&] { static_cast<Basis *>(b)->deriv(m, h_x, h_D); });
    }

    // ------------------------------------------------------------ mesh
    void *cuddh_mesh_uniform_rect(int nx, double ax, double bx, int ny, double ay, double by)
    {
        return guarded_new<Mesh2D>([&] { return new Mesh2D(Mesh2D::uniform_rect(nx, ax, bx, ny, ay, by)); });
    }
    void *cuddh_mesh_from_vertices(int n_pts, const double *h_xy, int n_elem, const int *h_elems)
    {
        return guarded_new<Mesh2D>([&] { return new Mesh2D(Mesh2D::from_vertices(n_pts, h_xy, n_elem, h_elems)); });
    }
    void *cuddh_mesh_load(const char *dir)
    {
        return guarded_new<Mesh2D>([&] { return new Mesh2D(load_mesh(dir)); });
    }
    void *cuddh_mesh_refined(void *m, int times)
    {
        return guarded_new<Mesh2D>([&]
        {
            const QuadMeshData fine = refine_quads(mesh_data(*static_cast<Mesh2D *>(m)), times);
            return new Mesh2D(Mesh2D::from_vertices(fine.n_pts(), fine.xy.data(), fine.n_elem(), fine.elems.data()));
        });
    }
    int cuddh_mesh_partition(void *m, int n_parts, int *h_labels)
    {
        return guarded([&]
        {
            const std::vector<int> labels = partition_elements(*static_cast<Mesh2D *>(m), n_parts);
            std::copy(labels.begin(), labels.end(), h_labels);
        });
    }
    void cuddh_mesh_destroy(void *m) { delete static_cast<Mesh2D *>(m); }
    int cuddh_mesh_n_elem(void *m) { return static_cast<Mesh2D *>(m)->n_elem(); }
    int cuddh_mesh_n_edges(void *m) { return static_cast<Mesh2D *>(m)->n_edges(); }
    int cuddh_mesh_n_nodes(void *m) { return static_cast<Mesh2D *>(m)->n_nodes(); }
    int cuddh_mesh_n_boundary_edges(void *m) { return static_cast<Mesh2D *>(m)->n_edges(FaceType::BOUNDARY); }
    int cuddh_mesh_boundary_edges(void *m, int *h_out)
    {
        return guarded([&]
        {
            ivec b = static_cast<Mesh2D *>(m)->boundary_edges();
            for (int i = 0; i < b.size(); ++i)
                h_out[i] = b[i];
        });
    }
    int cuddh_mesh_edges(void *m, int *h_out)
    {
        return guarded([&]
        {
            const Mesh2D *mesh = static_cast<Mesh2D *>(m);
            for (int e = 0; e < mesh->n_edges(); ++e)
            {
                const Edge *ed = mesh->edge(e);
                int *o = h_out + 8 * e;
                o[0] = ed->type == FaceType::BOUNDARY ? 1 : 0;
                o[1] = ed->nodes[0];
                o[2] = ed->nodes[1];
                o[3] = ed->elements[0];
                o[4] = ed->elements[1];
                o[5] = ed->sides[0];
                o[6] = ed->sides[1];
                o[7] = ed->delta;
            }
        });
    }
    double cuddh_mesh_min_h(void *m) { return static_cast<Mesh2D *>(m)->min_h(); }
    int cuddh_mesh_vertices(void *m, double *h_xy)
    {
        return guarded([&]
        {
            const Mesh2D *mesh = static_cast<Mesh2D *>(m);
            for (int i = 0; i < mesh->n_nodes(); ++i)
            {
                h_xy[2 * i] = mesh->node(i).x[0];
                h_xy[2 * i + 1] = mesh->node(i).x[1];
            }
        });
    }
    int cuddh_mesh_elements(void *m, int *h_out)
    {
        return guarded([&]
        {
            const Mesh2D *mesh = static_cast<Mesh2D *>(m);
            for (int e = 0; e < mesh->n_elem(); ++e)
                for (int c = 0; c < 4; ++c)
                    h_out[4 * e + c] = mesh->element(e)->nodes[c];
        });
    }

    // ------------------------------------------------------------ spaces
    void *cuddh_h1space_create(void *mesh, void *basis)
    {
        return guarded_new<H1Space>([&] { return new H1Space(*static_cast<Mesh2D *>(mesh), *static_cast<Basis *>(basis)); });
    }
    void cuddh_h1space_destroy(void *f) { delete static_cast<H1Space *>(f); }
    int cuddh_h1space_size(void *f) { return static_cast<H1Space *>(f)->size(); }
    int cuddh_h1space_global_indices(void *f, int *h_I)
    {
        return guarded([&]
        {
            auto I = static_cast<H1Space *>(f)->global_indices(MemorySpace::HOST);
            std::memcpy(h_I, I.data(), sizeof(int) * static_cast<std::size_t>(I.size()));
        });
    }
    int cuddh_h1space_coordinates(void *f, double *h_xy)
    {
        return guarded([&]
        {
            auto X = static_cast<H1Space *>(f)->physical_coordinates(MemorySpace::HOST);
            std::memcpy(h_xy, X.data(), sizeof(double) * static_cast<std::size_t>(X.size()));
        });
    }
    const int *cuddh_h1space_global_indices_device(void *f)
    {
        const int *p = nullptr;
        guarded([&] { p = static_cast<H1Space *>(f)->global_indices(MemorySpace::DEVICE).data(); });
        return p;
    }
    const double *cuddh_h1space_coordinates_device(void *f)
    {
        const double *p = nullptr;
        guarded([&] { p = static_cast<H1Space *>(f)->physical_coordinates(MemorySpace::DEVICE).data(); });
        return p;
    }

    void *cuddh_facespace_create(void *fem, int n_faces, const int *h_faces)
    {
        return guarded_new<FaceSpace>([&] { return new FaceSpace(*static_cast<H1Space *>(fem), n_faces, h_faces); });
    }
    void cuddh_facespace_destroy(void *fs) { delete static_cast<FaceSpace *>(fs); }
    int cuddh_facespace_size(void *fs) { return static_cast<FaceSpace *>(fs)->size(); }
    int cuddh_facespace_subspace_indices(void *fs, int *h_I)
    {
        return guarded([&]
        {
            auto I = static_cast<FaceSpace *>(fs)->subspace_indices(MemorySpace::HOST);
            std::memcpy(h_I, I.data(), sizeof(int) * static_cast<std::size_t>(I.size()));
        });
    }
    int cuddh_facespace_global_indices(void *fs, int *h_proj)
    {
        return guarded([&]
        {
            auto I = static_cast<FaceSpace *>(fs)->global_indices(MemorySpace::HOST);
            std::memcpy(h_proj, I.data(), sizeof(int) * static_cast<std::size_t>(I.size()));
        });
    }
    int cuddh_facespace_restrict(void *fs, const double *x, double *y) { return guarded([&] { static_cast<FaceSpace *>(fs)->restrict(x, y); }); }
    int cuddh_facespace_prolong(void *fs, const double *x, double *y) { return guarded([&] { static_cast<FaceSpace *>(fs)->prolong(x, y); }); }
    int cuddh_facespace_orth(void *fs, double *x) { return guarded([&] { static_cast<FaceSpace *>(fs)->orth(x); }); }

    // ------------------------------------------------------------ EnsembleSpace
    void *cuddh_ensemble_create(void *fem, int n_spaces, const int *h_labels)
    {
        return guarded_new<EnsembleSpace>([&] { return new EnsembleSpace(*static_cast<H1Space *>(fem), n_spaces, h_labels); });
    }
    void cuddh_ensemble_destroy(void *e) { delete static_cast<EnsembleSpace *>(e); }
    int cuddh_ensemble_dims(void *e, int *d)
    {
        return guarded([&]
        {
            const EnsembleSpace *E = static_cast<EnsembleSpace *>(e);
            d[0] = E->size();
            d[1] = E->elements(MemorySpace::HOST).shape(0);
            d[2] = E->faces(MemorySpace::HOST).shape(0);
            d[3] = E->global_indices(MemorySpace::HOST).shape(0);
            d[4] = E->face_proj(MemorySpace::HOST).shape(0);
            d[5] = E->connectivity_map(MemorySpace::HOST).shape(1);
        });
    }
    int cuddh_ensemble_array(void *e, const char *name_, int *h_out)
    {
        return guarded([&]
        {
            const EnsembleSpace *E = static_cast<EnsembleSpace *>(e);
            const std::string name(name_);
            const int *src = nullptr;
            int n = 0;
            auto take = [&](auto w)
            {
                src = w.data();
                n = w.size();
            };
            if (name == "gI") take(E->global_indices(MemorySpace::HOST));
            else if (name == "sizes") take(E->sizes(MemorySpace::HOST));
            else if (name == "elements") take(E->elements(MemorySpace::HOST));
            else if (name == "n_elems") take(E->n_elems(MemorySpace::HOST));
            else if (name == "faces") take(E->faces(MemorySpace::HOST));
            else if (name == "n_faces") take(E->n_faces(MemorySpace::HOST));
            else if (name == "sI") take(E->subspace_indices(MemorySpace::HOST));
            else if (name == "fI") take(E->face_indices(MemorySpace::HOST));
            else if (name == "pI") take(E->face_proj(MemorySpace::HOST));
            else if (name == "fsizes") take(E->fsizes(MemorySpace::HOST));
            else if (name == "cmap") take(E->connectivity_map(MemorySpace::HOST));
            else throw std::runtime_error("unknown EnsembleSpace array: " + name);
            if (n > 0)
                std::memcpy(h_out, src, sizeof(int) * static_cast<std::size_t>(n));
        });
    }

    // ------------------------------------------------------------ operators
    void *cuddh_stiffness_create(void *fem, int nq)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            const H1Space &f = *static_cast<H1Space *>(fem);
            if (nq > 0)
                h->op.reset(new StiffnessMatrix(f, QuadratureRule(nq, QuadratureRule::GaussLegendre)));
            else
                h->op.reset(new StiffnessMatrix(f));
            return h;
        });
    }
    void *cuddh_mass_create(void *fem, const double *coef)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            const H1Space &f = *static_cast<H1Space *>(fem);
            h->op.reset(coef ? new MassMatrix(coef, f) : new MassMatrix(f));
            return h;
        });
    }
    void *cuddh_diaginv_mass_create(void *fem, const double *coef)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            const H1Space &f = *static_cast<H1Space *>(fem);
            h->op.reset(coef ? new DiagInvMassMatrix(coef, f) : new DiagInvMassMatrix(f));
            return h;
        });
    }
    void *cuddh_facemass_create(void *fs, const double *coef)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            const FaceSpace &f = *static_cast<FaceSpace *>(fs);
            h->op.reset(coef ? new FaceMassMatrix(coef, f) : new FaceMassMatrix(f));
            return h;
        });
    }
    void *cuddh_diaginv_facemass_create(void *fs, const double *coef)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            const FaceSpace &f = *static_cast<FaceSpace *>(fs);
            h->op.reset(coef ? new DiagInvFaceMassMatrix(coef, f) : new DiagInvFaceMassMatrix(f));
            return h;
        });
    }
    void *cuddh_helmholtz_create(double omega, const double *a2x, const double *ax, void *fem, void *fs)
    {
        return guarded_new<OpHandle>([&]
        {
            auto h = new OpHandle;
            auto *H = new HelmholtzOperator(omega, a2x, ax, *static_cast<H1Space *>(fem), *static_cast<FaceSpace *>(fs));
            h->op.reset(H);
            h->helm = H;
            return h;
        });
    }
    void cuddh_operator_destroy(void *op) { delete static_cast<OpHandle *>(op); }
    int cuddh_operator_apply(void *op, const double *x, double *y) { return guarded([&] { static_cast<OpHandle *>(op)->op->action(x, y); }); }
    int cuddh_operator_apply_add(void *op, double c, const double *x, double *y)
    {
        return guarded([&] { static_cast<OpHandle *>(op)->op->action(c, x, y); });
    }
    int cuddh_helmholtz_apply_unfused(void *op, const double *x, double *y)
    {
        return guarded([&]
        {
            auto *h = static_cast<OpHandle *>(op);
            if (!h->helm)
                throw std::runtime_error("not a Helmholtz operator");
            h->helm->action_unfused(x, y);
        });
    }
    int cuddh_helmholtz_is_fused(void *op)
    {
        auto *h = static_cast<OpHandle *>(op);
        return h->helm && h->helm->fused() ? 1 : 0;
    }
    int cuddh_operator_kernel_name(void *op, char *buf, int cap)
    {
        return guarded([&]
        {
            auto *h = static_cast<OpHandle *>(op);
            std::string name = "generic";
            if (h->helm)
                name = h->helm->kernel_name();
            else if (auto *s = dynamic_cast<StiffnessMatrix *>(h->op.get()))
                name = s->kernel_name();
            else if (auto *m = dynamic_cast<MassMatrix *>(h->op.get()))
                name = m->kernel_name();
            std::snprintf(buf, cap, "%s", name.c_str());
        });
    }
    int cuddh_helmholtz_read_stamps(void *op, unsigned long long *h_out, int n_patches)
    {
        auto *h = static_cast<OpHandle *>(op);
        if (!h->helm)
            return -1;
        return cuddh_hip_helmholtz_plan_read_stamps(h->helm->kernel_plan(), h_out, n_patches);
    }
    size_t cuddh_helmholtz_bytes(void *op, int actual)
    {
        auto *h = static_cast<OpHandle *>(op);
        if (!h->helm)
            return 0;
        return actual == 3 ? h->helm->bytes_native() : (actual == 2 ? h->helm->bytes_affine() : h->helm->bytes_per_apply(actual != 0));
    }
    int cuddh_helmholtz_has_native(void *op)
    {
        auto *h = static_cast<OpHandle *>(op);
        return h->helm && h->helm->has_native() ? 1 : 0;
    }
    static HelmholtzOperator &helm_of(void *op)
    {
        auto *h = static_cast<OpHandle *>(op);
        if (!h->helm)
            throw std::runtime_error("not a Helmholtz operator");
        return *h->helm;
    }
    int cuddh_helmholtz_to_native(void *op, const double *x, double *z) { return guarded([&] { helm_of(op).to_native(x, z); }); }
    int cuddh_helmholtz_from_native(void *op, const double *z, double *y) { return guarded([&] { helm_of(op).from_native(z, y); }); }
    int cuddh_helmholtz_apply_native(void *op, const double *z_in, double *z_out) { return guarded([&] { helm_of(op).action_native(z_in, z_out); }); }

    // ------------------------------------------------------------ functionals
    int cuddh_linear_functional(void *fem, int nq, int integrand, double param, double c, int accumulate, double *F)
    {
        return guarded([&]
        {
            const H1Space &f = *static_cast<H1Space *>(fem);
            const Integrand g{integrand, param};
            if (nq > 0)
            {
                LinearFunctional l(f, QuadratureRule(nq, QuadratureRule::GaussLegendre));
                if (!accumulate)
                    zeros(f.size(), F);
                l.action(c, g, F);
                detail::check_hip(cuddh_hip_stream_sync(stream()), "linear functional"); // scratch dies with l
            }
            else
            {
                LinearFunctional l(f);
                if (!accumulate)
                    zeros(f.size(), F);
                l.action(c, g, F);
                detail::check_hip(cuddh_hip_stream_sync(stream()), "linear functional");
            }
        });
    }
    int cuddh_face_linear_functional(void *fs_, int nq, int integrand, double param, double c, int accumulate, double *F)
    {
        return guarded([&]
        {
            const FaceSpace &fs = *static_cast<FaceSpace *>(fs_);
            const Integrand g{integrand, param};
            if (nq > 0)
            {
                FaceLinearFunctional l(fs, QuadratureRule(nq, QuadratureRule::GaussLegendre));
                if (!accumulate)
                    zeros(fs.size(), F);
                l.action(c, g, F);
                detail::check_hip(cuddh_hip_stream_sync(stream()), "face linear functional");
            }
            else
            {
                FaceLinearFunctional l(fs);
                if (!accumulate)
                    zeros(fs.size(), F);
                l.action(c, g, F);
                detail::check_hip(cuddh_hip_stream_sync(stream()), "face linear functional");
            }
        });
    }
    int cuddh_nodal_values(void *fem, int integrand, double param, double *out)
    {
        return guarded([&]
        {
            const H1Space &f = *static_cast<H1Space *>(fem);
            const double *X = f.physical_coordinates(MemorySpace::DEVICE);
            const Integrand g{integrand, param};
            forall(f.size(), [=] __device__(int i) -> void
            {
                const double xy[2] = {X[2 * i], X[2 * i + 1]};
                out[i] = g(xy);
            });
        });
    }

    // ------------------------------------------------------------ DDH
    void *cuddh_ddh_create(double omega, const double *h_a, void *fem, int nx, int ny, int f64, int kernel)
    {
        return guarded_new<DdhHandle>([&]
        {
            auto h = new DdhHandle;
            const H1Space &f = *static_cast<H1Space *>(fem);
            if (f64)
                h->f64.reset(new DDH64(omega, h_a, f, nx, ny, kernel));
            else
                h->f32.reset(new DDH(omega, h_a, f, nx, ny, kernel));
            return h;
        });
    }
    void cuddh_ddh_destroy(void *d) { delete static_cast<DdhHandle *>(d); }
    int cuddh_ddh_size(void *d)
    {
        auto *h = static_cast<DdhHandle *>(d);
        return h->is64() ? h->f64->size() : h->f32->size();
    }
    int cuddh_ddh_info(void *d, int *info, double *h_dt)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            auto fill = [&](const auto &c)
            {
                info[0] = c.num_domains();
                info[1] = c.num_steps();
                info[2] = c.n_traces() / 2;
                info[3] = c.max_dof();
                info[4] = c.max_fdof();
                info[5] = c.elems_per_side();
                info[6] = cuddh_hip_device_count() > 0 ? c.kernel_kind() : -1;
                *h_dt = c.time_step();
            };
            if (h->is64())
                fill(h->f64->internals());
            else
                fill(h->f32->internals());
            info[7] = h->is64() ? 1 : 0;
        });
    }
    int cuddh_ddh_solve_multi_gpu(int nx, int nb, double omega, const double *h_a, const double *h_f, double *h_u, int world, int m,
                                  int maxit, double tol, int force_rccl, cuddh_multi_gpu_result *out, double *h_res)
    {
        return guarded([&]
        {
            const multi_gpu_result r = ddh_solve_multi_gpu(nx, nb, omega, h_a, h_f, h_u, world, m, maxit, static_cast<float>(tol), force_rccl & 3,
                                                           (force_rccl & 4) != 0, (force_rccl >> 8) & 0xFF, (force_rccl >> 16) & 0xFF);
            out->success = r.gmres.success ? 1 : 0;
            out->num_iter = r.gmres.num_iter;
            out->num_matvec = r.gmres.num_matvec;
            out->n_res = static_cast<int>(r.gmres.res_norm.size());
            out->world = r.world;
            out->used_rccl = r.used_rccl ? 1 : 0;
            out->t_setup = r.t_setup;
            out->t_rhs = r.t_rhs;
            out->t_gmres = r.t_gmres;
            out->t_postprocess = r.t_postprocess;
            out->bytes_sent_per_action_rank0 = r.bytes_sent_per_action_rank0;
            for (int i = 0; i < out->n_res && i < maxit + 2; ++i)
                h_res[i] = r.gmres.res_norm[i];
        });
    }
    int cuddh_helmholtz_multi_gpu(void *mesh, int nb, double omega, const double *h_a2x, const double *h_ax, const double *h_x, double *h_y,
                                  int world, int transport, int reps, int m, int maxit, double tol, cuddh_helmholtz_multi_gpu_result *out, double *h_res)
    {
        return guarded([&]
        {
            const QuadMeshData g = mesh_data(*static_cast<Mesh2D *>(mesh));
            const helmholtz_multi_gpu_result r = helmholtz_multi_gpu(g.n_pts(), g.xy.data(), g.n_elem(), g.elems.data(), nb, omega, h_a2x, h_ax, h_x, h_y,
                                                                     world, transport, reps, m, maxit, tol);
            out->success = r.gmres.success ? 1 : 0;
            out->num_iter = r.gmres.num_iter;
            out->num_matvec = r.gmres.num_matvec;
            out->n_res = static_cast<int>(r.gmres.res_norm.size());
            out->world = r.world;
            out->used_rccl = r.used_rccl ? 1 : 0;
            out->t_setup = r.t_setup;
            out->t_apply = r.t_apply;
            out->t_gmres = r.t_gmres;
            out->n_loc_max = r.n_loc_max;
            out->n_halo_max = r.n_halo_max;
            out->halo_bytes_per_apply_max = r.halo_bytes_per_apply_max;
            for (int i = 0; h_res && i < out->n_res && i < maxit + 2; ++i)
                h_res[i] = r.gmres.res_norm[i];
        });
    }
    int cuddh_helmholtz_partition_query(void *mesh, void *fem, void *fs, int rank, int world, int which, int peer, int *h_out)
    {
        int count = -1;
        const int err = guarded([&]
        {
            const H1Space &f = *static_cast<H1Space *>(fem);
            const HelmholtzPartition p = HelmholtzPartition::build(*static_cast<Mesh2D *>(mesh), f.basis(), f, *static_cast<FaceSpace *>(fs), rank, world);
            static const std::vector<int> none;
            const std::vector<int> *v = &none;
            if (which == 0)
                v = &p.my_elems;
            else if (which == 1)
                v = &p.l2g;
            else if (which == 2)
                v = &p.owned;
            else if (which == 3)
                v = &p.halo;
            else if (which == 4 || which == 5)
            {
                const auto &m = which == 4 ? p.own_to : p.halo_from;
                const auto it = m.find(peer);
                if (it != m.end())
                    v = &it->second;
            }
            else if (which == 6)
                v = &p.face_l2g;
            else if (which == 7)
                v = &p.faces;
            count = static_cast<int>(v->size());
            if (h_out)
                std::copy(v->begin(), v->end(), h_out);
        });
        return err ? -1 : count;
    }
    int cuddh_trace_exchange_query(const int *h_B, int n_domains, int mx_fdof, int n_lambda, int rank, int world, int which, int peer,
                                   int *h_out)
    {
        int count = -1;
        const int err = guarded([&]
        {
            const TraceExchangePlan p = TraceExchangePlan::build(h_B, n_domains, mx_fdof, n_lambda, rank, world);
            const std::vector<int> *v = nullptr;
            static const std::vector<int> none;
            if (which == 0)
                v = &p.owned;
            else if (which == 3)
                v = &p.boundary;
            else if (which == 4)
                v = &p.interior;
            else
            {
                const auto &m = which == 1 ? p.send : p.recv;
                const auto it = m.find(peer);
                v = it == m.end() ? &none : &it->second;
            }
            count = static_cast<int>(v->size());
            if (h_out)
                std::copy(v->begin(), v->end(), h_out);
        });
        return err ? -1 : count;
    }
    int cuddh_ddh_set_wh_iters(void *d, int n)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->internals().set_waveholtz_iterations(n);
            else
                h->f32->internals().set_waveholtz_iterations(n);
        });
    }
    int cuddh_ddh_set_wave_priority(void *d, int high)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->internals().set_wave_priority(high != 0);
            else
                h->f32->internals().set_wave_priority(high != 0);
        });
    }
    int cuddh_ddh_rhs(void *d, const double *f, void *b)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->rhs(f, static_cast<double *>(b));
            else
                h->f32->rhs(f, static_cast<float *>(b));
        });
    }
    int cuddh_ddh_postprocess(void *d, const void *lambda, const double *f, double *u)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->postprocess(static_cast<const double *>(lambda), f, u);
            else
                h->f32->postprocess(static_cast<const float *>(lambda), f, u);
        });
    }
    int cuddh_ddh_action(void *d, const void *x, void *y)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->action(static_cast<const double *>(x), static_cast<double *>(y));
            else
                h->f32->action(static_cast<const float *>(x), static_cast<float *>(y));
        });
    }
    int cuddh_ddh_local_traces(void *d, int d0, int d1, const double *f, const void *lambda, void *update)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->local_traces(d0, d1, f, static_cast<const double *>(lambda), static_cast<double *>(update));
            else
                h->f32->local_traces(d0, d1, f, static_cast<const float *>(lambda), static_cast<float *>(update));
        });
    }
    int cuddh_ddh_local_traces_listed(void *d, const int *d_domains, int n, const double *f, const void *lambda, void *update)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->internals().solve_listed(d_domains, n, f, static_cast<const double *>(lambda), static_cast<double *>(update));
            else
                h->f32->internals().solve_listed(d_domains, n, f, static_cast<const float *>(lambda), static_cast<float *>(update));
        });
    }
    int cuddh_ddh_local_solution_listed(void *d, const int *d_domains, int n, const void *lambda, const double *f, double *u, int zero_u)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->internals().solve_listed(d_domains, n, f, u, zero_u != 0, static_cast<const double *>(lambda), nullptr);
            else
                h->f32->internals().solve_listed(d_domains, n, f, u, zero_u != 0, static_cast<const float *>(lambda), nullptr);
        });
    }
    int cuddh_ddh_local_solution(void *d, int d0, int d1, const void *lambda, const double *f, double *u, int zero_u)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            if (h->is64())
                h->f64->local_solution(d0, d1, static_cast<const double *>(lambda), f, u, zero_u != 0);
            else
                h->f32->local_solution(d0, d1, static_cast<const float *>(lambda), f, u, zero_u != 0);
        });
    }
    long long cuddh_ddh_table(void *d, const char *name, void *h_out, int count_only)
    {
        long long n = -1;
        guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(d);
            n = h->is64() ? ddh_table(h->f64->internals(), name, h_out, count_only) : ddh_table(h->f32->internals(), name, h_out, count_only);
        });
        return n;
    }

    // ------------------------------------------------------------ GMRES
    int cuddh_gmres_f64(int n, double *x, void *op, const double *b, void *precond, int m, int maxit, double tol, int verbose,
                        double max_seconds, cuddh_solver_result *out, double *h_res, double *h_time)
    {
        return guarded([&]
        {
            const Operator *A = static_cast<OpHandle *>(op)->op.get();
            solver_out o = precond ? gmres(n, x, A, b, static_cast<OpHandle *>(precond)->op.get(), m, maxit, tol, verbose, max_seconds)
                                   : gmres(n, x, A, b, m, maxit, tol, verbose, max_seconds);
            fill_result(o, out, h_res, h_time);
        });
    }

    int cuddh_gmres_helmholtz(void *op, double *x, const double *b, int m, int maxit, double tol, int verbose, double max_seconds,
                              cuddh_solver_result *out, double *h_res, double *h_time)
    {
        return guarded([&]
        {
            solver_out o = helm_of(op).gmres(x, b, m, maxit, tol, verbose, max_seconds);
            fill_result(o, out, h_res, h_time);
        });
    }

    int cuddh_gmres_ddh(int n, void *x, void *ddh, const void *b, int m, int maxit, double tol, int verbose, double max_seconds,
                        cuddh_solver_result *out, double *h_res, double *h_time)
    {
        return guarded([&]
        {
            auto *h = static_cast<DdhHandle *>(ddh);
            solver_out o;
            if (h->is64())
                o = gmres(n, static_cast<double *>(x), h->f64.get(), static_cast<const double *>(b), m, maxit, tol, verbose, max_seconds);
            else
                o = gmres(n, static_cast<float *>(x), h->f32.get(), static_cast<const float *>(b), m, maxit, static_cast<float>(tol),
                          verbose, max_seconds);
            fill_result(o, out, h_res, h_time);
        });
    }

    int cuddh_gmres_callback(int n, void *x, cuddh_action_cb cb, void *ctx, const void *b, int is_f64, int m, int maxit, double tol,
                             int verbose, double max_seconds, cuddh_solver_result *out, double *h_res, double *h_time)
    {
        return guarded([&]
        {
            solver_out o;
            if (is_f64)
            {
                CallbackOp64 A(cb, ctx);
                o = gmres(n, static_cast<double *>(x), &A, static_cast<const double *>(b), m, maxit, tol, verbose, max_seconds);
            }
            else
            {
                CallbackOp32 A(cb, ctx);
                o = gmres(n, static_cast<float *>(x), &A, static_cast<const float *>(b), m, maxit, static_cast<float>(tol), verbose,
                          max_seconds);
            }
            fill_result(o, out, h_res, h_time);
        });
    }

    int cuddh_gmres_callback_sharded(int n, void *x, cuddh_action_cb cb, void *ctx, cuddh_reduce_cb reduce, void *reduce_ctx,
                                     const void *b, int is_f64, int m, int maxit, double tol, int verbose, double max_seconds,
                                     cuddh_solver_result *out, double *h_res, double *h_time)
    {
        return guarded([&]
        {
            solver_out o;
            const ScalarReduce red{reduce, reduce_ctx};
            if (is_f64)
            {
                CallbackOp64 A(cb, ctx);
                o = gmres(n, static_cast<double *>(x), &A, static_cast<const double *>(b), m, maxit, tol, verbose, max_seconds, red);
            }
            else
            {
                CallbackOp32 A(cb, ctx);
                o = gmres(n, static_cast<float *>(x), &A, static_cast<const float *>(b), m, maxit, static_cast<float>(tol), verbose,
                          max_seconds, red);
            }
            fill_result(o, out, h_res, h_time);
        });
    }
}
