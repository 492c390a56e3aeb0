// Constructors of the load-vector functionals (the kernels are header templates).
#include "cuddh/functionals.hpp"

namespace cuddh
{
    namespace
    {
        void store_weights(host_device_dvec &w, const QuadratureRule &quad)
        {
            double *h = w.host_write();
            for (int i = 0; i < quad.size(); ++i)
                h[i] = quad.w(i);
        }
    } // namespace

    LinearFunctional::LinearFunctional(const H1Space &fem_)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()),
          n_quad(fem_.basis().size()), collocated(true), metrics(fem_.mesh().element_metrics(fem_.basis().quadrature())),
          _w(n_quad)
    {
        store_weights(_w, fem.basis().quadrature());
    }

    LinearFunctional::LinearFunctional(const H1Space &fem_, const QuadratureRule &quad)
        : fem(fem_), ndof(fem_.size()), n_elem(fem_.mesh().n_elem()), n_basis(fem_.basis().size()), n_quad(quad.size()),
          collocated(false), metrics(fem_.mesh().element_metrics(quad)), _w(n_quad), _P(n_quad * n_basis)
    {
        store_weights(_w, quad);
        fem.basis().eval(n_quad, quad.x(), _P.host_write());
    }

    FaceLinearFunctional::FaceLinearFunctional(const FaceSpace &fs_)
        : fs(fs_), metrics(fs_.metrics(fs_.h1_space().basis().quadrature())), fdof(fs_.size()), n_faces(fs_.n_faces()),
          n_basis(fs_.h1_space().basis().size()), n_quad(n_basis), collocated(true), _w(n_quad)
    {
        store_weights(_w, fs.h1_space().basis().quadrature());
    }

    FaceLinearFunctional::FaceLinearFunctional(const FaceSpace &fs_, const QuadratureRule &quad)
        : fs(fs_), metrics(fs_.metrics(quad)), fdof(fs_.size()), n_faces(fs_.n_faces()),
          n_basis(fs_.h1_space().basis().size()), n_quad(quad.size()), collocated(false), _w(n_quad), _P(n_quad * n_basis)
    {
        store_weights(_w, quad);
        fs.h1_space().basis().eval(n_quad, quad.x(), _P.host_write());
    }
} // namespace cuddh
