// Continuous-Galerkin H1 space on a quad mesh and its restriction to a set of faces.
// Contract: reference include/H1Space.hpp:22-146.  The global numbering rule is
// the reference's (source/H1Space.cpp:11-127): walk the element-local flat index
// v = i + nb*(j + nb*el) upwards and give every node not aliased to an earlier
// owner the next number; interior-edge interior nodes of elements[1] alias those
// of elements[0] (reversed when delta < 0); every copy of a mesh vertex aliases the
// corner of the first element listed for that vertex.
#ifndef CUDDH_AMD_SPACES_HPP
#define CUDDH_AMD_SPACES_HPP

#include <memory>
#include <string>
#include <unordered_map>

#include "basis.hpp"
#include "launch.hpp"
#include "memory.hpp"
#include "mesh.hpp"
#include "operator.hpp"
#include "tensor.hpp"

#include <mutex>

namespace cuddh
{
    class H1Space
    {
    public:
        H1Space(const Mesh2D &mesh, const Basis &basis);

        int size() const { return ndof; }

        /// shape (n_basis, n_basis, n_elem): element node -> global dof
        const_icube_wrapper global_indices(MemorySpace m) const { return reshape(_I.read(m), n_basis, n_basis, n_elem); }

        const Mesh2D &mesh() const { return _mesh; }
        const Basis &basis() const { return _basis; }

        /// shape (2, ndof): collocation point of every dof.  Evaluated at the first request (the reference fills it in the
        /// constructor, source/H1Space.cpp:108-126; the solvers never read it, the drivers do once, for their output).
        const_dmat_wrapper physical_coordinates(MemorySpace m) const
        {
            std::call_once(_xy_once, [this] { build_collocation_points(); });
            return reshape(_xy.read(m), 2, ndof);
        }

    private:
        void build_collocation_points() const;

        const int n_elem;
        const int n_basis;
        const Mesh2D &_mesh;
        const Basis &_basis;
        int ndof;

        host_device_ivec _I;
        mutable host_device_dvec _xy;
        mutable std::once_flag _xy_once;
    };

    class FaceSpace
    {
    public:
        FaceSpace(const H1Space &fem, int n_faces, const int *faces);

        int size() const { return ndof; }
        int n_faces() const { return _n_faces; }

        const_ivec_wrapper faces(MemorySpace m) const { return reshape(_faces.read(m), _n_faces); }
        /// shape (n_basis, n_faces): face node -> FaceSpace dof
        const_imat_wrapper subspace_indices(MemorySpace m) const { return reshape(_I.read(m), n_basis, _n_faces); }
        /// FaceSpace dof -> H1Space dof
        const_ivec_wrapper global_indices(MemorySpace m) const { return reshape(_proj.read(m), ndof); }

        /// y[i] = x[proj(i)]          (x: H1 vector, y: face vector; DEVICE)
        void restrict(const double *x, double *y) const;
        /// y[proj(i)] += x[i]         (x: face vector, y: H1 vector; DEVICE)
        void prolong(const double *x, double *y) const;
        /// x[proj(i)] = 0             (x: H1 vector; DEVICE)
        void orth(double *x) const;

        const H1Space &h1_space() const { return fem; }

        const Mesh2D::EdgeMetricCollection &metrics(const QuadratureRule &quad) const;

    private:
        const H1Space &fem;
        const int _n_faces;
        const int n_basis;
        int ndof;

        host_device_ivec _I, _faces, _proj;
        mutable std::unordered_map<std::string, std::unique_ptr<Mesh2D::EdgeMetricCollection>> _metrics;
    };
} // namespace cuddh

#endif
