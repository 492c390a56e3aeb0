// A partition of an H1Space into element-disjoint subspaces (subdomains) with the
// gather/scatter maps the DDH local solves consume.
// Contract: reference include/EnsembleSpace.hpp:21-118; construction rules follow
// source/EnsembleSpace.cpp:11-287 (elements in increasing global order; subspace
// dofs numbered by first touch over (el, j, i); boundary faces in global edge
// order with side 0/1; face dofs by first touch over (face, i); shared-dof pairs
// cmap = [S0, S1, j0, j1] de-duplicated per subspace pair).  All arrays are
// int32, padded to the per-subspace maximum and filled with -1.
#ifndef CUDDH_AMD_ENSEMBLE_HPP
#define CUDDH_AMD_ENSEMBLE_HPP

#include "memory.hpp"
#include "spaces.hpp"
#include "tensor.hpp"

namespace cuddh
{
    class EnsembleSpace
    {
    public:
        /// element_labels[el] in [0, n_spaces): the subspace element el belongs to
        EnsembleSpace(const H1Space &fem, int n_spaces, const int *element_labels);

        int size() const { return n_spaces; }

        /// (mx_ndof, n_spaces): subspace dof -> global dof
        const_imat_wrapper global_indices(MemorySpace m) const { return reshape(gI.read(m), mx_ndof, n_spaces); }
        /// (n_spaces): number of dofs per subspace
        const_ivec_wrapper sizes(MemorySpace m) const { return reshape(s_dof.read(m), n_spaces); }
        /// (mx_elems, n_spaces): subspace element -> global element
        const_imat_wrapper elements(MemorySpace m) const { return reshape(elems.read(m), mx_elems, n_spaces); }
        const_ivec_wrapper n_elems(MemorySpace m) const { return reshape(s_elems.read(m), n_spaces); }
        /// (mx_faces, n_spaces): boundary faces (global edge ids) of each subspace
        const_imat_wrapper faces(MemorySpace m) const { return reshape(_faces.read(m), mx_faces, n_spaces); }
        const_ivec_wrapper n_faces(MemorySpace m) const { return reshape(s_faces.read(m), n_spaces); }
        /// (n_basis, n_basis, mx_elems, n_spaces): element node -> subspace dof
        TensorWrapper<4, const int> subspace_indices(MemorySpace m) const
        {
            return reshape(sI.read(m), n_basis, n_basis, mx_elems, n_spaces);
        }
        /// (n_basis, mx_faces, n_spaces): face node -> face-space dof
        const_icube_wrapper face_indices(MemorySpace m) const { return reshape(fI.read(m), n_basis, mx_faces, n_spaces); }
        /// (mx_fdof, n_spaces): face-space dof -> subspace dof
        const_imat_wrapper face_proj(MemorySpace m) const { return reshape(pI.read(m), mx_fdof, n_spaces); }
        const_ivec_wrapper fsizes(MemorySpace m) const { return reshape(s_fdof.read(m), n_spaces); }
        /// (4, n_shared): [p, q, i, j] = face dof i of subspace p coincides with face dof j of subspace q
        const_imat_wrapper connectivity_map(MemorySpace m) const { return reshape(cmap.read(m), 4, n_shared_dofs); }

    private:
        const int n_spaces;
        const int n_basis;
        int mx_elems = 0, mx_faces = 0, mx_ndof = 0, mx_fdof = 0, n_shared_dofs = 0;

        host_device_ivec gI, s_dof, elems, s_elems, _faces, s_faces, sI, fI, pI, s_fdof, cmap;
    };
} // namespace cuddh

#endif
