// One rank's share of the global operator apply over several devices (SURVEY 8e "global operator apply"; not in the reference,
// which is single-GPU: examples/Helmholtz.hpp:28-56 is the operator being partitioned).
//
// Elements are split into `world` consecutive runs of the Morton order of their centroids (compact regions,
// partition_elements).  A rank builds an ordinary Mesh2D / H1Space / FaceSpace of its own elements, so its local numbering is
// the reference's rule (source/H1Space.cpp:11-127) on the sub-mesh and every operator kernel and plan applies unchanged; l2g maps
// local to global dofs.  A dof touched by elements of several ranks is OWNED by the lowest of them; the others hold it as HALO.
// Vectors are local [u_loc; v_loc], zero at halo entries, so an inner product is the sum of the ranks' local ones.
// One apply: owners send x at the dofs others hold as halo -> local apply -> halo holders send their partial sums of y back
// and clear them.  Both exchanges list dofs in increasing GLOBAL id on both sides, so no index travels.
#ifndef CUDDH_AMD_PARTITION_HPP
#define CUDDH_AMD_PARTITION_HPP

#include <map>
#include <memory>
#include <vector>

#include "basis.hpp"
#include "mesh.hpp"
#include "spaces.hpp"

namespace cuddh
{
    struct HelmholtzPartition
    {
        int rank = 0, world = 1;
        int ndof_global = 0, n_loc = 0;
        std::vector<int> my_elems;            ///< global ids of this rank's elements, increasing
        std::unique_ptr<Mesh2D> mesh;         ///< the sub-mesh (vertices renumbered in increasing global id)
        std::unique_ptr<H1Space> fem;         ///< H1Space of the sub-mesh
        std::vector<int> faces;               ///< sub-mesh edges on the PHYSICAL boundary (its other boundary edges are cuts)
        std::unique_ptr<FaceSpace> fs;        ///< FaceSpace of `faces`
        std::vector<int> l2g;                 ///< local dof -> global dof
        std::vector<int> face_l2g;            ///< local FaceSpace dof -> global FaceSpace dof (coefficient pick-up)
        std::vector<int> owned, halo;         ///< local dofs this rank owns / holds for another rank, increasing local id
        std::map<int, std::vector<int>> own_to;    ///< rank s -> MY owned local dofs s holds as halo, ordered by global id
        std::map<int, std::vector<int>> halo_from; ///< rank s -> my halo local dofs owned by s, ordered by global id

        /// gmesh / gfem / gfs: the whole problem's mesh, H1Space and boundary FaceSpace (every rank builds them: the ownership rule
        /// needs the global numbering); basis must outlive the partition
        static HelmholtzPartition build(const Mesh2D &gmesh, const Basis &basis, const H1Space &gfem, const FaceSpace &gfs, int rank, int world);
    };
} // namespace cuddh

#endif
