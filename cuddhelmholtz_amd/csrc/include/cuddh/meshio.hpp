// Mesh ingestion for the irregular-mesh configurations (BASELINE config 5): the reference reads its only
// unstructured fixture with a test-side helper (tests/load_unstructured_square.cpp:11-55) and has neither a refinement
// routine nor a way to produce the `element_labels` EnsembleSpace takes (include/EnsembleSpace.hpp:21) on such a mesh.
//   load_mesh         the reference's text format: <dir>/info.txt "n_pts n_elem", coordinates.txt "x y" per vertex,
//                     elements.txt four 0-based counter-clockwise vertex ids per quadrilateral
//   refine_quads      uniform refinement: every quadrilateral -> 4 through its edge midpoints and centroid
//   partition_elements  n_parts compact, equally sized element sets (labels for EnsembleSpace): elements sorted along the
//                     Morton curve of their centroids, cut into n_parts consecutive runs whose sizes differ by at most one
#ifndef CUDDH_AMD_MESHIO_HPP
#define CUDDH_AMD_MESHIO_HPP

#include <string>
#include <vector>

#include "mesh.hpp"

namespace cuddh
{
    /// vertex coordinates (2, n_pts) and corner ids (4, n_elem) as Mesh2D::from_vertices takes them
    struct QuadMeshData
    {
        std::vector<double> xy;
        std::vector<int> elems;
        int n_pts() const { return static_cast<int>(xy.size() / 2); }
        int n_elem() const { return static_cast<int>(elems.size() / 4); }
    };

    /// reads <dir>/info.txt, coordinates.txt, elements.txt; cuddh_error (throws) when a file cannot be opened or is short
    QuadMeshData read_mesh_files(const std::string &dir);
    Mesh2D load_mesh(const std::string &dir);

    /// vertex and element arrays of an existing mesh
    QuadMeshData mesh_data(const Mesh2D &mesh);

    /// `times` rounds of uniform refinement.  New vertices: the midpoints of the unique edges in increasing order of
    /// (lower end) * (n_pts + 1) + (higher end), then the centroids in element order; child c of an element keeps corner c.
    QuadMeshData refine_quads(const QuadMeshData &mesh, int times);

    /// labels[el] in [0, n_parts): consecutive runs of the centroid Morton order, sizes n_elem / n_parts rounded both ways
    std::vector<int> partition_elements(const Mesh2D &mesh, int n_parts);
} // namespace cuddh

#endif
