"""Product-side mesh ingestion (csrc/include/cuddh/meshio.hpp): the reference's text format, uniform refinement and the
partitioner that produces EnsembleSpace labels on irregular meshes -- BASELINE config 5's "meshes/unstructured_square ...
256 subdomains" needs all three and the reference has only a test-side loader (tests/load_unstructured_square.cpp:11-55)."""
import numpy as np
import pytest

import cuddhelmholtz_amd as cd
import oracle
from conftest import GOLDEN, load_unstructured_square
from cuddhelmholtz_amd.meshtools import refine_quads

MESH_DIR = GOLDEN / "unstructured_square"


def test_load_matches_the_reference_fixture():
    xy, elems = load_unstructured_square()
    m = cd.Mesh2D.load(MESH_DIR)
    assert (m.n_nodes(), m.n_elem()) == (140, 119)
    assert np.array_equal(m.vertices(), xy) and np.array_equal(m.elements(), elems)
    om = oracle.Mesh(xy, elems)
    assert np.array_equal(m.boundary_edges(), np.asarray(om.boundary_edges))
    fem = cd.H1Space(m, cd.Basis(4))
    I, ndof = oracle.h1_numbering(om, 4)
    assert fem.size() == ndof and np.array_equal(fem.global_indices().reshape(-1, order="F"), I.reshape(-1, order="F"))


def test_load_reports_missing_files(tmp_path):
    with pytest.raises(RuntimeError, match="cannot open file"):
        cd.Mesh2D.load(tmp_path)
    (tmp_path / "info.txt").write_text("4 1\n")
    (tmp_path / "coordinates.txt").write_text("0 0\n1 0\n1 1\n")
    (tmp_path / "elements.txt").write_text("0 1 2 3\n")
    with pytest.raises(RuntimeError, match="shorter"):
        cd.Mesh2D.load(tmp_path)


@pytest.mark.parametrize("times", [1, 2])
def test_refinement_matches_the_python_utility(times):
    xy, elems = load_unstructured_square()
    x2, e2 = refine_quads(xy, elems, times)
    fine = cd.Mesh2D.load(MESH_DIR).refined(times)
    assert fine.n_elem() == 119 * 4**times
    assert np.array_equal(fine.elements(), e2)
    assert np.allclose(fine.vertices(), x2, rtol=0, atol=1e-15)


@pytest.mark.parametrize("times,n_parts", [(1, 256), (2, 256), (0, 7)])
def test_partition_and_ensemble_space_on_the_refined_fixture(times, n_parts):
    """config 5's EnsembleSpace part: 256 subdomains of the refined unstructured square (r = 1: 476 quads), built
    entirely through the product; every table identical to the oracle's restatement of source/EnsembleSpace.cpp."""
    mesh = cd.Mesh2D.load(MESH_DIR).refined(times) if times else cd.Mesh2D.load(MESH_DIR)
    labels = mesh.partition(n_parts)
    counts = np.bincount(labels, minlength=n_parts)
    assert labels.min() == 0 and labels.max() == n_parts - 1
    assert counts.max() - counts.min() <= 1
    # compact: on average a part's bounding box is a small fraction of the domain
    xy = mesh.vertices()[mesh.elements()].mean(axis=1)
    box = [np.ptp(xy[labels == p], axis=0).prod() for p in range(n_parts)]
    assert np.mean(box) < 4.0 * 4.0 / n_parts
    nb = 3
    fem = cd.H1Space(mesh, cd.Basis(nb))
    E = cd.EnsembleSpace(fem, n_parts, labels)
    om = oracle.Mesh(mesh.vertices(), mesh.elements())
    I, _ = oracle.h1_numbering(om, nb)
    ref = oracle.ensemble(om, I, n_parts, labels)
    assert list(E.dims) == [n_parts, ref.mx_elems, ref.mx_faces, ref.mx_ndof, ref.mx_fdof, ref.cmap.shape[1]]
    for name, want in [("gI", ref.gI), ("sizes", ref.s_dof), ("elements", ref.elems), ("n_elems", ref.s_elems), ("faces", ref.faces),
                       ("n_faces", ref.s_faces), ("sI", ref.sI), ("fI", ref.fI), ("pI", ref.pI), ("fsizes", ref.s_fdof), ("cmap", ref.cmap)]:
        assert np.array_equal(E.array(name), want), name
