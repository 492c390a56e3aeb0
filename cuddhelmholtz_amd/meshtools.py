"""Host-side mesh utilities that are not part of the reference's API.

`refine_quads`: uniform refinement of a straight-sided quadrilateral mesh (every quad -> 4 through its edge
midpoints and centroid), used to scale the reference's `meshes/unstructured_square` fixture (119 quads) up to
sizes that stress the irregular gather/scatter of the operator kernels (119 * 4^r quads)."""
from __future__ import annotations

import numpy as np


def refine_quads(xy: np.ndarray, elems: np.ndarray, times: int = 1) -> tuple[np.ndarray, np.ndarray]:
    """xy: (n_pts, 2) vertex coordinates; elems: (n_elem, 4) counter-clockwise vertex ids.  Returns the refined pair.
    Children keep the parent's orientation: child c owns the parent's corner c."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    elems = np.asarray(elems, dtype=np.int64).reshape(-1, 4)
    for _ in range(times):
        n_pts, n_elem = len(xy), len(elems)
        # unique edges -> midpoint vertices
        a = elems.reshape(-1)
        b = np.roll(elems, -1, axis=1).reshape(-1)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        key = lo * (n_pts + 1) + hi
        uniq, inverse = np.unique(key, return_inverse=True)
        mid_xy = 0.5 * (xy[uniq // (n_pts + 1)] + xy[uniq % (n_pts + 1)])
        mid = (n_pts + inverse).reshape(n_elem, 4)  # mid[e, s]: midpoint of side corner s -> corner s+1
        cen = n_pts + len(uniq) + np.arange(n_elem)
        cen_xy = xy[elems].mean(axis=1)
        xy = np.vstack([xy, mid_xy, cen_xy])
        c0, c1, c2, c3 = (elems[:, i] for i in range(4))
        m0, m1, m2, m3 = (mid[:, i] for i in range(4))
        children = np.stack([np.stack([c0, m0, cen, m3], 1), np.stack([m0, c1, m1, cen], 1),
                             np.stack([cen, m1, c2, m2], 1), np.stack([m3, cen, m2, c3], 1)], 1)
        elems = children.reshape(-1, 4)
    return xy, elems
