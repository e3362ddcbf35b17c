"""Element topology tables (data, not code).

These are the local point/face/edge orderings the reference ships as
`ninpol/utils/point_ordering.yaml:7-53` (meshio / VTK vertex order, faces counter-clockwise seen
from outside) together with the fixed array widths of `ninpol/_interpolator/ninpol_defines.pxd:2-8`.
They are part of the integer-parity contract: the local face order fixes the global face
numbering (`grid.pyx:304-345`), hence the row order of `fsup`/`esuf` and of every GLS system.
"""
import numpy as np

MAX_POINTS_PER_ELEMENT = 8
MAX_FACES_PER_ELEMENT = 6
MAX_POINTS_PER_FACE = 4
NUM_ELEMENT_TYPES = 8
MAX_EDGES_PER_ELEMENT = 12
MAX_ELEMENTS_PER_FACE = 2
MAX_POINTS_PER_EDGE = 2

# insertion order matters: it is the order `process_mesh` walks the YAML in (interpolator.pyx:300)
ELEMENTS = {
    "vertex": dict(element_type=0, number_of_points=1, edges=[], faces=[]),
    "line": dict(element_type=1, number_of_points=2, edges=[[0, 1]], faces=[]),
    "triangle": dict(element_type=2, number_of_points=3,
                     edges=[[0, 1], [1, 2], [2, 0]], faces=[]),
    "quad": dict(element_type=3, number_of_points=4,
                 edges=[[0, 1], [1, 2], [2, 3], [3, 0]], faces=[]),
    "tetra": dict(element_type=4, number_of_points=4,
                  edges=[[0, 1], [1, 2], [2, 0], [0, 3], [1, 3], [2, 3]],
                  faces=[[0, 2, 1], [0, 1, 3], [1, 2, 3], [0, 3, 2]]),
    "hexahedron": dict(element_type=5, number_of_points=8,
                       edges=[[0, 1], [1, 2], [2, 3], [3, 0], [4, 5], [5, 6], [6, 7], [7, 4],
                              [0, 4], [1, 5], [2, 6], [3, 7]],
                       faces=[[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [1, 2, 6, 5],
                              [2, 3, 7, 6], [3, 0, 4, 7]]),
    "wedge": dict(element_type=6, number_of_points=6,
                  edges=[[0, 1], [1, 2], [2, 0], [3, 4], [4, 5], [5, 3], [0, 3], [1, 4], [2, 5]],
                  faces=[[0, 2, 1], [3, 4, 5], [0, 1, 4, 3], [1, 2, 5, 4], [0, 3, 5, 2]]),
    "pyramid": dict(element_type=7, number_of_points=5,
                    edges=[[0, 1], [1, 2], [2, 3], [3, 0], [0, 4], [1, 4], [2, 4], [3, 4]],
                    faces=[[0, 3, 2, 1], [0, 1, 4], [1, 2, 4], [2, 3, 4], [3, 0, 4]]),
}

POINT_ORDERING = {"elements": ELEMENTS}

TYPES_PER_DIMENSION = {
    0: ["vertex"],
    1: ["line"],
    2: ["triangle", "quad"],
    3: ["tetra", "hexahedron", "wedge", "pyramid"],
}


def mesh_dimension(cell_types):
    """max topological dimension over the cell blocks (interpolator.pyx:291-294)."""
    dim = 1
    for t in cell_types:
        for d, names in TYPES_PER_DIMENSION.items():
            if t in names:
                dim = max(dim, d)
    return dim


def topology_tables(dim):
    """The six -1 padded int64 tables `process_mesh` hands to `Grid` (interpolator.pyx:274-330).

    npoel is filled for every type; nfael/lnofa/lpofa/nedel/lpoed only for the types of dimension
    `dim`.  In 2-D the reference uses each element's "edges" list as its faces
    (interpolator.pyx:296-298, :306-323): nfael = number of edges, lnofa = 2, lpofa = the edge's points.
    """
    npoel = -np.ones(NUM_ELEMENT_TYPES, dtype=np.int64)
    nfael = -np.ones(NUM_ELEMENT_TYPES, dtype=np.int64)
    lnofa = -np.ones((NUM_ELEMENT_TYPES, MAX_FACES_PER_ELEMENT), dtype=np.int64)
    lpofa = -np.ones((NUM_ELEMENT_TYPES, MAX_FACES_PER_ELEMENT, MAX_POINTS_PER_FACE), dtype=np.int64)
    nedel = -np.ones(NUM_ELEMENT_TYPES, dtype=np.int64)
    lpoed = -np.ones((NUM_ELEMENT_TYPES, MAX_EDGES_PER_ELEMENT, 2), dtype=np.int64)
    faces_key = "edges" if dim == 2 else "faces"
    for name, e in ELEMENTS.items():
        t = e["element_type"]
        npoel[t] = e["number_of_points"]
        if name not in TYPES_PER_DIMENSION[dim]:
            continue
        flist = e[faces_key]
        nfael[t] = len(flist)
        for i, f in enumerate(flist):
            lnofa[t, i] = len(f)
            for j, p in enumerate(f):
                lpofa[t, i, j] = p
        nedel[t] = len(e["edges"])
        for i, ed in enumerate(e["edges"]):
            for j, p in enumerate(ed):
                lpoed[t, i, j] = p
    return npoel, nfael, lnofa, lpofa, nedel, lpoed
