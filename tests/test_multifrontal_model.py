"""The numpy model of the multifrontal cube-node kernel (tools/proto_hex8_mf.py: the arithmetic of
csrc/kernels_gls_hex8mf.hip written lane by lane -- descriptor, per-lane fronts, row-distributed 32 x 12 phase over the
quad, back-substitution, weights) against the oracle on a small jittered hexahedron mesh.  CPU only: it pins the
ALGORITHM (column order, the z / u / s folding of the even cells' rows of R, the pivot-row schedule); the kernel itself
is checked against the oracle in the GPU suite."""
import os
import sys

import numpy as np

import util  # noqa: F401  (path setup via conftest)
from ninpol_amd import mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multifrontal_model_matches_oracle(oracle_lib):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_hex8_mf as P
    m = M.hex_mesh(4, 3, 4, jitter=0.15, seed=2)
    M.attach_fields(m, "u", perm="ALH")
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(m)
    W, _ = o.prepare("gls", "u")
    G = o.grid
    v2i = o.variable_to_index
    perm = o.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
    dmag = o.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
    interior = [p for p in range(G.n_points) if not G.boundary_points[p]]
    assert len(interior) == 3 * 2 * 3
    for p in interior:
        w = P.node_weights(p, G, perm, dmag)
        ref = W[p, :8]
        assert np.abs(w - ref).max() <= 1e-12 * np.abs(ref).max(), p


def test_cube_descriptor_rejects_other_graphs():
    """descriptor(): the cube graph is accepted with the canonical labelling (lane l not adjacent to odd slot 3 - l,
    neighbours in ascending slot order); a graph with a triangle (not bipartite) or a doubled face is not."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_hex8_mf as P
    cells = list(range(100, 108))
    cube = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]   # 12 edges
    fc = {f: (cells[a], cells[b]) for f, (a, b) in enumerate(cube)}
    lanes = P.descriptor(cells, list(range(12)), fc)
    assert lanes is not None and len(lanes) == 4
    even = [e for e, _, _ in lanes]
    assert all(bin(e).count("1") % 2 == 0 for e in even) and sorted(o for _, o, _ in lanes) == [1, 2, 4, 7]
    # 3-regular, 8 vertices, 12 edges, but with triangles: two K4-minus-an-edge joined
    bad = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 4), (3, 5), (4, 5), (4, 6), (5, 7), (6, 7), (6, 7)]
    assert P.descriptor(cells, list(range(12)), {f: (cells[a], cells[b]) for f, (a, b) in enumerate(bad)}) is None


def _model_vs_oracle(oracle_lib, mesh):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_mfw as P
    M.attach_fields(mesh, "u", perm="ALH")
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    W, _ = o.prepare("gls", "u")
    G = o.grid
    v2i = o.variable_to_index
    perm = o.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
    dmag = o.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
    taken = other = 0
    for p in range(G.n_points):
        if G.boundary_points[p]:
            continue
        w = P.node_weights(p, G, perm, dmag)
        if w is None:
            other += 1
            continue
        ref = W[p, :len(w)]
        assert np.abs(w - ref).max() <= 1e-12 * np.abs(ref).max(), p
        taken += 1
    return taken, other


def test_one_wavefront_model_matches_oracle(oracle_lib):
    """The numpy model of the one-wavefront multifrontal kernel (tools/proto_mfw.py: the arithmetic of
    csrc/kernels_gls_mfw.hip -- two-colouring or the largest greedy independent set, fronts with their z / u / s folding,
    free faces' rows, the dense problem with one reduction per column and the pivot row updated by its own reflector, the
    tail) against the oracle: Kuhn tetrahedra (12 fronts + 12 dense cells), wedges (6 + 6), cube nodes (4 + 4), and the
    mixed mesh whose interface and apex nodes are of the general kind (odd cycles, free faces, up to 15 dense cells)."""
    assert _model_vs_oracle(oracle_lib, M.tet_mesh(3, jitter=0.1, seed=1)) == (8, 0)
    assert _model_vs_oracle(oracle_lib, M.wedge_mesh(3, jitter=0.05, seed=1)) == (8, 0)
    assert _model_vs_oracle(oracle_lib, M.hex_mesh(3, jitter=0.1, seed=1)) == (8, 0)
    taken, other = _model_vs_oracle(oracle_lib, M.mixed_mesh(6, 4, 4, jitter=0.1, seed=1))
    assert taken > 0 and other == 0


def test_descriptor_kinds():
    """descriptor(): the cube graph is two-coloured with 4 fronts in esup order; the same graph with one more face between
    two of its dense cells is of the general kind (an odd cycle; that face is free); a plain cycle (its cells have 2 faces,
    not 3) and a node with a boundary face are refused."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import proto_mfw as P
    cells = list(range(50, 58))
    cube = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]
    fc = {f: (cells[a], cells[b]) for f, (a, b) in enumerate(cube)}
    kind, fronts, dense, ff, free = P.descriptor(cells, list(range(12)), fc)
    assert kind == 1 and fronts == [0, 3, 5, 6] and dense == [1, 2, 4, 7] and all(len(x) == 3 for x in ff) and not free
    fc2 = dict(fc)
    fc2[12] = (cells[1], cells[2])            # 0-1 and 0-2 are faces of the cube: 0-1-2 is a triangle now
    kind, fronts, dense, ff, free = P.descriptor(cells, list(range(13)), fc2)
    assert kind == 2 and len(free) >= 1 and all(len(x) == 3 for x in ff) and len(fronts) + len(dense) == 8
    ring = [(i, (i + 1) % 6) for i in range(6)]
    assert P.descriptor(cells[:6], list(range(6)), {f: (cells[a], cells[b]) for f, (a, b) in enumerate(ring)}) is None
    fcb = dict(fc)
    fcb[0] = (cells[0], -1)
    assert P.descriptor(cells, list(range(12)), fcb) is None
