"""The structural invariants of the reference's own grid test (tests/grid_test.py:126-244, dead there behind an
unconditional return at :60) as property tests of our builders: on every mesh family with the host builder (CPU),
and at 1 M cells with the device builder (GPU)."""
import numpy as np
import pytest

import util  # noqa: F401
from ninpol_amd import mesh as M
from ninpol_amd import topology as T


def check_invariants(g, sample=None):
    P, E, F = g.n_points, g.n_elems, g.n_faces
    inpoel, etype = g.inpoel, g.element_types
    npoel = np.array([T.topology_tables(g.dim)[0][t] for t in etype])            # points per element
    # inpoel: exactly npoel valid ids then -1 padding (grid_test.py:126-137)
    cols = np.arange(inpoel.shape[1])[None, :]
    assert np.all((inpoel >= 0) == (cols < npoel[:, None]))
    assert inpoel.max() < P
    # esup: every element of row i contains node i; rows ascending; total = sum npoel (:201-207)
    ptr, esup = g.esup_ptr, g.esup
    assert ptr[0] == 0 and ptr[-1] == len(esup) == npoel.sum()
    rows = np.repeat(np.arange(P), np.diff(ptr))
    assert np.all((inpoel[esup] == rows[:, None]).any(axis=1))
    same_row = rows[1:] == rows[:-1]
    assert np.all(esup[1:][same_row] > esup[:-1][same_row])
    assert g.MX_ELEMENTS_PER_POINT == np.diff(ptr).max()
    # infael / inpofa: every face of an element is made of that element's points (:140-160)
    infael, inpofa = g.infael, g.inpofa
    nfael = np.array([T.topology_tables(g.dim)[1][t] for t in etype])
    fcols = np.arange(infael.shape[1])[None, :]
    assert np.all((infael >= 0) == (fcols < nfael[:, None]))
    assert infael.max() == F - 1 and len(np.unique(infael[infael >= 0])) == F
    idx = np.arange(E) if sample is None else sample
    for e in idx[:2000]:
        pts = set(inpoel[e][inpoel[e] >= 0])
        for f in infael[e][infael[e] >= 0]:
            fp = inpofa[f][inpofa[f] >= 0]
            assert set(fp) <= pts
    # esuf: every element of face f lists f in infael; 1 or 2 elements; boundary flag = one element (:222-227)
    eptr, esuf = g.esuf_ptr, g.esuf
    cnt = np.diff(eptr)
    assert set(np.unique(cnt)) <= {1, 2}
    assert np.array_equal(g.boundary_faces.astype(bool), cnt == 1)
    frows = np.repeat(np.arange(F), cnt)
    assert np.all((infael[esuf] == frows[:, None]).any(axis=1))
    # fsup: the faces listed for node i contain node i, ascending, and every (face, point) pair is listed
    fptr, fsup = g.fsup_ptr, g.fsup
    prow = np.repeat(np.arange(P), np.diff(fptr))
    assert np.all((inpofa[fsup] == prow[:, None]).any(axis=1))
    assert len(fsup) == (inpofa >= 0).sum()
    same = prow[1:] == prow[:-1]
    assert np.all(fsup[1:][same] > fsup[:-1][same])
    # boundary points = points of boundary faces
    bp = np.zeros(P, dtype=bool)
    bf = inpofa[g.boundary_faces.astype(bool)]
    bp[bf[bf >= 0]] = True
    assert np.array_equal(bp, g.boundary_points.astype(bool))
    # esuel: symmetric, -1 exactly on boundary faces
    esuel = g.esuel
    assert (esuel[infael >= 0] == -1).sum() == int(g.boundary_faces.sum())
    e_idx, j_idx = np.nonzero(esuel >= 0)
    nb = esuel[e_idx, j_idx]
    assert np.all((esuel[nb] == e_idx[:, None]).any(axis=1))
    # centroids are the vertex means (grid.pyx:699-704; grid_test.py:22-27)
    X = g.point_coords
    X3 = np.zeros((P, 3)); X3[:, :X.shape[1]] = X
    safe = np.where(inpoel >= 0, inpoel, 0)
    mean = (X3[safe] * (inpoel >= 0)[:, :, None]).sum(axis=1) / npoel[:, None]
    assert np.allclose(g.centroids, mean, rtol=0, atol=1e-14)
    # unit normals, positive areas
    nrm = np.linalg.norm(g.normal_faces, axis=1)
    assert np.allclose(nrm, 1.0, atol=1e-6) and np.all(g.faces_areas > 0)


def check_psup(g):
    """psup: symmetric, no self, exactly the other points of the surrounding elements (grid_test.py:210-219)."""
    ptr, psup = g.psup_ptr, g.psup
    P = g.n_points
    rows = np.repeat(np.arange(P), np.diff(ptr))
    assert np.all(psup != rows)
    pairs = set(zip(rows.tolist(), psup.tolist()))
    assert all((b, a) in pairs for a, b in pairs)
    assert g.MX_POINTS_PER_POINT == np.diff(ptr).max()


def _meshes():
    yield "hex", M.hex_mesh(5, 4, 6, jitter=0.1, seed=1)
    yield "tet", M.tet_mesh(4, jitter=0.1, seed=2)
    yield "wedge", M.wedge_mesh(4, 3, 3, jitter=0.05, seed=3)
    yield "mixed", M.mixed_mesh(7, 4, 4, jitter=0.1, seed=4)
    yield "fan", M.wedge_fan(17, 3, jitter=0.02, seed=5)
    yield "quad_tri_2d", M.quad_tri_mesh_2d(7, 5, jitter=0.1, seed=6)


@pytest.mark.parametrize("name,mesh", list(_meshes()), ids=[m[0] for m in _meshes()])
def test_host_grid_invariants(name, mesh):
    import ninpol_amd
    M.attach_fields(mesh, "u", perm="LIN")
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    check_invariants(I.grid)
    check_psup(I.grid)


@pytest.mark.gpu
def test_device_grid_invariants_1m_cells():
    import ninpol_amd
    mesh = M.mixed_mesh(100, 60, 60, jitter=0.1, seed=7)     # 1.26 M cells: hexahedra, pyramids, tetrahedra
    M.attach_fields(mesh, "u", perm="LIN")
    I = ninpol_amd.Interpolator(grid_build="device")
    I.load_mesh(mesh_obj=mesh)
    rng = np.random.default_rng(0)
    check_invariants(I.grid, sample=rng.choice(I.grid.n_elems, 2000, replace=False))
