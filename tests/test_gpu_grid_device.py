"""the Grid built on the device (csrc/grid_device.hip) against the native host builder
(csrc/grid_host.cpp, itself pinned bit for bit to the reference's fixtures in test_host.py / test_oracle.py) and
against the golden grids.  Every array, integers and geometry, must be identical; the weights computed on an
adopted device grid must be those of the host-built path."""
import numpy as np
import pytest

import util
from ninpol_amd import mesh as M

pytestmark = pytest.mark.gpu

ARRAYS = ("esup_ptr", "esup", "psup_ptr", "psup", "fsup_ptr", "fsup", "esuf_ptr", "esuf", "esuel", "infael", "inpofa",
          "inpoel", "boundary_faces", "boundary_points", "point_coords", "centroids", "faces_centers", "normal_faces",
          "faces_areas", "element_types")
SCALARS = ("dim", "n_elems", "n_points", "n_faces", "MX_ELEMENTS_PER_POINT", "MX_POINTS_PER_POINT",
           "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")


def _both(mesh, **kw):
    import ninpol_amd
    a = ninpol_amd.Interpolator(grid_build="host", **kw)
    a.load_mesh(mesh_obj=mesh)
    b = ninpol_amd.Interpolator(grid_build="device", **kw)
    b.load_mesh(mesh_obj=mesh)
    return a, b


def _meshes():
    yield "hex12_jitter", M.hex_mesh(12, jitter=0.15, seed=0)
    yield "hex_slab", M.hex_mesh(9, 7, 5)
    yield "tet6_jitter", M.tet_mesh(6, jitter=0.1, seed=1)
    yield "wedge7", M.wedge_mesh(7, 5, 4, jitter=0.05, seed=2)
    yield "mixed", M.mixed_mesh(10, 6, 6, jitter=0.1, seed=3)
    yield "quad_tri_2d", M.quad_tri_mesh_2d(11, 8, jitter=0.1, seed=4)
    yield "delaunay9", M.delaunay_tet_mesh(9, seed=7)


@pytest.mark.parametrize("name,mesh", list(_meshes()), ids=[m[0] for m in _meshes()])
def test_device_grid_equals_host_grid(name, mesh):
    M.attach_fields(mesh, "u", perm="ALH" if name != "quad_tri_2d" else "LIN", neumann_plane=(0, 0.0), seed=5)
    a, b = _both(mesh)
    for k in SCALARS:
        assert getattr(a.grid, k) == getattr(b.grid, k), k
    for k in ARRAYS:
        x, y = getattr(a.grid, k), getattr(b.grid, k)
        assert x.dtype == y.dtype and x.shape == y.shape, k
        assert np.array_equal(x, y), k      # floats too: same operations in the same order, no contraction
    methods = ("idw", "ls") if name == "quad_tri_2d" else ("idw", "ls", "gls")
    for meth in methods:
        Wa, na = a.interpolate("u", meth)
        Wb, nb = b.interpolate("u", meth)
        assert np.array_equal(Wa.indptr, Wb.indptr) and np.array_equal(Wa.indices, Wb.indices), meth
        assert np.array_equal(Wa.data, Wb.data, equal_nan=True), meth
        assert np.array_equal(na, nb, equal_nan=True), meth


@pytest.mark.parametrize("case", util.GOLDEN_CASES)
def test_device_grid_matches_golden(case):
    import ninpol_amd
    mesh, z = util.load_golden(case)
    I = ninpol_amd.Interpolator(grid_build="device")
    I.load_mesh(mesh_obj=mesh)
    util.assert_grid_equal(I.grid, z)


def test_device_grid_edges_and_errors():
    import ninpol_amd
    mesh = M.hex_mesh(5, jitter=0.1, seed=1)
    M.attach_fields(mesh, "u", perm="LIN")
    a, b = _both(mesh, build_edges=True)
    assert a.grid.n_edges == b.grid.n_edges
    for k in ("inedel", "inpoed"):
        assert np.array_equal(getattr(a.grid, k), getattr(b.grid, k)), k
    bad = M.hex_mesh(3)
    bad.cells[0].data[0, 0] = 10 ** 6   # a point id outside the mesh: the reference would read out of bounds
    I = ninpol_amd.Interpolator(grid_build="device")
    with pytest.raises(ValueError):
        I.load_mesh(mesh_obj=bad)


def test_device_grid_large_hex_properties():
    """1 M-cell hexahedron mesh: closed-form counts (SURVEY 8 header) and a checksum of every array against the
    host builder."""
    import ninpol_amd
    N = 100
    mesh = M.hex_mesh(N, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="LIN")
    a, b = _both(mesh)
    g = b.grid
    assert (g.n_points, g.n_elems, g.n_faces) == ((N + 1) ** 3, N ** 3, 3 * N * N * (N + 1))
    assert int(g.boundary_faces.sum()) == 6 * N * N
    assert g.MX_ELEMENTS_PER_POINT == 8 and g.MX_FACES_PER_POINT == 12
    for k in ARRAYS:
        if k in ("psup", "psup_ptr"):
            continue
        assert np.array_equal(getattr(a.grid, k), getattr(g, k)), k
