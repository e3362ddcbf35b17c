"""CPU-side tests of the product's host layer (no GPU needed): the C-ABI library loads and exports
everything include/ninpol_amd.h declares, the native host Grid reproduces the reference's arrays bit
for bit (golden fixtures + live oracle), the Interpolator mirrors the reference's tables, errors and
quirks, and the weight kernels refuse to run without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import util
from ninpol_amd import mesh as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ninpol_amd import build as nbuild
    nbuild.build()
    from ninpol_amd import _lib
    return _lib


def test_abi_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "ninpol_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(nin_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 15
    L = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/ninpol_amd.h but not exported"
    assert set(declared) == set(lib.EXPORTS)
    assert lib.load().nin_version().decode().startswith("ninpol_amd")


@pytest.mark.parametrize("case", util.GOLDEN_CASES)
def test_host_grid_matches_golden(lib, case):
    import ninpol_amd
    mesh, z = util.load_golden(case)
    I = ninpol_amd.Interpolator(build_edges=True)
    I.load_mesh(mesh_obj=mesh)
    util.assert_grid_equal(I.grid, z)
    np.testing.assert_array_equal(I.cells_data, z["cells_data"])
    np.testing.assert_array_equal(I.points_data, z["points_data"])
    assert list(I.variable_to_index["cells"]) == [str(s) for s in z["cells_vars"]]
    assert list(I.variable_to_index["points"]) == [str(s) for s in z["points_vars"]]
    for k in util.GRID_ARRAYS:
        assert getattr(I.grid, k).dtype in (np.int64, np.float64)


def test_host_grid_matches_oracle_large_and_threads(lib, oracle_lib):
    import ninpol_amd
    mesh = M.mixed_mesh(14, 9, 8, jitter=0.1, seed=6)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0))
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    for nt in (1, 3, 8):
        I = ninpol_amd.Interpolator(num_threads=nt)
        I.load_mesh(mesh_obj=mesh)
        for k in util.GRID_SCALARS:
            assert getattr(I.grid, k) == getattr(o.grid, k), k
        for k in util.GRID_ARRAYS:
            np.testing.assert_array_equal(getattr(I.grid, k), getattr(o.grid, k), err_msg=k)


def test_edges_match_reference(lib, oracle_lib):
    """build_edges=True: inedel / inpoed / n_edges, including the reference's hash-keyed numbering."""
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built")
    import ninpol_amd
    mesh = M.mixed_mesh(6, 4, 4, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u")
    args = oracle_lib.process_mesh(mesh)
    drv = oracle_lib._ref_driver()
    ref = drv.build_grid(*args, np.ascontiguousarray(mesh.points), 1).get_data()
    I = ninpol_amd.Interpolator(build_edges=True)
    I.load_mesh(mesh_obj=mesh)
    mine = I.grid.get_data()
    assert sorted(mine) == sorted(ref)
    for k in ref:
        np.testing.assert_array_equal(np.asarray(mine[k]), np.asarray(ref[k]), err_msg=k)


def test_reference_errors_and_quirks(lib):
    import ninpol_amd
    I = ninpol_amd.Interpolator()
    with pytest.raises(ValueError, match="Filename for the mesh or meshio.Mesh object must be provided."):
        I.load_mesh()
    with pytest.raises(ValueError, match="Grid not initialized. Please load a mesh first."):
        I.interpolate("u", "gls")
    assert list(I.supported_methods) == ["gls", "idw", "ls"]
    mesh = M.hex_mesh(3)
    M.attach_fields(mesh, "u")
    mesh.cell_data["vec"] = [np.zeros((27, 3))]
    I.load_mesh(mesh_obj=mesh)
    with pytest.raises(ValueError, match=r"Method 'foo' not supported. Supported methods are: \['gls', 'idw', 'ls'\]"):
        I.interpolate("u", "foo")
    with pytest.raises(ValueError, match="Variable 'nope' not found in cells data. Point -> Cell interpolation not supported yet."):
        I.interpolate("nope", "idw")
    with pytest.raises(ValueError, match="Variable 'vec' has more than one dimension. Vector data not supported yet."):
        I.interpolate("vec", "idw")
    with pytest.raises(ValueError, match="Invalid shape in axis 0: 0."):
        I.grid.get_data()
    with pytest.raises(ValueError, match="not found in cells data"):
        I.get_data("cells", np.arange(3), "nope")
    assert I.get_data("cells", np.arange(3), "u").shape == (3,)
    d = I.get_dict()
    assert list(d["variable_to_index"]["cells"]) == ["permeability", "diff_mag", "u", "vec"]
    assert d["cells_data"].shape == (4, 27 * 9)
    # diff_mag as the reference computes it: (1 - 3 / tr K)^2 (cdivision quirk, see test_kat.py)
    K = mesh.cell_data["permeability"][0].reshape(-1, 3, 3)
    np.testing.assert_array_equal(I.get_data("cells", np.arange(27), "diff_mag"),
                                  (1 - (3 * np.ones(27)) / np.trace(K, axis1=1, axis2=2)) ** 2)
    with pytest.raises(ValueError, match="The number of elements must be greater than 0."):
        ninpol_amd.Grid(3, 0, 8, *[np.zeros(1)] * 6, np.zeros((0, 8)), np.zeros(0), coords=np.zeros((8, 3)))


def test_load_face_data(lib):
    import ninpol_amd
    mesh = M.hex_mesh(2)
    M.attach_fields(mesh, "u")
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    F = I.grid.n_faces
    vals = np.arange(F, dtype=float)
    I.load_face_data({"flux": vals})
    np.testing.assert_array_equal(I.faces_data[0], vals)
    perm = np.random.default_rng(0).permutation(F)
    I.load_face_data({"flux": vals}, face_connectivity=I.grid.inpofa[perm])
    np.testing.assert_array_equal(I.faces_data[I.variable_to_index["faces"]["flux"]], vals[perm])


def test_no_cpu_fallback(lib):
    """Without a GPU the hot path must fail loudly (NIN_ENODEVICE), never compute on the host."""
    import ninpol_amd
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the -m gpu tests")
    mesh = M.hex_mesh(3)
    M.attach_fields(mesh, "u")
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    with pytest.raises(lib.NinpolError) as ei:
        I.interpolate("u", "idw")
    assert ei.value.code == lib.NIN_ENODEVICE


def test_device_grid_build_has_no_cpu_fallback(lib):
    """`grid_build="device"` without a GPU is NIN_ENODEVICE, not a silent host build."""
    import ninpol_amd
    if lib.device_count() > 0:
        pytest.skip("a GPU is visible: covered by tests/test_gpu_grid_device.py")
    mesh = M.hex_mesh(3)
    M.attach_fields(mesh, "u")
    I = ninpol_amd.Interpolator(grid_build="device")
    with pytest.raises(lib.NinpolError) as ei:
        I.load_mesh(mesh_obj=mesh)
    assert ei.value.code == lib.NIN_ENODEVICE
    with pytest.raises(ValueError):
        ninpol_amd.Interpolator(grid_build="somewhere")


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ninpol_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"ninpol_oracle|oracle/|libninpol_oracle|ninpol_ref_driver", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_2d_mesh_host_grid_matches_oracle(lib, oracle_lib):
    """dim = 2: faces are the elements' edges (interpolator.pyx:296-298); grid arrays against the oracle
    (and against the reference build when present)."""
    import ninpol_amd
    mesh = M.quad_tri_mesh_2d(8, 6, jitter=0.1, seed=3)
    M.attach_fields(mesh, "u", perm="LIN")
    backends = ["port"] + (["reference"] if oracle_lib.have_reference() else [])
    I = ninpol_amd.Interpolator()
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.dim == 2
    for be in backends:
        o = oracle_lib.OracleInterpolator(be, threads=1)
        o.load_mesh(mesh)
        for k in util.GRID_SCALARS:
            assert getattr(I.grid, k) == getattr(o.grid, k), (be, k)
        for k in util.GRID_ARRAYS:
            np.testing.assert_array_equal(getattr(I.grid, k), getattr(o.grid, k), err_msg=f"{be}:{k}")


def test_load_arrays_equals_load_mesh():
    """the array-based entry builds the same tables and grid as the mesh-object entry."""
    import ninpol_amd
    from ninpol_amd import mesh as M
    m = M.mixed_mesh(6, 4, 4, jitter=0.1, seed=1)
    M.attach_fields(m, "u", perm="ALH", neumann_plane=(2, 0.0))
    a = ninpol_amd.Interpolator()
    a.load_mesh(mesh_obj=m)
    b = ninpol_amd.Interpolator()
    b.load_arrays(m.points, [(c.type, c.data) for c in m.cells],
                  cell_data={k: np.concatenate(v) for k, v in m.cell_data.items()}, point_data=m.point_data)
    assert a.variable_to_index == b.variable_to_index
    assert np.array_equal(a.cells_data, b.cells_data) and np.array_equal(a.points_data, b.points_data)
    for k in ("esup", "esup_ptr", "fsup", "inpofa", "centroids", "normal_faces"):
        assert np.array_equal(getattr(a.grid, k), getattr(b.grid, k)), k


def test_native_table_packing_matches_numpy(lib):
    """process_mesh / load_data / diff_mag run in the native library (csrc/pack_host.cpp); the tables must
    be what the vectorised numpy formulation (and, through the golden fixtures, the reference) gives, bit for bit."""
    import ninpol_amd
    from ninpol_amd import topology as T
    mesh = M.mixed_mesh(7, 4, 4, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=1)
    I = ninpol_amd.Interpolator()
    args = I.process_mesh(mesh)
    conn, etypes = args[9], args[10]
    n_elems = sum(len(b.data) for b in mesh.cells)
    ref_c = -np.ones((n_elems, 8), dtype=np.int64)
    ref_t = -np.ones(n_elems, dtype=np.int64)
    at = 0
    for b in mesh.cells:
        d = np.asarray(b.data)
        ref_c[at:at + len(d), :d.shape[1]] = d
        ref_t[at:at + len(d)] = T.ELEMENTS[b.type]["element_type"]
        at += len(d)
    np.testing.assert_array_equal(conn, ref_c)
    np.testing.assert_array_equal(etypes, ref_t)
    K = np.concatenate(mesh.cell_data["permeability"]).reshape(-1, 9)
    tr = (K[:, 0] + K[:, 4]) + K[:, 8]
    np.testing.assert_array_equal(I.compute_diffusion_magnitude(K), (1 - (3 * 1.0 / tr)) ** 2)
    I.load_mesh(mesh_obj=mesh)
    v2i = I.variable_to_index
    assert list(v2i["cells"]) == ["permeability", "diff_mag", "u"]          # dict-insertion order of the reference
    E = I.grid.n_elems
    np.testing.assert_array_equal(np.asarray(I.cells_data)[v2i["cells"]["permeability"]][:E * 9], K.reshape(-1))
    np.testing.assert_array_equal(np.asarray(I.cells_data)[v2i["cells"]["u"]][:E], np.concatenate(mesh.cell_data["u"]))
    np.testing.assert_array_equal(np.asarray(I.points_data)[v2i["points"]["neumann_u"]], mesh.point_data["neumann_u"])
