"""ninpol_amd.vtk_legacy: the file format the reference's own tests write through meshio (legacy VTK, tests/accuracy_test.py:46,
tests/performance_test.py:46-49) read without meshio.  meshio is absent here, so these are round trips against the published
format (parity unpinned against meshio's writer, see the module docstring): written by our writer in the layout meshio uses
(version 5.1, binary, FIELD arrays) and in the older one (4.2, ASCII), hand-written SCALARS / VECTORS sections, error paths."""
import numpy as np
import pytest

import ninpol_amd
from ninpol_amd import mesh as M, vtk_legacy


def _same_mesh(a, b):
    assert np.array_equal(a.points, b.points)
    assert [c.type for c in a.cells] == [c.type for c in b.cells]
    for ca, cb in zip(a.cells, b.cells):
        assert np.array_equal(ca.data, cb.data)
    assert set(a.point_data) == set(b.point_data) and set(a.cell_data) == set(b.cell_data)
    for k in a.point_data:
        assert np.array_equal(np.asarray(a.point_data[k]), np.asarray(b.point_data[k])), k
    for k in a.cell_data:
        for x, y in zip(a.cell_data[k], b.cell_data[k]):
            assert np.array_equal(np.asarray(x), np.asarray(y)), k


@pytest.mark.parametrize("binary,version", [(True, "5.1"), (False, "5.1"), (True, "4.2"), (False, "4.2")])
@pytest.mark.parametrize("kind", ["hex", "mixed"])
def test_round_trip(tmp_path, binary, version, kind):
    mesh = M.hex_mesh(4, 3, 2, jitter=0.1, seed=1) if kind == "hex" else M.mixed_mesh(6, 3, 3, jitter=0.1, seed=1)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=3)
    fn = str(tmp_path / "m.vtk")
    vtk_legacy.write(fn, mesh, binary=binary, version=version)
    back = vtk_legacy.read(fn)
    _same_mesh(mesh, back)
    assert back.cell_data["permeability"][0].shape[1] == 9
    assert set(back.cell_data_dict["u"]) == {c.type for c in mesh.cells}


def test_load_mesh_from_a_file_equals_the_object(tmp_path):
    """Interpolator.load_mesh(filename=...) on a .vtk file (no meshio here) builds the same grid and tables as mesh_obj=."""
    mesh = M.mixed_mesh(6, 3, 3, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(0, 0.0), seed=4)
    fn = str(tmp_path / "mixed.vtk")
    vtk_legacy.write(fn, mesh)
    A, B = ninpol_amd.Interpolator(), ninpol_amd.Interpolator()
    A.load_mesh(mesh_obj=mesh)
    B.load_mesh(filename=fn)
    for k in ("esup", "esup_ptr", "fsup", "fsup_ptr", "inpoel", "inpofa", "centroids", "normal_faces", "boundary_points"):
        assert np.array_equal(np.asarray(getattr(A.grid, k)), np.asarray(getattr(B.grid, k))), k
    assert A.variable_to_index == B.variable_to_index
    assert np.array_equal(A.cells_data, B.cells_data) and np.array_equal(A.points_data, B.points_data)


def test_classic_attribute_sections_and_errors(tmp_path):
    fn = tmp_path / "hand.vtk"
    fn.write_text("""# vtk DataFile Version 3.0
one tetrahedron and one pyramid
ASCII
DATASET UNSTRUCTURED_GRID
POINTS 6 float
0 0 0  1 0 0  1 1 0
0 1 0  0.5 0.5 1  0.5 0.5 -1
CELLS 2 11
5 0 1 2 3 4
4 0 1 2 5
CELL_TYPES 2
14
10
POINT_DATA 6
SCALARS neumann_flag_u double 1
LOOKUP_TABLE default
0 0 1 1 0 0
VECTORS v float
0 0 0 1 1 1 2 2 2 3 3 3 4 4 4 5 5 5
CELL_DATA 2
TENSORS permeability double
1 0 0 0 1 0 0 0 1
2 0 0 0 2 0 0 0 2
FIELD FieldData 1
u 1 2 double
3.5 4.5
""")
    m = vtk_legacy.read(str(fn))
    assert [c.type for c in m.cells] == ["pyramid", "tetra"] and m.points.dtype == np.float64
    assert np.array_equal(m.cells[0].data, [[0, 1, 2, 3, 4]]) and np.array_equal(m.cells[1].data, [[0, 1, 2, 5]])
    assert np.array_equal(m.point_data["neumann_flag_u"], [0, 0, 1, 1, 0, 0]) and m.point_data["v"].shape == (6, 3)
    assert m.cell_data["permeability"][1].tolist() == [[2, 0, 0, 0, 2, 0, 0, 0, 2]] and m.cell_data["u"][0].tolist() == [3.5]
    bad = tmp_path / "bad.vtk"
    bad.write_text("# vtk DataFile Version 3.0\nt\nASCII\nDATASET STRUCTURED_POINTS\n")
    with pytest.raises(ValueError):
        vtk_legacy.read(str(bad))
    with pytest.raises(ImportError):
        ninpol_amd.Interpolator().load_mesh(filename=str(tmp_path / "mesh.msh"))


def test_metadata_blocks_behind_arrays_are_skipped(tmp_path):
    """VTK / ParaView 9 writers put a METADATA block (INFORMATION ..., ended by a blank line) behind an array -- between two
    SCALARS of a section, and between / behind the arrays of a FIELD.  The section must go on (advisor finding, round 3)."""
    from ninpol_amd import vtk_legacy
    path = tmp_path / "meta.vtk"
    path.write_text("""# vtk DataFile Version 5.1
vtk output
ASCII
DATASET UNSTRUCTURED_GRID
POINTS 4 double
0 0 0 1 0 0 0 1 0 0 0 1
METADATA
INFORMATION 1
NAME L2_NORM_RANGE LOCATION vtkDataArray
DATA 2 0 1

CELLS 2 4
OFFSETS vtktypeint64
0 4
CONNECTIVITY vtktypeint64
0 1 2 3
CELL_TYPES 1
10
POINT_DATA 4
SCALARS a double 1
LOOKUP_TABLE default
1 2 3 4
METADATA
INFORMATION 0

SCALARS b double 1
LOOKUP_TABLE default
5 6 7 8
FIELD FieldData 2
c 1 4 double
9 10 11 12
METADATA
INFORMATION 0

d 3 4 double
1 2 3 4 5 6 7 8 9 10 11 12
METADATA
INFORMATION 0

CELL_DATA 1
FIELD FieldData 1
permeability 9 1 double
1 0 0 0 1 0 0 0 1
METADATA
INFORMATION 0

""")
    m = vtk_legacy.read(str(path))
    assert m.points.shape == (4, 3) and m.cells[0].type == "tetra" and m.cells[0].data.tolist() == [[0, 1, 2, 3]]
    assert m.point_data["a"].tolist() == [1, 2, 3, 4] and m.point_data["b"].tolist() == [5, 6, 7, 8]
    assert m.point_data["c"].tolist() == [9, 10, 11, 12] and m.point_data["d"].shape == (4, 3)
    assert m.cell_data["permeability"][0].shape == (1, 9)
