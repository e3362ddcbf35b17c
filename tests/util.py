"""Shared helpers for the test-suite (golden loading, tolerances)."""
import glob
import os

import numpy as np

from ninpol_amd import mesh as M

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))

GRID_ARRAYS = ("esup", "esup_ptr", "psup", "psup_ptr", "fsup", "fsup_ptr", "esuf", "esuf_ptr", "esuel",
               "infael", "inpofa", "inpoel", "boundary_faces", "boundary_points", "point_coords",
               "centroids", "faces_centers", "normal_faces", "faces_areas")
GRID_SCALARS = ("dim", "n_elems", "n_points", "n_faces", "MX_ELEMENTS_PER_POINT", "MX_POINTS_PER_POINT",
                "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")

# north_star: "within 1e-10 relative for the float64 weights".  Weights of one node are O(1/n_elem)
# and sum to ~1, so the error is measured relative to the largest weight of the row.
WEIGHT_RTOL = 1e-10


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    nb = int(z["n_blocks"])
    cells = [M.CellBlock(str(z[f"block{b}_type"]), z[f"block{b}_data"]) for b in range(nb)]
    mesh = M.Mesh(z["points"], cells,
                  point_data={"neumann_flag_u": z["neumann_flag_u"], "neumann_u": z["neumann_u"]},
                  cell_data={"permeability": [z[f"permeability_block{b}"] for b in range(nb)],
                             "u": [z[f"u_block{b}"] for b in range(nb)]})
    return mesh, z


def assert_grid_equal(grid, z, prefix="grid_"):
    """bit-exact: integers AND geometry (centroids, face centres, float32 normals, areas)."""
    for k in GRID_SCALARS:
        assert int(getattr(grid, k)) == int(z[prefix + k]), k
    for k in GRID_ARRAYS:
        a = np.asarray(getattr(grid, k))
        b = z[prefix + k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        np.testing.assert_array_equal(a, b, err_msg=k)


def rowscaled_err(a, b):
    """max over rows of |a-b| / max|b_row| for dense (n, w) tables; NaNs must coincide."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    if a.size == 0:
        return 0.0
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), "inf pattern differs"
    d = np.where(fin, np.abs(np.where(fin, a, 0.0) - np.where(fin, b, 0.0)), 0.0)
    if a.ndim == 1:
        scale = max(np.abs(np.where(fin, b, 0.0)).max(), 1e-300)
        return float(d.max() / scale)
    scale = np.abs(np.where(fin, b, 0.0)).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    return float((d / scale).max())


def csr_rowscaled_err(W, indptr, indices, data):
    """compare a scipy CSR with a golden (indptr, indices, data): pattern exact, values row-scaled."""
    np.testing.assert_array_equal(W.indptr, indptr)
    np.testing.assert_array_equal(W.indices, indices)
    assert W.indices.dtype == indices.dtype
    a, b = W.data, data
    assert np.array_equal(np.isnan(a), np.isnan(b))
    if len(b) == 0:
        return 0.0
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    bf = np.where(np.isfinite(b), np.abs(b), 0.0)
    scale = np.zeros(len(indptr) - 1)
    np.maximum.at(scale, rows, bf)
    scale[scale == 0] = 1.0
    d = np.where(np.isfinite(b), np.abs(np.nan_to_num(a) - np.nan_to_num(b)), 0.0)
    return float((d / scale[rows]).max())
