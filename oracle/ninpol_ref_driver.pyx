# cython: language_level=3
"""Thin driver (OUR code, test infrastructure only) over the reference's compiled Grid and
method plugins.  The reference reaches these only through `Interpolator` (interpolator.pyx:194-207
for the grid, :657-665 for the plugin call); `Interpolator` cannot be imported here because it
imports meshio, so this file performs exactly those two call sequences and nothing else.
"""
import numpy as np

from ninpol._interpolator.grid cimport Grid
from ninpol._methods.idw cimport IDWInterpolation
from ninpol._methods.ls cimport LSInterpolation
from ninpol._methods.gls cimport GLSInterpolation

ctypedef long long I_t
ctypedef double F_t


def build_grid(I_t dim, I_t n_elems, I_t n_points,
               I_t[::1] npoel, I_t[::1] nfael, I_t[:, ::1] lnofa, I_t[:, :, ::1] lpofa,
               I_t[::1] nedel, I_t[:, :, ::1] lpoed,
               I_t[:, ::1] connectivity, I_t[::1] element_types,
               F_t[:, ::1] coords, int build_edges=False):
    """Grid(*args); build(); load_point_coords(); calculate_centroids(); calculate_normal_faces()
    -- the sequence of interpolator.pyx:194 and :204-207."""
    cdef Grid g = Grid(dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed,
                       connectivity, element_types, False, build_edges)
    g.build()
    g.load_point_coords(coords)
    g.calculate_centroids()
    g.calculate_normal_faces()
    return g


_GRID_ARRAYS = ("esup", "esup_ptr", "psup", "psup_ptr", "fsup", "fsup_ptr", "esuf", "esuf_ptr",
                "esuel", "infael", "inpofa", "inpoel", "boundary_faces", "boundary_points",
                "point_coords", "centroids", "faces_centers", "normal_faces", "faces_areas")
_GRID_SCALARS = ("dim", "n_elems", "n_points", "n_faces", "n_edges", "MX_ELEMENTS_PER_POINT",
                 "MX_POINTS_PER_POINT", "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")


def grid_to_dict(Grid g):
    """Copy every readonly attribute the hot path reads (grid.pxd:128-187) into numpy arrays."""
    out = {}
    for name in _GRID_ARRAYS:
        out[name] = np.array(getattr(g, name))
    for name in _GRID_SCALARS:
        out[name] = int(getattr(g, name))
    return out


def run_method(str method, Grid grid,
               F_t[:, ::1] cells_data, F_t[:, ::1] points_data, F_t[:, ::1] faces_data,
               dict variable_to_index, str variable, I_t[::1] target_points,
               F_t[:, ::1] weights, F_t[::1] neumann_ws):
    """The plugin call of interpolator.pyx:657-665: prepare(grid, cells_data, points_data,
    faces_data, variable_to_index, variable, target_points, weights[out], neumann_ws[out])."""
    cdef IDWInterpolation idw
    cdef LSInterpolation ls
    cdef GLSInterpolation gls
    if method == "idw":
        idw = IDWInterpolation(False)
        idw.prepare(grid, cells_data, points_data, faces_data, variable_to_index, variable,
                    target_points, weights, neumann_ws)
    elif method == "ls":
        ls = LSInterpolation(False)
        ls.prepare(grid, cells_data, points_data, faces_data, variable_to_index, variable,
                   target_points, weights, neumann_ws)
    elif method == "gls":
        gls = GLSInterpolation(False)
        gls.prepare(grid, cells_data, points_data, faces_data, variable_to_index, variable,
                    target_points, weights, neumann_ws)
    else:
        raise ValueError(method)
