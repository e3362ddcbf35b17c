"""Python face of the oracle: loads oracle/ninpol_oracle.c (our CPU restatement) and, when it has
been built, oracle/_ref (the reference's own compiled Grid + IDW/LS/GLS).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by anything under ninpol_amd/.

Two oracles with one interface:

  OracleInterpolator(backend="port")       C restatement in ninpol_oracle.c
  OracleInterpolator(backend="reference")  reference .so files in oracle/_ref through ref_driver

Both share the L3 glue below, which restates the parts of `Interpolator` that cannot be imported
(interpolator.pyx imports meshio, absent from the image):
  process_mesh      interpolator.pyx:255-369
  load_data         interpolator.pyx:372-426
  load_cell_data    interpolator.pyx:428-451  (+ compute_diffusion_magnitude :501-509)
  interpolate       interpolator.pyx:549-629  (dense -> COO -> csr_matrix -> eliminate_zeros)
The glue is pinned by the reference's published accuracy table (tests/results/yaml/accuracy.yaml),
see tests/test_kat.py.
"""
import ctypes
import os
import subprocess
import sys
import sysconfig

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ninpol_amd import topology  # data tables only (element orderings); no compute  # noqa: E402

_LIB = None
I64 = np.int64
F64 = np.float64


def lib_path():
    # NINPOL_ORACLE_LIB: another build of the same C file (tools/sanitize_host.sh: -fsanitize=address,undefined)
    return os.environ.get("NINPOL_ORACLE_LIB") or os.path.join(HERE, "libninpol_oracle.so")


def build_port(force=False):
    """gcc the C restatement.  -ffp-contract=off: see the header of ninpol_oracle.c."""
    src = os.path.join(HERE, "ninpol_oracle.c")
    out = lib_path()
    if os.environ.get("NINPOL_ORACLE_LIB"):
        return out
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-std=c99",
           "-o", out, src, "-lm"]
    subprocess.check_call(cmd)
    return out


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(lib_path()):
            build_port()
        L = ctypes.CDLL(lib_path())
        vp, i64, cp = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_char_p
        L.oracle_grid_create.restype = vp
        L.oracle_grid_create.argtypes = [i64, i64, i64] + [vp] * 7
        L.oracle_grid_destroy.argtypes = [vp]
        L.oracle_grid_get.argtypes = [vp, cp, ctypes.POINTER(vp), ctypes.POINTER(i64),
                                      ctypes.POINTER(ctypes.c_int)]
        L.oracle_grid_scalar.restype = i64
        L.oracle_grid_scalar.argtypes = [vp, cp]
        L.oracle_idw.argtypes = [vp, vp, i64, vp, vp, ctypes.c_int]
        L.oracle_ls.argtypes = [vp, vp, i64, vp, vp, ctypes.c_int]
        L.oracle_gls.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, ctypes.c_int]
        L.oracle_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


REFERENCE_TREE = os.environ.get("NINPOL_REFERENCE", "/root/reference")


def have_reference():
    """True only in the dev container: oracle/_ref holds the compiled reference AND the reference tree it was
    built from is mounted.  oracle/_ref is listed in .gpurunignore -- the reference never travels to the GPU box
    in any form -- and the second condition keeps this False there even if it did."""
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    return (os.path.isdir(os.path.join(REFERENCE_TREE, "ninpol"))
            and os.path.exists(os.path.join(HERE, "_ref", "ninpol_ref_driver" + suffix)))


def _ref_driver():
    p = os.path.join(HERE, "_ref")
    if p not in sys.path:
        sys.path.insert(0, p)
    import ninpol_ref_driver
    return ninpol_ref_driver


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------------------------------------
# L3 glue restated
# ---------------------------------------------------------------------------------------------

def process_mesh(mesh):
    """interpolator.pyx:255-369 -> the positional args of Grid (minus logging / build_edges)."""
    dim = topology.mesh_dimension([b.type for b in mesh.cells])
    npoel, nfael, lnofa, lpofa, nedel, lpoed = topology.topology_tables(dim)
    blocks = [b for b in mesh.cells if b.type in topology.TYPES_PER_DIMENSION[dim]]
    n_elems = sum(len(b.data) for b in blocks)
    conn = -np.ones((n_elems, topology.MAX_POINTS_PER_ELEMENT), dtype=I64)
    etypes = -np.ones(n_elems, dtype=I64)
    at = 0
    for b in blocks:
        d = np.asarray(b.data)
        conn[at:at + len(d), :d.shape[1]] = d
        etypes[at:at + len(d)] = topology.ELEMENTS[b.type]["element_type"]
        at += len(d)
    return dim, n_elems, int(mesh.points.shape[0]), npoel, nfael, lnofa, lpofa, nedel, lpoed, conn, etypes


def compute_diffusion_magnitude(perm):
    """interpolator.pyx:501-509.  The source reads `(1 - (3 * (detKs ** (1 / 3)) / trKs)) ** 2`, but the
    module is compiled with cdivision=True (setup.py:100-108) and `1 / 3` between two C integer
    literals is C integer division = 0, so det ** 0 == 1 and what the reference computes is
    (1 - 3 / tr K)^2.  Confirmed by the published accuracy table: only this form reproduces the
    FAN / ALH GLS columns of tests/results/yaml/accuracy.yaml (to 1e-15; tests/test_kat.py)."""
    Ks = np.reshape(perm, (len(perm), 3, 3))
    det = np.linalg.det(Ks)
    tr = np.trace(Ks, axis1=1, axis2=2)
    return (1 - (3 * (det ** 0) / tr)) ** 2


def load_data(data_dict, n_entities, v2i_section):
    """interpolator.pyx:372-426: (n_vars, n * max_shape) table, variable order = dict order."""
    n_vars = len(data_dict)
    dims = np.zeros(n_vars, dtype=I64)
    max_shape = 1
    for idx, (name, arr) in enumerate(data_dict.items()):
        arr = np.asarray(arr)
        cur = arr.shape[1] if arr.ndim > 1 else 1
        max_shape = max(max_shape, cur)
        v2i_section[name] = idx
        dims[idx] = cur
    table = np.zeros((n_vars, n_entities * max_shape), dtype=F64)
    for name, arr in data_dict.items():
        arr = np.asarray(arr, dtype=F64)
        idx = v2i_section[name]
        cur = dims[idx]
        if cur == 1:
            table[idx, :n_entities] = arr.reshape(len(arr), -1)[:n_entities, 0]
        else:
            table[idx, :n_entities * cur] = arr[:n_entities].reshape(-1)
    return table, dims


def load_cell_data(mesh, dim, n_elems, v2i):
    """interpolator.pyx:428-451: concatenate per-type arrays; 'permeability' adds 'diff_mag'."""
    cdd = mesh.cell_data_dict
    cell_data = {}
    for var in cdd:
        parts = [np.asarray(cdd[var][t]) for t in cdd[var] if t in topology.TYPES_PER_DIMENSION[dim]]
        cell_data[var] = np.concatenate(parts) if parts else np.zeros(0)
        if var == "permeability":
            cell_data["diff_mag"] = compute_diffusion_magnitude(cell_data["permeability"])
    return load_data(cell_data, n_elems, v2i["cells"])


class _Grid:
    """Attribute bag with the reference Grid's readonly names (grid.pxd:128-187), int64/float64."""


_ARRAYS_2D = {"esuel": 6, "infael": 6, "inpofa": 4, "inpoel": 8, "point_coords": 3, "centroids": 3,
              "faces_centers": 3, "normal_faces": 3}
_ARRAY_NAMES = ("esup", "esup_ptr", "psup", "psup_ptr", "fsup", "fsup_ptr", "esuf", "esuf_ptr", "esuel",
                "infael", "inpofa", "inpoel", "boundary_faces", "boundary_points", "point_coords",
                "centroids", "faces_centers", "normal_faces", "faces_areas")
_SCALAR_NAMES = ("dim", "n_elems", "n_points", "n_faces", "MX_ELEMENTS_PER_POINT", "MX_POINTS_PER_POINT",
                 "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")


class OracleInterpolator:
    """load_mesh(mesh_obj=...) / interpolate(variable, method) with the reference's semantics."""

    def __init__(self, backend="port", threads=None):
        assert backend in ("port", "reference")
        if backend == "reference" and not have_reference():
            raise RuntimeError("oracle/_ref is not built (run oracle/build_ref.py in the dev container)")
        self.backend = backend
        self.threads = threads
        self.variable_to_index = {"points": {}, "cells": {}, "faces": {}}
        self.grid = None
        self._h = None
        self._refgrid = None

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().oracle_grid_destroy(self._h)
            self._h = None

    # -- grid ---------------------------------------------------------------------------------
    def load_mesh(self, mesh_obj):
        args = process_mesh(mesh_obj)
        dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed, conn, etypes = args
        coords = np.ascontiguousarray(np.asarray(mesh_obj.points).astype(F64))
        g = _Grid()
        if self.backend == "reference":
            drv = _ref_driver()
            self._refgrid = drv.build_grid(dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel,
                                           lpoed, conn, etypes, coords, 0)
            for k, v in drv.grid_to_dict(self._refgrid).items():
                setattr(g, k, v)
        else:
            L = _lib()
            self._keep = (npoel, nfael, lnofa, lpofa, conn, etypes, coords)
            self._h = L.oracle_grid_create(dim, n_elems, n_points, _ptr(npoel), _ptr(nfael), _ptr(lnofa),
                                           _ptr(lpofa), _ptr(conn), _ptr(etypes), _ptr(coords))
            for name in _SCALAR_NAMES:
                setattr(g, name, int(L.oracle_grid_scalar(self._h, name.encode())))
            for name in _ARRAY_NAMES:
                p, n, isf = ctypes.c_void_p(), ctypes.c_longlong(), ctypes.c_int()
                rc = L.oracle_grid_get(self._h, name.encode(), ctypes.byref(p), ctypes.byref(n), ctypes.byref(isf))
                assert rc == 0, name
                ct = ctypes.c_double if isf.value else ctypes.c_longlong
                if n.value:
                    a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(n.value,)).copy()
                else:
                    a = np.zeros(0, dtype=F64 if isf.value else I64)
                if name in _ARRAYS_2D:
                    a = a.reshape(-1, _ARRAYS_2D[name])
                setattr(g, name, a)
        self.grid = g
        self.variable_to_index = {"points": {}, "cells": {}, "faces": {}}
        if mesh_obj.cell_data:
            self.cells_data, self.cells_data_dimensions = load_cell_data(mesh_obj, dim, n_elems,
                                                                         self.variable_to_index)
        else:
            self.cells_data, self.cells_data_dimensions = np.zeros((1, 1)), np.zeros(1, dtype=I64)
        if mesh_obj.point_data:
            self.points_data, self.points_data_dimensions = load_data(mesh_obj.point_data, n_points,
                                                                      self.variable_to_index["points"])
        else:
            self.points_data, self.points_data_dimensions = np.zeros((1, 1)), np.zeros(1, dtype=I64)
        self.faces_data = np.zeros((1, 1))

    # -- the plugin call, interpolator.pyx:631-670 -----------------------------------------------
    def prepare(self, method, variable, target_points=None):
        g = self.grid
        if target_points is None or len(target_points) == 0:
            target_points = np.arange(g.n_points, dtype=I64)
        target_points = np.ascontiguousarray(target_points, dtype=I64)
        # the methods index weights[point, :] (idw.pyx:72-84 ...): size by n_points so that subsets
        # can be checked too; for the full target set this is exactly interpolator.pyx:650-651
        weights = np.zeros((g.n_points, g.MX_ELEMENTS_PER_POINT), dtype=F64)
        neumann_ws = np.zeros(g.n_points, dtype=F64)
        v2i = self.variable_to_index
        threads = self.threads or 16
        if self.backend == "reference":
            _ref_driver().run_method(method, self._refgrid, self.cells_data, self.points_data,
                                     self.faces_data, v2i, variable, target_points, weights, neumann_ws)
            return weights, neumann_ws
        L = _lib()
        flag = np.ascontiguousarray(self.points_data[v2i["points"]["neumann_flag_" + variable]]).astype(I64)
        nt = len(target_points)
        if method == "idw":
            L.oracle_idw(self._h, _ptr(target_points), nt, _ptr(flag), _ptr(weights), threads)
        elif method == "ls":
            L.oracle_ls(self._h, _ptr(target_points), nt, _ptr(flag), _ptr(weights), threads)
        elif method == "gls":
            E = g.n_elems
            perm = np.ascontiguousarray(self.cells_data[v2i["cells"]["permeability"]][:E * 9])
            dmag = np.ascontiguousarray(self.cells_data[v2i["cells"]["diff_mag"]][:E])
            nval = np.ascontiguousarray(self.points_data[v2i["points"]["neumann_" + variable]][:g.n_points])
            L.oracle_gls(self._h, _ptr(target_points), nt, _ptr(perm), _ptr(dmag), _ptr(flag), _ptr(nval),
                         _ptr(weights), _ptr(neumann_ws), threads)
        else:
            raise ValueError(method)
        return weights, neumann_ws

    # -- interpolator.pyx:549-629 ----------------------------------------------------------------
    def interpolate(self, variable, method, target_points=None):
        g = self.grid
        if method not in ("gls", "idw", "ls"):
            raise ValueError(f"Method '{method}' not supported. Supported methods are: ['gls', 'idw', 'ls']")
        if variable not in self.variable_to_index["cells"]:
            raise ValueError(f"Variable '{variable}' not found in cells data. "
                             "Point -> Cell interpolation not supported yet.")
        weights, neumann_ws = self.prepare(method, variable, target_points)
        n_points = g.n_points
        counts = np.diff(g.esup_ptr)
        rows = np.repeat(np.arange(n_points, dtype=I64), counts)
        local = np.arange(len(g.esup), dtype=I64) - np.repeat(g.esup_ptr[:-1], counts)
        data = weights[rows, local] + neumann_ws[rows]          # interpolator.pyx:618
        W = sp.csr_matrix((data, (rows, g.esup)), shape=(n_points, g.n_elems))
        W.eliminate_zeros()
        return W, neumann_ws
