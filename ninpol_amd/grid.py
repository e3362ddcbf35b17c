"""`Grid`: host mirror of the reference's `ninpol.Grid` (ninpol/_interpolator/grid.pyx, grid.pxd:128-187)
over the native builder in csrc/grid_host.cpp.  Same constructor arguments, same readonly attribute
names, same dtypes on the Python side (int64 / float64); the arrays themselves live in the C++
object as int32 and are converted on first access.
"""
import ctypes
import warnings

import numpy as np

from . import _lib
from . import topology as T

_SCALARS = ("dim", "n_elems", "n_points", "n_faces", "n_edges", "MX_ELEMENTS_PER_POINT",
            "MX_POINTS_PER_POINT", "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")
_SHAPES = {"esuel": T.MAX_FACES_PER_ELEMENT, "infael": T.MAX_FACES_PER_ELEMENT, "inpofa": T.MAX_POINTS_PER_FACE,
           "inpoel": T.MAX_POINTS_PER_ELEMENT, "inedel": T.MAX_EDGES_PER_ELEMENT, "inpoed": 2,
           "centroids": 3, "faces_centers": 3, "normal_faces": 3, "point_coords": 3}
_ARRAYS = ("esup", "esup_ptr", "psup", "psup_ptr", "fsup", "fsup_ptr", "esuf", "esuf_ptr", "esuel", "infael",
           "inpofa", "inpoel", "inpoed", "inedel", "boundary_faces", "boundary_points", "point_coords",
           "centroids", "faces_centers", "normal_faces", "faces_areas", "element_types")


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _PlanCounts(dict):
    """Nodes per kernel of the GLS launch plan; `plan["mfx"]` = the wide multifrontal kernel's five size classes together (interior
    nodes; its boundary nodes' list is `mfx_boundary`)."""

    def __missing__(self, key):
        if key == "mfx":
            return sum(v for k, v in self.items() if k.startswith("mfx_") and k != "mfx_boundary")
        raise KeyError(key)


class Grid:
    """Grid(dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed, connectivity,
    element_types, logging=False, build_edges=False)  -- grid.pyx:47-53.

    Unlike the reference, the point coordinates are part of construction (`coords=`) and everything
    -- connectivity, centroids, normals -- is built in one native call; `build()`,
    `load_point_coords()` etc. are therefore no-ops kept for call-compatibility.
    """

    def __init__(self, dim, n_elems, n_points, npoel, nfael, lnofa, lpofa, nedel, lpoed, connectivity,
                 element_types, logging=False, build_edges=False, coords=None, num_threads=0, build_device=None):
        # grid.pyx:55-60
        if dim < 1:
            raise ValueError("The number of dimensions must be greater than 0.")
        if n_elems < 1:
            raise ValueError("The number of elements must be greater than 0.")
        if n_points < 1:
            raise ValueError("The number of points must be greater than 0.")
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        tabs = [i64(npoel), i64(nfael), i64(lnofa), i64(lpofa), i64(nedel), i64(lpoed)]
        expected = [(T.NUM_ELEMENT_TYPES,), (T.NUM_ELEMENT_TYPES,),
                    (T.NUM_ELEMENT_TYPES, T.MAX_FACES_PER_ELEMENT),
                    (T.NUM_ELEMENT_TYPES, T.MAX_FACES_PER_ELEMENT, T.MAX_POINTS_PER_FACE),
                    (T.NUM_ELEMENT_TYPES,), (T.NUM_ELEMENT_TYPES, T.MAX_EDGES_PER_ELEMENT, T.MAX_POINTS_PER_EDGE)]
        for a, shp in zip(tabs, expected):  # grid.pyx:80-101 _validate_shape
            if a.shape != shp:
                raise ValueError(f"The array must have shape {shp}, not {a.shape}.")
        conn = i64(connectivity)
        if conn.shape != (n_elems, T.MAX_POINTS_PER_ELEMENT):
            raise ValueError(f"The array must have shape {(n_elems, T.MAX_POINTS_PER_ELEMENT)}, not {conn.shape}.")
        etypes = i64(element_types)
        if coords is None:
            raise ValueError("The point coordinates have not been set.")
        xyz = np.ascontiguousarray(coords, dtype=np.float64)
        if xyz.ndim != 2 or xyz.shape[0] != n_points or not 1 <= xyz.shape[1] <= 3:
            raise ValueError(f"coords must have shape ({n_points}, 1..3), not {xyz.shape}.")
        self.logging = bool(logging)
        self.build_edges = bool(build_edges)
        self._cache = {}
        self._perm_key = None   # fingerprint of the permeability table resident on the device (interpolator.py)
        self._fields_variable = None   # the variable whose Neumann flags are resident (DevicePlan.ensure_current)
        self._h = ctypes.c_void_p()
        L = _lib.load()
        if build_device is None:   # the native OpenMP builder (csrc/grid_host.cpp)
            rc = L.nin_grid_create(int(dim), int(n_elems), int(n_points), *[_ptr(a) for a in tabs], _ptr(conn),
                                   _ptr(etypes), _ptr(xyz), int(xyz.shape[1]), int(bool(build_edges)),
                                   int(num_threads), ctypes.byref(self._h))
        else:                      # the same arrays built by HIP kernels on that GPU (csrc/grid_device.hip)
            rc = L.nin_grid_create_on_device(int(dim), int(n_elems), int(n_points), *[_ptr(a) for a in tabs], _ptr(conn),
                                             _ptr(etypes), _ptr(xyz), int(xyz.shape[1]), int(bool(build_edges)),
                                             int(build_device), ctypes.byref(self._h))
        if rc == _lib.NIN_EINVAL:
            raise ValueError(L.nin_last_error().decode())
        _lib.check(rc)
        self._coords_dim = int(xyz.shape[1])
        self.are_elements_loaded = True
        self.are_coords_loaded = True
        self.are_structures_built = True
        self.are_centroids_calculated = True
        self.are_normals_calculated = True

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.load().nin_grid_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- reference method names, kept so reference-style call sequences run unchanged -------------
    def build(self):
        return None

    def load_point_coords(self, coords=None):
        return None

    def calculate_centroids(self):
        return None

    def calculate_normal_faces(self):
        return None

    # -- attributes --------------------------------------------------------------------------------
    def _scalar(self, name):
        return int(_lib.load().nin_grid_scalar(self._h, name.encode()))

    def _array(self, name):
        if name in self._cache:
            return self._cache[name]
        L = _lib.load()
        n, dt = ctypes.c_int64(), ctypes.c_int()
        _lib.check(L.nin_grid_array_info(self._h, name.encode(), ctypes.byref(n), ctypes.byref(dt)))
        a = np.empty(n.value, dtype=np.float64 if dt.value == 1 else np.int64)
        _lib.check(L.nin_grid_array_copy(self._h, name.encode(), _ptr(a), n.value))
        if name in _SHAPES:
            a = a.reshape(-1, _SHAPES[name])
            if name == "point_coords" and self._coords_dim != 3:
                a = np.ascontiguousarray(a[:, :self._coords_dim])   # grid.pyx:666 keeps the caller's width
        self._cache[name] = a
        return a

    @property
    def nnz_esup(self):
        """len(esup) without bringing the array to the host (a device-built grid mirrors arrays lazily)."""
        return self._scalar("nnz_esup")

    def __getattr__(self, name):
        if name in _SCALARS:
            return self._scalar(name)
        if name in _ARRAYS:
            return self._array(name)
        raise AttributeError(name)

    def get_data(self):
        """grid.pyx:583-658, including its quirk: with build_edges=False the reference dies on
        `self.inpoed.copy()` of an empty (0, 0) view with this ValueError."""
        if not self.build_edges:
            raise ValueError("Invalid shape in axis 0: 0.")
        data = {k: getattr(self, k) for k in ("n_elems", "n_points", "n_faces", "n_edges", "MX_ELEMENTS_PER_POINT",
                                              "MX_POINTS_PER_POINT", "MX_ELEMENTS_PER_FACE", "MX_FACES_PER_POINT")}
        for k in ("point_coords", "centroids", "normal_faces", "faces_centers", "faces_areas", "boundary_faces",
                  "boundary_points", "inpoel", "element_types", "inpofa", "infael", "inpoed", "inedel"):
            data[k] = getattr(self, k).copy()

        def dense(ptr, idx, n_rows, width):
            out = -np.ones((n_rows, width), dtype=np.int64)
            cnt = np.diff(ptr)
            rows = np.repeat(np.arange(n_rows), cnt)
            cols = np.arange(len(idx)) - np.repeat(ptr[:-1], cnt)
            out[rows, cols] = idx
            return out
        data["esup"] = dense(self.esup_ptr, self.esup, self.n_points, self.MX_ELEMENTS_PER_POINT)
        data["psup"] = dense(self.psup_ptr, self.psup, self.n_points, self.MX_POINTS_PER_POINT)
        data["esuf"] = dense(self.esuf_ptr, self.esuf, self.n_faces, self.MX_ELEMENTS_PER_FACE)
        data["fsup"] = dense(self.fsup_ptr, self.fsup, self.n_points, self.MX_FACES_PER_POINT)
        return data

    # -- device ------------------------------------------------------------------------------------
    def to_device(self, device=0):
        self._perm_key = None   # a fresh device copy holds no fields
        self._fields_variable = None   # ... and nobody's Neumann flags: every DevicePlan re-uploads at its next launch
        _lib.check(_lib.load().nin_grid_to_device(self._h, int(device)))
        return self

    @property
    def device(self):
        return int(_lib.load().nin_grid_device(self._h))

    def release_scratch(self):
        """Free the device buffers interpolate() / apply() keep between calls (nin_grid_release_scratch: ~2.3 GB of HBM
        at 10 M cells); the next call allocates them again."""
        _lib.check(_lib.load().nin_grid_release_scratch(self._h))

    PLAN_KERNELS = ("block1", "block2", "block4", "block8", "scratch", "hex8", "mfw_large", "mfw_small", "mfw_general", "small4", "small8",
                    "small12", "quad4", "mfx_6x10", "mfx_7x11", "mfx_8x13", "mfx_9x15", "mfx_10x16", "mfx_boundary", "mfg_tiles", "mfx_4x7", "mfx_7x12")

    def gls_plan_flops(self):
        """Per kernel of the GLS launch plan (nin_gls_plan_flops): {kernel: (algorithmic flops, reference-equivalent dgels flops,
        nodes computed)} for one launch over all nodes; needs the fields on the device (a DevicePlan or an interpolate() first)."""
        alg, ref, comp = np.zeros(22), np.zeros(22), np.zeros(22, dtype=np.int64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(_lib.load().nin_gls_plan_flops(self._h, p(alg), p(ref), p(comp)))
        return {k: (float(alg[i]), float(ref[i]), int(comp[i])) for i, k in enumerate(self.PLAN_KERNELS)}

    def gls_plan(self):
        """Nodes per GLS kernel of the device copy (nin_gls_plan): block kernel classes 1 / 2 / 4 / 8 wavefronts per
        node and global scratch, the cube-node kernel, the one-wavefront multifrontal kernel (two-coloured nodes large / small, general kind), the one-wavefront dense
        kernel for small nodes (at most 4 / 8 / 12 cells), the two-lanes-per-node kernel for the nodes inside a boundary
        face of a hexahedron mesh, the wide one-wavefront multifrontal kernel (interior nodes of unstructured meshes)."""
        counts = np.zeros(22, dtype=np.int64)
        _lib.check(_lib.load().nin_gls_plan(self._h, counts.ctypes.data_as(ctypes.c_void_p)))
        return _PlanCounts(zip(self.PLAN_KERNELS, counts.tolist()))
