"""Legacy VTK files (`.vtk`, DATASET UNSTRUCTURED_GRID) without meshio.

The reference reads mesh FILES through `meshio.read` (`interpolator.pyx:188`) and its own tests produce them with
`meshio.write(<name>.vtk, ...)` (`tests/accuracy_test.py:46`, `tests/performance_test.py:46-49`; read back at
`tests/utils/analytical.py:127`): legacy VTK, version 5.1, binary by default, every point / cell array as a FIELD entry.
meshio is not a dependency of this package (and absent from the build image), so `Interpolator.load_mesh(filename=...)` reads
that format here -- ASCII and BINARY, file versions 2.0 - 4.2 (`CELLS n size` with inline counts) and 5.x (`OFFSETS` /
`CONNECTIVITY`), data sections SCALARS / VECTORS / NORMALS / TENSORS / FIELD -- into the same `Mesh` / `CellBlock` shape a
`meshio.Mesh` has: consecutive cells of one type form a block, `cell_data[name]` is a list of per-block arrays.
`write()` produces the file meshio's writer produces for the same mesh (version 5.1, FIELD arrays), for the tests and tools.

Parity: the format is VTK's published one; meshio itself is absent here, so files WRITTEN BY meshio have not been read by this
code in this repository -- "parity unpinned" against meshio's writer, pinned against the specification by round trips.
"""
import numpy as np

from .mesh import CellBlock, Mesh

# VTK cell type ids <-> meshio names (linear cells: node order identical in both)
VTK_TO_NAME = {1: "vertex", 3: "line", 5: "triangle", 9: "quad", 10: "tetra", 12: "hexahedron", 13: "wedge", 14: "pyramid"}
NAME_TO_VTK = {v: k for k, v in VTK_TO_NAME.items()}
_NODES = {"vertex": 1, "line": 2, "triangle": 3, "quad": 4, "tetra": 4, "hexahedron": 8, "wedge": 6, "pyramid": 5}
_DTYPES = {"bit": "u1", "unsigned_char": "u1", "char": "i1", "unsigned_short": "u2", "short": "i2", "unsigned_int": "u4",
           "int": "i4", "unsigned_long": "u8", "long": "i8", "float": "f4", "double": "f8", "vtktypeint32": "i4",
           "vtktypeint64": "i8", "vtktypeuint8": "u1", "vtkidtype": "i8"}
_NAMES = {"f8": "double", "f4": "float", "i4": "int", "i8": "vtktypeint64", "u1": "unsigned_char"}


class _Reader:
    """Token / block access to a legacy VTK file held in memory (ASCII numbers and big-endian binary blocks)."""

    def __init__(self, raw):
        self.raw = raw
        self.pos = 0
        self.binary = False

    def line(self):
        """Next non-empty line, stripped (None at the end of the file)."""
        while self.pos < len(self.raw):
            end = self.raw.find(b"\n", self.pos)
            end = len(self.raw) if end < 0 else end
            ln = self.raw[self.pos:end].strip()
            self.pos = end + 1
            if ln:
                return ln.decode("ascii", errors="replace")
        return None

    def peek(self):
        p = self.pos
        ln = self.line()
        self.pos = p
        return ln

    def skip_metadata(self):
        """The body of a METADATA block (VTK / ParaView 9 write one behind an array: INFORMATION <n>, NAME / DATA lines):
        everything up to the blank line that ends it.  The keyword line itself has been consumed."""
        while self.pos < len(self.raw):
            end = self.raw.find(b"\n", self.pos)
            end = len(self.raw) if end < 0 else end
            blank = not self.raw[self.pos:end].strip()
            self.pos = end + 1
            if blank:
                break

    def skip_metadata_if_next(self):
        nxt = self.peek()
        if nxt is not None and nxt.split()[0].upper() == "METADATA":
            self.line()
            self.skip_metadata()

    def array(self, count, type_name):
        dt = _DTYPES.get(type_name.lower())
        if dt is None:
            raise ValueError(f"legacy VTK: unknown data type {type_name!r}")
        if self.binary:
            nbytes = count * np.dtype(dt).itemsize
            if self.pos + nbytes > len(self.raw):
                raise ValueError("legacy VTK: file ends inside a binary block")
            a = np.frombuffer(self.raw, dtype=np.dtype(dt).newbyteorder(">"), count=count, offset=self.pos)
            self.pos += nbytes
            if self.pos < len(self.raw) and self.raw[self.pos:self.pos + 1] == b"\n":
                self.pos += 1
            return a.astype(np.dtype(dt))
        vals = []
        while len(vals) < count:
            ln = self.line()
            if ln is None:
                raise ValueError("legacy VTK: file ends inside an ASCII block")
            vals.extend(ln.split())
        if len(vals) != count:
            raise ValueError("legacy VTK: an ASCII block does not end at a line end")
        return np.array(vals, dtype=np.float64 if dt[0] == "f" else np.int64).astype(np.dtype(dt))


def _read_data_section(r, n, where):
    """The arrays of a POINT_DATA / CELL_DATA section of `n` tuples -> {name: (n,) or (n, k) array}."""
    out = {}
    while True:
        ln = r.peek()
        if ln is None:
            break
        key = ln.split()[0].upper()
        if key in ("POINT_DATA", "CELL_DATA"):
            break
        r.line()
        tok = ln.split()
        if key == "METADATA":            # belongs to the array before it: skipped, the section goes on (what meshio's reader does)
            r.skip_metadata()
        elif key == "SCALARS":
            name, typ, ncomp = tok[1], tok[2], int(tok[3]) if len(tok) > 3 else 1
            nxt = r.peek()
            if nxt is not None and nxt.split()[0].upper() == "LOOKUP_TABLE":
                r.line()
            a = r.array(n * ncomp, typ)
            out[name] = a if ncomp == 1 else a.reshape(n, ncomp)
        elif key in ("VECTORS", "NORMALS"):
            out[tok[1]] = r.array(3 * n, tok[2]).reshape(n, 3)
        elif key == "TENSORS":
            out[tok[1]] = r.array(9 * n, tok[2]).reshape(n, 9)
        elif key == "FIELD":
            for _ in range(int(tok[2])):
                r.skip_metadata_if_next()            # (a METADATA block may sit between two arrays of a FIELD)
                head = r.line().split()
                name, ncomp, ntup, typ = head[0], int(head[1]), int(head[2]), head[3]
                a = r.array(ncomp * ntup, typ)
                if ntup != n:
                    raise ValueError(f"legacy VTK: {where} array {name!r} has {ntup} tuples, expected {n}")
                out[name] = a if ncomp == 1 else a.reshape(ntup, ncomp)
            r.skip_metadata_if_next()
        elif key in ("LOOKUP_TABLE",):
            r.array(4 * int(tok[2]), "unsigned_char" if r.binary else "float")
        elif key in ("COLOR_SCALARS", "TEXTURE_COORDINATES"):
            raise ValueError(f"legacy VTK: {key} attributes are not supported")
        else:
            raise ValueError(f"legacy VTK: unexpected line in {where}: {ln!r}")
    return out


def read(filename):
    """Read a legacy VTK unstructured grid -> ninpol_amd.mesh.Mesh (points (P, 3) float64, CellBlocks by type runs)."""
    with open(filename, "rb") as f:
        raw = f.read()
    r = _Reader(raw)
    head = r.line()
    if head is None or not head.lower().startswith("# vtk datafile version"):
        raise ValueError(f"{filename}: not a legacy VTK file")
    r.line()   # title
    fmt = (r.line() or "").upper()
    if fmt not in ("ASCII", "BINARY"):
        raise ValueError(f"{filename}: expected ASCII or BINARY, found {fmt!r}")
    r.binary = fmt == "BINARY"
    ds = (r.line() or "").split()
    if len(ds) != 2 or ds[0].upper() != "DATASET" or ds[1].upper() != "UNSTRUCTURED_GRID":
        raise ValueError(f"{filename}: only DATASET UNSTRUCTURED_GRID is supported")
    points = offsets = conn = types = None
    point_data, cell_data = {}, {}
    while True:
        ln = r.line()
        if ln is None:
            break
        tok = ln.split()
        key = tok[0].upper()
        if key == "POINTS":
            points = r.array(3 * int(tok[1]), tok[2]).reshape(-1, 3).astype(np.float64)
        elif key == "CELLS":
            a, b = int(tok[1]), int(tok[2])
            nxt = r.peek()
            if nxt is not None and nxt.split()[0].upper() == "OFFSETS":       # version 5.x: a = len(offsets), b = len(connectivity)
                offsets = r.array(a, r.line().split()[1]).astype(np.int64)
                head2 = r.line().split()
                if head2[0].upper() != "CONNECTIVITY":
                    raise ValueError(f"{filename}: CONNECTIVITY expected after OFFSETS")
                conn = r.array(b, head2[1]).astype(np.int64)
            else:                                                              # versions up to 4.2: a cells, b integers (count, ids ...)
                flat = r.array(b, "int").astype(np.int64)
                offsets = np.zeros(a + 1, dtype=np.int64)
                conn_parts, p = [], 0
                for c in range(a):
                    k = int(flat[p])
                    conn_parts.append(flat[p + 1:p + 1 + k])
                    offsets[c + 1] = offsets[c] + k
                    p += 1 + k
                conn = np.concatenate(conn_parts) if conn_parts else np.zeros(0, dtype=np.int64)
        elif key == "CELL_TYPES":
            types = r.array(int(tok[1]), "int").astype(np.int64)
        elif key == "POINT_DATA":
            point_data.update(_read_data_section(r, int(tok[1]), "POINT_DATA"))
        elif key == "CELL_DATA":
            cell_data.update(_read_data_section(r, int(tok[1]), "CELL_DATA"))
        elif key == "METADATA":
            r.skip_metadata()                                                  # (information blocks: skipped up to the blank line)
        elif key == "FIELD":                                                   # a dataset-level field: read and dropped
            for _ in range(int(tok[2])):
                r.skip_metadata_if_next()
                h = r.line().split()
                r.array(int(h[1]) * int(h[2]), h[3])
        else:
            raise ValueError(f"{filename}: unexpected line {ln!r}")
    if points is None or offsets is None or types is None:
        raise ValueError(f"{filename}: POINTS, CELLS and CELL_TYPES are all required")
    n_cells = len(types)
    if len(offsets) != n_cells + 1:
        raise ValueError(f"{filename}: {len(offsets)} offsets for {n_cells} cells")
    # consecutive cells of one type form a block (what meshio does)
    cells, cuts = [], [0]
    for c in range(1, n_cells + 1):
        if c == n_cells or types[c] != types[c - 1]:
            cuts.append(c)
    for b0, b1 in zip(cuts[:-1], cuts[1:]):
        t = int(types[b0])
        if t not in VTK_TO_NAME:
            raise ValueError(f"{filename}: VTK cell type {t} is not supported")
        name = VTK_TO_NAME[t]
        k = _NODES[name]
        if np.any(np.diff(offsets[b0:b1 + 1]) != k):
            raise ValueError(f"{filename}: a {name} cell without {k} nodes")
        cells.append(CellBlock(name, conn[offsets[b0]:offsets[b1]].reshape(b1 - b0, k)))
    cd = {name: [a[b0:b1] for b0, b1 in zip(cuts[:-1], cuts[1:])] for name, a in cell_data.items()}
    return Mesh(points, cells, point_data=point_data, cell_data=cd)


def write(filename, mesh, binary=True, version="5.1"):
    """Write `mesh` (Mesh-shaped: .points, .cells, .point_data, .cell_data) as a legacy VTK file the way meshio does: every data
    array a FIELD entry; version "5.1" (OFFSETS / CONNECTIVITY, int64) or "4.2" (inline counts, int32); BINARY big-endian."""
    pts = np.asarray(mesh.points, dtype=np.float64)
    if pts.shape[1] == 2:
        pts = np.column_stack([pts, np.zeros(len(pts))])
    blocks = [(c.type, np.asarray(c.data, dtype=np.int64)) for c in mesh.cells]
    n_cells = sum(len(d) for _, d in blocks)

    def put(f, a):
        a = np.ascontiguousarray(a)
        if binary:
            f.write(a.astype(a.dtype.newbyteorder(">")).tobytes())
            f.write(b"\n")
        else:
            flat = a.reshape(-1)
            fmtv = (lambda v: repr(float(v))) if a.dtype.kind == "f" else (lambda v: str(int(v)))
            for i in range(0, len(flat), 9):
                f.write((" ".join(fmtv(v) for v in flat[i:i + 9]) + "\n").encode())

    def put_fields(f, data, n):
        f.write(f"FIELD FieldData {len(data)}\n".encode())
        for name, a in data.items():
            a = np.asarray(a)
            a = a.astype(np.float64) if a.dtype.kind == "f" else a.astype(np.int64 if a.dtype.itemsize > 4 else np.int32)
            ncomp = 1 if a.ndim == 1 else int(np.prod(a.shape[1:]))
            if a.shape[0] != n:
                raise ValueError(f"array {name!r} has {a.shape[0]} tuples, expected {n}")
            f.write(f"{name.replace(' ', '_')} {ncomp} {n} {_NAMES[a.dtype.str[1:]]}\n".encode())
            put(f, a)

    with open(filename, "wb") as f:
        f.write(f"# vtk DataFile Version {version}\nwritten by ninpol_amd\n{'BINARY' if binary else 'ASCII'}\nDATASET UNSTRUCTURED_GRID\n".encode())
        f.write(f"POINTS {len(pts)} double\n".encode())
        put(f, pts)
        if version.startswith("5"):
            offsets = np.concatenate([[0], np.cumsum(np.concatenate([np.full(len(d), d.shape[1]) for _, d in blocks]))]).astype(np.int64)
            conn = np.concatenate([d.reshape(-1) for _, d in blocks]).astype(np.int64)
            f.write(f"CELLS {len(offsets)} {len(conn)}\nOFFSETS vtktypeint64\n".encode())
            put(f, offsets)
            f.write(b"CONNECTIVITY vtktypeint64\n")
            put(f, conn)
        else:
            rows = [np.column_stack([np.full(len(d), d.shape[1]), d]).reshape(-1) for _, d in blocks]
            flat = np.concatenate(rows).astype(np.int32)
            f.write(f"CELLS {n_cells} {len(flat)}\n".encode())
            put(f, flat)
        f.write(f"CELL_TYPES {n_cells}\n".encode())
        put(f, np.concatenate([np.full(len(d), NAME_TO_VTK[t]) for t, d in blocks]).astype(np.int32))
        if mesh.point_data:
            f.write(f"POINT_DATA {len(pts)}\n".encode())
            put_fields(f, mesh.point_data, len(pts))
        if mesh.cell_data:
            f.write(f"CELL_DATA {n_cells}\n".encode())
            put_fields(f, {k: np.concatenate([np.asarray(x) for x in v]) for k, v in mesh.cell_data.items()}, n_cells)
