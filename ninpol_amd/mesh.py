"""Mesh containers and synthetic mesh generators.

`Interpolator.load_mesh(mesh_obj=...)` is duck-typed exactly as in the reference
(`interpolator.pyx:258,291-294,333-361,217-227,433-447,454`): anything with `.points`,
`.cells` (blocks with `.type` / `.data`), `.cell_data`, `.cell_data_dict`, `.point_data` works --
a real `meshio.Mesh` included.  `Mesh` / `CellBlock` below are the minimal containers of that
shape used by our own generators (meshio is not a dependency of this package).

The reference ships no mesh files (`tests/mesh/.gitkeep`), so the generators here are ours:
structured hexahedra, the Kuhn 6-tetrahedra split, wedges, and a conforming hex | pyramid+tet |
tet mix.  Node numbering is x-fastest, z-slowest so a contiguous node block is a z-slab.
"""
import numpy as np


class CellBlock:
    def __init__(self, type, data):
        self.type = type
        self.data = np.ascontiguousarray(data)

    def __len__(self):
        return len(self.data)

    def __repr__(self):
        return f"<CellBlock {self.type} x{len(self.data)}>"


class Mesh:
    """points (P, 3); cells: list of CellBlock; point_data {name: (P,) array};
    cell_data {name: [array per block]} -- the meshio.Mesh layout."""

    def __init__(self, points, cells, point_data=None, cell_data=None):
        self.points = np.asarray(points)
        self.cells = [c if hasattr(c, "type") else CellBlock(*c) for c in cells]
        self.point_data = dict(point_data or {})
        self.cell_data = dict(cell_data or {})

    @property
    def cell_data_dict(self):
        return {k: {cb.type: a for cb, a in zip(self.cells, v)} for k, v in self.cell_data.items()}

    @property
    def n_cells(self):
        return sum(len(c) for c in self.cells)


# ------------------------------------------------------------------------------------------------
# structured generators
# ------------------------------------------------------------------------------------------------

def _lattice_points(nx, ny, nz, lengths, origin, jitter, seed, k0=0, k1=None):
    """Lattice nodes of planes k0..k1 (inclusive) of the global (nx, ny, nz) box, x fastest.

    Jitter is drawn per z-plane from default_rng([seed, k]) so that any z-slab of the mesh can be
    generated on its own (one slab per GPU rank) and agrees with the whole mesh on shared planes."""
    k1 = nz if k1 is None else k1
    x = np.linspace(0.0, lengths[0], nx + 1) + origin[0]
    y = np.linspace(0.0, lengths[1], ny + 1) + origin[1]
    z = np.linspace(0.0, lengths[2], nz + 1)[k0:k1 + 1] + origin[2]
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if jitter:
        h = np.array([lengths[0] / nx, lengths[1] / ny, lengths[2] / nz])
        npl = (nx + 1) * (ny + 1)
        I = np.arange(npl)
        i, j = I % (nx + 1), I // (nx + 1)
        for k in range(k0, k1 + 1):
            d = np.random.default_rng([seed, k]).uniform(-jitter, jitter, size=(npl, 3)) * h
            # keep the box: boundary nodes only move inside their boundary plane
            d[(i == 0) | (i == nx), 0] = 0.0
            d[(j == 0) | (j == ny), 1] = 0.0
            if k == 0 or k == nz:
                d[:, 2] = 0.0
            pts[(k - k0) * npl:(k - k0 + 1) * npl] += d
    return np.ascontiguousarray(pts)


def _hex_corner_ids(nx, ny, nz, i0=0, i1=None):
    """(E, 8) node ids of the lattice cells i0 <= i < i1 in meshio hexahedron order:
    0=(i,j,k) 1=(i+1,j,k) 2=(i+1,j+1,k) 3=(i,j+1,k), 4..7 the same at k+1.  Cell order x-fastest."""
    if i1 is None:
        i1 = nx
    sx, sy = nx + 1, (nx + 1) * (ny + 1)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(i0, i1), indexing="ij")
    n0 = (i + j * sx + k * sy).ravel().astype(np.int64)
    off = np.array([0, 1, 1 + sx, sx, sy, sy + 1, sy + 1 + sx, sy + sx], dtype=np.int64)
    return n0[:, None] + off[None, :]


def hex_mesh(nx, ny=None, nz=None, lengths=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
             jitter=0.0, seed=0):
    """nx*ny*nz hexahedra on a box."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    pts = _lattice_points(nx, ny, nz, lengths, origin, jitter, seed)
    return Mesh(pts, [CellBlock("hexahedron", _hex_corner_ids(nx, ny, nz))])


def hex_slab(nx, ny, nz, plane_lo, plane_hi, lengths=(1.0, 1.0, 1.0), jitter=0.0, seed=0):
    """The part of hex_mesh(nx, ny, nz) a rank owning node planes [plane_lo, plane_hi) needs: those
    planes plus one halo plane on each interior side, and every cell between them.  Returns
    (mesh, node_offset, cell_offset, owned_lo, owned_hi): local ids + offset = global ids, and the
    owned nodes are the local range [owned_lo, owned_hi).  Nodes partition by contiguous index
    blocks with the neighbouring cells replicated (north_star)."""
    k0 = max(plane_lo - 1, 0)
    k1 = min(plane_hi, nz)            # last node plane kept (inclusive)
    pts = _lattice_points(nx, ny, nz, lengths, (0.0, 0.0, 0.0), jitter, seed, k0, k1)
    cells = _hex_corner_ids(nx, ny, k1 - k0)
    npl = (nx + 1) * (ny + 1)
    mesh = Mesh(pts, [CellBlock("hexahedron", cells)])
    return mesh, k0 * npl, k0 * nx * ny, (plane_lo - k0) * npl, (plane_hi - k0) * npl


# the six Kuhn simplices of the unit cube as paths 0 -> 6 through hexahedron-local corners,
# each re-ordered to positive orientation
_KUHN = np.array([
    [0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6], [0, 5, 1, 6],
], dtype=np.int64)


def _kuhn_from_hex(hexes):
    return hexes[:, _KUHN].reshape(-1, 4)


def tet_mesh(nx, ny=None, nz=None, lengths=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
             jitter=0.0, seed=0):
    """Kuhn split: 6 tetrahedra per lattice cell around the 0-6 diagonal (conforming)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    pts = _lattice_points(nx, ny, nz, lengths, origin, jitter, seed)
    return Mesh(pts, [CellBlock("tetra", _kuhn_from_hex(_hex_corner_ids(nx, ny, nz)))])


def wedge_mesh(nx, ny=None, nz=None, lengths=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0),
               jitter=0.0, seed=0):
    """2 wedges per lattice cell, split along the 0-2 diagonal of the bottom quad."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    pts = _lattice_points(nx, ny, nz, lengths, origin, jitter, seed)
    h = _hex_corner_ids(nx, ny, nz)
    w = np.stack([h[:, [0, 1, 2, 4, 5, 6]], h[:, [0, 2, 3, 4, 6, 7]]], axis=1).reshape(-1, 6)
    return Mesh(pts, [CellBlock("wedge", w)])


def wedge_fan(n_sectors, n_layers=2, jitter=0.0, seed=0):
    """A disc of `n_sectors` triangles around its centre, extruded into `n_layers` layers of wedges: the interior
    nodes on the axis have 2 * n_sectors cells and 5 * n_sectors faces around them -- the high-degree nodes that no
    lattice generator produces (wide GLS systems: column slots 3-4, the global-scratch class)."""
    rng = np.random.default_rng(seed)
    th = 2.0 * np.pi * np.arange(n_sectors) / n_sectors
    pts = []
    for k in range(n_layers + 1):
        z = k / n_layers
        pts.append([0.0, 0.0, z])
        pts.extend([[np.cos(t), np.sin(t), z] for t in th])
    pts = np.asarray(pts, dtype=float)
    if jitter:
        pts[:, :2] += rng.uniform(-jitter, jitter, size=(len(pts), 2)) * (np.abs(pts[:, :1]) + np.abs(pts[:, 1:2]) > 0)
    per = n_sectors + 1
    w = []
    for k in range(n_layers):
        lo, hi = k * per, (k + 1) * per
        for i in range(n_sectors):
            a, b = 1 + i, 1 + (i + 1) % n_sectors
            w.append([lo, lo + a, lo + b, hi, hi + a, hi + b])   # bottom triangle counter-clockwise from above
    return Mesh(pts, [CellBlock("wedge", np.asarray(w, dtype=np.int64))])


def mixed_mesh(nx, ny=None, nz=None, n_hex=None, lengths=(1.0, 1.0, 1.0), jitter=0.0, seed=0):
    """Conforming hex | transition | tet mesh along x.

    Lattice columns i < n_hex stay hexahedra; column i == n_hex is the transition layer: each cell
    gets a centre node and six pyramids (quad faces meet the hexahedra and each other), and the
    pyramid on the +x face is cut into two tetrahedra along the diagonal the Kuhn split uses there;
    columns i > n_hex are Kuhn tetrahedra.  Blocks are [hexahedron, pyramid, tetra] (meshio order of
    appearance), so the global cell numbering is the concatenation in that order.
    """
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    n_hex = nx // 2 if n_hex is None else n_hex
    assert 0 < n_hex < nx - 1
    pts = _lattice_points(nx, ny, nz, lengths, (0.0, 0.0, 0.0), jitter, seed)
    hexes = _hex_corner_ids(nx, ny, nz, 0, n_hex)
    trans = _hex_corner_ids(nx, ny, nz, n_hex, n_hex + 1)
    tets_src = _hex_corner_ids(nx, ny, nz, n_hex + 1, nx)
    centres = pts[trans].mean(axis=1)
    c_id = pts.shape[0] + np.arange(trans.shape[0], dtype=np.int64)
    pts = np.vstack([pts, centres])
    # pyramid base = the hexahedron face reversed (so its own base face [0,3,2,1] points outward)
    hex_faces = np.array([[0, 3, 2, 1], [4, 5, 6, 7], [0, 1, 5, 4], [1, 2, 6, 5],
                          [2, 3, 7, 6], [3, 0, 4, 7]])
    pyr, tet_extra = [], []
    for f in hex_faces:
        base = trans[:, f[::-1]]
        if set(f.tolist()) == {1, 2, 6, 5}:          # the +x face: two tetrahedra, diagonal 1-6
            # base (reversed face) = [5, 6, 2, 1]; Kuhn cuts the quad {1,2,6,5} along 1-6
            q = trans
            tet_extra.append(np.stack([q[:, 1], q[:, 6], q[:, 2], c_id], axis=1))
            tet_extra.append(np.stack([q[:, 1], q[:, 5], q[:, 6], c_id], axis=1))
        else:
            pyr.append(np.concatenate([base, c_id[:, None]], axis=1))
    pyr = np.stack(pyr, axis=1).reshape(-1, 5)
    tet_extra = np.stack(tet_extra, axis=1).reshape(-1, 4)
    tets = np.vstack([tet_extra, _kuhn_from_hex(tets_src)])
    tets = _fix_tet_orientation(pts, tets)
    return Mesh(pts, [CellBlock("hexahedron", hexes), CellBlock("pyramid", pyr),
                      CellBlock("tetra", tets)])


def delaunay_tet_mesh(n, jitter=0.25, seed=0, lattice="bcc", renumber=True):
    """UNSTRUCTURED tetrahedra: the Delaunay tetrahedrisation (scipy.spatial.Delaunay = Qhull) of a jittered point cloud
    in the unit box -- the mesh class the reference's tetra / prism / misc numbers are taken on (gmsh-type meshes,
    tests/results/yaml/performance.yaml:184-246: ~23 cells per node); the reference ships no mesh files, so this generator
    stands in for them.  Node degree is irregular: 2 V - 4 cells around an interior node with V neighbours -- 14 .. 40, mean 25
    for the default cloud (Kuhn meshes: always 24; no two-colouring of the cells either).

    lattice = "bcc": the corners and the centres of an n^3 lattice (2 n^3 + ... points, ~12.5 n^3 cells), each moved by
              U(-jitter, jitter) * h -- slivers stay rare (smallest cell volume ~1e-3 h^3 at jitter 0.25);
              "random": the lattice corners on the boundary + (n - 1)^3 + n^3 uniform random interior points (a wider degree
              distribution: up to ~50 cells per node, and worse cells).
    Boundary points stay in their boundary plane (corners fixed), so the hull is the box.  renumber: nodes in (z, y, x) order of
    their lattice cell and cells in the order of their lowest node, so that contiguous node blocks are z-slabs as in the
    structured generators (Qhull's own order is kept otherwise)."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    h = 1.0 / n
    g = np.arange(n + 1) * h
    Z, Y, X = np.meshgrid(g, g, g, indexing="ij")
    corners = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    c = (np.arange(n) + 0.5) * h
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    centres = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if lattice == "bcc":
        pts = np.vstack([corners, centres])
        d = rng.uniform(-jitter, jitter, size=pts.shape) * h
    elif lattice == "random":
        on_b = np.any((corners < 1e-12) | (corners > 1 - 1e-12), axis=1)
        n_int = (n - 1) ** 3 + n ** 3
        pts = np.vstack([corners[on_b], rng.uniform(0.2 * h, 1.0 - 0.2 * h, size=(n_int, 3))])
        d = np.zeros_like(pts)
        d[:int(on_b.sum())] = rng.uniform(-jitter, jitter, size=(int(on_b.sum()), 3)) * h
    else:
        raise ValueError("lattice must be 'bcc' or 'random'")
    for a in range(3):                                   # boundary points only move inside their boundary plane
        on = (np.abs(pts[:, a]) < 1e-12) | (np.abs(pts[:, a] - 1.0) < 1e-12)
        d[on, a] = 0.0
    pts = pts + d
    tets = Delaunay(pts).simplices.astype(np.int64)
    if renumber:
        ijk = np.clip(np.floor(pts / h - 1e-9).astype(np.int64), 0, n - 1)
        order = np.lexsort((pts[:, 0], ijk[:, 0], ijk[:, 1], ijk[:, 2]))
        new_id = np.empty(len(pts), dtype=np.int64)
        new_id[order] = np.arange(len(pts))
        pts = pts[order]
        tets = new_id[tets]
        tets = tets[np.lexsort((tets.max(axis=1), tets.min(axis=1)))]
    tets = _fix_tet_orientation(pts, tets)
    return Mesh(np.ascontiguousarray(pts), [CellBlock("tetra", tets)])


def delaunay_wedge_mesh(n, layers=None, jitter=0.3, seed=0, lattice="grid"):
    """UNSTRUCTURED prisms: the Delaunay triangulation (scipy / Qhull) of a jittered point cloud in the unit square, extruded into
    `layers` layers of wedges (default n) -- the "prism" mesh class of the reference's numbers (performance.yaml; it ships no mesh
    files).  A node inside has 2 V wedges around it, V = its valence in the triangulation: 4 .. 9 (the structured wedge_mesh: always
    6), and the ring of wedges is two-coloured only where V is even.

    lattice = "grid": the (n + 1)^2 lattice points, each moved by U(-jitter, jitter) * h; "random": the lattice points on the
    boundary + (n - 1)^2 uniform random points inside.  Boundary points stay on their boundary line.  Nodes are numbered layer by
    layer ((z, y, x) order of their lattice cell), cells layer by layer, bottom triangle counter-clockwise seen from above."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    layers = n if layers is None else layers
    h = 1.0 / n
    g = np.arange(n + 1) * h
    Y, X = np.meshgrid(g, g, indexing="ij")
    p2 = np.stack([X.ravel(), Y.ravel()], axis=1)
    on_b = np.any((p2 < 1e-12) | (p2 > 1 - 1e-12), axis=1)
    if lattice == "grid":
        d = rng.uniform(-jitter, jitter, size=p2.shape) * h
    elif lattice == "random":
        p2 = np.vstack([p2[on_b], rng.uniform(0.2 * h, 1.0 - 0.2 * h, size=((n - 1) ** 2, 2))])
        on_b = np.arange(len(p2)) < int(on_b.sum())
        d = np.zeros_like(p2)
        d[on_b] = rng.uniform(-jitter, jitter, size=(int(on_b.sum()), 2)) * h
    else:
        raise ValueError("lattice must be 'grid' or 'random'")
    for a in range(2):
        on = (np.abs(p2[:, a]) < 1e-12) | (np.abs(p2[:, a] - 1.0) < 1e-12)
        d[on, a] = 0.0
    p2 = p2 + d
    ij = np.clip(np.floor(p2 / h - 1e-9).astype(np.int64), 0, n - 1)
    order = np.lexsort((p2[:, 0], ij[:, 0], ij[:, 1]))
    p2 = p2[order]
    tri = Delaunay(p2).simplices.astype(np.int64)
    a, b, c = p2[tri[:, 0]], p2[tri[:, 1]], p2[tri[:, 2]]
    cw = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0]) < 0
    tri[cw] = tri[cw][:, [0, 2, 1]]
    tri = tri[np.lexsort((tri.max(axis=1), tri.min(axis=1)))]
    per = len(p2)
    z = np.arange(layers + 1) / layers
    pts = np.concatenate([np.column_stack([p2, np.full(per, zk)]) for zk in z])
    w = np.concatenate([np.hstack([tri + k * per, tri + (k + 1) * per]) for k in range(layers)])
    return Mesh(np.ascontiguousarray(pts), [CellBlock("wedge", w.astype(np.int64))])


def quad_tri_mesh_2d(nx, ny=None, jitter=0.0, seed=0):
    """2-D mesh on the unit square: left half quads, right half triangles (each lattice cell cut along its
    0-2 diagonal).  Points are (P, 3) with z = 0, as meshio delivers 2-D meshes."""
    ny = nx if ny is None else ny
    pts = _lattice_points(nx, ny, 1, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0), jitter, seed, 0, 0)
    pts[:, 2] = 0.0
    sx = nx + 1
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    n0 = (i + j * sx).ravel().astype(np.int64)
    q = np.stack([n0, n0 + 1, n0 + 1 + sx, n0 + sx], axis=1)
    left = (i.ravel() < nx // 2)
    quads = q[left]
    qt = q[~left]
    tris = np.stack([qt[:, [0, 1, 2]], qt[:, [0, 2, 3]]], axis=1).reshape(-1, 3)
    return Mesh(pts, [CellBlock("quad", quads), CellBlock("triangle", tris)])


def _fix_tet_orientation(pts, tets):
    a, b, c, d = (pts[tets[:, i]] for i in range(4))
    vol = np.einsum("ij,ij->i", np.cross(b - a, c - a), d - a)
    neg = vol < 0
    tets = tets.copy()
    tets[neg, 1], tets[neg, 2] = tets[neg, 2].copy(), tets[neg, 1].copy()
    return tets


# ------------------------------------------------------------------------------------------------
# synthetic fields (the shapes the reference's analytical cases use, tests/utils/analytical.py)
# ------------------------------------------------------------------------------------------------

def cell_centroids(mesh):
    """vertex mean per cell, block order (what analytical.py:142 evaluates K at)."""
    return np.vstack([mesh.points[c.data].mean(axis=1) for c in mesh.cells])


def permeability_alh(centroids):
    """Heterogeneous SPD tensor of the ALH case (tests/utils/analytical.py:303-323)."""
    x, y, z = centroids[:, 0], centroids[:, 1], centroids[:, 2]
    K = np.zeros((len(x), 3, 3))
    K[:, 0, 0] = y ** 2 + z ** 2 + 1
    K[:, 0, 1] = -x * y
    K[:, 0, 2] = -x * z
    K[:, 1, 0] = -y * x
    K[:, 1, 1] = x ** 2 + z ** 2 + 1
    K[:, 1, 2] = -y * z
    K[:, 2, 0] = -z * x
    K[:, 2, 1] = -z * y
    K[:, 2, 2] = x ** 2 + y ** 2 + 1
    return K


def permeability_const(n, kind="LIN"):
    """Constant tensors of the LIN/QUAD and FAN cases (analytical.py:253-296)."""
    if kind == "FAN":
        Ku = np.array([[2464.36, 0.0, 1148.68], [0.0, 536.64, 0.0], [1148.68, 0.0, 536.64]])
    else:
        Ku = np.array([[1.0, 0.5, 0.0], [0.5, 1.0, 0.5], [0.0, 0.5, 1.0]])
    K = np.zeros((n, 3, 3))
    K[:] = Ku
    return K


def attach_fields(mesh, variable="u", perm="ALH", neumann_plane=None, seed=1, values=None):
    """Attach `permeability`, a scalar cell variable and the Neumann point arrays to `mesh`.

    neumann_plane: None (all-Dirichlet boundary, flags 0) or (axis, coordinate): nodes with
    |x[axis] - coordinate| < 1e-12 get flag 1 and a U(0,1) Neumann value (default_rng(seed)).
    """
    cen = cell_centroids(mesh)
    E = cen.shape[0]
    K = permeability_alh(cen) if perm == "ALH" else permeability_const(E, perm)
    K = K.reshape(E, 9)
    if values is None:
        values = cen[:, 0] + cen[:, 1] + cen[:, 2]
    P = mesh.points.shape[0]
    flag = np.zeros(P)
    val = np.zeros(P)
    if neumann_plane is not None:
        axis, coord = neumann_plane
        on = np.abs(mesh.points[:, axis] - coord) < 1e-12
        flag[on] = 1.0
        val[on] = np.random.default_rng(seed).uniform(0.0, 1.0, int(on.sum()))
    sizes = np.cumsum([0] + [len(c) for c in mesh.cells])
    split = lambda a: [a[sizes[b]:sizes[b + 1]] for b in range(len(mesh.cells))]
    mesh.cell_data = {"permeability": split(K), variable: split(np.asarray(values, dtype=float))}
    mesh.point_data = {"neumann_flag_" + variable: flag, "neumann_" + variable: val}
    return mesh
