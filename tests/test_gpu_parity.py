"""Parity tests proper: the HIP path (through the C ABI) against the oracle on the same inputs and
against the committed golden fixtures.  Run with `-m gpu` on an MI355X.

Bars (north_star): integers bit-exact; float64 weights within 1e-10 relative (row-scaled, util.py).
IDW and LS are held to a much tighter bar (1e-14 row-relative, NaN rows on the same nodes) since the
kernels follow the reference's operation order with FMA contraction off."""
import numpy as np
import pytest

import util
from ninpol_amd import mesh as M

pytestmark = pytest.mark.gpu

TIGHT = 1e-14


def _interp():
    import ninpol_amd
    return ninpol_amd.Interpolator()


@pytest.mark.parametrize("case", util.GOLDEN_CASES)
def test_gpu_matches_golden(case):
    mesh, z = util.load_golden(case)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    util.assert_grid_equal(I.grid, z)
    for meth in ("idw", "ls", "gls"):
        w, nw = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        tol = util.WEIGHT_RTOL if meth == "gls" else TIGHT
        assert util.rowscaled_err(w, z[f"{meth}_weights"]) <= tol, meth
        assert util.elementwise_err(w, z[f"{meth}_weights"]) <= util.elementwise_rtol(meth, "FAN" if "fan" in case else "ALH"), meth
        assert util.rowscaled_err(nw, z[f"{meth}_neumann_ws"]) <= tol, meth
        W, nws = I.interpolate("u", meth)
        assert W.shape == (I.grid.n_points, I.grid.n_elems)
        err = util.csr_rowscaled_err(W, z[f"{meth}_indptr"], z[f"{meth}_indices"], z[f"{meth}_data"])
        assert err <= tol, (meth, err)


def _meshes():
    yield "hex20_jitter_neu", M.hex_mesh(20, jitter=0.15, seed=0), "ALH", (2, 0.0)
    yield "hex16_uniform_neu", M.hex_mesh(16), "LIN", (0, 1.0)
    yield "tet7_jitter", M.tet_mesh(7, jitter=0.1, seed=1), "ALH", (1, 0.0)
    yield "wedge8", M.wedge_mesh(8, 6, 5, jitter=0.05, seed=2), "ALH", (2, 1.0)
    yield "mixed1266", M.mixed_mesh(12, 6, 6, jitter=0.1, seed=3), "ALH", (2, 1.0)
    yield "delaunay10", M.delaunay_tet_mesh(10, seed=4), "ALH", (2, 0.0)
    yield "delaunay8_random_cloud_fan", M.delaunay_tet_mesh(8, seed=9, lattice="random"), "FAN", (0, 1.0)
    yield "delaunay_prisms14", M.delaunay_wedge_mesh(14, 8, seed=5), "ALH", (2, 0.0)
    yield "delaunay_prisms12_random_fan", M.delaunay_wedge_mesh(12, 6, seed=6, lattice="random"), "FAN", (1, 0.0)


@pytest.mark.parametrize("name,mesh,perm,plane", list(_meshes()), ids=[m[0] for m in _meshes()])
def test_gpu_matches_oracle(oracle_lib, name, mesh, perm, plane):
    M.attach_fields(mesh, "u", perm=perm, neumann_plane=plane, seed=7)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    for k in util.GRID_SCALARS:
        assert getattr(I.grid, k) == getattr(o.grid, k), k
    for k in util.GRID_ARRAYS:
        np.testing.assert_array_equal(getattr(I.grid, k), getattr(o.grid, k), err_msg=k)
    for meth in ("idw", "ls", "gls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        tol = util.WEIGHT_RTOL if meth == "gls" else TIGHT
        assert util.rowscaled_err(w, wo) <= tol, meth
        assert util.rowscaled_err(nw, no) <= tol, meth
        Wo, _ = o.interpolate("u", meth)
        W, _ = I.interpolate("u", meth)
        assert util.csr_rowscaled_err(W, Wo.indptr, Wo.indices, Wo.data) <= tol, meth
        # element-wise relative, every entry down to 1e-3 of its row's largest (util.py)
        ew = max(util.elementwise_err(w, wo), util.csr_elementwise_err(W, Wo.indptr, Wo.indices, Wo.data))
        print(f"{name} {meth}: row-scaled {util.rowscaled_err(w, wo):.2e}, element-wise (floor {util.ELEMENTWISE_FLOOR:g}) {ew:.2e}")
        assert ew <= util.elementwise_rtol(meth, perm), (meth, ew)


def test_gpu_target_subset(oracle_lib):
    mesh = M.hex_mesh(8, jitter=0.1, seed=9)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0))
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    targets = np.array([400, 3, 77, 500, 81], dtype=np.int64)
    for meth in ("idw", "ls", "gls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", targets)
        assert util.rowscaled_err(w, wo[targets]) <= util.WEIGHT_RTOL
        assert util.rowscaled_err(nw, no[targets]) <= util.WEIGHT_RTOL
        W, nws = I.interpolate("u", meth, targets)
        assert W.shape == (len(targets), I.grid.n_elems)


def test_gpu_target_subset_spans_gls_classes(oracle_lib):
    """A target list that hits every GLS launch class of a mixed mesh in ONE call -- cube nodes (the multifrontal
    kernel, which also needs its descriptors built for the list), boundary nodes (one wave per node), pyramid / tet
    transition and Kuhn-tet nodes (2 and 4 waves per node): the per-class device lists share one allocation that must
    outlive all the launches (the bug fixed in 3f9bef5), in an order that is not the node order."""
    mesh = M.mixed_mesh(12, 6, 6, jitter=0.1, seed=6)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=2)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    bp = np.asarray(I.grid.boundary_points).astype(bool)
    rng = np.random.default_rng(1)
    picks = []
    for sel in (~bp & (ne == 8), ~bp & (ne == 24), ~bp & (ne > 8) & (ne < 24), ~bp & (ne > 24), bp):
        ids = np.nonzero(sel)[0]
        assert len(ids) > 0
        picks.append(rng.choice(ids, min(40, len(ids)), replace=False))
    targets = rng.permutation(np.concatenate(picks)).astype(np.int64)
    wo, no = o.prepare("gls", "u")
    for _ in range(3):        # repeated calls reuse / reallocate the list buffer
        w, nw = I.prepare_interpolator("gls", "u", targets)
        assert util.rowscaled_err(w, wo[targets]) <= util.WEIGHT_RTOL
        assert util.rowscaled_err(nw, no[targets]) <= util.WEIGHT_RTOL


def test_gpu_interior_neumann_flags(oracle_lib):
    """The Neumann flag on INTERIOR nodes (gls.pyx:470-472, interpolator.pyx:618): neumann_ws = the last cell's weight,
    added to every stored entry of the row -- through the multifrontal kernel (cube nodes of the hexahedron region) and
    the block kernel (tetrahedron / transition nodes) alike."""
    mesh = M.mixed_mesh(10, 5, 5, jitter=0.1, seed=5)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=6)
    rng = np.random.default_rng(3)
    P = mesh.points.shape[0]
    flagged = rng.choice(P, P // 3, replace=False)
    mesh.point_data["neumann_flag_u"][flagged] = 1.0
    mesh.point_data["neumann_u"][flagged] = rng.uniform(0.0, 1.0, len(flagged))
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    bp = np.asarray(I.grid.boundary_points).astype(bool)
    assert (~bp[flagged]).sum() > 50
    for meth in ("gls", "idw", "ls"):
        W, nws = I.interpolate("u", meth)
        Wo, nwo = o.interpolate("u", meth)
        tol = util.WEIGHT_RTOL if meth == "gls" else 1e-14
        assert util.csr_rowscaled_err(W, Wo.indptr, Wo.indices, Wo.data) <= tol, meth
        assert util.rowscaled_err(nws, nwo) <= tol, meth
        if meth == "gls":    # the interior flagged rows do carry a neumann_ws
            assert np.abs(nws[flagged][~bp[flagged]]).min() > 0.0


def test_gpu_2d_idw_ls(oracle_lib):
    """2-D quad + triangle mesh (z = 0): IDW uses grid.dim coordinates (idw.pyx:66), LS always three."""
    mesh = M.quad_tri_mesh_2d(12, 9, jitter=0.1, seed=1)
    M.attach_fields(mesh, "u", perm="LIN")
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    for meth in ("idw", "ls"):
        wo, _ = o.prepare(meth, "u")
        w, _ = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        assert util.rowscaled_err(w, wo) <= TIGHT, meth


def test_gpu_gls_global_scratch_path(oracle_lib, monkeypatch):
    """Systems too large for LDS run the general kernel on a global-memory slot per wave; forced here
    (NIN_GLS_FORCE_GLOBAL) on a small mixed mesh so that the path is exercised."""
    monkeypatch.setenv("NIN_GLS_FORCE_GLOBAL", "1")
    monkeypatch.setenv("NIN_GLS_NO_GROUP", "1")
    mesh = M.mixed_mesh(6, 4, 4, jitter=0.1, seed=8)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=2)
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    wo, no = o.prepare("gls", "u")
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL


@pytest.mark.parametrize("no_mfw", [True, False])
def test_gpu_generic_kernel_on_hex(oracle_lib, monkeypatch, no_mfw):
    """Hexahedron interior nodes with the cube-node kernel switched off: through the block kernel (both special kernels
    off), and through the small instantiation of the one-wavefront multifrontal kernel (4 fronts + 4 dense cells)."""
    monkeypatch.setenv("NIN_GLS_NO_GROUP", "1")
    if no_mfw:
        monkeypatch.setenv("NIN_GLS_NO_MFW", "1")
        monkeypatch.setenv("NIN_GLS_NO_SMALL", "1")   # (44 rows: the small-node kernel would take them)
    mesh = M.hex_mesh(9, jitter=0.15, seed=5)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(0, 1.0), seed=2)
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    I.grid.to_device(0)
    plan = I.grid.gls_plan()
    assert plan["hex8"] == 0 and plan["mfw_large"] == 0
    assert plan["mfw_small"] == (0 if no_mfw else 8 ** 3)
    wo, no = o.prepare("gls", "u")
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL


def test_gpu_cube_kernel_forms(oracle_lib, monkeypatch):
    """The cube-node kernel against the oracle: two wavefronts per SIMD (blocks a column at a time, fill entries parked in LDS, index
    levels of the next pass by LDS-DMA; round 2's one-wavefront kernel was deleted in round 4).  More groups than resident waves, a
    ragged last group, a Neumann plane, target subsets."""
    mesh = M.hex_mesh(41, 37, 29, jitter=0.15, seed=6)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(1, 0.0), seed=3)
    o = oracle_lib.OracleInterpolator("port", threads=16)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    I.grid.to_device(0)
    assert I.grid.gls_plan()["hex8"] == 40 * 36 * 28
    wo, no = o.prepare("gls", "u")
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL
    rng = np.random.default_rng(0)
    sub = np.sort(rng.choice(I.grid.n_points, size=1237, replace=False)).astype(np.int64)
    ws, nws = I.prepare_interpolator("gls", "u", sub)
    assert np.array_equal(ws, w[sub]) and np.array_equal(nws, nw[sub])


@pytest.mark.parametrize("axis,side", [(0, 0.0), (1, 1.0), (2, 0.0), (2, 1.0)])
def test_gpu_quad_node_kernel(oracle_lib, monkeypatch, axis, side):
    """The nodes inside a boundary face of a hexahedron mesh (4 cells, 4 internal + 4 boundary faces) on a Neumann plane: the
    two-lanes-per-node kernel (kernels_gls_quad4.hip) against the oracle and against the small-node kernel that serves them
    when it is switched off; planes of either orientation and side (which cell of a face is its first differs), a ragged last
    group, Dirichlet quad nodes of the other faces in the same list (zero rows), target subsets."""
    mesh = M.hex_mesh(13, 11, 9, jitter=0.15, seed=4)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(axis, side), seed=6)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    got = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("NIN_GLS_NO_QUAD4", "1")
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        I.grid.to_device(0)
        plan = I.grid.gls_plan()
        assert plan["quad4"] == (0 if off else 2 * (12 * 10 + 12 * 8 + 10 * 8))
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL, off
        assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, off
        got[off] = (w, nw, I)
    flag = np.asarray(mesh.point_data["neumann_flag_u"]).astype(bool)
    ne = np.diff(np.asarray(got[False][2].grid.esup_ptr))
    assert np.count_nonzero(got[False][1][flag & (ne == 4)]) > 50          # the plane's quad nodes carry a Neumann value
    sub = np.sort(np.random.default_rng(1).choice(mesh.points.shape[0], size=317, replace=False)).astype(np.int64)
    ws, nws = got[False][2].prepare_interpolator("gls", "u", sub)
    assert np.array_equal(ws, got[False][0][sub]) and np.array_equal(nws, got[False][1][sub])


def test_gpu_cube_list_in_locality_order(monkeypatch):
    """NIN_GLS_LOCALITY_ORDER: the cube-node kernel's list in strips of mesh rows walked plane by plane (the default), in Morton
    order of the node coordinates ("m"), in node order ("off") -- runs of 16 entries, inside the pieces of interpolate()'s
    pipeline.  A node's arithmetic does not depend on its neighbours in the list: the weights are bit-identical in all three,
    through prepare_interpolator and through the pipelined interpolate()."""
    mesh = M.hex_mesh(30, 26, 22, jitter=0.15, seed=8)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(0, 0.0), seed=3)
    monkeypatch.setenv("NIN_E2E_MIN_NODES", "1024")
    got = {}
    for mode in ("off", None, "m", "s3"):
        if mode:
            monkeypatch.setenv("NIN_GLS_LOCALITY_ORDER", mode)
        else:
            monkeypatch.delenv("NIN_GLS_LOCALITY_ORDER", raising=False)
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        W, neu = I.interpolate("u", "gls")
        got[mode] = (w, nw, W.indptr.copy(), W.indices.copy(), W.data.copy(), neu)
        assert I.grid.gls_plan()["hex8"] == 29 * 25 * 21
    for mode in (None, "m", "s3"):
        for a, b in zip(got["off"], got[mode]):
            assert np.array_equal(a, b), mode
    assert np.count_nonzero(got[None][0]) > 8 * 29 * 25 * 21 - 10


@pytest.mark.parametrize("kind", ["tet", "wedge", "mixed"])
def test_gpu_block_kernel_where_mfw_would_run(oracle_lib, monkeypatch, kind):
    """Interior nodes of tetrahedron / wedge / mixed meshes go to the one-wavefront multifrontal kernel by default; with
    it switched off they take the block kernel (sparse first phase, 2 and 4 wavefronts per node), which still serves every
    node the multifrontal kernels refuse (boundary faces, odd cycles in the cell graph, pyramid apexes) -- and the plan
    says which kernel ran."""
    mesh = {"tet": lambda: M.tet_mesh(5, jitter=0.1, seed=3), "wedge": lambda: M.wedge_mesh(5, jitter=0.05, seed=3),
            "mixed": lambda: M.mixed_mesh(8, 4, 4, jitter=0.1, seed=3)}[kind]()
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=4)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    plans = {}
    for off in (False, True):
        if off:
            monkeypatch.setenv("NIN_GLS_NO_MFW", "1")
            monkeypatch.setenv("NIN_GLS_NO_MFX", "1")
            monkeypatch.setenv("NIN_GLS_NO_SMALL", "1")
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        I.grid.to_device(0)
        plans[off] = I.grid.gls_plan()
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL, off
        assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, off
    assert plans[True]["mfw_large"] == 0 and plans[True]["mfw_small"] == 0 and plans[True]["mfw_general"] == 0 and plans[True]["mfx"] == 0
    taken = plans[False]["mfw_large"] + plans[False]["mfw_small"]
    assert taken == {"tet": 4 ** 3, "wedge": 4 ** 3}.get(kind, taken) and taken > 0
    # the nodes that are not two-coloured (free faces): the hex | pyramid | tet interfaces of the mix -- the wide kernel's by default
    assert (plans[False]["mfx"] > 0) == (kind == "mixed") and plans[False]["mfw_general"] == 0
    assert sum(plans[False].values()) == sum(plans[True].values()) == I.grid.n_points


@pytest.mark.parametrize("kind", ["tet", "wedge"])
def test_gpu_multifrontal_first_form(oracle_lib, monkeypatch, kind):
    """The one-wavefront multifrontal kernel's first form (lane = column, pivot rows in LDS), kept behind
    NIN_MFW_LANE_COLUMNS as the A/B baseline of the row-lane form: still correct."""
    monkeypatch.setenv("NIN_MFW_LANE_COLUMNS", "1")
    mesh = M.tet_mesh(4, jitter=0.1, seed=7) if kind == "tet" else M.wedge_mesh(4, jitter=0.05, seed=7)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=4)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    assert I.grid.gls_plan()["mfw_large" if kind == "tet" else "mfw_small"] == 27
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL


@pytest.mark.parametrize("kind,switch", [("tet", None), ("tet", "NIN_MFW_NO_STRIPS"), ("wedge", None), ("wedge", "NIN_MFW_SMALL_STRIPS"),
                                         ("hex", "NIN_MFW_SMALL_STRIPS"), ("mixed", "NIN_MFW_NO_STRIPS")])
def test_gpu_multifrontal_dense_phase_forms(oracle_lib, monkeypatch, kind, switch):
    """The dense phase of the one-wavefront multifrontal kernel in each of its forms against the oracle: strips (16 x 4 tiles,
    panels of four reflectors, trailing updates on the FP64 matrix unit: the default of the large instantiation), row lanes
    (NIN_MFW_NO_STRIPS: round 2's form, the default of the small instantiation), strips in the small instantiation
    (NIN_MFW_SMALL_STRIPS: wedge nodes 6 + 6 -- a partial last panel with c inside --, cube nodes 4 + 4 with the cube-node
    kernel off -- c in a block of its own).  Nodes with fewer cells than the instantiation holds (mixed mesh: zero column
    blocks, zero row tiles) ride along."""
    if kind == "hex":
        monkeypatch.setenv("NIN_GLS_NO_GROUP", "1")
    if switch:
        monkeypatch.setenv(switch, "1")
    mesh = {"tet": lambda: M.tet_mesh(6, jitter=0.1, seed=9), "wedge": lambda: M.wedge_mesh(6, 5, 4, jitter=0.06, seed=9),
            "hex": lambda: M.hex_mesh(7, jitter=0.15, seed=9), "mixed": lambda: M.mixed_mesh(10, 5, 5, jitter=0.1, seed=9)}[kind]()
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(1, 0.0), seed=4)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    plan = I.grid.gls_plan()
    assert plan["mfw_large" if kind in ("tet", "mixed") else "mfw_small"] > 0
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL


def test_gpu_multifrontal_general_kind(oracle_lib, monkeypatch):
    """Interior nodes whose cell graph has odd cycles (hex | pyramid and pyramid | tet interfaces: 16 and 26 cells) or
    cells with 4 faces at the node (pyramid apexes): the one-wavefront multifrontal kernel's general kind -- fronts = a
    maximal independent set of 3-face cells, the faces between two dense cells as free rows, up to 15 dense cells and 128 rows --
    against the oracle, and against the block kernel that takes them when the kind is switched off."""
    mesh = M.mixed_mesh(10, 5, 5, jitter=0.1, seed=11)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    got = {}
    monkeypatch.setenv("NIN_GLS_MFW_GENERAL", "1")       # (round 4: by default the wide kernel takes these nodes, next test)
    monkeypatch.setenv("NIN_GLS_NO_MFX", "1")
    for off in (False, True):
        if off:
            monkeypatch.setenv("NIN_GLS_NO_MFW_GENERAL", "1")
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        I.grid.to_device(0)
        plan = I.grid.gls_plan()
        assert (plan["mfw_general"] == 0) == off and plan["mfw_large"] > 0 and plan["mfx"] == 0
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL, off
        assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, off
        got[off] = plan
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    bp = np.asarray(I.grid.boundary_points).astype(bool)
    # (the pyramid apexes -- 7 cells, 43 rows -- fit the small-node kernel and go there; with it off they are general-kind nodes)
    assert got[False]["mfw_general"] == int(np.sum(~bp & np.isin(ne, (16, 26))))
    assert got[False]["small8"] >= int(np.sum(~bp & (ne == 7)))


def test_gpu_wide_kernel_takes_the_general_kind_by_default(oracle_lib):
    """The hex | pyramid and pyramid | tet interface nodes of a mixed mesh (16 and 26 cells, odd cycles) go to the wide
    multifrontal kernel by default (round 4); the pyramid apexes (7 cells) stay with the small-node kernel."""
    mesh = M.mixed_mesh(10, 5, 5, jitter=0.1, seed=11)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL and util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL
    plan = I.grid.gls_plan()
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    bp = np.asarray(I.grid.boundary_points).astype(bool)
    assert plan["mfx"] == int(np.sum(~bp & np.isin(ne, (16, 26)))) and plan["mfw_general"] == 0
    assert plan["small8"] >= int(np.sum(~bp & (ne == 7)))


@pytest.mark.parametrize("lattice,perm", [("bcc", "ALH"), ("bcc", "FAN"), ("random", "LIN")])
def test_gpu_wide_multifrontal_kernel_on_unstructured_tetrahedra(oracle_lib, monkeypatch, lattice, perm):
    """A Delaunay tetrahedrisation: 14 .. 40+ cells around an interior node, no two-colouring, a third of the nodes beyond the
    general kind's 12 + 15 cells (the mesh class of the reference's tetra numbers).  Three routes
    against the oracle: the default plan (every node that is not two-coloured through the wide kernel), NIN_GLS_MFW_GENERAL (the
    nodes that fit 12 + 15 cells through kernels_gls_mfw.hip's general kind, the rest through the wide kernel), NIN_GLS_NO_MFX
    (round 3's route: general kind + block kernel) -- and the plan itself: at least 95 % of the interior nodes
    stay off the block / scratch kernels by default."""
    mesh = M.delaunay_tet_mesh(11, seed=21, lattice=lattice)
    M.attach_fields(mesh, "u", perm=perm, neumann_plane=(1, 0.0), seed=2)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    plans = {}
    for route in ("default", "NIN_GLS_MFW_GENERAL", "NIN_GLS_NO_MFX", "NIN_GLS_MFX_NO_BOUNDARY", "NIN_GLS_NO_MFX_7X12"):
        if route != "default":
            monkeypatch.setenv(route, "1")
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL, route
        assert util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, route
        plans[route] = I.grid.gls_plan()
        if route != "default":
            monkeypatch.delenv(route)
    n_interior = int(np.sum(~np.asarray(I.grid.boundary_points).astype(bool)))
    one_wave = lambda p: p["mfw_large"] + p["mfw_small"] + p["mfw_general"] + p["mfx"] + p["hex8"]
    assert plans["default"]["mfx"] > 0 and plans["default"]["mfw_general"] == 0 and plans["NIN_GLS_NO_MFX"]["mfx"] == 0
    assert plans["NIN_GLS_MFW_GENERAL"]["mfw_general"] > 0 and 0 < plans["NIN_GLS_MFW_GENERAL"]["mfx"] < plans["default"]["mfx"]
    # (the interior nodes the block kernel keeps: more than 16 + 21 cells -- the random cloud has a few per cent of them)
    assert one_wave(plans["default"]) >= (0.95 if lattice == "bcc" else 0.85) * n_interior, (plans["default"], n_interior)
    assert one_wave(plans["NIN_GLS_NO_MFX"]) < one_wave(plans["default"])
    # the Neumann plane's nodes (half a node's cells, boundary faces as one row each): the wide kernel's boundary instantiation by
    # default, the block / small-node kernels with NIN_GLS_MFX_NO_BOUNDARY (round 3's route) -- same weights either way (above)
    assert plans["default"]["mfx_boundary"] > 0 and plans["NIN_GLS_MFX_NO_BOUNDARY"]["mfx_boundary"] == 0
    assert plans["NIN_GLS_MFX_NO_BOUNDARY"]["mfx"] == plans["default"]["mfx"]
    # the class between (7, 11) and (8, 13): its nodes are (8, 13)'s with NIN_GLS_NO_MFX_7X12 -- same weights either way (above)
    assert plans["NIN_GLS_NO_MFX_7X12"]["mfx_7x12"] == 0 and plans["NIN_GLS_NO_MFX_7X12"]["mfx"] == plans["default"]["mfx"]
    assert plans["NIN_GLS_NO_MFX_7X12"]["mfx_8x13"] == plans["default"]["mfx_8x13"] + plans["default"]["mfx_7x12"]
    if lattice == "bcc":
        assert plans["default"]["mfx_7x12"] > 0


@pytest.mark.parametrize("perm", ["ALH", "FAN"])
def test_gpu_tiles_in_global_memory_kernel_on_a_random_cloud(oracle_lib, monkeypatch, perm):
    """The Delaunay mesh of a random point cloud: a few per cent of its interior nodes have more cells than the wide kernel's 16 fronts +
    21 dense cells (up to 60 cells here).  kernels_gls_mfg.hip takes them -- up to 32 fronts + 40 dense cells, the tiles of the dense
    problem in a global-memory slot, left-looking panels -- where the block / global-scratch kernels ran before (NIN_GLS_NO_MFG: that
    route).  Both routes against the oracle, row-scaled and element by element; through the full launch, an explicit target list (the
    descriptors are made per call there) and the pieces of interpolate()'s pipeline."""
    mesh = M.delaunay_tet_mesh(10, seed=4, lattice="random")
    M.attach_fields(mesh, "u", perm=perm, neumann_plane=(0, 1.0), seed=3)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    Wo, _ = o.interpolate("u", "gls")
    plans = {}
    for route in ("default", "NIN_GLS_NO_MFG"):
        with monkeypatch.context() as mp:
            if route != "default":
                mp.setenv(route, "1")
            mp.setenv("NIN_E2E_MIN_NODES", "256")
            I = _interp()
            I.load_mesh(mesh_obj=mesh)
            P = I.grid.n_points
            w, nw = I.prepare_interpolator("gls", "u", np.arange(P))
            plans[route] = I.grid.gls_plan()
            assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL and util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, route
            assert util.elementwise_err(w, wo) <= util.elementwise_rtol("gls", perm), route
            targets = np.arange(1, P, 3)
            wt, nwt = I.prepare_interpolator("gls", "u", targets)
            assert np.array_equal(wt, w[targets]) and np.array_equal(nwt, nw[targets]), route
            W, _ = I.interpolate("u", "gls")                     # (the pipeline: sub-ranges of every list)
            assert util.csr_rowscaled_err(W, Wo.indptr, Wo.indices, Wo.data) <= util.WEIGHT_RTOL, route
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    interior = ~np.asarray(I.grid.boundary_points).astype(bool)
    d, b = plans["default"], plans["NIN_GLS_NO_MFG"]
    assert d["mfg_tiles"] >= int(np.sum(interior & (ne > 37))) > 20 and b["mfg_tiles"] == 0, (d, b)
    assert d["mfx"] == b["mfx"]
    heavy = ("block4", "block8", "scratch")
    assert sum(b[k] for k in heavy) == sum(d[k] for k in heavy) + d["mfg_tiles"]
    # every interior node of the cloud is on a one-wavefront multifrontal kernel now
    assert d["mfx"] + d["mfg_tiles"] + d["mfw_large"] + d["mfw_small"] + d["mfw_general"] + d["hex8"] >= int(interior.sum()) - int(np.sum(interior & (ne <= 12)))


def test_gpu_unstructured_prisms_small_class(oracle_lib, monkeypatch):
    """Unstructured PRISMS (a 2-D Delaunay triangulation extruded: the reference's "prism" mesh family): a node of valence V has 2 V
    wedges in a ring; V = 4 is the cube graph (cube-node kernel), V = 6 / 8 are two-coloured (kernels_gls_mfw.hip), V = 5 / 7 / 9 are
    not: 10- and 14-cell nodes go to the wide kernel's SMALL class (4 x 7 tiles: their dense problem is 43 x 19 / 59 x 25 after the
    fronts), where the small-node kernel (a dense 55 x 31 sweep) and the (6, 10) class ran before (NIN_GLS_NO_MFX_SMALL: that route)."""
    mesh = M.delaunay_wedge_mesh(12, 6, seed=3, lattice="random")
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=4)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    wo, no = o.prepare("gls", "u")
    plans = {}
    for route in ("default", "NIN_GLS_NO_MFX_SMALL"):
        with monkeypatch.context() as mp:
            if route != "default":
                mp.setenv(route, "1")
            I = _interp()
            I.load_mesh(mesh_obj=mesh)
            w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
            plans[route] = I.grid.gls_plan()
        assert util.rowscaled_err(w, wo) <= util.WEIGHT_RTOL and util.rowscaled_err(nw, no) <= util.WEIGHT_RTOL, route
        assert util.elementwise_err(w, wo) <= util.elementwise_rtol("gls", "ALH"), route
    ne = np.diff(np.asarray(I.grid.esup_ptr))
    interior = ~np.asarray(I.grid.boundary_points).astype(bool)
    d, b = plans["default"], plans["NIN_GLS_NO_MFX_SMALL"]
    assert d["mfx_4x7"] == int(np.sum(interior & np.isin(ne, (10, 14)))) > 50 and b["mfx_4x7"] == 0, (d, b)
    assert b["small12"] == d["small12"] + int(np.sum(interior & (ne == 10))) and b["mfx_6x10"] == d["mfx_6x10"] + int(np.sum(interior & (ne == 14)))
    assert d["hex8"] == int(np.sum(interior & (ne == 8))) and d["mfw_small"] == int(np.sum(interior & (ne == 12)))
    assert d["mfx"] == d["mfx_4x7"] + d["mfx_6x10"] + d["mfx_7x11"] + d["mfx_7x12"] + d["mfx_8x13"] + d["mfx_9x15"] + d["mfx_10x16"]


def test_gpu_wide_kernel_dense_phase_alone():
    """mfx_strips.hpp's xstrip_factor -- the blocked Householder QR in register tiles behind the wide kernel -- on random
    problems of every size class against a host QR (tools/test_xstrip.hip, compiled by __graft_entry__.build())."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "_bin", "test_xstrip")
    assert os.path.exists(exe), "python -c 'import __graft_entry__ as g; g.build()' builds it"
    r = subprocess.run([exe, "--no-timing"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


@pytest.mark.parametrize("sectors,layers", [(15, 3), (30, 3), (50, 2), (80, 2)])
def test_gpu_gls_high_degree_nodes(oracle_lib, sectors, layers):
    """Wedge fans: the axis nodes have 2 * sectors cells -- 91 / 181 / 301 / 481 unknowns -- which takes the block
    kernel through its wider column slots (2, 3) and, past 256 columns, the wave kernel on global scratch."""
    mesh = M.wedge_fan(sectors, layers, jitter=0.02, seed=sectors)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=3)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    assert I.grid.MX_ELEMENTS_PER_POINT == 2 * sectors
    for meth in ("idw", "ls", "gls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        tol = util.WEIGHT_RTOL if meth == "gls" else TIGHT
        assert util.rowscaled_err(w, wo) <= tol, meth
        assert util.rowscaled_err(nw, no) <= tol, meth


def test_gpu_gls_oversize_node_is_an_error_not_a_crash():
    """More than 1024 rows in one node's system is beyond the fallback kernel: NIN_ERANGE, nothing launched."""
    import ninpol_amd
    mesh = M.wedge_fan(120, 2)
    M.attach_fields(mesh, "u", perm="LIN")
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    with pytest.raises(ninpol_amd.NinpolError) as e:
        I.interpolate("u", "gls")
    assert e.value.code == -5
    W, _ = I.interpolate("u", "idw")      # the grid and the other methods are unaffected
    assert W.shape == (I.grid.n_points, I.grid.n_elems)


def test_gpu_apply_matches_oracle_matrix_product(oracle_lib):
    """Interpolator.apply against the ORACLE's W . u (the reference callers' next step, analytical.py:236): one field
    and a batch of three through nin_apply_fields_host, which computes the weights once."""
    mesh = M.mixed_mesh(10, 6, 6, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=4)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    u = np.concatenate(mesh.cell_data["u"])
    rng = np.random.default_rng(0)
    fields = np.stack([u, np.sin(3.0 * u), rng.uniform(-1.0, 1.0, len(u))])
    for meth in ("idw", "ls", "gls"):
        Wo, nwo = o.interpolate("u", meth)
        vals, nws = I.apply("u", meth)
        assert vals.shape == (I.grid.n_points,)
        ref = Wo.dot(u)
        assert util.rowscaled_err(nws, nwo) <= util.WEIGHT_RTOL
        assert np.abs(vals - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), meth
        many, nws3 = I.apply("u", meth, values=fields)
        assert many.shape == (3, I.grid.n_points)
        np.testing.assert_array_equal(nws3, nws)
        for k in range(3):
            ref = Wo.dot(fields[k])
            assert np.abs(many[k] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (meth, k)
        np.testing.assert_array_equal(many[0], vals)      # a batch of k and k single calls are the same sums
    with pytest.raises(ValueError):
        I.apply("u", "gls", values=np.zeros((2, 5)))


def test_gpu_apply_fused_in_the_cube_kernel(oracle_lib, monkeypatch):
    """GLS apply on a mesh with cube nodes: the cube-node kernel forms W . u itself (no row of weights is written for those
    nodes), the other kernels' rows go through a list kernel.  Against the oracle's W . u, and against the unfused path
    (NIN_APPLY_NO_FUSION: weights, then the apply kernel) to rounding; one field and three; a Neumann plane, so that the
    fused nodes, the quad nodes and the small nodes all carry values."""
    mesh = M.hex_mesh(17, 14, 11, jitter=0.15, seed=12)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(0, 1.0), seed=9)
    o = oracle_lib.OracleInterpolator("port", threads=8)
    o.load_mesh(mesh)
    Wo, nwo = o.interpolate("u", "gls")
    u = np.concatenate(mesh.cell_data["u"])
    rng = np.random.default_rng(5)
    fields = np.stack([u, np.cos(2.0 * u), rng.uniform(-1.0, 1.0, len(u))])
    got = {}
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("NIN_APPLY_NO_FUSION", "1")
        I = _interp()
        I.load_mesh(mesh_obj=mesh)
        vals, nws = I.apply("u", "gls")
        many, nws3 = I.apply("u", "gls", values=fields)
        assert I.grid.gls_plan()["hex8"] == 16 * 13 * 10
        assert util.rowscaled_err(nws, nwo) <= util.WEIGHT_RTOL
        np.testing.assert_array_equal(nws3, nws)
        np.testing.assert_array_equal(many[0], vals)
        for k in range(3):
            ref = Wo.dot(fields[k])
            assert np.abs(many[k] - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (fused, k)
        got[fused] = (many, nws)
    assert np.abs(got[True][0] - got[False][0]).max() <= 1e-13 * np.abs(got[False][0]).max()
    np.testing.assert_array_equal(got[True][1], got[False][1])


def _gls_degenerate_nodes(g, flag):
    """The nodes outside GLS's parity set, from the grid arrays alone (gls.pyx:165-182,266-267): computed (not a
    Dirichlet boundary node) but with no internal face at all (n_bface >= n_face: the reference leaves Mi empty) or
    with fewer rows than unknowns next to the node value (m < n - 1: rank-deficient by count)."""
    P = g.n_points
    ep, fp = np.asarray(g.esup_ptr), np.asarray(g.fsup_ptr)
    bf = np.asarray(g.boundary_faces).astype(bool)
    bp = np.asarray(g.boundary_points).astype(bool)
    neu = np.asarray(flag).astype(np.int64) != 0
    fs = np.asarray(g.fsup)
    nb_face = np.add.reduceat(np.append(bf[fs], False).astype(np.int64), fp[:-1])[:P] * (np.diff(fp) > 0)
    ne, nf = np.diff(ep), np.diff(fp)
    n_if = nf - nb_face
    m = ne + 3 * n_if + np.where(neu, nb_face, 0)
    n = 3 * ne + 1
    computed = ~(bp & ~neu)
    return computed, computed & ((n_if == 0) | (m < n - 1))


@pytest.mark.parametrize("kind", ["hex", "tet", "mixed"])
def test_gpu_gls_degenerate_set(oracle_lib, kind):
    """The zero-row policy of the GLS kernels is applied on EXACTLY the nodes enumerated by _gls_degenerate_nodes -- on
    a Neumann plane those are the box corners (one cell, no internal face) and, on tetrahedra, thin edge nodes -- and
    nowhere else; every other computed node is non-empty and matches the oracle (DESIGN.md 'parity set')."""
    mesh = {"hex": lambda: M.hex_mesh(5, 4, 4, jitter=0.1, seed=3), "tet": lambda: M.tet_mesh(4, jitter=0.1, seed=3),
            "mixed": lambda: M.mixed_mesh(8, 4, 4, jitter=0.1, seed=3)}[kind]()
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=5)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    o = oracle_lib.OracleInterpolator("port", threads=2)
    o.load_mesh(mesh)
    flag = mesh.point_data["neumann_flag_u"]
    computed, degenerate = _gls_degenerate_nodes(I.grid, flag)
    if kind == "hex":      # (a Kuhn-split corner has several tetrahedra, hence internal faces: nothing degenerate there)
        assert degenerate.sum() == 4, "the four corners of the Neumann plane: one cell, no internal face"
    full = np.arange(I.grid.n_points)
    w, nw = I.prepare_interpolator("gls", "u", full)
    wo, no = o.prepare("gls", "u", full)
    nonzero = np.abs(w).max(axis=1) > 0
    assert not nonzero[degenerate].any(), "zero row on every degenerate node"
    assert not nonzero[~computed].any(), "Dirichlet boundary nodes are skipped"
    regular = computed & ~degenerate
    assert nonzero[regular].all(), "every other computed node has weights"
    assert util.rowscaled_err(w[regular], wo[regular]) <= util.WEIGHT_RTOL
    assert util.rowscaled_err(nw[regular], no[regular]) <= util.WEIGHT_RTOL


# every way a node can reach a GLS kernel: name -> environment switches (read when the launch plan is built)
_GLS_ROUTES = {
    "default": (),                                                        # cube-node kernel / mfw strips (two-coloured) / the wide kernel / small, quad, block for the boundary
    "no_cube_kernel": ("NIN_GLS_NO_GROUP",),                              # cube nodes -> mfw small instantiation
    "mfw_lane_columns": ("NIN_GLS_NO_GROUP", "NIN_MFW_LANE_COLUMNS"),     # the mfw kernel's first form
    "mfw_row_lanes": ("NIN_MFW_NO_STRIPS",),                              # its second form (round 2's default) where the strip form runs now
    "mfw_small_strips": ("NIN_GLS_NO_GROUP", "NIN_MFW_SMALL_STRIPS"),     # the strip form in the small instantiation (wedge / cube nodes)
    "general_kind": ("NIN_GLS_MFW_GENERAL",),                             # nodes that are not two-coloured: kernels_gls_mfw.hip's general kind where it fits (round 3's default)
    "no_general_kind": ("NIN_GLS_NO_MFW_GENERAL", "NIN_GLS_NO_MFX"),      # ... -> block kernel
    "wide_for_two_coloured": ("NIN_GLS_NO_MFW",),                         # Kuhn / wedge nodes through the wide kernel (small ones: the small-node kernel)
    "no_quad_kernel": ("NIN_GLS_NO_QUAD4",),                              # nodes inside a boundary face -> small-node kernel
    "no_small_kernel": ("NIN_GLS_NO_SMALL", "NIN_GLS_NO_QUAD4"),          # boundary nodes -> block kernel (round 2's route)
    "small_where_it_fits": ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW"),        # the small-node kernel for every node of <= 12 cells and <= 64 rows
    "no_boundary_in_wide": ("NIN_GLS_MFX_NO_BOUNDARY",),                  # boundary nodes beyond the small-node kernel -> block kernel (round 3's route)
    "block_only": ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW", "NIN_GLS_NO_MFX", "NIN_GLS_NO_SMALL", "NIN_GLS_NO_QUAD4"),   # block kernel, 1 / 2 / 4 / 8 wavefronts per node
    "global_scratch": ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW", "NIN_GLS_FORCE_GLOBAL"),   # the wave kernel
}


@pytest.mark.parametrize("kind", ["hex", "tet", "wedge", "mixed"])
def test_gpu_gls_degenerate_zero_pivot_column(monkeypatch, kind):
    """Degenerate class (iii) of the parity set: a pivot column that is zero altogether.  On
    util.flat_mesh the z-column of EVERY cell is exactly zero in every node's system, so each computed node meets a zero
    pivot column -- in the cube-node kernel as the NaN its branch-free Householder scalars produce (isfinite test), in the
    mfw / block / wave kernels at their explicit guards.  The rule is ONE: the zero row, and neumann_ws = 0.  (The
    reference reads what dgels' singular exit left in B there, gls.pyx:457-472 ignores `info`: outside the parity set.)
    Every route to a kernel is forced in turn and must give exactly that, with the plan saying which kernel ran."""
    mesh = util.flat_mesh(kind)
    flag = mesh.point_data["neumann_flag_u"]
    plans = {}
    for route, switches in _GLS_ROUTES.items():
        with monkeypatch.context() as mp:
            for sw in switches:
                mp.setenv(sw, "1")
            I = _interp()
            I.load_mesh(mesh_obj=mesh)
            I.grid.to_device(0)
            plans[route] = I.grid.gls_plan()
            w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        assert np.isfinite(np.asarray(I.grid.normal_faces)).all()
        assert not np.any(w), (route, int(np.count_nonzero(np.abs(w).max(axis=1))))
        assert not np.any(nw), route
        # IDW / LS are untouched by any of this: rows of the computed nodes are there
        W, _ = I.interpolate("u", "idw")
        computed = ~(np.asarray(I.grid.boundary_points).astype(bool) & (flag == 0))
        assert np.array_equal(np.diff(W.indptr) > 0, computed)
    d = plans["default"]
    assert (d["hex8"] > 0) == (kind in ("hex", "mixed"))
    assert (d["mfw_large"] > 0) == (kind in ("tet", "mixed")) and (d["mfw_small"] > 0) == (kind == "wedge")
    assert (d["mfx"] > 0) == (kind == "mixed") and d["mfw_general"] == 0
    assert (plans["general_kind"]["mfw_general"] > 0) == (kind == "mixed") and plans["general_kind"]["mfx"] == 0
    assert plans["no_cube_kernel"]["hex8"] == 0 and plans["no_cube_kernel"]["mfw_small"] >= d["hex8"]
    assert plans["no_general_kind"]["mfw_general"] == 0 and plans["no_general_kind"]["mfx"] == 0
    assert plans["wide_for_two_coloured"]["mfx"] >= d["mfw_large"] + d["mfx"]
    assert plans["no_boundary_in_wide"]["mfx_boundary"] == 0
    small = ("small4", "small8", "small12")
    assert sum(d[k] for k in small) > 0 and all(plans["no_small_kernel"][k] == 0 for k in small)
    assert (d["quad4"] > 0 or kind == "tet") and plans["no_quad_kernel"]["quad4"] == 0   # (wedge meshes have quad nodes too: their lateral faces)
    assert sum(plans["no_quad_kernel"][k] for k in small) == sum(d[k] for k in small) + d["quad4"]
    assert sum(plans["small_where_it_fits"][k] for k in small) >= sum(d[k] for k in small) + (d["hex8"] if kind == "hex" else 0)
    for route in ("block_only", "global_scratch"):
        assert all(plans[route][k] == 0 for k in ("hex8", "mfw_large", "mfw_small", "mfw_general", "mfx", "mfx_boundary", "quad4") + small), route
    assert all(sum(p.values()) == I.grid.n_points for p in plans.values())


@pytest.mark.parametrize("n", [32, 64])
def test_gpu_gls_fan_permeability(oracle_lib, n):
    """The reference's own FAN tensor (tests/utils/analytical.py:285-293), element by element: cond(M_v) = 3e5 .. 6e5
    here, so every correct float64 QR -- the reference included -- is ~cond * eps from the exact solution and two of
    them differ by up to the sum (numbers in util.py).  All nodes against the C restatement under util.FAN_RTOL (1e-10 at
    32^3, 3e-10 at 64^3, this case only); IDW / LS do not see K and keep their bit-level bar.  The sharper statement is
    test_gpu_gls_fan_exact_sample."""
    mesh = M.hex_mesh(n, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="FAN", neumann_plane=(2, 0.0), seed=7)
    o = oracle_lib.OracleInterpolator("port", threads=16)
    o.load_mesh(mesh)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    full = np.arange(I.grid.n_points)
    for meth in ("gls", "idw", "ls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", full)
        tol = util.fan_rtol(n) if meth == "gls" else TIGHT
        e = max(util.rowscaled_err(w, wo), util.rowscaled_err(nw, no))
        print(f"FAN hex {n}^3 {meth}: HIP vs port {e:.3e} (bound {tol:.2e})")
        assert e <= tol, (meth, e)
    # the same nodes through the generic kernels (cube-node kernel off): the bound is a property of the case, not of a kernel
    if n == 32:
        import os
        for switches in (("NIN_GLS_NO_GROUP",), ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW"), ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW", "NIN_GLS_NO_SMALL"),   # mfw small / small-node / wide kernel
                         ("NIN_GLS_NO_GROUP", "NIN_GLS_NO_MFW", "NIN_GLS_NO_MFX", "NIN_GLS_NO_SMALL", "NIN_GLS_NO_QUAD4")):                              # block kernel
            for sw in switches:
                os.environ[sw] = "1"
            try:
                J = _interp()
                J.load_mesh(mesh_obj=mesh)
                w, nw = J.prepare_interpolator("gls", "u", full)
            finally:
                for sw in switches:
                    del os.environ[sw]
            assert J.grid.gls_plan()["hex8"] == 0
            wo, no = o.prepare("gls", "u")
            assert max(util.rowscaled_err(w, wo), util.rowscaled_err(nw, no)) <= util.fan_rtol(n), switches


@pytest.mark.parametrize("n", [32, 64])
def test_gpu_gls_fan_exact_sample(n):
    """FAN tensor, element-wise, against the committed pin tests/golden/pins/fan_exact.npz (generator beside it): a
    sample of interior and Neumann-plane nodes with the weights of THE REFERENCE ITSELF (`ref`, oracle/_ref in the dev
    container) and the exact solution of the same least-squares problem in 80-bit arithmetic (`exact`).  The HIP path
    must be (1) within the sum of the two distances of the reference, and (2) no further from exact than
    util.FAN_EXACT_SLACK x the reference's own distance (never tighter than 1e-10)."""
    z = util.load_fan_exact()
    nodes, exact, ref, ref_err = z[f"nodes_{n}"], z[f"exact_{n}"], z[f"ref_{n}"], float(z[f"ref_err_{n}"])
    mesh = M.hex_mesh(n, jitter=0.15, seed=0)
    M.attach_fields(mesh, "u", perm="FAN", neumann_plane=(2, 0.0), seed=7)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    w, _ = I.prepare_interpolator("gls", "u", nodes)
    e_exact = util.rowscaled_err(w, exact)
    e_ref = util.rowscaled_err(w, ref)
    bound = max(util.WEIGHT_RTOL, util.FAN_EXACT_SLACK * ref_err)
    print(f"FAN hex {n}^3, {len(nodes)} nodes: HIP vs exact {e_exact:.3e} (reference vs exact {ref_err:.3e}, bound {bound:.2e}); "
          f"HIP vs reference {e_ref:.3e}")
    assert e_exact <= bound
    assert e_ref <= e_exact + ref_err + 1e-16


def test_gpu_integration_md_plugin_stub_runs_verbatim():
    """the ctypes plugin a ninpol maintainer would add is executed as printed (only the
    library path is made absolute) against ninpol's 9-argument plugin convention, and must fill the dense
    (n_target, MX_ELEMENTS_PER_POINT) table exactly as our own Interpolator does."""
    import os, re
    import ninpol_amd
    from ninpol_amd import build as nbuild
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    code = code.replace('ctypes.CDLL("libninpol_amd.so")', f'ctypes.CDLL({nbuild.LIB!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    mesh = M.mixed_mesh(8, 5, 5, jitter=0.1, seed=4)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=6)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    args = I.process_mesh(mesh)
    g = I.grid
    for meth in ("gls", "idw", "ls"):
        plug = ns["AMDPlugin"](meth, args, mesh.points, device=0)
        weights = np.zeros((g.n_points, g.MX_ELEMENTS_PER_POINT))
        nws = np.zeros(g.n_points)
        plug.prepare(g, I.cells_data, I.points_data, I.faces_data, I.variable_to_index, "u",
                     np.arange(g.n_points), weights, nws)
        w, nw = I.prepare_interpolator(meth, "u", np.arange(g.n_points))
        assert np.array_equal(weights, w, equal_nan=True), meth
        assert np.array_equal(nws, nw, equal_nan=True), meth


def test_gpu_permeability_stays_resident_and_follows_edits():
    """interpolate() uploads permeability / diff_mag once per mesh; a table that is rebuilt or edited in place is
    uploaded again (the cache key samples the values)."""
    mesh = M.hex_mesh(7, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u", perm="ALH")
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    W1, _ = I.interpolate("u", "gls")
    key = I.grid._perm_key
    W2, _ = I.interpolate("u", "gls")
    assert I.grid._perm_key == key and np.array_equal(W1.data, W2.data)
    row = I.variable_to_index["cells"]["permeability"]
    I.cells_data[row, :I.grid.n_elems * 9] *= 1.0 + 0.3 * np.tile(np.arange(9) % 4 == 0, I.grid.n_elems)   # scale the diagonals
    I.cells_data[I.variable_to_index["cells"]["diff_mag"], :I.grid.n_elems] = I.compute_diffusion_magnitude(
        I.cells_data[row, :I.grid.n_elems * 9].reshape(-1, 9))
    W3, _ = I.interpolate("u", "gls")
    assert I.grid._perm_key != key
    J = _interp()                      # the same edited tables through a fresh object
    J.load_mesh(mesh_obj=mesh)
    J.cells_data[:] = I.cells_data
    W4, _ = J.interpolate("u", "gls")
    assert np.array_equal(W3.data, W4.data) and not np.array_equal(W3.data, W1.data)


def test_gpu_csr_result_is_canonical_and_scratch_can_be_released():
    """interpolate() hands scipy the compaction's arrays without the constructor's validation pass: they must BE what
    that pass would have accepted (indptr monotone from 0 to nnz, sorted duplicate-free in-range columns) and behave as a
    csr_matrix in the operations callers use; release_scratch() returns the kept device buffers and the next call works."""
    import scipy.sparse as sp
    import ninpol_amd
    mesh = M.mixed_mesh(9, 5, 5, jitter=0.1, seed=2)
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=4)
    I = _interp()
    I.load_mesh(mesh_obj=mesh)
    u = np.concatenate(mesh.cell_data["u"])
    for meth in ("gls", "idw", "ls"):
        W, nws = I.interpolate("u", meth)
        assert isinstance(W, sp.csr_matrix) and W.dtype == np.float64 and W.indices.dtype == np.int32
        assert W.indptr[0] == 0 and W.indptr[-1] == W.nnz == len(W.data) == len(W.indices)
        assert np.all(np.diff(W.indptr) >= 0) and W.indices.min() >= 0 and W.indices.max() < I.grid.n_elems
        rows = np.repeat(np.arange(W.shape[0]), np.diff(W.indptr))
        assert np.all((np.diff(W.indices) > 0) | (np.diff(rows) > 0)), "columns sorted and unique inside every row"
        V = sp.csr_matrix((W.data.copy(), W.indices.copy(), W.indptr.copy()), shape=W.shape)    # the validating constructor
        assert (W != V).nnz == 0 and np.array_equal(W.dot(u), V.dot(u)) and np.array_equal(W.T.dot(nws), V.T.dot(nws))
        assert np.array_equal(W[5:50].toarray(), V[5:50].toarray()) and (W + V).nnz == W.nnz
        first = (W.data.copy(), nws.copy())
        I.release_scratch()
        assert ninpol_amd.pinned_pool().idle_bytes() == 0
        W2, nws2 = I.interpolate("u", meth)
        assert np.array_equal(W2.data, first[0], equal_nan=True) and np.array_equal(nws2, first[1], equal_nan=True)
        vals, _ = I.apply("u", meth)
        I.release_scratch(pinned=False)
        vals2, _ = I.apply("u", meth)
        assert np.array_equal(vals, vals2, equal_nan=True)


@pytest.mark.parametrize("kind", ["hex", "mixed", "tet"])
def test_gpu_interpolate_pipeline_equals_one_piece(monkeypatch, kind):
    """interpolate() in pieces (weights, count, seeded scan, compaction and PCIe slices per quarter of the node range -- the
    default from 64 k nodes on) against interpolate() in one piece: identical CSR and neumann_ws, every method, with Neumann
    nodes and rows that vanish; and the chunk boundaries really cut every GLS list (all kernels run in every piece)."""
    mesh = {"hex": lambda: M.hex_mesh(13, 11, 9, jitter=0.15, seed=2), "mixed": lambda: M.mixed_mesh(12, 6, 6, jitter=0.1, seed=2),
            "tet": lambda: M.tet_mesh(7, jitter=0.1, seed=2)}[kind]()
    M.attach_fields(mesh, "u", perm="ALH", neumann_plane=(2, 0.0), seed=3)
    res = {}
    for mode in ("one", "pieces"):
        with monkeypatch.context() as mp:
            if mode == "one":
                mp.setenv("NIN_E2E_NO_PIPELINE", "1")
            else:
                mp.setenv("NIN_E2E_MIN_NODES", "0")
            I = _interp()
            I.load_mesh(mesh_obj=mesh)
            assert I.grid.n_points >= 256
            for meth in ("gls", "idw", "ls"):
                W, nws = I.interpolate("u", meth)
                W2, nws2 = I.interpolate("u", meth)          # buffers of the path are reused
                assert np.array_equal(W.data, W2.data, equal_nan=True)
                res[(mode, meth)] = (W.indptr.copy(), W.indices.copy(), W.data.copy(), np.array(nws))
    for meth in ("gls", "idw", "ls"):
        a, b = res[("one", meth)], res[("pieces", meth)]
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True), meth
