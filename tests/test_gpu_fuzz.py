"""Random meshes / fields / Neumann planes through the HIP path against the oracle (a fixed-seed slice of
tools/fuzz_parity.py): every mesh family, both grid builders, all three methods, every GLS kernel."""
import numpy as np
import pytest

import util
from ninpol_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(25))
def test_fuzz_case(oracle_lib, case):
    import ninpol_amd
    rng = np.random.default_rng(7000 + case)
    kind = ["hex", "tet", "wedge", "mixed", "fan"][case % 5] if case < 15 else "delaunay" if case < 21 else "prisms"   # (cases 0 .. 14 are round 3's, unchanged)
    nx, ny, nz = (int(v) for v in rng.integers(3, 9, size=3))
    jit = float(rng.uniform(0.0, 0.2))
    seed = int(rng.integers(1 << 30))
    if kind == "hex":
        m = M.hex_mesh(nx, ny, nz, jitter=jit, seed=seed)
    elif kind == "tet":
        m = M.tet_mesh(max(nx - 2, 2), max(ny - 2, 2), max(nz - 2, 2), jitter=min(jit, 0.1), seed=seed)
    elif kind == "wedge":
        m = M.wedge_mesh(nx, ny, nz, jitter=min(jit, 0.08), seed=seed)
    elif kind == "mixed":
        m = M.mixed_mesh(max(nx, 4) + 2, ny, nz, jitter=min(jit, 0.1), seed=seed)
    elif kind == "prisms":     # unstructured prisms: a 2-D Delaunay triangulation (jittered grid, or a random cloud) extruded
        m = M.delaunay_wedge_mesh(max(nx, 4) + 2, max(nz - 2, 2), jitter=0.1 + jit, seed=seed, lattice="random" if case % 2 == 0 else "grid")
    elif kind == "delaunay":   # unstructured tetrahedra: jittered body-centred cloud or (every third case) a random cloud
        m = M.delaunay_tet_mesh(max(nx, 4), jitter=0.1 + jit, seed=seed, lattice="random" if case % 3 == 0 else "bcc")
    else:
        m = M.wedge_fan(int(rng.integers(5, 70)), int(rng.integers(2, 4)), jitter=0.02, seed=seed)
    plane = None if rng.random() < 0.3 else (int(rng.integers(0, 3)), float(rng.integers(0, 2)))
    perm = ["ALH", "LIN", "FAN"][int(rng.integers(0, 3))]    # FAN: cond(M_v) ~1e5, still inside 1e-10 at these sizes
    M.attach_fields(m, "u", perm=perm, neumann_plane=plane, seed=seed % 1000)
    o = oracle_lib.OracleInterpolator("port", threads=4)
    o.load_mesh(m)
    I = ninpol_amd.Interpolator(grid_build=["host", "device"][case % 2])
    I.load_mesh(mesh_obj=m)
    for k in util.GRID_ARRAYS:
        np.testing.assert_array_equal(getattr(I.grid, k), getattr(o.grid, k), err_msg=k)
    for meth in ("idw", "ls", "gls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        tol = util.WEIGHT_RTOL if meth == "gls" else 1e-14
        assert util.rowscaled_err(w, wo) <= tol, (kind, meth)
        assert util.rowscaled_err(nw, no) <= tol, (kind, meth)
        # element-wise relative on every entry down to 1e-3 of its row's largest (util.py)
        assert util.elementwise_err(w, wo) <= util.elementwise_rtol(meth, perm), (kind, meth, perm)
