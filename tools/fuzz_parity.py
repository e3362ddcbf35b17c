"""Random meshes / fields / Neumann planes: GPU weights against the oracle port (not part of the test suite).
python tools/fuzz_parity.py [n_cases]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util
import numpy as np
import util
import ninpol_amd
from ninpol_amd import mesh as M
spec = importlib.util.spec_from_file_location("ninpol_oracle", os.path.join(os.getcwd(), "oracle", "ninpol_oracle.py"))
O = importlib.util.module_from_spec(spec); spec.loader.exec_module(O)
O.build_port()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(2026)
worst = {"idw": 0.0, "ls": 0.0, "gls": 0.0}
for case in range(n_cases):
    kind = ["hex", "tet", "wedge", "mixed", "fan", "delaunay", "prisms"][case % 7]
    nx, ny, nz = (int(v) for v in rng.integers(3, 9, size=3))
    jit = float(rng.uniform(0.0, 0.2)); seed = int(rng.integers(1 << 30))
    if kind == "hex": m = M.hex_mesh(nx, ny, nz, jitter=jit, seed=seed)
    elif kind == "tet": m = M.tet_mesh(max(nx - 2, 2), max(ny - 2, 2), max(nz - 2, 2), jitter=min(jit, 0.1), seed=seed)
    elif kind == "wedge": m = M.wedge_mesh(nx, ny, nz, jitter=min(jit, 0.08), seed=seed)
    elif kind == "mixed": m = M.mixed_mesh(max(nx, 4) + 2, ny, nz, jitter=min(jit, 0.1), seed=seed)
    elif kind == "prisms": m = M.delaunay_wedge_mesh(max(nx, 4) + 2, max(nz - 2, 2), jitter=0.1 + jit, seed=seed, lattice="random" if case % 3 == 0 else "grid")
    elif kind == "delaunay": m = M.delaunay_tet_mesh(max(nx, 4), jitter=0.1 + jit, seed=seed, lattice="random" if case % 4 == 0 else "bcc")
    else: m = M.wedge_fan(int(rng.integers(5, 70)), int(rng.integers(2, 4)), jitter=0.02, seed=seed)
    plane = None if rng.random() < 0.3 else (int(rng.integers(0, 3)), float(rng.integers(0, 2)))
    perm = ["ALH", "LIN", "FAN"][int(rng.integers(0, 3))]
    M.attach_fields(m, "u", perm=perm, neumann_plane=plane, seed=seed % 1000)
    o = O.OracleInterpolator("port", threads=8); o.load_mesh(m)
    I = ninpol_amd.Interpolator(grid_build=["host", "device"][case % 2]); I.load_mesh(mesh_obj=m)
    for k in util.GRID_ARRAYS:
        assert np.array_equal(getattr(I.grid, k), getattr(o.grid, k)), (case, kind, k)
    for meth in ("idw", "ls", "gls"):
        wo, no = o.prepare(meth, "u")
        w, nw = I.prepare_interpolator(meth, "u", np.arange(I.grid.n_points))
        e = max(util.rowscaled_err(w, wo), util.rowscaled_err(nw, no))
        worst[meth] = max(worst[meth], e)
        tol = util.WEIGHT_RTOL if meth == "gls" else 1e-14
        assert e <= tol, (case, kind, meth, e)
    print(f"case {case:2d} {kind:5s} P={I.grid.n_points:5d} E={I.grid.n_elems:5d} MX={I.grid.MX_ELEMENTS_PER_POINT:3d} plane={plane} perm={perm} ok", flush=True)
print("worst row-scaled errors:", worst)
