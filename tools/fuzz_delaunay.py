"""Random unstructured meshes (Delaunay clouds of 1-4 k nodes, body-centred with jitter up to 0.45 or random; Neumann planes of every
orientation; ALH / LIN / FAN) through the HIP path against the oracle, with the launch plan printed: python tools/fuzz_delaunay.py [cases]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import util, ninpol_amd, ninpol_oracle as O
from ninpol_amd import mesh as M
O.build_port()
rng = np.random.default_rng(424242)
worst = 0.0
for case in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    n = int(rng.integers(5, 15)); lat = "random" if case % 3 == 0 else "bcc"; jit = float(rng.uniform(0.05, 0.45)); seed = int(rng.integers(1 << 30))
    m = M.delaunay_tet_mesh(n, jitter=jit, seed=seed, lattice=lat)
    plane = None if rng.random() < 0.25 else (int(rng.integers(0, 3)), float(rng.integers(0, 2)))
    perm = ["ALH", "LIN", "FAN"][int(rng.integers(0, 3))]
    M.attach_fields(m, "u", perm=perm, neumann_plane=plane, seed=seed % 997)
    o = O.OracleInterpolator("port", threads=16); o.load_mesh(m)
    I = ninpol_amd.Interpolator(grid_build=["host", "device"][case % 2]); I.load_mesh(mesh_obj=m)
    for k in util.GRID_ARRAYS:
        assert np.array_equal(getattr(I.grid, k), getattr(o.grid, k)), (case, k)
    wo, no = o.prepare("gls", "u")
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    e = max(util.rowscaled_err(w, wo), util.rowscaled_err(nw, no)); ew = util.elementwise_err(w, wo)
    worst = max(worst, e)
    plan = {k: v for k, v in I.grid.gls_plan().items() if v}
    print(f"case {case:2d} n={n:2d} {lat:6s} jitter {jit:.2f} plane={plane} perm={perm}: P={I.grid.n_points} MX={I.grid.MX_ELEMENTS_PER_POINT} row-scaled {e:.2e} element-wise {ew:.2e} {plan}", flush=True)
    assert e <= util.WEIGHT_RTOL and ew <= util.elementwise_rtol("gls", perm), (case, e, ew)
print("worst row-scaled error", worst)
