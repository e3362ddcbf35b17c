"""The reference's FAN tensor (tests/utils/analytical.py:285-293) on hex_mesh(n, jitter=0.15, seed=0), Neumann plane z = 0 -- the case of
tests/test_gpu_parity.py::test_gpu_gls_fan_permeability -- at growing n: the HIP path against the C restatement, row-scaled and
element-wise, so that the kappa ~ 1/h trend is on record up to the benchmark's neighbourhood.
python tools/fan_trend.py [edges ...]     (GPU box; the restatement runs on the host cores)"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import ninpol_amd, ninpol_oracle as O, util
from ninpol_amd import mesh as M
O.build_port()
for n in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    for perm in ("FAN", "ALH"):
        m = M.hex_mesh(n, jitter=0.15, seed=0); M.attach_fields(m, "u", perm=perm, neumann_plane=(2, 0.0), seed=7)
        o = O.OracleInterpolator("port", threads=os.cpu_count() or 8); o.load_mesh(m)
        t0 = time.time(); wo, no = o.prepare("gls", "u"); t_o = time.time() - t0
        I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
        w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
        print(f"hex {n}^3 {perm}: HIP vs restatement, all {I.grid.n_points} nodes: row-scaled {util.rowscaled_err(w, wo):.2e}, element-wise "
              f"(entries >= {util.ELEMENTWISE_FLOOR:g} of their row's largest) {util.elementwise_err(w, wo):.2e}, neumann_ws {util.rowscaled_err(nw, no):.2e}"
              f"   (restatement {t_o:.1f} s)", flush=True)
        del m, o, I, w, wo
