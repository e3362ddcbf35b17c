"""FAN tensor on a random-cloud Delaunay mesh: the row-scaled error against the oracle by kernel -- the nodes beyond 37 cells through
kernels_gls_mfg.hip (default) and through the dense QR of the block / fallback kernels (NIN_GLS_NO_MFG=1), the other interior nodes beside them.
GPU box: python tools/fan_by_kernel.py"""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np
import ninpol_amd, ninpol_oracle, util
from ninpol_amd import mesh as M
ninpol_oracle.build_port()
m = M.delaunay_tet_mesh(24, seed=0, lattice="random"); M.attach_fields(m, "u", perm="FAN", neumann_plane=(0, 1.0), seed=3)
o = ninpol_oracle.OracleInterpolator("port", threads=16); o.load_mesh(m)
wo, no = o.prepare("gls", "u")
res = {}
for route in ("default", "NIN_GLS_NO_MFG"):
    if route != "default": os.environ[route] = "1"
    I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
    w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
    res[route] = w
    ne = np.diff(np.asarray(I.grid.esup_ptr)); bp = np.asarray(I.grid.boundary_points).astype(bool)
    scale = np.abs(wo).max(axis=1); scale[scale == 0] = 1
    err = np.abs(w - wo).max(axis=1) / scale
    big = ~bp & (ne > 37)
    print(route, {k: v for k, v in I.grid.gls_plan().items() if v and k in ("mfg_tiles", "block8", "block4", "scratch")})
    print("   nodes > 37 cells: max %.2e  median %.2e   other interior: max %.2e median %.2e" % (err[big].max(), np.median(err[big]), err[~bp & ~big].max(), np.median(err[~bp & ~big])))
d = np.abs(res["default"] - res["NIN_GLS_NO_MFG"]).max(axis=1) / scale
print("default vs NO_MFG route on the big nodes: max %.2e" % d[big].max())
