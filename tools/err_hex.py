"""GLS weights on a jittered hexahedron mesh against the oracle (C restatement): the largest row-scaled error, per library variant
(NINPOL_AMD_LIB / NIN_* switches from the environment).  python tools/err_hex.py [edge] [perm]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import ninpol_amd
from ninpol_amd import mesh as M
import util
from oracle import ninpol_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
perm = sys.argv[2] if len(sys.argv) > 2 else "ALH"
m = M.hex_mesh(n, jitter=0.15, seed=3); M.attach_fields(m, "u", perm=perm, neumann_plane=(2, 0.0), seed=5)
o = O.OracleInterpolator("port", threads=8); o.load_mesh(m)
wo, no = o.prepare("gls", "u")
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
print(f"hex {n}^3 perm {perm}: row-scaled error vs oracle {util.rowscaled_err(w, wo):.3e} (neumann_ws {util.rowscaled_err(nw, no):.3e}); plan {I.grid.gls_plan()['hex8']} cube nodes")
