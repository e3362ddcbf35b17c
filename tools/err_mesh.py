"""GLS weights on a tetrahedron / wedge / mixed mesh against the oracle (C restatement): the largest row-scaled error
(NINPOL_AMD_LIB / NIN_* switches from the environment).  python tools/err_mesh.py [tet|wedge|mixed] [perm]"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import ninpol_amd
from ninpol_amd import mesh as M
import util
from oracle import ninpol_oracle as O
kind = sys.argv[1] if len(sys.argv) > 1 else "tet"
perm = sys.argv[2] if len(sys.argv) > 2 else "ALH"
m = {"tet": lambda: M.tet_mesh(10, jitter=0.1, seed=3), "wedge": lambda: M.wedge_mesh(12, jitter=0.05, seed=3),
     "mixed": lambda: M.mixed_mesh(16, 8, 8, jitter=0.1, seed=3)}[kind]()
M.attach_fields(m, "u", perm=perm, neumann_plane=(2, 0.0), seed=5)
o = O.OracleInterpolator("port", threads=8); o.load_mesh(m)
wo, no = o.prepare("gls", "u")
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nw = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
print(f"{kind} perm {perm}: row-scaled error vs oracle {util.rowscaled_err(w, wo):.3e} (neumann_ws {util.rowscaled_err(nw, no):.3e}); plan { {k: v for k, v in I.grid.gls_plan().items() if v} }")
