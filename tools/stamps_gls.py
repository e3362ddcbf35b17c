import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import ninpol_amd
from ninpol_amd import mesh as M
m = M.hex_mesh(48, jitter=0.15, seed=0); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nws = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
r = int(np.argmax(w.max(axis=1))); st = w[r] - 1.0e6
print("stamps (cycles since pass start): loads, faces, staging, QR, back, end:", st, "diffs", np.diff(st))
