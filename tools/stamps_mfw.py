"""Phase stamps (s_memtime) of the one-wavefront multifrontal GLS kernel on a Kuhn-tet node (needs a -DNIN_MFW_STAMPS build):
NIN_EXTRA_HIPCC_FLAGS=-DNIN_MFW_STAMPS python -m ninpol_amd.build --force; python tools/stamps_mfw.py [tet | wedge]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ninpol_amd import mesh as M
kind = sys.argv[1] if len(sys.argv) > 1 else "tet"          # "tet": a Kuhn node (24 cells); "wedge": a wedge node (12 cells, the small instantiation)
NE = 24 if kind == "tet" else 12
m = (M.tet_mesh(24, jitter=0.1) if kind == "tet" else M.wedge_mesh(30, jitter=0.05)); M.attach_fields(m, "u", perm="ALH")
import ninpol_amd
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nws = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
g = I.grid
ne = np.diff(np.asarray(g.esup_ptr)); bp = np.asarray(g.boundary_points).astype(bool)
first = np.nonzero((ne == NE) & ~bp)[0][:7]          # the first nodes of the kernel's list
st = np.asarray(nws)[first]
names = ["node, descriptor, CSR row starts", "phase 1 (loads, fronts in lanes)", "rows gathered into the lanes", "dense factorisation",
         "back-substitution", "residuals, weights, stores"]
print(f"one {kind} node ({NE} cells), wavefront 0 of workgroup 0, its 5th node (cycles):")
for i, n in enumerate(names):
    print(f"  {n:36s} {st[i + 1] - st[i]:8.0f}")
print(f"  {'node':36s} {st[6]:8.0f}")

sub = np.asarray(nws)[np.nonzero((ne == NE) & ~bp)[0][8:13]]
if sub.sum() > 0 and sub.max() < 1e9:
    print("  inside the dense factorisation (strip form), summed over the panels:")
    for n, v in zip(["panels factored (4 columns each, vector unit)", "T", "W = V^T C, T^T W", "C -= V W'", "rows of R stored, blocks shifted"], sub):
        print(f"    {n:48s} {v:8.0f}")
