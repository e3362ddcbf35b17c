"""Stamps (s_memtime) inside the dense phase of the wide multifrontal GLS kernel on one Delaunay node (needs a -DNIN_MFX_STAMPS build:
bash tools/build_variant.sh mfxstamps kernels_gls_mfx.hip -DNIN_MFX_STAMPS; NINPOL_AMD_LIB=$PWD/tools/_bin/lib_mfxstamps.so python tools/stamps_mfx.py)"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ninpol_amd import mesh as M
m = M.delaunay_tet_mesh(24, seed=0); M.attach_fields(m, "u", perm="ALH")
import ninpol_amd
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nws = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
import ctypes
from ninpol_amd import _lib
cls = np.asarray(I.grid.node_class()) if hasattr(I.grid, "node_class") else None
# the kernel's list = the nodes of class 247 in ascending order: recover them from the plan by exclusion is not possible from Python,
# so the stamps build writes into nws[nodes[0..7]]: the eight smallest node ids that hold a value > 100 there
cand = np.nonzero(np.asarray(nws) > 3.5)[0][:8]
st = np.asarray(nws)[cand]
print("plan", I.grid.gls_plan())
names = ["panels factored (vector unit)", "V, T, first rows of R", "W = V^T C (matrix unit)", "T^T W", "C -= V W', rows of R stored", "whole factorisation", "rows", "columns"]
for n, v in zip(names, st):
    print(f"  {n:36s} {v:9.0f}")
