"""One wavefront's cycles by phase in kernels_gls_mfg.hip (library built with -DNIN_MFG_STAMPS: bash tools/rebuild_unit.sh kernels_gls_mfg.hip
-DNIN_MFG_STAMPS): the kernel writes its s_memtime differences over the first 8 weights of each of its nodes.  GPU box: python tools/stamps_mfg.py [delr24]"""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import numpy as np
import ninpol_amd
from ninpol_amd import mesh as M
name = sys.argv[1] if len(sys.argv) > 1 else "delr24"
n = int(name[4:])
m = M.delaunay_tet_mesh(n, seed=0, lattice="random"); M.attach_fields(m, "u", perm="ALH", seed=3)
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
W, _ = I.interpolate("u", "gls")
print(I.grid.gls_plan())
W = W.tocsr(); ip = W.indptr; d = W.data
rows = [p for p in range(len(ip) - 1) if ip[p + 1] - ip[p] >= 8 and d[ip[p]] > 1000.0]
A = np.array([d[ip[p]:ip[p] + 8] for p in rows])
names = ["zero fill + phase 1 + scatter", "group loads / stores, r.r", "reflectors of the panels left of the group", "panel steps", "in-group applies", "back substitution", "weights", "nq*1000+ncb"]
print(f"{len(rows)} nodes; memtime ticks (100 MHz) per node, mean / max:")
for k in range(7): print(f"  {names[k]:45s} {A[:, k].mean():10.0f} {A[:, k].max():10.0f}")
print("  total", A[:, :7].sum(axis=1).mean(), " mean nq, ncb:", (A[:, 7] // 1000).mean(), (A[:, 7] % 1000).mean())
