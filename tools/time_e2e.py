"""End-to-end interpolate() (weights -> device-side compaction -> PCIe -> scipy.sparse.csr_matrix) on the 10 M-cell mesh."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import ninpol_amd
from ninpol_amd import mesh as M
m = M.hex_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 216, jitter=0.15); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
for meth in ("gls", "idw"):
    for i in range(4):
        t0 = time.time(); W, nws = I.interpolate("u", meth); dt = time.time() - t0
        print(f"{meth}: interpolate() {dt:.3f} s = {I.grid.n_points / dt / 1e6:.1f} Mnodes/s, nnz {W.nnz}, checksum {W.data[::1000].sum():.12f}")
        del W, nws
