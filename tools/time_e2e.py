"""End-to-end interpolate() (field upload -> weights -> device-side compaction -> PCIe -> scipy.sparse.csr_matrix) on the
10 M-cell mesh (or another), with a phase breakdown: `python tools/time_e2e.py [edge | del54 | delr40]`; NIN_TIMING=1 adds the native call's own laps."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ninpol_amd
from ninpol_amd import interpolator as NI
from ninpol_amd import mesh as M
arg = sys.argv[1] if len(sys.argv) > 1 else "216"   # an edge of the hexahedron mesh, or del<n> / delr<n>: a Delaunay mesh
m = (M.delaunay_tet_mesh(int(arg[4:]), seed=0, lattice="random") if arg.startswith("delr") else M.delaunay_tet_mesh(int(arg[3:]), seed=0)
     if arg.startswith("del") else M.hex_mesh(int(arg), jitter=0.15)); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
phases = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(mod, name, g)
wrap(NI, "_upload_fields")
wrap(NI, "_native_interpolate")
wrap(NI, "_wrap_csr")
for meth in ("gls", "idw"):
    for i in range(5):
        phases.clear()
        t0 = time.perf_counter(); W, nws = I.interpolate("u", meth); dt = time.perf_counter() - t0
        ph = ", ".join(f"{k.strip('_')} {v * 1e3:.1f} ms" for k, v in phases.items())
        print(f"{meth}: interpolate() {dt:.4f} s = {I.grid.n_points / dt / 1e6:.1f} Mnodes/s, nnz {W.nnz}, checksum {W.data[::1000].sum():.12f}  [{ph}, "
              f"other {(dt - sum(phases.values())) * 1e3:.1f} ms]")
        del W, nws
