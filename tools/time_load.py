"""load_mesh at 216^3 cells with the host and the device grid builder (NIN_TIMING=1 prints the phases)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ninpol_amd
from ninpol_amd import mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
t = time.time(); m = M.hex_mesh(n, jitter=0.15); M.attach_fields(m, "u", perm="ALH"); print("mesh gen", round(time.time() - t, 2))
for build in ("device", "host", "device"):
    I = ninpol_amd.Interpolator(logging=bool(os.environ.get("NIN_TIMING")), grid_build=build)
    t = time.time(); I.load_mesh(mesh_obj=m); t1 = time.time() - t
    t = time.time(); W, nw = I.interpolate("u", "gls"); t2 = time.time() - t
    print(f"grid_build={build}: load_mesh {t1:.2f} s, first interpolate(gls) {t2:.2f} s, nnz {W.nnz}")
    del I
