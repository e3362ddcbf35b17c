"""Per-step s_memtime stamps of the GLS block kernel (needs a -DNIN_BLOCK_STAMPS build):
NIN_EXTRA_HIPCC_FLAGS=-DNIN_BLOCK_STAMPS python -m ninpol_amd.build --force; python tools/stamps_block.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ninpol_amd import mesh as M
N = 24
m = M.tet_mesh(N, jitter=0.1); M.attach_fields(m, "u", perm="ALH")
node = (N + 1) ** 2 * (N // 2) + (N + 1) * (N // 2) + N // 2     # an interior node
os.environ["NIN_GLS_BLOCK_DEBUG"] = str(node << 8)
import ninpol_amd
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
w, nws = I.prepare_interpolator("gls", "u", np.arange(I.grid.n_points))
nws = np.asarray(nws)
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 8
st = nws[: waves * 256 * 4].reshape(waves, 256, 4)
ks = [k for k in range(0, 200) if st[0, k, 0] != 0]       # the dense steps (the sparse first phase has no stamps)
t0 = st[0, ks[0], 0]
np.set_printoptions(linewidth=200, suppress=True)
for k in ks:
    d = st[0, k]
    print(f"k={k:3d} w0: start {d[0]-t0:8.0f}  math {d[1]-d[0]:6.0f}  sweep {d[2]-d[1]:6.0f}  publish+barrier {d[3]-d[2]:6.0f} | sweep per wave", (st[:, k, 2] - st[:, k, 1]).astype(int), "| barrier wait", (st[:, k, 3] - st[:, k, 2]).astype(int))
print("dense steps", ks[0], "..", ks[-1], "total", st[0, ks[-1], 3] - t0)
c0, c1 = st[0, 200], st[0, 201]
print(f"node timeline (wave 0, cycles): plan+assembly {c0[1]-c0[0]:.0f}  fronts {c0[2]-c0[1]:.0f}  first dots {c0[3]-c0[2]:.0f}  "
      f"dense steps {c1[0]-c0[3]:.0f}  tail {c1[1]-c1[0]:.0f}  total {c1[1]-c0[0]:.0f}")
