"""Per-phase s_memtime stamps of one pass (16 nodes) of the cube-node kernel, nin_gls_hex8w2_kernel (round 2's one-wave kernel is
gone since round 4), wave 0 of workgroup 0, pass NIN_MF_STAMP_PASS (default 8)
(needs a -DNIN_MF_STAMPS build of kernels_gls_hex8mf.hip; the build clobbers neumann_ws of the first 8 cube nodes):
    bash tools/build_variant.sh stamps kernels_gls_hex8mf.hip -DNIN_MF_STAMPS
    NINPOL_AMD_LIB=tools/_bin/lib_stamps.so python tools/stamps_hex8mf.py [edge]
With two waves per SIMD a phase's cycles include what the other wave took of the SIMD in the meantime."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = M.hex_mesh(n, jitter=0.15); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
plan = I.device_plan("u", "gls")
out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.zeros(plan.n_points, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
first = np.nonzero(np.asarray(I.grid.boundary_points) == 0)[0][:8]      # the first 8 interior (= cube) nodes
s = nws.cpu().numpy()[first]
names = ["ids in (LDS-DMA), geometry loads, face rows", "phase 1 (panel + 10 columns)", "tile put together (LDS, selects)",
         "phase 2, steps 0-5", "phase 2, steps 6-11", "back-substitution", "residuals, weights, stores"]
for i in range(7): print(f"{names[i]:50s} {s[i + 1] - s[i]:8.0f} cycles")
print(f"{'pass (16 nodes)':50s} {s[7]:8.0f} cycles")
