import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
m = M.hex_mesh(48, jitter=0.15, seed=0); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
plan = I.device_plan("u", "gls")
out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.empty(plan.n_points, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
for _ in range(2): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
b.record(st); torch.cuda.synchronize()
ms = a.elapsed_time(b)/3
nint = 47**3
blocks = int(os.environ.get("NIN_GLS_MAX_BLOCKS", "0"))
passes_per_wave = nint/4/(blocks*4) if blocks else 0
print(f"blocks {blocks} ms {ms:.3f} interior {nint} passes/wave {passes_per_wave:.1f} us/pass {ms*1e3/passes_per_wave if blocks else 0:.2f} cycles/pass@2.4GHz {ms*1e-3*2.4e9/passes_per_wave if blocks else 0:.0f}")
