import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
m = M.mixed_mesh(200, 120, 120, jitter=0.1); M.attach_fields(m, "u", perm="ALH")
I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
plan = I.device_plan("u", "gls")
out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.empty(plan.n_points, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
for _ in range(2): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
b.record(st); torch.cuda.synchronize()
ne = np.diff(np.asarray(I.grid.esup_ptr))
print("mixed10m gls ms", a.elapsed_time(b) / 3, {k: v for k, v in I.grid.gls_plan().items() if v})
