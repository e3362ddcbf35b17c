import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
for name, planes in (("all-Dirichlet", None), ("Neumann plane z=0", (2, 0.0))):
    m = M.hex_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 216, jitter=0.15); M.attach_fields(m, "u", perm="ALH", neumann_plane=planes)
    I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
    st = torch.cuda.current_stream()
    plan = I.device_plan("u", "gls")
    out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.empty(plan.n_points, dtype=torch.float64, device="cuda")
    for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(5): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
    b.record(st); torch.cuda.synchronize()
    print(name, "gls ms", a.elapsed_time(b) / 5, I.grid.gls_plan())
    del I, plan, out, nws
