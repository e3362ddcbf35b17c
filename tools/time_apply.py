"""W . u on the device (DevicePlan.apply: weights + row-block apply) per mesh and method: python tools/time_apply.py del54 [mixed10m ...]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
cases = {"tet40": lambda: M.tet_mesh(40, jitter=0.1), "mixed10m": lambda: M.mixed_mesh(200, 120, 120, jitter=0.1), "hex216": lambda: M.hex_mesh(216, jitter=0.15),
         "del54": lambda: M.delaunay_tet_mesh(54, seed=0), "delr40": lambda: M.delaunay_tet_mesh(40, seed=0, lattice="random")}
for name in sys.argv[1:] or ["del54"]:
    m = cases[name](); M.attach_fields(m, "u", perm="ALH")
    I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
    st = torch.cuda.current_stream()
    for meth in os.environ.get("NIN_METHODS", "idw,ls").split(","):
        plan = I.device_plan("u", meth)
        for k in (1, 4):
            u = torch.rand(k, I.grid.n_elems, dtype=torch.float64, device="cuda"); v = torch.empty(k, I.grid.n_points, dtype=torch.float64, device="cuda")
            nws = torch.empty(I.grid.n_points, dtype=torch.float64, device="cuda")
            plan.launch_apply(u.data_ptr(), k, v.data_ptr(), nws.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(5): plan.launch_apply(u.data_ptr(), k, v.data_ptr(), nws.data_ptr(), st.cuda_stream)
            b.record(st); torch.cuda.synchronize()
            print(f"{name}: P={I.grid.n_points} MX={I.grid.MX_ELEMENTS_PER_POINT} {meth} apply, {k} field(s): {a.elapsed_time(b) / 5:.3f} ms (weights + W.u)", flush=True)
