"""Time the three methods on a few mesh families (kernel-only, through DevicePlan)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
cases = {"tet40": lambda: M.tet_mesh(40, jitter=0.1), "wedge60": lambda: M.wedge_mesh(60, jitter=0.05),
         "mixed": lambda: M.mixed_mesh(100, 60, 60, jitter=0.1), "hex100": lambda: M.hex_mesh(100, jitter=0.15),
         "hex216": lambda: M.hex_mesh(216, jitter=0.15), "hex80m": lambda: M.hex_mesh(432, jitter=0.15),
         "mixed10m": lambda: M.mixed_mesh(200, 120, 120, jitter=0.1), "tet10m": lambda: M.tet_mesh(119, jitter=0.1),
         "del24": lambda: M.delaunay_tet_mesh(24, seed=0), "del40": lambda: M.delaunay_tet_mesh(40, seed=0),
         "del54": lambda: M.delaunay_tet_mesh(54, seed=0), "delr24": lambda: M.delaunay_tet_mesh(24, seed=0, lattice="random"),
         "delw100": lambda: M.delaunay_wedge_mesh(100, 60, seed=0), "delwr100": lambda: M.delaunay_wedge_mesh(100, 60, seed=0, lattice="random"),
         "delr40": lambda: M.delaunay_tet_mesh(40, seed=0, lattice="random"), "delr54": lambda: M.delaunay_tet_mesh(54, seed=0, lattice="random")}
for name in (sys.argv[1:] or [c for c in cases if not c.endswith(("10m", "80m", "216")) and not c.startswith("del")]):
    m = cases[name](); M.attach_fields(m, "u", perm="ALH")
    I = ninpol_amd.Interpolator(grid_build=os.environ.get("NIN_GRID_BUILD", "host")); t0 = time.time(); I.load_mesh(mesh_obj=m); print(f"{name}: load_mesh {time.time() - t0:.2f} s")
    st = torch.cuda.current_stream()
    for meth in os.environ.get("NIN_METHODS", "idw,ls,gls").split(","):
        plan = I.device_plan("u", meth)
        out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.empty(plan.n_points, dtype=torch.float64, device="cuda")
        plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
        b.record(st); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        if meth == "gls": print(f"{name}: plan {I.grid.gls_plan()}")
        print(f"{name}: E={I.grid.n_elems} P={I.grid.n_points} MX={I.grid.MX_ELEMENTS_PER_POINT}/{I.grid.MX_FACES_PER_POINT} {meth}: {ms:.3f} ms = {I.grid.n_points/ms/1e3:.2f} Mnodes/s")
