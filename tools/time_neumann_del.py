"""GLS on the 2 M-cell Delaunay mesh with an all-Dirichlet boundary and with the plane z = 0 flagged Neumann (its boundary nodes are then
computed: half a node's cells, boundary faces -- the small-node / block kernels): python tools/time_neumann_del.py [n]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ninpol_amd
from ninpol_amd import mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 54
for plane in (None, (2, 0.0)):
    m = M.delaunay_tet_mesh(n, seed=0); M.attach_fields(m, "u", perm="ALH", neumann_plane=plane)
    I = ninpol_amd.Interpolator(grid_build="device"); I.load_mesh(mesh_obj=m)
    plan = I.device_plan("u", "gls")
    out = torch.empty(plan.nnz, dtype=torch.float64, device="cuda"); nws = torch.empty(plan.n_points, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream()
    plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(5): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
    b.record(st); torch.cuda.synchronize()
    fl = I.grid.gls_plan_flops()
    print(f"Delaunay {n}: Neumann plane {plane}: {a.elapsed_time(b) / 5:.3f} ms; computed nodes per kernel { {k: v[2] for k, v in fl.items() if v[2]} }", flush=True)
    for k, name in enumerate(I.grid.PLAN_KERNELS):
        if fl[name][2] and name.startswith(("block", "small", "scratch")):
            os.environ["NIN_GLS_ONLY"] = str(k)
            plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream); torch.cuda.synchronize()
            a.record(st)
            for _ in range(3): plan.launch(out.data_ptr(), nws.data_ptr(), st.cuda_stream)
            b.record(st); torch.cuda.synchronize()
            del os.environ["NIN_GLS_ONLY"]
            print(f"    {name}: {a.elapsed_time(b) / 3:.3f} ms for {fl[name][2]} computed nodes")
