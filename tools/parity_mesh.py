"""GLS / IDW / LS parity of the HIP path against the oracle on one generated mesh (GPU box): tools/parity_mesh.py del12 [plane]"""
import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np
import ninpol_amd, ninpol_oracle, util
from ninpol_amd import mesh as M
cases = {"del8": lambda: M.delaunay_tet_mesh(8, seed=1), "del12": lambda: M.delaunay_tet_mesh(12, seed=2),
         "del16": lambda: M.delaunay_tet_mesh(16, seed=3), "delr10": lambda: M.delaunay_tet_mesh(10, seed=4, lattice="random"),
         "delw12": lambda: M.delaunay_wedge_mesh(12, 8, seed=2), "delwr12": lambda: M.delaunay_wedge_mesh(12, 8, seed=3, lattice="random"),
         "delr24": lambda: M.delaunay_tet_mesh(24, seed=0, lattice="random"), "delr30": lambda: M.delaunay_tet_mesh(30, seed=3, lattice="random"),
         "tet8": lambda: M.tet_mesh(8, jitter=0.1, seed=1), "mixed": lambda: M.mixed_mesh(10, 6, 6, jitter=0.1, seed=1)}
ninpol_oracle.build_port()
for name in sys.argv[1:] or ["del8", "del12", "delr10"]:
    for plane, perm in ((None, "ALH"), ((2, 0.0), "LIN"), ((0, 1.0), "FAN")):
        m = cases[name](); M.attach_fields(m, "u", perm=perm, neumann_plane=plane, seed=3)
        I = ninpol_amd.Interpolator(); I.load_mesh(mesh_obj=m)
        o = ninpol_oracle.OracleInterpolator("port", threads=8); o.load_mesh(m)
        for meth in ("gls", "idw", "ls"):
            W, nws = I.interpolate("u", meth); Wo, nwo = o.interpolate("u", meth)
            err = util.csr_rowscaled_err(W, Wo.indptr, Wo.indices, Wo.data)
            print(f"{name} plane={plane} perm={perm} {meth}: nnz={W.nnz} err={err:.2e} nws={util.rowscaled_err(nws, nwo):.1e}" + (f" plan={I.grid.gls_plan()}" if meth == "gls" else ""), flush=True)
