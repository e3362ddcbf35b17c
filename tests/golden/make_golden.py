#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref = the reference's own Grid,
IDW, LS and GLS compiled in the dev container by oracle/build_ref.py).

Run once in the dev container (`python tests/golden/make_golden.py`); the .npz files are data --
inputs (points, cell blocks, fields) and the reference's outputs (every Grid array, and per method
the dense weight table and neumann_ws written by the reference's prepare()) -- and are committed;
the reference code never is.  `cells_data` / `points_data` and the CSR triplets come from the L3
glue restated in oracle/ninpol_oracle.py (Interpolator itself needs meshio, absent here); that glue
is pinned separately by the reference's published accuracy table (tests/test_kat.py).  Set OPENBLAS_NUM_THREADS=1 as the reference asks (gls.pyx:63-70).
"""
import os
import sys

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

import ninpol_oracle as O  # noqa: E402
from ninpol_amd import mesh as M  # noqa: E402

CASES = {
    # name: (generator, kwargs, permeability kind, neumann plane)
    "hex4_uniform": (M.hex_mesh, dict(nx=4), "ALH", (2, 0.0)),
    "hex543_jitter": (M.hex_mesh, dict(nx=5, ny=4, nz=3, jitter=0.2, seed=3), "ALH", (2, 0.0)),
    "hex6_lin_dirichlet": (M.hex_mesh, dict(nx=6, jitter=0.15, seed=5), "LIN", None),
    "tet3_jitter": (M.tet_mesh, dict(nx=3, jitter=0.1, seed=1), "ALH", (0, 0.0)),
    "wedge3_jitter": (M.wedge_mesh, dict(nx=3, jitter=0.1, seed=2), "ALH", (1, 1.0)),
    "mixed533_jitter": (M.mixed_mesh, dict(nx=5, ny=3, nz=3, n_hex=2, jitter=0.1, seed=4), "ALH", (2, 0.0)),
    # round 3: the reference's own FAN tensor (tests/utils/analytical.py:285-293, cond(K) = 3e3, cond(M_v) = 7e4 here) with
    # a Neumann plane; and BASELINE config [0]'s mesh exactly -- the Kuhn split of 6 x 6 x 5 hexahedra (E = 1 080, P = 294,
    # SURVEY 8d), all-Dirichlet boundary
    "hex8_fan": (M.hex_mesh, dict(nx=8, jitter=0.1, seed=6), "FAN", (2, 0.0)),
    "tet665": (M.tet_mesh, dict(nx=6, ny=6, nz=5), "ALH", None),
}


def build_case(name):
    gen, kw, perm, plane = CASES[name]
    m = gen(**kw)
    M.attach_fields(m, "u", perm=perm, neumann_plane=plane, seed=11)
    return m


def main():
    assert O.have_reference(), "build oracle/_ref first (python oracle/build_ref.py)"
    only = sys.argv[1:]          # `make_golden.py hex8_fan tet665` regenerates just those
    for name in CASES:
        if only and name not in only:
            continue
        m = build_case(name)
        ref = O.OracleInterpolator("reference")
        ref.load_mesh(m)
        out = {"points": m.points}
        for b, blk in enumerate(m.cells):
            out[f"block{b}_type"] = np.array(blk.type)
            out[f"block{b}_data"] = blk.data
        out["n_blocks"] = np.array(len(m.cells))
        for b in range(len(m.cells)):
            out[f"permeability_block{b}"] = m.cell_data["permeability"][b]
            out[f"u_block{b}"] = m.cell_data["u"][b]
        out["neumann_flag_u"] = m.point_data["neumann_flag_u"]
        out["neumann_u"] = m.point_data["neumann_u"]
        for k in O._ARRAY_NAMES:
            out["grid_" + k] = getattr(ref.grid, k)
        for k in O._SCALAR_NAMES:
            out["grid_" + k] = np.array(getattr(ref.grid, k))
        out["cells_data"] = ref.cells_data
        out["points_data"] = ref.points_data
        out["cells_vars"] = np.array(list(ref.variable_to_index["cells"].keys()))
        out["points_vars"] = np.array(list(ref.variable_to_index["points"].keys()))
        for meth in ("idw", "ls", "gls"):
            w, nw = ref.prepare(meth, "u")
            W, _ = ref.interpolate("u", meth)
            out[f"{meth}_weights"] = w
            out[f"{meth}_neumann_ws"] = nw
            out[f"{meth}_indptr"] = W.indptr
            out[f"{meth}_indices"] = W.indices
            out[f"{meth}_data"] = W.data
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "P", ref.grid.n_points, "E", ref.grid.n_elems, "F", ref.grid.n_faces,
              os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
