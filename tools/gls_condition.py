#!/usr/bin/env python3
"""Design aid (numpy, not product code): condition number of the GLS least-squares matrix M_v (gls.pyx:252-356) at the
interior nodes of a hexahedron mesh, per permeability kind -- the kappa that multiplies eps in "two Householder codes
agree to kappa * eps" (DESIGN.md section 2).  Dense assembly: rows = 8 cell rows [d_e | 1] + 3 rows per internal face
[-B_a | +B_b], B = [K N; T; tau U].

    python tools/gls_condition.py [edge] [jitter]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def node_matrix(p, G, perm, dmag):
    cells = G.esup[G.esup_ptr[p]:G.esup_ptr[p + 1]]
    faces = G.fsup[G.fsup_ptr[p]:G.fsup_ptr[p + 1]]
    loc = {int(c): i for i, c in enumerate(cells)}
    ne = len(cells)
    xv = G.point_coords[p]
    rows = []
    for e, c in enumerate(cells):
        r = np.zeros(3 * ne + 1)
        r[3 * e:3 * e + 3] = G.centroids[c] - xv
        r[-1] = 1.0
        rows.append(r)
    for f in faces:
        a, b = G.esuf_ptr[f], G.esuf_ptr[f + 1]
        if b - a != 2:
            continue
        ca, cb = int(G.esuf[a]), int(G.esuf[a + 1])
        N = G.normal_faces[f]
        T = xv - G.faces_centers[f]
        U = np.cross(N, T)
        eta = max(dmag[ca], dmag[cb], 0.0)
        tau = np.linalg.norm(U) ** (-eta)
        for k, (va, vb) in enumerate(((perm[ca].reshape(3, 3) @ N, perm[cb].reshape(3, 3) @ N), (T, T), (tau * U, tau * U))):
            r = np.zeros(3 * ne + 1)
            r[3 * loc[ca]:3 * loc[ca] + 3] = -va
            r[3 * loc[cb]:3 * loc[cb] + 3] = vb
            rows.append(r)
    return np.array(rows)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    jit = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
    import ninpol_oracle as O
    from ninpol_amd import mesh as M
    for kind in ("LIN", "ALH", "FAN"):
        m = M.hex_mesh(n, jitter=jit, seed=0)
        M.attach_fields(m, "u", perm=kind)
        o = O.OracleInterpolator("port", threads=4)
        o.load_mesh(m)
        G = o.grid
        v2i = o.variable_to_index
        perm = o.cells_data[v2i["cells"]["permeability"]][:G.n_elems * 9].reshape(-1, 9)
        dmag = o.cells_data[v2i["cells"]["diff_mag"]][:G.n_elems]
        ks = [np.linalg.cond(node_matrix(p, G, perm, dmag)) for p in range(G.n_points) if not G.boundary_points[p]]
        print(f"{kind}: hex {n}^3 jitter {jit}: cond(M_v) over {len(ks)} interior nodes: median {np.median(ks):.3g}, max {np.max(ks):.3g}")


if __name__ == "__main__":
    main()
